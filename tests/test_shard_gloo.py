"""Multi-rank path on CPU: two gloo ranks, range-partitioned by reference interval with a halo, the oracle as backend.
The merged result must equal the single-rank result (this is what bench.py --gpus N does with the HIP backend and RCCL)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

import oracle_lib as O
from seeksv_amd import host, shard, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_one(tmp_path):
    out = str(tmp_path / "merged.json")
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_shard_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    m = json.load(open(out))

    import _shard_worker as W
    w = synth.Workload(genome_frac=1 / 4096, depth=60, n_sv=30)
    b = w.generate_host(0, w.n_total)
    d = O.getclip([b])
    rc, n, mean, sd = O.isize_stats([w.generate_host(0, min(w.n_total, 200000))], 20, 100000)
    hdr = host.Header(w.names, w.lens)
    plan = host.Plan(hdr, w.junctions, mean, sd)
    counts = O.discordant([b], plan.junctions, mean, sd, 4, 20)
    rs, pd, _ = O.depth([b], plan.windows, plan.ranges, plan.points, 20)
    assert (m["mean"], m["sd"]) == (mean, sd)
    assert m["n_events"] == d["n_events"] == m["support_sum"] and m["n_clusters"] == d["n_clusters"]
    assert min(m["per_rank_events"]) > 0, "both ranks must own clip events"
    assert m["counts"] == counts.tolist() and m["rs"] == [int(x) for x in rs] and m["pd"] == pd.tolist()
    assert sum(m["counts"]) > 0
    single = sorted(W.cluster_rows(d), key=lambda r: r[:3])
    merged = sorted([tuple(r) for r in m["rows"]], key=lambda r: r[:3])
    assert merged == single


def test_shard_plan_covers_everything_once():
    w = synth.Workload(genome_frac=1 / 4096, depth=30, n_sv=10)
    for world in (1, 2, 3, 8):
        plans = [shard.shard_plan(w, r, world) for r in range(world)]
        assert plans[0]["own_lo_rec"] == 0 and plans[-1]["own_hi_rec"] == w.n_total
        for a, b in zip(plans, plans[1:]):
            assert a["own_hi_rec"] == b["own_lo_rec"] and a["own"][1] == b["own"][0]
            assert b["scan_lo_rec"] < b["own_lo_rec"] and b["scan_lo_rec"] % 8 == 0 and b["own_lo_rec"] % 8 == 0
