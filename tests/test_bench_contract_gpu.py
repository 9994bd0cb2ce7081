"""-m gpu: bench.py at a small size prints ONE JSON line that keeps the driver's contract (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, roofline, cpu_baseline), and the two-rank code path
gives the same result as the one-rank path on the same total workload (two ranks on one GPU over gloo: the test hook of bench.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-400:]
    return json.loads(lines[0])


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--genome-frac", "0.00390625", "--n-sv", "200", "--steps", "3", "--warmup", "1", "--cpu-sample", "300000", "--ref-sample", "0", "--config5-frac", "0.0009765625"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-600:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "records/s" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["records_total"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "records/s" and cb["value"] > 0 and cb["sample"]
    assert d["result"]["support_sum"] == d["result"]["n_events"] > 0
    # the file leg (the same sample written as a BAM, inflated and decoded on the GPU) must report the counts of the resident leg
    fp = d.get("file_path", {})
    assert "error" not in fp, fp
    if "same_result_as_resident_path" in fp:
        assert fp["same_result_as_resident_path"] is True, (fp["result"], d["result"])
    # the file is read inside the leg's timed region; round 3's footing (the file in pinned memory beforehand) is reported beside it; the
    # resident leg says where its counter bytes come from
    assert fp["includes_file_read"] is True and fp["from_pinned"]["includes_file_read"] is False and fp["from_pinned"]["value"] > 0 and fp["two_reads"]["value"] > 0
    # the PCIe-inclusive leg (host batches through the staging path) and BASELINE config 3's shape are in the same line, and say what they are
    hb = d.get("host_batch_path", {})
    assert "error" not in hb, hb
    assert hb["clip_scan_pinned_prefetch"]["records_per_s"] > 0 and hb["clip_scan_pageable"]["events"] == hb["clip_scan_pinned"]["events"] > 0 and "not `value`" in hb["what"]
    c3 = d.get("config3_path", {})
    assert "error" not in c3, c3
    assert c3["value"] > 0 and c3["device_kernels_ms"] > 0 and c3["records"] > 0
    # BASELINE config 5 through the product (tumor + normal on a human + HBV reference, pairs with an unmapped end): every planted integration is called, and called somatic
    c5 = d.get("config5_path", {})
    assert "error" not in c5, c5
    assert c5["viral_planted"] > 0 and c5["viral_somatic"] == c5["viral_found"] >= 10, c5   # (fifty breakpoints on a 3.2 kb contig: neighbours 60 bp apart take reads from each other)
    assert c5["germline_called_somatic"] * 10 <= c5["germline_found"] and c5["normal_unmapped_pairs_written"] > 0 and c5["somatic_s"] > 0 and c5["normal_cluster_rows"] > 0
    assert "traffic_source" in rf and ("profiles/traffic.json" in rf["traffic_source"])
    assert (rf["traffic"] is None) == rf["traffic_source"].startswith("none")


def test_bench_under_torchrun_with_one_rank_is_the_plain_line():
    """the driver's scaling run starts N = 1 through torch.distributed.run too: same footing as the plain `python bench.py` line (same workload, same legs
    beside the headline, same counts) - SCALE's N = 1 must agree with BENCH"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    common = ["--genome-frac", "0.00390625", "--n-sv", "200", "--steps", "2", "--warmup", "1", "--cpu-sample", "200000", "--ref-sample", "0"]
    r1 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                         os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-800:]
    a = _line(r1.stdout)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-600:]
    b = _line(r2.stdout)
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["scaling"] == b["scaling"] and a["config"]["workload"] == b["config"]["workload"]
    assert a["config"]["records_total"] == b["config"]["records_total"] and a["result"] == b["result"]
    assert set(a) == set(b) and "file_path" in a and "cpu_baseline" in a
    assert a["file_path"]["result"] == b["file_path"]["result"]


def test_bench_two_ranks_equal_one():
    """`python bench.py --gpus 2` with no RANK in the environment starts its two ranks itself (torch.distributed.run as a child process, before the parent
    touches the GPU) and relays rank 0's line - what the driver's `--gpus N` command does on an 8-GPU node; same counts as one rank on the same total
    workload; ranks_path (the PRODUCT's `-N 2` on a file, its own exchange) is in the line and equals the product's one-rank outputs"""
    env = dict(os.environ, SSV_FORCE_DEVICE="0", SSV_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genome-frac", "0.00390625", "--n-sv", "200", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-800:]
    a = _line(r2.stdout)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--genome-frac", "0.00390625", "--n-sv", "200", "--depth", "60", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--config5-frac", "0"],
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-600:]
    b = _line(r1.stdout)
    assert a["n_gpus"] == 2 and a["config"]["records_total"] == b["config"]["records_total"]
    assert a["result"] == b["result"]
    rp = a.get("ranks_path", {})
    assert "error" not in rp, rp
    assert rp["same_outputs_as_one_rank"] is True and rp["ranks"]["ranks"] == 2 and rp["one_rank"]["ranks"] == 1 and rp["value"] > 0
    assert "exchange over" in (rp["ranks"]["exchange"] or ""), rp["ranks"]
    assert "ranks_path" in a["config"]["multi_gpu"]


def test_bench_eight_forced_ranks_strong_scaling_dry_run():
    """the 8-GPU day's command, `python bench.py --gpus 8 --scaling strong`, as a dry run: eight ranks forced onto the one GPU of the box (gloo), a small
    sample split eight ways: one line, n_gpus 8, strong"""
    env = dict(os.environ, SSV_FORCE_DEVICE="0", SSV_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--scaling", "strong", "--strong-depth", "60", "--genome-frac", "0.00390625", "--n-sv", "200", "--steps", "2", "--warmup", "1",
                        "--ranks-frac", "0.5"], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    a = _line(r.stdout)
    assert a["n_gpus"] == 8 and a["scaling"] == "strong" and a["config"]["records_per_gpu"] * 7 < a["config"]["records_total"]
    assert a["ranks_path"]["same_outputs_as_one_rank"] is True and a["ranks_path"]["ranks"]["ranks"] == 8


def test_bench_strong_scaling_two_ranks_equal_one():
    """--scaling strong (BASELINE config 4: ONE sample split N ways): two ranks on the same sample as one rank give the same counts, the line
    says `strong`, and a share that cannot fit a GPU is refused with a message instead of an out-of-memory failure"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SSV_FORCE_DEVICE="0", SSV_DIST_BACKEND="gloo")
    common = ["--genome-frac", "0.00390625", "--n-sv", "200", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--strong-depth", "60", "--file-frac", "0", "--no-cpu-baseline"]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                         os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-800:]
    a = _line(r2.stdout)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-600:]
    b = _line(r1.stdout)
    assert a["scaling"] == b["scaling"] == "strong" and a["n_gpus"] == 2
    assert a["config"]["records_total"] == b["config"]["records_total"] and a["config"]["records_per_gpu"] < b["config"]["records_per_gpu"]
    assert a["result"] == b["result"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scaling", "strong", "--file-frac", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not fit one GPU" in r.stderr
