"""CPU: the host stage of `seeksv somatic` (normal-cluster look-ups for every tumor junction row), through the CLI's -J dump hook (no GPU, no
BAM pass), against the first 25 columns of the real reference's output on the 227-row branch-coverage table (tests/golden/somatic)."""
import gzip
import os
import subprocess

import numpy as np

import golden_util as G
import oracle_lib
from seeksv_amd import host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEKSV = os.environ.get("SSV_CLI") or os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
SYNTH_FULL = dict(genome_frac=1 / 8192, depth=40, n_sv=24)


def test_somatic_lookups_match_reference(tmp_path):
    if not os.path.exists(SEEKSV):
        subprocess.check_call(["make", "-C", ROOT, "cli"], stdout=subprocess.DEVNULL)
    # the normal's clip.gz: oracle getclip on the regenerated synthetic sample (the reference's getclip output equals it, test_oracle_golden)
    w = synth.Workload(**SYNTH_FULL)
    b = w.generate_host(0, w.n_total)
    table = oracle_lib.getclip([b], 0.9, 1, False)
    rows, _ = host.format_clip_outputs(table, w.names)
    clip = str(tmp_path / "normal.clip.gz")
    with gzip.open(clip, "wt") as f:
        f.write(rows)
    for tag, flags in (("", []), (".l0", ["-l", "0"]), (".l60", ["-l", "60"]), (".t05", ["-t", "0.5"]), (".m60", ["-m", "60"])):
        dump = str(tmp_path / f"dump{tag}.txt")
        r = subprocess.run([SEEKSV, "somatic", "-J", dump] + flags + ["unused.bam", clip, os.path.join(G.GOLDEN, "somatic", "tumor.sv"), "unused.out"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = [l.split("\t") for l in open(dump).read().splitlines()]
        exp = [l.split("\t") for l in G.read_text("somatic", f"somatic{tag}.sv").splitlines()]
        assert len(got) == len(exp) == 199
        assert got[0] == exp[0]
        for g, e in zip(got[1:], exp[1:]):
            assert g == e[:25], (tag, g[:9], g[23:], e[23:])
        msgs = [l for l in G.read_text("somatic", f"somatic{tag}.stderr").splitlines() if not l.startswith(("Bam/sam", "Mean deviation"))]
        assert r.stderr.splitlines() == msgs
