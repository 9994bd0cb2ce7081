"""-m gpu: the HIP path (through the C ABI of libseeksv_hip.so) against the reference's own outputs
(tests/golden/) and against the CPU oracle on the same inputs.  Bit-exact: this is integer / byte work;
the two fp64 match-rate compares are done with the same IEEE division on both sides."""
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from seeksv_amd import host
from test_oracle_golden import GETCLIP_CASES, GETSV_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from seeksv_amd.device import Context
    c = Context(0)
    yield c
    c.close()


TABLE_KEYS = ("tid", "pos", "side", "support", "left_len", "right_len", "qual_missing", "str_off", "str", "cigar_off", "n_cigar", "cigar")


def assert_tables_equal(a, b):
    assert a["n_clusters"] == b["n_clusters"] and a["n_events"] == b["n_events"]
    for k in TABLE_KEYS:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("sub,bam,prefix,kw", GETCLIP_CASES, ids=[c[2] for c in GETCLIP_CASES])
@pytest.mark.parametrize("batch_records", [1 << 20, 1000])
def test_getclip_hip_matches_reference(ctx, sub, bam, prefix, kw, batch_records):
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, sub, bam), batch_records)
    d = ctx.getclip(batches, **kw)
    clip, fq = host.format_clip_outputs(d, names)
    assert clip == G.read_text(sub, prefix + ".clip.txt")
    assert fq == G.read_text(sub, prefix + ".clip.fq.txt")
    assert_tables_equal(d, O.getclip(batches, **kw))


@pytest.mark.parametrize("sample", ["cancer", "normal"])
def test_isize_hip_matches_reference_example(ctx, sample):
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, "example", sample + ".sort.bam"), 5000)
    rc, n, mean, sd = ctx.isize_stats(batches, 20, 5000000)
    assert rc == 0 and [str(mean), str(sd)] == G.read_text("example", sample + ".isize.txt").split()
    assert (rc, n, mean, sd) == O.isize_stats(batches, 20, 5000000)
    # the cut-off after max_pairs qualifying records (cluster.cpp:68)
    for cap in (1, 1000, 4097):
        assert ctx.isize_stats(batches, 20, cap) == O.isize_stats(batches, 20, cap)


@pytest.mark.parametrize("case,prefix,kw", GETSV_CASES, ids=[c[1] for c in GETSV_CASES])
@pytest.mark.parametrize("batch_records", [1 << 20, 3000])
def test_getsv_passes_hip_matches_reference(ctx, case, prefix, kw, batch_records):
    base = os.path.join(G.GOLDEN, "getsv")
    rows = G.read_junction_file(os.path.join(base, case + ".junctions.txt"))
    stats, junctions, folded = G.run_getsv_case(os.path.join(base, case + ".bam"), rows, ctx, batch_records=batch_records, **kw)
    assert [str(stats[2]), str(stats[3])] == G.read_text("getsv", prefix + ".isize.txt").split()
    golden = G.parse_sv_outputs(os.path.join(base, prefix + ".sv"), os.path.join(base, prefix + ".stdout"))
    assert G.check_getsv_against_golden(junctions, folded, golden) > 3 * len(junctions)


def test_no_seq_and_empty_batches(ctx):
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, "getclip", "filters.bam"))
    empty = {k: v[:0] for k, v in batches[0].items() if isinstance(v, np.ndarray)}
    d = ctx.getclip([empty, batches[0], empty])
    assert_tables_equal(d, O.getclip([batches[0]]))
    d0 = ctx.getclip([empty])
    assert d0["n_clusters"] == 0 and d0["n_events"] == 0
