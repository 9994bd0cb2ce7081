"""-m gpu: host batches in page-locked memory announced one ahead (ssv_batch_prefetch: the copy of batch k+1 runs on the upload stream while
batch k is scanned) give the tables / tallies of the same batches handed over one by one; the order rule is enforced."""
import numpy as np
import pytest

import oracle_lib as O
from seeksv_amd import _abi, host, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from seeksv_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _cut(hb, i0, i1):
    """records [i0, i1) of a host batch as its own batch (shared variable parts, absolute offsets)"""
    out = {}
    for k, v in hb.items():
        out[k] = v[i0:i1] if isinstance(v, np.ndarray) and k not in ("cigar", "seqqual") else v
    return out


@pytest.mark.parametrize("with_rec", [False, True])
def test_prefetched_batches_equal_plain(ctx, with_rec):
    from seeksv_amd.device import PinnedArrays
    w = synth.Workload(genome_frac=1 / 512, depth=30, n_sv=50)
    hb = w.generate_host(0, w.n_total, with_rec=with_rec) if with_rec else w.generate_host(0, w.n_total)
    cuts = [0, w.n_total // 5, w.n_total // 2 + 3, w.n_total - 1000, w.n_total]
    want = ctx.getclip([_cut(hb, cuts[i], cuts[i + 1]) for i in range(4)])
    o = O.getclip([hb])
    assert want["n_clusters"] == o["n_clusters"] > 1000
    with PinnedArrays() as pin:
        parts = [pin.batch(_cut(hb, cuts[i], cuts[i + 1])) for i in range(4)]
        for p in parts:  # (the variable parts are shared by the four cuts: one pinned copy each is fine, the offsets are absolute)
            assert p["tid"].ctypes.data % 64 == 0
        batches = [_abi.make_batch(p) for p in parts]
        ctx.clip_begin()
        ctx.prefetch(batches[0][0])
        for k in range(4):
            if k + 1 < 4:
                ctx.prefetch(batches[k + 1][0])
            ctx.clip_scan(batches[k][0])
        got = ctx.clip_cluster()
        for k in ("tid", "pos", "side", "support", "left_len", "right_len", "str", "cigar", "n_cigar"):
            assert np.array_equal(got[k], want[k]), k
        # getsv: discordant pairs + depth through the same pipeline
        hdr = host.Header(w.names, w.lens)
        stats = O.isize_stats([hb], 20, 5000000)
        plan = host.Plan(hdr, w.junctions, stats[2], stats[3])
        c0, r0, p0 = ctx.discordant_and_depth([hb], plan, stats[2], stats[3], 20, hdr.target_lens)
        ctx.getsv_begin(plan.junctions, plan.windows, stats[2], stats[3], hdr.target_lens)
        ctx.prefetch(batches[0][0])
        for k in range(4):
            if k + 1 < 4:
                ctx.prefetch(batches[k + 1][0])
            ctx.getsv_scan(batches[k][0])
        c1, r1, p1 = ctx.getsv_finish(plan.ranges, plan.points)[:3]
        assert np.array_equal(c0, c1) and np.array_equal(r0, r1) and np.array_equal(p0, p1) and c0.sum() > 0
        plan.close()
        hdr.close()


def test_prefetch_order_is_enforced(ctx):
    from seeksv_amd.device import SeeksvError
    w = synth.Workload(genome_frac=1 / 8192, depth=30, n_sv=5)
    a = w.generate_host(0, w.n_total // 2)
    b = w.generate_host(w.n_total // 2, w.n_total - w.n_total // 2)
    ba, bb = _abi.make_batch(a), _abi.make_batch(b)
    ctx.clip_begin()
    ctx.prefetch(ba[0])
    with pytest.raises(SeeksvError, match="prefetched batch is pending"):
        ctx.clip_scan(bb[0])
    ctx.prefetch(bb[0])
    with pytest.raises(SeeksvError, match="two prefetched batches"):
        ctx.prefetch(ba[0])
    ctx.clip_scan(ba[0])
    ctx.clip_scan(bb[0])
    n2 = ctx.clip_cluster()["n_clusters"]
    assert n2 == ctx.getclip([a, b])["n_clusters"]
    # an announcement belongs to the stream of batches, not to a pass: it survives ssv_clip_begin (a driver ends a pass in the middle of a
    # batch while the next batch is on its way, seeksv_cli.cpp:getclip_single) ...
    ctx.clip_begin()
    ctx.prefetch(ba[0])
    ctx.clip_begin()
    with pytest.raises(SeeksvError, match="prefetched batch is pending"):
        ctx.clip_scan(bb[0])
    ctx.prefetch(bb[0])
    ctx.clip_scan_range(ba[0], 0, 100)      # a leading part: the batch stays announced
    ctx.clip_begin()
    ctx.clip_scan_range(ba[0], 100, ba[0].n)
    ctx.clip_scan(bb[0])
    ctx.clip_cluster()
    # ... until the caller gives the stream up
    ctx.clip_begin()
    ctx.prefetch(ba[0])
    ctx.prefetch_drop()
    assert ctx.getclip([a, b])["n_clusters"] == n2
    # ... or a new reading of the file starts (insert sizes, the fused getsv pass)
    ctx.prefetch(ba[0])
    assert ctx.isize_stats([a, b], 20, 5000000)[1] > 0
    dev, keep = w.generate_device(0, 1000, 0)
    with pytest.raises(SeeksvError, match="host batches"):
        ctx.prefetch(dev)
