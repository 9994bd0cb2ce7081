"""-m gpu: the C ABI's error behaviour (include/seeksv_hip.h: every call returns a status, nothing throws or exits, the message is in
ssv_last_error): calls out of sequence, null / malformed arguments, inputs the path refuses; the context stays usable afterwards."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_util as G
from seeksv_amd import _abi, host

pytestmark = pytest.mark.gpu
E_ARG, E_STATE, E_RANGE = -3, -4, -6


@pytest.fixture(scope="module")
def ctx():
    from seeksv_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _err(ctx):
    return ctx._lib.ssv_last_error(ctx._h).decode()


def test_calls_out_of_sequence(ctx):
    lib, h = ctx._lib, ctx._h
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, "getclip", "filters.bam"))
    b, keep = _abi.make_batch(batches[0])
    fresh = type(ctx)(0)
    try:
        assert lib.ssv_clip_scan(fresh._h, C.byref(b)) == E_STATE and "ssv_clip_begin" in _err(fresh)
        t = _abi.HipClusterTable()
        assert lib.ssv_clip_cluster(fresh._h, C.byref(t)) == E_STATE
        assert lib.ssv_getsv_scan(fresh._h, C.byref(b)) == E_STATE and "ssv_getsv_begin" in _err(fresh)
        md = C.c_int32()
        assert lib.ssv_getsv_finish(fresh._h, None, None, 0, None, None, 0, None, C.byref(md)) == E_STATE
        hits = np.zeros(1, dtype=np.dtype(_abi.REALIGN_HIT))
        off = np.array([0, 4], np.uint64)
        assert lib.ssv_realign_query(fresh._h, C.c_char_p(b"ACGT"), off.ctypes.data, 1, hits.ctypes.data) == E_STATE and "ssv_realign_index" in _err(fresh)
    finally:
        fresh.close()
    # the shared context still works
    assert ctx.getclip(batches)["n_clusters"] > 0


def test_bad_arguments(ctx):
    lib, h = ctx._lib, ctx._h
    assert lib.ssv_clip_begin(h, None) == E_ARG
    assert lib.ssv_clip_table_format(h, 7) == E_ARG
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, "getclip", "filters.bam"))
    ctx.clip_begin()
    assert lib.ssv_clip_scan(h, None) == E_ARG
    bad = dict(batches[0]); bad["cigar_off"] = None
    b, keep = _abi.make_batch(bad)
    assert lib.ssv_clip_scan(h, C.byref(b)) == E_ARG and "null" in _err(ctx)
    b2, keep2 = _abi.make_batch(batches[0])
    b2.n = -1
    assert lib.ssv_clip_scan(h, C.byref(b2)) == E_ARG
    b2.n = len(batches[0]["tid"]); b2.mem = 5
    assert lib.ssv_clip_scan(h, C.byref(b2)) == E_ARG and "mem" in _err(ctx)
    # a device batch must be 16-byte aligned
    import torch
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.uint8).copy()).cuda() for k, v in batches[0].items() if isinstance(v, np.ndarray) and v.size}
    arrays = {k: dev[k].data_ptr() for k in dev}
    arrays["tid"] += 4
    arrays.update(n_cigar_total=len(batches[0]["cigar"]), seqqual_bytes=len(batches[0]["seqqual"]), max_ref_span=batches[0]["max_ref_span"], xc=None)
    b3, keep3 = _abi.make_batch(arrays, mem=_abi.MEM_DEVICE, n=len(batches[0]["tid"]) - 1)
    assert lib.ssv_clip_scan(h, C.byref(b3)) == E_ARG and "aligned" in _err(ctx)
    # realign: offsets that do not describe the reference
    words = np.zeros(4, np.uint64)
    off = np.array([0, 50, 90], np.int64)
    assert lib.ssv_realign_index(h, words.ctypes.data, 0, 100, off.ctypes.data, 2, None) == E_ARG
    assert ctx.getclip(batches)["n_clusters"] > 0


def test_getsv_begin_refuses_unsorted_windows(ctx):
    lens = np.array([10000], np.int32)
    w = np.zeros(2, dtype=_abi.INTERVAL_DTYPE)
    w["tid"] = [0, 0]; w["beg"] = [500, 400]; w["end"] = [600, 450]
    p = _abi.GetsvParams()
    p.windows = w.ctypes.data; p.n_windows = 2; p.n_targets = 1; p.target_len = lens.ctypes.data; p.times = 4
    assert ctx._lib.ssv_getsv_begin(ctx._h, C.byref(p)) == E_ARG and "sorted" in _err(ctx)
