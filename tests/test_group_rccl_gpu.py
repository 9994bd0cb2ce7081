"""-m gpu: the one exchange step of a range-partitioned run (ssv_group_*, include/seeksv_hip.h) over REAL devices: with two or more GPUs the
group is an RCCL communicator (ncclCommInitAll) and the ranks' vectors meet in ncclAllGather over xGMI; on a one-GPU box the group is ONE
rank under SSV_GROUP_RCCL_ONE=1, which is still an RCCL communicator: dlopen, ncclCommInitAll, ncclAllGather, the polled stream, the copy
back, ncclCommAbort and ncclCommDestroy all run (tests/test_cli_ranks_gpu.py covers the host-memory exchange of ranks that share a GPU)."""
import ctypes as C
import threading

import numpy as np
import pytest

from seeksv_amd import _abi

pytestmark = pytest.mark.gpu


def _group(lib, devices):
    ctxs = []
    for d in devices:
        h = C.c_void_p()
        assert lib.ssv_ctx_create(d, C.byref(h)) == 0, lib.ssv_last_error(None)
        ctxs.append(h)
    arr = (C.c_void_p * len(ctxs))(*[h.value for h in ctxs])
    g = C.c_void_p()
    lib.ssv_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
    lib.ssv_group_allgather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.ssv_group_destroy.argtypes = [C.c_void_p]
    lib.ssv_group_uses_rccl.argtypes = [C.c_void_p]
    assert lib.ssv_group_create(arr, len(ctxs), C.byref(g)) == 0, lib.ssv_last_error(None)
    return ctxs, g


@pytest.mark.parametrize("n_bytes", [0, 8, 4096, 3 * 1024 * 1024 + 5])
def test_allgather_over_rccl(n_bytes, monkeypatch):
    lib = _abi.hip_lib()
    n_dev = lib.ssv_device_count()
    if n_dev < 2:
        monkeypatch.setenv("SSV_GROUP_RCCL_ONE", "1")   # RCCL cannot put two ranks on one device: a communicator of one
    n = min(n_dev, 8)
    ctxs, g = _group(lib, list(range(n)))
    try:
        assert lib.ssv_group_uses_rccl(g) == 1
        rng = np.random.default_rng(5)
        send = [rng.integers(0, 256, n_bytes, dtype=np.uint8) for _ in range(n)]
        recv = [np.zeros(n_bytes * n, dtype=np.uint8) for _ in range(n)]
        rcs = [None] * n

        def rank(r):
            rcs[r] = lib.ssv_group_allgather(g, r, send[r].ctypes.data if n_bytes else None, n_bytes, recv[r].ctypes.data if n_bytes else None)
        for rep in range(3):   # the communicator is reused
            th = [threading.Thread(target=rank, args=(r,)) for r in range(n)]
            [t.start() for t in th]
            [t.join(timeout=120) for t in th]
            assert all(not t.is_alive() for t in th), "a rank hangs in the exchange"
            assert rcs == [0] * n
            want = np.concatenate(send) if n_bytes else np.zeros(0, np.uint8)
            for r in range(n):
                assert np.array_equal(recv[r], want)
    finally:
        lib.ssv_group_destroy(g)
        for h in ctxs:
            lib.ssv_ctx_destroy(h)


def _run_ranks(lib, g, n, sizes, skip=()):
    """every rank's ssv_group_allgather on a thread of its own; returns (return codes, received arrays, hung?)"""
    send = [np.full(max(1, sizes[r]), r + 1, dtype=np.uint8) for r in range(n)]
    recv = [np.zeros(max(1, max(sizes) * n), dtype=np.uint8) for _ in range(n)]
    rcs = [None] * n

    def rank(r):
        rcs[r] = lib.ssv_group_allgather(g, r, send[r].ctypes.data, sizes[r], recv[r].ctypes.data)
    th = [threading.Thread(target=rank, args=(r,)) for r in range(n) if r not in skip]
    [t.start() for t in th]
    [t.join(timeout=60) for t in th]
    return rcs, recv, any(t.is_alive() for t in th)


def drop_group(lib, ctxs, g):
    lib.ssv_group_destroy(g)
    for h in ctxs:
        lib.ssv_ctx_destroy(h)


def _one_rank_rccl_failures(lib, fresh, drop, E_HIP, E_STATE):
    """what a communicator of one rank can show of the failure paths: the injected failures of every stage, ncclCommAbort on a live
    communicator (before and behind an issued collective), the broken group's answer afterwards, and a working exchange in between"""
    # a failure before the exchange: reported, nothing aborted, the group works afterwards
    ctxs, g = fresh(SSV_GROUP_FAIL="0:1")
    assert lib.ssv_group_uses_rccl(g) == 1
    rcs, _, hung = _run_ranks(lib, g, 1, [512])
    assert not hung and rcs == [E_HIP] and "injected" in _err(lib, ctxs[0])
    drop(lib, ctxs, g)
    # the collective's call fails: the communicator is aborted (ncclCommAbort), the group only answers SSV_E_STATE from then on
    ctxs, g = fresh(SSV_GROUP_FAIL="0:2", SSV_GROUP_TIMEOUT_S="20")
    rcs, _, hung = _run_ranks(lib, g, 1, [512])
    assert not hung and rcs == [E_HIP] and "ncclAllGather" in _err(lib, ctxs[0])
    rcs, _, hung = _run_ranks(lib, g, 1, [512])
    assert not hung and rcs == [E_STATE] and "aborted" in _err(lib, ctxs[0])
    drop(lib, ctxs, g)
    # the collective is issued for real and the communicator aborted behind it
    ctxs, g = fresh(SSV_GROUP_FAIL="0:3", SSV_GROUP_TIMEOUT_S="20")
    rcs, _, hung = _run_ranks(lib, g, 1, [3 * 1024 * 1024])
    assert not hung and rcs == [E_STATE] and "aborted" in _err(lib, ctxs[0])
    rcs, _, hung = _run_ranks(lib, g, 1, [512])
    assert not hung and rcs == [E_STATE]
    drop(lib, ctxs, g)
    # and an undisturbed group: several exchanges of different sizes over one communicator
    ctxs, g = fresh()
    for size in (512, 4096, 1, 100000):
        rcs, recv, hung = _run_ranks(lib, g, 1, [size])
        assert not hung and rcs == [0] and np.array_equal(recv[0][:size], np.full(size, 1, np.uint8))
    drop(lib, ctxs, g)


def _err(lib, h):
    lib.ssv_last_error.restype = C.c_char_p
    lib.ssv_last_error.argtypes = [C.c_void_p]
    return lib.ssv_last_error(h).decode()


@pytest.mark.parametrize("transport", ["host-exchange", "rccl"])
def test_exchange_never_leaves_a_rank_waiting(transport, monkeypatch):
    """whatever goes wrong on ONE rank - a vector of another size, a failure before the exchange, a failure inside it, a
    thread that never arrives - every rank RETURNS, all with an error, and (unless the group was broken) the next exchange works."""
    lib = _abi.hip_lib()
    n_dev = lib.ssv_device_count()
    if transport == "rccl":
        n, devices = min(n_dev, 4), list(range(min(n_dev, 4)))
    else:
        n, devices = 3, [0, 0, 0]
    E_HIP, E_ARG, E_STATE = -2, -3, -4

    def fresh(**env):
        for k in ("SSV_GROUP_FAIL", "SSV_GROUP_TIMEOUT_S", "SSV_GROUP_RCCL_ONE"):
            monkeypatch.delenv(k, raising=False)
        if transport == "rccl" and n == 1:
            monkeypatch.setenv("SSV_GROUP_RCCL_ONE", "1")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return _group(lib, devices)

    if n == 1:
        _one_rank_rccl_failures(lib, fresh, drop_group, E_HIP, E_STATE)
        return

    def drop(ctxs, g):
        lib.ssv_group_destroy(g)
        for h in ctxs:
            lib.ssv_ctx_destroy(h)

    # vectors of different sizes: agreed on BEFORE anything is exchanged; the group stays usable
    ctxs, g = fresh()
    assert lib.ssv_group_uses_rccl(g) == (1 if transport == "rccl" else 0)
    sizes = [4096] * n
    sizes[n - 1] = 4000
    rcs, _, hung = _run_ranks(lib, g, n, sizes)
    assert not hung and rcs == [E_ARG] * n and "different sizes" in _err(lib, ctxs[0])
    rcs, recv, hung = _run_ranks(lib, g, n, [4096] * n)
    assert not hung and rcs == [0] * n
    assert all(np.array_equal(recv[r][:4096 * n], np.repeat(np.arange(1, n + 1, dtype=np.uint8), 4096)) for r in range(n))
    drop(ctxs, g)

    # a rank that fails before the exchange (allocation, copy ...): it reports its own error, the others SSV_E_STATE; usable afterwards
    ctxs, g = fresh(SSV_GROUP_FAIL="1:1")
    rcs, _, hung = _run_ranks(lib, g, n, [512] * n)
    assert not hung and rcs[1] == E_HIP and all(rcs[r] == E_STATE for r in range(n) if r != 1)
    assert "injected" in _err(lib, ctxs[1]) and "another rank failed" in _err(lib, ctxs[0])
    drop(ctxs, g)

    # a rank that fails INSIDE the exchange (ncclAllGather returns an error on it alone): the peers come back too
    ctxs, g = fresh(SSV_GROUP_FAIL="0:2", SSV_GROUP_TIMEOUT_S="20")
    rcs, _, hung = _run_ranks(lib, g, n, [512] * n)
    assert not hung and rcs[0] == E_HIP and all(rcs[r] in (E_STATE, E_HIP) for r in range(1, n)), rcs
    drop(ctxs, g)

    # a rank whose thread never arrives: the others give up after SSV_GROUP_TIMEOUT_S, and the group says so from then on
    ctxs, g = fresh(SSV_GROUP_TIMEOUT_S="0.5")
    rcs, _, hung = _run_ranks(lib, g, n, [512] * n, skip=(n - 1,))
    assert not hung and all(rcs[r] == E_STATE for r in range(n - 1)) and "did not reach the exchange" in _err(lib, ctxs[0])
    rcs, _, hung = _run_ranks(lib, g, n, [512] * n)
    assert not hung and rcs == [E_STATE] * n
    drop(ctxs, g)


@pytest.mark.parametrize("inflate", ["host", "-Z"])
def test_cli_ranks_on_real_devices(tmp_path, inflate):
    """`seeksv getsv -N n` with one rank per real GPU: the tallies and depths meet in ncclAllGather and equal the single-GPU table"""
    import os
    import subprocess
    import bamio
    import golden_util as G
    from seeksv_amd import host
    lib = _abi.hip_lib()
    n_dev = lib.ssv_device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.environ.get("SSV_CLI") or os.path.join(root, "seeksv_amd", "bin", "seeksv")
    base = os.path.join(G.GOLDEN, "getsv")
    bam = os.path.join(base, "pairs1.bam")
    with host.BamReader(bam) as r:
        names, lens = r.target_names, [int(x) for x in r.target_lens]
    empty_bam = str(tmp_path / "empty.clip.bam")
    bamio.write_bam(empty_bam, names, lens, [])
    empty_clip = str(tmp_path / "empty.clip")
    open(empty_clip, "w").close()
    sv = str(tmp_path / "out.sv")
    env = dict(os.environ, SSV_TIMING="1")
    if n_dev < 2:
        env["SSV_GROUP_RCCL_ONE"] = "1"   # one run of records, its vector through a one-rank RCCL communicator
    r = subprocess.run([exe, "getsv"] + ([inflate] if inflate != "host" else []) + ["-N", str(min(n_dev, 4)), "-d", "0", "-f", "0", "-b", "0", "-T", "100000", "-B", os.path.join(base, "pairs1.junctions.txt"), empty_bam, bam, empty_clip, sv,
                        str(tmp_path / "x.fq")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "exchange over RCCL" in r.stderr
    assert open(sv).read() == G.read_text("getsv", "pairs1.sv")
