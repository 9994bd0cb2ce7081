#!/usr/bin/env python3
"""The GPU re-aligner (ssv_realign_*) at the size of the bench workload: index of the whole synthetic genome (generated in
HBM), then queries cut from it - half of them real placements (both strands, 0.5 % substitutions), half random sequence like the bulk of
a sample's soft clips.  usage: python tools/realign_bench.py [genome_frac] [n_queries] [query_len]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seeksv_amd import _abi, synth  # noqa: E402
from seeksv_amd.device import Context  # noqa: E402


def main():
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    qlen = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    import torch
    w = synth.Workload(genome_frac=frac, depth=1, n_sv=0)
    t = time.perf_counter()
    words, off = w.reference_2bit(0)
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t
    ctx = Context(0)
    ctx.prof_enable(1)
    t = time.perf_counter()
    dropped = ctx.realign_index(words.data_ptr(), off, _abi.MEM_DEVICE)
    t_index = time.perf_counter() - t
    # queries
    rng = np.random.RandomState(11)
    G = int(off[-1])
    n_real = nq // 2
    tid = rng.randint(0, len(off) - 1, n_real)
    start = (off[tid] + (rng.random_sample(n_real) * (np.diff(off)[tid] - qlen)).astype(np.int64)).astype(np.int64)
    wh = words.cpu().numpy().view(np.uint64)
    idx = start[:, None] + np.arange(qlen)[None, :]
    codes = ((wh[idx >> 5] >> ((idx & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)
    sub = rng.random_sample(codes.shape) < 0.005
    codes = np.where(sub, (codes + 1 + rng.randint(0, 3, codes.shape)) & 3, codes).astype(np.uint8)
    rev = rng.random_sample(n_real) < 0.5
    codes[rev] = (3 - codes[rev])[:, ::-1]
    junk = rng.randint(0, 4, (nq - n_real, qlen)).astype(np.uint8)
    allc = np.concatenate([codes, junk])
    order = rng.permutation(nq)
    allc = allc[order]
    lut = np.frombuffer(b"ACGT", np.uint8)
    text = lut[allc].tobytes().decode()
    seqs = [text[i * qlen:(i + 1) * qlen] for i in range(nq)]
    t = time.perf_counter()
    hits = ctx.realign(seqs)
    t_query = time.perf_counter() - t
    prof = ctx.prof_all()
    is_real = order < n_real
    exp_tid = np.full(nq, -1)
    exp_tid[is_real] = tid[order[is_real]]
    exp_pos = np.full(nq, -1, np.int64)
    exp_pos[is_real] = (start - off[tid])[order[is_real]]
    ok_real = int(((hits["tid"] == exp_tid) & (hits["pos"] - hits["q_beg"] == exp_pos) & (hits["mapq"] > 0))[is_real].sum())
    junk_unaligned = int((hits["tid"][~is_real] == -1).sum())
    out = {"genome_bases": G, "reference_2bit_s": round(t_ref, 3), "index_wall_s": round(t_index, 3), "index_kernel_ms": round(prof["realign_index"]["total_ms"], 2),
           "index_positions_per_s": round(G / 4 / (prof["realign_index"]["total_ms"] * 1e-3)), "index_dropped": dropped, "table_GB": round(4 * (1 << int(np.ceil(np.log2(G / 4 * 2)))) / 1e9, 2),
           "queries": nq, "query_len": qlen, "query_wall_s": round(t_query, 3), "query_kernel_ms": round(prof["realign_query"]["total_ms"], 2),
           "queries_per_s_kernel": round(nq / (prof["realign_query"]["total_ms"] * 1e-3)), "real_placed_correctly": ok_real, "real": int(is_real.sum()),
           "junk_unaligned": junk_unaligned, "junk": int((~is_real).sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
