#!/usr/bin/env python3
"""bench.py - BAM records/s through the getclip + getsv device path on synthetic WGS records resident in HBM.

One "step" = one full pass of the hot path over this rank's shard:
    clip_scan -> event sort -> cluster_bins -> cluster table to host            (seeksv getclip)
    insert-size stats on the global file prefix                                 (getsv pass 1)
    host bookkeeping (junction windows, flank windows) -> fused getsv_scan -> depth finish   (getsv passes 2+3)
    one all-gather of the per-rank result vector (N > 1)
Workload (BASELINE.json configs[1] + planted SVs so that the getsv passes have junctions to serve): synthetic 30x WGS,
150 bp paired end, ~617 M records per GPU, 1 % random soft clips, 10 k planted DEL/INV/TRA at VAF 0.5.  With N GPUs the
depth is 30x * N over the same genome (weak scaling: per-GPU records fixed), range-partitioned by reference interval.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--genome-frac F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# minimum traffic of the whole path if every field of a record were read once by one fused pass (33 B fixed part +
# 4.1 B CIGAR + 1 % x 225 B bases/qualities + ~0.7 B out).  `roofline` in the bench line prices ALL device kernels of one step with it:
# achieved = 40 B x records / (sum of the kernels' time).
PATH_BYTES_PER_RECORD = 40.0
# HBM bytes each kernel group must move per unit with the implemented algorithm; unit = what the group's launches process.
#   clip_scan   : cigar_ends 1 per record - only records with an S at either end of the CIGAR (~1 %) go on (without that column:
#                 n_cigar 2 per record, ~3 % go on)                                                                                   = 1 B/record
#   getsv_scan  : pos 4 per record (tile map stays in L2); the tid column comes as runs with the batch (ssv_batch_t.tid_runs: one per contig) and is read
#                 only in the tiles a run boundary falls into (without runs: tid 4 + pos 4 = 8 B/record)                             = 4 B/record
#   clip_place  : per candidate: staged index 4 + 4, record line 64, count 1; per event (0.31 per candidate): line 64 + staged key 16 written,
#                 staged key 16 read, side-list key 12 + (l_qseq, n_cigar) 8 + slot 4 written                                        = 110 B/candidate
#   event_sort  : '3' events (half): one windowed rank pass (12 read + 12 written + check 12); all: key 12 + line 64 read, 8 + 64 written = 166 B/event
#   cluster_pack: the byte-by-byte model (line 32 + 9, sizes 16; two scans 48; line 64 + 30, row 12 + descriptor 32 + CIGAR 10; descriptor 32 + read 228 +
#                 block 100 = 613 B/slot) overstates what the kernels move: the PMC passes count 2.56 GB for 5.69 M slots (profiles/traffic.json: caches
#                 serve the second look at a line)                                                                                    = 450 B/slot (measured)
ALGO_BYTES = {"clip_scan": 1.0, "getsv_scan": 4.0, "clip_place": 110.0, "event_sort": 166.0, "cluster_pack": 450.0}
DEVICE_GROUPS = ("clip_scan", "clip_place", "clip_gather", "event_sort", "cluster_bins", "cluster_pack", "isize_stats", "getsv_scan", "getsv_cand", "depth_finish")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-frac", type=float, default=1.0, help="scale the contigs (and the record count) down, e.g. 0.015625 for a quick run")
    ap.add_argument("--depth", type=float, default=30.0, help="per-GPU coverage")
    ap.add_argument("--n-sv", type=int, default=10000)
    ap.add_argument("--cpu-sample", type=int, default=12_000_000, help="records of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--ref-sample", type=int, default=3_000_000, help="records of the sample the real reference binary (oracle/_ref, if it travelled) is timed on (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--file-frac", type=float, default=-1.0, help="(default: --genome-frac, i.e. the bench workload itself = BASELINE config 2, 617 M records / 44 GB, where >= 192 host cores write the file in about a minute; a quarter of it from 64 cores; else 1/16 or 1/64) genome fraction of the BAM FILE leg (file_path in the bench line: compressed bytes in pinned host memory -> device "
                    "inflate + decode -> scans -> tables on the host); 0 = skip")
    ap.add_argument("--file-level", default="auto", help="deflate level of the file leg's BAM: 1..9 = zlib, fast = the repository's own single-probe LZ77 + Huffman coder (huff_gz.h: deflate_fast, "
                    "~5 x zlib level 4's speed, 93 instead of 77 B/record); auto (default): 6 = samtools' default where >= 64 CPUs write the file, fast where 16 CPUs have to write the WHOLE sample, else 4")
    ap.add_argument("--qual-alphabet", choices=("binned5", "hiseq40"), default="binned5", help="the synthetic reads' base qualities: five binned values {2, 11, 25, 37, 40} (NovaSeq-like, the default) or forty values 2..41 (HiSeq-like: the compact table's 11-bit pairs, a file that deflates less)")
    ap.add_argument("--no-host-batch", action="store_true", help="skip host_batch_path (the PCIe-inclusive rate: host SoA batches of a sixteenth of the sample through ssv_clip_scan)")
    ap.add_argument("--no-config3", action="store_true", help="skip config3_path (BASELINE config 3's shape: 300x over a tenth of the genome, the same number of records, resident in HBM)")
    ap.add_argument("--config5-frac", type=float, default=-1.0, help="genome fraction of config5_path (BASELINE config 5 through the product: human + HBV tumor at 2 x --depth against its normal, `seeksv run` + `getclip` + `somatic`); "
                    "default: 1/16 of --genome-frac (two BAM files are written for it: 116 M records in all); 1 = the real size (1.85 G records, 175 GB of files: profiles/r06_config5_full_size.json); 0 = skip")
    ap.add_argument("--n-integrations", type=int, default=50, help="planted human <-> HBV junctions of config5_path's tumor")
    ap.add_argument("--unmapped-frac", type=float, default=0.02, help="share of config5_path's records that come in pairs with one unmapped end (getclip's unmapped-pair side channel, clip_reads.h:172-219)")
    ap.add_argument("--ranks-frac", type=float, default=0.25, help="N > 1: the share of --genome-frac that ranks_path writes as a BAM file for `seeksv getclip|getsv -Z -N <gpus>`; 0 = skip")
    ap.add_argument("--no-cli-leg", action="store_true", help="skip cli_path (the `seeksv` binary as child processes on the file leg's BAM)")
    ap.add_argument("--ascii-table", action="store_true", help="cluster tables with ASCII sequences (the C ABI's default layout) instead of 4-bit codes")
    ap.add_argument("--table-format", type=int, default=3, help="ssv_clip_table_format: 3 = the compact table (default), 0 = ASCII")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="weak (default): every GPU holds its own --depth sample, N GPUs = N x depth over the same genome.  "
                    "strong = BASELINE config 4: ONE sample of --strong-depth (300x tumor WGS, 6.18 G records) range-partitioned N ways; refused where a rank's share does not fit its GPU")
    ap.add_argument("--strong-depth", type=float, default=300.0, help="coverage of the one sample that --scaling strong splits over the GPUs")
    ap.add_argument("--no-overlap", action="store_true", help="collect every cluster table in its own step (no copy in flight while other kernels run); use under rocprofv3, which serialises dispatches of different streams")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` by itself: this process starts the N ranks - one fresh process per GPU through torch.distributed.run, the launch the
        # driver's contract names - BEFORE it has touched the GPU (no torch import, no HIP call: a plain child process, never a re-exec), relays
        # what they print (rank 0's one JSON line) and leaves with their exit code
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # test hooks (a 1-GPU box can still exercise the N > 1 code path): SSV_FORCE_DEVICE pins every rank to one GPU,
    # SSV_DIST_BACKEND=gloo replaces RCCL (which needs one GPU per rank) by gloo over host memory
    if os.environ.get("SSV_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["SSV_FORCE_DEVICE"])
    backend = os.environ.get("SSV_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = dev if (world > 1 and backend == "nccl") else None
    # a host-side group for the waits around rank 0's reporting legs: a rank parked in an RCCL barrier keeps a kernel spinning on its GPU - the GPUs `seeksv -N` (ranks_path) runs on
    wait_group = None
    if world > 1:
        try:
            wait_group = dist.new_group(backend="gloo")
        except Exception as e:   # (a host on which gloo finds no interface: the waits go through the job's own backend)
            print(f"[bench] rank {rank}: no gloo group for the waits ({type(e).__name__}: {e})", file=sys.stderr)
    # N > 1: every rank keeps its threads - and with them the pinned buffers it allocates (the 0.66 GB table lands in them every step) - on the CPUs
    # next to its own GPU, so that eight tables a step do not cross the sockets' link
    near_cpus = bind_near_gpu(torch, local_rank) if world > 1 and os.environ.get("SSV_NO_CPU_BINDING") is None else 0

    from seeksv_amd import host, shard, synth
    from seeksv_amd.device import Context

    def resident(args, ctx=None):
        """the resident-input leg on one workload (args.depth / args.genome_frac / args.n_sv): generation, K timed steps, the breakdown steps"""

        strong = args.scaling == "strong"
        w = synth.Workload(genome_frac=args.genome_frac, depth=args.strong_depth if strong else args.depth * world, n_sv=args.n_sv, qual_model=1 if args.qual_alphabet == "hiseq40" else 0)
        sp = shard.shard_plan(w, rank, world)
        t0 = time.time()
        n_scan = sp["own_hi_rec"] - sp["scan_lo_rec"]
        n_own = sp["own_hi_rec"] - sp["own_lo_rec"]
        # a rank's records stay resident for the whole run: ~80 bytes each (hot columns 11, one 64-byte line, CIGAR 4, bases of the 1 % soft-clipped reads)
        # plus the pass's events, tables and scratch; a batch holds fewer than 2^31 records (include/seeksv_hip.h)
        hbm = torch.cuda.get_device_properties(local_rank).total_memory
        if n_scan >= (1 << 31) - 64 or n_scan * 80.0 > 0.75 * hbm:
            raise SystemExit(f"bench.py: a share of {n_scan} records ({n_scan * 80 / 1e9:.0f} GB resident) does not fit one GPU ({hbm / 1e9:.0f} GB, < 2^31 records per batch): "
                             f"{'--scaling strong at ' + format(args.strong_depth, 'g') + 'x needs more GPUs (300x: 4 or 8) or a lower --strong-depth' if strong else 'lower --depth or --genome-frac'}")
        # the batch as the device decoder hands it over: hot columns (tid, pos, n_cigar), one 64-byte line per record with the cold fields, CIGARs,
        # packed bases + qualities of the soft-clipped records.  It stays resident and unchanged over the timed region (SSV_MEM_PERSISTENT):
        # clip events point at the reads' own bytes and the cluster table is cut straight out of them.
        scan_batch, scan_t = w.generate_device(sp["scan_lo_rec"], n_scan, local_rank, soa=False, persistent=True)
        # the rank's own records = the scan batch minus the leading halo (same cigar / seqqual blobs, offsets are absolute)
        from seeksv_amd import _abi

        def sub_batch(batch, first, n):
            arrays = {}
            for name, dt in _abi.BATCH_FIELDS:
                ptr = getattr(batch, name)
                arrays[name] = ptr if name in ("cigar", "seqqual", "xc") or ptr is None else ptr + first * np.dtype(dt).itemsize
            arrays["rec"] = batch.rec + first * 64 if batch.rec else None
            arrays["n_cigar_total"], arrays["seqqual_bytes"], arrays["max_ref_span"] = batch.n_cigar_total, batch.seqqual_bytes, batch.max_ref_span
            arrays["tid_runs"] = _abi.rebase_runs(batch.get_tid_runs(), first, n)
            return _abi.make_batch(arrays, mem=batch.mem, n=n)[0]

        own_batch = sub_batch(scan_batch, sp["own_lo_rec"] - sp["scan_lo_rec"], n_own)
        # global file prefix for the insert-size statistics (cluster.cpp:68 stops after 5 M qualifying records)
        n_prefix = min(w.n_total, 6_500_000)
        if sp["scan_lo_rec"] == 0 and n_prefix <= n_scan:
            prefix_batch, prefix_t = sub_batch(scan_batch, 0, n_prefix), None
        else:
            prefix_batch, prefix_t = w.generate_device(0, n_prefix, local_rank, soa=False)
        torch.cuda.synchronize()
        gen_s = time.time() - t0

        if ctx is None:
            ctx = Context(local_rank)
        table_format = 0 if args.ascii_table else args.table_format
        # the table is the getclip stage's output and PCIe bounds the step: the compact format ships 12 bytes of fixed columns per cluster, bases
        # at 2 bits, qualities as alphabet indices; the host rebuilds contig / side / offsets (ssv_clip_table_expand, inside the timed step)
        ctx.clip_table_format(table_format)
        hdr = host.Header(w.names, w.lens)
        own = sp["own"] if world > 1 else None
        jtable = host.JunctionTable(w.junctions)

        wall = {}
        state = {"pending": False, "support_sum": 0, "tables": 0}

        def collect_table(prev):
            tw = time.perf_counter()
            t = ctx.clip_table_wait(prev=prev)
            n = t.n_clusters
            if t.format == 3:
                te = time.perf_counter()
                ctx.clip_table_expand(t, 0)
                state.setdefault("expand_ms", []).append((time.perf_counter() - te) * 1e3)
                state.setdefault("wait_ms", []).append((te - tw) * 1e3)
                state["table_bytes"] = n * (4 + 2 * t.len_bytes + t.support_bytes + t.ncig_bytes + 1) + t.str_bytes + t.cigar_bytes * t.cigar_ops + 16 * t.n_runs + 8 * t.n_base_exc
            else:
                state["table_bytes"] = n * 42 + t.str_bytes + 4 * t.cigar_ops
            state["table_info"] = dict(format=int(t.format), qual_bits=int(t.qual_bits), qual_group=int(t.qual_group), cigar_bytes=int(t.cigar_bytes), base_bits=int(t.base_bits), base_exceptions=int(t.n_base_exc))
            ssum = int(t.support_sum) if t.format == 3 else (int(np.ctypeslib.as_array(t.support, shape=(n,)).sum()) if n else 0)
            assert ssum == t.n_events, "clip events were lost or duplicated"
            assert bool(t.seq_packed) == (not args.ascii_table)
            state["support_sum"] = ssum
            state["tables"] += 1

        def start_plan_worker():
            box = {}
            th = threading.Thread(target=lambda: box.setdefault("plan", host.Plan(hdr, jtable, 0, 0)))
            th.start()
            state["plan_worker"], state["plan_box"] = th, box

        def drain():
            if state["pending"]:
                collect_table(prev=False)
                state["pending"] = False


        trace = [] if os.environ.get("SSV_BENCH_TRACE") else None   # debugging: wall-clock laps of every step on stderr

        def step(timed=None):
            t = [time.perf_counter()]
            if trace is not None and timed is None:
                timed = {}
                trace.append(timed)

            def lap(name):
                if timed is not None:
                    now = time.perf_counter()
                    timed[name] = timed.get(name, 0.0) + (now - t[0]) * 1e3
                    t[0] = now

            # host bookkeeping that does not depend on the insert-size statistics (flank windows, depth ranges, points: ~8 ms on one host core for
            # 10 k junctions) is built by a worker thread beside the GPU work; only the junction windows are refreshed once mean / sd are known.
            # Like the table copy, it runs one step ahead: the plan a step consumes was started when the step before began its getsv pass (every
            # step still builds exactly one plan).
            if state.get("plan_worker") is None:
                start_plan_worker()
            worker, box = state["plan_worker"], state["plan_box"]
            ctx.clip_begin(0.9, 1, False, own, sp["initial_last_tid"])
            ctx.clip_scan(scan_batch)
            lap("clip_scan+events")
            # the cluster table's copy to the host (1.8 GB over PCIe) is queued on a second stream and collected one step later, so it
            # overlaps with the getsv passes and with the next step's kernels; every table is collected before the timed region ends
            n_clusters, n_events = ctx.clip_cluster_async()
            lap("clip_cluster(kernels)")
            if args.no_overlap or (timed is not None and trace is None):
                collect_table(prev=False)   # the per-kernel breakdown step and --no-overlap runs: nothing else in flight
                lap("table_d2h(wait)")
            rc, npairs, mean, sd = ctx.isize_stats([prefix_batch], 20, 5000000)
            lap("isize_stats")
            worker.join()
            plan = box["plan"]
            start_plan_worker()   # the next step's
            plan.update_isize(mean, sd)
            lap("host_plan(wait)")
            ctx.getsv_begin(plan.junctions, plan.windows, mean, sd, hdr.target_lens, 4, 20, 20)
            lap("getsv_begin")
            ctx.getsv_scan(own_batch)
            counts, rs, pd, max_depth = ctx.getsv_finish(plan.ranges, plan.points)
            lap("getsv_scan+finish")
            if not (args.no_overlap or (timed is not None and trace is None)):
                if state["pending"]:
                    collect_table(prev=True)
                state["pending"] = True
                lap("table_wait(previous step)")
            vec = shard.pack_results(counts, rs, pd, n_clusters, n_events, state["support_sum"])
            stacked = shard.all_gather_vector(vec, coll_dev)
            merged = shard.merge_results(stacked, len(counts), len(rs), len(pd))
            lap("exchange")
            folded = plan.fold(merged[0], merged[1], merged[2])
            plan.close()
            lap("host_fold")
            return dict(n_clusters=merged[3], n_events=merged[4], support_sum=merged[5], mean=mean, sd=sd, abnormal_sum=int(folded["abnormal"].sum()),
                        depth_sum=int(folded["up_depth"].sum() + folded["down_depth"].sum()), flank_sum=int(folded["flank"].sum()), max_depth=max_depth)

        def barrier():
            drain()
            torch.cuda.synchronize()
            ctx.sync()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        res = None
        # setup, not warmup: both table sets (device + pinned host buffers), both sets of host rebuild buffers and the host thread pool only exist
        # after four passes have gone through the two-deep table pipeline (the first use of each costs page faults and pinning: 5-10 ms);
        # whatever W the caller asks for, the timed steps then find them in place
        for _ in range(max(0, 4 - args.warmup)):
            step()
        for _ in range(args.warmup):
            res = step()
        import gc
        gc.collect()
        gc.disable()   # (a collection inside a 12 ms step is a visible outlier)
        ctx.prof_reset()
        ctx.prof_enable(2)  # HIP events around the two streaming kernels only, on the context's stream
        barrier()
        t0 = time.perf_counter()
        step_walls = []
        for _ in range(args.steps):
            ts = time.perf_counter()
            res = step()
            step_walls.append(round((time.perf_counter() - ts) * 1e3, 2))
        tb = time.perf_counter()
        barrier()
        gc.enable()
        if state.get("plan_worker") is not None:   # the plan built ahead for a step that does not come
            state["plan_worker"].join()
            state["plan_box"]["plan"].close()
            state["plan_worker"] = None
        if trace is not None:
            for k, tr in enumerate(trace[-(args.steps):]):
                print("[bench trace] step", k, {a: round(b, 2) for a, b in tr.items()}, file=sys.stderr)
            trace = None
        step_walls.append(("final_barrier", round((time.perf_counter() - tb) * 1e3, 2)))
        dt = time.perf_counter() - t0
        prof = {k: ctx.prof_get(k) for k in ("clip_scan", "getsv_scan")}
        ctx.prof_enable(0)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev is not None else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # three extra, untimed steps with every kernel group bracketed by events, nothing else in flight: the per-kernel breakdown (per-step means:
        # one launch of a sub-millisecond kernel varies by a few percent)
        BREAKDOWN_STEPS = 3
        ctx.prof_reset()
        ctx.prof_enable(1)
        for _ in range(BREAKDOWN_STEPS):
            step(wall)
            drain()
        allprof = ctx.prof_all()
        for v in allprof.values():   # per step
            v["total_ms"] /= BREAKDOWN_STEPS
            v["launches"] = v["launches"] // BREAKDOWN_STEPS if v["launches"] >= BREAKDOWN_STEPS else v["launches"]
            v["units"] = v["units"] // BREAKDOWN_STEPS
        for k in list(wall):
            wall[k] /= BREAKDOWN_STEPS
        breakdown = {k: round(v["total_ms"], 4) for k, v in allprof.items() if v["launches"]}
        ctx.prof_enable(0)
        del scan_batch, scan_t, own_batch, prefix_batch, prefix_t   # (the sample leaves HBM with the leg)
        return dict(locals())

    link = {"what": "device <-> pinned host through hipMemcpyAsync (the runtime's choice of SDMA engine), 256 MB, [median, slowest] GB/s of four copies, taken between the legs (host_link_probe)"}
    gpu_dev = dev
    link["before"] = host_link_probe(torch, gpu_dev)
    R = resident(args)
    link["after the resident leg"] = host_link_probe(torch, gpu_dev)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    (w, sp, n_own, hdr, ctx, strong, gen_s, state, res, dt, prof, allprof, breakdown, wall, step_walls, BREAKDOWN_STEPS) = (R[k] for k in (
        "w", "sp", "n_own", "hdr", "ctx", "strong", "gen_s", "state", "res", "dt", "prof", "allprof", "breakdown", "wall", "step_walls", "BREAKDOWN_STEPS"))

    del R
    gc.collect()
    torch.cuda.empty_cache()
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier(group=wait_group)   # (every rank has given its sample back before rank 0 starts the product's ranks on all GPUs: ranks_path)
    if rank == 0:
        assert state["tables"] >= args.steps, "every step's cluster table must have reached the host"
        total_records = w.n_total
        ms_per_step = dt / args.steps * 1e3
        # Roofline: the whole path.  achieved = SURVEY 8(d)'s 40 B/record x this rank's records / the sum of ALL device kernels of one step
        # (HIP events on the context's stream around every kernel group, nothing else in flight: the extra breakdown steps above); every
        # group is listed with its own time, its byte model where it has one and - from the committed PMC passes (profiles/traffic.json) - the
        # HBM bytes it really moved; the longest group is named.  The two streaming kernels (the only ones that touch every record) are also
        # timed inside the K timed steps, the table copy of the previous step in flight beside them.
        dev = {k: v for k, v in allprof.items() if k in DEVICE_GROUPS and v["launches"]}
        dev_ms = sum(v["total_ms"] for v in dev.values())
        longest = max(dev, key=lambda k: dev[k]["total_ms"])
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = {}
        scale = n_own / traffic["records"] if traffic.get("records") else None   # the PMC passes ran the default workload

        def group_entry(k, v):
            per = v["total_ms"]
            units = v["units"] / max(v["launches"], 1) if k in ("clip_scan", "getsv_scan", "clip_place", "event_sort", "cluster_pack") else None
            e = {"ms": round(per, 4)}
            if k in ALGO_BYTES and units:
                e["units"] = int(units)
                e["algorithmic_bytes_per_unit"] = ALGO_BYTES[k]
                e["achieved_GBs"] = round(ALGO_BYTES[k] * units / (per * 1e-3) / 1e9, 1)
                e["frac"] = round(e["achieved_GBs"] / HBM_PEAK_GBS, 3)
            t = traffic.get("groups", {}).get(k)
            if t is not None and scale is not None and abs(scale - 1.0) < 0.02:
                e["traffic_bytes"] = t
                e["hbm_frac_measured"] = round(t / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)   # the group's counter bytes over its time: the share of the memory system it really uses
            return e

        groups = {k: group_entry(k, v) for k, v in dev.items()}
        path_bytes = PATH_BYTES_PER_RECORD * n_own
        achieved = path_bytes / (dev_ms * 1e-3) / 1e9
        total_traffic = traffic.get("path_bytes_per_step") if (scale is not None and abs(scale - 1.0) < 0.02) else None
        timed = {}
        for k in ("clip_scan", "getsv_scan"):
            if prof.get(k, {}).get("launches"):
                timed[k] = {"avg_launch_ms": round(prof[k]["total_ms"] / prof[k]["launches"], 4), "launches_timed": int(prof[k]["launches"]),
                            "achieved_GBs": round(ALGO_BYTES[k] * (prof[k]["units"] / prof[k]["launches"]) / (prof[k]["total_ms"] / prof[k]["launches"] * 1e-3) / 1e9, 1)}
        line = {
            "metric": "BAM records/sec through getclip+getsv",
            "value": total_records * args.steps / dt,
            "unit": "records/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "i32/u8 (fp64 match-rate compare)", "data": "synthetic",
            "config": {"workload": (f"synthetic {args.strong_depth:g}x WGS (BASELINE config 3/4), ONE sample range-partitioned over {world} GPU(s), " if strong else f"synthetic {args.depth:g}x-per-GPU WGS, ") +
                                   f"150 bp PE, 1% random soft clips, {len(w.junctions)} planted DEL/INV/TRA (VAF 0.5), genome_frac {args.genome_frac:g}, {n_own} records/GPU resident in HBM" + (", forty-value base qualities (--qual-alphabet hiseq40)" if args.qual_alphabet == "hiseq40" else ""),
                       "records_total": total_records, "records_per_gpu": n_own, "junctions": len(w.junctions), "parallelism": f"range-partition x{world}",
                       "multi_gpu": (f"strong scaling (--scaling strong): BASELINE config 4 - the fixed {args.strong_depth:g}x sample split {world} ways by reference interval, halo at the cuts, one all-gather" if strong else
                                     "weak scaling (default): 30x per GPU over the same genome (N GPUs = 30N x); `--scaling strong` runs BASELINE config 4's fixed 300x sample split N ways") +
                                    ("; this leg's exchange: ONE torch.distributed.all_gather per step (RCCL, one process per GPU); the product's own exchange (ssv_group_allgather = ncclAllGather from the host library, ranks as threads of `seeksv -N`) is timed in ranks_path" if world > 1 else ""),
                       "batch_layout": "hot columns tid/pos/n_cigar + cigar_ends (a one-byte copy of the first and last CIGAR operation codes: the getclip stream reads it and applies the soft-clip test to every record) + the tid column also as runs (one per contig: the getsv stream reads pos only) + one 64-byte line per record (ssv_record) + CIGARs + packed bases/qualities of soft-clipped records; SSV_MEM_PERSISTENT",
                       "host_cpus_bound_near_gpu": near_cpus, "generation_s": round(gen_s, 2)},
            "roofline": {"kernel": "path: all device kernels of one step (getclip + insert size + getsv passes; PCIe copy excluded)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": total_traffic,
                         # `traffic` is NOT measured in this run: PMC passes cannot run inside a bench line.  It is the committed counter file of the builder's
                         # PMC passes over this same command; used only when this run's records per GPU equal that file's (within 2 %: the rank split), null otherwise
                         "traffic_source": (f"profiles/traffic.json ({traffic.get('_round', 'builder-run')} rocprofv3 PMC passes over `bench.py --steps 1 --warmup 1 --no-overlap`, "
                                            f"{traffic.get('records')} records; this run: {n_own} records per GPU)" if total_traffic else
                                            f"none: profiles/traffic.json holds {traffic.get('records')} records per GPU, this run {n_own}"),
                         # the other way to read the same launches: the bytes the kernels really moved (PMC counters) over their time.  The path reads 18 of the 40
                         # "read every field once" bytes (the cold 64-byte lines are touched for 1-2 % of the records only), so `frac` (work done per second against
                         # the peak) is higher than the share of the memory system that is in use.
                         "hbm_frac_measured": (round(total_traffic / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) if total_traffic else None),
                         "note": "frac is the contract's budget figure: SURVEY 8(d)'s 40 B/record x records / the step's device time / 8 TB/s - the path reads ~14 of those 40 bytes (the cold "
                                 "64-byte lines are touched for 1-2 % of the records), so kernels-only it can exceed 1 and says how fast the WORK is done, not how busy the memory is; "
                                 "hbm_frac_measured (counter bytes over the same time, here and per group) is the utilisation figure",
                         "algorithmic_bytes_per_record": PATH_BYTES_PER_RECORD, "records_per_launch": float(n_own), "avg_launch_ms": round(dev_ms, 4),
                         "launches_timed": BREAKDOWN_STEPS, "dominant_group": {"name": longest, **groups[longest]},
                         "groups": groups, "streaming_kernels_in_timed_region": timed},
            "kernel_ms_one_step": breakdown,
            "wall_ms_one_step": {k: round(v, 3) for k, v in wall.items()},
            "wall_ms_timed_steps": step_walls,
            "result": res,
            "table": dict(state.get("table_info", {}), host_expand_ms=[round(x, 2) for x in state.get("expand_ms", [])[-6:]], wait_ms=[round(x, 2) for x in state.get("wait_ms", [])[-6:]], bytes=int(state.get("table_bytes", 0)), bytes_per_cluster=round(state.get("table_bytes", 0) / max(1, res["n_clusters"]), 1),
                          note="what crosses PCIe per step; format 3: contig / side / offsets are rebuilt on the host inside the step (ssv_clip_table_expand)"),
        }
        if world == 1 and not args.no_config3 and not strong:
            # BASELINE config 3's shape on one GPU: ten times the depth over a tenth of the genome - the same number of records resident in HBM, the
            # same 10,000 planted SVs, but every planted breakpoint's bin holds 50-150 clipped reads and every junction window ten times the records
            # (the full 6.18 G-record sample does not fit one GPU at once: tests/test_full_size_gpu.py streams it).  Same step, same timing rules.
            try:
                a3 = argparse.Namespace(**vars(args))
                a3.depth, a3.genome_frac, a3.steps, a3.warmup = args.depth * 10.0, args.genome_frac * 0.1, 5, 2
                R3 = resident(a3, ctx=ctx)
                dev3 = {k: v for k, v in R3["allprof"].items() if k in DEVICE_GROUPS and v["launches"]}
                dev3_ms = sum(v["total_ms"] for v in dev3.values())
                n3 = R3["n_own"]
                ev3 = int(R3["res"]["n_events"])
                line["config3_path"] = {
                    "workload": f"synthetic {a3.depth:g}x WGS (BASELINE config 3's depth) over genome_frac {a3.genome_frac:g}, 150 bp PE, 1% random soft clips, {len(R3['w'].junctions)} planted DEL/INV/TRA (VAF 0.5), {n3} records resident in HBM",
                    "records": n3, "steps": a3.steps, "warmup": a3.warmup, "ms_per_step": R3["dt"] / a3.steps * 1e3, "value": R3["w"].n_total * a3.steps / R3["dt"], "unit": "records/s",
                    "kernel_ms_one_step": R3["breakdown"], "device_kernels_ms": round(dev3_ms, 4),
                    "roofline": {"bound": "hbm", "achieved": PATH_BYTES_PER_RECORD * n3 / (dev3_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": PATH_BYTES_PER_RECORD * n3 / (dev3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "algorithmic_bytes_per_record": PATH_BYTES_PER_RECORD, "groups": {k: round(v["total_ms"], 4) for k, v in dev3.items()},
                                 "dominant_group": max(dev3, key=lambda k: dev3[k]["total_ms"])},
                    "cluster_bins_ns_per_event": round(dev3["cluster_bins"]["total_ms"] * 1e6 / max(1, ev3), 3) if "cluster_bins" in dev3 else None,
                    "cluster_bins_ns_per_event_default_workload": round(dev["cluster_bins"]["total_ms"] * 1e6 / max(1, int(res["n_events"])), 3) if "cluster_bins" in dev else None,
                    "table": dict(R3["state"].get("table_info", {}), bytes=int(R3["state"].get("table_bytes", 0)), bytes_per_cluster=round(R3["state"].get("table_bytes", 0) / max(1, R3["res"]["n_clusters"]), 1)),
                    "result": R3["res"], "generation_s": round(R3["gen_s"], 2)}
                del R3
                link["after config3_path"] = host_link_probe(torch, gpu_dev)
                gc.collect()
                torch.cuda.empty_cache()   # (the 49 GB sample goes back to the device: the file leg below keeps its own 49 GB of decoded records)
            except BaseException as e:  # (SystemExit too: a report beside the headline)
                line["config3_path"] = {"error": f"{type(e).__name__}: {e}"}
        if args.file_frac < 0:
            cores_here = os.cpu_count() or 1
            try:
                avail_gb = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable:")) / (1 << 20)
            except Exception:
                avail_gb = 0.0
            # Every record with its bases and qualities: 72-76 B/record in the file.  What bounds the leg's size is writing the file with zlib on the
            # host (outside the timed region, but inside the run): level 6 deflates ~11 MB/s per core, level 4 (lazy matching too, 76 instead of
            # 72 B/record) ~40 MB/s.  The driver's box grants 16 CPUs of time: HALF of the sample (309 M records, 85 GB of records, a 23.7 GB file) at
            # level 4 takes about two minutes (round 3 took an eighth: a command's 0.5 s of start-up - process, HIP runtime, first allocations - is
            # as long as its work on a 6 GB file); >= 192 CPUs write the whole sample.
            eff = effective_cpus()
            # Round 5: the WHOLE sample (617.65 M records) also on a 16-CPU box - written with the repository's own fast deflate coder (a 4-byte hash with one
            # candidate, greedy, then the tokens' two Huffman codes: 93 B/record instead of level 4's 77, at ~5 x its speed), where round 4 took half the sample
            # at level 4.  The file (~58 GB) lives in /dev/shm: 160 GB of available memory are asked for.
            whole_fast = eff >= 16 and eff < 192 and avail_gb >= 160 and args.file_level in ("auto", "fast")
            args.file_frac = args.genome_frac * (1.0 if (eff >= 192 and avail_gb >= 256) or whole_fast else 0.5 if eff >= 16 and avail_gb >= 160 else 0.25 if eff >= 12 and avail_gb >= 64 else 0.125 if eff >= 8 and avail_gb >= 32 else 1 / 64)
            if args.file_level == "auto":
                args.file_level = "6" if eff >= 64 else "fast" if whole_fast else "4"
        args.config5_level = (6 if effective_cpus() >= 64 else -2) if args.file_level == "auto" else (-2 if args.file_level == "fast" else int(args.file_level))
        if args.file_level == "auto":
            args.file_level = "6"
        args.file_level = -2 if args.file_level == "fast" else int(args.file_level)
        if world == 1 and args.file_frac > 0:
            try:
                line["file_path"] = file_path_leg(ctx, args, local_rank)
                if args.file_frac == args.genome_frac and isinstance(line.get("result"), dict):
                    # the same sample once resident in HBM and once as a BAM file (written, inflated, decoded): every count both legs report must agree
                    fr = line["file_path"]["result"]
                    line["file_path"]["same_result_as_resident_path"] = all(fr[k] == line["result"][k] for k in fr if k in line["result"])
            except Exception as e:  # the leg is a report beside the headline, never a reason to lose the line
                line["file_path"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.file_frac > 0:
            link["after file_path + cli_path"] = host_link_probe(torch, gpu_dev)
        if world == 1 and args.config5_frac != 0 and not strong:
            try:
                if args.config5_frac < 0:
                    args.config5_frac = args.genome_frac / 16
                gc.collect(); torch.cuda.empty_cache()
                line["config5_path"] = config5_path_leg(args)
            except Exception as e:
                line["config5_path"] = {"error": f"{type(e).__name__}: {e}"}
        if world > 1 and args.ranks_frac > 0:
            try:
                line["ranks_path"] = ranks_path_leg(args, world, os.environ.get("SSV_FORCE_DEVICE"))
            except Exception as e:
                line["ranks_path"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_host_batch and not strong:
            # the boundary handing over HOST buffers (what libseeksv_host's reader produces): SoA batches, 49.5 B/record, through the staging path - never `value`
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import host_batch_rate
                gc.collect(); torch.cuda.empty_cache()
                hbp = host_batch_rate.measure(min(args.genome_frac, 1 / 16), 1 << 22, ctx)
                hbp["what"] = "host SoA batches (4 M records each; pageable / page-locked / page-locked and announced one ahead) -> H2D staging -> ssv_clip_scan, best of 3; PCIe-inclusive, not `value`"
                line["host_batch_path"] = hbp
            except Exception as e:
                line["host_batch_path"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline and args.cpu_sample > 0:
            # the three host legs are reports beside the headline; the two single-threaded ones (the oracle in this process, the real reference
            # binary as a child process) run side by side, then every core gets an oracle worker for a few seconds: ~20 s in all
            box = {}
            t_ref = threading.Thread(target=lambda: box.setdefault("ref", cpu_reference(w, min(args.ref_sample, w.n_total))))
            t_ref.start()
            line["cpu_baseline"] = cpu_baseline(w, hdr, min(args.cpu_sample, w.n_total), min_seconds=10.0)
            t_ref.join()
            if box.get("ref"):
                line["cpu_reference"] = box["ref"]
            allc = cpu_baseline_all_cores(args, w, seconds=4.0)
            if allc:
                line["cpu_baseline_all_cores"] = allc
        line["host_link_GBs"] = link
        print(json.dumps(line))
    if world > 1:
        dist.barrier(group=wait_group)
        dist.destroy_process_group()


def host_link_probe(torch, dev):
    """what the host link gives RIGHT NOW through the runtime's own choice of SDMA engine: 256 MB device -> pinned host and back, four copies each way, (median, slowest) GB/s.  One of the engines
    hipMemcpyAsync alternates between is busy for seconds behind every large free of device memory (profiles/r06_host_link.txt) and a copy on it gets half the link: the library's large copies go
    to an engine by name and move on when that happens (LinkCopy, seeksv_hip.hip); these numbers say what the box was like around the legs."""
    n = 256 << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    out = {}
    for name, (a, b) in (("d2h", (h, d)), ("h2d", (d, h))):
        ms = []
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            a.copy_(b, non_blocking=True)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t) * 1e3)
        ms = sorted(ms[1:])
        out[name] = [round(n / ms[len(ms) // 2] / 1e6, 1), round(n / ms[-1] / 1e6, 1)]
    del d, h
    return out


def launch_ranks(n, argv):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port <free> bench.py <argv>` as a child process;
    its stdout / stderr pass through; -> its exit code"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (RCCL between processes: the host driver only supports dmabuf IPC)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def effective_cpus():
    """CPUs this process can really use: the visible ones, the affinity mask, and the cgroup's CPU quota (a container may see 256 CPUs and be
    granted the time of 16: more threads than that only queue behind the throttle)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except Exception:
        pass
    return n


def bind_near_gpu(torch, index):
    """sched_setaffinity to the GPU's local CPUs (sysfs local_cpulist of its PCI function); returns how many, 0 = left alone"""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        cpus = set()
        for part in open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) < 8:
            return 0
        os.sched_setaffinity(0, cpus)
        return len(cpus)
    except Exception:
        return 0


def write_workload_bam(w, bam, level):
    """a workload as a BAM file, EVERY record with its bases and qualities (reference bases at the record's position with 0.2 % substitutions, qualities
    from the workload's model): records are generated by a pool of host threads (a C loop, GIL released) that runs ahead of the writer (which serialises and
    deflates a batch on all cores); a bounded queue keeps at most ~2 x cores batches of 0.5 M records (~120 MB each) in host memory.  level: 1..9 zlib, -2 the
    repository's fast coder (huff_gz.h)"""
    import queue
    from concurrent.futures import ThreadPoolExecutor
    from seeksv_amd import host
    os.environ["SSV_BGZF_LEVEL"] = str(level)
    cores = effective_cpus()
    os.environ.setdefault("SSV_WRITE_THREADS", str(cores))
    chunk = 500_000
    starts = list(range(0, w.n_total, chunk))

    def batches():
        n_gen = max(1, min(cores, len(starts)))
        with ThreadPoolExecutor(max_workers=n_gen) as ex:
            pending = queue.Queue()
            it = iter(starts)

            def submit():
                g = next(it, None)
                if g is not None:
                    pending.put(ex.submit(w.generate_host, g, min(chunk, w.n_total - g), False, True))
            for _ in range(2 * n_gen):
                submit()
            while not pending.empty():
                f = pending.get()
                submit()
                yield f.result()
    host.write_bam(bam, w.names, w.lens, batches())


def junction_file(path, junctions):
    """planted junctions as the 23-column rows the reference's -B harness reads (getsv.cpp:1292-1320)"""
    with open(path, "w") as f:
        for j in junctions:
            f.write("\t".join(str(x) for x in (j[0], j[1], j[2], 0, j[3], j[4], j[5], 0, 0, 0, "NA", 0, 0, 0, 0, 0, 0, 0, 0, "50M", "50M", "ACGT", "ACGT")) + "\n")


def timing_phases(stderr):
    """`[timing] <name> <seconds> s` lines of a command run with SSV_TIMING=1 -> {name: seconds}"""
    out = {}
    for line in stderr.splitlines():
        if line.startswith("[timing] "):
            try:
                name, sec, _ = line[9:].rsplit(" ", 2)
                out[name] = round(float(sec), 3)
            except ValueError:
                pass
    return out


def run_command(cmd, env):
    """a command as a child process with SSV_TIMING=1: -> (CompletedProcess, dict(total_s, phases_s, exec_to_main_s, exit_to_reaped_s)) - the last two from the
    command's own wall-clock stamps: what the process costs before main() and after its last statement (the kernel tearing its device memory down)"""
    import resource
    import subprocess
    c0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0, w0 = time.perf_counter(), time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    dt, w1 = time.perf_counter() - t0, time.time()
    c1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    # cpu_s: the command's user + system time on all its threads - over total_s, the CPUs it kept busy (a box that gives 16 CPUs of host time bounds a command at 16)
    out = dict(total_s=round(dt, 3), cpu_s=round(c1.ru_utime - c0.ru_utime + c1.ru_stime - c0.ru_stime, 2), phases_s=timing_phases(r.stderr))
    stamps = [float(l.split(":")[1]) for l in r.stderr.splitlines() if l.startswith("[stamp] wall clock at")]
    if len(stamps) == 2:
        out.update(exec_to_main_s=round(stamps[0] - w0, 3), exit_to_reaped_s=round(w1 - stamps[1], 3))
    return r, out


def room_for(need):
    """a directory with `need` bytes free: /dev/shm (the page cache), else the temp directory; None: neither"""
    import tempfile

    def room(path):
        try:
            st = os.statvfs(path)
            return st.f_bavail * st.f_frsize
        except OSError:
            return 0
    return next((p for p in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(p) and room(p) > need), None)


def file_path_leg(ctx, args, device):
    """The same workload from a BAM FILE: a smaller genome fraction of the same synthetic 30x sample is written as a real BAM by the
    repository's writer (libseeksv_host: BGZF level 6 like samtools), its compressed bytes are put into pinned host memory in chunks of
    whole BGZF blocks, and the timed region runs what `seeksv getclip` + `seeksv getsv` do with a file: per chunk H2D of the compressed
    bytes (the next chunk's announced ahead, ssv_bamdec_prefetch, so it runs beside this chunk's kernels) -> device BGZF inflate + BAM decode (ssv_bamdec_*) -> scans; pass 1 = getclip (clip events -> cluster table on the host), pass 2 =
    insert-size statistics on the file's first records, then the fused discordant + depth scan of every record - over the records pass 1
    decoded and kept in HBM (ssv_batch_retain): one inflate of the file -> counts / depths on the host.  Rate = records / (pass 1 + pass 2)."""
    import ctypes as C
    import tempfile
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from seeksv_amd import _abi, host, synth
    w = synth.Workload(genome_frac=args.file_frac, depth=args.depth, n_sv=max(1, round(args.n_sv * args.file_frac / args.genome_frac)), qual_model=1 if args.qual_alphabet == "hiseq40" else 0)  # the same density of planted junctions
    need = int(w.n_total * 110)  # the file is 72-93 bytes a record
    def room(path):
        try:
            st = os.statvfs(path)
            return st.f_bavail * st.f_frsize
        except OSError:
            return 0
    where = next((p for p in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(p) and room(p) > need), None)
    if where is None:
        raise RuntimeError(f"no room for a {need >> 20} MB BAM file in /dev/shm or {tempfile.gettempdir()}")
    d = tempfile.mkdtemp(prefix="ssv_file_", dir=where)
    try:
        t0 = time.perf_counter()
        # EVERY record carries its bases and qualities (reference bases at the record's position with 0.2 % substitutions, qualities
        # from {2, 11, 25, 37, 40} with fixed weights), deflate level 6 like samtools: ~72 B/record compressed, ~275 B/record inflated.
        # Records are generated by a pool of host threads (a C loop, GIL released) that runs ahead of the writer (which serialises and
        # deflates a batch on all cores): a bounded queue keeps at most ~2 x cores batches of 0.5 M records (~120 MB each) in host memory.
        bam = os.path.join(d, "sample.bam")
        write_workload_bam(w, bam, args.file_level)
        bam_bytes = os.path.getsize(bam)
        make_s = time.perf_counter() - t0
        # the file's BGZF blocks into pinned host memory, in chunks of ~2 GB of inflated data
        hl = _abi.host_lib()
        chunks = []
        t_read = time.perf_counter()
        with host.BamReader(bam) as r:
            first = C.c_uint64()
            if hl.ssvh_bam_raw_begin(r.handle, C.byref(first)) != 0:
                raise IOError(hl.ssvh_last_error().decode())
            n_targets = len(r.target_names)
            # chunks of ~2 GB of inflated data (about 33 K BGZF blocks, 0.6 GB of file each): both inflate passes take a wavefront per block and run at
            # the same rate whatever the chunk's size (round 3: 5 GB chunks, which the lane-per-block kernels needed to fill the chip)
            chunk_inflated = 2 << 30
            max_blocks = 1 << 17
            cap = min(bam_bytes + (1 << 20), 3 << 28)
            while True:
                buf = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
                blocks = (_abi.BgzfBlock * max_blocks)()
                nb, nbytes = C.c_int64(), C.c_size_t()
                if hl.ssvh_bam_read_blocks(r.handle, C.c_void_p(buf.data_ptr()), cap, chunk_inflated, blocks, max_blocks, C.byref(nb), C.byref(nbytes)) != 0:
                    raise IOError(hl.ssvh_last_error().decode())
                if nb.value == 0:
                    break
                chunks.append((buf, blocks, nb.value, nbytes.value))
        read_s = time.perf_counter() - t_read
        blk_dt = np.dtype([("c_off", np.uint64), ("c_len", np.uint32), ("u_len", np.uint32)])
        inflated_total = sum(int(np.frombuffer(blk, dtype=blk_dt, count=nb_)["u_len"].sum()) for _, blk, nb_, _ in chunks)
        hdr = host.Header(w.names, w.lens)
        jtable = host.JunctionTable(w.junctions)
        lib = ctx._lib

        class Preloaded:
            """the file's chunks as they lie in pinned host memory already (read before the timed region): round 3's footing, kept as `from_pinned`"""
            def open(self):
                return self
            def get(self, k):
                return chunks[k] if k < len(chunks) else None
            def ready(self, k):
                return k < len(chunks)
            def release(self, k):
                pass
            def close(self):
                pass

        class Streamed:
            """the file read INSIDE the timed region, the way `seeksv`'s BatchSource reads it: a thread runs up to three chunks ahead of the decoder, every
            chunk pread by all host threads (ssvh_bam_read_blocks) into one of three pinned buffers; a slot is the reader's again when its chunk is decoded"""
            NS = 3

            def __init__(self):
                # (like the preloaded chunks: allocated outside the timed region; from the library's own allocator - page-locked huge pages on the GPU's NUMA node, what the CLI's
                # staging buffers are since round 6 - so that the leg's copies do not depend on which socket the pages happened to land on)
                from seeksv_amd.device import PinnedArrays
                self.pinned = PinnedArrays()

                class Buf:
                    def __init__(self, a):
                        self.a = a

                    def data_ptr(self):
                        return self.a.ctypes.data
                self.bufs = [Buf(self.pinned.empty(cap, np.uint8)) for _ in range(self.NS)]
                self.blocks = [(_abi.BgzfBlock * max_blocks)() for _ in range(self.NS)]

            def open(self):
                self.cv = threading.Condition()
                self.produced, self.released, self.stop, self.err, self.slots = 0, 0, False, None, [None] * self.NS
                self.reader = host.BamReader(bam)
                fo = C.c_uint64()
                if hl.ssvh_bam_raw_begin(self.reader.handle, C.byref(fo)) != 0:
                    raise IOError(hl.ssvh_last_error().decode())
                self.th = threading.Thread(target=self._run)
                self.th.start()
                return self

            def _run(self):
                k = 0
                while True:
                    with self.cv:
                        self.cv.wait_for(lambda: self.stop or k - self.released < self.NS)
                        if self.stop:
                            return
                    buf, blocks = self.bufs[k % self.NS], self.blocks[k % self.NS]
                    nb, nbytes = C.c_int64(), C.c_size_t()
                    rc = hl.ssvh_bam_read_blocks(self.reader.handle, C.c_void_p(buf.data_ptr()), cap, chunk_inflated, blocks, max_blocks, C.byref(nb), C.byref(nbytes))
                    with self.cv:
                        if rc != 0:
                            self.err = hl.ssvh_last_error().decode()
                        self.slots[k % self.NS] = (buf, blocks, nb.value, nbytes.value) if rc == 0 and nb.value else None
                        self.produced = k + 1
                        self.cv.notify_all()
                    if rc != 0 or nb.value == 0:
                        return
                    k += 1

            def get(self, k):
                with self.cv:
                    self.cv.wait_for(lambda: self.produced > k)
                    if self.err:
                        raise IOError(self.err)
                    return self.slots[k % self.NS]

            def ready(self, k):
                with self.cv:
                    return self.produced > k and self.slots[k % self.NS] is not None and not self.err

            def release(self, k):
                with self.cv:
                    self.released = k + 1
                    self.cv.notify_all()

            def close(self):
                with self.cv:
                    self.stop = True
                    self.cv.notify_all()
                self.th.join()
                ctx._check(lib.ssv_bamdec_prefetch_drop(ctx._h), "ssv_bamdec_prefetch_drop")
                self.reader.close()

        def decoded_chunks(source, keep_all_seq=0):
            """every chunk of the file through the device decoder, in order; chunk k+1 is announced (ssv_bamdec_prefetch: its compressed bytes start for the GPU)
            before chunk k is decoded whenever it is there already"""
            src = source.open()
            try:
                k, announced = 0, -1
                c = src.get(0)
                while c is not None:
                    if announced < k + 1 and src.ready(k + 1):
                        nbuf, _, _, nnbytes = src.get(k + 1)
                        ctx._check(lib.ssv_bamdec_prefetch(ctx._h, C.c_void_p(nbuf.data_ptr()), nnbytes), "ssv_bamdec_prefetch")
                        announced = k + 1
                    buf, blocks, nb, nbytes = c
                    b = _abi.Batch()
                    ctx._check(lib.ssv_bamdec_decode(ctx._h, C.c_void_p(buf.data_ptr()), nbytes, blocks, nb, keep_all_seq, C.byref(b)), "ssv_bamdec_decode")
                    yield b     # (valid until the next decode; the chunk's slot stays the decoder's until the consumer comes back)
                    src.release(k)
                    k += 1
                    c = src.get(k)
            finally:
                src.close()

        def end_of_input():
            b = _abi.Batch()
            ctx._check(lib.ssv_bamdec_decode(ctx._h, None, 0, None, 0, 0, C.byref(b)), "ssv_bamdec_decode")

        def one_run(single_decode=True, source=None):
            """single_decode: every chunk is inflated and decoded ONCE; its records stay in HBM (ssv_batch_retain: 80 B/record) and the getsv
            passes scan them there.  False: the two-command shape of the reference - getclip reads the file, getsv reads it again.
            source: where the chunks come from (Preloaded / Streamed: the file read inside the timed region)"""
            source = source or preloaded
            t = {}
            t0 = time.perf_counter()
            kept = []
            try:
                # ---- pass 1: seeksv getclip ----
                ctx._check(lib.ssv_bamdec_begin(ctx._h, n_targets, first.value), "ssv_bamdec_begin"); ctx.bamdec_target_lens(w.lens)
                ctx.clip_begin(0.9, 1, False, None, 0)
                n = 0
                lap = {"decode(+reader wait)": 0.0, "retain": 0.0, "clip_scan": 0.0}   # SSV_BENCH_TRACE: where pass 1's wall time goes, summed over the chunks
                tl = time.perf_counter()
                for b in decoded_chunks(source):
                    n += b.n
                    tn = time.perf_counter(); lap["decode(+reader wait)"] += tn - tl; tl = tn
                    if single_decode:
                        b = ctx.batch_retain(b)
                        kept.append(b)
                        tn = time.perf_counter(); lap["retain"] += tn - tl; tl = tn
                    ctx.clip_scan(b)
                    tn = time.perf_counter(); lap["clip_scan"] += tn - tl; tl = tn
                end_of_input()
                if os.environ.get("SSV_BENCH_TRACE"):
                    print("[bench trace] file leg pass 1 (single_decode %s, %s):" % (single_decode, type(source).__name__), {k: round(v, 3) for k, v in lap.items()}, file=sys.stderr)
                nc, ne = ctx.clip_cluster_async()
                tab = ctx.clip_table_wait()
                if tab.format == 3:
                    ctx.clip_table_expand(tab, 0)
                ssum = int(np.ctypeslib.as_array(tab.support, shape=(tab.n_clusters,)).sum()) if tab.n_clusters else 0
                assert ssum == tab.n_events == ne and n == w.n_total
                t["getclip_s"] = time.perf_counter() - t0
                # ---- pass 2: seeksv getsv (insert size on the file's first records, then discordant pairs + depth of every record) ----
                t1 = time.perf_counter()
                ctx._check(lib.ssv_isize_begin(ctx._h, 20, 5000000), "ssv_isize_begin")
                done, used = C.c_int32(0), 0
                head = []   # two_reads: the batches the insert-size pass decoded, kept for the scan (like the CLI's IsizeCarry)
                stream = None
                if single_decode:
                    while used < len(kept) and not done.value:
                        ctx._check(lib.ssv_isize_accumulate(ctx._h, C.byref(kept[used]), C.byref(done)), "ssv_isize_accumulate")
                        used += 1
                else:
                    # CalculateInsertsizeDeviation reads the file until it has its 5,000,000 pairs (cluster.cpp:68): the first chunk or the first few; their
                    # batches stay (ssv_batch_retain) and the fused scan goes on from the chunk behind them - the second reading of the file is ONE reading
                    ctx._check(lib.ssv_bamdec_begin(ctx._h, n_targets, first.value), "ssv_bamdec_begin"); ctx.bamdec_target_lens(w.lens)
                    stream = decoded_chunks(source)
                    for b0 in stream:
                        ctx._check(lib.ssv_isize_accumulate(ctx._h, C.byref(b0), C.byref(done)), "ssv_isize_accumulate")
                        head.append(ctx.batch_retain(b0))
                        kept.append(head[-1])
                        if done.value:
                            break
                npairs, mean, sd = C.c_int64(), C.c_int32(0), C.c_int32(0)
                ctx._check(lib.ssv_isize_finish(ctx._h, C.byref(npairs), C.byref(mean), C.byref(sd)), "ssv_isize_finish")
                npairs, mean, sd = npairs.value, mean.value, sd.value
                plan = host.Plan(hdr, jtable, mean, sd)
                ctx.getsv_begin(plan.junctions, plan.windows, mean, sd, hdr.target_lens, 4, 20, 20)
                if single_decode:
                    for b in kept:
                        ctx.getsv_scan(b)
                else:
                    for b in head:
                        ctx.getsv_scan(b)
                    for b in stream:
                        ctx.getsv_scan(b)
                    end_of_input()
                counts, rs, pd, max_depth = ctx.getsv_finish(plan.ranges, plan.points)
                folded = plan.fold(counts, rs, pd)
                plan.close()
            finally:
                for b in kept:
                    ctx.batch_release(b)
            t["getsv_s"] = time.perf_counter() - t1
            t["total_s"] = time.perf_counter() - t0
            t["result"] = dict(n_clusters=int(nc), n_events=int(ne), mean=int(mean), sd=int(sd), pairs_used=int(npairs), abnormal_sum=int(folded["abnormal"].sum()),
                               depth_sum=int(folded["up_depth"].sum() + folded["down_depth"].sum()), max_depth=int(max_depth))
            return t

        preloaded = Preloaded()
        one_run()                      # warm-up: buffers grow to size
        ctx.prof_reset(); ctx.prof_enable(1)
        runs = [one_run() for _ in range(3)]
        prof = ctx.prof_all()
        ctx.prof_enable(0)
        pinned_best = min(runs, key=lambda t: t["total_s"])
        pinned_two = min((one_run(False) for _ in range(2)), key=lambda t: t["total_s"])   # the reference's shape: each command reads the file
        assert pinned_two["result"] == pinned_best["result"]
        # ... and with the FILE READ inside the timed region: the reader thread preads the file out of the page cache while the GPU decodes
        streamed = Streamed()
        one_run(True, streamed)
        best = min((one_run(True, streamed) for _ in range(3)), key=lambda t: t["total_s"])
        two = min((one_run(False, streamed) for _ in range(2)), key=lambda t: t["total_s"])
        assert best["result"] == pinned_best["result"] and two["result"] == best["result"]
        streamed.bufs = []
        streamed.pinned.close()
        inflated = None
        kernel_ms = {k: round(v["total_ms"] / len(runs), 3) for k, v in prof.items() if v["launches"]}
        coder = "the repository's fast deflate coder (single-probe LZ77 + Huffman, huff_gz.h)" if args.file_level == -2 else "deflate level " + str(args.file_level)
        out = {"value": w.n_total / best["total_s"], "unit": "records/s",
               "workload": f"synthetic {args.depth:g}x WGS, genome_frac {args.file_frac:g}: {w.n_total} records, every one with its bases (reference + 0.2 % substitutions) and qualities "
                           f"({'forty values 2..41' if args.qual_alphabet == 'hiseq40' else 'from {2,11,25,37,40}'}), as a BAM file of {bam_bytes} bytes = {bam_bytes / w.n_total:.1f} B/record (BGZF, {coder}, written by libseeksv_host; "
                           f"{inflated_total / w.n_total:.1f} B/record inflated), read in {len(chunks)} chunks of whole BGZF blocks",
               "records": w.n_total, "bam_bytes": bam_bytes, "bam_bytes_per_record": round(bam_bytes / w.n_total, 2), "inflated_bytes_per_record": round(inflated_total / w.n_total, 2), "chunks": len(chunks),
               "includes_file_read": True,
               "coder": {"name": "deflate_fast (huff_gz.h: single-probe greedy LZ77 + Huffman)" if args.file_level == -2 else "zlib", "level": int(args.file_level),
                         "note": "not samtools' level 6 unless level says 6: fewer and shorter matches inflate slower per output byte; inflate_level6 has the same records at level 6"},
               "timed_region": "opening the file, then per pass and chunk: the chunk pread out of the page cache (/dev/shm) into one of three pinned buffers by a reader thread that runs ahead "
                               "(ssvh_bam_read_blocks: all host threads, block headers scanned there) -> H2D of the compressed bytes (the next chunk's announced ahead) -> device BGZF inflate -> BAM record "
                               "decode -> scans -> tables / tallies on the host; NOT in it: file creation, allocating the three pinned buffers",
               "getclip_s": round(best["getclip_s"], 4), "getsv_s": round(best["getsv_s"], 4), "total_s": round(best["total_s"], 4),
               "runs_total_s": [round(t["total_s"], 4) for t in runs],
               "pcie_in_GBs": round(bam_bytes / best["total_s"] / 1e9, 2),   # (the compressed bytes cross PCIe once)
               "kernel_ms_per_run": kernel_ms, "result": best["result"],
               "two_reads": {"total_s": round(two["total_s"], 4), "getclip_s": round(two["getclip_s"], 4), "getsv_s": round(two["getsv_s"], 4), "value": w.n_total / two["total_s"],
                             "what": "the same with the file read, inflated and decoded twice, once per command like the reference (seeksv.cpp:128,157): same counts"},
               "from_pinned": {"value": w.n_total / pinned_best["total_s"], "total_s": round(pinned_best["total_s"], 4), "getclip_s": round(pinned_best["getclip_s"], 4), "getsv_s": round(pinned_best["getsv_s"], 4),
                               "two_reads_value": w.n_total / pinned_two["total_s"], "two_reads_total_s": round(pinned_two["total_s"], 4), "includes_file_read": False,
                               "what": "round 3's footing: the whole file in pinned host memory before the clock starts (reading it there took %.1f s, outside); kernel_ms_per_run was measured on these runs" % read_s},
               "note": "the file is inflated and decoded ONCE: the decoded records (80 B each) stay in HBM (ssv_batch_retain) and the getsv passes scan them there - the reference reads the file once per command; file creation (%.1f s) is outside the timed region" % make_s}
        hdr.close()
        if args.file_level != 6:
            try:
                out["inflate_level6"] = inflate_level6_leg(ctx, args)
            except Exception as e:
                out["inflate_level6"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cli_leg:
            try:
                out["cli_path"] = cli_path_leg(args, w, bam, d, out["result"])
            except Exception as e:
                out["cli_path"] = {"error": f"{type(e).__name__}: {e}"}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def inflate_level6_leg(ctx, args):
    """The device inflate on a file as samtools writes it (zlib level 6: longer matches, fewer literals than the fast coder the whole-sample file leg has to use where 16
    CPUs write it): a slice of the same records, its chunks in pinned memory, decode only - the inflate kernels' rate in bytes of OUTPUT, and the PCIe floor beside it."""
    import ctypes as C
    import shutil
    import tempfile
    import torch
    from seeksv_amd import _abi, host, synth
    frac = args.genome_frac / 128
    w = synth.Workload(genome_frac=frac, depth=args.depth, n_sv=max(1, round(args.n_sv / 128)), qual_model=1 if args.qual_alphabet == "hiseq40" else 0)
    where = room_for(int(w.n_total * 110))
    d = tempfile.mkdtemp(prefix="ssv_l6_", dir=where)
    try:
        bam = os.path.join(d, "l6.bam")
        t0 = time.perf_counter()
        write_workload_bam(w, bam, 6)
        make_s = time.perf_counter() - t0
        os.environ.pop("SSV_BGZF_LEVEL", None)
        hl, lib = _abi.host_lib(), ctx._lib
        chunks = []
        with host.BamReader(bam) as r:
            first = C.c_uint64()
            if hl.ssvh_bam_raw_begin(r.handle, C.byref(first)) != 0:
                raise IOError(hl.ssvh_last_error().decode())
            n_targets = len(r.target_names)
            cap, max_blocks = min(os.path.getsize(bam) + (1 << 20), 3 << 28), 1 << 17
            while True:
                buf = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
                blocks = (_abi.BgzfBlock * max_blocks)()
                nb, nbytes = C.c_int64(), C.c_size_t()
                if hl.ssvh_bam_read_blocks(r.handle, C.c_void_p(buf.data_ptr()), cap, 1 << 30, blocks, max_blocks, C.byref(nb), C.byref(nbytes)) != 0:
                    raise IOError(hl.ssvh_last_error().decode())
                if nb.value == 0:
                    break
                chunks.append((buf, blocks, nb.value, nbytes.value))
        blk_dt = np.dtype([("c_off", np.uint64), ("c_len", np.uint32), ("u_len", np.uint32)])
        inflated = sum(int(np.frombuffer(blk, dtype=blk_dt, count=nb_)["u_len"].sum()) for _, blk, nb_, _ in chunks)

        def run():
            ctx._check(lib.ssv_bamdec_begin(ctx._h, n_targets, first.value), "ssv_bamdec_begin"); ctx.bamdec_target_lens(w.lens)
            n = 0
            for buf, blocks, nb, nbytes in chunks:
                b = _abi.Batch()
                ctx._check(lib.ssv_bamdec_decode(ctx._h, C.c_void_p(buf.data_ptr()), nbytes, blocks, nb, 0, C.byref(b)), "ssv_bamdec_decode")
                n += b.n
            b = _abi.Batch()
            ctx._check(lib.ssv_bamdec_decode(ctx._h, None, 0, None, 0, 0, C.byref(b)), "ssv_bamdec_decode")
            assert n == w.n_total
        run()
        ctx.prof_reset(); ctx.prof_enable(1)
        REPS = 3
        for _ in range(REPS):
            run()
        prof = ctx.prof_all()
        ctx.prof_enable(0)
        ms = {k: prof[k]["total_ms"] / REPS for k in ("bam_inflate", "bam_resolve", "bam_records", "bam_decode", "bam_upload") if prof.get(k, {}).get("launches")}
        size = os.path.getsize(bam)
        return {"coder": {"name": "zlib", "level": 6}, "records": w.n_total, "bam_bytes": size, "bam_bytes_per_record": round(size / w.n_total, 2), "inflated_bytes": inflated,
                "kernel_ms": {k: round(v, 3) for k, v in ms.items()}, "inflate_GBs_of_output": round(inflated / (ms["bam_inflate"] * 1e-3) / 1e9, 1),
                "whole_sample_at_this_rate": {"file_GB": round(size / w.n_total * args.genome_frac / frac * w.n_total / 1e9, 1), "pcie_floor_s_at_56_GBs": round(size / frac * args.genome_frac / 56e9, 3),
                                              "inflate_s": round(ms["bam_inflate"] * 1e-3 / frac * args.genome_frac, 3), "decode_s": round((ms.get("bam_records", 0) + ms.get("bam_decode", 0)) * 1e-3 / frac * args.genome_frac, 3)},
                "what": f"genome_frac {frac:g} of the same sample written with zlib level 6 like samtools ({make_s:.1f} s), {len(chunks)} chunk(s) out of pinned memory through ssv_bamdec_decode only, mean of {REPS} runs; "
                        "bam_inflate holds both inflate passes (bam_resolve: pass 2's share)"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cli_path_leg(args, w, bam, d, expect):
    """The product binary at scale (north_star: "a C++ host that keeps the seeksv getclip / seeksv getsv CLI"): `seeksv getclip -Z` and `seeksv getsv -Z -B`
    as child processes on the file leg's BAM (in /dev/shm), wall clock from exec to exit - process start, HIP context, reading the file, device
    inflate + decode, kernels, row formatting, gzip, every output file.  Junctions come in through the reference's own -B harness (the external
    re-alignment step between the two commands is not part of the path)."""
    import gzip
    import subprocess
    from seeksv_amd import host
    exe = os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
    env = dict(os.environ, SSV_TIMING="1")
    env.pop("SSV_BGZF_LEVEL", None)   # (the file leg's own BAM writer: not what the commands write their clip.bam with)

    phases = timing_phases
    jfile = os.path.join(d, "junctions.txt")
    junction_file(jfile, w.junctions)
    empty_bam = os.path.join(d, "empty.clip.bam")
    host.write_bam(empty_bam, w.names, w.lens, [])
    empty_clip = os.path.join(d, "empty.clip")
    open(empty_clip, "w").close()
    res = {}
    best = None
    for rep in range(2):  # the second run finds the binary, the libraries and the file's pages warm; the better one is reported
        t0 = time.perf_counter()
        r1, c1 = run_command([exe, "getclip", "-Z", "-o", os.path.join(d, "cli"), bam], env)
        t1 = time.perf_counter()
        r2, c2 = run_command([exe, "getsv", "-Z", "-d", "0", "-f", "0", "-b", "0", "-B", jfile, empty_bam, bam, empty_clip, os.path.join(d, "cli.sv"), os.path.join(d, "cli.x.fq")], env)
        t2 = time.perf_counter()
        if r1.returncode != 0 or r2.returncode != 0:
            raise RuntimeError((r1.stderr + r2.stderr)[-400:])
        cur = dict(getclip_s=round(t1 - t0, 3), getsv_s=round(t2 - t1, 3), total_s=round(t2 - t0, 3), getclip_phases_s=c1["phases_s"], getsv_phases_s=c2["phases_s"],
                   process_s={"getclip": {k: c1.get(k) for k in ("exec_to_main_s", "exit_to_reaped_s", "cpu_s")}, "getsv": {k: c2.get(k) for k in ("exec_to_main_s", "exit_to_reaped_s", "cpu_s")},
                              "host_cpus": effective_cpus()})
        if best is None or cur["total_s"] < best["total_s"]:
            best = cur
    res.update(best)
    res["value"] = w.n_total / best["total_s"]
    res["unit"] = "records/s"
    # the whole pipeline in ONE process: `seeksv run` = getclip + the GPU re-aligner (in the place of the external `bwa mem` step) + getsv on the junctions it
    # finds itself; the BAM is inflated and decoded once and its records stay in HBM for the getsv passes
    try:
        fa = os.path.join(d, "ref.fa")
        w.write_fasta(fa, effective_cpus())
        rbest = None
        for rep in range(2):
            r, cur = run_command([exe, "run", bam, fa, os.path.join(d, "one")], env)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-400:])
            if rbest is None or cur["total_s"] < rbest["total_s"]:
                rbest = cur
        rows_sv = [l.split("\t") for l in open(os.path.join(d, "one.sv.txt")) if not l.startswith("@")]
        found = {(c[0], int(c[1]), c[2], c[4], int(c[5]), c[6]) for c in rows_sv}
        planted = {tuple(j[:6]) for j in w.junctions}
        rbest.update(value=w.n_total / rbest["total_s"], unit="records/s", sv_rows=len(rows_sv), planted=len(planted), planted_found=len(planted & found),
                     what="`seeksv run <bam> <ref.fa> <prefix>`: getclip -> re-aligner -> getsv in one process, one decode of the BAM, every output file of the three commands written")
        res["run"] = rbest
    except Exception as e:
        res["run"] = {"error": f"{type(e).__name__}: {e}"}
    # the same answers as the ABI path on the same file: one clip.gz row per cluster, the discordant pairs of the SV table
    rows = 0
    with gzip.open(os.path.join(d, "cli.clip.gz"), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            rows += chunk.count(b"\n")
    abnormal = sum(int(l.split("\t")[9]) for l in open(os.path.join(d, "cli.sv")) if l and not l.startswith("@"))
    abnormal += sum(int(l.split("\t")[10]) for l in r2.stdout.splitlines() if l.count("\t") >= 15)  # junctions the filter chain sent to stdout (reason + the 15 fields)
    res["clip_rows"], res["abnormal_sum"] = rows, abnormal
    res["same_result_as_abi_path"] = bool(rows == expect["n_clusters"] and abnormal == expect["abnormal_sum"])
    res["outputs_bytes"] = {n: os.path.getsize(os.path.join(d, n)) for n in ("cli.clip.gz", "cli.clip.fq.gz", "cli.sv")}
    res["what"] = "`seeksv getclip -Z` + `seeksv getsv -Z -B <planted junctions>` as child processes on the file leg's BAM in /dev/shm, wall clock exec to exit (two whole-file reads, two HIP contexts)"
    return res


def config5_path_leg(args):
    """BASELINE config 5 through the PRODUCT: a human + HBV hybrid reference, tumor at twice the depth with planted virus integrations against its normal, 2 % of the
    records in pairs with one unmapped end (what getclip's unmapped-pair side channel is for): `seeksv run` on the tumor (getclip -> re-aligner -> getsv), `seeksv getclip`
    on the normal, `seeksv somatic` (the tumor's table against the normal's clusters and BAM) - child processes, wall clock exec to exit, SSV_TIMING phases.
    Reference: somatic.h:40-70, somatic.cpp:14-427 (tumor rows against the normal), clip_reads.h:172-219 (the side channel)."""
    import shutil
    import subprocess
    import tempfile
    from seeksv_amd import synth
    frac = args.config5_frac
    kw = dict(genome_frac=frac, n_sv=max(8, round(args.n_sv * frac)), hbv=True, unmap_permille=int(round(args.unmapped_frac * 1000)), qual_model=1 if args.qual_alphabet == "hiseq40" else 0)
    tumor = synth.Workload(depth=2 * args.depth, n_integrations=args.n_integrations, **kw)
    normal = synth.Workload(depth=args.depth, **kw)   # the same seed: the same reference, the same germline SVs, no virus
    need = int((tumor.n_total + normal.n_total) * 125 + tumor.genome_len * 1.1)
    where = room_for(need)
    if where is None:
        raise RuntimeError(f"no room for {need >> 20} MB of BAM files")
    d = tempfile.mkdtemp(prefix="ssv_c5_", dir=where)
    exe = os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
    env = dict(os.environ, SSV_TIMING="1")
    try:
        t0 = time.perf_counter()
        tbam, nbam, fa = os.path.join(d, "tumor.bam"), os.path.join(d, "normal.bam"), os.path.join(d, "ref.fa")
        # the files' coder: what --file-level says; by itself the repository's fast coder where 16 CPUs write (zlib level 6 took 27 minutes for the real size's 1.85 G records), level 6 from 64 CPUs on
        level = args.config5_level
        write_workload_bam(tumor, tbam, level)
        write_workload_bam(normal, nbam, level)
        tumor.write_fasta(fa, effective_cpus())
        make_s = time.perf_counter() - t0
        env.pop("SSV_BGZF_LEVEL", None)
        best = None
        for rep in range(2):   # (the second run finds the binary, the libraries and the files' pages warm; the better one is reported)
            cur = {}
            ta = time.perf_counter()
            r1 = subprocess.run([exe, "run", tbam, fa, os.path.join(d, "T")], capture_output=True, text=True, env=env)
            tb = time.perf_counter()
            if r1.returncode != 0:
                raise RuntimeError("seeksv run: " + r1.stderr[-400:])
            r2 = subprocess.run([exe, "getclip", "-Z", "-o", os.path.join(d, "N"), nbam], capture_output=True, text=True, env=env)
            tc = time.perf_counter()
            if r2.returncode != 0:
                raise RuntimeError("seeksv getclip: " + r2.stderr[-400:])
            r3 = subprocess.run([exe, "somatic", "-Z", nbam, os.path.join(d, "N.clip.gz"), os.path.join(d, "T.sv.txt"), os.path.join(d, "T.somatic.sv")], capture_output=True, text=True, env=env)
            td = time.perf_counter()
            if r3.returncode != 0:
                raise RuntimeError("seeksv somatic: " + r3.stderr[-400:])
            cur = dict(total_s=round(td - ta, 3), run_tumor_s=round(tb - ta, 3), getclip_normal_s=round(tc - tb, 3), somatic_s=round(td - tc, 3),
                       run_tumor_phases_s=timing_phases(r1.stderr), getclip_normal_phases_s=timing_phases(r2.stderr), somatic_phases_s=timing_phases(r3.stderr),
                       somatic_detail=[l[9:] for l in r3.stderr.splitlines() if l.startswith("[timing] (")], getclip_detail=[l[9:] for l in r2.stderr.splitlines() if l.startswith("[timing] (")])
            if best is None or cur["total_s"] < best["total_s"]:
                best = cur
        rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(d, "T.somatic.sv")) if not l.startswith("@")]
        planted = {tuple(j[:6]) for j in tumor.junctions}
        viral = {j for j in planted if "HBV" in (j[0], j[3])}
        found = {(c[0], int(c[1]), c[2], c[4], int(c[5]), c[6]): c for c in rows}
        somatic = {k for k, c in found.items() if c[23] == "0" and c[24] == "0" and c[25] == "0"}   # example/seeksv.somatic.sh:5-6: awk '$24==0 && $25==0 && $26==0'
        import gzip
        un_lines = 0
        with gzip.open(os.path.join(d, "N.unmapped_1.fq.gz"), "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                un_lines += chunk.count(b"\n")
        n_rows = int(next((x.split(" rows")[0].split(": ")[-1] for x in best["somatic_detail"] if x.startswith("(normal clusters:")), 0) or 0)
        out = dict(best)
        out.update(records=tumor.n_total + normal.n_total, value=(tumor.n_total + normal.n_total) / best["total_s"], unit="records/s",
                   workload=f"BASELINE config 5: human + HBV hybrid reference (genome_frac {frac:g}, {len(tumor.names)} contigs), tumor {2 * args.depth:g}x = {tumor.n_total} records with {args.n_integrations} planted "
                            f"virus integrations and {len(planted) - len(viral)} DEL/INV/TRA, normal {args.depth:g}x = {normal.n_total} records with the same germline SVs, {args.unmapped_frac * 100:g} % of the records in pairs with one unmapped end; "
                            f"BAM files of {os.path.getsize(tbam)} + {os.path.getsize(nbam)} bytes",
                   sv_rows=len(rows), planted=len(planted), planted_found=len(planted & set(found)), viral_planted=len(viral), viral_found=len(viral & set(found)), viral_somatic=len(viral & somatic),
                   germline_found=len((planted - viral) & set(found)), germline_called_somatic=len((planted - viral) & somatic),
                   normal_cluster_rows=n_rows, normal_unmapped_pairs_written=un_lines // 4, files_made_s=round(make_s, 1),
                   what="`seeksv run tumor.bam ref.fa T` + `seeksv getclip -Z -o N normal.bam` + `seeksv somatic -Z normal.bam N.clip.gz T.sv.txt T.somatic.sv`: child processes, wall clock exec to exit, "
                        "every output file written; file creation outside")
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def ranks_path_leg(args, world, force_device):
    """N > 1: the PRODUCT's own multi-GPU path on a file - `seeksv getclip -Z -N <world>` + `seeksv getsv -Z -N <world> -B <planted junctions>`: the BAM cut into <world> runs
    of records (bam_partition.cpp), one rank thread and one GPU per run, the partial tallies / depth sums through ssv_group_allgather = ONE ncclAllGather over xGMI issued
    from the host library (group_api.inc; sums: getsv.cpp:1116, bam2depth.cpp:101-124) - beside the same two commands on one GPU.  Run by rank 0 once the resident leg is
    over and every rank has given its device memory back.  (The resident leg's exchange is torch.distributed's all_gather over the same RCCL: one process per GPU, as
    the bench contract asks; the product's ranks are threads of one process.)"""
    import shutil
    import subprocess
    import tempfile
    from seeksv_amd import host, synth
    frac = args.genome_frac * args.ranks_frac
    w = synth.Workload(genome_frac=frac, depth=args.depth, n_sv=max(8, round(args.n_sv * args.ranks_frac)), qual_model=1 if args.qual_alphabet == "hiseq40" else 0)
    where = room_for(int(w.n_total * 125))
    if where is None:
        raise RuntimeError("no room for the BAM file")
    d = tempfile.mkdtemp(prefix="ssv_ranks_", dir=where)
    exe = os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
    env = dict(os.environ, SSV_TIMING="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):   # (the commands are not ranks of this job)
        env.pop(k, None)
    try:
        t0 = time.perf_counter()
        bam = os.path.join(d, "sample.bam")
        write_workload_bam(w, bam, -2)
        make_s = time.perf_counter() - t0
        env.pop("SSV_BGZF_LEVEL", None)
        jfile, empty_bam, empty_clip = os.path.join(d, "junctions.txt"), os.path.join(d, "empty.clip.bam"), os.path.join(d, "empty.clip")
        junction_file(jfile, w.junctions)
        host.write_bam(empty_bam, w.names, w.lens, [])
        open(empty_clip, "w").close()
        dev = ["-G", str(force_device)] if force_device is not None else []

        def pair(tag, n):
            best = None
            for rep in range(2):
                ta = time.perf_counter()
                r1 = subprocess.run([exe, "getclip", "-Z", "-N", str(n)] + dev + ["-o", os.path.join(d, tag), bam], capture_output=True, text=True, env=env)
                tb = time.perf_counter()
                r2 = subprocess.run([exe, "getsv", "-Z", "-N", str(n)] + dev + ["-d", "0", "-f", "0", "-b", "0", "-B", jfile, empty_bam, bam, empty_clip, os.path.join(d, tag + ".sv"), os.path.join(d, tag + ".x.fq")],
                                    capture_output=True, text=True, env=env)
                tc = time.perf_counter()
                if r1.returncode != 0 or r2.returncode != 0:
                    raise RuntimeError((r1.stderr + r2.stderr)[-400:])
                cur = dict(ranks=n, getclip_s=round(tb - ta, 3), getsv_s=round(tc - tb, 3), total_s=round(tc - ta, 3), value=w.n_total / (tc - ta), getclip_phases_s=timing_phases(r1.stderr), getsv_phases_s=timing_phases(r2.stderr),
                           exchange=next((l[9:] for l in r2.stderr.splitlines() if l.startswith("[timing] exchange over")), None), stdout=r2.stdout)
                if best is None or cur["total_s"] < best["total_s"]:
                    best = cur
            return best
        many, one = pair("many", world), pair("one", 1)
        import gzip
        same = open(os.path.join(d, "many.sv")).read() == open(os.path.join(d, "one.sv")).read() and many.pop("stdout") == one.pop("stdout") and all(
            gzip.open(os.path.join(d, f"many.{e}")).read() == gzip.open(os.path.join(d, f"one.{e}")).read() for e in ("clip.gz", "clip.fq.gz"))
        return dict(value=many["value"], unit="records/s", records=w.n_total, ranks=many, one_rank=one, speedup=round(one["total_s"] / many["total_s"], 3), same_outputs_as_one_rank=bool(same), file_made_s=round(make_s, 1),
                    workload=f"synthetic {args.depth:g}x WGS, genome_frac {frac:g}: {w.n_total} records as a BAM file of {os.path.getsize(bam)} bytes, {len(w.junctions)} planted junctions through -B",
                    what=f"`seeksv getclip -Z -N {world}` + `seeksv getsv -Z -N {world} -B`: the product's range partition (halo at the cuts) and its ONE ssv_group_allgather (ncclAllGather over xGMI between "
                         f"different GPUs; host memory when the ranks share one), child processes of rank 0, wall clock exec to exit; beside them the same commands with -N 1")
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(w, hdr, n_sample, min_seconds=10.0):
    """The CPU oracle (plain-C restatement of the reference, oracle/) on a bounded prefix of the same workload, 1 thread:
    whole passes (getclip + insert size + discordant + depth) are repeated until >= min_seconds of CPU work have been timed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from seeksv_amd import host
    b = w.generate_host(0, n_sample)
    passes, t_clip = 0, 0.0
    t0 = time.perf_counter()
    while True:
        t1 = time.perf_counter()
        d = O.getclip([b])
        t_clip += time.perf_counter() - t1
        rc, npairs, mean, sd = O.isize_stats([b], 20, 5000000)
        plan = host.Plan(hdr, w.junctions, mean, sd)
        O.discordant([b], plan.junctions, mean, sd, 4, 20)
        O.depth([b], plan.windows, plan.ranges, plan.points, 20)
        plan.close()
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or passes >= 64:
            break
    return {"value": n_sample * passes / dt, "unit": "records/s", "cores": 1, "kind": "port",
            "sample": f"{passes} passes over the first {n_sample} records of the same synthetic workload = {dt:.1f} s of CPU work (getclip {t_clip:.1f} s); "
                      "oracle = plain-C restatement of seeksv v1.2.3 on decoded SoA records (no BGZF inflate / BAM parse), pinned to the real reference on tests/golden",
            "clusters": int(d["n_clusters"]), "events": int(d["n_events"])}


def cpu_baseline_all_cores(args, w, seconds=6.0, n_per_worker=2_000_000):
    """The same oracle on every host core at once: one single-threaded worker process per core (tools/cpu_baseline_worker.py, started as
    plain child processes that never touch the GPU), each on its own slice of the workload, range-partitioned like the multi-GPU run.
    Aggregate records/s - what the host of this GPU box could do with the reference's algorithm if it were parallelised by contig range."""
    import subprocess
    cores = effective_cpus()   # (a container that sees 256 CPUs may be granted the time of 16: that many workers, and `cores` says so)
    n = min(n_per_worker, max(1, w.n_total // cores))
    worker = os.path.join(ROOT, "tools", "cpu_baseline_worker.py")
    try:
        procs = [subprocess.Popen([sys.executable, worker, repr(args.genome_frac), repr(w.depth), str(args.n_sv), str(k * (w.n_total // cores)), str(n), str(seconds)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
                 for k in range(cores)]
        rate, ok = 0.0, 0
        for p in procs:
            out, _ = p.communicate(timeout=120)
            if p.returncode == 0 and out.strip():
                nn, passes, dt = out.split()
                rate += int(nn) * int(passes) / float(dt)
                ok += 1
        if not ok:
            return None
        return {"value": rate, "unit": "records/s", "cores": ok, "kind": "port",
                "sample": f"{ok} single-threaded oracle processes at once (the CPUs this process may use: {os.cpu_count()} visible, cgroup quota / affinity {cores}), each {seconds:g} s of whole passes over its own {n}-record slice of the same workload"}
    except Exception as e:  # a courtesy, like cpu_reference
        return {"value": None, "unit": "records/s", "cores": cores, "kind": "port", "sample": f"failed: {e}"}


def cpu_reference(w, n_sample):
    """The REAL reference (oracle/_ref/seeksv_ref, built from the reference's own sources by `make -C oracle ref` where they exist; the binary
    travels with the snapshot) on a bounded prefix of the same workload written as a BAM file: `getclip`, then `getsv -B` (insert size +
    discordant tally + depth for the planted junctions that fall into the sample).  File to file, so unlike `cpu_baseline` it includes
    the BGZF inflate + BAM parse the reference spends most of its time in.  None when the binary is not there."""
    import shutil
    import subprocess
    import tempfile
    from seeksv_amd import host
    ref, bamidx = os.path.join(ROOT, "oracle", "_ref", "seeksv_ref"), os.path.join(ROOT, "oracle", "_ref", "bamidx")
    if n_sample <= 0 or not (os.path.exists(ref) and os.path.exists(bamidx)):
        return None
    d = tempfile.mkdtemp(prefix="ssv_ref_")
    try:
        b = w.generate_host(0, n_sample)
        bam = os.path.join(d, "sample.bam")
        host.write_bam(bam, w.names, w.lens, [b])
        subprocess.run([bamidx, bam], check=True, capture_output=True)
        last_tid, last_pos = int(b["tid"][-1]), int(b["pos"][-1])
        rows = [j for j in w.junctions if (w.names.index(j[0]), j[1]) <= (last_tid, last_pos)][:1000]
        jfile = os.path.join(d, "junctions.txt")
        with open(jfile, "w") as f:
            for j in rows:
                f.write("\t".join(str(x) for x in (j[0], j[1], j[2], 0, j[3], j[4], j[5], 0, 0, 0, "NA", 0, 0, 0, 0, 0, 0, 0, 0, "50M", "50M", "ACGT", "ACGT")) + "\n")
        empty_bam, empty_clip = os.path.join(d, "empty.clip.bam"), os.path.join(d, "empty.clip")
        host.write_bam(empty_bam, w.names, w.lens, [])
        open(empty_clip, "w").close()
        t0 = time.perf_counter()
        subprocess.run([ref, "getclip", "-o", os.path.join(d, "ref"), bam], check=True, capture_output=True)
        t1 = time.perf_counter()
        subprocess.run([ref, "getsv", "-d", "0", "-f", "0", "-b", "0", "-B", jfile, empty_bam, bam, empty_clip, os.path.join(d, "ref.sv"), os.path.join(d, "x.fq")], check=True, capture_output=True)
        t2 = time.perf_counter()
        return {"value": n_sample / (t2 - t0), "unit": "records/s", "cores": 1, "kind": "reference", "junctions_served": len(rows),
                "note": f"its getsv leg serves the {len(rows)} planted junctions that fall into its sample, the GPU legs all {len(w.junctions)}: a rate beside the others, not against them",
                "sample": f"seeksv v1.2.3 binary on a BAM of the first {n_sample} records of the same workload ({os.path.getsize(bam) >> 20} MB): getclip {t1 - t0:.2f} s + "
                          f"getsv -B with {len(rows)} junctions {t2 - t1:.2f} s, file to file (BGZF inflate and BAM parse included)"}
    except Exception as e:  # the baseline is a courtesy: never fail the bench for it
        return {"value": None, "unit": "records/s", "cores": 1, "kind": "reference", "sample": f"failed: {e}"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
