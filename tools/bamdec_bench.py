#!/usr/bin/env python3
"""Whole-file decode rates: the device-side BGZF inflate + BAM decode (ssv_bamdec_*) next to the multi-threaded host reader, on a synthetic
BAM written by this repository's writer.  usage: python tools/bamdec_bench.py [genome_frac] [depth] [chunk_inflated_GB]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seeksv_amd import device, host, synth  # noqa: E402


class _Total:
    def __init__(self, n):
        self.n_total = n


def main():
    d = tempfile.mkdtemp(prefix="ssv_bamdec_")
    bam = os.path.join(d, "synth.bam")
    if len(sys.argv) > 1 and sys.argv[1] == "example":
        # real reads, real base qualities: the records of the reference's bundled example BAM (tests/golden/example/cancer.sort.bam), written
        # K times over - what a lane sees inside one BGZF block (literal / match mix, Huffman code lengths) is then that of real data
        k = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
        chunk_gb = int(sys.argv[3]) if len(sys.argv) > 3 else 7
        with host.BamReader(os.path.join(ROOT, "tests", "golden", "example", "cancer.sort.bam")) as r:
            names, lens, b = r.target_names, r.target_lens, r.read_batch(1 << 22, keep_all_seq=True)
        host.write_bam(bam, names, lens, [b] * k)
        w = _Total(len(b["tid"]) * k)
        label = f"example/cancer.sort.bam x {k}"
    elif len(sys.argv) > 1 and sys.argv[1] == "real":
        # the synthetic sample with bases and qualities for EVERY record (reference bases + 0.2 % substitutions, qualities from {2, 11, 25, 37, 40},
        # SURVEY 8d), deflate level 6 like samtools: 72 B/record compressed, 330 B/record inflated - what bench.py's file leg reads
        from concurrent.futures import ThreadPoolExecutor
        frac = float(sys.argv[2]) if len(sys.argv) > 2 else 1 / 32
        chunk_gb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
        os.environ.setdefault("SSV_BGZF_LEVEL", "6")
        w = synth.Workload(genome_frac=frac, depth=30, n_sv=200)
        chunk = 1_000_000
        starts = list(range(0, w.n_total, chunk))
        nw = min(64, os.cpu_count() or 1)
        def batches():
            with ThreadPoolExecutor(max_workers=nw) as ex:
                for i in range(0, len(starts), nw):
                    yield from ex.map(lambda g: w.generate_host(g, min(chunk, w.n_total - g), all_seq=True), starts[i:i + nw])
        t0 = time.perf_counter()
        host.write_bam(bam, w.names, w.lens, batches())
        label = f"synthetic 30x with bases and qualities for every record, genome_frac {frac}, BGZF level {os.environ['SSV_BGZF_LEVEL']} (written in {time.perf_counter() - t0:.1f} s)"
    else:
        frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1 / 32
        depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30
        chunk_gb = int(sys.argv[3]) if len(sys.argv) > 3 else 7
        w = synth.Workload(genome_frac=frac, depth=depth, n_sv=200)
        chunk = 2_000_000
        host.write_bam(bam, w.names, w.lens, (w.generate_host(g, min(chunk, w.n_total - g)) for g in range(0, w.n_total, chunk)))
        label = f"synthetic 30x, genome_frac {frac}"
    out = {"input": label, "chunk_inflated_GB": chunk_gb, "records": w.n_total, "bam_bytes": os.path.getsize(bam), "host_cpus": os.cpu_count()}
    # host reader (all threads, read-ahead)
    for rep in range(2):
        t = time.perf_counter()
        n = 0
        with host.BamReader(bam, readahead=True) as r:
            import ctypes as C
            from seeksv_amd import _abi
            b = _abi.Batch()
            while True:
                if r._lib.ssvh_bam_read_batch(r.handle, 1 << 22, 0, C.byref(b)) != 0:
                    raise IOError(r._lib.ssvh_last_error().decode())
                if b.n == 0:
                    break
                n += b.n
        out["host_reader_s"] = round(time.perf_counter() - t, 3)
    assert n == w.n_total
    with device.Context(0) as ctx:
        for rep in range(3):
            ctx.prof_reset()
            ctx.prof_enable(1 if rep == 2 else 0)
            t = time.perf_counter()
            n = inflated = comp = repaired = chunks = 0
            with host.BamReader(bam) as r:
                for b, info in ctx.bam_batches(r, chunk_bytes=int(float(os.environ.get("SSV_CHUNK_COMP_GB", "1")) * (1 << 30)), max_blocks=1 << 18, chunk_inflated=chunk_gb << 30):
                    n += info["n_records"]; inflated += info["inflated_bytes"]; comp += info["compressed_bytes"]; repaired += info["repaired_blocks"]; chunks += 1
            dt = time.perf_counter() - t
            assert n == w.n_total
        prof = ctx.prof_all()
        out.update({"device_decode_s": round(dt, 3), "inflated_bytes": inflated, "chunks": chunks, "repaired_blocks": repaired,
                    "device_records_per_s": round(n / dt), "host_records_per_s": round(n / out["host_reader_s"]),
                    "kernel_ms": {k: round(v["total_ms"], 3) for k, v in prof.items() if k.startswith("bam_")}})
        ms = out["kernel_ms"]
        if ms.get("bam_inflate"):
            out["inflate_GBs_out"] = round(inflated / ms["bam_inflate"] / 1e6, 1)  # the inflate kernels (both passes)
            out["inflate_with_upload_GBs_out"] = round(inflated / (ms["bam_inflate"] + ms.get("bam_upload", 0.0)) / 1e6, 1)  # + the compressed bytes' H2D, not overlapped
        out["device_kernels_records_per_s"] = round(n / (sum(v for k, v in ms.items() if k != "bam_resolve") / 1e3)) if ms else None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
