#!/usr/bin/env python3
"""End-to-end, file to file: the `seeksv` CLI of this repository next to the REAL reference binary (oracle/_ref/seeksv_ref, built from
/root/reference by `make -C oracle ref` in the build container; it travels to the GPU box with the snapshot) on the same synthetic BAM.
Checks that the outputs are identical and prints the wall times.  usage: python tools/cli_vs_reference.py [genome_frac] [depth] [n_sv] [full] [unmapped permille] [virus integrations]
(round 6: pairs with one unmapped end - the side channel's two FASTQ files are compared too -, `seeksv run` against the three commands and the reference's table, and `seeksv somatic` with a normal sample of half the depth
against the real reference's somatic table)
With `full` the junction stage runs too (BASELINE config 3 at a size the reference finishes in minutes): `seeksv realign` (the GPU stand-in for
bwa mem; there is no bwa on the GPU box) makes ONE clip.bam, and both programs run `getsv clip.bam in.bam clip.gz` on it - the two SV tables
must be identical and must hold every planted junction."""
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seeksv_amd import host, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "seeksv_ref")
BAMIDX = os.path.join(ROOT, "oracle", "_ref", "bamidx")
OURS = os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")


def phases(stderr):
    """the `[timing] phase seconds s` lines the CLI prints under SSV_TIMING=1"""
    out = {}
    for line in stderr.splitlines():
        if line.startswith("[timing] ") and not line.startswith("[timing] ("):   # (the bracketed lines are detail, not phases)
            try:
                name, sec, _ = line[9:].rsplit(" ", 2)
                out[name] = round(float(sec), 3)
            except ValueError:
                pass
    return out


def sha(path):
    """sha256 of a file's content - of the INFLATED text for a .gz (the two programs' gzip members differ, their text must not)"""
    import hashlib
    data = gzip.open(path).read() if path.endswith(".gz") else open(path, "rb").read()
    return hashlib.sha256(data).hexdigest()


def timed(cmd, **kw):
    t = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    return time.perf_counter() - t, r


def main():
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1 / 128
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30
    n_sv = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    unmap = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    n_int = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    wkw = dict(genome_frac=frac, n_sv=n_sv, unmap_permille=unmap, hbv=n_int > 0)
    w = synth.Workload(depth=depth, n_integrations=n_int, **wkw)
    d = tempfile.mkdtemp(prefix="ssv_cli_")
    bam = os.path.join(d, "synth.bam")
    t = time.perf_counter()
    chunk = 2_000_000
    host.write_bam(bam, w.names, w.lens, (w.generate_host(g, min(chunk, w.n_total - g)) for g in range(0, w.n_total, chunk)))
    t_write = time.perf_counter() - t
    out = {"records": w.n_total, "bam_bytes": os.path.getsize(bam), "write_s": round(t_write, 2), "junctions": len(w.junctions), "host_cpus": os.cpu_count(), "unmapped_permille": unmap, "virus_integrations": n_int}
    EXTS = ("clip.gz", "clip.fq.gz", "unmapped_1.fq.gz", "unmapped_2.fq.gz")
    have_ref = os.path.exists(REF)
    if have_ref:
        subprocess.run([BAMIDX, bam], capture_output=True)
        out["ref_getclip_s"], r = timed([REF, "getclip", "-o", os.path.join(d, "ref"), bam])
        assert r.returncode == 0, r.stderr
    out["ours_getclip_s"], r = timed([OURS, "getclip", "-o", os.path.join(d, "ours"), bam])
    assert r.returncode == 0, r.stderr
    if phases(r.stderr):
        out["ours_getclip_phases_s"] = phases(r.stderr)
    # the same with the BAM inflated and decoded on the GPU
    out["ours_getclip_Z_s"], r = timed([OURS, "getclip", "-Z", "-o", os.path.join(d, "oursz"), bam])
    assert r.returncode == 0, r.stderr
    if phases(r.stderr):
        out["ours_getclip_Z_phases_s"] = phases(r.stderr)
    for ext in EXTS:
        assert gzip.open(os.path.join(d, "ours." + ext)).read() == gzip.open(os.path.join(d, "oursz." + ext)).read(), ext
    # ... and cut into three runs of records (three ranks on the box's one GPU), inflated on the device
    out["ours_getclip_N3_Z_s"], r = timed([OURS, "getclip", "-Z", "-N", "3", "-o", os.path.join(d, "oursn"), bam])
    assert r.returncode == 0, r.stderr
    digests = out.setdefault("sha256", {})
    for ext in EXTS:
        digests[ext] = {who: sha(os.path.join(d, who_prefix + "." + ext)) for who, who_prefix in (("ours", "ours"), ("ours_Z", "oursz"), ("ours_N3_Z", "oursn"))}
        assert len(set(digests[ext].values())) == 1, ext
    out["unmapped_pairs_written"] = gzip.open(os.path.join(d, "ours.unmapped_1.fq.gz")).read().count(b"\n") // 4
    if have_ref:
        for ext in EXTS:
            digests[ext]["reference"] = sha(os.path.join(d, "ref." + ext))
        for ext in EXTS:
            assert gzip.open(os.path.join(d, "ref." + ext)).read() == gzip.open(os.path.join(d, "ours." + ext)).read(), ext
        out["getclip_outputs_identical"] = True
    # getsv on the planted junctions through the -B harness (no bwa on the box)
    jfile = os.path.join(d, "junctions.txt")
    with open(jfile, "w") as f:
        for j in w.junctions:
            f.write("\t".join(str(x) for x in (j[0], j[1], j[2], 0, j[3], j[4], j[5], 0, 0, 0, "NA", 0, 0, 0, 0, 0, 0, 0, 0, "50M", "50M", "ACGT", "ACGT")) + "\n")
    empty_bam = os.path.join(d, "empty.clip.bam")
    host.write_bam(empty_bam, w.names, w.lens, [])
    empty_clip = os.path.join(d, "empty.clip")
    open(empty_clip, "w").close()
    args = ["-d", "0", "-f", "0", "-b", "0", "-B", jfile, empty_bam, bam, empty_clip]
    if have_ref:
        out["ref_getsv_s"], r = timed([REF, "getsv"] + args + [os.path.join(d, "ref.sv"), os.path.join(d, "x.fq")])
        assert r.returncode == 0, r.stderr
        ref_stdout = r.stdout
    out["ours_getsv_s"], r = timed([OURS, "getsv"] + args + [os.path.join(d, "ours.sv"), os.path.join(d, "y.fq")])
    assert r.returncode == 0, r.stderr
    if phases(r.stderr):
        out["ours_getsv_phases_s"] = phases(r.stderr)
    out["ours_getsv_Z_s"], rz = timed([OURS, "getsv", "-Z"] + args + [os.path.join(d, "oursz.sv"), os.path.join(d, "z.fq")])
    assert rz.returncode == 0, rz.stderr
    if phases(rz.stderr):
        out["ours_getsv_Z_phases_s"] = phases(rz.stderr)
    assert open(os.path.join(d, "ours.sv")).read() == open(os.path.join(d, "oursz.sv")).read() and rz.stdout == r.stdout
    out["ours_getsv_N3_Z_s"], rn = timed([OURS, "getsv", "-Z", "-N", "3"] + args + [os.path.join(d, "oursn.sv"), os.path.join(d, "n.fq")])
    assert rn.returncode == 0, rn.stderr
    assert rn.stdout == r.stdout
    digests["getsv -B sv table"] = {who: sha(os.path.join(d, f + ".sv")) for who, f in (("ours", "ours"), ("ours_Z", "oursz"), ("ours_N3_Z", "oursn"))}
    assert len(set(digests["getsv -B sv table"].values())) == 1
    if have_ref:
        digests["getsv -B sv table"]["reference"] = sha(os.path.join(d, "ref.sv"))
        assert open(os.path.join(d, "ref.sv")).read() == open(os.path.join(d, "ours.sv")).read()
        assert ref_stdout == r.stdout
        out["getsv_outputs_identical"] = True
        out["ref_records_per_s"] = round(w.n_total / (out["ref_getclip_s"] + out["ref_getsv_s"]))
        out["speedup_getclip"] = round(out["ref_getclip_s"] / out["ours_getclip_s"], 2)
        out["speedup_getsv"] = round(out["ref_getsv_s"] / out["ours_getsv_s"], 2)
    out["ours_records_per_s"] = round(w.n_total / (out["ours_getclip_s"] + out["ours_getsv_s"]))
    if len(sys.argv) > 4 and sys.argv[4] == "full":
        fa = os.path.join(d, "ref.fa")
        with open(fa, "w") as f:
            f.write(w.reference_fasta())
        clip_bam = os.path.join(d, "ours.clip.bam")
        out["ours_realign_s"], r = timed([OURS, "realign", fa, os.path.join(d, "ours.clip.fq.gz"), clip_bam])
        assert r.returncode == 0, r.stderr
        out["realign_summary"] = r.stderr.strip().splitlines()[-1]
        if phases(r.stderr):
            out["ours_realign_phases_s"] = phases(r.stderr)
        full = [clip_bam, bam, os.path.join(d, "ours.clip.gz")]
        out["ours_getsv_full_s"], r = timed([OURS, "getsv"] + full + [os.path.join(d, "ours.full.sv"), os.path.join(d, "u1.fq")])
        assert r.returncode == 0, r.stderr
        if phases(r.stderr):
            out["ours_getsv_full_phases_s"] = phases(r.stderr)
        out["ours_getsv_full_Z_s"], rz = timed([OURS, "getsv", "-Z"] + full + [os.path.join(d, "oursz.full.sv"), os.path.join(d, "u3.fq")])
        assert rz.returncode == 0, rz.stderr
        assert rz.stdout == r.stdout
        digests["full pipeline sv table"] = {"ours": sha(os.path.join(d, "ours.full.sv")), "ours_Z": sha(os.path.join(d, "oursz.full.sv"))}
        assert len(set(digests["full pipeline sv table"].values())) == 1
        digests["full pipeline unmapped fq"] = {"ours": sha(os.path.join(d, "u1.fq")), "ours_Z": sha(os.path.join(d, "u3.fq"))}
        rows = [l.split("\t") for l in open(os.path.join(d, "ours.full.sv")) if not l.startswith("@")]
        found = {(c[0], int(c[1]), c[2], c[4], int(c[5]), c[6]) for c in rows}
        planted = {tuple(j[:6]) for j in w.junctions}
        out["planted"], out["planted_found"], out["sv_rows"] = len(planted), len(planted & found), len(rows)
        if have_ref:
            out["ref_getsv_full_s"], rr = timed([REF, "getsv"] + full + [os.path.join(d, "ref.full.sv"), os.path.join(d, "u2.fq")])
            assert rr.returncode == 0, rr.stderr
            assert open(os.path.join(d, "ref.full.sv")).read() == open(os.path.join(d, "ours.full.sv")).read()
            assert rr.stdout == r.stdout
            digests["full pipeline sv table"]["reference"] = sha(os.path.join(d, "ref.full.sv"))
            digests["full pipeline unmapped fq"]["reference"] = sha(os.path.join(d, "u2.fq"))
            assert len(set(digests["full pipeline unmapped fq"].values())) == 1
            out["full_pipeline_sv_table_identical"] = True
            out["speedup_getsv_full"] = round(out["ref_getsv_full_s"] / out["ours_getsv_full_s"], 2)
        # ---- round 6: the whole pipeline in ONE process, and the tumor's table against a normal sample ----
        out["ours_run_s"], r1 = timed([OURS, "run", bam, fa, os.path.join(d, "one")], env=dict(os.environ, SSV_TIMING="1"))
        assert r1.returncode == 0, r1.stderr
        out["ours_run_phases_s"] = {k: v for k, v in phases(r1.stderr).items() if k.startswith("run:")}
        assert open(os.path.join(d, "one.sv.txt")).read() == open(os.path.join(d, "ours.full.sv")).read() and r1.stdout == r.stdout
        for ext in EXTS:
            assert sha(os.path.join(d, "one." + ext)) == digests[ext]["ours"], ext
        digests["full pipeline sv table"]["ours_run"] = sha(os.path.join(d, "one.sv.txt"))
        wn = synth.Workload(depth=depth / 2, **wkw)   # the same patient's normal: same reference, same germline SVs, no virus
        nbam = os.path.join(d, "normal.bam")
        host.write_bam(nbam, wn.names, wn.lens, (wn.generate_host(g, min(chunk, wn.n_total - g)) for g in range(0, wn.n_total, chunk)))
        out["normal_records"] = wn.n_total
        out["ours_getclip_normal_Z_s"], r2 = timed([OURS, "getclip", "-Z", "-o", os.path.join(d, "N"), nbam])
        assert r2.returncode == 0, r2.stderr
        for tag, extra in (("ours_somatic_s", []), ("ours_somatic_Z_s", ["-Z"])):
            out[tag], r3 = timed([OURS, "somatic"] + extra + [nbam, os.path.join(d, "N.clip.gz"), os.path.join(d, "ours.full.sv"), os.path.join(d, tag + ".sv")], env=dict(os.environ, SSV_TIMING="1"))
            assert r3.returncode == 0, r3.stderr
            out[tag.replace("_s", "_phases_s")] = phases(r3.stderr)
        digests["somatic table"] = {"ours": sha(os.path.join(d, "ours_somatic_s.sv")), "ours_Z": sha(os.path.join(d, "ours_somatic_Z_s.sv"))}
        rows_s = [l.rstrip("\n").split("\t") for l in open(os.path.join(d, "ours_somatic_s.sv")) if not l.startswith("@")]
        out["somatic_rows"], out["somatic_calls"] = len(rows_s), sum(1 for c in rows_s if c[23] == "0" and c[24] == "0" and c[25] == "0")
        if have_ref:
            subprocess.run([BAMIDX, nbam], capture_output=True)
            out["ref_getclip_normal_s"], rr2 = timed([REF, "getclip", "-o", os.path.join(d, "refN"), nbam])
            assert rr2.returncode == 0, rr2.stderr
            assert sha(os.path.join(d, "refN.clip.gz")) == sha(os.path.join(d, "N.clip.gz"))
            out["ref_somatic_s"], rr3 = timed([REF, "somatic", nbam, os.path.join(d, "refN.clip.gz"), os.path.join(d, "ref.full.sv"), os.path.join(d, "ref.somatic.sv")])
            assert rr3.returncode == 0, rr3.stderr
            digests["somatic table"]["reference"] = sha(os.path.join(d, "ref.somatic.sv"))
            out["speedup_somatic"] = round(out["ref_somatic_s"] / out["ours_somatic_Z_s"], 2)
        assert len(set(digests["somatic table"].values())) == 1, digests["somatic table"]
        out["somatic_table_identical"] = True
    for k in list(out):
        if k.endswith("_s") and not isinstance(out[k], dict):
            out[k] = round(out[k], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
