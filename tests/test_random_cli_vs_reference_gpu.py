"""-m gpu (and only where the REAL reference binary travelled along: oracle/_ref/seeksv_ref): the host junction stage (clip.gz x clip.bam join,
GetAlignInfo, GetJunction, MergeJunction, OutputBreakpoint: SURVEY 8f #1) on randomized inputs, `seeksv getsv` of this repository against
the reference's, byte for byte.  The inputs are made directly (no aligner): clip rows of planted junctions in all orientations whose clipped
sequences re-align to the partner breakpoint (forward / reverse, with microhomology shifts, soft- and hard-clipped ends, MAPQ 0, secondary
records, both-sided and one-sided support), duplicated and near-duplicate junctions that MergeJunction folds, and noise rows that re-align
nowhere or anywhere; the BAM passes run on a committed background BAM."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import bamio
import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "seeksv_ref")
SEEKSV = os.environ.get("SSV_CLI") or os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF), reason="the real reference binary is not here")]
COMP = str.maketrans("ACGT", "TGCA")


def rnd(rng, n):
    return "".join("ACGT"[x] for x in rng.randint(0, 4, n))


def revcomp(s):
    return s.translate(COMP)[::-1]


def make_inputs(seed, d):
    rng = np.random.RandomState(seed)
    bg = os.path.join(G.GOLDEN, "getsv", "pairs1.bam")
    names, recs0 = bamio.read_bam_records(bg)
    lens = {"chrA": 40000, "chrB": 15000, "HBV": 3215}
    rows, aln = [], {}  # rows: (chr, pos, side, cigar, aligned_seq, clipped_seq, support); aln[clipped_seq] = list of records

    def qual(n):
        return "".join(chr(33 + int(x)) for x in rng.choice([2, 11, 25, 37, 40], n))

    def add_row(ch, pos, side, aligned, clipped, support, cigar=None):
        rows.append((ch, int(pos), side, cigar or f"{len(aligned)}M", aligned, clipped, int(support)))

    def add_aln(clipped, ch, pos1, reverse=False, mapq=60, lead=0, trail=0, hard=False, secondary=False, unmapped=False):
        L = len(clipped)
        if unmapped:
            aln.setdefault(clipped, []).append(dict(flag=4, tid=-1, pos=-1, mapq=0, cigar=[], seq=clipped))
            return
        m = L - lead - trail
        op = 5 if hard else 4
        cig = ([(lead, op)] if lead else []) + [(m, 0)] + ([(trail, op)] if trail else [])
        seq = revcomp(clipped) if reverse else clipped
        if hard:
            seq = seq[lead:L - trail]
        aln.setdefault(clipped, []).append(dict(flag=(16 if reverse else 0) | (256 if secondary else 0), tid=names.index(ch), pos=int(pos1) - 1, mapq=mapq, cigar=cig, seq=seq))

    for k in range(70):
        ca, cb = rng.choice(names, 2)
        same = rng.rand() < 0.6
        if same: cb = ca
        A = int(rng.randint(300, lens[ca] - 300)); B = int(rng.randint(300, lens[cb] - 300))
        if same and abs(A - B) < 120: B = A + 150 + int(rng.randint(0, 3000)) if A + 3300 < lens[ca] else A - 200
        left, right = rnd(rng, 140), rnd(rng, 140)   # sequence up to A (ends at A) and from B on
        kind = k % 5
        ka, kb = int(rng.randint(25, 70)), int(rng.randint(25, 70))
        sa, sb = int(rng.randint(1, 12)), int(rng.randint(1, 12))
        mq = int(rng.choice([60, 60, 60, 30, 0]))
        if kind in (0, 1, 2):   # (A,+) -> (B,+): right-clipped reads at A, left-clipped reads at B
            if kind != 2:
                clipped = right[:ka]
                add_row(ca, A, "3", left[-(100 - ka):], clipped, sa)
                add_aln(clipped, cb, B + int(rng.choice([0, 0, 0, 1, -2, 5])), mapq=mq, trail=int(rng.choice([0, 0, 4])))
            if kind != 1:
                clipped = left[-kb:]
                add_row(cb, B, "5", right[:100 - kb], clipped, sb)
                add_aln(clipped, ca, A - kb + 1 + int(rng.choice([0, 0, 0, -1, 3])), mapq=int(rng.choice([60, 60, 0])), lead=int(rng.choice([0, 0, 5])))
        elif kind == 3:         # inversion-like: the clipped part re-aligns on the reverse strand
            clipped = revcomp(right[:ka])
            add_row(ca, A, "3", left[-(100 - ka):], clipped, sa)
            add_aln(clipped, cb, B, reverse=True, mapq=mq)
            clipped2 = revcomp(left[-kb:])
            add_row(cb, B + ka, "3", rnd(rng, 100 - kb), clipped2, sb)
            add_aln(clipped2, ca, A - kb + 1, reverse=True, mapq=60, hard=bool(rng.rand() < 0.2), lead=int(rng.choice([0, 3])))
        else:                   # left-clipped at A, partner reverse
            clipped = revcomp(rnd(rng, ka))
            add_row(ca, A, "5", right[:100 - ka], clipped, sa)
            add_aln(clipped, cb, B, reverse=bool(rng.rand() < 0.7), mapq=mq, secondary=bool(rng.rand() < 0.15))
        if rng.rand() < 0.3:    # a near-duplicate a few bases away: MergeJunction candidates
            sh = int(rng.choice([1, 2, 3, 7, 20]))
            clipped = right[sh:sh + ka]
            add_row(ca, A + sh, "3", (left + right[:sh])[-(100 - ka):], clipped, int(rng.randint(1, 5)))
            add_aln(clipped, cb, B + sh, mapq=60)
    for _ in range(60):         # noise
        ch = str(rng.choice(names)); p = int(rng.randint(200, lens[ch] - 200)); L = int(rng.randint(20, 75))
        clipped = rnd(rng, L)
        add_row(ch, p, "53"[int(rng.randint(0, 2))], rnd(rng, 100 - L), clipped, int(rng.randint(1, 4)), cigar=str(rng.choice([f"{100 - L}M", f"{60 - L // 2}M2D{40 - (L - L // 2)}M"])))
        r = rng.rand()
        if r < 0.6: add_aln(clipped, None, 0, unmapped=True)
        else:
            c2 = str(rng.choice(names))
            add_aln(clipped, c2, int(rng.randint(100, lens[c2] - 200)), reverse=bool(rng.rand() < 0.5), mapq=int(rng.choice([0, 0, 13, 60])), lead=int(rng.choice([0, 0, 6])), trail=int(rng.choice([0, 0, 6])))
            if rng.rand() < 0.3: add_aln(clipped, str(rng.choice(names)), int(rng.randint(100, 3000)), secondary=True, mapq=0)
    # getclip's output order: per contig, side 5 by position, then side 3 by position; rows of one clipped sequence need not be adjacent
    order = sorted(range(len(rows)), key=lambda i: (names.index(rows[i][0]), rows[i][2] == "3", rows[i][1]))
    rows = [rows[i] for i in order]
    clip_gz, clip_bam = os.path.join(d, "in.clip.gz"), os.path.join(d, "in.clip.bam")
    with gzip.open(clip_gz, "wt") as f:
        for ch, pos, side, cigar, aligned, clipped, support in rows:
            f.write("\t".join([ch, str(pos), side, cigar, aligned, qual(len(aligned)), clipped, qual(len(clipped)), str(support)]) + "\n")
    out, done = [], set()
    for ch, pos, side, cigar, aligned, clipped, support in rows:  # like bwa mem: records follow the FASTQ, all records of a read together
        if clipped in done:
            continue
        done.add(clipped)
        for a in aln[clipped]:
            out.append(dict(qname=clipped, flag=a["flag"], tid=a["tid"], pos=a["pos"], mapq=a["mapq"], cigar=a["cigar"], mtid=-1, mpos=-1, isize=0, seq=a["seq"], qual=None))
    bamio.write_bam(clip_bam, names, [lens[n] for n in names], out)
    return bg, clip_bam, clip_gz


@pytest.mark.parametrize("seed", range(8))
def test_getsv_full_equals_reference_on_random_junction_inputs(tmp_path, seed):
    d = str(tmp_path)
    bg, clip_bam, clip_gz = make_inputs(seed, d)
    for tag, flags in (("default", []), ("loose", ["-f", "0", "-b", "0", "-d", "0"]), ("l90", ["-l", "90", "-e", "1"])):
        ref = subprocess.run([REF, "getsv"] + flags + [clip_bam, bg, clip_gz, os.path.join(d, f"ref.{tag}.sv"), os.path.join(d, "r.fq")], capture_output=True, text=True)
        assert ref.returncode == 0, ref.stderr[-400:]
        ours = subprocess.run([SEEKSV, "getsv"] + flags + [clip_bam, bg, clip_gz, os.path.join(d, f"ours.{tag}.sv"), os.path.join(d, "o.fq")], capture_output=True, text=True)
        assert ours.returncode == 0, ours.stderr[-400:]
        want, got = open(os.path.join(d, f"ref.{tag}.sv")).read(), open(os.path.join(d, f"ours.{tag}.sv")).read()
        assert got == want, (seed, tag)
        assert ours.stdout == ref.stdout, (seed, tag)
        if tag == "loose":
            assert want.count("\n") > 30 and ref.stdout.count("\n") >= 0, want.count("\n")
    # seeksv somatic with the loose table as the tumor's and the same clip clusters / BAM as the control: every look-up kind
    # finds something, and a shifted copy of the clusters makes the range probes walk
    table = os.path.join(d, "ref.loose.sv")
    for tag, flags in (("s", []), ("s_l60", ["-l", "60", "-m", "30"]), ("s_t", ["-t", "0.5", "-q", "0"])):
        ref = subprocess.run([REF, "somatic"] + flags + [bg, clip_gz, table, os.path.join(d, f"ref.{tag}.sv")], capture_output=True, text=True)
        assert ref.returncode == 0, ref.stderr[-400:]
        ours = subprocess.run([SEEKSV, "somatic"] + flags + [bg, clip_gz, table, os.path.join(d, f"ours.{tag}.sv")], capture_output=True, text=True)
        assert ours.returncode == 0, ours.stderr[-400:]
        assert open(os.path.join(d, f"ours.{tag}.sv")).read() == open(os.path.join(d, f"ref.{tag}.sv")).read(), (seed, tag)


@pytest.mark.parametrize("seed", range(200, 206))
@pytest.mark.parametrize("mode", [[], ["-Z"]], ids=["host-inflate", "device-inflate"])
def test_cli_getclip_equals_reference_on_random_samples(tmp_path, seed, mode):
    """`seeksv getclip` file to file against the real reference on generated BAMs (the generator of tests/test_random_differential_gpu.py,
    restricted to inputs the reference defines): the four gzip outputs decompress to the same bytes, with the BAM decoded on the host or on the GPU"""
    from test_random_differential_gpu import random_sample
    from test_random_oracle_vs_reference import write_sample
    names, lens, b, rng, recs = random_sample(seed, n=2500, safe=True, want_records=True)
    bam = str(tmp_path / "s.bam")
    write_sample(bam, names, lens, recs)
    for tag, flags in (("a", []), ("b", ["-t", "0.8", "-q", "0", "-s"])):
        r = subprocess.run([REF, "getclip"] + flags + ["-o", str(tmp_path / ("ref" + tag)), bam], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-300:]
        o = subprocess.run([SEEKSV, "getclip"] + mode + flags + ["-o", str(tmp_path / ("our" + tag)), bam], capture_output=True, text=True)
        assert o.returncode == 0, o.stderr[-300:]
        for ext in ("clip.gz", "clip.fq.gz", "unmapped_1.fq.gz", "unmapped_2.fq.gz"):
            assert gzip.open(str(tmp_path / f"our{tag}.{ext}")).read() == gzip.open(str(tmp_path / f"ref{tag}.{ext}")).read(), (seed, tag, ext)


@pytest.mark.parametrize("mode,env", [([], {}), (["-Z"], {}), (["-Z"], {"SSV_CHUNK_INFLATED_MB": "1"}), ([], {"SSV_HOST_BATCH_RECORDS": "997"})], ids=["host", "device", "device-small-chunks", "host-small-batches"])
def test_cli_getclip_unmapped_pairs_of_the_synthetic_sample(tmp_path, mode, env):
    """The generator's pairs with one unmapped end (synth_core.h: unmap_permille, 3 % here) through `seeksv getclip`: the side channel (unmapped_pairs.h: a
    thread of its own over the readers' raw records) writes what the reference's std::map loop writes (clip_reads.h:172-219), mates a batch apart included."""
    from seeksv_amd import host, synth
    w = synth.Workload(genome_frac=1 / 4096, depth=30, n_sv=12, unmap_permille=30)
    b = w.generate_host(0, w.n_total, all_seq=True)
    n_un = int(((b["flag"] & 12) != 0).sum())
    assert 0.02 * w.n_total < n_un < 0.04 * w.n_total and n_un % 2 == 0
    bam = str(tmp_path / "u.bam")
    host.write_bam(bam, w.names, w.lens, [b])
    r = subprocess.run([REF, "getclip", "-o", str(tmp_path / "ref"), bam], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-300:]
    o = subprocess.run([SEEKSV, "getclip"] + mode + ["-o", str(tmp_path / "our"), bam], capture_output=True, text=True, env=dict(os.environ, **env))
    assert o.returncode == 0, o.stderr[-300:]
    for ext in ("clip.gz", "clip.fq.gz", "unmapped_1.fq.gz", "unmapped_2.fq.gz"):
        want = gzip.open(str(tmp_path / f"ref.{ext}")).read()
        assert gzip.open(str(tmp_path / f"our.{ext}")).read() == want, ext
        if ext.startswith("unmapped"):
            assert want.count(b"\n") == 4 * (n_un // 2)


def test_cli_getclip_table_columns_both_ways(tmp_path):
    """The cluster table's columns - dense cluster index, CIGAR offset, string offset of every sorted slot - come out of scans over TILES of 256 slots
    (k_cluster_tile_sums / k_scan_sums_lists / k_cluster_cols3_tiles, table3_kernels.h), or the round-5 way out of two words per slot and two device-wide
    scans (SSV_PACK_COLS=split).  A sample with some hundred tiles: both write what the real reference writes."""
    from seeksv_amd import host, synth
    w = synth.Workload(genome_frac=1 / 128, depth=30, n_sv=80)
    b = w.generate_host(0, w.n_total, all_seq=True)
    bam = str(tmp_path / "t.bam")
    host.write_bam(bam, w.names, w.lens, [b])
    r = subprocess.run([REF, "getclip", "-o", str(tmp_path / "ref"), bam], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-300:]
    want = {ext: gzip.open(str(tmp_path / f"ref.{ext}")).read() for ext in ("clip.gz", "clip.fq.gz")}
    assert want["clip.gz"].count(b"\n") > 64 * 256 * 2   # (rows = clusters: more than a hundred tiles)
    for tag, env in (("tiles", {}), ("split", {"SSV_PACK_COLS": "split"})):
        o = subprocess.run([SEEKSV, "getclip", "-Z", "-o", str(tmp_path / tag), bam], capture_output=True, text=True, env=dict(os.environ, **env))
        assert o.returncode == 0, o.stderr[-300:]
        for ext, data in want.items():
            assert gzip.open(str(tmp_path / f"{tag}.{ext}")).read() == data, (tag, ext)
