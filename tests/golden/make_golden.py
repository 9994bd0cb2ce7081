#!/usr/bin/env python3
"""Regenerate tests/golden/ by running the REAL reference (oracle/_ref/seeksv_ref, built from
/root/reference by `make -C oracle ref`) on the bundled example BAMs and on crafted BAMs written here.

Runs only in the build container (needs /root/reference).  What is committed is data: input BAMs,
junction lists and the reference's outputs.  bwa (example/bin/bwa, 0.7.10) is only needed for the
example clip.bam fixtures, exactly as example/seeksv.sh:3 uses it.

usage: python tests/golden/make_golden.py
"""
import gzip
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamio  # noqa: E402

REF = "/root/reference"
BIN = os.path.join(ROOT, "oracle", "_ref")
SEEKSV = os.path.join(BIN, "seeksv_ref")
TMP = "/tmp/seeksv_golden_work"


def run(cmd, stdout=None, stderr=None, cwd=None):
    with open(stdout, "wb") if stdout else open(os.devnull, "wb") as so, open(stderr, "wb") if stderr else open(os.devnull, "wb") as se:
        subprocess.check_call(cmd, stdout=so, stderr=se, cwd=cwd)


def gunzip_to(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read())


# ------------------------------------------------------------------------------------------------
# 1. bundled example BAMs: full pipeline of example/seeksv.sh + seeksv.somatic.sh
# ------------------------------------------------------------------------------------------------

def example():
    out = os.path.join(HERE, "example")
    os.makedirs(out, exist_ok=True)
    os.makedirs(TMP, exist_ok=True)
    bwa = os.path.join(TMP, "bwa")
    shutil.copy(os.path.join(REF, "example/bin/bwa"), bwa)
    os.chmod(bwa, 0o755)
    for s in ("cancer", "normal"):
        bam = os.path.join(out, f"{s}.sort.bam")
        shutil.copy(os.path.join(REF, f"example/{s}.sort.bam"), bam)
        shutil.copy(os.path.join(REF, f"example/{s}.sort.bam.bai"), bam + ".bai")
        os.chmod(bam, 0o644), os.chmod(bam + ".bai", 0o644)
        run([SEEKSV, "getclip", "-o", s, bam], cwd=TMP, stderr=os.path.join(out, f"{s}.getclip.stderr"))
        gunzip_to(os.path.join(TMP, f"{s}.clip.gz"), os.path.join(out, f"{s}.clip.txt"))
        gunzip_to(os.path.join(TMP, f"{s}.clip.fq.gz"), os.path.join(out, f"{s}.clip.fq.txt"))
        gunzip_to(os.path.join(TMP, f"{s}.unmapped_1.fq.gz"), os.path.join(out, f"{s}.unmapped_1.fq.txt"))
        gunzip_to(os.path.join(TMP, f"{s}.unmapped_2.fq.gz"), os.path.join(out, f"{s}.unmapped_2.fq.txt"))
        run([bwa, "mem", os.path.join(REF, "example/reference/example.fa"), f"{s}.clip.fq.gz"], cwd=TMP, stdout=os.path.join(TMP, f"{s}.clip.sam"))
        run([os.path.join(BIN, "sam2bam"), f"{s}.clip.sam", os.path.join(out, f"{s}.clip.bam")], cwd=TMP)
        run([SEEKSV, "getsv", os.path.join(out, f"{s}.clip.bam"), bam, f"{s}.clip.gz", os.path.join(out, f"{s}.sv"), f"{s}.clipunmap.fq"],
            cwd=TMP, stdout=os.path.join(out, f"{s}.getsv.stdout"), stderr=os.path.join(TMP, f"{s}.getsv.stderr"))
        # stderr carries the absolute path of the BAM: keep only the insert size line values
        with open(os.path.join(TMP, f"{s}.getsv.stderr")) as f, open(os.path.join(out, f"{s}.isize.txt"), "w") as g:
            txt = f.read()
            mean = txt.split("Mean insert size : ")[1].split()[0]
            sd = txt.split("Mean deviation: ")[1].split()[0]
            g.write(f"{mean}\t{sd}\n")
        # every filter reason at least once
        for tag, flags in (("b40", ["-b", "40"]), ("f2", ["-f", "2"]), ("d400", ["-d", "400"]), ("e60", ["-e", "60"]), ("D", ["-D"]), ("n0", ["-n", "0"])):
            run([SEEKSV, "getsv"] + flags + [os.path.join(out, f"{s}.clip.bam"), bam, f"{s}.clip.gz", os.path.join(out, f"{s}.{tag}.sv"), f"{s}.clipunmap.fq"],
                cwd=TMP, stdout=os.path.join(out, f"{s}.{tag}.getsv.stdout"))
    run([SEEKSV, "somatic", os.path.join(out, "normal.sort.bam"), os.path.join(TMP, "normal.clip.gz"), os.path.join(out, "cancer.sv"),
         os.path.join(out, "cancer.somatic.temp.sv")], cwd=TMP)


# ------------------------------------------------------------------------------------------------
# 2. crafted getclip cases
# ------------------------------------------------------------------------------------------------

def rnd_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def rnd_qual(rng, n):
    return bytes(rng.choice((2, 11, 25, 37, 40)) for _ in range(n))


def mutate(rng, s, rate):
    return "".join(rng.choice("ACGT".replace(c, "")) if rng.random() < rate else c for c in s)


def getclip_filters_case():
    """One record per filter / CIGAR shape (SURVEY 8c 'behaviours verified with crafted BAMs')."""
    rng = random.Random(11)
    names, lens = ["c1", "c2", "c3"], [5000, 5000, 3000]
    recs = []

    def add(tid, pos, cigar, flag=0, mapq=60, aux=b"", qual="rand", mtid=None, mpos=None, isize=0, seq=None):
        L = sum(l for l, op in bamio.parse_cigar(cigar) if op in (0, 1, 4, 7, 8))
        recs.append(dict(qname=f"r{len(recs)}", flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, mtid=tid if mtid is None else mtid,
                         mpos=pos + 200 if mpos is None else mpos, isize=isize, seq=seq or rnd_seq(rng, L),
                         qual=rnd_qual(rng, L) if qual == "rand" else qual, aux=aux))

    xc = b"XCi" + (7).to_bytes(4, "little")
    xc_c = b"XCC" + bytes([3])
    xc0 = b"XCi" + (0).to_bytes(4, "little")
    nm = b"NMC" + bytes([1])
    add(0, 100, "50M")                       # first record of contig 0 (tid == initial last_tid 0): processed, no clip
    add(0, 299, "10S30M10S")                 # both sides: '5' at 300, '3' at 329
    add(0, 400, "20S80M")                    # left clip
    add(0, 500, "80M20S")                    # right clip -> key 500+80
    add(0, 600, "20S80M", flag=256)          # secondary: kept
    add(0, 610, "20S80M", flag=512)          # QC fail: kept
    add(0, 620, "20S80M", flag=2048)         # supplementary: kept
    add(0, 700, "20S80M", mapq=0)            # MAPQ 0: dropped (default -q 1)
    add(0, 710, "20S80M", mapq=1)            # MAPQ 1: kept
    add(0, 800, "20S80M", flag=1024)         # duplicate: dropped
    add(0, 900, "20H80M")                    # hard clip: dropped
    add(0, 910, "20S60M20H")                 # hard clip at the other end: dropped
    add(0, 1000, "20S80M", flag=8)           # mate unmapped: unmapped side channel, no clip
    add(0, 1010, "20S80M", flag=4)           # unmapped flag: same
    add(0, 1100, "20S40M5D40M")              # deletion in the span
    add(0, 1200, "40M5I35M20S")              # insertion; right clip key = 1200 + 75
    add(0, 1300, "30M100N50M20S")            # ref skip counts in the span
    add(0, 1400, "30=5X45M20S")              # '=' counts, 'X' does NOT (clip_reads.cpp:322)
    add(0, 1500, "20S80M", aux=xc)           # XC != 0, one sided: dropped
    add(0, 1510, "20S80M", aux=nm + xc_c)    # XC as type C after another tag: dropped
    add(0, 1520, "20S80M", aux=xc0)          # XC == 0: kept
    add(0, 1600, "15S70M15S", aux=xc)        # XC != 0, both sided, forward: '5' only
    add(0, 1700, "15S70M15S", aux=xc, flag=16)  # reverse: '3' only
    add(0, 1800, "20S80M", qual=None)        # qualities absent: "*"
    add(0, 1900, "100M")
    add(1, 50, "20S80M")                     # first record of contig 1: triggers the flush and is LOST
    add(1, 60, "20S80M")                     # kept
    add(1, 70, "80M20S", flag=4)             # unmapped in between does not touch last_tid
    add(1, 80, "80M20S")
    add(2, 10, "100M")                       # switch, lost (no clip anyway)
    add(2, 20, "30S70M")
    add(2, 20, "35S65M", flag=16)
    return names, lens, recs


def getclip_stress_case(seed, n_bp=40, depth=30, L=100, alleles=(1, 1, 2)):
    """Deep bins: reads of varying clip length piled on shared breakpoints, with base errors and low
    qualities so that consensus updates, prepend/append growth, cigar replacement and second clusters all occur."""
    rng = random.Random(seed)
    names, lens = ["chrS", "chrT"], [60000, 30000]
    genome = [rnd_seq(rng, l) for l in lens]
    recs = []
    for tid in (0, 1):
        recs.append(dict(qname=f"lead{tid}", flag=0, tid=tid, pos=5, mapq=60, cigar=f"{L}M", mtid=tid, mpos=300, isize=0,
                         seq=genome[tid][5:5 + L], qual=rnd_qual(rng, L)))
    for b in range(n_bp):
        tid = rng.randrange(2)
        bp = rng.randrange(1000, lens[tid] - 1000)  # 0-based first aligned base (left clip) / last aligned+1 (right clip)
        n_alleles = rng.choice(alleles)
        partner = [rnd_seq(rng, L) for _ in range(n_alleles)]
        left = rng.random() < 0.5
        err = rng.choice((0.0, 0.01, 0.05, 0.12))
        for _ in range(rng.randrange(2, depth)):
            clip = rng.randrange(3, L - 20)
            al = L - clip
            a = rng.randrange(n_alleles)
            q = rnd_qual(rng, L)
            if left:
                seq = partner[a][L - clip:] + genome[tid][bp:bp + al]
                if rng.random() < 0.15:  # an indel inside the aligned part changes cigar_vec
                    cigar, pos = f"{clip}S{al - 10}M2D10M", bp
                else:
                    cigar, pos = f"{clip}S{al}M", bp
            else:
                seq = genome[tid][bp - al:bp] + partner[a][:clip]
                cigar, pos = f"{al}M{clip}S", bp - al
            recs.append(dict(qname=f"b{b}_{len(recs)}", flag=rng.choice((0, 16, 99, 147)), tid=tid, pos=pos, mapq=rng.choice((0, 1, 20, 60, 60, 60)),
                             cigar=cigar, mtid=tid, mpos=pos + 250, isize=350, seq=mutate(rng, seq, err), qual=q))
        if rng.random() < 0.3:  # double-clipped reads sharing the bin
            for _ in range(rng.randrange(1, 6)):
                c1, c2 = rng.randrange(5, 30), rng.randrange(5, 30)
                al = L - c1 - c2
                seq = partner[0][L - c1:] + genome[tid][bp:bp + al] + rnd_seq(rng, c2)
                recs.append(dict(qname=f"d{b}_{len(recs)}", flag=0, tid=tid, pos=bp, mapq=60, cigar=f"{c1}S{al}M{c2}S", mtid=tid, mpos=bp + 250,
                                 isize=350, seq=mutate(rng, seq, err), qual=rnd_qual(rng, L)))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    return names, lens, recs


def crafted_getclip():
    out = os.path.join(HERE, "getclip")
    os.makedirs(out, exist_ok=True)
    cases = {"filters": getclip_filters_case(), "stress1": getclip_stress_case(1), "stress2": getclip_stress_case(2, n_bp=60, depth=80, L=150),
             # a few breakpoints with thousands of reads from ~100 different partner sequences: bins with far more than 64 clusters
             "stress3": getclip_stress_case(3, n_bp=4, depth=2500, L=100, alleles=(90, 120))}
    for name, (names, lens, recs) in cases.items():
        bam = os.path.join(out, f"{name}.bam")
        bamio.write_bam(bam, names, lens, recs)
        variants = [("", [])]
        if name == "filters":
            variants += [(".s", ["-s"]), (".q0", ["-q", "0"]), (".q30", ["-q", "30"])]
        if name == "stress1":
            variants += [(".t08", ["-t", "0.8"]), (".t1", ["-t", "1"])]
        if name == "stress3":
            variants += [(".t08", ["-t", "0.8"])]
        for tag, flags in variants:
            pre = f"{name}{tag}"
            run([SEEKSV, "getclip"] + flags + ["-o", pre, bam], cwd=TMP)
            gunzip_to(os.path.join(TMP, f"{pre}.clip.gz"), os.path.join(out, f"{pre}.clip.txt"))
            gunzip_to(os.path.join(TMP, f"{pre}.clip.fq.gz"), os.path.join(out, f"{pre}.clip.fq.txt"))


def lone_s_case():
    """records whose whole CIGAR is one soft clip (`50S`) between ordinary clipped reads: the reference reads that single operation as BOTH
    ends of the CIGAR (clip_reads.cpp:150-175: a negative "middle" length) and prints two rows with an empty aligned part for it.
    (The reads here have 51 bases under a 50S CIGAR: the '3' row's clipped part is bases [1, 51).)"""
    out = os.path.join(HERE, "getclip")
    seq = "ACGTTGCAAGCTTAGGCTAACGTAGCTAGGATCCGATAGCTAGCTAGGCTA"

    def rec(name, pos, cigar):
        return dict(qname=name, flag=99, tid=0, pos=pos, mapq=60, cigar=cigar, mtid=0, mpos=pos + 200, isize=300, seq=seq, qual=bytes([30] * len(seq)))
    recs = [rec("a", 1000, "20S30M"), rec("lone1", 1500, "50S"), rec("b", 2000, "30M20S"), rec("lone2", 2500, "50S"), rec("c", 3000, "10S40M"), rec("d", 3000, "12S38M")]
    bam = os.path.join(out, "lone_s.bam")
    bamio.write_bam(bam, ["c1"], [100000], recs)
    run([SEEKSV, "getclip", "-o", "lone_s", bam], cwd=TMP)
    gunzip_to(os.path.join(TMP, "lone_s.clip.gz"), os.path.join(out, "lone_s.clip.txt"))
    gunzip_to(os.path.join(TMP, "lone_s.clip.fq.gz"), os.path.join(out, "lone_s.clip.fq.txt"))


def lone_s2_case():
    """more of the same: lone soft clips whose length equals the read's, several of them on one position (an empty part makes the match
    rate 0/0 = NaN: they never merge, clip_reads.cpp:204,216,266), on the position of ordinary clipped reads, with an XC tag on either
    strand (clip_reads.cpp:160-175), without qualities, duplicates / MAPQ 0 (dropped), as the first record of a contig (lost); under
    the default flags, -s and -q 0"""
    out = os.path.join(HERE, "getclip")
    rng = random.Random(77)
    names, lens = ["c1", "c2"], [50000, 20000]
    recs = []
    xc = b"XCi" + (5).to_bytes(4, "little")

    def add(tid, pos, cigar, flag=0, mapq=60, aux=b"", qual="rand", seq=None, L=None):
        if L is None:
            L = sum(l for l, op in bamio.parse_cigar(cigar) if op in (0, 1, 4, 7, 8))
        recs.append(dict(qname=f"r{len(recs)}", flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, mtid=tid, mpos=pos + 200, isize=0,
                         seq=seq or rnd_seq(rng, L), qual=rnd_qual(rng, L) if qual == "rand" else qual, aux=aux))
    shared = rnd_seq(rng, 60)
    add(0, 100, "60M")
    add(0, 1000, "60S", seq=shared)                # two identical lone clips on one position: two clusters per side, support 1 each
    add(0, 1000, "60S", seq=shared)
    add(0, 1000, "20S40M", seq=shared)             # an ordinary left clip in the same '5' bin (1001): no merge with the lone ones either way
    add(0, 1000, "20S40M", seq=shared)             # ... but with its twin
    add(0, 1940, "60M20S")                         # '3' at 2000
    add(0, 2000, "80S")                            # lone clip whose '3' key (pos + 0 = 2000) is the bin of the read above
    add(0, 3000, "40S", aux=xc)                    # XC != 0, forward: the '5' row only (without -s)
    add(0, 3100, "40S", aux=xc, flag=16)           # reverse: the '3' row only
    add(0, 3200, "40S", qual=None)                 # no qualities: "*" also for the empty part
    add(0, 3300, "40S", flag=1024)                 # duplicate: dropped
    add(0, 3400, "40S", mapq=0)                    # MAPQ 0: dropped unless -q 0
    add(0, 3500, "40S", flag=256)                  # secondary: kept
    add(0, 3600, "30S", L=45)                      # read longer than the clip, shorter than twice the clip: middle = -15
    add(0, 3700, "30S", L=75)                      # longer than twice the clip: middle = +15, an ordinary two-sided event
    add(0, 3800, "1S")                             # one base
    add(1, 10, "50S")                              # first record of contig c2: triggers the flush and is lost
    add(1, 20, "50S")
    add(1, 20, "25S25M")
    bam = os.path.join(out, "lone_s2.bam")
    bamio.write_bam(bam, names, lens, recs)
    for tag, flags in (("", []), (".s", ["-s"]), (".q0", ["-q", "0"])):
        pre = f"lone_s2{tag}"
        run([SEEKSV, "getclip"] + flags + ["-o", pre, bam], cwd=TMP)
        gunzip_to(os.path.join(TMP, f"{pre}.clip.gz"), os.path.join(out, f"{pre}.clip.txt"))
        gunzip_to(os.path.join(TMP, f"{pre}.clip.fq.gz"), os.path.join(out, f"{pre}.clip.fq.txt"))


def unsorted_case():
    """a BAM whose contigs come back (c1, c2, c1, c3, c2, c2 after unmapped reads, c1): the reference flushes its two maps at EVERY change of
    contig among the mapped-pair records (clip_reads.h:423-438), so reads of the two visits of c1 never share a cluster although they
    share a position, every visit prints its own block of '5' then '3' rows (and its own stderr line), and the first record of every visit
    is lost.  Positions also run backwards inside a visit (rows come out in map order: by position)."""
    out = os.path.join(HERE, "getclip")
    rng = random.Random(78)
    names, lens = ["c1", "c2", "c3"], [50000, 20000, 9000]
    recs = []
    shared = rnd_seq(rng, 100)

    def add(tid, pos, cigar, flag=0, mapq=60, seq=None):
        L = sum(l for l, op in bamio.parse_cigar(cigar) if op in (0, 1, 4, 7, 8))
        recs.append(dict(qname=f"r{len(recs)}", flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, mtid=tid, mpos=pos + 200, isize=0,
                         seq=(seq or rnd_seq(rng, 100))[:L], qual=rnd_qual(rng, L)))
    add(0, 50, "100M")
    for _ in range(3):
        add(0, 1000, "30S70M", seq=shared)       # visit 1 of c1: one cluster of 3 at '5' 1001
    add(0, 1930, "70M30S", seq=shared)           # '3' at 2000
    add(0, 900, "20S80M")                        # positions going backwards inside the visit
    add(1, 100, "30S70M", seq=shared)            # first record of c2: lost
    add(1, 100, "30S70M", seq=shared)
    add(1, 500, "80M20S")
    add(0, 1000, "30S70M", seq=shared)           # c1 again: this record only triggers the flush
    for _ in range(2):
        add(0, 1000, "30S70M", seq=shared)       # visit 2 of c1: its own cluster of 2 on the same position
    add(0, 1930, "70M30S", seq=shared)
    add(0, 1930, "70M30S", seq=shared)
    add(2, 10, "100M")
    add(2, 20, "25S75M")
    add(1, 100, "30S70M", seq=shared)            # c2 again (lost), then
    add(1, 100, "30S70M", seq=shared)
    add(1, 300, "40S60M", flag=4)                # unmapped-pair records do not touch last_tid ...
    add(0, 400, "40S60M", flag=8)                # ... whatever contig they name
    add(1, 100, "30S70M", seq=shared)            # so this one still belongs to the same visit of c2: cluster of 2
    add(1, 640, "60M40S")
    add(0, 1000, "30S70M", seq=shared)           # c1 a third time: lost
    add(0, 1000, "10S90M", seq=shared[20:] + shared[:20])
    bam = os.path.join(out, "unsorted.bam")
    bamio.write_bam(bam, names, lens, recs)
    run([SEEKSV, "getclip", "-o", "unsorted", bam], cwd=TMP, stderr=os.path.join(out, "unsorted.getclip.stderr"))
    gunzip_to(os.path.join(TMP, "unsorted.clip.gz"), os.path.join(out, "unsorted.clip.txt"))
    gunzip_to(os.path.join(TMP, "unsorted.clip.fq.gz"), os.path.join(out, "unsorted.clip.fq.txt"))


# ------------------------------------------------------------------------------------------------
# 3. getsv BAM passes through the -B junction-injection harness
# ------------------------------------------------------------------------------------------------

def junction_row(uc, up, us, dc, dp, ds, prev_abnormal=0):
    return "\t".join(str(x) for x in (uc, up, us, 0, dc, dp, ds, 0, 0, prev_abnormal, "NA", 0, 0, 0, 0, 0, 0, 0, 0, "50M", "50M", "ACGT", "ACGT")) + "\n"


def getsv_case(seed, n_pairs=6000, L=100, with_small_contig=True):
    rng = random.Random(seed)
    names, lens = ["chrA", "chrB"], [40000, 15000]
    if with_small_contig:
        names.append("HBV")
        lens.append(3215)
    recs = []
    cig_pool = [f"{L}M"] * 12 + [f"40M2D{L - 40}M", f"30M3I{L - 33}M", f"20S{L - 20}M", f"{L - 25}M25S", f"20H{L - 20}M", f"50M100N{L - 50}M", f"10S{L - 20}M10S"]

    def qlen(c):
        return sum(l for l, op in bamio.parse_cigar(c) if op in (0, 1, 4, 7, 8))

    def mk(tid, pos, flag, mtid, mpos, isize, cigar=None, mapq=None):
        cigar = cigar or rng.choice(cig_pool)
        lq = qlen(cigar)
        mq = mapq if mapq is not None else rng.choice((0, 10, 19, 20, 30, 60, 60, 60, 60))
        return dict(qname=f"q{len(recs)}", flag=flag, tid=tid, pos=pos, mapq=mq, cigar=cigar, mtid=mtid, mpos=mpos, isize=isize,
                    seq="A" * lq, qual=b"\x1e" * lq)

    def extra_flags():
        f = 0
        if rng.random() < 0.05: f |= 1024
        if rng.random() < 0.02: f |= 256
        if rng.random() < 0.02: f |= 512
        return f

    # background proper pairs
    for _ in range(n_pairs):
        tid = rng.randrange(len(names))
        isz = max(L + 1, int(rng.gauss(300, 20)))
        p = rng.randrange(0, max(1, lens[tid] - isz - 150))
        ef = extra_flags()
        recs.append(mk(tid, p, 99 | ef, tid, p + isz - L, isz))
        recs.append(mk(tid, p + isz - L, 147 | ef, tid, p, -isz))
    junctions = []
    # planted junctions with supporting discordant reads
    for _ in range(60):
        kind = rng.choice(("del", "dup", "inv_mp", "inv_pm", "ctx", "ctx_mp"))
        ta = rng.randrange(len(names))
        tb = ta if not kind.startswith("ctx") else (ta + 1 + rng.randrange(len(names) - 1)) % len(names)
        up = rng.randrange(1, lens[ta])
        if kind == "del": down = min(lens[tb] - 1, up + rng.randrange(1, 3000)); us, ds = "+", "+"
        elif kind == "dup": down = max(1, up - rng.randrange(1, 400)); us, ds = "+", "+"
        elif kind == "inv_mp" or kind == "ctx_mp": down = rng.randrange(1, lens[tb]); us, ds = "-", "+"
        elif kind == "inv_pm": down = rng.randrange(1, lens[tb]); us, ds = "+", "-"
        else: down = rng.randrange(1, lens[tb]); us, ds = "+", "+"
        junctions.append((names[ta], up, us, names[tb], down, ds))
        for _ in range(rng.randrange(0, 25)):
            lq = L
            jitter = rng.randrange(-40, 40)
            if us == "+" and ds == "+":
                p = up - rng.randrange(lq - 8, 420) + (jitter if rng.random() < 0.2 else 0)
                mp = down - 1 + rng.randrange(-8, 300)
                flag = rng.choice((97, 97, 97, 65, 81, 113, 161))
            elif us == "-":
                p = up - 1 + rng.randrange(-8, 320)
                mp = down - 1 + rng.randrange(-8, 300)
                flag = rng.choice((113, 113, 113, 97, 177))
            else:
                p = up - rng.randrange(lq - 8, 420)
                mp = down - lq - rng.randrange(-8, 300)
                flag = rng.choice((65, 65, 65, 97, 129))
            if p < 0 or mp < 0 or p >= lens[ta]:
                continue
            isz = rng.choice((0, mp - p, 5000, -5000, 300))
            recs.append(mk(ta, p, flag | extra_flags(), tb, mp, isz, cigar=rng.choice((f"{L}M", f"{L}M", f"{L}M", None))))
    # junctions without any support, duplicates, near contig ends, unknown contig, identical positions
    for _ in range(25):
        ta, tb = rng.randrange(len(names)), rng.randrange(len(names))
        junctions.append((names[ta], rng.randrange(1, lens[ta]), rng.choice("+-"), names[tb], rng.randrange(1, lens[tb]), "+"))
    junctions.append(junctions[3])
    junctions.append((names[0], 5, "+", names[0], 150, "+"))
    junctions.append((names[0], 120, "-", names[1], 60, "+"))
    junctions.append((names[0], lens[0] - 3, "+", names[1], lens[1] - 10, "+"))
    junctions.append((names[0], lens[0] - 30, "-", names[1], 100, "+"))
    junctions.append((names[1], 700, "+", names[1], 701, "+"))     # l == 0 windows
    junctions.append((names[1], 900, "+", names[1], 902, "+"))     # l == 1 windows
    junctions.append((names[1], 1200, "+", names[1], 1230, "+"))   # l == 29
    if with_small_contig:
        junctions.append((names[0], 2000, "+", "HBV", 150, "+"))   # wraps below the contig start
        junctions.append(("HBV", 3200, "+", names[0], 2500, "+"))  # runs past the contig end
        junctions.append(("HBV", 100, "-", names[0], 9000, "+"))
    junctions.append(("chrZ", 100, "+", names[0], 300, "+"))       # up_chr not in the header: keeps its previous count
    junctions.append((names[0], 3000, "+", "chrZ", 300, "+"))      # down_chr not in the header
    recs.sort(key=lambda r: (r["tid"], r["pos"]))

    def jkey(j):
        return (j[0], j[3], j[2], j[5], j[1], j[4])  # Junction::operator<, getsv.h:187-225
    junctions.sort(key=jkey)
    return names, lens, recs, junctions


def eqx_case():
    """reads whose CIGARs use '=' / 'X' (and an 'S' + '=' + 'M' mix) among plain M reads: pins what the depth pass of libbam 0.1.16 does with them"""
    names, lens = ["chrA", "chrB"], [20000, 5000]
    recs = []

    def mk(tid, pos, cigar, flag=99, mapq=60):
        lq = sum(l for l, op in bamio.parse_cigar(cigar) if op in (0, 1, 4, 7, 8))
        recs.append(dict(qname=f"q{len(recs)}", flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, mtid=tid, mpos=pos + 200, isize=280 + pos % 41, seq="A" * lq, qual=b"\x1e" * lq))
    for p in range(900, 1300, 7): mk(0, p, "100M")
    for p in range(1000, 1100, 3): mk(0, p, "100=")
    for p in range(1001, 1100, 5): mk(0, p, "40=5X55=")
    for p in range(1002, 1100, 11): mk(0, p, "30M10X60=")
    for p in range(1003, 1100, 13): mk(0, p, "10S40=50M")
    for p in range(1004, 1100, 17): mk(0, p, "20M5D10=3I20X30M")
    for p in range(300, 500, 9): mk(1, p, "50=50M", flag=163)
    for p in range(305, 500, 19): mk(1, p, "25X25M2D25=25M", flag=83, mapq=30)
    junctions = [("chrA", 1050, "+", "chrA", 1150, "+"), ("chrA", 1100, "+", "chrA", 1250, "+"), ("chrA", 950, "+", "chrA", 1201, "+"),
                 ("chrA", 1030, "-", "chrB", 400, "+"), ("chrB", 350, "+", "chrB", 460, "-")]
    # discordant pairs so that the junctions reach the SV table with their flank depths (some of the supporting reads use '=' too)
    for k, (uc, up, us, dc, dp, ds) in enumerate(junctions):
        ta, tb = names.index(uc), names.index(dc)
        for r in range(4):
            cigar = "100M" if r % 2 == 0 else "60=40M"
            if us == "+" and ds == "+":
                pos, mpos, flag = up - 130 - 9 * r, dp + 15 + r, 97
            elif us == "-":
                pos, mpos, flag = up + 10 + 7 * r, dp + 140 + r, 113
            else:
                pos, mpos, flag = up - 140 - 5 * r, dp - 100 - 120 - r, 65
            lq = 100
            recs.append(dict(qname=f"d{k}_{r}", flag=flag, tid=ta, pos=pos, mapq=60, cigar=cigar, mtid=tb, mpos=mpos, isize=0 if ta != tb else mpos - pos, seq="A" * lq, qual=b"\x1e" * lq))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    junctions.sort(key=lambda j: (j[0], j[3], j[2], j[5], j[1], j[4]))
    return names, lens, recs, junctions


def deep_case():
    """stacks of more than 8000 reads: pins the read cap of libbam 0.1.16's pileup (bam_plp_push) - single stacks, stacks that fill up over
    several start positions, a 500-bp stretch whose uncapped depth would be ~9500x with mixed CIGARs and filtered reads in between, a second deep contig"""
    rng = random.Random(99)
    names, lens = ["chrA", "chrB", "chrC"], [30000, 8000, 6000]
    recs = []

    def mk(tid, pos, cigar="100M", flag=99, mapq=60, mtid=None, mpos=None, isize=None):
        lq = sum(l for l, op in bamio.parse_cigar(cigar) if op in (0, 1, 4, 7, 8))
        recs.append(dict(qname=f"q{len(recs)}", flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, mtid=tid if mtid is None else mtid,
                         mpos=pos + 200 if mpos is None else mpos, isize=(280 + pos % 41) if isize is None else isize, seq="A" * lq, qual=b"\x1e" * lq))
    for p in range(200, 29000, 5): mk(0, p)                                   # background, depth 20
    for _ in range(9000): mk(0, 2000)                                         # one stack
    for _ in range(8000): mk(0, 5000)                                         # full stack, then ten more 10 bp further
    for _ in range(10): mk(0, 5010)
    for _ in range(5000): mk(0, 8000)                                         # two half stacks
    for _ in range(5000): mk(0, 8020)
    for _ in range(7990): mk(0, 11000)                                        # nearly full, then 5, then 20
    for _ in range(5): mk(0, 11007)
    for _ in range(20): mk(0, 11015)
    cig_pool = ["100M"] * 6 + ["50M", "30M10D60M", "20S80M", "40M20N40M", "60M3I37M", "10M1D10M1D70M", "100M", "75M25S"]
    for p in range(15000, 15500):                                             # ~9500x for 500 bp, mixed CIGARs, filtered reads in between
        for _ in range(110):
            r = rng.random()
            flag = 99 | (1024 if r < 0.04 else 0) | (256 if 0.04 <= r < 0.06 else 0) | (512 if 0.06 <= r < 0.07 else 0)
            mk(0, p, rng.choice(cig_pool), flag=flag, mapq=rng.choice((60, 60, 60, 60, 30, 20, 19, 0)))
    for p in range(3000, 3300):                                               # a second deep contig right after: the state starts afresh
        for _ in range(95): mk(1, p, rng.choice(("100M", "100M", "80M", "50M10D50M")))
    for p in range(100, 5000, 7): mk(2, p)
    junctions = [("chrA", 2050, "+", "chrA", 5050, "+"), ("chrA", 5012, "+", "chrA", 8030, "+"), ("chrA", 8025, "-", "chrA", 11010, "+"),
                 ("chrA", 11020, "+", "chrA", 15100, "+"), ("chrA", 15250, "+", "chrA", 15480, "+"), ("chrA", 15050, "-", "chrB", 3100, "+"),
                 ("chrB", 3250, "+", "chrC", 1000, "+"), ("chrA", 15400, "+", "chrB", 3010, "-"), ("chrA", 20000, "+", "chrA", 25000, "+")]
    for k, (uc, up, us, dc, dp, ds) in enumerate(junctions):
        ta, tb = names.index(uc), names.index(dc)
        for r in range(4):
            if us == "+" and ds == "+": pos, mpos, flag = up - 130 - 9 * r, dp + 15 + r, 97
            elif us == "-": pos, mpos, flag = up + 10 + 7 * r, dp + 140 + r, 113
            else: pos, mpos, flag = up - 140 - 5 * r, dp - 100 - 120 - r, 65
            mk(ta, pos, "100M", flag=flag, mtid=tb, mpos=mpos, isize=0 if ta != tb else mpos - pos)
    order = sorted(range(len(recs)), key=lambda i: (recs[i]["tid"], recs[i]["pos"]))   # stable: equal starts keep their creation order
    recs = [recs[i] for i in order]
    junctions.sort(key=lambda j: (j[0], j[3], j[2], j[5], j[1], j[4]))
    return names, lens, recs, junctions


def crafted_getsv():
    out = os.path.join(HERE, "getsv")
    os.makedirs(out, exist_ok=True)
    # header-only clip.bam and empty clip file for the harness
    for name, seed, kw in (("pairs1", 5, {}), ("pairs2", 6, dict(n_pairs=12000, L=150)), ("pairs3", 7, dict(n_pairs=3000, with_small_contig=False)), ("eqx", None, {}), ("deep", None, {})):
        names, lens, recs, junctions = getsv_case(seed, **kw) if seed is not None else (eqx_case() if name == "eqx" else deep_case())
        bam = os.path.join(out, f"{name}.bam")
        bamio.write_bam(bam, names, lens, recs)
        run([os.path.join(BIN, "bamidx"), bam])
        empty_bam = os.path.join(TMP, f"{name}.empty.clip.bam")
        bamio.write_bam(empty_bam, names, lens, [])
        empty_clip = os.path.join(TMP, "empty.clip")
        open(empty_clip, "w").close()
        jfile = os.path.join(out, f"{name}.junctions.txt")
        with open(jfile, "w") as f:
            for j in junctions:
                f.write(junction_row(*j, prev_abnormal=7 if j[0] == "chrZ" else 0))
        variants = [("", [])]
        if name == "pairs1":
            variants += [(".q0", ["-q", "0"]), (".L50", ["-L", "50"]), (".L1", ["-L", "1"])]
        for tag, flags in variants:
            pre = os.path.join(out, f"{name}{tag}")
            # -d 0 -f 0 -b 0: every junction with a discordant pair reaches the table with all depth columns
            cmd = [SEEKSV, "getsv", "-d", "0", "-f", "0", "-b", "0", "-T", "100000"] + flags + ["-B", jfile, empty_bam, bam, empty_clip, pre + ".sv", os.path.join(TMP, "x.fq")]
            with open(pre + ".stdout", "wb") as so, open(os.path.join(TMP, "stderr.txt"), "wb") as se:
                rc = subprocess.call(cmd, stdout=so, stderr=se)
            txt = open(os.path.join(TMP, "stderr.txt")).read()
            with open(pre + ".isize.txt", "w") as g:
                if "Mean insert size : " in txt:
                    g.write(txt.split("Mean insert size : ")[1].split()[0] + "\t" + txt.split("Mean deviation: ")[1].split()[0] + "\n")
                else:
                    g.write("NA\tNA\n")
            if rc != 0:
                print(f"  reference exited with {rc} on {name}{tag} (kept: this is reference behaviour)")
                with open(pre + ".rc", "w") as g:
                    g.write(f"{rc}\n")


# ------------------------------------------------------------------------------------------------
# 4. the synthetic generator's records through the real reference (inputs are regenerated by the tests,
#    only the reference's outputs are committed)
# ------------------------------------------------------------------------------------------------

SYNTH_CASES = {"synth30x": dict(genome_frac=1 / 2048, depth=30, n_sv=40), "synth300x": dict(genome_frac=1 / 8192, depth=300, n_sv=24),
               # BASELINE config 5 in small: human + HBV hybrid reference, virus integrations (two of them within 200 bp of the HBV contig's ends)
               "synthhbv": dict(genome_frac=1 / 8192, depth=60, n_sv=8, n_integrations=10)}


def synthetic():
    sys.path.insert(0, ROOT)
    from seeksv_amd import synth
    out = os.path.join(HERE, "synth")
    os.makedirs(out, exist_ok=True)
    for name, kw in SYNTH_CASES.items():
        w = synth.Workload(**kw)
        b = w.generate_host(0, w.n_total)
        bam = os.path.join(TMP, f"{name}.bam")
        bamio.soa_to_bam(bam, w.names, w.lens, b)
        run([os.path.join(BIN, "bamidx"), bam])
        run([SEEKSV, "getclip", "-o", name, bam], cwd=TMP)
        for ext in ("clip", "clip.fq"):
            with gzip.open(os.path.join(TMP, f"{name}.{ext}.gz"), "rb") as f, gzip.GzipFile(os.path.join(out, f"{name}.{ext}.txt.gz"), "wb", mtime=0) as g:
                g.write(f.read())
        empty_bam = os.path.join(TMP, f"{name}.empty.clip.bam")
        bamio.write_bam(empty_bam, w.names, [int(x) for x in w.lens], [])
        empty_clip = os.path.join(TMP, "empty.clip")
        open(empty_clip, "w").close()
        jfile = os.path.join(out, f"{name}.junctions.txt")
        with open(jfile, "w") as f:
            for j in w.junctions:
                f.write(junction_row(*j))
        pre = os.path.join(out, name)
        with open(pre + ".stdout", "wb") as so, open(os.path.join(TMP, "stderr.txt"), "wb") as se:
            subprocess.check_call([SEEKSV, "getsv", "-d", "0", "-f", "0", "-b", "0", "-T", "100000", "-B", jfile, empty_bam, bam, empty_clip, pre + ".sv",
                                   os.path.join(TMP, "x.fq")], stdout=so, stderr=se)
        txt = open(os.path.join(TMP, "stderr.txt")).read()
        with open(pre + ".isize.txt", "w") as g:
            g.write(txt.split("Mean insert size : ")[1].split()[0] + "\t" + txt.split("Mean deviation: ")[1].split()[0] + "\n")
        print(f"  {name}: {w.n_total} records, {len(w.junctions)} junctions")



# ------------------------------------------------------------------------------------------------
# 5. full pipeline on a synthetic sample with planted DEL / INV / TRA: getclip -> bwa mem against the
#    hash-generated reference -> getsv.  Exercises the junction stage (reverse-strand re-alignments,
#    translocations, MergeJunction) far beyond the three deletions of the bundled example.
# ------------------------------------------------------------------------------------------------

SYNTH_FULL = {"synthfull": dict(genome_frac=1 / 8192, depth=40, n_sv=24),
              "hbvfull": dict(genome_frac=1 / 8192, depth=60, n_sv=8, n_integrations=10)}   # the tumor of config 5
HBV_NORMAL = dict(genome_frac=1 / 8192, depth=30, n_sv=8, hbv=True)                          # its normal: same reference, same germline SVs, no virus


def synthetic_full():
    sys.path.insert(0, ROOT)
    from seeksv_amd import synth
    out = os.path.join(HERE, "synth")
    os.makedirs(out, exist_ok=True)
    bwa = os.path.join(TMP, "bwa")
    if not os.path.exists(bwa):
        shutil.copy(os.path.join(REF, "example/bin/bwa"), bwa)
        os.chmod(bwa, 0o755)
    for name, kw in SYNTH_FULL.items():
        w = synth.Workload(**kw)
        b = w.generate_host(0, w.n_total)
        bam = os.path.join(TMP, f"{name}.bam")
        bamio.soa_to_bam(bam, w.names, w.lens, b)
        run([os.path.join(BIN, "bamidx"), bam])
        fa = os.path.join(TMP, f"{name}.fa")
        with open(fa, "w") as f:
            f.write(w.reference_fasta())
        run([bwa, "index", fa], cwd=TMP)
        run([SEEKSV, "getclip", "-o", name, bam], cwd=TMP)
        run([bwa, "mem", fa, f"{name}.clip.fq.gz"], cwd=TMP, stdout=os.path.join(TMP, f"{name}.clip.sam"))
        run([os.path.join(BIN, "sam2bam"), f"{name}.clip.sam", os.path.join(out, f"{name}.clip.bam")], cwd=TMP)
        for tag, flags in (("", []), (".l90", ["-l", "90"]), (".loose", ["-f", "0", "-b", "0", "-d", "0"]), (".strict", ["-e", "10", "-b", "20"])):
            run([SEEKSV, "getsv"] + flags + [os.path.join(out, f"{name}.clip.bam"), bam, f"{name}.clip.gz", os.path.join(out, f"{name}{tag}.sv"), "x.fq"], cwd=TMP,
                stdout=os.path.join(out, f"{name}{tag}.stdout"))
        n_sv = sum(1 for l in open(os.path.join(out, f"{name}.sv")) if not l.startswith("@"))
        print(f"  {name}: {w.n_total} records, {len(w.junctions)} planted, {n_sv} SV rows")


# ------------------------------------------------------------------------------------------------
# 6. somatic: the synthetic sample as the "normal", and a tumor SV table made of the sample's own
#    junctions plus systematic variants of them, so that every branch of ReadTumorFileAndOutputSomaticInfo
#    (strand pair x microhomology known or -1 x which side has no clipped reads, shifted positions for the
#    anchored range searches, the two error messages) is taken.
# ------------------------------------------------------------------------------------------------

def somatic_table(sv_rows):
    out = []
    for k, row in enumerate(sv_rows):
        c = row.rstrip("\n").split("\t")
        out.append(c)
        for var in range(8):
            v = list(c)
            if var == 0:
                v[8], v[3] = "-1", "0"                      # only the right breakend clipped
            elif var == 1:
                v[8], v[7] = "-1", "0"                      # only the left breakend clipped
            elif var == 2:
                v[8] = "-1"                                 # both clipped but no microhomology: the reference prints a message
            elif var == 3:
                v[8], v[3], v[1] = "-1", "0", str(int(c[1]) - 3 - k % 5)   # range probe has to walk to the bin
            elif var == 4:
                v[8], v[7], v[5] = "-1", "0", str(int(c[5]) + 2 + k % 7)
            elif var == 5:
                v[8] = str(1 + k % 3)                       # bins move by the microhomology
            elif var == 6:
                v[8], v[3], v[1] = "-1", "0", str(int(c[1]) + 2)
            elif var == 7:
                v[8], v[7], v[5] = "-1", "0", str(int(c[5]) - 4)
            out.append(v)
        if k % 6 == 0:
            v = list(c); v[2], v[6] = "-", "-"; out.append(v)   # unsupported strand pair
        if k % 4 == 1:
            v = list(c); v[21] = c[21][:8]; v[22] = c[22][:9]; v[8], v[3] = "-1", "0"; out.append(v)  # sequences shorter than the 10-base anchor
    return out


def somatic_synth():
    sys.path.insert(0, ROOT)
    from seeksv_amd import synth
    out = os.path.join(HERE, "somatic")
    os.makedirs(out, exist_ok=True)
    name, kw = "synthfull", SYNTH_FULL["synthfull"]
    w = synth.Workload(**kw)
    bam = os.path.join(TMP, f"{name}.bam")
    if not os.path.exists(bam):
        bamio.soa_to_bam(bam, w.names, w.lens, w.generate_host(0, w.n_total))
    run([os.path.join(BIN, "bamidx"), bam])
    run([SEEKSV, "getclip", "-o", name, bam], cwd=TMP)
    lines = open(os.path.join(HERE, "synth", f"{name}.loose.sv")).read().splitlines()
    header, rows = lines[0], [l for l in lines[1:] if l]
    table = os.path.join(out, "tumor.sv")
    with open(table, "w") as f:
        f.write(header + "\n")
        for v in somatic_table(rows):
            f.write("\t".join(v) + "\n")
    for tag, flags in (("", []), (".n0", ["-n", "0"]), (".l0", ["-l", "0"]), (".l60", ["-l", "60"]), (".t05", ["-t", "0.5"]), (".m60", ["-m", "60"]), (".q0", ["-q", "0"])):
        run([SEEKSV, "somatic"] + flags + [bam, f"{name}.clip.gz", table, os.path.join(out, f"somatic{tag}.sv")], cwd=TMP, stderr=os.path.join(out, f"somatic{tag}.stderr"))
    n = sum(1 for _ in open(os.path.join(out, "somatic.sv")))
    print(f"  somatic: {len(open(table).read().splitlines())} table rows -> {n} output rows")


# ------------------------------------------------------------------------------------------------
# 7. BASELINE config 5 in small: tumor (human + HBV, planted integrations) against its normal -
#    getclip on both, the tumor's SV table from section 5 (bwa mem on the hybrid reference), somatic.
# ------------------------------------------------------------------------------------------------

def somatic_hbv():
    sys.path.insert(0, ROOT)
    from seeksv_amd import synth
    out = os.path.join(HERE, "somatic")
    os.makedirs(out, exist_ok=True)
    n = synth.Workload(**HBV_NORMAL)
    bam = os.path.join(TMP, "hbvnormal.bam")
    bamio.soa_to_bam(bam, n.names, n.lens, n.generate_host(0, n.n_total))
    run([os.path.join(BIN, "bamidx"), bam])
    run([SEEKSV, "getclip", "-o", "hbvnormal", bam], cwd=TMP)
    table = os.path.join(HERE, "synth", "hbvfull.loose.sv")
    for tag, flags in (("", []), (".n0", ["-n", "0"]), (".q0", ["-q", "0"])):
        run([SEEKSV, "somatic"] + flags + [bam, "hbvnormal.clip.gz", table, os.path.join(out, f"hbv.somatic{tag}.sv")], cwd=TMP, stderr=os.path.join(out, f"hbv.somatic{tag}.stderr"))
    rows = [l.split("\t") for l in open(os.path.join(out, "hbv.somatic.sv")) if not l.startswith("@")]
    print(f"  hbv somatic: {len(rows)} rows, {sum(1 for r in rows if 'HBV' in (r[0], r[4]))} with a viral end")


if __name__ == "__main__":
    if not os.path.exists(SEEKSV):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(TMP, exist_ok=True)
    if len(sys.argv) > 1:   # only the named sections, e.g. `make_golden.py lone_s_case`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    example()
    crafted_getclip()
    lone_s_case()
    lone_s2_case()
    unsorted_case()
    crafted_getsv()
    synthetic()
    synthetic_full()
    somatic_synth()
    somatic_hbv()
    print("goldens regenerated under", HERE)
