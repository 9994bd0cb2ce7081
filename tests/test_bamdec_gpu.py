"""-m gpu: the device-side BGZF inflate + BAM record decode (ssv_bamdec_*) against the host reader, array for array, and
through the getclip kernels against the reference goldens.  Chunk sizes are chosen so that records straddle chunks and blocks."""
import os

import numpy as np
import pytest

import bamio
import golden_util as G
from seeksv_amd import device, host, synth
from test_bam_reader import NAMES, LENS, _records

pytestmark = pytest.mark.gpu
KEYS = ("tid", "pos", "flag", "mapq", "n_cigar", "l_qseq", "mtid", "mpos", "isize", "xc")
NO_SEQ = np.uint64(2 ** 64 - 1)


def _host_all(path, keep_all_seq=False):
    out, unm = [], []
    with host.BamReader(path) as r:
        while True:
            b = r.read_batch(1 << 20, keep_all_seq)
            if b is None:
                break
            unm += r.unmapped()
            out.append(b)
    return out, unm


def _flatten(batches):
    """batches of owned arrays -> one comparable view: fixed columns, CIGARs per record, shipped bases+qualities per record"""
    cat = {k: np.concatenate([b[k] for b in batches]) if batches else np.zeros(0) for k in KEYS}
    cig, seqs, shipped = [], [], []
    for b in batches:
        co, nc = b["cigar_off"].astype(np.int64), b["n_cigar"].astype(np.int64)
        for i in range(len(nc)):
            cig.append(b["cigar"][co[i]:co[i] + nc[i]].tobytes())
        for i in np.nonzero(b["seq_off"] != NO_SEQ)[0]:
            lq, so = int(b["l_qseq"][i]), int(b["seq_off"][i])
            seqs.append(bytes(b["seqqual"][so:so + (lq + 1) // 2 + lq]))
        shipped.append(b["seq_off"] != NO_SEQ)
    cat["cigars"], cat["seqs"] = cig, seqs
    cat["shipped"] = np.concatenate(shipped) if shipped else np.zeros(0, bool)
    cat["max_ref_span"] = max([int(b["max_ref_span"]) for b in batches] or [0])
    return cat


def _runs_reference(flag, tid):
    runs, last = [], 0
    for i in range(len(tid)):
        if flag[i] & 12:
            continue
        if tid[i] != last:
            runs.append((i, int(tid[i])))
            last = int(tid[i])
    return runs


def _device_all(ctx, path, chunk_bytes, max_blocks, keep_all_seq=False, verify_crc=False):
    out, unm, runs, base, repaired = [], [], [], 0, 0
    with host.BamReader(path) as r:
        for b, info in ctx.bam_batches(r, chunk_bytes=chunk_bytes, max_blocks=max_blocks, keep_all_seq=keep_all_seq, verify_crc=verify_crc):
            if b.n:
                out.append(ctx.batch_to_host(b))
            unm += info["unmapped"]
            runs += [(base + i, t) for i, t in info["tid_runs"]]
            base += info["n_records"]
            repaired += info["repaired_blocks"]
    return out, unm, runs, repaired


@pytest.fixture(scope="module")
def ctx():
    with device.Context(0) as c:
        yield c


@pytest.fixture(scope="module")
def ragged(tmp_path_factory):
    d = tmp_path_factory.mktemp("bamdec")
    path = str(d / "ragged.bam")
    bamio.write_bam(path, NAMES, LENS, _records(6000, 11))  # blocks cut every 0xff00 bytes, anywhere inside a record
    return path


CHUNKS = [(64 << 20, 1 << 16), (1 << 17, 1 << 16), (64 << 20, 1), (64 << 20, 3), (200_000, 7)]


@pytest.mark.parametrize("chunk_bytes,max_blocks", CHUNKS, ids=[f"{c}B-{m}blk" for c, m in CHUNKS])
def test_device_decode_equals_host_reader_ragged(ctx, ragged, chunk_bytes, max_blocks):
    hb, hunm = _host_all(ragged)
    db, dunm, druns, _ = _device_all(ctx, ragged, chunk_bytes, max_blocks)
    h, d = _flatten(hb), _flatten(db)
    for k in KEYS + ("shipped",):
        assert np.array_equal(h[k], d[k]), k
    assert h["cigars"] == d["cigars"] and h["seqs"] == d["seqs"]
    assert d["max_ref_span"] >= 1 and h["max_ref_span"] == d["max_ref_span"]
    assert hunm == dunm and len(hunm) > 0
    assert druns == _runs_reference(h["flag"], h["tid"])


@pytest.mark.parametrize("sub,name", [("example", "cancer.sort.bam"), ("example", "normal.sort.bam"), ("getclip", "filters.bam"), ("getclip", "stress2.bam"), ("getsv", "pairs1.bam")])
@pytest.mark.parametrize("keep_all", [False, True], ids=["clipseq", "allseq"])
def test_device_decode_equals_host_reader_goldens(ctx, sub, name, keep_all):
    path = os.path.join(G.GOLDEN, sub, name)
    hb, hunm = _host_all(path, keep_all)
    db, dunm, druns, _ = _device_all(ctx, path, 1 << 18, 5, keep_all)
    h, d = _flatten(hb), _flatten(db)
    for k in KEYS + ("shipped",):
        assert np.array_equal(h[k], d[k]), k
    assert h["cigars"] == d["cigars"] and h["seqs"] == d["seqs"]
    assert hunm == dunm
    assert druns == _runs_reference(h["flag"], h["tid"])


def test_chunks_out_of_a_locked_mapping_of_the_file(ctx, ragged):
    """The no-copy way in (round 4): ssvh_bam_map_blocks hands a chunk out where it lies in a mapping of the file, the caller page-locks its pages (ssv_host_register:
    whole pages around it) and gives that pointer to ssv_bamdec_prefetch / ssv_bamdec_decode - the GPU's DMA engines fetch the file out of the page cache.  With
    ssv_bamdec_expect before the first (small) chunk.  Same batches as the host reader's; chunk sizes of a few blocks so that the three mappings take turns."""
    import ctypes as C
    from seeksv_amd import _abi
    lib, hl = ctx._lib, _abi.host_lib()
    hl.ssvh_bam_map_blocks.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(_abi.BgzfBlock), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    hb, _ = _host_all(ragged)
    page = os.sysconf("SC_PAGESIZE")
    out, locked = [], []
    with host.BamReader(ragged) as r:
        first = C.c_uint64()
        assert hl.ssvh_bam_raw_begin(r.handle, C.byref(first)) == 0
        ctx._check(lib.ssv_bamdec_begin(ctx._h, len(r.target_names), first.value), "ssv_bamdec_begin")
        ctx._check(lib.ssv_bamdec_expect(ctx._h, 1 << 20), "ssv_bamdec_expect")
        blocks = (_abi.BgzfBlock * 64)()
        sizes = [70_000, 140_000, 200_000]
        k = 0
        while True:
            nb, nbytes, ptr = C.c_int64(), C.c_size_t(), C.c_void_p()
            assert hl.ssvh_bam_map_blocks(r.handle, sizes[k % 3], 1 << 30, blocks, 64, C.byref(nb), C.byref(ptr), C.byref(nbytes)) == 0, hl.ssvh_last_error()
            if nb.value == 0:
                break
            lo = ptr.value & ~(page - 1)
            hi = (ptr.value + nbytes.value + page - 1) & ~(page - 1)
            rc = lib.ssv_host_register(C.c_void_p(lo), hi - lo)
            if rc == 0:
                locked.append(lo)
                if k % 2:   # every other chunk announced first: its bytes travel on the upload stream
                    ctx._check(lib.ssv_bamdec_prefetch(ctx._h, ptr, nbytes.value), "ssv_bamdec_prefetch")
            b = _abi.Batch()
            ctx._check(lib.ssv_bamdec_decode(ctx._h, ptr, nbytes.value, blocks, nb.value, 0, C.byref(b)), "ssv_bamdec_decode")
            if b.n:
                out.append(ctx.batch_to_host(b))
            if rc == 0 and len(locked) > 2:   # (a chunk's pages are let go of once it is decoded; two stay for the overlap of neighbouring page ranges in other mappings)
                assert lib.ssv_host_unregister(C.c_void_p(locked.pop(0))) == 0
            k += 1
        b = _abi.Batch()
        ctx._check(lib.ssv_bamdec_decode(ctx._h, None, 0, None, 0, 0, C.byref(b)), "ssv_bamdec_decode")
        for lo in locked:
            assert lib.ssv_host_unregister(C.c_void_p(lo)) == 0
    assert k >= 3
    h, d = _flatten(hb), _flatten(out)
    for key in KEYS + ("shipped",):
        assert np.array_equal(h[key], d[key]), key
    assert h["cigars"] == d["cigars"] and h["seqs"] == d["seqs"]


def test_announced_chunks_equal_copied_chunks(ctx, ragged):
    """ssv_bamdec_prefetch: chunks announced ahead (bytes on the upload stream into one of two device slots) decode to the same batches as
    chunks copied by the decode call itself; announcing more than two, or a chunk that is never decoded, is harmless"""
    a = _flatten(_device_all(ctx, ragged, 200_000, 7)[0])                      # bam_batches announces chunk k+1 before it decodes chunk k
    import ctypes as C
    out = []
    with host.BamReader(ragged) as r:
        for b, _ in ctx.bam_batches(r, chunk_bytes=200_000, max_blocks=7, prefetch=False):
            if b.n:
                out.append(ctx.batch_to_host(b))
    c = _flatten(out)
    for k in KEYS + ("shipped",):
        assert np.array_equal(a[k], c[k]), k
    assert a["cigars"] == c["cigars"] and a["seqs"] == c["seqs"]
    # three announcements in a row (the third finds both slots taken and is left to its decode call), then a fresh file
    pin = device.PinnedArrays()
    bufs = [pin.empty(4096, np.uint8) for _ in range(3)]
    for x in bufs:
        x[:] = 7
        assert ctx._lib.ssv_bamdec_prefetch(ctx._h, C.c_void_p(x.ctypes.data), x.size) == 0
    b = _flatten(_device_all(ctx, ragged, 1 << 17, 1 << 16)[0])
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), k


def test_device_decode_feeds_getclip(ctx):
    """BAM bytes -> device inflate/decode -> clip kernels, nothing decoded on the host: the reference's clip rows"""
    path = os.path.join(G.GOLDEN, "example", "cancer.sort.bam")
    ctx.clip_begin(0.9, 1, False)
    with host.BamReader(path) as r:
        names = r.target_names
        for b, _ in ctx.bam_batches(r, chunk_bytes=1 << 18, max_blocks=4):
            if b.n:
                ctx.clip_scan(b)
    table = ctx.clip_cluster()
    rows, fq = host.format_clip_outputs(table, names)
    assert rows == G.read_text("example", "cancer.clip.txt")
    assert fq == G.read_text("example", "cancer.clip.fq.txt")


def test_device_decode_synthetic_large(ctx, tmp_path):
    """a few hundred BGZF blocks written by the host writer; every speculated record start must be right or repaired"""
    w = synth.Workload(genome_frac=1 / 1024, depth=30, n_sv=20)
    b = w.generate_host(0, w.n_total)
    path = str(tmp_path / "s.bam")
    host.write_bam(path, w.names, w.lens, [b])
    db, _, _, repaired = _device_all(ctx, path, 8 << 20, 1 << 16)
    d = _flatten(db)
    for k in ("tid", "pos", "flag", "mapq", "n_cigar", "l_qseq", "mtid", "mpos", "isize"):
        assert np.array_equal(d[k], b[k]), k
    assert len(d["tid"]) == w.n_total
    assert repaired <= 2 * len(db) + 2  # only chunk heads may need the hand-followed chain


def test_device_decode_rejects_damage(ctx, ragged, tmp_path):
    raw = bytearray(open(ragged, "rb").read())
    raw[len(raw) // 2] ^= 0x5A  # inside some block's deflate payload
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises((device.SeeksvError, IOError)):
        _device_all(ctx, bad, 64 << 20, 1 << 16)


def _pattern_records(n, seed):
    """records whose bases and qualities are runs and repeating patterns of every period 1..70 and of every length up to a few hundred bytes -
    in the deflate stream: matches with dist < len (a pattern that feeds itself), dist == len, long and short matches, next to plain literals"""
    import random
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        l = rng.choice([30, 76, 151, 151, 259, 600, 2000])
        period = 1 + i % 70
        unit_q = bytes(rng.randrange(42) for _ in range(period))
        unit_s = "".join(rng.choice("ACGT") for _ in range(2 * period))
        kind = i % 4
        qual = (unit_q * (l // period + 1))[:l] if kind != 3 else bytes(rng.randrange(42) for _ in range(l))
        seq = (unit_s * (l // (2 * period) + 1))[:l] if kind in (1, 2) else "".join(rng.choice("ACGTN") for _ in range(l))
        if kind == 2 and l > 40:  # a pattern broken by a few literals every now and then
            q = bytearray(qual)
            for k in range(17, l, 37):
                q[k] = rng.randrange(42)
            qual = bytes(q)
        recs.append(dict(qname=f"p{i}", flag=rng.choice([99, 147, 83, 163]), tid=0, pos=10 + 3 * i, mapq=60, cigar=f"12S{l - 12}M" if i % 3 else f"{l - 9}M9S",
                         mtid=0, mpos=100 + 3 * i, isize=rng.randrange(-900, 900), seq=seq, qual=qual, aux=b"NMC\x01"))
    return recs


@pytest.mark.parametrize("chunk_bytes,max_blocks", [(64 << 20, 1 << 16), (64 << 20, 2)], ids=["one-chunk", "2blk"])
def test_device_decode_match_shapes(ctx, tmp_path, chunk_bytes, max_blocks):
    """the inflate kernels' copy paths one by one (k_bgzf_resolve: a round's independent matches, matches inside the round, repeating patterns,
    long matches), on files written by zlib at level 6 (bamio) and by the repository's own writer"""
    recs = _pattern_records(9000, 5)
    p1 = str(tmp_path / "patterns.bam")
    bamio.write_bam(p1, NAMES, [10_000_000] * len(NAMES), recs)
    for path in (p1,):
        hb, hunm = _host_all(path, True)
        db, dunm, druns, _ = _device_all(ctx, path, chunk_bytes, max_blocks, True)
        h, d = _flatten(hb), _flatten(db)
        for k in KEYS + ("shipped",):
            assert np.array_equal(h[k], d[k]), k
        assert h["cigars"] == d["cigars"] and h["seqs"] == d["seqs"]
        assert len(h["seqs"]) == len(recs)


@pytest.mark.parametrize("mode", ["wave", "win"])
def test_resolve_modes(mode):
    """Pass 2 of the inflate, a wavefront per BGZF block, in its two forms: working in global memory (k_bgzf_resolve_wave: rounds of 64 tokens, the earlier holes a
    source touches as a lane range found by binary search, readiness by one AND with the ballot of the open lanes - the reference form) and with a window of the
    block in LDS (k_bgzf_resolve_win, the default: the round's bytes in and out with coalesced 16-byte accesses, matches copied inside the window).  Both,
    forced, through the same device-decode == host-reader checks: every match shape (near, far, overlapping its own hole, 3..258 bytes), every kind of deflate
    block, records and blocks straddling chunks, damaged input"""
    import subprocess
    import sys
    if os.environ.get("SSV_RESOLVE") or os.environ.get("SSV_TOKENS"):
        pytest.skip("already inside a mode run")
    env = dict(os.environ, SSV_RESOLVE=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "ragged or goldens or synthetic_large or match_shapes or rejects_damage or longer_than or every_deflate or fuzzed"], env=env,
                       capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("chunk_bytes", [64 << 20, 1 << 20], ids=["one-chunk", "1MB-chunks"])
def test_device_decode_record_longer_than_64_blocks(ctx, tmp_path, chunk_bytes):
    """a read of 5.5 M bases = one 8 MB record = ~130 BGZF blocks without a record start in them: the per-block check of the speculated record starts
    looks back over at most 64 such blocks and then leaves the chunk to the sequential walk (k_stitch_check / k_stitch_blocks); same batches as the host reader's"""
    rng = np.random.default_rng(9)
    recs = _records(300, 21)
    n = 5_500_000
    long = dict(qname="ultralong", flag=0, tid=0, pos=1000, mapq=60, cigar=[(100, 0), (n - 100, 4)], mtid=-1, mpos=-1, isize=0,
                seq="".join(np.array(list("ACGT"))[rng.integers(0, 4, n)]), qual=bytes(rng.integers(2, 41, n, dtype=np.uint8)))
    path = str(tmp_path / "long.bam")
    bamio.write_bam(path, NAMES, LENS, recs[:150] + [long] + recs[150:])
    hb, hunm = _host_all(path)
    db, dunm, druns, _ = _device_all(ctx, path, chunk_bytes, 1 << 16)
    h, d = _flatten(hb), _flatten(db)
    for k in KEYS + ("shipped",):
        assert np.array_equal(h[k], d[k]), k
    assert h["cigars"] == d["cigars"] and h["seqs"] == d["seqs"] and hunm == dunm
    assert int(np.max(h["l_qseq"])) == n and any(len(s) == (n + 1) // 2 + n for s in d["seqs"])


@pytest.mark.parametrize("mode", ["wave", "lanes"])
def test_tokens_modes(mode):
    """Pass 1 of the inflate has two forms: a wavefront per BGZF block - 64 lanes decode the block's symbols speculatively from guessed bit offsets and
    re-synchronise, tables shared in LDS (inflate_wave.h; the default up to 48 K blocks per chunk) - and a lane per block (the default above that).  Both,
    forced, through the same device-decode == host-reader checks: stored, fixed and dynamic blocks, every match shape, records and blocks straddling
    chunks, damaged input"""
    import subprocess
    import sys
    if os.environ.get("SSV_RESOLVE") or os.environ.get("SSV_TOKENS"):
        pytest.skip("already inside a mode run")
    env = dict(os.environ, SSV_TOKENS=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "ragged or goldens or synthetic_large or match_shapes or rejects_damage or longer_than or every_deflate or fuzzed"], env=env,
                       capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def _bam_with_block_kinds(path, records, seed):
    """a BAM whose BGZF blocks are compressed every which way zlib can: stored (level 0), fixed Huffman (Z_FIXED), Huffman only, run-length, levels 1-9,
    and payloads of SEVERAL deflate blocks (full flushes inside a block: a dynamic block, an empty stored block, another dynamic block ...); block payloads
    of 1 byte to 65280 bytes"""
    import struct
    import zlib
    rng = np.random.default_rng(seed)
    raw = bytearray()
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(NAMES, LENS))
    raw += b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(NAMES))
    for n, l in zip(NAMES, LENS):
        nb = n.encode() + b"\0"
        raw += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    for r in records:
        raw += bamio.encode_record(r)
    kinds = [(0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (4, zlib.Z_FILTERED), "flushes"]
    with open(path, "wb") as f:
        at, k = 0, 0
        while at < len(raw):
            size = int(rng.choice([1, 7, 300, 4000, 30000, 0xff00]))
            data = bytes(raw[at:at + size])
            at += len(data)
            kind = kinds[k % len(kinds)]
            k += 1
            if kind == "flushes":
                c = zlib.compressobj(6, zlib.DEFLATED, -15)
                comp = b""
                for j in range(0, len(data), 1500):
                    comp += c.compress(data[j:j + 1500]) + c.flush(zlib.Z_FULL_FLUSH if (j // 1500) % 2 == 0 else zlib.Z_SYNC_FLUSH)
                comp += c.flush()
            else:
                c = zlib.compressobj(kind[0], zlib.DEFLATED, -15, 8, kind[1])
                comp = c.compress(data) + c.flush()
            if len(comp) + 26 > 65536:  # (stored random-ish bytes can outgrow a BGZF block: halve it)
                at -= len(data)
                size = len(data) // 2
                data = bytes(raw[at:at + size])
                at += len(data)
                c = zlib.compressobj(0, zlib.DEFLATED, -15)
                comp = c.compress(data) + c.flush()
            hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25)
            f.write(hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
        f.write(bamio.BGZF_EOF)


@pytest.mark.parametrize("chunk_bytes,max_blocks", [(64 << 20, 1 << 16), (1 << 18, 5)], ids=["one-chunk", "256KB-5blk"])
def test_device_decode_every_deflate_block_kind(ctx, tmp_path, chunk_bytes, max_blocks):
    """stored, fixed, dynamic, Huffman-only and run-length blocks, several deflate blocks inside one BGZF block (with empty stored blocks between them), tiny
    and full-size payloads: both forms of pass 1 (this test runs again under SSV_TOKENS=wave and =lanes in test_tokens_modes) against the host reader"""
    path = str(tmp_path / "kinds.bam")
    _bam_with_block_kinds(path, _records(4000, 31) + _pattern_records(3000, 7), 3)
    hb, hunm = _host_all(path, True)
    db, dunm, druns, _ = _device_all(ctx, path, chunk_bytes, max_blocks, True)
    h, d = _flatten(hb), _flatten(db)
    for k in KEYS + ("shipped",):
        assert np.array_equal(h[k], d[k]), k
    assert h["cigars"] == d["cigars"] and h["seqs"] == d["seqs"] and hunm == dunm
    assert len(h["tid"]) == 7000


def test_verify_crc_refuses_what_the_structure_hides(ctx, tmp_path):
    """ssv_bamdec_verify_crc (`-C`): damage that leaves a block's deflate structure valid - one literal turned into another of the same code length, a byte of
    a stored block - inflates to the right number of wrong bytes.  libbam 0.1.16 hands them out and so does the device decoder by default (the damaged file
    decodes to what the host reader, which checks no CRC either, makes of it); with the check on, the decoder refuses the chunk and names the CRC32.  The
    undamaged file passes with the check on, under every chunking."""
    import struct
    import zlib
    path = str(tmp_path / "f.bam")
    _bam_with_block_kinds(path, _records(1500, 5) + _pattern_records(800, 9), 11)
    raw = open(path, "rb").read()
    want = _flatten(_host_all(path, True)[0])
    for chunk_bytes, max_blocks in ((64 << 20, 1 << 16), (1 << 17, 1 << 16), (64 << 20, 3)):
        got = _flatten(_device_all(ctx, path, chunk_bytes, max_blocks, True, verify_crc=True)[0])
        for k in KEYS + ("shipped",):
            assert np.array_equal(want[k], got[k]), k
    blocks, at = [], 0
    while at < len(raw):
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        if bsize > 28:
            blocks.append((at, bsize))
        at += bsize
    rng = np.random.default_rng(23)
    silent = 0
    for _ in range(400):
        at, bsize = blocks[int(rng.integers(1, len(blocks)))]   # (not the block with the BAM header)
        b = bytearray(raw)
        i = at + 18 + int(rng.integers(0, bsize - 26))
        b[i] ^= 1 << int(rng.integers(0, 8))
        isize, crc = struct.unpack_from("<I", b, at + bsize - 4)[0], struct.unpack_from("<I", b, at + bsize - 8)[0]
        try:
            z = zlib.decompressobj(-15)
            data = z.decompress(bytes(b[at + 18:at + bsize - 8]))
            if not (len(data) == isize and z.eof and zlib.crc32(data) != crc):
                continue
        except zlib.error:
            continue
        bad = str(tmp_path / "bad.bam")
        open(bad, "wb").write(bytes(b))
        try:
            hb = _flatten(_host_all(bad, True)[0])
        except Exception:
            continue   # (the wrong bytes fell into a record's header: both readers refuse the record, CRC or not)
        db = _flatten(_device_all(ctx, bad, 1 << 19, 9, True)[0])   # silently: the same wrong bytes as the host reader's
        for k in KEYS + ("shipped",):
            assert np.array_equal(hb[k], db[k]), k
        assert hb["seqs"] == db["seqs"]
        with pytest.raises(device.SeeksvError, match="CRC32"):
            _device_all(ctx, bad, 1 << 19, 9, True, verify_crc=True)
        silent += 1
        if silent >= 12:
            break
    assert silent >= 6, "too few structure-valid damages found"
    got = _flatten(_device_all(ctx, path, 1 << 19, 9, True, verify_crc=True)[0])   # the decoder's state is intact afterwards
    for k in KEYS + ("shipped",):
        assert np.array_equal(want[k], got[k]), k


def test_device_decode_fuzzed_payloads(ctx, tmp_path):
    """120 copies of a BAM with one to three random bytes of the deflate payloads overwritten (headers and trailers left alone): the decoder must refuse
    the file or - when the damage happens to decode to the same number of bytes - hand out batches, and never fault, hang or corrupt its own state: the
    undamaged file still decodes to the host reader's batches afterwards.  (Runs again under both forms of pass 1 in test_tokens_modes.)"""
    import struct
    import zlib
    path = str(tmp_path / "f.bam")
    _bam_with_block_kinds(path, _records(1500, 5) + _pattern_records(800, 9), 11)
    raw = open(path, "rb").read()
    spans, at = [], 0
    while at < len(raw):  # the payload of every BGZF block
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        if bsize > 28:
            spans.append((at + 18, at + bsize - 8))
        at += bsize
    rng = np.random.default_rng(17)
    refused = 0
    for k in range(120):
        b = bytearray(raw)
        for _ in range(int(rng.integers(1, 4))):
            lo, hi = spans[int(rng.integers(0, len(spans)))]
            b[int(rng.integers(lo, hi))] = int(rng.integers(0, 256))
        bad = str(tmp_path / "bad.bam")
        open(bad, "wb").write(bytes(b))
        # zlib's verdict on every block of the damaged file: the device decoders apply zlib's rules to a block's structure (over-subscribed and
        # incomplete codes, missing end-of-block code, distances in front of the block, symbols that do not exist, input or output that ends early),
        # so whatever form of pass 1 runs refuses every file zlib refuses
        zlib_ok, at = True, 0
        while at < len(b) and zlib_ok:
            bsize = struct.unpack_from("<H", b, at + 16)[0] + 1
            isize = struct.unpack_from("<I", b, at + bsize - 4)[0]
            try:
                z = zlib.decompressobj(-15)
                zlib_ok = len(z.decompress(bytes(b[at + 18:at + bsize - 8]))) == isize and z.eof
            except zlib.error:
                zlib_ok = False
            at += bsize
        try:
            _device_all(ctx, bad, 1 << 19, 9)
            assert zlib_ok, f"fuzz case {k}: the device decoder accepted a file zlib refuses"
        except (device.SeeksvError, IOError):
            refused += 1
    # (a byte that turns one literal into another - in a stored block, or a code of the same length - inflates to the same number of bytes: like libbam 0.1.16's
    # reader, the decoder does not check the blocks' CRC32; such files come through as batches with one wrong byte, or are refused later by the record decoder)
    assert refused >= 60
    hb, hunm = _host_all(path, True)
    db, dunm, _, _ = _device_all(ctx, path, 1 << 19, 9, True)
    h, d = _flatten(hb), _flatten(db)
    for key in KEYS + ("shipped",):
        assert np.array_equal(h[key], d[key]), key
    assert h["cigars"] == d["cigars"] and h["seqs"] == d["seqs"] and hunm == dunm
