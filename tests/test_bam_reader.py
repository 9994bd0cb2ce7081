"""Host BAM reader (seeksv_amd/host/bam_reader.cpp): the batches must not depend on how the BGZF blocks cut the record stream, on the
batch size, or on read-ahead.  Stands where the reference relies on samtools' samread() (sam/sam.h:73)."""
import os
import random
import struct

import numpy as np
import pytest

import bamio
from seeksv_amd import host

NAMES, LENS = ["chr1", "chr2", "chrX"], [100000, 50000, 30000]
KEYS = ("tid", "pos", "flag", "mapq", "n_cigar", "l_qseq", "mtid", "mpos", "isize", "xc")


def _records(n, seed):
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        l = rng.choice([1, 2, 35, 100, 151, 3000 if i % 97 == 0 else 76])
        shape = rng.randrange(5)
        if shape == 0:
            cig = f"{l}M"
        elif shape == 1 and l > 20:
            cig = f"10S{l - 10}M"
        elif shape == 2 and l > 20:
            cig = f"{l - 7}M7S"
        elif shape == 3 and l > 30:
            cig = f"5S{l - 25}M3I4M2D8M5S"
        else:
            cig = f"{l}M"
        flag = rng.choice([99, 147, 83, 163, 65, 129, 73, 133, 1024 + 99])
        aux = b"XCi" + struct.pack("<i", 1) if rng.random() < 0.1 else b"NMC\x01"
        recs.append(dict(qname=f"read{i}" + "x" * rng.randrange(0, 40), flag=flag, tid=rng.randrange(3), pos=rng.randrange(20000), mapq=rng.randrange(61), cigar=cig,
                         mtid=rng.randrange(3), mpos=rng.randrange(20000), isize=rng.randrange(-900, 900), seq="".join(rng.choice("ACGTN") for _ in range(l)),
                         qual=None if rng.random() < 0.05 else bytes(rng.randrange(42) for _ in range(l)), aux=aux))
    return recs


def _read_all(path, batch, readahead):
    out, unm = [], []
    with host.BamReader(path, readahead=readahead) as r:
        while True:
            b = r.read_batch(batch)
            if b is None:
                break
            unm += r.unmapped()
            out.append(b)
    cat = {k: np.concatenate([x[k] for x in out]) for k in KEYS}
    cat["cigar"] = np.concatenate([x["cigar"][:int(x["cigar_off"][-1]) + int(x["n_cigar"][-1])] for x in out])
    seqs = []
    for x in out:
        for i in np.nonzero(x["seq_off"] != np.uint64(2 ** 64 - 1))[0]:
            lq, so = int(x["l_qseq"][i]), int(x["seq_off"][i])
            seqs.append(bytes(x["seqqual"][so:so + (lq + 1) // 2 + lq]))
    cat["seqs"] = seqs
    cat["unmapped"] = unm
    return cat


@pytest.fixture(scope="module")
def bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("bamreader")
    recs = _records(6000, 11)
    path = str(d / "ragged.bam")
    bamio.write_bam(path, NAMES, LENS, recs)  # blocks cut every 0xff00 bytes, wherever that falls inside a record
    return path, recs


def test_batches_independent_of_batch_size_and_readahead(bam):
    path, recs = bam
    base = _read_all(path, 1 << 20, False)
    assert len(base["tid"]) == len(recs)
    assert list(base["pos"]) == [r["pos"] for r in recs]
    assert list(base["isize"]) == [r["isize"] for r in recs]
    soft = [r for r in recs if "S" in r["cigar"]]
    assert len(base["seqs"]) == len(soft)
    assert [int(x) for x in base["xc"]] == [int("S" in r["cigar"] and r["aux"].startswith(b"XC")) for r in recs]
    n_unm = sum(1 for r in recs if r["flag"] & 12)
    assert len(base["unmapped"]) == n_unm and n_unm > 0
    for batch, ra in ((1, False), (7, True), (1000, True), (4096, False), (1 << 20, True)):
        if batch == 1:
            continue  # one record per batch is covered by the 7-record case; keeps the test fast
        got = _read_all(path, batch, ra)
        for k in KEYS + ("cigar",):
            assert np.array_equal(got[k], base[k]), (k, batch, ra)
        assert got["seqs"] == base["seqs"] and got["unmapped"] == base["unmapped"], (batch, ra)


def test_truncated_and_corrupt_inputs_fail_loudly(bam, tmp_path):
    path, _ = bam
    raw = open(path, "rb").read()
    cut = str(tmp_path / "cut.bam")
    # drop the last data block and the EOF marker: the stream then ends inside a record
    blocks, o = [], 0
    while o < len(raw):
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        blocks.append(raw[o:o + bsize])
        o += bsize
    open(cut, "wb").write(b"".join(blocks[:-2]))
    with pytest.raises(IOError):
        _read_all(cut, 1000, False)
    with pytest.raises(IOError):
        _read_all(cut, 1000, True)
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(raw[:len(raw) // 2] + b"\0" * 64 + raw[len(raw) // 2 + 64:])
    with pytest.raises(IOError):
        _read_all(bad, 1000, False)


def test_next_record_refused_during_readahead(bam):
    path, _ = bam
    import ctypes as C
    lib = host._abi.host_lib()
    rec = (C.c_uint8 * 256)()  # room for an ssvh_record; the call must fail before it writes one
    with host.BamReader(path, readahead=True) as r:
        assert r.read_batch(100) is not None
        assert lib.ssvh_bam_next_record(r.handle, C.byref(rec)) == -1
        assert b"read-ahead" in lib.ssvh_last_error()


def test_mapped_chunks_equal_copied_chunks(bam):
    """ssvh_bam_map_blocks (the chunk's bytes handed out where they lie in a mapping of the file: what the device decoder's reader page-locks and
    lets the GPU fetch by DMA) cuts the file into the same chunks, with the same block tables, as ssvh_bam_read_blocks (bytes copied into the
    caller's buffer) - at chunk sizes from one block up, including sizes that end a chunk in the middle of a block header"""
    import ctypes as C
    from seeksv_amd import _abi
    path, _ = bam
    lib = _abi.host_lib()
    lib.ssvh_bam_map_blocks.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(_abi.BgzfBlock), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    size = os.path.getsize(path)
    for max_bytes, max_inflated, max_blocks in [(1 << 20, 1 << 30, 1 << 16), (70000, 1 << 30, 1 << 16), (1 << 20, 100000, 1 << 16), (1 << 20, 1 << 30, 3), (200001, 150000, 5), (size + 100, 1 << 40, 1 << 16)]:
        chunks = []
        for mapped in (False, True):
            h = C.c_void_p()
            assert lib.ssvh_bam_open(path.encode(), C.byref(h)) == 0
            first = C.c_uint64()
            assert lib.ssvh_bam_raw_begin(h, C.byref(first)) == 0
            blocks = (_abi.BgzfBlock * max_blocks)() if max_blocks < 100 else (_abi.BgzfBlock * 4096)()
            buf = (C.c_uint8 * (max_bytes + 8))()
            got = []
            while True:
                nb, nbytes, ptr = C.c_int64(), C.c_size_t(), C.c_void_p()
                if mapped:
                    assert lib.ssvh_bam_map_blocks(h, max_bytes, max_inflated, blocks, min(max_blocks, 4096), C.byref(nb), C.byref(ptr), C.byref(nbytes)) == 0, lib.ssvh_last_error()
                    data = C.string_at(ptr, nbytes.value) if nbytes.value else b""
                else:
                    assert lib.ssvh_bam_read_blocks(h, buf, max_bytes + 8, max_inflated, blocks, min(max_blocks, 4096), C.byref(nb), C.byref(nbytes)) == 0, lib.ssvh_last_error()
                    data = bytes(buf[:nbytes.value])
                if nb.value == 0:
                    break
                got.append((data, [(blocks[k].c_off, blocks[k].c_len, blocks[k].u_len) for k in range(nb.value)]))
            lib.ssvh_bam_close(h)
            chunks.append(got)
        assert len(chunks[0]) == len(chunks[1]) and len(chunks[0]) >= 1
        for (d0, b0), (d1, b1) in zip(*chunks):
            assert b0 == b1 and d0 == d1
        assert sum(len(d) for d, _ in chunks[1]) in (size, size - 28)  # (the end-of-file block may or may not be part of the last chunk)


def test_mapped_reader_refuses_a_truncated_file(bam, tmp_path):
    """a file cut short (or replaced by a shorter one) while the mapped reader works on it: its next chunk would touch pages of the mapping behind the
    file's end - SIGBUS in the header walk or the DMA; the reader notices the size before and reports an error"""
    import ctypes as C
    import shutil
    from seeksv_amd import _abi
    path, _ = bam
    lib = _abi.host_lib()
    lib.ssvh_bam_map_blocks.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(_abi.BgzfBlock), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.ssvh_last_error.restype = C.c_char_p
    mine = str(tmp_path / "cut.bam")
    shutil.copy(path, mine)
    h = C.c_void_p()
    assert lib.ssvh_bam_open(mine.encode(), C.byref(h)) == 0
    first = C.c_uint64()
    assert lib.ssvh_bam_raw_begin(h, C.byref(first)) == 0
    blocks = (_abi.BgzfBlock * 4096)()
    nb, nbytes, ptr = C.c_int64(), C.c_size_t(), C.c_void_p()
    assert lib.ssvh_bam_map_blocks(h, 70000, 1 << 30, blocks, 4096, C.byref(nb), C.byref(ptr), C.byref(nbytes)) == 0 and nb.value >= 1
    os.truncate(mine, os.path.getsize(mine) // 2)
    assert lib.ssvh_bam_map_blocks(h, 1 << 20, 1 << 30, blocks, 4096, C.byref(nb), C.byref(ptr), C.byref(nbytes)) != 0
    assert b"truncated" in lib.ssvh_last_error()
    lib.ssvh_bam_close(h)


def _block_table_digest(path):
    """sha256 over the block tables of the file read as ONE chunk (ssvh_bam_read_blocks), and the number of blocks"""
    import ctypes as C
    import hashlib
    from seeksv_amd import _abi
    lib = _abi.host_lib()
    size = os.path.getsize(path)
    h = C.c_void_p()
    assert lib.ssvh_bam_open(path.encode(), C.byref(h)) == 0
    first = C.c_uint64()
    assert lib.ssvh_bam_raw_begin(h, C.byref(first)) == 0
    max_blocks = size // 64 + 16
    blocks = (_abi.BgzfBlock * max_blocks)()
    buf = (C.c_uint8 * (size + 64))()
    dig, n = hashlib.sha256(), 0
    while True:
        nb, nbytes = C.c_int64(), C.c_size_t()
        assert lib.ssvh_bam_read_blocks(h, buf, size + 64, 1 << 40, blocks, max_blocks, C.byref(nb), C.byref(nbytes)) == 0, lib.ssvh_last_error()
        if nb.value == 0:
            break
        dig.update(np.ctypeslib.as_array(C.cast(blocks, C.POINTER(C.c_uint8)), shape=(nb.value * C.sizeof(_abi.BgzfBlock),)).tobytes())
        dig.update(bytes(buf[:nbytes.value])[-64:])
        n += nb.value
    lib.ssvh_bam_close(h)
    return dig.hexdigest(), n


def test_parallel_header_walk_equals_serial_walk(tmp_path):
    """ssvh_bam_read_blocks finds the headers of a chunk's BGZF blocks with all host threads at once (every thread looks for the first block that begins in
    its segment and walks to the next segment; segments are taken over as far as they join) - on a file large enough for that path the block table is the one
    the one-block-after-the-other walk (SSV_SERIAL=walk, in a process of its own) makes"""
    import subprocess
    import sys
    from seeksv_amd import synth
    w = synth.Workload(genome_frac=1 / 2048, depth=30, n_sv=4)
    path = str(tmp_path / "walk.bam")
    old = os.environ.get("SSV_BGZF_LEVEL")
    os.environ["SSV_BGZF_LEVEL"] = "1"
    try:
        host.write_bam(path, w.names, w.lens, [w.generate_host(0, w.n_total, all_seq=True)])
    finally:
        if old is None:
            os.environ.pop("SSV_BGZF_LEVEL", None)
        else:
            os.environ["SSV_BGZF_LEVEL"] = old
    assert os.path.getsize(path) > (20 << 20)   # (the parallel walk starts at 16 MB of chunk)
    par = _block_table_digest(path)
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_bam_reader as t; print(*t._block_table_digest(%r))" % (
        os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SSV_SERIAL="walk"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ser = r.stdout.split()
    assert (ser[0], int(ser[1])) == par and par[1] > 1000


@pytest.mark.parametrize("level", ["-2", "-1"])
def test_writer_levels_without_zlib_give_the_same_records(tmp_path, level):
    """SSV_BGZF_LEVEL=-1 (a literal-only Huffman block per BGZF block) and -2 (the same with the cheapest string matching: huff_gz.h's deflate_fast, what
    bench.py writes its whole-genome file with on a 16-CPU box): zlib inflates every block to the bytes of the level-1 file, CRC32 and size right, and the
    reader gives the same batches (the level is read per call: a child process per file is not needed)"""
    import gzip
    import hashlib
    from seeksv_amd import synth
    w = synth.Workload(genome_frac=1 / 8192, depth=30, n_sv=4)
    batch = w.generate_host(0, w.n_total, all_seq=True)
    digests = {}
    old = os.environ.get("SSV_BGZF_LEVEL")
    try:
        for lv in ("1", level):
            os.environ["SSV_BGZF_LEVEL"] = lv
            p = str(tmp_path / f"l{lv}.bam")
            host.write_bam(p, w.names, w.lens, [batch])
            digests[lv] = (hashlib.sha256(gzip.open(p, "rb").read()).hexdigest(), os.path.getsize(p))   # (gzip checks every member's CRC32 and ISIZE)
    finally:
        if old is None:
            os.environ.pop("SSV_BGZF_LEVEL", None)
        else:
            os.environ["SSV_BGZF_LEVEL"] = old
    assert digests["1"][0] == digests[level][0]
    if level == "-2":
        assert digests[level][1] < 1.25 * digests["1"][1]   # string matching pays: close to zlib level 1's size
    names, lens, got = host.read_bam(str(tmp_path / f"l{level}.bam"))
    assert sum(len(b["tid"]) for b in got) == w.n_total


def test_generator_pairs_with_an_unmapped_end_reach_the_side_channel(tmp_path):
    """synth_core.h's unmap_permille: records 2k / 2k + 1 as (mapped read with MUNMAP, unmapped mate), both named after 2k by the writer; the reader hands them
    over raw (ssvh_bam_unmapped_raw, the form the device decoder uses) and decoded - the same records either way, in file order"""
    import ctypes as C
    from seeksv_amd import _abi, synth
    w = synth.Workload(genome_frac=1 / 8192, depth=30, n_sv=4, unmap_permille=25)
    b = w.generate_host(0, w.n_total, all_seq=True)
    un = np.nonzero(b["flag"] & 12)[0]
    assert len(un) > 100 and len(un) % 2 == 0 and (un[0::2] % 2 == 0).all() and (un[1::2] == un[0::2] + 1).all()
    assert ((b["flag"][un[0::2]] & 0x4c) == 0x48).all() and ((b["flag"][un[1::2]] & 0x8c) == 0x84).all()
    assert (np.diff(b["pos"].astype(np.int64) + (b["tid"].astype(np.int64) << 32)) >= 0).all()   # still coordinate sorted
    w0 = synth.Workload(genome_frac=1 / 8192, depth=30, n_sv=4)
    b0 = w0.generate_host(0, w0.n_total)
    rest = np.setdiff1d(np.arange(w.n_total), un)
    for k in ("tid", "pos", "flag", "mapq", "mpos", "isize"):
        assert np.array_equal(b[k][rest], b0[k][rest]), k   # the other records are the workload without them
    bam = str(tmp_path / "u.bam")
    host.write_bam(bam, w.names, w.lens, [b])
    lib = _abi.host_lib()
    lib.ssvh_bam_unmapped_raw.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.ssvh_raw_record_fastq.restype = C.c_size_t
    seen = []
    with host.BamReader(bam) as r:
        while r.read_batch(777) is not None:
            dec = r.unmapped()
            raw, nbytes = C.c_void_p(), C.c_size_t()
            assert lib.ssvh_bam_unmapped_raw(r.handle, C.byref(raw), C.byref(nbytes)) == 0
            off, k = 0, 0
            q, s, u, r1 = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_int()
            while nbytes.value:
                nxt = lib.ssvh_raw_record_fastq(raw, nbytes, C.c_size_t(off), C.byref(q), C.byref(s), C.byref(u), C.byref(r1))
                if not nxt:
                    break
                assert (q.value.decode(), s.value.decode(), u.value.decode(), bool(r1.value)) == dec[k]
                off, k = nxt, k + 1
            assert k == len(dec)
            seen += dec
    assert len(seen) == len(un)
    assert [x[0] for x in seen] == [f"s{g & ~1}" for g in un]
    assert all(a[3] and not c[3] for a, c in zip(seen[0::2], seen[1::2]))
