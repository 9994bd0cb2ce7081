"""Minimal BAM writer for test fixtures (BGZF via zlib).  Test tooling only."""
import struct
import zlib

NT16 = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def _bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    c = comp.compress(data) + comp.flush()
    bsize = len(c) + 25  # total block size - 1
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return hdr + c + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def parse_cigar(text):
    out, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num), CIGAR_OPS.index(ch)))
            num = ""
    return out


def encode_record(r):
    """r: dict(qname, flag, tid, pos(0-based), mapq, cigar (text or [(len,op)]), mtid, mpos, isize, seq, qual(None|bytes|str), aux(bytes))"""
    qname = r.get("qname", "r").encode() + b"\0"
    cig = r.get("cigar", "")
    if isinstance(cig, str):
        cig = parse_cigar(cig) if cig not in ("", "*") else []
    seq = r.get("seq", "")
    l_seq = len(seq)
    qual = r.get("qual")
    if qual is None:
        qual = b"\xff" * l_seq
    elif isinstance(qual, str):
        qual = bytes(ord(c) - 33 for c in qual)
    ref_span = sum(l for l, op in cig if op in (0, 2, 3, 7, 8))
    pos = r.get("pos", -1)
    bin_ = reg2bin(pos, pos + max(ref_span, 1)) if pos >= 0 else 4680
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq):
        v = NT16.index(ch.upper())
        packed[i >> 1] |= v << (4 if i % 2 == 0 else 0)
    body = struct.pack("<iiBBHHHiiii", r.get("tid", -1), pos, len(qname), r.get("mapq", 0), bin_, len(cig), r.get("flag", 0),
                       l_seq, r.get("mtid", -1), r.get("mpos", -1), r.get("isize", 0))
    body += qname + b"".join(struct.pack("<I", (l << 4) | op) for l, op in cig) + bytes(packed) + bytes(qual) + r.get("aux", b"")
    return struct.pack("<i", len(body)) + body


def write_bam(path, target_names, target_lens, records, sam_header_text=None):
    """records: iterable of dicts (see encode_record), already coordinate sorted if an index is wanted."""
    if sam_header_text is None:
        sam_header_text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(target_names, target_lens))
    text = sam_header_text.encode()
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(target_names))
    for n, l in zip(target_names, target_lens):
        nb = n.encode() + b"\0"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    with open(path, "wb") as f:
        f.write(_bgzf_block(hdr))
        buf = bytearray()
        for r in records:
            buf += encode_record(r) if isinstance(r, dict) else r
            while len(buf) >= 0xff00:
                f.write(_bgzf_block(bytes(buf[:0xff00])))
                del buf[:0xff00]
        if buf:
            f.write(_bgzf_block(bytes(buf)))
        f.write(BGZF_EOF)


def soa_to_bam(path, names, lens, b):
    import numpy as np
    NT = "=ACMGRSVTWYHKDBN"
    recs = []
    no_seq = np.uint64(2 ** 64 - 1)
    for i in range(len(b["tid"])):
        co, nc, lq = int(b["cigar_off"][i]), int(b["n_cigar"][i]), int(b["l_qseq"][i])
        cig = [(int(c) >> 4, int(c) & 15) for c in b["cigar"][co:co + nc]]
        if b["seq_off"][i] != no_seq:
            so = int(b["seq_off"][i])
            packed = b["seqqual"][so:so + (lq + 1) // 2]
            seq = "".join(NT[(int(packed[k >> 1]) >> (4 if k % 2 == 0 else 0)) & 15] for k in range(lq))
            qual = bytes(b["seqqual"][so + (lq + 1) // 2: so + (lq + 1) // 2 + lq])
        else:
            seq, qual = "A" * lq, b"\x1e" * lq
        recs.append(dict(qname=f"s{i}", flag=int(b["flag"][i]), tid=int(b["tid"][i]), pos=int(b["pos"][i]), mapq=int(b["mapq"][i]), cigar=cig,
                         mtid=int(b["mtid"][i]), mpos=int(b["mpos"][i]), isize=int(b["isize"][i]), seq=seq, qual=qual))
    write_bam(path, names, [int(x) for x in lens], recs)




def read_bam_records(path):
    """-> (target_names, [dict(qname, flag, tid, pos, mapq, cigar=[(len, op_char)], l_qseq)]) - plain Python, small files only"""
    import gzip
    import struct
    data = gzip.open(path, "rb").read()  # BGZF = concatenated gzip members
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, p)
    p += 4
    names = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, p)
        names.append(data[p + 4:p + 4 + l - 1].decode())
        p += 4 + l + 4
    recs = []
    while p < len(data):
        bs, tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiiBBHHHi", data, p)
        q = p + 36
        qname = data[q:q + l_rn - 1].decode()
        q += l_rn
        cig = [(c >> 4, "MIDNSHP=X"[c & 15]) for c in struct.unpack_from("<%dI" % n_cig, data, q)]
        recs.append(dict(qname=qname, flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cig, l_qseq=l_seq))
        p += 4 + bs
    return names, recs
