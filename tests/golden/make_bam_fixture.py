"""Hand-assembles tests/golden/spec_fixture.bam byte by byte from the tables of the SAM/BAM specification (SAMv1 sections
4.1 "The BGZF compression format" and 4.2 "The BAM format") with nothing but struct and zlib -- it imports NOTHING from
midas_amd, so the decoder under test (midas_amd/csrc/hostio.cpp) and the writer the other tests use (midas_amd/bam.py) cannot
share a misreading of the spec.  The expected columns are written next to it (spec_fixture.json) from the same literals.

What the file exercises: records that straddle BGZF block borders (blocks are cut every 97 bytes of the stream), a
multi-reference header, every integer width of the NM aux tag (c C s S i I), a 'B' array, 'Z', 'H', 'A' and 'f' tags in
front of NM, a record without NM, a record without SEQ/QUAL, QUAL absent (0xFF), an unmapped record (refID -1: the path never
sees it), CIGARs with every op, and the 28-byte EOF block.

  python tests/golden/make_bam_fixture.py        (rewrites both files; they are committed)
"""
import json
import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
NT16 = "=ACMGRSVTWYHKDBN"          # SAMv1 4.2: 4-bit encoded read
CIGAR_OPS = "MIDNSHP=X"            # SAMv1 4.2: op_len << 4 | op

REFS = [("chrA", 1000), ("scaffold_2|x", 50000), ("c3", 77)]

# (refID, pos, mapq, flag, cigar, seq, qual (list | None = absent), aux as [(tag, type, value)])
RECORDS = [
    (0, 0, 42, 0, "10M", "ACGTACGTAC", [40] * 10, [("NM", "C", 0)]),
    (0, 5, 30, 16, "3S7M", "NNNACGTACG", list(range(30, 40)), [("XS", "A", "+"), ("NM", "c", 3)]),
    (0, 17, 255, 99, "4M2I4M", "ACGTTTACGT", [2, 3, 41, 41, 0, 93, 10, 20, 30, 40], [("ZB", "B", ("S", [1, 65535, 7])), ("NM", "S", 300)]),
    (0, 17, 0, 147, "4M2D4M1P1M", "ACGTACGTA", [35] * 9, [("MD", "Z", "4^AC5"), ("NM", "s", 2)]),
    (1, 100, 60, 0, "5H20M9000N5M2H", "A" * 25, [37] * 25, [("XH", "H", "1AE301"), ("NM", "i", 70000)]),
    (1, 49990, 20, 1024, "8=2X", "GGGGGGGGTT", [33] * 10, [("XF", "f", 1.5), ("NM", "I", 4000000000)]),
    (1, 49999, 3, 0, "1M", "T", [41], []),                                           # no NM at all
    (2, 0, 42, 0, "77M", "ACGT" * 19 + "A", None, [("NM", "C", 1)]),                  # QUAL absent: 0xFF bytes
    (2, 10, 42, 256, "*", "", [], [("NM", "C", 0)]),                                  # no SEQ, no CIGAR
    (-1, -1, 0, 4, "*", "ACGTN", [10] * 5, []),                                       # unmapped: dropped by the decoder
]


def reg2bin(beg, end):             # SAMv1 5.3, verbatim
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def parse_cigar(c):
    if c == "*":
        return []
    out, n = [], ""
    for ch in c:
        if ch.isdigit():
            n += ch
        else:
            out.append((int(n), CIGAR_OPS.index(ch)))
            n = ""
    return out


def aux_bytes(tag, typ, val):
    b = tag.encode() + typ.encode()
    if typ == "A": return b + val.encode()
    if typ in "cCsSiI": return b + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[typ], val)
    if typ == "f": return b + struct.pack("<f", val)
    if typ in "ZH": return b + val.encode() + b"\0"
    if typ == "B":
        sub, vals = val
        return b + sub.encode() + struct.pack("<i", len(vals)) + b"".join(
            struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub], v) for v in vals)
    raise ValueError(typ)


def record_bytes(k, rec):
    refid, pos, mapq, flag, cigar, seq, qual, aux = rec
    name = ("read%d" % k).encode() + b"\0"
    cg = parse_cigar(cigar)
    l_seq = len(seq)
    ref_len = sum(n for n, op in cg if op in (0, 2, 3, 7, 8))
    codes = [NT16.index(c) for c in seq] + ([0] if l_seq & 1 else [])
    seq4 = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    q = bytes([0xFF] * l_seq) if qual is None else bytes(qual)
    body = struct.pack("<iiBBHHHIiii", refid, pos, len(name), mapq, reg2bin(max(pos, 0), max(pos, 0) + max(ref_len, 1)), len(cg), flag,
                       l_seq, -1, -1, 0)
    body += name + b"".join(struct.pack("<I", (n << 4) | op) for n, op in cg) + seq4 + q
    body += b"".join(aux_bytes(*a) for a in aux)
    return struct.pack("<i", len(body)) + body


def bgzf_block(data):
    """SAMv1 4.1: a gzip member with the BC extra subfield holding the block's total size - 1."""
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = co.compress(data) + co.flush()
    total = 12 + 6 + len(cdata) + 8
    return (b"\x1f\x8b\x08\x04" + struct.pack("<IBBH", 0, 0, 0xFF, 6) + b"BC" + struct.pack("<HH", 2, total - 1) + cdata +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")   # SAMv1 4.1.2


def main():
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in REFS)
    stream = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(REFS))
    for name, ln in REFS:
        stream += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    for k, rec in enumerate(RECORDS):
        stream += record_bytes(k, rec)
    out = b"".join(bgzf_block(stream[lo:lo + 97]) for lo in range(0, len(stream), 97)) + EOF_BLOCK
    assert len(EOF_BLOCK) == 28 and zlib.decompress(EOF_BLOCK[18:-8], -15) == b""      # the spec's empty block inflates to nothing
    with open(os.path.join(HERE, "spec_fixture.bam"), "wb") as f:
        f.write(out)
    kept = [r for r in RECORDS if r[0] >= 0]
    exp = {"refs": [[n, l] for n, l in REFS], "records": []}
    for refid, pos, mapq, flag, cigar, seq, qual, aux in kept:
        nm = [v for t, ty, v in aux if t == "NM"]
        exp["records"].append({"refid": refid, "pos": pos, "mapq": mapq, "flag": flag,
                               "cigar": [(n << 4) | op for n, op in parse_cigar(cigar)], "seq": seq,
                               "qual": [0xFF] * len(seq) if qual is None else qual,
                               "nm": -1 if not nm else (nm[0] if nm[0] <= 0x7FFFFFFF else "overflow")})
    with open(os.path.join(HERE, "spec_fixture.json"), "w") as f:
        json.dump(exp, f, indent=1)
    print("wrote spec_fixture.bam (%d bytes, %d BGZF blocks) and spec_fixture.json" % (len(out), (len(stream) + 96) // 97 + 1))


if __name__ == "__main__":
    main()
