"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
include/midas_snps.h declares; host-only entry points behave; there is no silent CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from midas_amd import abi, build, synth
from tests import helpers as H
from tests import mirror

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "midas_snps.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(midas_(?:snps|bam|merge|genes|comm|fasta)_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_for_gfx950_and_loads():
    path = build.build_native()
    assert os.path.exists(path)
    lib = abi.load_library(build_if_missing=False)
    assert lib.midas_snps_abi_version() == abi.ABI_VERSION


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_symbols()
    assert len(declared) >= 18
    lib = C.CDLL(build.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), "header declares %s but the library does not export it" % sym
    assert sorted(abi.EXPORTED_SYMBOLS) == declared, "python binding and header disagree"


def test_library_contains_gfx950_code_object():
    blob = open(build.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"pileup_tiles_kernel" in blob and b"index_reads_kernel" in blob


def test_status_strings():
    lib = abi.load_library()
    assert lib.midas_snps_status_string(0) == b"ok"
    assert b"NM" in lib.midas_snps_status_string(abi.ERR_READ_NO_NM)
    assert lib.midas_snps_status_string(12345) == b"unknown status"


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this box has a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(abi.MidasSnpsError) as ei:
        abi.Context(0)
    assert ei.value.status == abi.ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "midas_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "pileup_oracle" not in text, f
    for f in ("scripts/run_midas.py",):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert "oracle" not in open(p).read()


# ---- the host packer (runs without a GPU) -----------------------------------------------------

LANE_BASES = 31      # layout.h: a lane carries 31 bases in 32 payload slots of one byte, the last slot is padding


def n_lanes(n_bases):
    return (n_bases + LANE_BASES - 1) // LANE_BASES


def base_bytes(seq, quals, lane_bases=LANE_BASES):
    """layout.h: one byte per base, (min(qual, 62) + 1) << 2 | code with A, C, G, T = 0..3; 0 for any other letter, for the
    padding slot of a 31-base lane and past the end of the record; 32 slots per lane."""
    out = []
    for c0 in range(0, len(seq), lane_bases):
        lane = [((min(q, 62) + 1) << 2) | "ACGT".index(ch) if ch in "ACGT" else 0
                for ch, q in zip(seq[c0:c0 + lane_bases], quals[c0:c0 + lane_bases])]
        out += lane + [0] * (32 - len(lane))
    return out


REC_DTYPE = np.dtype([("pos", "<i4"), ("off8", "<u4"), ("l", "<u2"), ("n", "<u2"), ("nm", "<u2"),
                      ("mapq", "u1"), ("flags", "u1")])


def seg_fields(r):
    """Read-level numbers a segment record carries (layout.h): (l_seq of the read, aligned length, NM, first?)."""
    n, nm = r["n"].astype(int), r["nm"].astype(int)
    return (n & 0x3FF).tolist(), ((((n >> 10) & 0xF) << 6) | (nm >> 10)).tolist(), (nm & 0x3FF).tolist(), ((n >> 14) & 1).tolist()


def qmean_of(r):
    return ((r["l"] >> 11).astype(int) | (((r["flags"].astype(int) >> 4) & 7) << 5)).tolist()


def test_pack_layout_matches_design():
    reads = H.reads_from_dicts([
        dict(pos=7, cigar="3S7M", seq="TTTACGTACG", qual=list(range(10)), nm=1, mapq=33, flag=16),
        dict(pos=9, cigar="5M", seq="ACGTN", nm=None, mapq=0),
        dict(pos=11, cigar="4M", seq="ACGT", qual="absent"),
    ])
    rec, blob, maxl = mirror.pack_reads(reads)
    assert maxl == 10 and rec.shape == (3, 16)
    r = rec.view(REC_DTYPE).reshape(3)
    # read 0 is served as its one match segment: the 7 aligned bases at pos 7 (the soft clip is in no record);
    # read 1 has no NM tag, so it keeps its CIGAR (the device's general path raises the reference's KeyError);
    # read 2 is a plain single match
    assert r["pos"].tolist() == [7, 9, 11] and (r["l"] & 0x7FF).tolist() == [7, 5, 4]
    # flags: bit0 QUAL absent, bit1 "simple" (one gap-free match segment), bit2 generic clips, bit3 overrun
    assert (r["flags"] & 0x8F).tolist() == [2, 0, 3]
    assert r["mapq"].tolist() == [33, 0, 42]
    lr, al, nm, first = seg_fields(r[[0, 2]])
    assert (lr, al, nm, first) == ([10, 4], [7, 4], [1, 0], [1, 1])
    assert r["n"][1] == 1 and r["nm"][1] == 0xFFFF
    # qmean = floor(sum(qual) / l) over the WHOLE read: low five bits above l_seq, high three bits in flags 4-6
    assert qmean_of(r) == [45 // 10, 40, 255]           # 0..9 ; default quality 40 ; absent = 0xFF bytes
    # read 0: 7 bases -> 32 bytes; read 1: 5 bases -> 32 | cigar 4 -> 8 = 40 bytes; read 2 like read 0
    assert r["off8"].tolist() == [0, 4, 9]
    b0 = blob[:32]
    assert b0.tolist() == base_bytes("ACGTACG", list(range(3, 10)))        # quality 3..9 above codes 0,1,2,3,0,1,2
    assert b0[:7].tolist() == [(q + 1) << 2 | k % 4 for k, q in enumerate(range(3, 10))] and not b0[7:32].any()
    b1 = blob[32:72]
    assert b1[:5].tolist() == [41 << 2 | 0, 41 << 2 | 1, 41 << 2 | 2, 41 << 2 | 3, 0]    # the N is stored as 0: it can never count
    assert b1[:32].tolist() == base_bytes("ACGTN", [40] * 5)
    assert b1[32:36].view("<u4").tolist() == [(5 << 4) | 0]
    assert blob.size == 32 + 40 + 32


def test_pack_serves_reads_as_match_segments():
    """CIGAR -> records: one per gap-free run of aligned bases; clipped and inserted bases are in no record."""
    seq = "".join("ACGT"[(7 * i) % 4] for i in range(150))
    reads = H.reads_from_dicts([
        dict(pos=100, cigar="75M2I73M", seq=seq, nm=3),          # -> (100, bases 0..74), (175, bases 77..149)
        dict(pos=300, cigar="5S70M2D75M", seq=seq, nm=2),        # -> (300, 5..74), (372, 75..149)
        dict(pos=500, cigar="10=1X20=119S", seq=seq, nm=1),      # -> one segment of 31
        dict(pos=700, cigar="2H10S30M5N40M3D60M10S4H", seq=seq, nm=3),   # three segments
    ])
    rec, blob, _ = mirror.pack_reads(reads)
    r = rec.view(REC_DTYPE).reshape(-1)
    assert r["pos"].tolist() == [100, 175, 300, 372, 500, 700, 735, 778]
    assert (r["l"] & 0x7FF).tolist() == [75, 73, 70, 75, 31, 30, 40, 60]
    assert (r["flags"] & 0x8F).tolist() == [2] * 8
    lr, al, nm, first = seg_fields(r)
    assert lr == [150] * 8 and nm == [3, 3, 2, 2, 1, 3, 3, 3]
    assert al == [150, 150, 145, 145, 31, 130, 130, 130]          # l_seq minus the soft clips (insertions count)
    assert first == [1, 0, 1, 0, 1, 1, 0, 0]
    # payload of the second record of read 0: read bases 77..149
    o = int(r["off8"][1]) * 8
    assert blob[o:o + 96].tolist() == base_bytes(seq[77:150], [40] * 73)              # 73 bases: three lanes


def test_pack_uses_all_32_slots_where_that_saves_a_lane():
    """layout.h lane_bases_for: 125 bp reads pack 32 bases per lane (4 lanes, no padding slot), 150 bp reads 31 (5 lanes)."""
    seq = "".join("ACGT"[(5 * i) % 4] for i in range(125))
    quals = [30 + (i % 10) for i in range(125)]
    reads = H.reads_from_dicts([dict(pos=10, cigar="125M", seq=seq, nm=0, qual=quals)])
    rec, blob, maxl = mirror.pack_reads(reads)
    assert maxl == 125 and blob.size == 128
    assert blob.tolist() == base_bytes(seq, quals, lane_bases=32)
    assert not blob[125:128].any()


def test_pack_keeps_qualities_in_six_bits():
    """Qualities above 62 are stored as 62 (exact for every baseq <= 62, layout.h); 0xFF bytes (QUAL absent) too."""
    reads = H.reads_from_dicts([dict(pos=0, cigar="6M", seq="ACGTAC", qual=[0, 1, 61, 62, 63, 93]),
                                dict(pos=0, cigar="4M", seq="ACGT", qual="absent")])
    rec, blob, _ = mirror.pack_reads(reads)
    assert blob[:6].tolist() == [1 << 2 | 0, 2 << 2 | 1, 62 << 2 | 2, 63 << 2 | 3, 63 << 2 | 0, 63 << 2 | 1]
    assert blob[32:36].tolist() == [63 << 2 | 0, 63 << 2 | 1, 63 << 2 | 2, 63 << 2 | 3]


def test_pack_keeps_the_cigar_of_reads_it_cannot_segment():
    seq = "A" * 60
    cases = [("10S", 10, "no aligned base"), ("5S5S50M", 60, "two clips at one end"), ("30M2P30M", 60, "pad op"),
             ("20M5S35M", 60, "clip in the middle"), ("50M", 60, "query length does not add up"),
             ("70M", 60, "CIGAR longer than SEQ"), ("60I", 60, "no match op")]
    reads = H.reads_from_dicts([dict(pos=0, cigar=cg, seq=seq[:l]) for cg, l, _ in cases] +
                               [dict(pos=0, cigar="60M", seq=seq, nm=None), dict(pos=0, cigar="60M", seq=seq, nm=1024),
                                dict(pos=-1, cigar="60M", seq=seq)])
    rec, _, _ = mirror.pack_reads(reads)
    r = rec.view(REC_DTYPE).reshape(-1)
    assert rec.shape[0] == len(cases) + 3
    assert ((r["flags"] & 2) == 0).all()                        # none of them is "simple"
    assert r["n"].tolist() == [1, 3, 3, 3, 1, 1, 1, 1, 1, 1]    # the op counts, i.e. the CIGARs are stored
    # seven match ops separated by insertions: more segments than a read may be served as
    many = H.reads_from_dicts([dict(pos=0, cigar="5M1I" * 6 + "5M", seq="A" * 41, nm=6)])
    rec, _, _ = mirror.pack_reads(many)
    assert rec.shape[0] == 1 and not (rec[0, 15] & 2)
    six = H.reads_from_dicts([dict(pos=0, cigar="5M1I" * 5 + "5M", seq="A" * 35, nm=5)])
    rec, _, _ = mirror.pack_reads(six)
    assert rec.shape[0] == 6 and all(rec[:, 15] & 2)


def test_pack_overrun_flag_needs_the_contig_length():
    reads = H.reads_from_dicts([dict(pos=0, cigar="12M", seq="ACGTACGTAC"),      # query 10..11 -> sites 10..11
                                dict(pos=15, cigar="12M", seq="ACGTACGTAC")])    # query 10..11 -> sites 25..26
    rec, _, _ = mirror.pack_reads(reads, H.single_contig(20, 2))
    assert (rec[:, 15] & 8).tolist() == [8, 0]      # the second read overruns only beyond the contig end
    rec, _, _ = mirror.pack_reads(reads, None)
    assert (rec[:, 15] & 8).tolist() == [8, 8]


def test_pack_cigar_fast_path_flags():
    """Decode-time CIGAR facts the kernels rely on (layout.h kRec*): they must be exact, not heuristics.
    (cigar, l_seq, flags of the first record & 0x0F, number of records)"""
    cases = [("150M", 150, 2, 1), ("150=", 150, 2, 1), ("150X", 150, 2, 1), ("149M", 150, 0, 1), ("10S140M", 150, 2, 1),
             ("140M10S", 150, 2, 1), ("10S130M10S", 150, 2, 1), ("5H10S135M", 145, 2, 1), ("135M10S5H", 145, 2, 1),
             ("5S5S140M", 150, 4, 1), ("140M5S5S", 150, 4, 1), ("2H148M", 148, 2, 1), ("10S", 10, 0, 1), ("5S5S", 10, 4, 1),
             ("75M2I73M", 150, 2, 2), ("75M2D75M", 150, 2, 2), ("150I", 150, 0, 1)]
    for cg, l, f, k in cases:
        rec, _, _ = mirror.pack_reads(H.reads_from_dicts([dict(pos=0, cigar=cg, seq="A" * l)]))
        assert rec.shape[0] == k and (rec[0, 15] & 0x0F) == f, (cg, rec.shape[0], rec[0, 15] & 0x0F)


def test_pack_rejects_malformed_input_with_status():
    ok = H.reads_from_dicts([dict(pos=0, cigar="4M", seq="ACGT")])
    bad = abi.ReadsSoA(**{**ok.as_dict(), 'qual_off': np.array([0, 2], dtype=np.int64)})
    with pytest.raises(abi.MidasSnpsError) as ei:
        mirror.pack_reads(bad)
    assert ei.value.status == abi.ERR_BAD_LAYOUT
    long_read = H.reads_from_dicts([dict(pos=0, cigar="1025M", seq="A" * 1025)])
    with pytest.raises(abi.MidasSnpsError) as ei:
        mirror.pack_reads(long_read)
    assert ei.value.status == abi.ERR_UNSUPPORTED


def test_pack_round_trips_synthetic_reads():
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=9000, n_reads=3000,
                                        seed=5, var_len=True)
    rec, blob, maxl = mirror.pack_reads(reads, contigs)
    r = rec.view(REC_DTYPE).reshape(-1)
    nt16 = "=ACMGRSVTWYHKDBN"
    # walk the records in input order (no tile order without a tile length): read i owns k_i consecutive records
    j = 0
    multi = 0
    for i in range(reads.n_reads):
        l = int(reads.l_seq[i])
        s4 = reads.seq4[reads.seq_off[i]:reads.seq_off[i] + (l + 1) // 2]
        seq = "".join(nt16[b >> 4] + nt16[b & 15] for b in s4)[:l]
        q = reads.qual[reads.qual_off[i]:reads.qual_off[i] + l].astype(int)
        cig = [(int(v) & 15, int(v) >> 4) for v in reads.cigar[reads.cigar_off[i]:reads.cigar_off[i + 1]]]
        # match segments by the textbook walk
        segs, qp, rp = [], 0, int(reads.pos[i])
        for op, ln in cig:
            if op in (0, 7, 8):
                segs.append((qp, rp, ln)); qp += ln; rp += ln
            elif op in (1, 4):
                qp += ln
            elif op in (2, 3):
                rp += ln
        multi += len(segs) > 1
        for s, (qs, rs, ln) in enumerate(segs):
            assert r["flags"][j] & 2 and int(r["pos"][j]) == rs and int(r["l"][j]) & 0x7FF == ln, (i, s)
            assert qmean_of(r[j:j + 1])[0] == int(q.sum()) // l
            o = int(r["off8"][j]) * 8
            # non-ACGT bases carry quality 0; so do the padding slots (the self-masking tail)
            assert blob[o:o + 32 * n_lanes(ln)].tolist() == base_bytes(seq[qs:qs + ln], q[qs:qs + ln].tolist())
            lr, al, nm, first = seg_fields(r[j:j + 1])
            clips = sum(ln2 for op, ln2 in cig if op == 4)
            assert (lr[0], al[0], nm[0], first[0]) == (l, l - clips, int(reads.nm[i]), int(s == 0))
            j += 1
    assert j == rec.shape[0] and multi > 50
