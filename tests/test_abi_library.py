"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
include/midas_snps.h declares; host-only entry points behave; there is no silent CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from midas_amd import abi, build, synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "midas_snps.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(midas_(?:snps|bam|merge)_[a-z_0-9]+)\s*\(", src)))


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

CALL = {"A": 0x0, "C": 0x4, "G": 0x8, "T": 0xC}


def call_codes(seq):
    """layout.h: 16 bytes per 32-base chunk; byte k = code(base k) | code(base k + 16) << 4; padding = 0x2"""
    code = {"A": 0x0, "C": 0x4, "G": 0x8, "T": 0xC}
    out = bytearray()
    for c0 in range(0, max(len(seq), 1), 32):
        chunk = [code.get(ch, 0x2) for ch in seq[c0:c0 + 32]]
        chunk += [0x2] * (32 - len(chunk))
        out += bytes(chunk[k] | (chunk[k + 16] << 4) for k in range(16))
    return bytes(out)


REC_DTYPE = np.dtype([("pos", "<i4"), ("off8", "<u4"), ("l", "<u2"), ("n", "<u2"), ("nm", "<u2"),
                      ("mapq", "u1"), ("flags", "u1")])


def test_pack_layout_matches_design():
    reads = H.reads_from_dicts([
        dict(pos=7, cigar="3S7M", seq="TTTACGTACG", qual=list(range(10)), nm=1, mapq=33, flag=16),
        dict(pos=9, cigar="5M", seq="ACGTN", nm=None, mapq=0),
        dict(pos=11, cigar="4M", seq="ACGT", qual="absent"),
    ])
    rec, blob, maxl = abi.pack_reads(reads)
    assert maxl == 10 and rec.shape == (3, 16)
    r = rec.view(REC_DTYPE).reshape(3)
    assert r["pos"].tolist() == [7, 9, 11] and (r["l"] & 0x7FF).tolist() == [10, 5, 4] and r["n"].tolist() == [2, 1, 1]
    assert r["nm"].tolist() == [1, 0xFFFF, 0] and r["mapq"].tolist() == [33, 0, 42]
    # flags: bit0 QUAL absent, bit1 "simple" (one M/=/X op spanning l_seq), bit2 generic clips, bit3 overrun
    assert (r["flags"] & 0x8F).tolist() == [0, 2, 3]
    # qmean = floor(sum(qual) / l): low five bits above l_seq, high three bits in flags 4-6
    qmean = ((r["l"] >> 11) | ((r["flags"].astype(int) >> 4) & 7) << 5).tolist()
    assert qmean == [45 // 10, 40, 255]           # 0..9 ; default quality 40 ; absent = 0xFF bytes
    # read 0: qual 10 -> 32 | calls 16 | cigar 8            = 56 bytes
    # read 1: qual  5 -> 32 | calls 16 | (simple: no cigar) = 48 bytes; read 2 likewise
    assert r["off8"].tolist() == [0, 7, 13]
    b0 = blob[:56]
    assert b0[:10].tolist() == list(range(10)) and not b0[10:32].any()
    assert bytes(b0[32:48]) == call_codes("TTTACGTACG")
    assert b0[48:56].view("<u4").tolist() == [(3 << 4) | 4, (7 << 4) | 0]
    b1 = blob[56:104]
    assert b1[:5].tolist() == [40, 40, 40, 40, 0]    # the N's quality is stored as 0: it can never count
    assert bytes(b1[32:48]) == call_codes("ACGTN")
    assert blob.size == 56 + 48 + 48
    # the sentinel record after the last read carries its own flag
    assert rec.shape[0] == 3


def test_pack_overrun_flag_needs_the_contig_length():
    reads = H.reads_from_dicts([dict(pos=0, cigar="12M", seq="ACGTACGTAC"),      # query 10..11 -> sites 10..11
                                dict(pos=15, cigar="12M", seq="ACGTACGTAC")])    # query 10..11 -> sites 25..26
    rec, _, _ = abi.pack_reads(reads, H.single_contig(20, 2))
    assert (rec[:, 15] & 8).tolist() == [8, 0]      # the second read overruns only beyond the contig end
    rec, _, _ = abi.pack_reads(reads, None)
    assert (rec[:, 15] & 8).tolist() == [8, 8]


def test_pack_cigar_fast_path_flags():
    """Decode-time CIGAR facts the kernels rely on (layout.h kRec*): they must be exact, not heuristics."""
    cases = [("150M", 150, 2), ("150=", 150, 2), ("150X", 150, 2), ("149M", 150, 0), ("10S140M", 150, 0),
             ("140M10S", 150, 0), ("10S130M10S", 150, 0), ("5H10S135M", 145, 4), ("135M10S5H", 145, 4),
             ("5S5S140M", 150, 4), ("140M5S5S", 150, 4), ("2H148M", 148, 4), ("10S", 10, 0), ("5S5S", 10, 4),
             ("75M2I73M", 150, 0), ("75M2D75M", 150, 0), ("150I", 150, 0)]
    reads = H.reads_from_dicts([dict(pos=0, cigar=cg, seq="A" * l) for cg, l, _ in cases])
    rec, _, _ = abi.pack_reads(reads)
    flags = (rec[:, 15] & 0x8F).tolist()       # bits 4-6 carry qmean
    assert flags == [f for _, _, f in cases], list(zip([c[0] for c in cases], flags))


def test_pack_rejects_malformed_input_with_status():
    ok = H.reads_from_dicts([dict(pos=0, cigar="4M", seq="ACGT")])
    bad = abi.ReadsSoA(**{**ok.as_dict(), 'qual_off': np.array([0, 2], dtype=np.int64)})
    with pytest.raises(abi.MidasSnpsError) as ei:
        abi.pack_reads(bad)
    assert ei.value.status == abi.ERR_BAD_LAYOUT
    long_read = H.reads_from_dicts([dict(pos=0, cigar="1025M", seq="A" * 1025)])
    with pytest.raises(abi.MidasSnpsError) as ei:
        abi.pack_reads(long_read)
    assert ei.value.status == abi.ERR_UNSUPPORTED


def test_pack_round_trips_synthetic_reads():
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=9000, n_reads=3000,
                                        seed=5, var_len=True)
    rec, blob, maxl = abi.pack_reads(reads, contigs)
    r = rec.view(REC_DTYPE).reshape(-1)
    np.testing.assert_array_equal(r["pos"], reads.pos)
    np.testing.assert_array_equal(r["l"] & 0x7FF, reads.l_seq)
    qmean = (r["l"] >> 11).astype(int) | (((r["flags"].astype(int) >> 4) & 7) << 5)
    saw_n = False
    for i in (0, 17, 1234, reads.n_reads - 1):
        l = int(reads.l_seq[i])
        o = int(r["off8"][i]) * 8
        nt16 = "=ACMGRSVTWYHKDBN"
        s4 = reads.seq4[reads.seq_off[i]:reads.seq_off[i] + (l + 1) // 2]
        seq = "".join(nt16[b >> 4] + nt16[b & 15] for b in s4)[:l]
        q = reads.qual[reads.qual_off[i]:reads.qual_off[i] + l].astype(int)
        assert qmean[i] == int(q.sum()) // l
        acgt = np.array([ch in "ACGT" for ch in seq])
        saw_n |= bool((~acgt).any())
        np.testing.assert_array_equal(blob[o:o + l], np.where(acgt, q, 0))   # non-ACGT bases carry quality 0
        assert not blob[o + l:o + ((l + 31) & ~31)].any()            # zero padding = self-masking read tail
        so = o + ((l + 31) & ~31)
        assert bytes(blob[so:so + 16 * ((l + 31) // 32)]) == call_codes(seq)
