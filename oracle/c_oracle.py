"""ctypes loader for oracle/pileup_oracle.c.  TEST INFRASTRUCTURE ONLY (parity unpinned, see the .c header)."""

import ctypes as C
import os
import subprocess

import numpy as np

from midas_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpileup_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "pileup_oracle.c")
    hdr = os.path.join(HERE, "..", "include", "midas_snps.h")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["make", "-C", HERE, "-s", "-B", "libpileup_oracle.so"], check=True)
    return LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.midas_oracle_pileup.restype = C.c_int32
        _lib.midas_oracle_pileup.argtypes = [C.POINTER(abi.Thresholds), C.POINTER(abi._Contigs),
                                             C.POINTER(abi._Reads), C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_int64)]
    return _lib


def pileup(thr, contigs, reads, want_allele=True):
    """-> (status, err_read, counts[n_sites,4] u32, allele u8 | None, stats[n_species,4] i64)"""
    lib = _load()
    n = contigs.n_sites
    counts = np.zeros((n, 4), dtype=np.uint32)
    allele = np.zeros(n, dtype=np.uint8) if want_allele else None
    stats = np.zeros((contigs.n_species, abi.NUM_STATS), dtype=np.int64)
    err_read = C.c_int64(-1)
    c, r = contigs._c(), reads._c()
    st = lib.midas_oracle_pileup(C.byref(thr), C.byref(c), C.byref(r), counts.ctypes.data_as(C.c_void_p),
                                 allele.ctypes.data_as(C.c_void_p) if want_allele else None,
                                 stats.ctypes.data_as(C.c_void_p), C.byref(err_read))
    return int(st), int(err_read.value), counts, allele, stats
