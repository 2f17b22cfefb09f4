"""TEST INFRASTRUCTURE: the host mirror of the device packer (pack_mirror.cpp), built here and loaded with ctypes.

The product library does not contain it: the CPU tests pin the packed device layout (midas_amd/csrc/layout.h) on this
mirror, the GPU tests hold the device packer (pack_reads.hip) to it bit for bit.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from midas_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmidas_snps_mirror.so")
TILE_SITES = 2048          # the library's tile (midas_amd/csrc/kernels.h kTileSites)


def build(force=False):
    src = [os.path.join(HERE, "pack_mirror.cpp"), os.path.join(HERE, "pack_mirror.h"),
           os.path.join(HERE, "..", "..", "midas_amd", "csrc", "layout.h"), os.path.join(HERE, "..", "..", "midas_amd", "csrc", "workers.h"),
           os.path.join(HERE, "..", "..", "include", "midas_snps.h")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in src):
        tmp = LIB + ".tmp.%d" % os.getpid()
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__host__=", "-D__device__=", src[0], "-o", tmp, "-lpthread"],
                       check=True)
        os.replace(tmp, LIB)
    return LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIB)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        lib.midas_mirror_pack_reads.restype = i32
        lib.midas_mirror_pack_reads.argtypes = [C.POINTER(abi._Reads), C.POINTER(abi._Contigs), i32, vp, vp, i64, C.POINTER(i64),
                                                C.POINTER(i64), C.POINTER(i32), C.c_char_p]
        lib.midas_mirror_pack_reads_tiled.restype = i32
        lib.midas_mirror_pack_reads_tiled.argtypes = [C.POINTER(abi._Reads), C.POINTER(abi._Contigs), i32, i32, vp, vp, i64, vp, vp,
                                                      C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.c_char_p]
        _lib = lib
    return _lib


def pack_reads(reads, contigs=None, pad_rule=abi.PAD_SPEC):
    """Run the packer's mirror and return (rec[n_records,16] uint8, blob uint8, max_l_seq).  n_records >= n_reads:
    a read with indels or clips is served as one record per match segment (layout.h)."""
    lib = _load()
    r = reads._c()
    cc = contigs._c() if contigs is not None else None
    cp = C.byref(cc) if cc is not None else None
    nbytes, nrec, maxl = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    err = C.create_string_buffer(256)
    st = lib.midas_mirror_pack_reads(C.byref(r), cp, int(pad_rule), None, None, 0, C.byref(nbytes), C.byref(nrec), C.byref(maxl), err)
    if st != 0:
        raise abi.MidasSnpsError(st, err.value.decode())
    n = int(nrec.value)
    rec = np.zeros((n + 1, 16), dtype=np.uint8)   # + sentinel record
    blob = np.zeros(max(int(nbytes.value), 1), dtype=np.uint8)
    st = lib.midas_mirror_pack_reads(C.byref(r), cp, int(pad_rule), rec.ctypes.data_as(C.c_void_p), blob.ctypes.data_as(C.c_void_p),
                                     blob.size, C.byref(nbytes), C.byref(nrec), C.byref(maxl), err)
    if st != 0:
        raise abi.MidasSnpsError(st, err.value.decode())
    return rec[:n], blob[:int(nbytes.value)], int(maxl.value)


def pack_reads_tiled(reads, contigs, pad_rule=abi.PAD_SPEC):
    """The mirror in a batch's tile order -> (rec[n+1,16] u8 incl. sentinel, blob, orig u32, key u32)."""
    lib = _load()
    r, cc = reads._c(), contigs._c()
    nbytes, nrec, maxl = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    err = C.create_string_buffer(256)
    st = lib.midas_mirror_pack_reads_tiled(C.byref(r), C.byref(cc), int(pad_rule), TILE_SITES, None, None, 0, None, None, C.byref(nbytes),
                                           C.byref(nrec), C.byref(maxl), err)
    if st != 0:
        raise abi.MidasSnpsError(st, err.value.decode())
    n = int(nrec.value)
    rec = np.zeros((n + 1, 16), dtype=np.uint8)
    blob = np.zeros(max(int(nbytes.value), 1), dtype=np.uint8)
    orig = np.zeros(max(n, 1), np.uint32)
    key = np.zeros(max(n, 1), np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    st = lib.midas_mirror_pack_reads_tiled(C.byref(r), C.byref(cc), int(pad_rule), TILE_SITES, p(rec), p(blob), blob.size, p(orig), p(key),
                                           C.byref(nbytes), C.byref(nrec), C.byref(maxl), err)
    if st != 0:
        raise abi.MidasSnpsError(st, err.value.decode())
    return rec, blob[:int(nbytes.value)], orig[:n], key[:n]
