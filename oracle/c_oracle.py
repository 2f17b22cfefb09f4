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


def set_pad_rule(pysam_rule: bool):
    """What the CIGAR op P does to the query position in the walk: False = nothing (SAM specification, the default), True =
    it advances (pysam's get_aligned_pairs of MIDAS's time).  include/midas_snps.h, midas_snps_set_pad_rule."""
    lib = _load()
    lib.midas_oracle_set_pad_rule.restype = None
    lib.midas_oracle_set_pad_rule.argtypes = [C.c_int]
    lib.midas_oracle_set_pad_rule(1 if pysam_rule else 0)


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


def _contig_slice(contigs, reads, c0, c1):
    """(sub-ContigTable, sub-ReadsSoA) of contigs [c0, c1): views, no payload copies."""
    rb = contigs.read_begin
    lo, hi = int(rb[c0]), int(rb[c1])
    so, qo, co = reads.seq_off, reads.qual_off, reads.cigar_off
    sub = abi.ReadsSoA(pos=reads.pos[lo:hi], mapq=reads.mapq[lo:hi], flag=reads.flag[lo:hi], nm=reads.nm[lo:hi],
                       l_seq=reads.l_seq[lo:hi], seq_off=so[lo:hi + 1] - so[lo], qual_off=qo[lo:hi + 1] - qo[lo],
                       cigar_off=co[lo:hi + 1] - co[lo], seq4=reads.seq4[int(so[lo]):int(so[hi])],
                       qual=reads.qual[int(qo[lo]):int(qo[hi])], cigar=reads.cigar[int(co[lo]):int(co[hi])])
    off = contigs.site_offsets()
    sp = contigs.species[c0:c1]
    uniq = sorted(set(int(x) for x in sp))
    remap = {s: k for k, s in enumerate(uniq)}
    table = abi.ContigTable(length=contigs.length[c0:c1], species=[remap[int(x)] for x in sp], read_begin=rb[c0:c1 + 1] - lo,
                            ref=contigs.ref[int(off[c0]):int(off[c1])], n_species=len(uniq))
    return table, sub, uniq, int(off[c0]), int(off[c1])


def pileup_parallel(thr, contigs, reads, workers, grain="contig"):
    """The same oracle over a thread pool (ctypes releases the GIL): one task per contig, or per species -- the
    reference's own grain (midas/run/snps.py:225-228, one Pool task per species).  -> (status, counts, stats)"""
    from concurrent.futures import ThreadPoolExecutor
    _load()
    nc = contigs.n_contigs
    if grain == "species":     # runs of contigs of one species (contig tables keep a species' contigs together)
        cuts = [0] + [k for k in range(1, nc) if contigs.species[k] != contigs.species[k - 1]] + [nc]
    else:
        cuts = list(range(nc + 1))
    counts = np.zeros((contigs.n_sites, 4), dtype=np.uint32)
    stats = np.zeros((contigs.n_species, abi.NUM_STATS), dtype=np.int64)

    def task(k):
        table, sub, uniq, s0, s1 = _contig_slice(contigs, reads, cuts[k], cuts[k + 1])
        st, _, c, _, s = pileup(thr, table, sub, want_allele=False)
        counts[s0:s1] = c
        return st, uniq, s
    status = 0
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as ex:
        for st, uniq, s in ex.map(task, range(len(cuts) - 1)):
            status = status or st
            for k, sp in enumerate(uniq):
                stats[sp] += s[k]
    return status, counts, stats
