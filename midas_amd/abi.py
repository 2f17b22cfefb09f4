"""ctypes binding of include/midas_snps.h (libmidas_snps_hip.so).

There is no CPU fallback: if the library is missing, or no gfx950 GPU is visible,
the calls raise.  Nothing under ``oracle/`` is imported from here.
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import build as _build

ABI_VERSION = 4

STAT_ALIGNED_READS, STAT_MAPPED_READS, STAT_COVERED_BASES, STAT_TOTAL_DEPTH = range(4)
NUM_STATS = 4

ERR_READ_NO_SEQ, ERR_READ_NO_NM, ERR_READ_ZERO_ALIGN, ERR_READ_NO_QUAL, ERR_READ_CIGAR_OVERRUN, \
    ERR_READ_BAD_CIGAR_OP = range(1, 7)
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY, ERR_UNSUPPORTED, ERR_BAD_LAYOUT = \
    -1, -2, -3, -4, -5, -6


class MidasSnpsError(RuntimeError):
    def __init__(self, status: int, message: str, read_index: int = -1):
        super().__init__("midas_snps status %d: %s" % (status, message))
        self.status = status
        self.message = message
        self.read_index = read_index


class Thresholds(C.Structure):
    """scripts/run_midas.py:410-419 defaults."""
    _fields_ = [("baseq", C.c_int32), ("mapq", C.c_int32), ("readq", C.c_int32), ("reserved", C.c_int32),
                ("mapid", C.c_double), ("aln_cov", C.c_double)]

    @classmethod
    def from_args(cls, args: dict) -> "Thresholds":
        return cls(int(args['baseq']), int(args['mapq']), int(args['readq']), 0,
                   float(args['mapid']), float(args['aln_cov']))


DEFAULT_ARGS = {'mapid': 94.0, 'mapq': 20, 'baseq': 30, 'readq': 20, 'aln_cov': 0.75}

ERR_MERGE_ZERO_MEAN_DEPTH = 7
SNP_TYPE_BITS = {'any': 1, 'mono': 2, 'bi': 4, 'tri': 8, 'quad': 16}
SNP_TYPE_NAMES = [None, 'mono', 'bi', 'tri', 'quad']


class MergeParams(C.Structure):
    """scripts/merge_midas.py:229-252 (site filters) as the kernel takes them."""
    _fields_ = [("allele_freq", C.c_double), ("site_ratio", C.c_double), ("site_prev", C.c_double),
                ("site_depth", C.c_int32), ("snp_types", C.c_int32)]

    @classmethod
    def from_args(cls, args: dict) -> "MergeParams":
        bits = 0
        for t in args['snp_type']:
            bits |= SNP_TYPE_BITS[t]
        return cls(float(args['allele_freq']), float(args['site_ratio']), float(args['site_prev']),
                   int(args['site_depth']), bits)


DEFAULT_MERGE_ARGS = {'snp_type': ['bi'], 'allele_freq': 0.01, 'site_depth': 1, 'site_ratio': 2.0, 'site_prev': 0.95}


class _Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("pos", C.c_void_p), ("mapq", C.c_void_p), ("flag", C.c_void_p),
                ("nm", C.c_void_p), ("l_seq", C.c_void_p), ("seq_off", C.c_void_p), ("qual_off", C.c_void_p),
                ("cigar_off", C.c_void_p), ("seq4", C.c_void_p), ("qual", C.c_void_p), ("cigar", C.c_void_p)]


class _Contigs(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("n_species", C.c_int32), ("length", C.c_void_p),
                ("species", C.c_void_p), ("read_begin", C.c_void_p), ("ref", C.c_void_p), ("origin", C.c_void_p)]


class BatchInfo(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_sites", C.c_int64), ("n_tiles", C.c_int64),
                ("packed_bytes", C.c_int64), ("algorithmic_bytes", C.c_int64),
                ("tile_sites", C.c_int32), ("lanes_per_read", C.c_int32), ("n_work_items", C.c_int64),
                ("path", C.c_int32), ("path_auto", C.c_int32), ("lane_bases", C.c_int32), ("layout_build_us", C.c_int32),
                ("direct_general_reads", C.c_int64), ("direct_reach", C.c_int64),
                ("direct_stream_reads", C.c_int64), ("direct_max_tile_reads", C.c_int64),
                ("direct_chunk_tiles", C.c_int32), ("direct_overhang", C.c_int32)]


ROWS_DEVICE, ROWS_HOST = 0, 1    # who formats and deflates a batch's rows (midas_snps_set_row_coder)
PAD_SPEC, PAD_PYSAM = 0, 1       # what the CIGAR op P does to the query position (midas_snps_set_pad_rule)
PATH_AUTO, PATH_DIRECT, PATH_PACKED, PATH_LONG = 0, 1, 2, 3
PATH_NAMES = {PATH_AUTO: "auto", PATH_DIRECT: "direct", PATH_PACKED: "packed", PATH_LONG: "long"}


_SOA_DTYPES = {
    'pos': np.int32, 'mapq': np.uint8, 'flag': np.uint16, 'nm': np.int32, 'l_seq': np.int32,
    'seq_off': np.int64, 'qual_off': np.int64, 'cigar_off': np.int64,
    'seq4': np.uint8, 'qual': np.uint8, 'cigar': np.uint32,
}


@dataclass
class ReadsSoA:
    """Alignment records in BAM-native encodings (see midas_snps_reads in the header)."""
    pos: np.ndarray
    mapq: np.ndarray
    flag: np.ndarray
    nm: np.ndarray
    l_seq: np.ndarray
    seq_off: np.ndarray
    qual_off: np.ndarray
    cigar_off: np.ndarray
    seq4: np.ndarray
    qual: np.ndarray
    cigar: np.ndarray
    # (seq4, qual, cigar) as DEVICE addresses + whatever keeps them alive (read_bam(..., payload_on_device=True)): the three
    # arrays above are empty then, and only a Context's batch / pileup -- which copy device to device -- can take the reads
    device: Optional[tuple] = None

    def __post_init__(self):
        for k, dt in _SOA_DTYPES.items():
            setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=dt))

    @property
    def n_reads(self) -> int:
        return int(self.pos.shape[0])

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k in _SOA_DTYPES}

    @classmethod
    def from_dict(cls, d: dict) -> "ReadsSoA":
        return cls(**{k: d[k] for k in _SOA_DTYPES})

    @classmethod
    def empty(cls) -> "ReadsSoA":
        z = lambda dt, n=0: np.zeros(n, dtype=dt)
        return cls(z(np.int32), z(np.uint8), z(np.uint16), z(np.int32), z(np.int32),
                   z(np.int64, 1), z(np.int64, 1), z(np.int64, 1), z(np.uint8), z(np.uint8), z(np.uint32))

    def _c(self) -> _Reads:
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # zero-size arrays still have a valid address
        payload = (p(self.seq4), p(self.qual), p(self.cigar)) if self.device is None else \
            tuple(C.c_void_p(int(x)) for x in self.device[:3])
        return _Reads(self.n_reads, p(self.pos), p(self.mapq), p(self.flag), p(self.nm), p(self.l_seq),
                      self.seq_off.ctypes.data_as(C.c_void_p), self.qual_off.ctypes.data_as(C.c_void_p),
                      self.cigar_off.ctypes.data_as(C.c_void_p), *payload)


class ResidentReads:
    """The records of a BAM decoded on a Context's device with EVERY column left there, in the pileup kernel's own layout
    (read_bam(..., resident=True), midas_bam_load_resident): what the host holds is the count, the bases' total and the native
    handle.  Only that context's Batch takes them -- records [first, first + n) of the handle, where they lie; a host that must
    look into the columns asks Context.fetch_payload, which turns them into an ordinary ReadsSoA (midas_bam_resident_to_columns)."""
    device = ('resident',)        # (read like ReadsSoA.device: "not in host memory")

    def __init__(self, lib, handle, owner, n_reads: int, l_seq_total: int, first: int = 0):
        self._lib, self._h, self.owner = lib, handle, owner
        self._n, self.l_seq_total, self.first = int(n_reads), int(l_seq_total), int(first)

    @property
    def n_reads(self) -> int:
        return self._n

    def to_columns(self, ctx) -> "ReadsSoA":
        """The same records as a ReadsSoA whose SEQ / QUAL / CIGAR are device columns (as read_bam(payload_on_device=True)'s)."""
        sb, qb, nc = C.c_int64(), C.c_int64(), C.c_int64()
        err = C.create_string_buffer(256)
        st = self._lib.midas_bam_resident_to_columns(self._h, ctx._h, C.byref(sb), C.byref(qb), C.byref(nc), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        _, reads = _bam_columns(self._lib, self._h, self._n, int(sb.value), int(qb.value), int(nc.value), self.owner, on_device=True)
        return reads


@dataclass
class ContigTable:
    """Flattened `contigs` dict of midas/run/snps.py:55-67 (see midas_snps_contigs)."""
    length: np.ndarray        # [n_contigs] int64
    species: np.ndarray       # [n_contigs] int32 index into species ids
    read_begin: np.ndarray    # [n_contigs+1] int64
    ref: np.ndarray           # [sum(length)] uint8
    n_species: int
    ids: list = field(default_factory=list)           # contig ids, table order
    species_ids: list = field(default_factory=list)   # species ids, index order
    origin: Optional[np.ndarray] = None               # [n_contigs] int64 or None: the entries are pieces of longer contigs

    def __post_init__(self):
        if self.origin is not None:
            self.origin = np.ascontiguousarray(self.origin, dtype=np.int64)
        self.length = np.ascontiguousarray(self.length, dtype=np.int64)
        self.species = np.ascontiguousarray(self.species, dtype=np.int32)
        self.read_begin = np.ascontiguousarray(self.read_begin, dtype=np.int64)
        self.ref = np.ascontiguousarray(self.ref, dtype=np.uint8)

    @property
    def n_contigs(self) -> int:
        return int(self.length.shape[0])

    @property
    def n_sites(self) -> int:
        return int(self.length.sum())

    def site_offsets(self) -> np.ndarray:
        out = np.zeros(self.n_contigs + 1, dtype=np.int64)
        np.cumsum(self.length, out=out[1:])
        return out

    def _c(self) -> _Contigs:
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        return _Contigs(self.n_contigs, int(self.n_species), p(self.length), p(self.species),
                        self.read_begin.ctypes.data_as(C.c_void_p), p(self.ref), p(self.origin) if self.origin is not None else None)


_lib = None


def library_path() -> str:
    return _build.LIB_PATH


def load_library(build_if_missing: bool = True):
    """dlopen the in-tree libmidas_snps_hip.so; raises if it cannot be found or built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MIDAS_SNPS_LIBRARY") or _build.LIB_PATH     # override: developer builds (tools/)
    if not os.path.exists(path):
        if not build_if_missing:
            raise MidasSnpsError(ERR_NO_DEVICE, "native library %s is missing (run `python -m midas_amd.build`)" % path)
        _build.build_native()
    lib = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    sig = {
        'midas_snps_abi_version': (i32, []),
        'midas_snps_cpu_budget': (i32, []),
        'midas_snps_status_string': (C.c_char_p, [i32]),
        'midas_snps_create': (i32, [i32, C.POINTER(vp)]),
        'midas_snps_destroy': (None, [vp]),
        'midas_snps_last_error': (C.c_char_p, [vp]),
        'midas_snps_last_error_read': (i64, [vp]),
        'midas_snps_set_stream': (i32, [vp, vp]),
        'midas_snps_device_info': (i32, [vp, C.c_char_p, C.POINTER(i32), C.POINTER(i64)]),
        'midas_snps_host_alloc': (vp, [i64]),
        'midas_snps_host_free': (None, [vp]),
        'midas_snps_pileup': (i32, [vp, C.POINTER(Thresholds), C.POINTER(_Contigs), C.POINTER(_Reads), vp, vp, vp]),
        'midas_snps_batch_create': (i32, [vp, C.POINTER(_Contigs), C.POINTER(_Reads), C.POINTER(vp)]),
        'midas_snps_batch_destroy': (None, [vp]),
        'midas_snps_batch_run': (i32, [vp, C.POINTER(Thresholds)]),
        'midas_snps_batch_sync': (i32, [vp]),
        'midas_snps_batch_fetch': (i32, [vp, vp, vp, vp]),
        'midas_snps_batch_get_info': (i32, [vp, C.POINTER(BatchInfo)]),
        'midas_snps_batch_enable_timing': (i32, [vp, i32]),
        'midas_snps_batch_timing': (i32, [vp, i32, C.POINTER(C.c_float)]),
        'midas_snps_batch_time_pileup_only': (i32, [vp, i32]),
        'midas_snps_batch_stats_to_device': (i32, [vp, vp]),
        'midas_snps_batch_pack': (i32, [vp]),
        'midas_snps_batch_select_path': (i32, [vp, i32]),
        'midas_snps_set_default_path': (i32, [vp, i32]),
        'midas_snps_set_pad_rule': (i32, [vp, i32]),
        'midas_snps_stream_rates': (i32, [vp, i64, i32, C.POINTER(C.c_double)]),
        'midas_snps_calibration_pass': (i32, [vp, i64]),
        'midas_snps_set_row_coder': (i32, [vp, i32]),
        'midas_snps_copy_rate': (i32, [vp, i64, i32, C.POINTER(C.c_double)]),
        'midas_snps_batch_fetch_packed': (i32, [vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]),
        'midas_snps_batch_pack_timing': (i32, [vp, i32, C.POINTER(C.c_float)]),
    }
    sig.update({
        'midas_bam_open': (i32, [C.c_char_p, C.POINTER(vp), C.c_char_p]),
        'midas_bam_close': (None, [vp]),
        'midas_bam_write': (i32, [C.c_char_p, i32, C.POINTER(C.c_char_p), vp, C.POINTER(_Reads), vp, i32, i32, C.c_char_p]),
        'midas_bam_n_refs': (i32, [vp]),
        'midas_bam_ref': (i32, [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i64)]),
        'midas_bam_load': (i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.c_char_p]),
        'midas_bam_copy': (i32, [vp] + [vp] * 12),
        'midas_bam_columns': (i32, [vp, vp]),
        'midas_bam_open_slice': (i32, [C.c_char_p, i32, i32, C.POINTER(vp), C.c_char_p]),
        'midas_bam_slice_facts': (i32, [vp, vp, vp, vp, vp]),
        'midas_bam_slice_marks': (i32, [vp, vp, vp, vp, C.c_int64]),
        'midas_bam_open_device': (i32, [C.c_char_p, vp, C.POINTER(vp), C.c_char_p]),
        'midas_bam_open_slice_device': (i32, [C.c_char_p, i32, i32, vp, C.POINTER(vp), C.c_char_p]),
        'midas_bam_load_device': (i32, [C.c_char_p, vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.c_char_p]),
        'midas_bam_payload_on_device': (i32, [vp]),
        'midas_bam_release_file': (None, [vp]),
        'midas_bam_load_resident': (i32, [C.c_char_p, vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.c_char_p]),
        'midas_bam_load_ranges_resident': (i32, [vp, vp, i32, vp, vp, C.POINTER(i64), C.POINTER(i64), C.c_char_p]),
        'midas_bam_is_resident': (i32, [vp]),
        'midas_bam_open_share_local': (i32, [C.c_char_p, i32, i32, C.POINTER(vp), vp, C.c_char_p]),
        'midas_bam_share_locate': (i32, [vp, i32, i64, i64, i64, vp, C.c_char_p]),
        'midas_bam_resident_to_columns': (i32, [vp, vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.c_char_p]),
        'midas_snps_batch_create_resident': (i32, [vp, C.POINTER(_Contigs), vp, i64, C.POINTER(vp)]),
        'midas_snps_copy_from_device': (i32, [vp, vp, vp, i64]),
        'midas_bam_load_ranges_device': (i32, [vp, vp, i32, vp, vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.c_char_p]),
        'midas_snps_inflate_blocks': (i32, [vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, i64, C.POINTER(i64)]),
        'midas_bam_load_ranges': (i32, [vp, i32, vp, vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.c_char_p]),
        'midas_snps_write_rows': (i32, [C.c_char_p, i32, C.c_char_p, i64, vp, vp, i32, i32, C.c_char_p]),
        'midas_merge_write_info': (i32, [C.c_char_p, C.c_char_p, i64, vp, vp, vp, vp, vp, vp, vp, i32, i64, C.c_char_p]),
        'midas_merge_write_matrix': (i32, [C.c_char_p, C.c_char_p, i64, vp, i32, i64, vp, vp, i32, i64, C.c_char_p]),
        'midas_snps_table_open_range': (i32, [C.c_char_p, i64, i64, i32, C.POINTER(vp), C.c_char_p]),
        'midas_snps_table_count_rows': (i32, [C.c_char_p, C.POINTER(i64), C.c_char_p]),
        'midas_snps_write_table': (i32, [C.c_char_p, i32, vp, vp, vp, vp, i32, i32, C.c_char_p]),
        'midas_snps_write_part': (i32, [C.c_char_p, i32, i32, vp, vp, vp, vp, i32, i32, C.c_char_p]),
        'midas_snps_write_pieces': (i32, [C.c_char_p, i32, i32, vp, vp, vp, vp, vp, i32, i32, C.c_char_p]),
        'midas_snps_deflate_rows': (i32, [vp, i64, vp, vp, i64, vp, i64, C.POINTER(i64)]),
        'midas_snps_batch_write_part': (i32, [vp, C.c_char_p, i32, i32, vp, vp, i32, i32]),
        'midas_fasta_load': (i32, [i32, vp, i32, C.POINTER(vp), C.c_char_p]),
        'midas_fasta_n_records': (i64, [vp]),
        'midas_fasta_columns': (i32, [vp, vp, vp]),
        'midas_fasta_close': (None, [vp]),
        'midas_snps_tableset_open': (i32, [i32, vp, C.POINTER(vp), vp, C.c_char_p]),
        'midas_snps_tableset_read_counts': (i32, [vp, i64, i64, vp, C.c_char_p]),
        'midas_snps_tableset_close': (None, [vp]),
        'midas_snps_table_open': (i32, [C.c_char_p, i64, i32, C.POINTER(vp), C.c_char_p]),
        'midas_snps_table_close': (None, [vp]),
        'midas_snps_table_rows': (i64, [vp]),
        'midas_snps_table_key_bytes': (i64, [vp]),
        'midas_snps_table_copy': (i32, [vp, vp, vp, vp]),
        'midas_genes_count': (i32, [vp, C.POINTER(Thresholds), C.POINTER(_Reads), vp, i64, vp, vp, vp, vp, C.POINTER(C.c_float)]),
        'midas_genes_terms': (i32, [vp, C.POINTER(Thresholds), C.POINTER(_Reads), vp, i64, vp, vp, C.POINTER(C.c_float)]),
        'midas_genes_sum': (i32, [vp, i64, vp, vp, i64, vp, vp, vp, C.POINTER(C.c_float)]),
        'midas_merge_sites': (i32, [vp, C.POINTER(MergeParams), i32, i64, C.POINTER(vp), vp] + [vp] * 5 + [C.POINTER(C.c_float)]),
        'midas_bam_open_share': (i32, [C.c_char_p, i32, i32, i64, C.POINTER(vp), vp, C.c_char_p]),
        'midas_comm_device_key': (i32, [vp, C.c_char_p]),
        'midas_comm_unique_id': (i32, [vp, C.c_char_p]),
        'midas_comm_probe': (i32, [C.POINTER(i32), C.c_char_p]),
        'midas_comm_create': (i32, [vp, vp, i32, i32, C.POINTER(vp), C.c_char_p]),
        'midas_comm_destroy': (None, [vp]),
        'midas_comm_all_gather': (i32, [vp, vp, vp, i64, C.c_char_p]),
        'midas_comm_all_to_all_v': (i32, [vp, vp, vp, vp, vp, C.c_char_p]),
    })
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.midas_snps_abi_version() != ABI_VERSION:
        raise MidasSnpsError(ERR_INVALID_ARG, "ABI version mismatch: library %d, binding %d"
                             % (lib.midas_snps_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    'midas_snps_abi_version', 'midas_snps_cpu_budget', 'midas_snps_status_string', 'midas_snps_create', 'midas_snps_destroy',
    'midas_snps_last_error', 'midas_snps_last_error_read', 'midas_snps_set_stream', 'midas_snps_device_info',
    'midas_snps_host_alloc', 'midas_snps_host_free',
    'midas_snps_pileup', 'midas_snps_batch_create', 'midas_snps_batch_destroy', 'midas_snps_batch_run',
    'midas_snps_batch_sync', 'midas_snps_batch_fetch', 'midas_snps_batch_get_info',
    'midas_snps_batch_enable_timing', 'midas_snps_batch_timing', 'midas_snps_batch_time_pileup_only',
    'midas_snps_batch_stats_to_device', 'midas_snps_batch_pack', 'midas_snps_batch_fetch_packed',
    'midas_snps_batch_select_path', 'midas_snps_set_default_path', 'midas_snps_copy_rate', 'midas_snps_stream_rates', 'midas_snps_calibration_pass', 'midas_snps_set_pad_rule', 'midas_snps_set_row_coder',
    'midas_snps_batch_pack_timing',
    'midas_bam_open', 'midas_bam_close', 'midas_bam_write', 'midas_bam_n_refs', 'midas_bam_ref', 'midas_bam_load', 'midas_bam_copy', 'midas_bam_columns',
    'midas_bam_open_slice', 'midas_bam_slice_facts', 'midas_bam_slice_marks', 'midas_bam_load_ranges',
    'midas_bam_open_device', 'midas_bam_open_slice_device', 'midas_bam_load_ranges_device', 'midas_snps_inflate_blocks', 'midas_bam_load_device',
    'midas_bam_payload_on_device', 'midas_snps_copy_from_device', 'midas_bam_release_file',
    'midas_bam_load_resident', 'midas_bam_load_ranges_resident', 'midas_bam_is_resident', 'midas_bam_resident_to_columns',
    'midas_snps_batch_create_resident', 'midas_bam_open_share_local', 'midas_bam_share_locate',
    'midas_snps_write_rows', 'midas_snps_write_table', 'midas_snps_write_part', 'midas_snps_write_pieces', 'midas_snps_deflate_rows',
    'midas_snps_tableset_open', 'midas_snps_tableset_read_counts', 'midas_snps_tableset_close', 'midas_snps_batch_write_part',
    'midas_fasta_load', 'midas_fasta_n_records', 'midas_fasta_columns', 'midas_fasta_close',
    'midas_snps_table_open', 'midas_snps_table_open_range', 'midas_snps_table_count_rows', 'midas_snps_table_close', 'midas_snps_table_rows', 'midas_snps_table_key_bytes',
    'midas_snps_table_copy', 'midas_merge_sites', 'midas_genes_count', 'midas_genes_terms', 'midas_genes_sum', 'midas_merge_write_info',
    'midas_merge_write_matrix',
    'midas_bam_open_share',
    'midas_comm_device_key', 'midas_comm_probe', 'midas_comm_unique_id', 'midas_comm_create', 'midas_comm_destroy', 'midas_comm_all_gather', 'midas_comm_all_to_all_v',
]


def deflate_rows(text: bytes, row_begin, tail_begin) -> bytes:
    """Raw DEFLATE stream of row-structured text by the library's row coder (midas_snps_deflate_rows)."""
    lib = load_library()
    buf = np.frombuffer(text, dtype=np.uint8)
    rb = np.ascontiguousarray(row_begin, dtype=np.uint32)
    tb = np.ascontiguousarray(tail_begin, dtype=np.uint32)
    out = np.empty(len(text) + len(text) // 8 + 4096, dtype=np.uint8)
    n_out = C.c_int64(0)
    st = lib.midas_snps_deflate_rows(buf.ctypes.data, len(text), rb.ctypes.data, tb.ctypes.data, rb.size, out.ctypes.data,
                                     out.size, C.byref(n_out))
    if st != 0:
        raise MidasSnpsError(st, "midas_snps_deflate_rows: bad row offsets or output too small")
    return out[:n_out.value].tobytes()


def write_rows(path: str, append: bool, ref_id: str, allele: np.ndarray, counts: np.ndarray,
               gz_level: int = 4, threads: int = 0):
    """Format + gzip the rows of ONE contig into <species>.snps.gz (midas_snps_write_rows)."""
    lib = load_library()
    allele = np.ascontiguousarray(allele, dtype=np.uint8)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    assert counts.shape == (allele.shape[0], 4)
    err = C.create_string_buffer(256)
    st = lib.midas_snps_write_rows(path.encode(), 1 if append else 0, ref_id.encode(), allele.shape[0],
                                   allele.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                   int(gz_level), int(threads), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())


ROWS_PER_MEMBER = 16384     # MIDAS_SNPS_ROWS_PER_MEMBER: pieces of a contig start at multiples of this


def write_table(path: str, ref_ids, alleles, counts, gz_level: int = 4, threads: int = 0, header=None, first_pos=None):
    """Header + the rows of every contig of one species in one call (midas_snps_write_table): ref_ids[k],
    alleles[k] (u8[n_k]) and counts[k] (u32[n_k,4]) describe contig k in output order.  header = True / False writes a
    PART of a species' table instead (midas_snps_write_part): with or without the header member in front.  first_pos:
    the entries are pieces of contigs, entry k's first row is position first_pos[k] + 1 (midas_snps_write_pieces)."""
    lib = load_library()
    n = len(ref_ids)
    al = [np.ascontiguousarray(a, dtype=np.uint8) for a in alleles]
    cn = [np.ascontiguousarray(c, dtype=np.uint32) for c in counts]
    assert all(c.shape == (a.shape[0], 4) for a, c in zip(al, cn))
    ids = (C.c_char_p * max(n, 1))(*[r.encode() for r in ref_ids])
    ns = (C.c_int64 * max(n, 1))(*[a.shape[0] for a in al])
    pa = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in al])
    pc = (C.c_void_p * max(n, 1))(*[c.ctypes.data for c in cn])
    err = C.create_string_buffer(256)
    if first_pos is not None:
        fp = (C.c_int64 * max(n, 1))(*[int(x) for x in first_pos])
        st = lib.midas_snps_write_pieces(path.encode(), 1 if (header is None or header) else 0, n, ids, ns, fp, pa, pc,
                                         int(gz_level), int(threads), err)
    elif header is None:
        st = lib.midas_snps_write_table(path.encode(), n, ids, ns, pa, pc, int(gz_level), int(threads), err)
    else:
        st = lib.midas_snps_write_part(path.encode(), 1 if header else 0, n, ids, ns, pa, pc, int(gz_level), int(threads), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())


class _Genes(C.Structure):
    _fields_ = [("n_genes", C.c_int64), ("scaffold_id", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p),
                ("strand", C.c_char_p), ("gene_type", C.c_void_p), ("gene_id", C.c_void_p), ("seq", C.c_void_p)]


def write_merge_info(path: str, header_line: str, keep: np.ndarray, keys, key_off: np.ndarray, res: dict, genes: list,
                     threads: int = 0, site_id_base: int = 0):
    """snps_info.txt for the kept sites (midas_merge_write_info): annotation + the per-site calls.  keys / key_off as
    returned by read_snps_table, res = Context.merge_sites(...), genes = the species' genes in the reference's order
    (dicts with scaffold_id, start, end, strand, gene_type, gene_id, seq)."""
    lib = load_library()
    keep = np.ascontiguousarray(keep, dtype=np.int64)
    key_off = np.ascontiguousarray(key_off, dtype=np.int64)
    kb = np.frombuffer(keys, dtype=np.uint8) if not isinstance(keys, np.ndarray) else keys
    calls = res['major'].base if res['major'].base is not None else np.stack([res['major'], res['minor'], res['snp_type'], res['flag']], 1)
    calls = np.ascontiguousarray(calls, dtype=np.uint8)
    cs = np.ascontiguousarray(res['count_samples'], dtype=np.uint32)
    pooled = np.ascontiguousarray(res['pooled'], dtype=np.uint64)
    n = len(genes)

    def strs(field):
        vals = [str(g[field]).encode() for g in genes]
        return (C.c_char_p * max(n, 1))(*vals)
    sid, gty, gid, seq = strs('scaffold_id'), strs('gene_type'), strs('gene_id'), strs('seq')
    start = np.array([g['start'] for g in genes], dtype=np.int64)
    end = np.array([g['end'] for g in genes], dtype=np.int64)
    strand = "".join(str(g['strand'])[:1] or '+' for g in genes).encode()
    gs = _Genes(n, C.cast(sid, C.c_void_p), start.ctypes.data_as(C.c_void_p), end.ctypes.data_as(C.c_void_p), strand,
                C.cast(gty, C.c_void_p), C.cast(gid, C.c_void_p), C.cast(seq, C.c_void_p))
    err = C.create_string_buffer(256)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    st = lib.midas_merge_write_info(path.encode(), header_line.encode(), keep.shape[0], p(keep), p(kb), p(key_off), p(calls),
                                    p(cs), p(pooled), C.byref(gs), int(threads), int(site_id_base), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())


def write_merge_matrix(path: str, header_line: str, keep: np.ndarray, depth: np.ndarray, minor_count=None, threads: int = 0,
                       site_id_base: int = 0):
    """snps_depth.txt (minor_count None) / snps_freq.txt of merge_midas.py snps for the kept sites (midas_merge_write_matrix).
    depth, minor_count: [n_samples, n_sites] uint32 as returned by Context.merge_sites."""
    lib = load_library()
    keep = np.ascontiguousarray(keep, dtype=np.int64)
    depth = np.ascontiguousarray(depth, dtype=np.uint32)
    S, n = depth.shape
    mc = None
    if minor_count is not None:
        mc = np.ascontiguousarray(minor_count, dtype=np.uint32)
        assert mc.shape == depth.shape
    err = C.create_string_buffer(256)
    st = lib.midas_merge_write_matrix(path.encode(), header_line.encode(), keep.shape[0], keep.ctypes.data_as(C.c_void_p),
                                      S, n, depth.ctypes.data_as(C.c_void_p),
                                      mc.ctypes.data_as(C.c_void_p) if mc is not None else None, int(threads), int(site_id_base), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())


def count_snps_rows(path: str) -> int:
    """Rows of a <species>.snps.gz without inflating it, or -1 when the file does not say (midas_snps_table_count_rows)."""
    lib = load_library()
    n = C.c_int64(-1)
    err = C.create_string_buffer(256)
    st = lib.midas_snps_table_count_rows(path.encode(), C.byref(n), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())
    return int(n.value)


def read_snps_table(path: str, max_rows: int = -1, want_keys: bool = True, row_begin: int = 0):
    """Parse one <species>.snps.gz with the native reader -> (counts[n,4] u32, keys | None, key_off | None);
    keys is a byte buffer, row i's 'ref_id|ref_pos|ref_allele' is bytes(keys[key_off[i]:key_off[i + 1]]).
    Rows [row_begin, max_rows) of the table (max_rows < 0: to the end)."""
    lib = load_library()
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    st = lib.midas_snps_table_open_range(path.encode(), int(row_begin), int(max_rows), 1 if want_keys else 0, C.byref(h), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())
    try:
        n = int(lib.midas_snps_table_rows(h))
        counts = np.empty((n, 4), np.uint32)
        keys = np.empty(int(lib.midas_snps_table_key_bytes(h)), np.uint8) if want_keys else None
        key_off = np.empty(n + 1, np.int64) if want_keys else None
        p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
        lib.midas_snps_table_copy(h, p(counts), p(keys), p(key_off))
    finally:
        lib.midas_snps_table_close(h)
    return counts, (memoryview(keys) if want_keys else None), key_off


def read_snps_counts(paths, row_begin: int = 0, max_rows: int = -1):
    """The count columns of several samples' <species>.snps.gz in one parallel region (midas_snps_tableset_*):
    -> list of counts[n,4] u32, one per path, n = rows [row_begin, max_rows) of the SHORTEST table (max_rows < 0: its
    end) -- the reference's zip over the samples' files stops there too.  None when one of the files does not announce
    its rows (written by the reference): read those with read_snps_table."""
    lib = load_library()
    n = len(paths)
    arr = (C.c_char_p * n)(*[p.encode() for p in paths])
    rows = np.empty(n, np.int64)
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    st = lib.midas_snps_tableset_open(n, arr, C.byref(h), rows.ctypes.data_as(C.c_void_p), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())
    try:
        if (rows < 0).any():
            return None
        hi = int(rows.min()) if max_rows < 0 else min(int(rows.min()), int(max_rows))
        m = max(0, hi - int(row_begin))
        out = [np.empty((m, 4), np.uint32) for _ in range(n)]
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in out])
        st = lib.midas_snps_tableset_read_counts(h, int(row_begin), m, ptrs, err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        return out
    finally:
        lib.midas_snps_tableset_close(h)


def read_bam(path: str, ctx=None, payload_on_device: bool = False, resident: bool = False):
    """Decode a BAM with the native reader -> (ref_names, ref_lengths, refid[int32], ReadsSoA).  The arrays are views of
    the decoder's own buffers (no copy); the native handle lives as long as any of them does.  ctx (a Context): the BGZF
    blocks are inflated on its device instead of by the host's threads (midas_bam_open_device); with payload_on_device SEQ,
    QUAL and CIGAR are also cut out of the inflated stream there and never come down (midas_bam_load_device): the ReadsSoA
    carries their device addresses (`device`) and only that context's batches can take it."""
    lib = load_library()
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    if resident:
        # ONE pass from the BAM's bytes to the kernel's input: every column stays on the device, in the direct layout; refID
        # alone comes down (midas_bam_load_resident).  More payload than the layout addresses: the columns' way.
        if ctx is None or not getattr(ctx, 'inflates', False):
            raise MidasSnpsError(ERR_INVALID_ARG, "resident needs a device context")
        n, total = C.c_int64(), C.c_int64()
        st = lib.midas_bam_load_resident(path.encode(), ctx._h, C.byref(h), C.byref(n), C.byref(total), err)
        if st == ERR_UNSUPPORTED:
            return read_bam(path, ctx, payload_on_device=True)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        owner = _BamOwner(lib, h)
        names, lens = _bam_refs(lib, h)
        ptrs = (C.c_void_p * 12)()
        lib.midas_bam_columns(h, ptrs)
        refid = np.asarray(_Column(owner, ptrs[0] or 0, int(n.value), np.int32))
        return names, lens, refid, ResidentReads(lib, h, owner, int(n.value), int(total.value))
    if payload_on_device:
        if ctx is None or not getattr(ctx, 'inflates', False):
            raise MidasSnpsError(ERR_INVALID_ARG, "payload_on_device needs a device context")
        n, sb, qb, nc = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        st = lib.midas_bam_load_device(path.encode(), ctx._h, C.byref(h), C.byref(n), C.byref(sb), C.byref(qb), C.byref(nc), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        owner = _BamOwner(lib, h)
        names, lens = _bam_refs(lib, h)
        refid, reads = _bam_columns(lib, h, int(n.value), int(sb.value), int(qb.value), int(nc.value), owner, on_device=True)
        return names, lens, refid, reads
    if ctx is not None and getattr(ctx, 'inflates', False):
        st = lib.midas_bam_open_device(path.encode(), ctx._h, C.byref(h), err)
    else:
        st = lib.midas_bam_open(path.encode(), C.byref(h), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())
    owner = _BamOwner(lib, h)
    names, lens = _bam_refs(lib, h)
    n, sb, qb, nc = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    st = lib.midas_bam_load(h, C.byref(n), C.byref(sb), C.byref(qb), C.byref(nc), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())
    refid, reads = _bam_columns(lib, h, int(n.value), int(sb.value), int(qb.value), int(nc.value), owner)
    return names, lens, refid, reads


def _bam_refs(lib, h):
    names, lens = [], []
    for i in range(lib.midas_bam_n_refs(h)):
        nm = C.c_char_p()
        ln = C.c_int64()
        lib.midas_bam_ref(h, i, C.byref(nm), C.byref(ln))
        names.append(nm.value.decode())
        lens.append(int(ln.value))
    return names, lens


class _BamOwner:
    """Keeps a decoded BAM alive for as long as a numpy view of one of its columns is (midas_bam_columns: no copies)."""

    def __init__(self, lib, h):
        self._lib, self._h = lib, h

    def __del__(self):
        if self._h:
            self._lib.midas_bam_close(self._h)
            self._h = None


class _Column:
    def __init__(self, owner, ptr, n, dtype):
        self._owner = owner
        dt = np.dtype(dtype)
        self.__array_interface__ = {'shape': (int(n),), 'typestr': dt.str, 'data': (int(ptr) if n else 0, True), 'version': 3} \
            if n else np.empty(0, dt).__array_interface__


class _FastaOwner:
    def __init__(self, lib, h):
        self._lib, self._h = lib, h

    def __del__(self):
        if self._h:
            self._lib.midas_fasta_close(self._h)
            self._h = None


def read_fasta_files(paths, threads: int = 0):
    """The FASTA files `paths` (plain or gzip) read by all cores (midas_fasta_load): (pool, records) -- every sequence back to
    back in one read-only uint8 array (whitespace out, ASCII letters upper-cased), records = [(id, file index, offset, length)]
    in file and record order: what midas_amd/fasta.py parse_bytes yields for each file, `seq.upper()` applied."""
    lib = load_library()
    n = len(paths)
    c_paths = (C.c_char_p * max(n, 1))(*[p.encode() for p in paths])
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    st = lib.midas_fasta_load(n, c_paths, int(threads), C.byref(h), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode() or "midas_fasta_load failed")
    owner = _FastaOwner(lib, h)
    nrec = int(lib.midas_fasta_n_records(h))
    ptrs, sizes = (C.c_void_p * 6)(), (C.c_int64 * 2)()
    st = lib.midas_fasta_columns(h, ptrs, sizes)
    if st != 0:
        raise MidasSnpsError(st, "midas_fasta_columns failed")
    pool = np.asarray(_Column(owner, ptrs[0] or 0, int(sizes[0]), np.uint8))
    off = np.asarray(_Column(owner, ptrs[1] or 0, nrec, np.int64))
    ln = np.asarray(_Column(owner, ptrs[2] or 0, nrec, np.int64))
    fi = np.asarray(_Column(owner, ptrs[3] or 0, nrec, np.int32))
    ids = bytes(np.asarray(_Column(owner, ptrs[4] or 0, int(sizes[1]), np.uint8)))
    io = np.asarray(_Column(owner, ptrs[5] or 0, nrec + 1, np.int64))
    recs = [(ids[int(io[k]):int(io[k + 1])].decode('latin-1'), int(fi[k]), int(off[k]), int(ln[k])) for k in range(nrec)]
    return pool, recs


def write_bam(path, ref_names, ref_lengths, refid, reads, level=6, threads=0):
    """The native BAM writer (midas_bam_write): records in the given order, names "r<i>", aux NM + YT:Z:UU -- the bytes of
    midas_amd/bam.py's pure-Python writer, made by all cores."""
    lib = load_library()
    names = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
    lens = np.ascontiguousarray(ref_lengths, np.int64)
    rid = np.ascontiguousarray(refid, np.int32)
    r = reads._c()
    err = C.create_string_buffer(256)
    st = lib.midas_bam_write(path.encode(), len(ref_names), names, lens.ctypes.data_as(C.c_void_p), C.byref(r),
                             rid.ctypes.data_as(C.c_void_p), int(level), int(threads), err)
    if st != 0:
        raise MidasSnpsError(st, err.value.decode())


def _bam_columns(lib, h, n, sb, qb, nc, owner=None, on_device=False):
    """(refid, ReadsSoA) over the decoder's own buffers.  `owner` (a _BamOwner) is kept alive by every array.  on_device:
    the last three columns are device addresses (midas_bam_load_device) and go into ReadsSoA.device."""
    ptrs = (C.c_void_p * 12)()
    st = lib.midas_bam_columns(h, ptrs)
    if st != 0:
        raise MidasSnpsError(st, "midas_bam_columns failed")
    if owner is None:
        owner = _BamOwner(lib, None)      # (the caller keeps the handle itself)
    spec = [('refid', np.int32, n), ('pos', np.int32, n), ('mapq', np.uint8, n), ('flag', np.uint16, n), ('nm', np.int32, n),
            ('l_seq', np.int32, n), ('seq_off', np.int64, n + 1), ('qual_off', np.int64, n + 1), ('cigar_off', np.int64, n + 1),
            ('seq4', np.uint8, sb), ('qual', np.uint8, qb), ('cigar', np.uint32, nc)]
    a = {name: np.asarray(_Column(owner, ptrs[k] or 0, cnt, dt)) for k, (name, dt, cnt) in enumerate(spec[:9 if on_device else 12])}
    refid = a.pop('refid')
    if on_device:
        a.update(seq4=np.zeros(0, np.uint8), qual=np.zeros(0, np.uint8), cigar=np.zeros(0, np.uint32),
                 device=(ptrs[9] or 0, ptrs[10] or 0, ptrs[11] or 0, owner))
    return refid, ReadsSoA(**a)


class BamSlice:
    """Rank-local view of a BAM (midas_bam_open_slice): this rank's share of the file walked, facts to exchange with the
    other ranks, then only the record ranges this rank owns decoded (midas_bam_load_ranges)."""

    def __init__(self, path: str, slice_index: int, n_slices: int, ctx=None):
        """ctx (a Context): the slice's blocks are inflated and walked on its device (midas_bam_open_slice_device)."""
        self._lib = load_library()
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        if ctx is not None and getattr(ctx, 'inflates', False):
            st = self._lib.midas_bam_open_slice_device(path.encode(), int(slice_index), int(n_slices), ctx._h, C.byref(h), err)
        else:
            st = self._lib.midas_bam_open_slice(path.encode(), int(slice_index), int(n_slices), C.byref(h), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        self._h = h
        self.ref_names, self.ref_lens = _bam_refs(self._lib, h)
        n = len(self.ref_names)
        out7 = np.zeros(7, np.int64)
        self.ref_reads, self.ref_bases, self.ref_first = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        self._lib.midas_bam_slice_facts(h, p(out7), p(self.ref_reads), p(self.ref_bases), p(self.ref_first))
        (self.first, self.end, self.sorted, self.first_ref, self.last_ref, self.rec_begin, self.total) = (int(x) for x in out7)

    def marks(self):
        """(pos_sorted, first_pos, last_pos, ref_span int64 [n_ref], marks int64 [n, 3] = refID, bin, offset): what a
        contig needs to be cut into pieces (midas_bam_slice_marks)."""
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        out4 = np.zeros(4, np.int64)
        span = np.zeros(len(self.ref_names), np.int64)
        self._lib.midas_bam_slice_marks(self._h, p(out4), p(span), None, 0)
        marks = np.zeros((int(out4[3]), 3), np.int64)
        if marks.size:
            self._lib.midas_bam_slice_marks(self._h, p(out4), None, p(marks), marks.shape[0])
        return int(out4[0]), int(out4[1]), int(out4[2]), span, marks

    def load_ranges(self, ranges, ctx=None, resident: bool = False):
        """[(begin, end)] uncompressed record ranges -> (refid int32, ReadsSoA) of the records in them, in file order.
        ctx (a Context): the ranges are decoded on its device and SEQ / QUAL / CIGAR stay there (midas_bam_load_ranges_device):
        the ReadsSoA carries their device addresses (`device`), as read_bam(..., payload_on_device=True)'s does."""
        rb = np.array([r[0] for r in ranges], np.int64)
        re_ = np.array([r[1] for r in ranges], np.int64)
        n, sb, qb, nc = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        err = C.create_string_buffer(256)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        if resident and ctx is not None and getattr(ctx, 'inflates', False) and len(ranges):
            # (everything stays on the device, in the direct layout: read_bam(..., resident=True)'s form for a rank's ranges)
            total = C.c_int64()
            st = self._lib.midas_bam_load_ranges_resident(self._h, ctx._h, len(ranges), p(rb), p(re_), C.byref(n), C.byref(total), err)
            if st == 0 and self._lib.midas_bam_is_resident(self._h):
                keeper = _BamOwner(self._lib, None)
                keeper._slice = self
                ptrs = (C.c_void_p * 12)()
                self._lib.midas_bam_columns(self._h, ptrs)
                refid = np.asarray(_Column(keeper, ptrs[0] or 0, int(n.value), np.int32))
                return refid, ResidentReads(self._lib, self._h, keeper, int(n.value), int(total.value))
            if st != ERR_UNSUPPORTED:
                if st != 0:
                    raise MidasSnpsError(st, err.value.decode())
        if ctx is not None and getattr(ctx, 'inflates', False):
            st = self._lib.midas_bam_load_ranges_device(self._h, ctx._h, len(ranges), p(rb), p(re_), C.byref(n), C.byref(sb),
                                                        C.byref(qb), C.byref(nc), err)
        else:
            st = self._lib.midas_bam_load_ranges(self._h, len(ranges), p(rb), p(re_), C.byref(n), C.byref(sb), C.byref(qb),
                                                 C.byref(nc), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        keeper = _BamOwner(self._lib, None)
        keeper._slice = self                # the arrays keep this object (and with it the native handle) alive
        return _bam_columns(self._lib, self._h, int(n.value), int(sb.value), int(qb.value), int(nc.value), keeper,
                            on_device=bool(self._lib.midas_bam_payload_on_device(self._h)))

    def release_file(self):
        """The ranges are loaded: the file's mapping is unmapped on a thread of its own (midas_bam_release_file)."""
        if getattr(self, '_h', None):
            self._lib.midas_bam_release_file(self._h)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.midas_bam_close(self._h)
            self._h = None

    __del__ = close


class BamShare(BamSlice):
    """A rank's CONTIGUOUS share of a coordinate-sorted BAM (midas_bam_open_share): where this rank's equal share of the file's
    bytes begins, moved forward to the next reference's first record.  `first` (-1: no reference border nearby), `total`,
    `rec_begin`; load_ranges as a slice's -- the one-pass rank-local decode of midas_amd/run/snps.py."""

    def __init__(self, path: str, slice_index: int, n_slices: int, max_walk: int = 256 << 20):
        self._lib = load_library()
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        out3 = np.zeros(3, np.int64)
        st = self._lib.midas_bam_open_share(path.encode(), int(slice_index), int(n_slices), int(max_walk), C.byref(h),
                                            out3.ctypes.data_as(C.c_void_p), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        self._h = h
        self.ref_names, self.ref_lens = _bam_refs(self._lib, h)
        self.first, self.total, self.rec_begin = (int(x) for x in out3)

    @classmethod
    def open_local(cls, path: str, slice_index: int, n_slices: int):
        """The share with a LOCAL block table (midas_bam_open_share_local): the rank walks the BGZF chain over its own 1 / N of the
        file only.  `walk` = (first block's file offset, where the walk ended, uncompressed bytes, file size): the ranks exchange
        these, check that they chain, and call locate()."""
        self = cls.__new__(cls)
        self._lib = load_library()
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        out4 = np.zeros(4, np.int64)
        st = self._lib.midas_bam_open_share_local(path.encode(), int(slice_index), int(n_slices), C.byref(h),
                                                  out4.ctypes.data_as(C.c_void_p), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        self._h, self._slice = h, int(slice_index)
        self.ref_names, self.ref_lens = _bam_refs(self._lib, h)
        self.walk = tuple(int(x) for x in out4)
        self.first = self.total = self.rec_begin = -1
        return self

    def locate(self, upos_base: int, total: int, max_walk: int = 256 << 20):
        out3 = np.zeros(3, np.int64)
        err = C.create_string_buffer(256)
        st = self._lib.midas_bam_share_locate(self._h, self._slice, int(upos_base), int(total), int(max_walk),
                                              out3.ctypes.data_as(C.c_void_p), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode())
        self.first, self.total, self.rec_begin = (int(x) for x in out3)
        return self


class PinnedPool:
    """Page-locked result buffers of a Context (midas_snps_host_alloc): grown on demand and REUSED across calls -- pinning
    a quarter of a gigabyte costs more than the copy it speeds up.  An array handed out under a key is overwritten by the
    next request for that key."""

    def __init__(self, lib):
        self._lib = lib
        self._bufs = {}

    def array(self, key, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = max(1, int(np.prod(shape)) * dtype.itemsize)
        ptr, cap = self._bufs.get(key, (None, 0))
        if cap < nbytes:
            if ptr:
                self._lib.midas_snps_host_free(C.c_void_p(ptr))
            ptr = self._lib.midas_snps_host_alloc(nbytes)
            if not ptr:
                self._bufs.pop(key, None)
                return np.empty(shape, dtype)          # no pinned memory: an ordinary array (staged copy)
            self._bufs[key] = (ptr, nbytes)
        return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        for ptr, _ in self._bufs.values():
            self._lib.midas_snps_host_free(C.c_void_p(ptr))
        self._bufs = {}


class Context:
    """One GPU.  Replaces a `mp.Pool` worker and its module globals (midas/run/snps.py:142,167-176)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._pool = None
        h = C.c_void_p()
        st = self._lib.midas_snps_create(int(device), C.byref(h))
        if st != 0:
            raise MidasSnpsError(st, "midas_snps_create(device=%d): %s"
                                 % (device, self._lib.midas_snps_status_string(st).decode()))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, '_pool', None):
            self._pool.close()
            self._pool = None
        if getattr(self, '_h', None):
            self._lib.midas_snps_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def pinned(self, key, shape, dtype):
        """A page-locked array owned by the context (see PinnedPool): valid until the same key is asked for again."""
        if self._pool is None:
            self._pool = PinnedPool(self._lib)
        return self._pool.array(key, shape, dtype)

    def _check(self, st: int):
        if st != 0:
            msg = self._lib.midas_snps_last_error(self._h).decode() or \
                self._lib.midas_snps_status_string(st).decode()
            raise MidasSnpsError(st, msg, int(self._lib.midas_snps_last_error_read(self._h)))

    def set_stream(self, hip_stream: Optional[int]):
        self._check(self._lib.midas_snps_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def set_default_path(self, path: int):
        """The path of every batch created on this context from now on (PATH_AUTO: each batch's own choice)."""
        self._check(self._lib.midas_snps_set_default_path(self._h, int(path)))

    def set_pad_rule(self, rule: int):
        """PAD_SPEC (default): the CIGAR op P consumes nothing; PAD_PYSAM: it advances the query position, as
        get_aligned_pairs of the pysam releases of MIDAS's time does.  For batches created afterwards."""
        self._check(self._lib.midas_snps_set_pad_rule(self._h, int(rule)))

    inflates = True      # read_bam / BamSlice.load_ranges may hand this context the BGZF blocks

    def inflate_blocks(self, comp, cpos, clen, upos, ulen, out_bytes: int, crc=None):
        """Raw DEFLATE streams comp[cpos[k], +clen[k]) -> out[upos[k], +ulen[k]) on the device (midas_snps_inflate_blocks).
        crc: the CRC-32 every stream's inflated bytes must have (None: not checked).
        Raises MidasSnpsError (ERR_BAD_LAYOUT, read_index = the stream) when a stream is corrupt."""
        comp = np.ascontiguousarray(np.frombuffer(comp, np.uint8) if not isinstance(comp, np.ndarray) else comp, dtype=np.uint8)
        cpos, upos = np.ascontiguousarray(cpos, np.int64), np.ascontiguousarray(upos, np.int64)
        clen, ulen = np.ascontiguousarray(clen, np.int32), np.ascontiguousarray(ulen, np.int32)
        out = np.zeros(max(int(out_bytes), 1), np.uint8)
        bad = C.c_int64(-1)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        crc = None if crc is None else np.ascontiguousarray(crc, np.uint32)
        st = self._lib.midas_snps_inflate_blocks(self._h, p(comp), comp.size, cpos.size, p(cpos), p(clen), p(upos), p(ulen),
                                                 None if crc is None else p(crc), p(out), int(out_bytes), C.byref(bad))
        if st != 0:
            msg = self._lib.midas_snps_last_error(self._h).decode() or self._lib.midas_snps_status_string(st).decode()
            raise MidasSnpsError(st, msg, int(bad.value))
        return out[:int(out_bytes)]

    def fetch_payload(self, reads: "ReadsSoA") -> "ReadsSoA":
        """A ReadsSoA whose SEQ / QUAL / CIGAR live on the device (read_bam(..., payload_on_device=True)) with those three
        columns copied down: for tests, and for host code that has to slice them."""
        if reads.device is None:
            return reads
        if isinstance(reads, ResidentReads):
            reads = reads.to_columns(self)
        n = reads.n_reads
        sizes = (int(reads.seq_off[n]), int(reads.qual_off[n]), int(reads.cigar_off[n]) * 4)
        out = [np.zeros(max(s, 1), np.uint8) for s in sizes]
        for buf, src, s in zip(out, reads.device[:3], sizes):
            self._check(self._lib.midas_snps_copy_from_device(self._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(int(src)), s))
        d = reads.as_dict()
        d.update(seq4=out[0][:sizes[0]], qual=out[1][:sizes[1]], cigar=out[2][:sizes[2]].view(np.uint32))
        return ReadsSoA(**d)

    def set_row_coder(self, coder: int):
        """ROWS_DEVICE (default): Batch.write_part formats and deflates the rows in a kernel; ROWS_HOST: the host's
        formatter threads do (the file then equals write_table's byte for byte).  Same text either way."""
        self._check(self._lib.midas_snps_set_row_coder(self._h, int(coder)))

    def copy_rate(self, nbytes: int = 1 << 30, reps: int = 10) -> float:
        """GB/s (read + written) of a device-to-device copy with the library's 16-bytes-per-lane copy kernel."""
        out = C.c_double(0.0)
        self._check(self._lib.midas_snps_copy_rate(self._h, int(nbytes), int(reps), C.byref(out)))
        return out.value

    def stream_rates(self, nbytes: int = 1 << 32, reps: int = 5):
        """GB/s of a saturating read stream, write stream and copy (read + written) over nbytes per buffer."""
        out = (C.c_double * 3)()
        self._check(self._lib.midas_snps_stream_rates(self._h, int(nbytes), int(reps), out))
        return {"read_GBps": out[0], "write_GBps": out[1], "copy_GBps": out[2]}

    def calibration_pass(self, nbytes: int = 1 << 30):
        """The known-byte-count kernels that calibrate FETCH_SIZE / WRITE_SIZE (run it under rocprofv3 --pmc)."""
        self._check(self._lib.midas_snps_calibration_pass(self._h, int(nbytes)))

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu = C.c_int32(0)
        mem = C.c_int64(0)
        self._check(self._lib.midas_snps_device_info(self._h, name, C.byref(ncu), C.byref(mem)))
        return {'name': name.value.decode(), 'compute_units': ncu.value, 'hbm_bytes': mem.value}

    def pileup(self, thr: Thresholds, contigs: ContigTable, reads: ReadsSoA, want_allele: bool = True, pinned_slot=None):
        """One-shot midas_snps_pileup(): returns (counts[n_sites,4] u32, allele[n_sites] u8 | None, stats[n_species,4] i64).
        pinned_slot: the per-site results land in the context's page-locked buffers of that name (one DMA, no staging)
        and stay valid until the next call with the same slot."""
        n = contigs.n_sites
        if pinned_slot is None:
            counts = np.empty((n, 4), dtype=np.uint32)
            allele = np.empty(n, dtype=np.uint8) if want_allele else None
        else:
            counts = self.pinned(('counts', pinned_slot), (n, 4), np.uint32)
            allele = self.pinned(('allele', pinned_slot), (n,), np.uint8) if want_allele else None
        stats = np.zeros((contigs.n_species, NUM_STATS), dtype=np.int64)
        c, r = contigs._c(), reads._c()
        st = self._lib.midas_snps_pileup(self._h, C.byref(thr), C.byref(c), C.byref(r),
                                         counts.ctypes.data_as(C.c_void_p),
                                         allele.ctypes.data_as(C.c_void_p) if want_allele else None,
                                         stats.ctypes.data_as(C.c_void_p))
        self._check(st)
        return counts, allele, stats

    def genes_count(self, thr: Thresholds, reads: "ReadsSoA", ref_id, gene_length):
        """midas_genes_count(): per gene (aligned_reads i64, mapped_reads i64, depth f64, kernel_ms)."""
        rid = np.ascontiguousarray(ref_id, dtype=np.int32)
        gl = np.ascontiguousarray(gene_length, dtype=np.int64)
        n = gl.shape[0]
        aligned, mapped, depth = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.float64)
        ms = C.c_float(0)
        r = reads._c()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        st = self._lib.midas_genes_count(self._h, C.byref(thr), C.byref(r), p(rid), n, p(gl), p(aligned), p(mapped), p(depth),
                                         C.byref(ms))
        self._check(st)
        return aligned, mapped, depth, float(ms.value)

    def genes_terms(self, thr: Thresholds, reads: "ReadsSoA", ref_id, gene_length):
        """midas_genes_terms(): per read of a slice its term (f64; +0.0 for a read keep_read drops)."""
        rid = np.ascontiguousarray(ref_id, dtype=np.int32)
        gl = np.ascontiguousarray(gene_length, dtype=np.int64)
        term = np.zeros(int(reads.n_reads), np.float64)
        ms = C.c_float(0)
        r = reads._c()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        st = self._lib.midas_genes_terms(self._h, C.byref(thr), C.byref(r), p(rid), gl.shape[0], p(gl), p(term), C.byref(ms))
        self._check(st)
        return term

    def genes_sum(self, gene, term, n_genes):
        """midas_genes_sum(): (gene, term) pairs in BAM order -> per gene (aligned_reads i64, mapped_reads i64, depth f64)."""
        g = np.ascontiguousarray(gene, dtype=np.int32)
        t = np.ascontiguousarray(term, dtype=np.float64)
        n = int(n_genes)
        aligned, mapped, depth = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.float64)
        ms = C.c_float(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        st = self._lib.midas_genes_sum(self._h, g.shape[0], p(g), p(t), n, p(aligned), p(mapped), p(depth), C.byref(ms))
        self._check(st)
        return aligned, mapped, depth

    def merge_sites(self, prm: "MergeParams", sample_counts, mean_depth):
        """midas_merge_sites(): sample_counts = list of [n_sites,4] uint32 arrays (one per sample).
        -> dict(major, minor, snp_type, flag, count_samples, pooled[n,4] u64, depth[S,n] u32, minor_count[S,n] u32, kernel_ms)"""
        S = len(sample_counts)
        arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in sample_counts]
        n = arrs[0].shape[0]
        assert all(a.shape == (n, 4) for a in arrs)
        ptrs = (C.c_void_p * S)(*[a.ctypes.data for a in arrs])
        md = np.ascontiguousarray(mean_depth, dtype=np.float64)
        calls = np.empty((n, 4), np.uint8)
        out = dict(major=calls[:, 0], minor=calls[:, 1], snp_type=calls[:, 2], flag=calls[:, 3],
                   count_samples=np.empty(n, np.uint32), pooled=np.empty((n, 4), np.uint64),
                   depth=np.empty((S, n), np.uint32), minor_count=np.empty((S, n), np.uint32))
        ms = C.c_float(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        st = self._lib.midas_merge_sites(self._h, C.byref(prm), S, n, ptrs, p(md), p(calls),
                                         p(out['count_samples']), p(out['pooled']),
                                         p(out['depth']), p(out['minor_count']), C.byref(ms))
        self._check(st)
        out['kernel_ms'] = float(ms.value)
        return out

    def batch(self, contigs: ContigTable, reads: ReadsSoA) -> "Batch":
        return Batch(self, contigs, reads)


class Comm:
    """An RCCL communicator of the ranks' contexts (comm.cpp): the all-gather of the summary rows over xGMI and the genes
    path's all-to-all, with no process group behind them.  midas_amd/dist.py makes one per job (the id travels through a file)."""

    def __init__(self, ctx: "Context", id128: bytes, rank: int, world: int):
        self._lib = ctx._lib
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        buf = C.create_string_buffer(bytes(id128), 128)
        st = self._lib.midas_comm_create(ctx._h, C.cast(buf, C.c_void_p), self.rank, self.world, C.byref(h), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode() or "midas_comm_create failed")
        self._h = h
        self._ctx = ctx          # (the communicator runs on the context's stream: keep it alive)

    @staticmethod
    def unique_id() -> bytes:
        lib = load_library()
        out, err = C.create_string_buffer(128), C.create_string_buffer(256)
        st = lib.midas_comm_unique_id(C.cast(out, C.c_void_p), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode() or "midas_comm_unique_id failed")
        return out.raw

    @staticmethod
    def probe() -> int:
        """RCCL's version when this process can use it at all (midas_comm_probe); raises MidasSnpsError when it cannot."""
        lib = load_library()
        v, err = C.c_int32(0), C.create_string_buffer(256)
        st = lib.midas_comm_probe(C.byref(v), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode() or "midas_comm_probe failed")
        return int(v.value)

    @staticmethod
    def device_key(ctx: "Context") -> str:
        out = C.create_string_buffer(64)
        ctx._check(ctx._lib.midas_comm_device_key(ctx._h, out))
        return out.value.decode()

    def close(self):
        if getattr(self, '_h', None):
            self._lib.midas_comm_destroy(self._h)
            self._h = None

    __del__ = close

    def all_gather(self, data: bytes):
        """every rank's `data` (the same length everywhere), by rank"""
        n = len(data)
        out, err = C.create_string_buffer(max(1, n * self.world)), C.create_string_buffer(256)
        src = C.create_string_buffer(bytes(data), max(1, n))
        st = self._lib.midas_comm_all_gather(self._h, C.cast(src, C.c_void_p), C.cast(out, C.c_void_p), n, err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode() or "midas_comm_all_gather failed")
        raw = out.raw
        return [raw[r * n:(r + 1) * n] for r in range(self.world)]

    def all_to_all_v(self, parts, recv_bytes):
        """parts[r]: the bytes for rank r; recv_bytes[r]: how many come from rank r -> the bytes received, by source rank"""
        send = b"".join(bytes(p) for p in parts)
        sb = np.array([len(p) for p in parts], np.int64)
        rb = np.array([int(x) for x in recv_bytes], np.int64)
        out, err = C.create_string_buffer(max(1, int(rb.sum()))), C.create_string_buffer(256)
        src = C.create_string_buffer(send, max(1, len(send)))
        st = self._lib.midas_comm_all_to_all_v(self._h, C.cast(src, C.c_void_p), sb.ctypes.data_as(C.c_void_p), C.cast(out, C.c_void_p),
                                               rb.ctypes.data_as(C.c_void_p), err)
        if st != 0:
            raise MidasSnpsError(st, err.value.decode() or "midas_comm_all_to_all_v failed")
        raw, got, at = out.raw, [], 0
        for n in rb:
            got.append(raw[at:at + int(n)])
            at += int(n)
        return got


class Batch:
    """Device-resident (contig table, reads): upload once, run many."""

    def __init__(self, ctx: Context, contigs: ContigTable, reads: ReadsSoA):
        self.ctx = ctx
        self._lib = ctx._lib
        self.n_sites = contigs.n_sites
        self.n_species = int(contigs.n_species)
        h = C.c_void_p()
        c = contigs._c()
        if isinstance(reads, ResidentReads):       # the handle's records where they lie (and the handle alive as long as the batch)
            self._keep = reads.owner
            ctx._check(self._lib.midas_snps_batch_create_resident(ctx._h, C.byref(c), reads._h, reads.first, C.byref(h)))
        else:
            r = reads._c()
            ctx._check(self._lib.midas_snps_batch_create(ctx._h, C.byref(c), C.byref(r), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, '_h', None):
            self._lib.midas_snps_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def run(self, thr: Thresholds):
        self.ctx._check(self._lib.midas_snps_batch_run(self._h, C.byref(thr)))
        if getattr(self, '_slots', 0):
            self._timed += 1

    def sync(self):
        self.ctx._check(self._lib.midas_snps_batch_sync(self._h))

    def fetch(self, counts: bool = True, allele: bool = True, stats: bool = True):
        oc = np.empty((self.n_sites, 4), dtype=np.uint32) if counts else None
        oa = np.empty(self.n_sites, dtype=np.uint8) if allele else None
        os_ = np.zeros((self.n_species, NUM_STATS), dtype=np.int64) if stats else None
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        self.ctx._check(self._lib.midas_snps_batch_fetch(self._h, p(oc), p(oa), p(os_)))
        return oc, oa, os_

    def write_part(self, path: str, contig_index, ref_ids, header: bool = True, gz_level: int = 4, threads: int = 0):
        """Rows of the contigs contig_index (indices into the batch's contig table, output order) straight from the device
        results into a gzip table or part of one (midas_snps_batch_write_part)."""
        n = len(contig_index)
        idx = np.ascontiguousarray(contig_index, dtype=np.int32)
        ids = (C.c_char_p * max(n, 1))(*[r.encode() for r in ref_ids])
        self.ctx._check(self._lib.midas_snps_batch_write_part(self._h, path.encode(), 1 if header else 0, n,
                                                              idx.ctypes.data_as(C.c_void_p), ids, int(gz_level), int(threads)))

    def info(self) -> BatchInfo:
        bi = BatchInfo()
        self.ctx._check(self._lib.midas_snps_batch_get_info(self._h, C.byref(bi)))
        return bi

    def enable_timing(self, n_slots: int = 1):
        """n_slots event triples; run k records into slot k % n_slots (0 turns timing off)."""
        self.ctx._check(self._lib.midas_snps_batch_enable_timing(self._h, int(n_slots)))
        self._slots = int(n_slots)
        self._timed = 0
        self._packs = 0

    def time_pileup_only(self, on: bool = True):
        """Timed runs record only the two events around the pileup kernel (index_ms reads 0, run_ms == pileup_ms)."""
        self.ctx._check(self._lib.midas_snps_batch_time_pileup_only(self._h, 1 if on else 0))

    def timing(self, slot: int = 0):
        ms = (C.c_float * 3)()
        self.ctx._check(self._lib.midas_snps_batch_timing(self._h, int(slot), ms))
        return {'index_ms': ms[0], 'pileup_ms': ms[1], 'run_ms': ms[2]}

    def last_timing(self):
        return self.timing((self._timed - 1) % self._slots)

    def select_path(self, path: int):
        """PATH_DIRECT: the pileup kernel reads the raw arrays; PATH_PACKED: tile-ordered records + payload; PATH_AUTO: the
        batch's own choice (midas_snps_batch_select_path)."""
        self.ctx._check(self._lib.midas_snps_batch_select_path(self._h, int(path)))

    def pack(self):
        """Re-run the device packer over the resident raw reads (midas_snps_batch_pack)."""
        self.ctx._check(self._lib.midas_snps_batch_pack(self._h))
        if getattr(self, '_slots', 0):
            self._packs += 1

    def pack_timing(self, slot: int = 0):
        ms = (C.c_float * 2)()
        self.ctx._check(self._lib.midas_snps_batch_pack_timing(self._h, int(slot), ms))
        return {'pack_ms': ms[0], 'scatter_ms': ms[1]}

    def fetch_packed(self):
        """The device-packed layout: (rec[n_records+1,16] u8 incl. the sentinel, blob u8, orig u32, key u32)."""
        n, nb = C.c_int64(0), C.c_int64(0)
        self.ctx._check(self._lib.midas_snps_batch_fetch_packed(self._h, None, None, None, None, C.byref(n), C.byref(nb)))
        rec = np.zeros((n.value + 1, 16), np.uint8)
        blob = np.zeros(max(nb.value, 1), np.uint8)
        orig = np.zeros(max(n.value, 1), np.uint32)
        key = np.zeros(max(n.value, 1), np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.ctx._check(self._lib.midas_snps_batch_fetch_packed(self._h, p(rec), p(blob), p(orig), p(key), C.byref(n), C.byref(nb)))
        return rec, blob[:nb.value], orig[:n.value], key[:n.value]

    def stats_to_device(self, dst_device_ptr: int):
        """Enqueue a D2D copy of the [n_species,4] int64 counters into caller-owned device memory."""
        self.ctx._check(self._lib.midas_snps_batch_stats_to_device(self._h, C.c_void_p(dst_device_ptr)))
