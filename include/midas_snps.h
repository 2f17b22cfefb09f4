/*
 * midas_snps.h -- C-ABI of the MI355X-native MIDAS SNP pileup (libmidas_snps_hip.so).
 *
 * This is the drop-in boundary for ONE path of snayfach/MIDAS: the pileup +
 * allele-count stage of `run_midas.py snps`.  The reference has no FFI for it;
 * the seam is Python (all citations are into /root/reference):
 *
 *   midas/run/snps.py:301      pysam_pileup(args, species, contigs)
 *   midas/run/snps.py:194-199  bamfile.count_coverage(contig.id, start=0, end=contig.length,
 *                                quality_threshold=args['baseq'], read_callback=keep_read)
 *   midas/run/snps.py:141-162  keep_read(aln)                  (per-read filter + 2 counters)
 *   midas/run/snps.py:201-213  per-site emit loop              (ref_allele, depth, 3 counters)
 *   midas/run/snps.py:130-137  index_bam                       (samtools index)
 *
 * Every entry point below names the reference line(s) it replaces.  Only plain C
 * types cross the boundary.  Ownership: the caller allocates and owns every host
 * buffer it passes in or receives results in; the library never keeps a host
 * pointer past the call that received it and never frees caller memory.  Device
 * memory is owned by the context / batch objects and released by their destroy
 * calls.  No global state; a context is bound to one GPU and may be used by one
 * thread at a time (use one context per thread / per rank).
 *
 * Error convention: every call returns MIDAS_SNPS_OK (0) or a negative /
 * positive status; midas_snps_last_error() returns a human-readable message for
 * the last failing call on that context.  Nothing aborts or throws across the
 * ABI.  The Python host turns a non-zero status into the reference's own
 * convention, sys.exit("\nError: ...\n") (midas/utility.py:227-232).
 *
 * There is NO CPU fallback behind these symbols: if no gfx950 device is present
 * midas_snps_create() fails with MIDAS_SNPS_ERR_NO_DEVICE.  The host-only
 * helpers (pack / plan / version) work without a GPU.
 */
#ifndef MIDAS_SNPS_H
#define MIDAS_SNPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIDAS_SNPS_ABI_VERSION 4      /* 4: midas_snps_batch_info grew by direct_chunk_tiles, direct_overhang */

/* ---- status codes ------------------------------------------------------- */
enum {
  MIDAS_SNPS_OK = 0,
  /* Per-read conditions under which the reference raises inside the worker
   * (numbering shared with oracle/pileup_oracle.py).  The offending read index
   * (lowest one) is in the error string and in midas_snps_last_error_read(). */
  MIDAS_SNPS_ERR_READ_NO_SEQ = 1,        /* len(aln.query_alignment_sequence) on None: TypeError  (snps.py:145) */
  MIDAS_SNPS_ERR_READ_NO_NM = 2,         /* dict(aln.tags)['NM']: KeyError                         (snps.py:148) */
  MIDAS_SNPS_ERR_READ_ZERO_ALIGN = 3,    /* /float(align_len) with align_len == 0: ZeroDivisionError (snps.py:148) */
  MIDAS_SNPS_ERR_READ_NO_QUAL = 4,       /* np.mean(None): TypeError                               (snps.py:151) */
  MIDAS_SNPS_ERR_READ_CIGAR_OVERRUN = 5, /* seq[qpos] past l_seq on a kept read: IndexError        (pysam count_coverage) */
  MIDAS_SNPS_ERR_READ_BAD_CIGAR_OP = 6,  /* CIGAR op code > 8                                      */
  /* Library / argument errors */
  MIDAS_SNPS_ERR_INVALID_ARG = -1,
  MIDAS_SNPS_ERR_NO_DEVICE = -2,         /* no HIP device / not gfx950 / HIP runtime failure at create */
  MIDAS_SNPS_ERR_HIP = -3,               /* a HIP runtime call failed (message has hipGetErrorString) */
  MIDAS_SNPS_ERR_OUT_OF_MEMORY = -4,
  MIDAS_SNPS_ERR_UNSUPPORTED = -5,       /* a batch beyond what one batch addresses: > 2*10^9 reads, > 32 GiB of read payload
                                            (the caller splits it); baseq > 62 on a PACKED batch holding qualities above 62;
                                            selecting the DIRECT / PACKED path on a batch that holds a read beyond their
                                            limits (l_seq > 1024, n_cigar / NM > 65534: such a batch RUNS, on the long path);
                                            a contig of length < 1 (pysam: "interval of size 0") or >= 2^31 (BAM's limit)     */
  MIDAS_SNPS_ERR_BAD_LAYOUT = -6         /* offsets out of range / not monotone, read_begin inconsistent */
};

/* ---- thresholds: the five `args[...]` values the path reads --------------
 * scripts/run_midas.py:410-419 (defaults mapid 94.0, mapq 20, baseq 30, readq 20,
 * aln_cov 0.75); consumed at midas/run/snps.py:148-157 and :198.               */
typedef struct midas_snps_thresholds {
  int32_t baseq;    /* count a base iff qual >= baseq (baseq <= 0: every base)   */
  int32_t mapq;     /* keep read iff mapq >= mapq                                 */
  int32_t readq;    /* keep read iff mean(qual over all l_seq bases) >= readq     */
  int32_t reserved; /* must be 0                                                  */
  double mapid;     /* keep read iff 100*(align_len-NM)/float(align_len) >= mapid */
  double aln_cov;   /* keep read iff align_len/float(l_seq) >= aln_cov            */
} midas_snps_thresholds;

/* ---- inputs ---------------------------------------------------------------
 * Alignment records in BAM-native encodings, struct-of-arrays, host memory,
 * grouped by contig in the order of the contig table and (for speed, not for
 * correctness) sorted by pos inside a contig -- i.e. the order of the
 * coordinate-sorted snps/temp/genomes.bam (midas/run/snps.py:116-120).
 *   seq4:  4-bit codes "=ACMGRSVTWYHKDBN", two per byte, first base in the high nibble
 *   qual:  phred bytes, l_seq per read; qual[0]==0xFF means QUAL absent
 *   cigar: u32 = len<<4 | op, op in MIDNSHP=X (0..8)
 *   nm:    NM aux tag, or -1 when the record has none
 *   *_off: CSR offsets, n_reads+1 entries each (seq_off/qual_off in bytes, cigar_off in u32 elements)
 * `flag` is carried for completeness; the path never looks at it (with a callable
 * read_callback pysam applies no flag filter), and it may be NULL.               */
typedef struct midas_snps_reads {
  int64_t n_reads;
  const int32_t* pos;
  const uint8_t* mapq;
  const uint16_t* flag;
  const int32_t* nm;
  const int32_t* l_seq;
  const int64_t* seq_off;
  const int64_t* qual_off;
  const int64_t* cigar_off;
  const uint8_t* seq4;        /* these three: host memory, or memory of the context's device (midas_bam_load_device leaves  */
  const uint8_t* qual;        /* them there) -- all three alike; midas_snps_batch_create / midas_snps_pileup copy either way;     */
  const uint32_t* cigar;      /* the host-only entry points (midas_snps_write_*, midas_genes_*) take host memory only                 */
} midas_snps_reads;

/* Contig table: what initialize_contigs() builds (midas/run/snps.py:55-67), flattened.
 * `ref` holds the FASTA letters of all contigs back to back in table order (any case;
 * the library upper-cases ASCII a-z exactly like str.upper() at snps.py:62).
 * Reads of contig c are read indices [read_begin[c], read_begin[c+1]).            */
typedef struct midas_snps_contigs {
  int32_t n_contigs;
  int32_t n_species;
  const int64_t* length;      /* [n_contigs]   > 0                                  */
  const int32_t* species;     /* [n_contigs]   species index in [0, n_species)      */
  const int64_t* read_begin;  /* [n_contigs+1] non-decreasing, last == n_reads      */
  const uint8_t* ref;         /* [sum(length)]                                      */
  /* NULL, or [n_contigs]: entry c is a PIECE of a longer contig, starting at its 0-based position origin[c] (0: the contig's
   * start).  A long contig can then be piled up piece by piece -- on different GPUs -- and the pieces' tables concatenated
   * (midas_snps_write_part takes the position its rows start at).  `length`, `ref` and the read positions are the piece's:
   * pos is relative to origin[c] and NEGATIVE for a read that starts in front of the piece and reaches into it -- such a
   * read is tallied where it covers the piece, but with origin[c] > 0 it is the PREVIOUS piece's read: it is not counted in
   * aligned_reads / mapped_reads here and what keep_read would raise for it is not reported here (its own piece does both;
   * a CIGAR that overruns SEQ at a site of THIS piece is reported here, where the walk meets it).
   * count_coverage is called per contig at midas/run/snps.py:194-199; the split is this library's.                       */
  const int64_t* origin;
} midas_snps_contigs;

/* Per-species counters, in this order (midas/run/snps.py:172-176, 211-213, 143, 161).
 * genome_length is sum(length) and is left to the caller.                          */
enum {
  MIDAS_SNPS_STAT_ALIGNED_READS = 0,
  MIDAS_SNPS_STAT_MAPPED_READS = 1,
  MIDAS_SNPS_STAT_COVERED_BASES = 2,
  MIDAS_SNPS_STAT_TOTAL_DEPTH = 3,
  MIDAS_SNPS_NUM_STATS = 4
};

typedef struct midas_snps_ctx midas_snps_ctx;
typedef struct midas_snps_batch midas_snps_batch;

/* ---- library ------------------------------------------------------------ */
int32_t midas_snps_abi_version(void);
/* Host threads the library's parallel regions use (BAM inflate / decode, row formatting + gzip, table parsing): the CPUs the
 * process may run on -- hardware threads, its affinity mask, a cgroup CPU quota, whichever is least -- divided by
 * LOCAL_WORLD_SIZE under torchrun.  Replaces the `threads` argument of utility.parallel's mp.Pool (midas/utility.py:81-107). */
int32_t midas_snps_cpu_budget(void);
/* Static string for a status code (never NULL). */
const char* midas_snps_status_string(int32_t status);

/* ---- context: one per GPU ------------------------------------------------
 * Replaces the per-worker-process module globals `aln_stats` / `global_args`
 * (midas/run/snps.py:142,167-176) and the mp.Pool worker itself
 * (midas/utility.py:81-107).                                                     */
int32_t midas_snps_create(int32_t device_ordinal, midas_snps_ctx** out_ctx);
void midas_snps_destroy(midas_snps_ctx* ctx);
const char* midas_snps_last_error(const midas_snps_ctx* ctx);
/* Lowest index of a read that made the last pileup fail with a MIDAS_SNPS_ERR_READ_* status, else -1. */
int64_t midas_snps_last_error_read(const midas_snps_ctx* ctx);
/* Launch on an existing HIP stream (hipStream_t as void*; e.g. torch's current stream) instead of
 * the context's own.  NULL restores the context's stream -- so HIP's null (legacy default) stream
 * cannot be named here: a caller whose framework runs on the default stream (torch.cuda.current_stream()
 * has the handle 0 until a stream is made current) makes a stream of its own current and passes that
 * one, as bench.py does; otherwise the framework's copies and collectives do not wait for this
 * library's kernels.                                                                             */
int32_t midas_snps_set_stream(midas_snps_ctx* ctx, void* hip_stream);
/* Device facts for logs: name (<=255 chars), compute units, HBM bytes. */
int32_t midas_snps_device_info(const midas_snps_ctx* ctx, char* name256, int32_t* n_cu, int64_t* hbm_bytes);

/* What a plain device-to-device copy reaches on this device right now, in GB/s (bytes read + bytes written per second):
 * `reps` passes of the library's own 16-bytes-per-lane copy kernel over `bytes` (>= 1 MiB), timed with HIP events.  The
 * practical HBM ceiling the roofline fractions of bench.py are also held against (measurement aid; no reference counterpart). */
int32_t midas_snps_copy_rate(midas_snps_ctx* ctx, int64_t bytes, int32_t reps, double* out_gbps);
/* The same question asked properly: a read stream, a write stream and a copy over `bytes` per buffer (use >= 4 GiB: far
 * beyond the 256 MiB Infinity Cache), `reps` passes each, by kernels built to saturate the memory pipe (a persistent grid
 * of 4 workgroups per CU, 16 bytes per lane, four independent accesses in flight).  out_gbps[0] bytes read per second by the
 * read stream, [1] bytes written per second by the write stream, [2] bytes read + written per second by the copy.  bench.py
 * quotes the largest of them as this box's ceiling (measurement aid; no reference counterpart).                       */
int32_t midas_snps_stream_rates(midas_snps_ctx* ctx, int64_t bytes, int32_t reps, double out_gbps[3]);
/* Counter calibration for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: reads `bytes` once with 4-, 8- and 16-byte-per-lane
 * loads and writes them once with 16-byte stores, one kernel each (midas_calib_read4_kernel, _read8_, _read16_,
 * midas_calib_write16_kernel): the counter value of each against the bytes it is known to move is the factor bench.py
 * applies to a product kernel with that access width (measurement aid; no reference counterpart).                     */
int32_t midas_snps_calibration_pass(midas_snps_ctx* ctx, int64_t bytes);

/* Page-locked host memory (hipHostMalloc).  Optional: every entry point takes ordinary memory; result buffers that come
 * from here are filled by one DMA instead of through the context's staging ring, and a caller that keeps them for the
 * lifetime of its context pays the pinning once.  NULL when no device / out of memory.                            */
void* midas_snps_host_alloc(int64_t bytes);
void midas_snps_host_free(void* ptr);

/* ---- one-shot: host buffers in, host buffers out --------------------------
 * Replaces, for every contig of the table at once, the call
 *   bamfile.count_coverage(contig.id, 0, contig.length, args['baseq'], keep_read)
 * (midas/run/snps.py:194-199) together with keep_read (:141-162) and the counter
 * part of the emit loop (:204-213).
 *   out_counts [sum(length) * 4] u32, per site A,C,G,T      (counts[0..3][i], :205-208)
 *   out_allele [sum(length)]     u8, upper-cased ref letter  (contig.seq[i],   :203) -- may be NULL
 *   out_stats  [n_species * 4]   i64, MIDAS_SNPS_STAT_* order
 * On a MIDAS_SNPS_ERR_READ_* status the outputs are unspecified, as in the
 * reference where the worker's exception discards them.                           */
int32_t midas_snps_pileup(midas_snps_ctx* ctx, const midas_snps_thresholds* thr,
                          const midas_snps_contigs* contigs, const midas_snps_reads* reads,
                          uint32_t* out_counts, uint8_t* out_allele, int64_t* out_stats);

/* ---- resident batches: upload once, run many ------------------------------
 * A batch is the device-resident form of (contig table, reads): the caller's BAM-native arrays
 * are uploaded as they are, every read is validated on the device, and the DIRECT layout is built from
 * them once -- one 16-byte record per read (pos, l_seq, n_cigar, NM, mapq, payload offset) and the read's
 * CIGAR / SEQ / QUAL bytes as one run of a payload in BAM's own order: columns re-encoded and bytes
 * gathered, nothing decided -- next to the reference letters and the tile table.  (The PACKED layout,
 * tile-ordered records + one byte per base, is built on first use: unsorted input, coverage hot spots.)
 * Creating it replaces what `pysam.AlignmentFile(bampath)` + htslib's record decode do per worker
 * (midas/run/snps.py:186); running it replaces index_bam (:130-137, the device
 * builds its own per-tile read index each run) and the calls named above.      */
int32_t midas_snps_batch_create(midas_snps_ctx* ctx, const midas_snps_contigs* contigs,
                                const midas_snps_reads* reads, midas_snps_batch** out_batch);
void midas_snps_batch_destroy(midas_snps_batch* batch);
/* The device paths of a batch.  DIRECT: the pileup kernel reads a read's 16-byte record and its BAM-order payload -- 4-bit
 * SEQ, QUAL, CIGAR as the BAM holds them -- and visits every read once: a pass over the positions alone (4 bytes per
 * read) finds, per 2048-site tile, the run of reads that can touch it; the kernel decides in registers what a read's CIGAR
 * is and tallies it; nothing is sorted or decided beforehand.  It wants position-sorted reads (what samtools sort
 * writes; any order is CORRECT, only slower).  PACKED: the reads are first laid out in tile order as records + one byte per
 * base (midas_snps_batch_pack), which handles any order and cuts coverage hot spots into parts.  batch_create picks DIRECT
 * unless the reads are badly ordered, one read spans many tiles or a tile holds a hot spot; AUTO restores that choice.
 * Both paths give bit-identical results (tests/test_gpu_direct.py).  LONG: a batch that holds a read beyond the two fast
 * paths' limits -- l_seq above 1024, more than 65 534 CIGAR ops, NM above 65 534; pysam knows no such limits
 * (midas/run/snps.py:187-199) -- is not refused: batch_create gives it the long path (pileup_long.hip: one thread per read,
 * op by op and base by base from the caller's arrays, global atomics, the ratio tests in fp64), exact and slow, the only
 * path such a batch has (selecting DIRECT or PACKED on it is MIDAS_SNPS_ERR_UNSUPPORTED; any batch may select LONG).      */
enum { MIDAS_SNPS_PATH_AUTO = 0, MIDAS_SNPS_PATH_DIRECT = 1, MIDAS_SNPS_PATH_PACKED = 2, MIDAS_SNPS_PATH_LONG = 3 };
int32_t midas_snps_batch_select_path(midas_snps_batch* batch, int32_t path);
/* The path of every batch created on the context from now on (the one-shot midas_snps_pileup included); AUTO = each
 * batch's own choice.                                                                                              */
int32_t midas_snps_set_default_path(midas_snps_ctx* ctx, int32_t path);
/* What the CIGAR op P (padding) does to the QUERY position in the walk behind count_coverage ([EXT] pysam
 * get_aligned_pairs(matches_only=True), reached from midas/run/snps.py:194-199):
 *   MIDAS_SNPS_PAD_SPEC   (default) nothing -- the SAM specification: P consumes neither query nor reference; the worked
 *                         example of SAMv1 section 1.1 (read r002, 3S6M1P1I4M) only comes out as printed under this rule;
 *   MIDAS_SNPS_PAD_PYSAM  the query advances, as get_aligned_pairs of the pysam releases of MIDAS's time does (BAM_CPAD sits
 *                         in the branch of BAM_CINS / BAM_CSOFT_CLIP there): bases behind a pad are read one position late and
 *                         a read whose last match op then runs past SEQ inside the contig raises IndexError when kept
 *                         (MIDAS_SNPS_ERR_READ_CIGAR_OVERRUN here).
 * bowtie2 -- the only aligner run_midas.py snps drives -- never writes P, so the two rules give the same tables on every BAM
 * the reference itself produces; the switch exists because bit-identity is promised against the dependency, whose behaviour on
 * this op cannot be pinned in an image without pysam.  Clip lengths (query_alignment_start / _end) never look at P.  Applies
 * to batches created on the context afterwards (the one-shot midas_snps_pileup included).                              */
enum { MIDAS_SNPS_PAD_SPEC = 0, MIDAS_SNPS_PAD_PYSAM = 1 };
int32_t midas_snps_set_pad_rule(midas_snps_ctx* ctx, int32_t rule);
/* Who formats and deflates the rows midas_snps_batch_write_part writes at gz levels 1-5 (the row coder's levels):
 *   MIDAS_SNPS_ROWS_DEVICE (default)  a kernel, one workgroup per gzip member: what crosses the link is the DEFLATE stream
 *                                     (~4 bytes a row) and the host only frames and writes the members;
 *   MIDAS_SNPS_ROWS_HOST              the host's formatter threads, from counts and alleles streamed through the pinned ring
 *                                     (the file is then byte for byte what midas_snps_write_part writes).
 * Both give gzip files that inflate to the same text -- the rows of midas/run/snps.py:201-210 -- but not the same bytes: the
 * two coders choose their matches differently.  Every rank of a job uses the same coder, so parts still concatenate into the
 * file one rank writes.  The environment variable MIDAS_SNPS_ROW_CODER=host sets the default of new contexts.          */
enum { MIDAS_SNPS_ROWS_DEVICE = 0, MIDAS_SNPS_ROWS_HOST = 1 };
int32_t midas_snps_set_row_coder(midas_snps_ctx* ctx, int32_t coder);
/* Re-run the device packer over the batch's resident BAM-native arrays (the arrays batch_create uploaded, unchanged):
 * per read the CIGAR walk into gap-free match segments (pysam get_aligned_pairs(matches_only=True), reached from
 * midas/run/snps.py:194-199), the clip structure (query_alignment_sequence, :145), floor(mean(query_qualities))
 * (:151, a wave reduction), the 'A','C','G','T'-only rule folded into the quality bytes, and the tile order the
 * index kernel expects (:130-137).  batch_create already ran it once; this entry exists so that a caller can time
 * "raw reads resident in HBM -> counts" (pack + run).  Returns after the pack has finished (the hot-spot plan needs
 * the per-tile read counts on the host).                                                                          */
int32_t midas_snps_batch_pack(midas_snps_batch* batch);
/* Copy the packed device layout back (tests: it must equal the host mirror tests/mirror/pack_mirror.cpp bit for bit).
 * rec16 [(n_records+1)*16], blob [blob_bytes], orig_index / key [n_records]; any pointer may be NULL; the two sizes
 * are always returned.                                                                                            */
int32_t midas_snps_batch_fetch_packed(midas_snps_batch* batch, void* rec16, void* blob, uint32_t* orig_index,
                                      uint32_t* key, int64_t* out_n_records, int64_t* out_blob_bytes);
/* Device-side durations of a timed pack (enable_timing slots, like batch_timing): [0] whole pack, [1] its scatter
 * kernel (the dominant one: raw reads in, records + payload out).                                                 */
int32_t midas_snps_batch_pack_timing(midas_snps_batch* batch, int32_t slot, float out_ms[2]);
/* Enqueue one full pass (index + filter + pileup + per-species counters) on the context's stream;
 * asynchronous.  Results stay on the device until midas_snps_batch_fetch().                    */
int32_t midas_snps_batch_run(midas_snps_batch* batch, const midas_snps_thresholds* thr);
/* Wait for the stream, then report the status of the last run (MIDAS_SNPS_ERR_READ_* possible). */
int32_t midas_snps_batch_sync(midas_snps_batch* batch);
/* sync + copy results to host.  Any of the three pointers may be NULL. */
int32_t midas_snps_batch_fetch(midas_snps_batch* batch, uint32_t* out_counts, uint8_t* out_allele,
                               int64_t* out_stats);
/* Facts about a batch, for roofline accounting: */
typedef struct midas_snps_batch_info {
  int64_t n_reads;
  int64_t n_sites;          /* sum(length)                                               */
  int64_t n_tiles;
  int64_t packed_bytes;     /* device bytes of packed records + payload                  */
  int64_t algorithmic_bytes;/* sum(ceil(l/2)+l+4*n_cigar+16) + 17*n_sites  */
  int32_t tile_sites;
  int32_t lanes_per_read;
  int64_t n_work_items;     /* >= n_tiles: a tile holding a coverage hot spot is processed as several parts */
  int32_t path;             /* MIDAS_SNPS_PATH_DIRECT / _PACKED / _LONG: what batch_run takes                 */
  int32_t path_auto;        /* what the batch's own numbers recommend (see midas_snps_batch_select_path)       */
  int32_t lane_bases;       /* bases per lane of the active path's kernel                                      */
  int32_t layout_build_us;  /* device time (HIP events, microseconds) batch_create spent building the direct layout -- the 16-byte records
                               and the payload runs -- from the caller's arrays; 0 for a batch over resident records (the decoder wrote it) */
  int64_t direct_general_reads;   /* reads that are not one or two gap-free match runs (walked op by op)      */
  int64_t direct_reach;           /* longest reference span of a read: how far back a tile's read range reaches */
  int64_t direct_stream_reads;    /* sum over tiles of the reads their streams hold: n_reads + straddlers when sorted */
  int64_t direct_max_tile_reads;
  int32_t direct_chunk_tiles;     /* the direct path's work items: chunks of this many consecutive tiles, a read visited once per chunk and a
                                     tile's overhang carried on (4), or 1: every tile by itself (unsorted positions, reads longer than the
                                     longest overhang, more outlier reads than their list holds)                                              */
  int32_t direct_overhang;        /* sites behind a tile's last that its tallies hold: 160, or 288 for a batch whose longest read asks for it */
} midas_snps_batch_info;
int32_t midas_snps_batch_get_info(const midas_snps_batch* batch, midas_snps_batch_info* out);
/* Device-side durations from HIP events recorded on the run's stream.  enable_timing(batch, n_slots)
 * allocates n_slots event triples (0 = off); run number k (counted from enable) records into slot
 * k % n_slots, so a caller can time K asynchronous runs and read all K after one sync.
 * out_ms: [0] index kernel (+ workspace memsets), [1] pileup kernel, [2] whole run.
 * time_pileup_only(batch, 1): runs record only the two events around the pileup kernel (each event record takes
 * ~4 us of stream time); out_ms[0] is then 0 and out_ms[2] == out_ms[1].                              */
int32_t midas_snps_batch_enable_timing(midas_snps_batch* batch, int32_t n_slots);
int32_t midas_snps_batch_time_pileup_only(midas_snps_batch* batch, int32_t on);
int32_t midas_snps_batch_timing(midas_snps_batch* batch, int32_t slot, float out_ms[3]);
/* Enqueue (same stream) a device-to-device copy of the per-species counters [n_species*4] i64 of the
 * last run into caller-owned device memory -- e.g. a torch tensor that is then all-gathered over
 * RCCL; the counters never round-trip through the host.  Replaces the pickled (species_id, aln_stats)
 * tuples returned through the Pool pipe (midas/run/snps.py:216, 228).                             */
int32_t midas_snps_batch_stats_to_device(midas_snps_batch* batch, void* dst_device_i64);

/* ---- host I/O (no GPU needed) ------------------------------------------------
 * BAM decode.  Replaces `pysam.AlignmentFile(bampath, 'rb')` and htslib's record decode
 * (midas/run/snps.py:186): the whole coordinate-sorted snps/temp/genomes.bam is inflated (BGZF blocks in
 * parallel) and every record with refID >= 0 -- everything fetch(contig, 0, length) can return -- is
 * decoded into the BAM-native SoA of midas_snps_reads plus a refID per record.  No .bai is needed
 * (index_bam, midas/run/snps.py:130-137, becomes a no-op: the device indexes).                     */
typedef struct midas_bam midas_bam;
int32_t midas_bam_open(const char* path, midas_bam** out, char* err256);
void midas_bam_close(midas_bam* bam);
int32_t midas_bam_n_refs(const midas_bam* bam);
int32_t midas_bam_ref(const midas_bam* bam, int32_t i, const char** name, int64_t* length);
/* Decode; returns the array sizes the caller must allocate for midas_bam_copy(). */
int32_t midas_bam_load(midas_bam* bam, int64_t* n_reads, int64_t* seq_bytes, int64_t* qual_bytes,
                       int64_t* n_cigar, char* err256);
/* The decoded columns where they are, without a copy: out12 = {refid, pos, mapq, flag, nm, l_seq, seq_off, qual_off,
 * cigar_off, seq4, qual, cigar} (types as in midas_bam_copy).  They belong to `bam` and are valid until midas_bam_close:
 * the caller keeps the handle for as long as it uses them (the Python host wraps them as numpy arrays that do).      */
int32_t midas_bam_columns(const midas_bam* bam, const void** out12);
/* Copy the decoded arrays out (any pointer may be NULL); *_off arrays have n_reads+1 entries. */
int32_t midas_bam_copy(const midas_bam* bam, int32_t* refid, int32_t* pos, uint8_t* mapq, uint16_t* flag,
                       int32_t* nm, int32_t* l_seq, int64_t* seq_off, int64_t* qual_off, int64_t* cigar_off,
                       uint8_t* seq4, uint8_t* qual, uint32_t* cigar);

/* Rank-local decode (N GPUs, one process each): a rank must not inflate the whole BAM to use an eighth of it.
 *   midas_bam_open_slice   maps the file, reads the BGZF block table and the header, and walks the records that START in
 *                          this rank's share of the file's bytes (slice of n_slices).  A slice that does not begin at
 *                          the header's end has to GUESS its first record boundary (32 plausible records in a row);
 *   midas_bam_slice_facts  hands back what the walk found -- out7 = {first record offset, offset behind the slice's last
 *                          record, coordinate-sorted so far (0/1), first and last refID seen, offset of the file's first
 *                          record, uncompressed size}; per reference: records, sum(l_seq), offset of its first record
 *                          (-1 = none in this slice).  All offsets are into the uncompressed stream.  The caller
 *                          exchanges these few numbers between the ranks and TRUSTS a guessed start only when the slice
 *                          before it ended on exactly that offset (the chain starts at the header's end, which is
 *                          exact): guesses are verified, never believed;
 *   midas_bam_load_ranges  inflates only the blocks that hold the given record ranges [begin, end) (e.g. the contigs
 *                          this rank owns: from a reference's first record to the next reference's) and decodes them;
 *                          midas_bam_copy then hands the columns out as after midas_bam_load.
 * Replaces the same `pysam.AlignmentFile` + `fetch(contig, ...)` of midas/run/snps.py:186, 194-199 (which goes through
 * the .bai the reference builds at :130-137; no index file is needed here).                                           */
int32_t midas_bam_open_slice(const char* path, int32_t slice, int32_t n_slices, midas_bam** out, char* err256);
/* The ONE-PASS form of the rank-local decode, for a coordinate-sorted BAM whose references are short beside a rank's share
 * (the usual metagenome: thousands of contigs): the file is dealt to the ranks as CONTIGUOUS runs of whole references.  Rank
 * `slice` of `n_slices` looks where its equal share of the file's bytes begins, guesses a record start there and walks the
 * records -- on the host, a few blocks -- to the first record of the next reference, at most max_walk uncompressed bytes on.
 * out3 = {first, total, rec_begin}: `first` = the uncompressed offset its share begins at (rec_begin for slice 0; `total` for an
 * empty share; -1: no reference border within max_walk -- a long chromosome: plan with midas_bam_open_slice instead).  The ranks
 * exchange their `first` (one all-gather of one number each) and every rank decodes [first of its own, first of the next) ONCE
 * with midas_bam_load_ranges / _device on this handle -- which also proves the next rank's guess: a range must end exactly on a
 * record border, and rank 0 starts at the header's end.  The reference reads the file once per worker through the index
 * (midas/run/snps.py:186-199); so does this, without an index.                                                          */
int32_t midas_bam_open_share(const char* path, int32_t slice, int32_t n_slices, int64_t max_walk, midas_bam** out, int64_t* out3,
                             char* err256);
/* The decoder's inverse (host only; what puts the synthetic samples of bench.py and the tests on disk -- samtools is not in
 * the image): records in the given order, refid[i] = reference of record i (-1: unmapped), names "r<i>", aux = NM + YT:Z:UU,
 * BGZF blocks of 0xff00 bytes deflated at `level` (0-9) by `threads` threads (0: the CPU budget), CRC-32 and the EOF block
 * as the specification wants them.  No reference counterpart (bowtie2 | samtools view | samtools sort write genomes.bam,
 * midas/run/snps.py:97-128).                                                                                         */
int32_t midas_bam_write(const char* path, int32_t n_ref, const char* const* ref_names, const int64_t* ref_lens,
                        const midas_snps_reads* reads, const int32_t* refid, int32_t level, int32_t threads, char* err256);
/* The same decoders with the BGZF blocks inflated ON THE DEVICE of `ctx` (bgzf_inflate.hip: one lane per block decodes its
 * Huffman codes -- by comparison against per-length limits in registers -- into tokens, a wavefront per block lays the bytes out) instead of by the host's threads: the compressed bytes go up, the inflated stream comes back, records are
 * walked and decoded into columns by the host as before.  On a 16-CPU host inflating is two thirds of the pileup stage once
 * the pileup and the row coder are on the device; with eight ranks sharing a node's CPUs it is more.  Same results, same
 * statuses (a corrupt block: MIDAS_SNPS_ERR_BAD_LAYOUT).  At this boundary a device failure is a status (ERR_OUT_OF_MEMORY,
 * ERR_HIP), never a silent fall-back; the Python host (midas_amd/run/snps.py, --device_inflate auto) answers those two with the
 * host decode and passes every other status on.
 *   midas_bam_open_device          = midas_bam_open (whole file; then midas_bam_load as usual)
 *   midas_bam_open_slice_device    = midas_bam_open_slice with the slice's blocks inflated AND walked on the device (bam_walk.hip):
 *                                  what comes down is refID / pos / l_seq / reference span / offset of the slice's records, which
 *                                  the host folds into the same facts (a rank of an 8-rank node has two CPUs for a gigabyte of BAM)
 *   midas_bam_load_ranges_device   = midas_bam_load_ranges on a handle of midas_bam_open_slice, decoded like midas_bam_load_device:
 *                                  SEQ / QUAL / CIGAR of the ranges' records stay on the device (midas_bam_payload_on_device == 1)
 *   midas_snps_inflate_blocks      the inflater on its own: n raw DEFLATE streams comp[cpos[k], +clen[k]) -> out[upos[k],
 *                                  +ulen[k]) (host pointers; every stream must inflate to exactly ulen[k] bytes and, when
 *                                  `crc` is not NULL, to bytes whose CRC-32 is crc[k]).  On a corrupt stream *bad_block (may
 *                                  be NULL) is its index.
 * Every BGZF block is held to the CRC-32 in its footer on both decoders, as htslib's bgzf_read_block does: a block that
 * inflates to the right size from damaged bytes is MIDAS_SNPS_ERR_BAD_LAYOUT naming the block's file offset.
 * Replaces the inflate inside pysam.AlignmentFile / htslib's bgzf.c behind midas/run/snps.py:186.
 *   midas_bam_load_device          open + load in one call, and SEQ / QUAL / CIGAR never come down: the host walks the
 *                                  records and decodes the small columns from the inflated stream as before, the three payload
 *                                  columns are cut out of the stream where it lies on the device.  midas_bam_columns then hands
 *                                  out DEVICE addresses for entries 9-11 (midas_bam_payload_on_device says so; midas_bam_copy
 *                                  refuses them), which midas_snps_batch_create / midas_snps_pileup accept in
 *                                  midas_snps_reads.seq4 / qual / cigar (they copy device to device).  They belong to `bam`.
 *   midas_snps_copy_from_device    bytes of such a column into host memory (tests; host code that must slice a payload).   */
int32_t midas_bam_open_device(const char* path, midas_snps_ctx* ctx, midas_bam** out, char* err256);
int32_t midas_bam_open_slice_device(const char* path, int32_t slice, int32_t n_slices, midas_snps_ctx* ctx, midas_bam** out, char* err256);
int32_t midas_bam_load_device(const char* path, midas_snps_ctx* ctx, midas_bam** out, int64_t* n_reads, int64_t* seq_bytes,
                              int64_t* qual_bytes, int64_t* n_cigar, char* err256);
int32_t midas_bam_payload_on_device(const midas_bam* bam);
/* A handle of midas_bam_open_slice / _open_share whose ranges are loaded needs its file no more: the mapping is handed to a
 * thread that unmaps it (a page-table walk of 0.2 s for a 9 GB BAM) while the caller piles the records up; the columns / the
 * resident records stay.  midas_bam_load_ranges* on the handle afterwards is MIDAS_SNPS_ERR_INVALID_ARG.                       */
void midas_bam_release_file(midas_bam* bam);
int32_t midas_snps_copy_from_device(midas_snps_ctx* ctx, void* dst, const void* src, int64_t bytes);
int32_t midas_bam_load_ranges_device(midas_bam* bam, midas_snps_ctx* ctx, int32_t n_ranges, const int64_t* range_begin,
                                     const int64_t* range_end, int64_t* n_reads, int64_t* seq_bytes, int64_t* qual_bytes,
                                     int64_t* n_cigar, char* err256);
/* ONE pass from the BAM's bytes to the pileup kernel's input (round 6).  The reference opens the BAM and counts in one go
 * (midas/run/snps.py:186-199); so does this: the decode above with NOTHING cut into columns and gathered back -- a record's CIGAR
 * ops, 4-bit SEQ and QUAL are one run of its bytes, in the order the direct path's payload has, and the device decoder copies that
 * run ONCE into the payload and writes the read's 16-byte record beside it.  Every column stays where the device decoded it
 * ("resident"); the host gets refID alone (midas_bam_columns: entry 0, the others NULL -- it groups the records by contig with it).
 *   midas_bam_load_resident          = midas_bam_load_device; *sum_l_seq: the bases of all records (what a caller sizes batches by)
 *   midas_bam_load_ranges_resident   = midas_bam_load_ranges_device
 *   midas_snps_batch_create_resident = midas_snps_batch_create over the handle's records [first_read, first_read +
 *                                      contigs->read_begin[n_contigs]) where they lie: nothing is uploaded, copied or built; the
 *                                      batch's facts pass validates them as it validates a caller's arrays.  The batch borrows the
 *                                      handle's device memory: midas_bam_close comes AFTER midas_snps_batch_destroy.  The packed and the
 *                                      long path (unsorted positions, coverage hot spots, reads beyond the fast paths' limits) cut
 *                                      their three payload columns out of the handle's inflated stream (or, after a streamed
 *                                      decode, out of the direct layout) when first asked for.
 *   midas_bam_resident_to_columns    the fall-back for a host that must regroup or slice the records (contigs it does not
 *                                      want among them, pieces of long contigs): the small columns come down, SEQ / QUAL /
 *                                      CIGAR are cut into device columns -- the handle then answers as after midas_bam_load_device
 *                                      (and stays resident as well).
 * A resident decode of ONE run of blocks that fills the device's decoder several times over (> ~51 000 BGZF blocks on MI355X: a BAM
 * of more than ~1.3 GB) is STREAMED: groups of blocks go up on a thread of their own while the groups before them are inflated,
 * walked and written behind one another into the resident layout; device memory is a few slots of ~2.5 x a group's inflated bytes
 * + the result instead of ~2.3 x the file's inflated bytes, and no inflated stream is kept.  Same records, same statuses (a corrupt
 * block is named by its index in the file).  Environment (read at every call): MIDAS_SNPS_DECODE_STREAM=0 switches it off,
 * MIDAS_SNPS_DECODE_GROUP_BLOCKS / _SLOTS (1-4) / _SLOT_MB size the groups, their number in flight and a slot's cap.
 * MIDAS_SNPS_ERR_UNSUPPORTED: more than 32 GiB of read payload in one decode (the record's offset is 32 bits of 8-byte units):
 * decode with midas_bam_load_device instead.                                                                                    */
int32_t midas_bam_load_resident(const char* path, midas_snps_ctx* ctx, midas_bam** out, int64_t* n_reads, int64_t* sum_l_seq, char* err256);
int32_t midas_bam_load_ranges_resident(midas_bam* bam, midas_snps_ctx* ctx, int32_t n_ranges, const int64_t* range_begin,
                                       const int64_t* range_end, int64_t* n_reads, int64_t* sum_l_seq, char* err256);
int32_t midas_bam_is_resident(const midas_bam* bam);
/* midas_bam_open_share with a LOCAL block table (round 6): a rank of N walks the BGZF chain over its own 1 / N of the file's bytes
 * only -- eight ranks that each walk, and page in the block headers of, the whole of a 9 GB file spend a third of a second each on
 * it.  _open_share_local: out4 = {file offset of the share's first block (a guess: a header from which eight headers follow one
 * another), where the rank's walk ended, uncompressed bytes of its blocks, file size}.  The ranks exchange these and believe them
 * only if they CHAIN (rank 0 starts at 0, every walk ends where the next begins, the last at the file's end); then
 * _share_locate(bam, slice, upos_base = the uncompressed bytes of the ranks in front, total = of all ranks, max_walk, out3) puts the
 * table in its place and finds the share's first record: out3 as midas_bam_open_share's.  midas_bam_load_ranges* walk the chain on
 * as far as a range reaches.  Ranks whose walks do not chain open their shares with midas_bam_open_share.                          */
int32_t midas_bam_open_share_local(const char* path, int32_t slice, int32_t n_slices, midas_bam** out, int64_t* out4, char* err256);
int32_t midas_bam_share_locate(midas_bam* bam, int32_t slice, int64_t upos_base, int64_t total, int64_t max_walk, int64_t* out3, char* err256);
int32_t midas_bam_resident_to_columns(midas_bam* bam, midas_snps_ctx* ctx, int64_t* seq_bytes, int64_t* qual_bytes, int64_t* n_cigar, char* err256);
int32_t midas_snps_batch_create_resident(midas_snps_ctx* ctx, const midas_snps_contigs* contigs, const midas_bam* bam, int64_t first_read,
                                         midas_snps_batch** out_batch);
int32_t midas_snps_inflate_blocks(midas_snps_ctx* ctx, const uint8_t* comp, int64_t comp_bytes, int64_t n_blocks,
                                  const int64_t* cpos, const int32_t* clen, const int64_t* upos, const int32_t* ulen,
                                  const uint32_t* crc, uint8_t* out, int64_t out_bytes, int64_t* bad_block);
int32_t midas_bam_slice_facts(const midas_bam* bam, int64_t* out7, int64_t* ref_reads, int64_t* ref_bases,
                              int64_t* ref_first);
/* What the walk of midas_bam_open_slice noted for cutting LONG references into pieces (midas_snps_contigs.origin), so that a
 * rank that owns a piece of a 20 Mb contig inflates that piece's records and not the contig's:
 *   out4     = {positions non-decreasing inside every reference so far (0/1), position of the slice's first and of its last
 *               mapped record (-1: none), number of marks};
 *   ref_span   [n_ref] the longest stretch of reference a record of that reference covers (M, D, N, =, X lengths summed): a
 *               piece that starts at `lo` needs the records from position lo - max-over-slices(ref_span) on;
 *   marks      [3 * n] {refID, bin, offset}: offset (uncompressed stream) of the first record of refID whose position is in
 *               [bin, bin + 1) * MIDAS_BAM_MARK_SPAN or, when that bin has none, in a later bin -- listed whenever the bin
 *               goes up, bin > 0 only (bin 0 is ref_first).  With positions sorted, every record of refID at a position >=
 *               bin * MIDAS_BAM_MARK_SPAN lies at or behind the smallest such offset over the slices.
 * marks may be NULL (ask for the count first); marks_capacity is in marks.                                               */
#define MIDAS_BAM_MARK_SPAN 65536
int32_t midas_bam_slice_marks(const midas_bam* bam, int64_t* out4, int64_t* ref_span, int64_t* marks, int64_t marks_capacity);
int32_t midas_bam_load_ranges(midas_bam* bam, int32_t n_ranges, const int64_t* range_begin, const int64_t* range_end,
                              int64_t* n_reads, int64_t* seq_bytes, int64_t* qual_bytes, int64_t* n_cigar, char* err256);

/* Row formatter + gzip writer for <outdir>/snps/output/<species>.snps.gz.  Replaces the per-site emit
 * loop of midas/run/snps.py:201-210 and utility.iopen(...,'w') (midas/utility.py:194-206) for ONE contig:
 * rows `ref_id \t i+1 \t allele[i] \t A+C+G+T \t A \t C \t G \t T \n` for i in [0, n_sites).
 * append == 0 creates the file and writes the header line first; append != 0 appends gzip members
 * (concatenated members are one valid gzip stream).  Rows are formatted and deflated by `threads`
 * workers (0 = all hardware threads) in 16 384-row members and written in order.  gz_level 1-5: the library's row
 * coder (midas_snps_deflate_rows below: smaller than zlib level 6 on these tables, several times faster);
 * 6-9: zlib at that level; 0: stored.                                                              */
int32_t midas_snps_write_rows(const char* path, int32_t append, const char* ref_id, int64_t n_sites,
                              const uint8_t* allele, const uint32_t* counts, int32_t gz_level,
                              int32_t threads, char* err256);
/* All contigs of one species in ONE call (header + rows, same text as repeated midas_snps_write_rows calls):
 * the emit loop of species_pileup over sorted(contigs) (midas/run/snps.py:183-213).  ref_ids[k], n_sites[k],
 * allele[k], counts[k] describe contig k in output order; the formatter/deflater pool works across contigs.  */
int32_t midas_snps_write_table(const char* path, int32_t n_contigs, const char* const* ref_ids,
                               const int64_t* n_sites, const uint8_t* const* allele, const uint32_t* const* counts,
                               int32_t gz_level, int32_t threads, char* err256);

/* A PART of a species' table: the rows of some of its contigs (consecutive in the sorted-contig order of the emit
 * loop), with or without the header member in front.  When a species' contigs are spread over several GPUs every
 * rank writes the parts it owns and the parts are concatenated in sorted-contig order: gzip members concatenate into
 * one valid gzip stream, and because a member never spans two contigs the concatenation is byte for byte the file
 * one midas_snps_write_table call would have written (midas/run/snps.py:183-213).                                */
int32_t midas_snps_write_part(const char* path, int32_t with_header, int32_t n_contigs, const char* const* ref_ids,
                              const int64_t* n_sites, const uint8_t* const* allele, const uint32_t* const* counts,
                              int32_t gz_level, int32_t threads, char* err256);
/* The same for entries that are PIECES of contigs (midas_snps_contigs.origin): entry k's first row is position
 * first_pos[k] + 1 of ref_ids[k] (first_pos NULL: all 0).  A piece's members are cut every MIDAS_SNPS_ROWS_PER_MEMBER rows
 * from its first row, so when every piece but a contig's last holds a multiple of that many rows the concatenated pieces
 * are byte for byte the rows of the whole contig.  (midas_snps_batch_write_part does this by itself from the origins the
 * batch was created with.)                                                                                          */
#define MIDAS_SNPS_ROWS_PER_MEMBER 16384
int32_t midas_snps_write_pieces(const char* path, int32_t with_header, int32_t n_contigs, const char* const* ref_ids,
                                const int64_t* n_sites, const int64_t* first_pos, const uint8_t* const* allele,
                                const uint32_t* const* counts, int32_t gz_level, int32_t threads, char* err256);

/* The count columns of SEVERAL samples' tables in one go: what the lock-step zip over the samples' files does in
 * build_temp_count_matrix (midas/merge/snps.py:236-271, one r[-4:] per sample and row).  _open reads all files (in
 * parallel) and walks their gzip members' headers; rows_each[t] is table t's row count, or -1 for a file that does not
 * announce it (written by the reference): such a table is read with midas_snps_table_open instead.  _read_counts fills
 * out_counts[t][4 * n_rows] with rows [row_begin, row_begin + n_rows) of every table t whose pointer is not NULL: every
 * gzip member of every table is one task of a single parallel region (inflate, then the last four fields of each line
 * straight into the caller's array) -- no per-table thread pools, no intermediate copies.                          */
typedef struct midas_snps_tableset midas_snps_tableset;
int32_t midas_snps_tableset_open(int32_t n_tables, const char* const* paths, midas_snps_tableset** out, int64_t* rows_each,
                                 char* err256);
int32_t midas_snps_tableset_read_counts(midas_snps_tableset* ts, int64_t row_begin, int64_t n_rows,
                                        uint32_t* const* out_counts, char* err256);
void midas_snps_tableset_close(midas_snps_tableset* ts);

/* The same rows as midas_snps_write_part, taken from a batch's results WHERE THEY ARE -- on the device -- for the
 * contigs contig_index[k] (indices into the batch's contig table, in output order) under the names ref_ids[k]: the
 * emit loop of species_pileup (midas/run/snps.py:183-213) without a host copy of the counts.  The formatter's
 * threads work on one slab of sites in the context's page-locked ring while the next slab crosses the link; the file
 * is byte for byte what batch_fetch + midas_snps_write_part write.  After midas_snps_batch_run.                  */
int32_t midas_snps_batch_write_part(midas_snps_batch* b, const char* path, int32_t with_header, int32_t n_contigs,
                                    const int32_t* contig_index, const char* const* ref_ids, int32_t gz_level,
                                    int32_t threads);

/* The row coder behind gz_level 1-5 of the writers above, on its own (tests and tools reach it here): a raw DEFLATE
 * stream (RFC 1951, one final dynamic-Huffman block) for text[0, n) whose rows start at row_begin[k] and whose row
 * tails -- the part that tends to repeat an earlier row: from the tab before ref_allele on -- start at tail_begin[k]
 * (row_begin[k] <= tail_begin[k] < row_begin[k + 1], the last row ends at n).  One table lookup per row instead of
 * zlib's hash chains: what utility.iopen's gzip.open(..., 'w') costs the reference (midas/utility.py:194-206,
 * midas/run/snps.py:179-210).  Any inflater reads the result.  Returns the stream's bytes in out[0, *out_len);
 * MIDAS_SNPS_ERR_INVALID_ARG when out_cap is too small (n + n/8 + 4096 always suffices).                          */
int32_t midas_snps_deflate_rows(const uint8_t* text, int64_t n, const uint32_t* row_begin, const uint32_t* tail_begin,
                                int64_t n_rows, uint8_t* out, int64_t out_cap, int64_t* out_len);

/* Parser of one sample's <species>.snps.gz: replaces read_run_midas_snps + the per-line split of
 * build_temp_count_matrix (midas/merge/snps.py:236-271): per row the site key '|'.join(r[0:3]) and the counts
 * r[-4:].  max_rows < 0 reads everything (args['max_sites']); want_keys == 0 skips the keys (only the first
 * sample's are used, as in the reference).  key_off has rows+1 entries into the key bytes.            */
typedef struct midas_snps_table midas_snps_table;
int32_t midas_snps_table_open(const char* path, int64_t max_rows, int32_t want_keys, midas_snps_table** out,
                              char* err256);
/* The same for the table rows [row_begin, row_end) only (row_end < 0: to the end) -- what one rank of a site-sharded
 * merge reads (the reference shards build_sharded_tables by line range, midas/merge/snps.py:366-386).  Tables written by
 * this library say in every gzip member how many rows it holds, so only the members that hold the range are inflated;
 * any other file is read whole and cut.  midas_snps_table_count_rows: the table's rows without inflating anything, or
 * -1 when the file does not say (a table written by the reference or by round 1 of this library).                    */
int32_t midas_snps_table_open_range(const char* path, int64_t row_begin, int64_t row_end, int32_t want_keys,
                                    midas_snps_table** out, char* err256);
int32_t midas_snps_table_count_rows(const char* path, int64_t* out_rows, char* err256);
void midas_snps_table_close(midas_snps_table* table);
int64_t midas_snps_table_rows(const midas_snps_table* table);
int64_t midas_snps_table_key_bytes(const midas_snps_table* table);
int32_t midas_snps_table_copy(const midas_snps_table* table, uint32_t* counts, char* keys, int64_t* key_off);

/* ---- merge_midas.py snps: per-site cross-sample arithmetic (SURVEY 8f "next" #1) -------------------
 * Replaces, for every genomic site of a species at once, GenomicSite.__init__/compute_pooled_counts,
 * call_alleles, compute_per_sample_mafs, compute_prevalence and flag
 * (midas/merge/snps.py:13-114, driven from build_sharded_tables :324-364).  Annotation (:116-174) and the
 * text emission (:176-201) stay on the host.  Inputs are the samples' per-site count tables -- exactly what
 * the pileup stage emits -- and each sample's mean_coverage (midas/merge/merge.py:18-21).          */
typedef struct midas_merge_params {
  double allele_freq;   /* args['allele_freq']: freq >= allele_freq counts an allele as present (0.01)   */
  double site_ratio;    /* args['site_ratio']:  site_depth/mean_depth > site_ratio fails the sample (2.0) */
  double site_prev;     /* args['site_prev']:   prevalence < site_prev flags the site (0.95)              */
  int32_t site_depth;   /* args['site_depth']:  site_depth < site_depth fails the sample (1)              */
  int32_t snp_types;    /* args['snp_type'] as a bit set: 1 any, 2 mono, 4 bi, 8 tri, 16 quad             */
} midas_merge_params;
#define MIDAS_MERGE_ERR_ZERO_MEAN_DEPTH 7 /* ZeroDivisionError in compute_prevalence (snps.py:99) */
/* sample_counts[s] -> [n_sites*4] u32 (A,C,G,T per site) of sample s, host memory; mean_depth[s] = float(mean_coverage).
 * Outputs (host, caller-owned): calls [n_sites*4] = per site {major, minor, snp_type, flag}: major/minor 0..3 =
 * A,C,G,T, 255 = None; snp_type 0 None, 1 mono, 2 bi, 3 tri, 4 quad; flag 0 keep, 1 'min_prev', 2 'snp_type';
 * count_samples [n_sites], pooled [n_sites*4] u64, depth and minor_count [n_samples*n_sites] u32
 * (sample_depths and the numerator of sample_mafs; maf = float(minor_count)/depth if depth > 0 else 0.0).
 * out_kernel_ms (nullable): device time of the kernel(s), HIP events.                                */
int32_t midas_merge_sites(midas_snps_ctx* ctx, const midas_merge_params* params, int32_t n_samples, int64_t n_sites,
                          const uint32_t* const* sample_counts, const double* mean_depth, uint8_t* out_calls,
                          uint32_t* out_count_samples, uint64_t* out_pooled, uint32_t* out_depth, uint32_t* out_minor_count, float* out_kernel_ms);

/* snps_freq.txt / snps_depth.txt of merge_midas.py snps (GenomicSite.write, midas/merge/snps.py:196-201): header_line,
 * then one line per kept site `site_id \t v[sample 0] \t ...` with site_id = site_id_base + keep[r] + 1 (site_id_base: the
 * table row of the arrays' first site -- 0 unless the caller holds a row range of the table; header_line may be "").  depth / minor_count are
 * the [n_samples * n_sites] outputs of midas_merge_sites.  minor_count == NULL prints str(depth); otherwise
 * '{0:.3g}'.format(float(minor_count) / depth if depth > 0 else 0.0).  Host only, formatted by a thread pool.  */
int32_t midas_merge_write_matrix(const char* path, const char* header_line, int64_t n_keep, const int64_t* keep,
                                 int32_t n_samples, int64_t n_sites, const uint32_t* depth,
                                 const uint32_t* minor_count, int32_t threads, int64_t site_id_base, char* err256);

/* snps_info.txt of merge_midas.py snps: GenomicSite.annotate + fetch_ref_codon + the info line of GenomicSite.write
 * (midas/merge/snps.py:116-195) with utility.translate / index_replace (midas/utility.py:306-332) for the kept sites.
 * keys / key_off: 'ref_id|ref_pos|ref_allele' of every table row (midas_snps_table_copy); calls / count_samples /
 * pooled: outputs of midas_merge_sites; genes: the species' genes in the reference's order (scaffold_id, start, -end) --
 * scaffold ids, 1-based inclusive start/end, strand '+'/'-', gene_type and gene_id strings, and the gene sequence
 * oriented start to stop (get_gene_seq).  A site takes the first gene in that order that is not entirely behind it.  */
typedef struct midas_merge_genes {
  int64_t n_genes;
  const char* const* scaffold_id;
  const int64_t* start;
  const int64_t* end;
  const char* strand;          /* n_genes chars */
  const char* const* gene_type;
  const char* const* gene_id;
  const char* const* seq;
} midas_merge_genes;
int32_t midas_merge_write_info(const char* path, const char* header_line, int64_t n_keep, const int64_t* keep,
                               const char* keys, const int64_t* key_off, const uint8_t* calls,
                               const uint32_t* count_samples, const uint64_t* pooled, const midas_merge_genes* genes,
                               int32_t threads, int64_t site_id_base, char* err256);

/* ---- run_midas.py genes: reads per pangenome gene (SURVEY 8f "next" #4) -----------------------------------------
 * Replaces count_mapped_bp's pass over the BAM (midas/run/genes.py:165-180) for every gene at once: per gene the number
 * of alignments (aligned_reads), of alignments passing keep_read (genes.py:148-163 -- the same predicate as the snps
 * path, thresholds mapid / readq / mapq / aln_cov of `thr`; baseq unused) and depth = the sum, IN BAM ORDER, of
 * len(query_alignment_sequence) / float(gene_length) over the kept ones (fp64, bit-identical to the reference's
 * running sum).  reads: the alignments in BAM order (seq4 may be NULL: only l_seq, CIGAR, NM, QUAL and MAPQ are read);
 * ref_id[i]: index of read i's gene in gene_length.  A record the reference would raise on returns
 * MIDAS_SNPS_ERR_READ_NO_SEQ / _NO_NM / _ZERO_ALIGN / _NO_QUAL with the lowest offending read in
 * midas_snps_last_error_read.  out_kernel_ms (nullable): device time of the kernel.                                  */
int32_t midas_genes_count(midas_snps_ctx* ctx, const midas_snps_thresholds* thr, const midas_snps_reads* reads,
                          const int32_t* ref_id, int64_t n_genes, const int64_t* gene_length, int64_t* out_aligned,
                          int64_t* out_mapped, double* out_depth, float* out_kernel_ms);

/* The two halves of midas_genes_count, for N ranks below the species (midas_amd/run/genes.py): a gene's running fp64 sum has
 * to be formed in BAM order on ONE rank, so every rank turns ITS slice of the unsorted BAM into terms, the (gene, term) pairs
 * travel to the gene's owner (one all-to-all; the slices are in file order, so are the pairs a rank receives), and the owner sums.
 *   midas_genes_terms: out_term[i] = len(query_alignment_sequence) / float(gene_length[ref_id[i]]) for a read keep_read
 *     passes, +0.0 for one it drops (adding it leaves a running sum unchanged bit for bit: the pair still counts as an
 *     alignment of its gene).  Statuses and midas_snps_last_error_read as midas_genes_count (the read index is the slice's).
 *   midas_genes_sum: the pairs (gene[i] in [0, n_genes), term[i]), in the order their reads have in the BAM -> per gene
 *     the number of pairs (aligned_reads), of pairs with a term above zero (mapped_reads) and their sum in that order.
 * midas_genes_count(reads) == midas_genes_sum(ref_id, midas_genes_terms(reads)) bit for bit (tests/test_gpu_genes.py).     */
int32_t midas_genes_terms(midas_snps_ctx* ctx, const midas_snps_thresholds* thr, const midas_snps_reads* reads,
                          const int32_t* ref_id, int64_t n_genes, const int64_t* gene_length, double* out_term,
                          float* out_kernel_ms);
int32_t midas_genes_sum(midas_snps_ctx* ctx, int64_t n_pairs, const int32_t* gene, const double* term, int64_t n_genes,
                        int64_t* out_aligned, int64_t* out_mapped, double* out_depth, float* out_kernel_ms);

/* ---- the exchange between ranks (comm.cpp): RCCL itself, no process group ------------------------------------------------
 * One process per GPU; the only thing ranks exchange on this path is the per-species summary rows -- the reference's pool
 * workers return (species_id, aln_stats) through a pipe, midas/run/snps.py:225-241, midas/utility.py:81-107 -- plus, on the
 * genes path, the (gene, term) pairs (midas/run/genes.py:165-199 computes them in one process).  librccl.so is loaded at run
 * time by the first of these calls (MIDAS_SNPS_ERR_UNSUPPORTED with the reason in err256 when it cannot be); a single-GPU run
 * never touches it.  Rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the others however the caller likes
 * (midas_amd/dist.py: a file in <outdir>/snps/temp); every rank then joins with its context's device (ncclCommInitRank:
 * collective, returns when all have).  Buffers are the caller's host memory; the collective runs on the context's stream.
 *   all_gather     every rank's bytes_per_rank bytes, rank-major, to every rank (ncclAllGather over xGMI)
 *   all_to_all_v   send_bytes[r] bytes to rank r (taken from `send` back to back in rank order), recv_bytes[r] from rank r
 *                  (left in `recv` in rank order): grouped ncclSend / ncclRecv                                             */
typedef struct midas_comm midas_comm;
/* the PCI bus id of the context's device (two ranks on ONE device cannot form an RCCL communicator: the caller checks first) */
/* Whether this process can use RCCL at all (librccl loads and answers); *out_version: ncclGetVersion.  The ranks ask BEFORE they
 * call midas_comm_create together, and stay on their other transport if any of them cannot -- ncclCommInitRank waits for every
 * rank of the communicator, one that never arrives would hold the others there.                                                 */
int32_t midas_comm_probe(int32_t* out_version, char* err256);
int32_t midas_comm_device_key(midas_snps_ctx* ctx, char* out64);
int32_t midas_comm_unique_id(uint8_t* out_id128, char* err256);
int32_t midas_comm_create(midas_snps_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world, midas_comm** out, char* err256);
void midas_comm_destroy(midas_comm* comm);
int32_t midas_comm_all_gather(midas_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, char* err256);
int32_t midas_comm_all_to_all_v(midas_comm* comm, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes,
                                char* err256);

/* The selected species' representative genomes read by all cores: initialize_contigs (midas/run/snps.py:55-67 -- Bio.SeqIO
 * over genome.fna[.gz], `str(rec.seq).upper()`).  Every file is read through zlib's gz layer (plain text passes through), cut
 * into records at the '>' that start a line (id = the header's first whitespace-separated word; what precedes the first header
 * is no record), whitespace removed, ASCII a-z upper-cased; the sequences of all files lie back to back in one pool, in file and
 * record order.  columns: out[0..5] = pool (uint8), rec_off (int64), rec_len (int64), rec_file (int32: index into paths), ids
 * (char, back to back), id_off (int64, n_records + 1); sizes[0..1] = bytes of the pool and of the ids.  They belong to the handle. */
typedef struct midas_fasta midas_fasta;
int32_t midas_fasta_load(int32_t n_files, const char* const* paths, int32_t threads, midas_fasta** out, char* err256);
int64_t midas_fasta_n_records(const midas_fasta* f);
int32_t midas_fasta_columns(const midas_fasta* f, const void** out, int64_t* sizes);
void midas_fasta_close(midas_fasta* f);

#ifdef __cplusplus
}
#endif
#endif /* MIDAS_SNPS_H */
