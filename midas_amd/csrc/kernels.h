// Launch interface between the C-ABI runtime (snps_abi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace midas {

// The shape of the pileup kernels (both paths): tiles of 2048 sites = 32 KiB of LDS tallies, workgroups of 256 threads, FOUR
// resident per CU (16 waves, 128 VGPRs each).  Until late in round 4: 4096 sites, 512 threads, two per CU -- the same waves and
// the same iterations per wave and tile, but four workgroups' load / barrier / write-out phases interleave more evenly than
// two's: 2 % faster on every workload, both paths (profiles/r04_kernel_experiments.txt, section 10).
#ifndef MIDAS_TILE_SHIFT
#define MIDAS_TILE_SHIFT 11
#endif
#ifndef MIDAS_PILEUP_BLOCK
#define MIDAS_PILEUP_BLOCK 256
#endif
#ifndef MIDAS_WORKGROUPS_PER_CU
#define MIDAS_WORKGROUPS_PER_CU 4
#endif
constexpr int kTileShift = MIDAS_TILE_SHIFT;
constexpr int kTileSites = 1 << kTileShift;
constexpr int kPileupBlock = MIDAS_PILEUP_BLOCK;          // 4 waves
constexpr int kWorkgroupsPerCU = MIDAS_WORKGROUPS_PER_CU;
constexpr int kIndexBlock = 256;

// Per-species counters, same order as MIDAS_SNPS_STAT_* in include/midas_snps.h.
constexpr int MIDAS_STATS = 4;
constexpr int MIDAS_STAT_ALIGNED = 0;
constexpr int MIDAS_STAT_MAPPED = 1;
constexpr int MIDAS_STAT_COVERED = 2;
constexpr int MIDAS_STAT_DEPTH = 3;

struct IndexParams {
  const ReadRec* rec;
  const uint8_t* blob;
  const uint32_t* key;               // [n_reads] tile << 7 | reach << 2 | class, from the packer
  const Tile* tiles;
  uint32_t* rbinv;                   // [3*n_tiles] slots 3t (S), 3t+1 (G), 3t+2 (I), see index_reads.hip; max of (n_reads - index), 0 = none
  uint32_t* rend;                    // [3*n_tiles] max of (index + 1)
  uint32_t* rbinv_next;              // the other parity's ranges: zeroed here for the next run
  uint32_t* rend_next;
  int32_t n_tiles;
  unsigned long long* stats;         // [n_species][4]  zeroed here, accumulated by the pileup kernel
  unsigned long long* err;           // set to kNoError here
  int32_t n_reads;
  int32_t n_stat_words;              // n_species * 4
  int32_t tile_len;                  // sites per tile of this batch (<= kTileSites)
  int32_t lane_bases;                // bases per lane of the blob layout (31 or 32)
};

// keep_read's two ratio tests as exact integer thresholds, built on the host per threshold set
// (snps_abi.hip: build_filter_tables) by evaluating the reference's own fp64 expressions:
//   min_match[a] = least x with !(100*x/float(a) < mapid)   -> keep iff (align_len - NM) >= min_match[align_len]
//   min_align[l] = least a with !(a/float(l) < aln_cov)     -> keep iff align_len >= min_align[l_seq]
struct FilterTables {
  int32_t min_match[kMaxLSeq + 1];
  int32_t min_align[kMaxLSeq + 1];
};

// host: tabulate the two thresholds for lengths 1..max_l (snps_abi.hip)
void build_filter_tables(double mapid, double aln_cov, int32_t max_l, FilterTables* t);

constexpr int kSchedGroups = 8;                      // XCDs
constexpr int kSchedWords = 32 * (kSchedGroups + 1);

struct PileupParams {
  const ReadRec* rec;
  const uint8_t* blob;
  const uint8_t* ref;
  const Tile* tiles;
  const uint32_t* rbinv;             // this run's parity, read-only here
  const uint32_t* rend;
  const FilterTables* filt;
  const uint32_t* orig;              // device record -> input index (read only when a record errs)
  uint32_t* out_counts;              // [n_sites][4]
  uint8_t* out_allele;               // [n_sites] or nullptr
  unsigned long long* stats;         // [n_species][4]
  unsigned long long* err;           // one word, atomicMin((read << 8) | kind)
  const uint32_t* items;             // [n_items][4] work items {tile, part, n_parts, 0}: a tile whose reads were split
                                     // into n_parts > 1 slices (hot spots) is accumulated with global atomics
  uint32_t* split_ticket;            // [n_tiles] arrival counter of a split tile's parts (self-resetting), then twice
                                     // kSchedWords: {8 item counters, workgroups done}, one cache line each, of the
                                     // whole-tile and of the parts launch
  // streaming kernel (pileup_stream.hip, developer variant): workgroup b owns tiles [wg_begin[b], wg_begin[b + 1])
  const uint32_t* wg_begin;          // [n_stream_wgs + 1]
  const uint8_t* tile_split;         // [n_tiles] 1 = the tile is processed as parts by the phased kernel; nullptr = none is
  int32_t n_stream_wgs;              // one per CU
  int32_t n_items;
  int32_t n_whole_items;             // items [0, n_whole_items) are whole tiles (n_parts == 1), the rest parts of split tiles
  int32_t n_tiles;
  int32_t n_reads;
  int32_t grid_blocks;               // persistent workgroups: 2 per CU
  int32_t lanes_per_read;            // ceil(max_l_seq / lane_bases)
  int32_t lane_bases;                // bases per lane of the blob layout (31 or 32, layout.h)
  int32_t reads_per_wave;            // 64 / lanes_per_read
  int32_t table_len;                 // entries of the filter tables in use (max_l_seq + 1)
  int32_t baseq, mapq, readq;
  int32_t pad_advances;              // the CIGAR op P advances the query position (MIDAS_SNPS_PAD_PYSAM)
};

// ---- device packer (pack_reads.hip): BAM-native SoA resident in HBM -> rec / blob / orig / key of layout.h -----------
constexpr int kPackBinsPerTile = 10;     // sort bins of a tile: segment records by first site mod 8, then records with a
                                         // CIGAR that stay inside the tile, then records reaching into a later tile
constexpr unsigned kPackBadLayout = 1, kPackUnsupported = 2;   // low byte of PackFacts::status

constexpr int kPackFactSlots = 64;       // workgroup b adds to slot b % 64 (own cache line each): 4096 atomics on ONE word take
                                         // ~50 us, the kernels they sit in ~100; the host adds the slots up
struct alignas(128) PackFacts {          // reductions of one pack, device resident (zeroed by launch_pack_plan)
  unsigned long long status;             // min((read << 8) | kPack*), kNoError when every read is well-formed
  unsigned long long alg_bytes;          // sum(ceil(l/2) + l + 4*n_cigar + 16)
  unsigned long long blob_bytes;         // payload bytes of all records
  unsigned long long n_records;          // device records of all reads (the u32 scan behind it may have wrapped: checked)
  uint32_t max_l;                        // longest read
  uint32_t high_qual;                    // a record holds a quality above kMaxPackedQual (set by the scatter kernel, slot 0)
};

struct PackParams {
  // the ABI's midas_snps_reads, uploaded as it is
  const int32_t* pos; const uint8_t* mapq; const int32_t* nm; const int32_t* l_seq;
  const int64_t* seq_off; const int64_t* qual_off; const int64_t* cigar_off;
  const uint8_t* seq4; const uint8_t* qual; const uint32_t* cigar;
  int64_t seq_bytes, qual_bytes, n_cigar;   // array sizes (the last CSR offsets)
  int32_t n_reads;
  // contig / tile geometry
  const int32_t* contig_read_begin;      // [n_contigs + 1]
  const int32_t* contig_tile_base;       // [n_contigs + 1]
  const int32_t* contig_len;             // [n_contigs]
  int32_t n_contigs, n_tiles, tile_len, tile_shift;   // tile_len == 1 << tile_shift
  int32_t lane_bases, lanes_per_read;    // known after the plan step (longest read)
  // per read
  uint8_t* nseg;                         // [n_reads] 0 = one record that keeps its CIGAR, k = k segment records
  uint32_t* cnt;                         // [n_reads + 1] records per read
  uint32_t* first;                       // [n_reads + 1] first record of a read (input order), [n_reads] = n_records
  // per record, input order
  uint32_t* sort_key; uint32_t* sort_val; uint32_t* bytes8; uint32_t* dest;
  uint4* desc;                           // 32-byte record descriptor, two words of 16 (pack_reads.hip make_desc)
  // per record, sorted / device order
  uint32_t* key_sorted; uint32_t* val_sorted;
  uint32_t* bin_start;                   // [n_tiles * kPackBinsPerTile + 1]
  uint32_t* bytes8_dev; uint32_t* off8;  // [n_records + 1]
  uint32_t* tile_extra; uint32_t* tile_reads;   // [n_tiles] records reaching in / all records a tile will see
  PackFacts* facts;                      // [kPackFactSlots]
  int32_t n_records;
  int32_t pad_advances;                  // the CIGAR op P advances the query position (MIDAS_SNPS_PAD_PYSAM)
  // outputs
  ReadRec* rec; uint8_t* blob; uint32_t* orig; uint32_t* key_out;
};

size_t pack_sort_temp_bytes(int64_t max_records, int key_bits);
int pack_key_bits(int32_t n_tiles);
hipError_t launch_pack_plan(const PackParams& p, void* tmp, size_t tmp_bytes, hipStream_t s);      // nseg, first, facts
hipError_t launch_pack_keys(const PackParams& p, hipStream_t s);                                   // sort keys, sizes
hipError_t launch_pack_order(const PackParams& p, void* tmp, size_t tmp_bytes, int key_bits, hipStream_t s);   // dest, off8, tile_reads
hipError_t launch_pack_scatter(const PackParams& p, hipStream_t s);                                // rec, blob, orig, key


// ---- direct path (index_direct.hip + pileup_direct.hip): the pileup kernel reads the BAM-native arrays themselves ------
// One visit per read.  A ranges pass over the positions alone (4 bytes per read) leaves, per tile, the run of read indices
// that can touch it; the pileup kernel fetches those reads' columns (pos, l_seq, NM, mapq, the CSR offsets), then their
// bases and the first four CIGAR ops, decides in registers whether the CIGAR is one or two gap-free match runs
// (direct_common.h: ReadShape -- everything with at most one indel) and tallies it; any other read (several indels, pads,
// odd clips, no NM / SEQ, a start off the contig ...) is walked op by op where it lies.
// Unsorted input only widens the ranges (slower, never wrong); the host falls back to the packed path when the ranges of a
// batch add up to much more than its reads.
constexpr uint32_t kGenIdle = 0x80;               // flag byte of a lane without a read
// Chunks of the direct pileup kernel (pileup_direct.hip "Work items"): a workgroup takes kDirectChunkTiles consecutive tiles at
// a time and carries what a tile's reads add behind its last site -- at most kDirectOverhang sites -- over to the next tile.
#ifndef MIDAS_DIRECT_CHUNK
#define MIDAS_DIRECT_CHUNK 4
#endif
constexpr int kDirectChunkTiles = MIDAS_DIRECT_CHUNK;
#ifndef MIDAS_DIRECT_TAIL_DIV
#define MIDAS_DIRECT_TAIL_DIV 8
#endif
constexpr int kDirectTailDiv = MIDAS_DIRECT_TAIL_DIV;      // the last 1 / kDirectTailDiv of the tiles are dealt one by one
constexpr int kDirectOverhang = 160;
// ... and for batches whose reads are longer than that (250 bp reads): a second instantiation of the kernel with this overhang --
// 2 KiB more LDS a workgroup, three workgroups a CU instead of four -- chosen per batch from its longest read (DirectParams::overhang)
constexpr int kDirectOverhangLong = 288;

constexpr int kDirectFactSlots = 64;
struct alignas(128) DirectFacts {                 // per-slot partial results of the facts pass (batch_create), added up by the host
  unsigned long long status;                      // min((read << 8) | kPack*), kNoError when every read is well-formed
  unsigned long long alg_bytes;                   // sum(ceil(l/2) + l + 4*n_cigar + 16)
  uint32_t n_general;                             // reads the pileup kernel walks op by op
  uint32_t max_l;                                 // longest read
  uint32_t max_span;                              // longest reference span (sum of M/=/X/D/N lengths) of a read
  uint32_t unsorted;                              // some contig's reads are not in position order
  uint32_t n_long;                                // reads beyond the fast paths' limits (l_seq > kMaxLSeq, n_cigar / NM > kMaxField16): the batch takes the long path
  uint32_t n_outliers;                            // reads whose reference span exceeds kDirectOverhang (a long deletion, an N skip) ...
  uint32_t max_span_common;                       // ... and the longest span of all the others
};
// A read that spans more than the overhang: the tiles behind the one it starts in, [t_first, t_last], must find it in their
// streams although the ranges pass reaches back over the COMMON span only (direct_outliers_kernel, every pass).
struct DirectOutlier { uint32_t read, t_first, t_last, pad; };

// the contig a workgroup of the direct path's index kernels starts in (index_direct.hip ContigCursor, as the facts pass left it)
struct DirectBlockCursor { int32_t c, begin, next_begin, tile_base, tile_end, pad; long long clen; };
struct DirectIndexParams {
  const int32_t* pos; const int32_t* nm; const int32_t* l_seq;
  const int64_t* seq_off; const int64_t* qual_off; const int64_t* cigar_off;
  const uint32_t* cigar;
  const DirectRec* rec; const uint8_t* payload;   // (facts pass of a resident batch, which has no CIGAR column: a read's ops open its payload run; else nullptr)
  DirectOutlier* outliers; uint32_t* n_outliers_listed; uint32_t outlier_cap;    // facts pass: the list (entries beyond the cap are counted, not listed)
  uint8_t* tile_flag;                             // facts pass: [n_tiles] 1 = an outlier leaves its tile's overhang here or reaches in: the tile's chunk is dealt tile by tile
  int64_t seq_bytes, qual_bytes, n_cigar;
  int32_t n_reads;
  const int32_t* contig_read_begin; const int32_t* contig_tile_base; const int32_t* contig_len;
  int32_t n_contigs, n_tiles, tile_shift;
  uint32_t* tbegin; uint32_t* tend;               // [n_tiles] this run's parity: [first, last + 1) of the reads that can touch a tile
  uint32_t* tbegin_next; uint32_t* tend_next;     // the other parity, reset here for the next run
  int32_t sorted;                                 // the facts pass found every contig's reads in position order
  int32_t reach;                                  // the longest reference span of the batch's reads: no read touches a site further from its start
  int32_t overhang = kDirectOverhang;             // facts pass: a read that spans more is an outlier (the batch's: kDirectOverhang or kDirectOverhangLong)
  DirectFacts* facts;                             // [kDirectFactSlots] (facts pass only)
  DirectBlockCursor* block_contig;                // [direct_index_blocks(n_reads)] the contig of a workgroup's first read and that contig's
                                                  // row of the contig tables: found by the facts pass (one binary search per workgroup), ONE
                                                  // 32-byte load of every ranges pass
  unsigned long long* stats; unsigned long long* err;
  int32_t n_stat_words;
};

// the direct layout (layout.h DirectRec + payload), built once per batch from the caller's arrays
struct DirectLayoutParams {
  const int32_t* pos; const uint8_t* mapq; const int32_t* nm; const int32_t* l_seq;
  const int64_t* seq_off; const int64_t* qual_off; const int64_t* cigar_off;
  const uint8_t* seq4; const uint8_t* qual; const uint32_t* cigar;
  int64_t n_reads;
  unsigned long long* block_units;                // [direct_index_blocks(n_reads) + 1] payload units per workgroup's reads, then their exclusive scan
  DirectRec* rec;                                 // [n_reads + 1] (the last one a sentinel: off8 = all units)
  uint8_t* payload;
};

struct DirectParams {
  const DirectRec* rec;                           // [n_reads + 1]
  const uint8_t* payload;                         // per read [cigar][seq][qual], 8-byte aligned (64 bytes of slack behind the last)
  // (the caller's arrays: read by developer variants of the kernel only)
  const int32_t* pos; const uint8_t* mapq; const int32_t* nm; const int32_t* l_seq;
  const int64_t* seq_off; const int64_t* qual_off; const int64_t* cigar_off;
  const uint8_t* seq4; const uint8_t* qual; const uint32_t* cigar;
  const uint32_t* tbegin; const uint32_t* tend;
  const uint8_t* ref;
  const Tile* tiles;
  const FilterTables* filt;
  uint32_t* out_counts; uint8_t* out_allele;
  unsigned long long* stats; unsigned long long* err;
  uint32_t* sched;                                // kSchedWords: {8 item counters, workgroups done}
  unsigned long long* probe;                      // developer builds (MIDAS_SNPS_DEBUG_BITS & 256): per-wave cycle counts
  int32_t n_tiles, n_reads, grid_blocks;
  int32_t lanes_per_read, reads_per_wave, table_len;
  int32_t baseq, mapq_min, readq;
  int32_t pad_advances;                           // the CIGAR op P advances the query position (MIDAS_SNPS_PAD_PYSAM)
  int32_t chunk_tiles, n_chunked_tiles;           // tiles [0, n_chunked_tiles) are dealt in chunks of chunk_tiles (1: every tile by itself)
  int32_t overhang = kDirectOverhang;             // which instantiation: the sites behind a tile's last that the tallies hold
  const uint8_t* chunk_ok;                        // nullptr, or per chunk: 0 = its tiles are piled up one by one (an outlier read), no overhang carried
};

// device_sort.hip: the library's own exclusive scan of 32-bit counters and stable 8-bit-digit radix sort of (key, value) pairs
size_t scan_scratch_words(long long n);                 // 32-bit words of `sums` a scan of n counters needs
hipError_t launch_scan_u32(const uint32_t* in, uint32_t* out, long long n, uint32_t* sums, hipStream_t s);     // in == out allowed
size_t sort_scratch_words(long long n);                 // 32-bit words of `scratch` a sort of n pairs needs
// keys below 2^bits; the sorted pairs end up in (*key_sorted, *val_sorted) = one of the two buffer pairs
hipError_t launch_sort_pairs_u32(uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, long long n, int bits, uint32_t* scratch,
                                 hipStream_t s, uint32_t** key_sorted, uint32_t** val_sorted);
hipError_t launch_sort_pairs_f64(uint32_t* key_a, double* val_a, uint32_t* key_b, double* val_b, long long n, int bits, uint32_t* scratch,
                                 hipStream_t s, uint32_t** key_sorted, double** val_sorted);

// rows_deflate.hip: the table's rows formatted and deflated on the device, one gzip member (<= 16 384 rows of one contig) at a time
struct RowsMember {
  long long site0;                 // the member's first site in the batch's counts / alleles
  long long pos0;                  // ref_pos of its first row (1-based; a piece's origin included)
  int32_t n_rows, id_off, id_len, pad;
};
struct RowsResult {
  unsigned long long off;          // the member's DEFLATE stream in the arena
  uint32_t n_bytes, crc, text_len;
  uint32_t status;                 // 0 done; 1 not taken (contig id too long); 2 the arena was full
};
struct RowsParams {
  const uint32_t* counts; const uint8_t* allele;
  const uint8_t* ids;              // the contig ids, back to back
  const RowsMember* members; int32_t n_members;
  uint8_t* arena; unsigned long long arena_bytes;      // zeroed
  unsigned long long* cursor;      // zeroed: bytes of the arena handed out
  RowsResult* results;
};
hipError_t launch_rows_deflate(const RowsParams& p, int grid_blocks, hipStream_t s);

// bgzf_inflate.hip: raw DEFLATE streams (BGZF blocks) inflated on the device, one thread per stream
struct InflateBlock { unsigned long long cpos, upos, mbase; uint32_t clen, ulen, mcap, pad; };   // mbase, mcap: the stream's room in `matches`
constexpr uint32_t kInflateMatchRoom = 9;      // status: the stream has more matches than its room (decode it again with more)
constexpr uint32_t kInflateCrc = 10;           // status: inflated to the right size, but the bytes' CRC-32 is not the footer's
struct InflateParams {
  const uint8_t* comp;             // the streams (8 bytes of slack behind the last one)
  const InflateBlock* blocks; long long n_blocks;
  uint8_t* out;
  uint32_t* status;                // per stream: 0 = inflated to exactly ulen bytes
  unsigned long long* matches;     // the matches the decoder noted for the resolver
  uint32_t* n_matches;             // per stream
  const uint32_t* want_crc;        // per stream: the CRC-32 its inflated bytes must have (BGZF footer); nullptr: not checked
};
hipError_t launch_bgzf_inflate(const InflateParams& p, hipStream_t s, int phases = 3);
long long bgzf_inflate_wave_blocks(int n_cu);      // blocks that fill the device once with the decoder's workgroups
struct PayloadParams {
  const uint8_t* stream;                   // the inflated BAM
  const unsigned long long* rec_off;       // where every record starts in it
  long long n_records;
  const long long* seq_off; const long long* qual_off; const long long* cigar_off;     // [n_records + 1], elements
  uint8_t* seq4; uint8_t* qual; uint32_t* cigar;
  // (a streamed decode keeps no inflated stream: the records' runs are cut out of the DIRECT layout then -- `stream` is its payload,
  // drec[i].off8 where record i's [cigar][seq][qual] run starts in it, the lengths from the offset columns)
  const DirectRec* drec = nullptr;
};
hipError_t launch_bam_payload(const PayloadParams& p, int grid_blocks, hipStream_t s);    // 1: decode, 2: resolve the matches

// bam_walk.hip: the record walk of an inflated BAM stream that lies in HBM
struct BamWalkParams {
  const uint8_t* d;                  // the inflated bytes: one or several SEGMENTS (runs of consecutive BGZF blocks) back to back
  const long long* ref_lens; int32_t n_ref;
  long long n_chunks;
  // per chunk (a chunk is <= 32 KiB of ONE segment; the host lays them out)
  const unsigned long long* lo;      // first byte of the chunk
  const unsigned long long* hi;      // one past its last byte (records STARTING in [lo, hi) are the chunk's)
  const unsigned long long* stop;    // the segment's stop: records starting at or behind it are not wanted
  const unsigned long long* limit;   // the segment's end: no record may reach past it
  const uint8_t* forced;             // 1: start[c] is given (a known record start, or where the chain stands); 0: guess one, scanning from lo
  unsigned long long* start;         // where the chunk's walk started; ~0: no boundary found
  unsigned long long* end;           // the first record start at or behind min(hi, stop) the walk reached
  uint32_t* kept;                    // records with refID >= 0 that start in the chunk (from `start` on, below stop)
  uint32_t* unmapped;                // records with refID < 0 among them ...
  unsigned long long* first_unmapped;   // ... and where the first one starts (~0: none)
  uint32_t* bad;                     // 1 = a block_size leaves the segment
};
struct BamColumnsParams {
  const uint8_t* d; const unsigned long long* rec_off; long long n;
  int32_t n_ref;                                    // references of the header: a kept record's refID must lie below it
  int32_t *refid, *pos, *nm, *l_seq; uint8_t* mapq; uint16_t* flag;
  long long *seq_off, *qual_off, *cigar_off;        // n + 1 entries: lengths at [i + 1] here, CSR offsets after the scans
  long long* unit_off;                              // (nullable) n + 1 entries: 8-byte units of the record's direct-layout payload, then their scan
  int32_t* span;                                    // (nullable) reference span: the lengths of the record's M / D / N / = / X ops
  unsigned long long* bad_record;                   // min index of a record whose variable parts overrun its block_size or whose
                                                    // refID names no reference of the header (~0: none)
  // A GROUP of a streamed decode (snps_abi.hip device_decode_stream): the records continue the columns of the groups before it --
  // the offsets' scans start at what those came to (entry [0] of this group = the last entry of the one before).
  long long base[4] = {0, 0, 0, 0};                 // seq_off, qual_off, cigar_off, unit_off
};
// the records of an inflated stream as the direct layout (layout.h): DirectRec[n + 1] + payload, one copy of every record's
// [cigar][seq][qual] run; pos / nm: the columns bam_columns_kernel decoded, unit_off: its scanned payload units
struct BamDirectParams {
  const uint8_t* stream; const unsigned long long* rec_off; long long n_records;
  const int32_t* pos; const int32_t* nm; const long long* unit_off;
  DirectRec* rec; uint8_t* payload;
};
hipError_t launch_bam_direct(const BamDirectParams& p, int grid_blocks, hipStream_t s);
hipError_t launch_bam_walk(const BamWalkParams& p, const long long* list, long long n_list, hipStream_t s);
hipError_t launch_bam_offsets(const BamWalkParams& p, const unsigned long long* base, unsigned long long* rec_off, hipStream_t s);
size_t bam_scan_scratch_bytes(long long n_records);
hipError_t launch_bam_columns(const BamColumnsParams& p, long long* scan_scratch, hipStream_t s);

// ---- long path (pileup_long.hip): batches holding a read beyond the fast paths' limits; one thread per read, global atomics ----
struct LongParams {
  const int32_t* pos; const uint8_t* mapq; const int32_t* nm; const int32_t* l_seq;
  const int64_t* seq_off; const int64_t* qual_off; const int64_t* cigar_off;
  const uint8_t* seq4; const uint8_t* qual; const uint32_t* cigar;
  long long n_reads, n_sites;
  const int32_t* contig_read_begin; const int32_t* contig_tile_base;
  int32_t n_contigs, n_tiles;
  const Tile* tiles;
  const uint8_t* ref;
  uint32_t* out_counts; uint8_t* out_allele;
  unsigned long long* stats; unsigned long long* err;
  int32_t n_stat_words;
  int32_t baseq, mapq_min, readq, pad_advances;
  double mapid, aln_cov;
};
hipError_t launch_pileup_long(const LongParams& p, hipStream_t s);

hipError_t launch_direct_facts(const DirectIndexParams& p, hipStream_t s);       // once per batch: validation, totals
hipError_t launch_direct_layout_sizes(const DirectLayoutParams& p, hipStream_t s);   // once per batch: payload units per workgroup + their scan
hipError_t launch_direct_layout_fill(const DirectLayoutParams& p, hipStream_t s);    // once per batch: records + payload
hipError_t launch_direct_ranges(const DirectIndexParams& p, hipStream_t s);      // every pass: the tile ranges, from the positions
hipError_t launch_direct_outliers(const DirectOutlier* list, uint32_t n, uint32_t* tbegin, hipStream_t s);   // every pass behind it, when the batch has outliers
hipError_t launch_pileup_direct(const DirectParams& p, int lane_bases, hipStream_t s);
int direct_lane_bases(int32_t max_l_seq);
int direct_index_blocks(int64_t n_reads);

hipError_t launch_index_reads(const IndexParams& p, hipStream_t stream);
hipError_t launch_pileup_tiles(const PileupParams& p, hipStream_t stream, bool whole_tiles, bool parts);   // barrier-phased
hipError_t launch_pileup_stream(const PileupParams& p, hipStream_t stream);   // whole tiles, barrier-free

}  // namespace midas
