// Device data layout of the MI355X SNP pileup, shared by the host packer and the kernels.
//
// HBM layout of one resident batch:
//
//   rec   [n_reads+1] 16 B   fixed part of a BAM record (one dwordx4 load per read); the extra
//                            last record is a sentinel whose blob_off8 is the end of the payload
//   blob  [...]              per record, 8-byte aligned, records back to back in device order
//                            (160 B for a 150 bp single-match read):
//                              bases  ONE byte per base, 32 per lane chunk: (min(qual, 62) + 1) << 2 | code with
//                                     code A=0 C=1 G=2 T=3; a base that is not A/C/G/T, the padding slot of a 31-base
//                                     lane and the slots past the end of the record are 0.  A base counts iff its byte
//                                     is >= 4 * (baseq + 1) -- ONE unsigned byte compare, exact for every baseq <= 62
//                                     (a batch that holds a quality above 62 refuses a baseq above 62: kMaxPackedQual);
//                                     baseq <= 0 compares against 4: every A/C/G/T base, none of the zeros.
//                                     (the whole-read mean the readq filter needs travels in the record, see ReadRec)
//                              cigar  n_cigar * u32 -- omitted when the record is kRecSimple
//                            coordinate-sorted input => the reads of a tile are one contiguous
//                            byte range of `blob`.
//   ref   [n_sites]   1 B    FASTA letters, contigs back to back
//   tiles [n_tiles]   32 B   {contig, start, len, species, site_base}
//   out counts [n_sites][4] u32 (A,C,G,T) ; out allele [n_sites] u8
//
// A pileup lane owns 31 (or 32, see lane_bases_for) consecutive bases of a read in kChunk = 32 slots of one byte (two
// dwordx4); slot 31 and the slots past the end of the record are padding.  The zero padding is self-masking: a zero byte
// never reaches a threshold >= 4.  (Round 1 and the first half of round 2 kept 32 quality bytes + 16 bytes of 4-bit call
// codes per lane: 48 B.  The kernel is HBM-bound and its reads are two thirds of its traffic, so a third less payload
// is worth more than the shift-and-mask per four bases that recovers the counter offsets.)
//
// Algorithmic bytes: ceil(l/2) + l + 4*n_cigar + 16 per read, 17 per site.
#pragma once
#include <stdint.h>

namespace midas {

struct ReadRec {            // 16 bytes, 16-byte aligned
  int32_t pos;              // BAM pos (0-based leftmost)
  uint32_t blob_off8;       // payload offset in 8-byte units
  uint16_t l_seq;           // bits 0-10: stored query length (soft clips included, <= kMaxLSeq);
                            // bits 11-15: low five bits of qmean = floor(sum(qual) / l_seq)  (see rec_qmean)
  uint16_t n_cigar;         // record with a CIGAR: number of ops.  kRecSimple record (a match segment): bits 0-9 the
                            // read's l_seq, bits 10-13 the high four bits of its aligned length, bit 14 "first segment
                            // of its read" (the one that counts the read and reports its errors)
  uint16_t nm;              // record with a CIGAR: NM tag, kNmAbsent when the record has none.  kRecSimple record:
                            // bits 0-9 NM, bits 10-15 the low six bits of the read's aligned length
  uint8_t mapq;
  uint8_t flags;            // kRec* bits; bits 4-6: high three bits of qmean
};
// qmean: `np.mean(query_qualities) < readq` (midas/run/snps.py:154) with an integer readq is exactly
// floor(sum / l) < readq, so eight bits per read replace a reduction over its quality bytes on the device.
constexpr int kRecLBits = 11;
static_assert(sizeof(ReadRec) == 16, "ReadRec must be 16 bytes");

constexpr uint16_t kNmAbsent = 0xFFFF;
// Record flag bits: decode-time facts about the record, set by the packer.
constexpr uint8_t kRecQualAbsent = 1;   // qual[0] == 0xFF (BAM: QUAL missing)
constexpr uint8_t kRecSimple = 2;       // the record is ONE gap-free match segment of its read (pos = the segment's first
                                        // site, l_seq = its length, payload = its bases): the walk is the identity
constexpr uint8_t kRecClipGeneric = 4;  // clip structure needs the general H/S loops (an H among the clips, or
                                        // several S at one end); when clear: lead = (op0 == S), trail = (opLast == S)
constexpr uint8_t kRecSentinel = 0x80; // the record after the last read (l_seq 0): stream positions past a tile's reads
constexpr uint8_t kRecOverrun = 8;      // some match op maps a query position >= l_seq onto a site inside the
                                        // contig: pysam would index past SEQ (IndexError) if the read is kept

constexpr int kChunk = 32;          // payload slots (= bytes) per lane
constexpr int kMaxPackedQual = 62;  // qualities above it are stored as 62: exact for every base-quality threshold <= 62
// Bases per lane: 31 or 32, fixed per batch (lane_bases_for).  31: the last slot of every lane is padding (quality 0,
// never counted).  The lanes of a read then start 31 sites apart, i.e. 124 dwords apart in the [site][A,C,G,T] tallies
// -- 4 banks short of a multiple of the bank count -- so their LDS atomics fall into different banks; 32 sites apart
// they all hit the same four banks (measured with 32 on 150 bp reads: 57 % of the LDS cycles were bank conflicts, kernel
// +6 %).  32 is kept for the read lengths where it saves a whole lane per read and a read has few lanes to collide
// (125 bp: 4 lanes instead of 5, measured 9 % faster; 250 bp: 8 instead of 9, measured 5 % slower, so 31 there).
__host__ __device__ inline uint32_t blob_chunks(uint32_t l_seq, uint32_t bases) {
  return bases == 32u ? (l_seq + 31u) >> 5 : (l_seq + 30u) / 31u;
}
inline int lane_bases_for(int32_t max_l_seq) {
  const uint32_t l = max_l_seq > 0 ? (uint32_t)max_l_seq : 1u;
  const uint32_t n32 = blob_chunks(l, 32u), n31 = blob_chunks(l, 31u);
  return (n32 < n31 && n32 <= 5u) ? 32 : 31;
}
constexpr int kMaxLSeq = 1024;      // at most 32 lanes per read
constexpr int kMaxField16 = 65534;  // l_seq / n_cigar / NM representable in the record
constexpr int kMaxSegments = 6;     // match segments a read may be served as (more: it keeps its CIGAR)
constexpr int kMaxPieces = 12;      // device records of one read: its segments, each cut at the tile boundaries it crosses
constexpr int kMaxSegField = 1023;  // l_seq / aligned length / NM representable in a segment record (10 bits each)

// payload byte of one base (0 = does not count at any threshold)
__host__ __device__ inline uint8_t base_byte(uint32_t qual, uint32_t code2) {
  const uint32_t q = qual > (uint32_t)kMaxPackedQual ? (uint32_t)kMaxPackedQual : qual;
  return (uint8_t)(((q + 1u) << 2) | code2);
}
// the byte threshold of a base-quality threshold: a base counts iff byte >= it (256: none does)
__host__ __device__ inline uint32_t base_threshold(int32_t baseq) {
  return baseq < 1 ? 4u : (baseq <= kMaxPackedQual ? 4u * ((uint32_t)baseq + 1u) : 256u);
}

struct Tile {               // 32 bytes
  int32_t contig;
  int32_t start;            // first site of the tile, contig coordinates
  int32_t len;              // sites in the tile (<= tile_sites)
  int32_t species;
  int64_t site_base;        // index of `start` in the concatenated site space
  int32_t contig_len;
  int32_t halo;             // 1: the contig entry is a piece with origin > 0 -- a read with pos < 0 belongs to the piece before it
};
static_assert(sizeof(Tile) == 32, "Tile must be 32 bytes");

// Offsets of the payload sections inside a read's blob.
__host__ __device__ inline uint32_t blob_cigar_off(uint32_t l_seq, uint32_t bases) { return blob_chunks(l_seq, bases) * 32u; }
__host__ __device__ inline uint32_t blob_bytes(uint32_t l_seq, uint32_t n_cigar_stored, uint32_t bases) {
  return (blob_cigar_off(l_seq, bases) + 4u * n_cigar_stored + 7u) & ~7u;
}

// ---- direct path (index_direct.hip + pileup_direct.hip) ------------------------------------------------------------------
// One 16-byte record per read -- the 16 fixed bytes SURVEY 8(d) counts per read -- and the read's variable part in BAM order,
// `[cigar 4 * n_cigar][seq ceil(l / 2)][qual l]`, padded to 8 bytes, reads back to back in input order: ONE dwordx4 load
// gives a lane everything it needs to find its bases.  Built once per batch from the caller's arrays (direct_layout_kernel);
// a re-encoding of columns and a gather of bytes, nothing is decided in it (no CIGAR shape, no filter outcome).
struct DirectRec {          // 16 bytes, 16-byte aligned
  int32_t pos;              // BAM pos (0-based leftmost)
  uint32_t l_nc;            // l_seq (bits 0-15) | n_cigar << 16
  uint32_t nmq;             // NM (bits 0-15, kNmAbsent = no tag) | mapq << 16 ; bits 24-31 zero (the kernel's own lane flags)
  uint32_t off8;            // the read's payload, in 8-byte units
};
static_assert(sizeof(DirectRec) == 16, "DirectRec must be 16 bytes");
__host__ __device__ inline uint32_t direct_payload_units(uint32_t l_seq, uint32_t n_cigar) {     // 8-byte units of one read
  return (4u * n_cigar + ((l_seq + 1u) >> 1) + l_seq + 7u) >> 3;
}
constexpr unsigned long long kMaxDirectPayloadUnits = 0xFFFFFFFFull;    // 32 GiB per batch

// Error word written by the kernels: (read_index << 8) | kind, reduced with atomicMin.
constexpr unsigned long long kNoError = ~0ull;

}  // namespace midas
