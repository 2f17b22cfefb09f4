#pragma once
#include "../../include/midas_snps.h"
#include <cstddef>

namespace midas {

constexpr int64_t kRowsPerMember = 1 << 14;   // table rows per gzip member

// Rows whose sites are not in host memory yet (the batch's results, still on the device): the writer asks for them slab
// by slab.  source_site[k]: where output contig k starts at the source; a slab is a run of sites that is contiguous
// there and holds whole members.  fetch() fills ring slot `slot` with sites [src_lo, src_lo + n) and hands back host
// pointers to them; it is called from one thread, slot s again only after every row of its previous slab is formatted.
struct RowFeed {
  const int64_t* source_site;
  int64_t slab_sites;          // most sites a slot holds (a multiple of kRowsPerMember)
  int n_slots;
  void* user;
  bool (*fetch)(void* user, int slot, int64_t src_lo, int64_t n, const uint8_t** allele, const uint32_t** counts);
};

int32_t write_rows_fed(const char* path, bool with_header, int32_t n_contigs, const char* const* ref_ids, const int64_t* n_sites,
                       int32_t gz_level, int32_t threads, const RowFeed& feed, char* err256, const int64_t* first_pos = nullptr);

// Many raw DEFLATE streams inflated at once by somebody else than the host's threads (the device: snps_abi.hip).  The
// compressed bytes are given as segments that the streams' cpos count through back to back; upos are offsets into out.
struct InflateJob { uint64_t cpos, upos; uint32_t clen, ulen; uint32_t crc, check_crc; };   // check_crc != 0: the inflated bytes' CRC-32 must be `crc`
struct InflateSegment { const uint8_t* p; size_t n; };
struct BlockInflater {
  void* user;
  // MIDAS_SNPS_OK, MIDAS_SNPS_ERR_BAD_LAYOUT (a stream is corrupt: *bad_job says which) or what went wrong on the way
  int32_t (*run)(void* user, const InflateSegment* segs, size_t n_segs, const InflateJob* jobs, size_t n_jobs, uint8_t* out,
                 size_t out_bytes, int64_t* bad_job, char* err256);
};
}  // namespace midas
struct midas_bam;
namespace midas {
// midas_bam_open / midas_bam_load_ranges with the BGZF blocks inflated by `inflater` (nullptr: the host's threads)
int32_t bam_open_with(const char* path, const BlockInflater* inflater, midas_bam** out, char* err256);
int32_t bam_load_ranges_with(midas_bam* bam, const BlockInflater* inflater, int32_t n_ranges, const int64_t* range_begin,
                             const int64_t* range_end, int64_t* n_reads, int64_t* seq_bytes, int64_t* qual_bytes,
                             int64_t* n_cigar, char* err256);

// midas_bam_load_device (snps_abi.hip): the payload columns are cut on the device
void bam_keep_payload_on_device(midas_bam* b);        // before midas_bam_load: decode everything but SEQ / QUAL / CIGAR
const uint64_t* bam_record_offsets(const midas_bam* b, size_t* n);
void bam_offsets(const midas_bam* b, const int64_t** seq_off, const int64_t** qual_off, const int64_t** cigar_off);
void bam_set_device_payload(midas_bam* b, void* seq4, void* qual, void* cigar, void* owner, void (*free_fn)(void*));   // owner: freed once, at close

// Members whose DEFLATE streams exist already (the device's row coder): frame them (this library's gzip header with the
// member's size and row count, CRC-32, ISIZE) and write them in order behind the header line's member.
struct CodedMember { const uint8_t* data; uint32_t n_bytes, crc, text_len, rows; };
int32_t write_coded_members(const char* path, bool with_header, int32_t gz_level, int64_t n_members, const CodedMember* members,
                            int32_t threads, char* err256);

}  // namespace midas
