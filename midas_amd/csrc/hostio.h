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
// A BAM's bytes are MAPPED for the few places that look into them (the header's blocks, the search for a share's first record),
// but bulk readers do not go through the mapping: every page read through it costs a page-table entry to set up and ~0.1 us to
// tear down again (unmapping a 9 GB BAM that had been read through its mapping: 0.18-0.33 s).  The mappings are registered
// here; a reader that is handed a pointer into one (the device upload's staging copy) asks for the file behind it and preads.
void register_file_mapping(const void* base, size_t size, int fd);
void unregister_file_mapping(const void* base);
bool file_of_mapping(const void* p, size_t n, int* fd, size_t* file_off);
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

// midas_bam_load_device, everything on the device (bam_device.hip): the file's BGZF blocks go up, are inflated and checked
// there, the records are found, their columns decoded and SEQ / QUAL / CIGAR cut out where the stream lies -- what comes down
// is the small columns (refID, pos, mapq, flag, NM, l_seq, the three CSR offset arrays).  hostio.cpp reads the file, builds
// the block table, parses the header (the first blocks, inflated by the host) and hands the device part to `dec`.
struct HostColumns {
  int32_t *refid, *pos, *nm, *l_seq; uint8_t* mapq; uint16_t* flag; int64_t *seq_off, *qual_off, *cigar_off;
  int32_t* span; uint64_t* rec_off;        // (only asked for by the slice walk: reference span and buffer offset of every record)
};
// payload == 2 ("resident"): every column stays where the device decoded it, the records also in the pileup kernel's own
// layout (layout.h DirectRec + payload, one copy of every record's [cigar][seq][qual] run): device addresses, all of them in
// the allocation `dev_owner` stands for.  stream / rec_off: the inflated bytes and every record's start in them, kept so that
// the three payload columns can still be cut (midas_bam_resident_to_columns) for the paths that read them.
struct ResidentReads {
  void* rec = nullptr; uint8_t* payload = nullptr;
  int32_t *refid = nullptr, *pos = nullptr, *nm = nullptr, *l_seq = nullptr; uint8_t* mapq = nullptr; uint16_t* flag = nullptr;
  int64_t *seq_off = nullptr, *qual_off = nullptr, *cigar_off = nullptr, *unit_off = nullptr;
  const uint8_t* stream = nullptr; const uint64_t* rec_off = nullptr;
  int64_t payload_units = 0;
};
struct DeviceDecodeResult {
  int64_t n_records = 0, seq_bytes = 0, qual_bytes = 0, n_cigar = 0;
  void *dev_seq = nullptr, *dev_qual = nullptr, *dev_cigar = nullptr;
  void* dev_owner = nullptr;
  void (*dev_free)(void*) = nullptr;
  ResidentReads resident;       // (payload == 2)
};
// A run of consecutive BGZF blocks of the file, inflated back to back into the decode buffer, and the records wanted from it.
struct DecodeSegment {
  size_t job_lo, job_hi;        // its blocks: jobs [job_lo, job_hi)
  uint64_t from;                // buffer offset of its first record (exact != 0), or of where to start guessing one
  int32_t exact;
  uint64_t stop;                // records that start at or behind this buffer offset are not wanted (the blocks reach a little further)
  // out
  uint64_t first = ~0ull;       // the first record found (~0: none)
  uint64_t end = 0;             // the first record start at or behind `stop` that the chain reached (the segment's end when it ran out)
  int64_t n_records = 0, n_unmapped = 0;
  uint64_t first_unmapped = ~0ull;
};
struct DeviceDecoder {
  void* user;
  // jobs[k]: cpos = offset of block k's DEFLATE stream from comp_base (the mapped / read file), upos = where it inflates to in
  // the decode buffer of buffer_bytes.  payload: 1 = cut SEQ / QUAL / CIGAR and leave them on the device (res->dev_*), 2 = leave
  // EVERYTHING on the device, in the direct layout (res->resident; only refID comes down: alloc may leave the other fields
  // nullptr); extra: also hand out span and rec_off.  alloc(sink, n): host arrays for n records (n + 1 offsets), nullptr fields when out of memory.
  // Statuses as BlockInflater's; MIDAS_SNPS_ERR_UNSUPPORTED: the device could not settle the record boundaries (the caller
  // decodes the host's way).  *bad_record: index of a record that overruns its block_size, -2: a block_size leaves its segment.
  int32_t (*run)(void* user, const uint8_t* comp_base, const InflateJob* jobs, size_t n_jobs, uint64_t buffer_bytes, DecodeSegment* segs,
                 size_t n_segs, const int64_t* ref_lens, int32_t n_ref, int payload, int extra, HostColumns (*alloc)(void* sink, int64_t n),
                 void* sink, DeviceDecodeResult* out, int64_t* bad_job, int64_t* bad_record, char* err256);
};
int32_t bam_decode_on_device(const char* path, const DeviceDecoder* dec, midas_bam** out, int64_t* n_reads, int64_t* seq_bytes,
                             int64_t* qual_bytes, int64_t* n_cigar, char* err256, int payload = 1);
// a handle decoded with payload == 2: its device columns (nullptr: it is not such a handle), record count and array totals
const ResidentReads* bam_resident(const midas_bam* b, int64_t* n_records, int64_t* seq_bytes, int64_t* qual_bytes, int64_t* n_cigar);
// midas_bam_resident_to_columns (snps_abi.hip): the handle's host columns for n records (false: out of memory); then the handle
// is as after midas_bam_load_device -- small columns in host memory, the three payload columns at the given device addresses,
// `owner` (freed with the handle, besides what it holds already) keeping them alive
bool bam_alloc_host_columns(midas_bam* b, int64_t n, HostColumns* c);
void bam_resident_became_columns(midas_bam* b, void* seq4, void* qual, void* cigar, void* owner, void (*free_fn)(void*));
// midas_bam_open_slice / midas_bam_load_ranges with the device doing the work (dec == nullptr: the host's threads, as before):
// the slice's blocks are inflated and walked on the device, the host folds the records' columns into the slice's facts; a
// rank's record ranges are decoded like a whole file, SEQ / QUAL / CIGAR staying on the device
int32_t bam_open_slice_with(const char* path, int32_t slice, int32_t n_slices, const DeviceDecoder* dec, midas_bam** out, char* err256);
int32_t bam_open_share(const char* path, int32_t slice, int32_t n_slices, int64_t max_walk, midas_bam** out, int64_t* out3, char* err256);
int32_t bam_load_ranges_on_device(midas_bam* bam, const DeviceDecoder* dec, int32_t n_ranges, const int64_t* range_begin,
                                  const int64_t* range_end, int64_t* n_reads, int64_t* seq_bytes, int64_t* qual_bytes,
                                  int64_t* n_cigar, char* err256, int payload = 1);

// Members whose DEFLATE streams exist already (the device's row coder): frame them (this library's gzip header with the
// member's size and row count, CRC-32, ISIZE) and write them in order behind the header line's member.
struct CodedMember { const uint8_t* data; uint32_t n_bytes, crc, text_len, rows; };
int32_t write_coded_members(const char* path, bool with_header, int32_t gz_level, int64_t n_members, const CodedMember* members,
                            int32_t threads, char* err256);

}  // namespace midas
