#pragma once
#include "../../include/midas_snps.h"

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

// Members whose DEFLATE streams exist already (the device's row coder): frame them (this library's gzip header with the
// member's size and row count, CRC-32, ISIZE) and write them in order behind the header line's member.
struct CodedMember { const uint8_t* data; uint32_t n_bytes, crc, text_len, rows; };
int32_t write_coded_members(const char* path, bool with_header, int32_t gz_level, int64_t n_members, const CodedMember* members,
                            int32_t threads, char* err256);

}  // namespace midas
