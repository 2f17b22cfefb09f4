#pragma once
#include "../../include/midas_snps.h"
#include "../../midas_amd/csrc/layout.h"

namespace midas {

struct PackSummary {
  int64_t blob_bytes = 0;
  int64_t n_records = 0;               // device records: one per match segment, or one per read that keeps its CIGAR
  int64_t read_algorithmic_bytes = 0;  // sum(ceil(l/2) + l + 4*n_cigar + 16)
  int32_t max_l_seq = 0;
  int32_t lane_bases = 31;             // bases per lane of this batch's blob layout (layout.h lane_bases_for)
  int32_t has_high_qual = 0;           // a read with QUAL present holds a quality above kMaxPackedQual
};

// rec == blob == nullptr: size query only (blob_bytes, n_records).  rec must hold n_records + 1 records (sentinel);
// orig_index / key_out n_records entries.
// `contigs` (may be nullptr) supplies the contig lengths the kRecOverrun flag is defined against;
// without it every contig is taken as unbounded.
// tile_len > 0 (needs `contigs`): device order = per tile window [simple reads][other reads]; orig_index[j]
// (nullable, n_reads entries) receives the input index of device record j, key_out[j] (nullable) its index key
// tile << 7 | min(reach, 31) << 2 | class (see index_reads.hip).
extern thread_local int g_pad_advances;   // the CIGAR op P advances the query position (set by the entry points from their pad_rule argument)
int32_t pack_reads(const midas_snps_reads* reads, const midas_snps_contigs* contigs, int32_t tile_len, ReadRec* rec,
                   uint8_t* blob, uint32_t* orig_index, uint32_t* key_out, int64_t blob_capacity, PackSummary* out,
                   char* err256);


}  // namespace midas
