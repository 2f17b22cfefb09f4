// Validation of the caller's contig table (midas_snps_contigs), shared by the entry points that take one.
#pragma once
#include "../../include/midas_snps.h"

namespace midas {

int32_t validate_contigs(const midas_snps_contigs* contigs, int64_t n_reads, int64_t* out_sites, char* err256);

}  // namespace midas
