// The caller's contig table checked before anything is read through it: malformed input is an error status, never UB.
#include "contigs.h"

#include <cstdio>

namespace midas {

namespace {
void set_err(char* err256, const char* fmt, long long a = 0, long long b = 0, long long c = 0) {
  if (err256) snprintf(err256, 256, fmt, a, b, c);
}
}  // namespace

int32_t validate_contigs(const midas_snps_contigs* c, int64_t n_reads, int64_t* out_sites, char* err256) {
  if (!c || c->n_contigs < 0 || c->n_species < 0) {
    set_err(err256, "bad contig table header");
    return MIDAS_SNPS_ERR_INVALID_ARG;
  }
  if (c->n_contigs > 0 && (!c->length || !c->species || !c->read_begin || !c->ref)) {
    set_err(err256, "NULL array in midas_snps_contigs");
    return MIDAS_SNPS_ERR_INVALID_ARG;
  }
  int64_t sites = 0;
  for (int32_t i = 0; i < c->n_contigs; ++i) {
    if (c->length[i] <= 0) {
      // reference: pysam raises ValueError("interval of size 0") for an empty contig
      set_err(err256, "contig %lld has length %lld (count_coverage: interval of size 0)", i, c->length[i]);
      return MIDAS_SNPS_ERR_UNSUPPORTED;
    }
    if (c->length[i] > 0x7FFFFFFFLL) {
      set_err(err256, "contig %lld longer than 2^31-1 (BAM limit)", i);
      return MIDAS_SNPS_ERR_UNSUPPORTED;
    }
    if (c->species[i] < 0 || c->species[i] >= c->n_species) {
      set_err(err256, "contig %lld: species index %lld out of range", i, c->species[i]);
      return MIDAS_SNPS_ERR_INVALID_ARG;
    }
    if (c->read_begin[i] < 0 || c->read_begin[i + 1] < c->read_begin[i]) {
      set_err(err256, "read_begin not monotone at contig %lld", i);
      return MIDAS_SNPS_ERR_BAD_LAYOUT;
    }
    if (c->origin && (c->origin[i] < 0 || c->origin[i] + c->length[i] > 0x7FFFFFFFLL)) {
      set_err(err256, "contig %lld: piece origin %lld out of range", i, c->origin[i]);
      return MIDAS_SNPS_ERR_INVALID_ARG;
    }
    sites += c->length[i];
  }
  if (c->n_contigs > 0 && (c->read_begin[0] != 0 || c->read_begin[c->n_contigs] != n_reads)) {
    set_err(err256, "read_begin must start at 0 and end at n_reads (%lld)", (long long)n_reads);
    return MIDAS_SNPS_ERR_BAD_LAYOUT;
  }
  if (c->n_contigs == 0 && n_reads != 0) {
    set_err(err256, "reads without contigs");
    return MIDAS_SNPS_ERR_BAD_LAYOUT;
  }
  *out_sites = sites;
  return MIDAS_SNPS_OK;
}

}  // namespace midas
