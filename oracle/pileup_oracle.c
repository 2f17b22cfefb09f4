/*
 * pileup_oracle.c -- scalar C restatement of the MIDAS SNP pileup hot path.
 *
 * TEST INFRASTRUCTURE ONLY (the checker and the "port" CPU baseline of bench.py).
 * Nothing under midas_amd/ links, loads or calls this file.
 *
 * PARITY UNPINNED: the arithmetic lives in pysam >= 0.8.1 (unpinned; reference
 * setup.py:15), which is absent from /root/reference and from this image, and the
 * reference's tests hold no golden vectors for this path (test/test_midas.py:98-102
 * checks exit codes only).  This file follows, line by line where possible:
 *
 *   keep_read()              /root/reference/midas/run/snps.py:141-162
 *   count_coverage() walk    call site /root/reference/midas/run/snps.py:194-199
 *                            ([EXT] pysam count_coverage + get_aligned_pairs(matches_only=True))
 *   per-site depth/counters  /root/reference/midas/run/snps.py:201-213
 *   contig.seq = ...upper()  /root/reference/midas/run/snps.py:62
 *
 * and is itself checked against oracle/pileup_oracle.py and the hand-derived cases
 * in tests/golden/kat_cases.json.  It takes the same SoA structs as the C-ABI so the
 * parity tests hand both sides literally the same buffers.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/midas_snps.h"

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

typedef struct {
  int64_t aligned_reads, mapped_reads;
} read_stats;

/* What the op P does to the query position in get_aligned_pairs: 0 = nothing (SAM specification, the default), 1 = it
 * advances (the pysam releases of MIDAS's time put BAM_CPAD in the branch of BAM_CINS / BAM_CSOFT_CLIP).  See
 * midas_snps_set_pad_rule in include/midas_snps.h. */
static int g_pad_advances = 0;
void midas_oracle_set_pad_rule(int rule) { g_pad_advances = rule == 1; }

/* [EXT] pysam getQueryStart */
static int64_t query_alignment_start(const uint32_t* cig, int64_t n) {
  int64_t start = 0;
  for (int64_t k = 0; k < n; ++k) {
    uint32_t op = cig[k] & 15u;
    if (op == OP_H) continue;
    else if (op == OP_S) start += cig[k] >> 4;
    else break;
  }
  return start;
}

/* [EXT] pysam getQueryEnd: backward walk over indices n-1 .. 1 */
static int64_t query_alignment_end(const uint32_t* cig, int64_t n, int64_t l_qseq) {
  int64_t end = l_qseq;
  for (int64_t k = n - 1; k >= 1; --k) {
    uint32_t op = cig[k] & 15u;
    if (op == OP_H) continue;
    else if (op == OP_S) end -= cig[k] >> 4;
    else break;
  }
  return end;
}

/* snps.py:141-162.  Returns 1 keep, 0 reject, <0: -(error kind). */
static int keep_read(const midas_snps_thresholds* a, int64_t l_seq, const uint32_t* cig, int64_t n_cigar,
                     int32_t nm, const uint8_t* qual, int mapq, read_stats* st) {
  st->aligned_reads += 1;
  if (l_seq == 0) return -MIDAS_SNPS_ERR_READ_NO_SEQ;
  int64_t align_len = query_alignment_end(cig, n_cigar, l_seq) - query_alignment_start(cig, n_cigar);
  if (align_len < 0) align_len = 0; /* len(seq[start:end]) */
  int64_t query_len = l_seq;
  if (nm < 0) return -MIDAS_SNPS_ERR_READ_NO_NM;
  if (align_len == 0) return -MIDAS_SNPS_ERR_READ_ZERO_ALIGN;
  if ((double)(100 * (align_len - (int64_t)nm)) / (double)align_len < a->mapid) return 0;
  if (qual[0] == 0xFF) return -MIDAS_SNPS_ERR_READ_NO_QUAL;
  {
    double sum = 0.0; /* np.mean: float64 accumulation of exact small integers */
    for (int64_t i = 0; i < l_seq; ++i) sum += (double)qual[i];
    if (sum / (double)l_seq < (double)a->readq) return 0;
  }
  if (mapq < a->mapq) return 0;
  if ((double)align_len / (double)query_len < a->aln_cov) return 0;
  st->mapped_reads += 1;
  return 1;
}

static inline int base_code(const uint8_t* seq4, int64_t i) {
  uint8_t b = seq4[i >> 1];
  return (i & 1) ? (b & 15) : (b >> 4);
}

/*
 * Same contract as midas_snps_pileup(): out_counts [sum(length)*4] u32 (A,C,G,T per site),
 * out_allele [sum(length)] u8 or NULL, out_stats [n_species*4] i64.  Returns 0 or the
 * MIDAS_SNPS_ERR_READ_* kind; *err_read gets the offending read index (first in input order).
 */
int32_t midas_oracle_pileup(const midas_snps_thresholds* thr, const midas_snps_contigs* contigs,
                            const midas_snps_reads* reads, uint32_t* out_counts, uint8_t* out_allele,
                            int64_t* out_stats, int64_t* err_read) {
  int64_t site0 = 0;
  if (err_read) *err_read = -1;
  memset(out_stats, 0, sizeof(int64_t) * (size_t)contigs->n_species * MIDAS_SNPS_NUM_STATS);
  for (int32_t c = 0; c < contigs->n_contigs; ++c) {
    const int64_t length = contigs->length[c];
    uint32_t* counts = out_counts + site0 * 4;
    int64_t* st = out_stats + (int64_t)contigs->species[c] * MIDAS_SNPS_NUM_STATS;
    read_stats rs = {0, 0};
    memset(counts, 0, sizeof(uint32_t) * 4 * (size_t)length);
    for (int64_t r = contigs->read_begin[c]; r < contigs->read_begin[c + 1]; ++r) {
      const int64_t l_seq = reads->l_seq[r];
      const uint32_t* cig = reads->cigar + reads->cigar_off[r];
      const int64_t n_cigar = reads->cigar_off[r + 1] - reads->cigar_off[r];
      const uint8_t* qual = reads->qual + reads->qual_off[r];
      const uint8_t* seq4 = reads->seq4 + reads->seq_off[r];
      /* a piece of a longer contig (midas_snps_contigs.origin): a read that starts in front of it is the previous piece's --
       * tallied here where it reaches in, counted there; what keep_read raises is reported there, a walk that overruns here, here */
      const int halo = contigs->origin && contigs->origin[c] > 0 && reads->pos[r] < 0;
      read_stats scratch = {0, 0};
      int k = keep_read(thr, l_seq, cig, n_cigar, reads->nm[r], qual, reads->mapq[r], halo ? &scratch : &rs);
      if (k < 0) {
        if (halo) continue;
        if (err_read) *err_read = r;
        return -k;
      }
      if (!k) continue;
      /* get_aligned_pairs(matches_only=True) + the counting body of count_coverage */
      int64_t qpos = 0, rpos = reads->pos[r];
      for (int64_t ci = 0; ci < n_cigar; ++ci) {
        const uint32_t op = cig[ci] & 15u;
        const int64_t len = cig[ci] >> 4;
        if (op == OP_M || op == OP_EQ || op == OP_X) {
          for (int64_t i = 0; i < len; ++i) {
            const int64_t q = qpos + i, refpos = rpos + i;
            if (refpos >= 0 && refpos < length) {
              if (q >= l_seq) {        /* (reported for a halo read too: this is where its walk fails) */
                if (err_read) *err_read = r;
                return MIDAS_SNPS_ERR_READ_CIGAR_OVERRUN;
              }
              if ((thr->baseq != 0 && (int)qual[q] >= thr->baseq) || thr->baseq == 0) {
                switch (base_code(seq4, q)) {
                  case 1: counts[refpos * 4 + 0]++; break; /* 'A' */
                  case 2: counts[refpos * 4 + 1]++; break; /* 'C' */
                  case 4: counts[refpos * 4 + 2]++; break; /* 'G' */
                  case 8: counts[refpos * 4 + 3]++; break; /* 'T' */
                  default: break;                          /* N / IUPAC / '=' count nowhere */
                }
              }
            }
          }
          qpos += len;
          rpos += len;
        } else if (op == OP_I || op == OP_S || (op == OP_P && g_pad_advances)) {
          qpos += len;
        } else if (op == OP_D || op == OP_N) {
          rpos += len;
        } /* H, B, and P under the specification's rule: no effect */
      }
    }
    /* snps.py:201-213 */
    for (int64_t i = 0; i < length; ++i) {
      const int64_t depth = (int64_t)counts[i * 4] + counts[i * 4 + 1] + counts[i * 4 + 2] + counts[i * 4 + 3];
      st[MIDAS_SNPS_STAT_TOTAL_DEPTH] += depth;
      if (depth > 0) st[MIDAS_SNPS_STAT_COVERED_BASES] += 1;
      if (out_allele) {
        uint8_t ch = contigs->ref[site0 + i];
        if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 32);
        out_allele[site0 + i] = ch;
      }
    }
    st[MIDAS_SNPS_STAT_ALIGNED_READS] += rs.aligned_reads;
    st[MIDAS_SNPS_STAT_MAPPED_READS] += rs.mapped_reads;
    site0 += length;
  }
  return MIDAS_SNPS_OK;
}
