// TEST INFRASTRUCTURE (not part of the product library): the host mirror of the device packer (midas_amd/csrc/pack_reads.hip).
// BAM-native SoA records -> the device layout of layout.h, computed the plain way on the CPU: the CPU tests pin the layout on
// it, the GPU tests hold the device packer to it bit for bit (tests/test_gpu_pack.py).  Built by tests/mirror/__init__.py
// into tests/mirror/libmidas_snps_mirror.so.
#include "pack_mirror.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../midas_amd/csrc/workers.h"

namespace midas {

thread_local int g_pad_advances = 0;   // the pad rule of the call in progress (every entry point takes it as an argument)

namespace {

int hw_threads() {
  unsigned n = (unsigned)cpu_budget();
  if (n > 64) n = 64;
  return (int)n;
}

template <class F>
void parallel_ranges(int64_t n, F&& fn) {
  int nt = hw_threads();
  if (n < (int64_t)1 << 16) nt = 1;
  if (nt == 1) {
    fn(0, (int64_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  int64_t per = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    int64_t lo = t * per, hi = std::min(n, lo + per);
    if (lo >= hi) break;
    th.emplace_back([&fn, t, lo, hi] { fn(t, lo, hi); });
  }
  for (auto& x : th) x.join();
}

// BAM base code ("=ACMGRSVTWYHKDBN") -> 2-bit code A,C,G,T = 0..3, 4 for anything else (layout.h)
struct CallCodeTable {
  uint8_t v[16];
  constexpr CallCodeTable() : v() {
    for (int b = 0; b < 16; ++b) v[b] = b == 1 ? 0 : b == 2 ? 1 : b == 4 ? 2 : b == 8 ? 3 : 4;
  }
  constexpr uint8_t operator[](uint8_t b) const { return v[b]; }
};
constexpr CallCodeTable kCallCode{};

void set_err(char* err256, const char* fmt, long long a = 0, long long b = 0, long long c = 0) {
  if (err256) snprintf(err256, 256, fmt, a, b, c);
}

inline bool op_match(uint32_t op) { return op == 0u || op == 7u || op == 8u; }  // M = X
inline bool op_clip(uint32_t op) { return op == 4u || op == 5u; }              // S H

// Decode-time facts about one record's CIGAR (layout.h kRec* bits) for a record that keeps its CIGAR: which clip
// path applies, plus the one condition (kRecOverrun) under which the reference would raise IndexError for a kept read.
uint8_t cigar_flags(const uint32_t* cg, uint32_t nc, uint32_t l, int64_t pos, int64_t contig_len, int64_t* reflen) {
  *reflen = l;
  uint8_t f = 0;
  if (nc > 0) {
    // leading clip run: maximal prefix of {S,H}; trailing: maximal suffix of {S,H} within indices >= 1
    uint32_t lead = 0;
    while (lead < nc && op_clip(cg[lead] & 15u)) ++lead;
    uint32_t trail = 0;
    while (trail + 1 < nc && op_clip(cg[nc - 1 - trail] & 15u)) ++trail;
    const bool lead_plain = lead == 0 || (lead == 1 && (cg[0] & 15u) == 4u);
    const bool trail_plain = trail == 0 || (trail == 1 && (cg[nc - 1] & 15u) == 4u);
    if (!lead_plain || !trail_plain) f |= kRecClipGeneric;
  }
  int64_t qpos = 0, rpos = pos;
  for (uint32_t k = 0; k < nc; ++k) {
    const uint32_t op = cg[k] & 15u;
    const int64_t len = cg[k] >> 4;
    if (op_match(op)) {
      if (qpos + len > (int64_t)l) {
        const int64_t qs = std::max<int64_t>(qpos, l);
        const int64_t rs = rpos + (qs - qpos), rend = rpos + len;
        if (rs < contig_len && rend > 0) f |= kRecOverrun;
      }
      qpos += len;
      rpos += len;
    } else if (op == 1u || op == 4u || (op == 6u && g_pad_advances)) {
      qpos += len;
    } else if (op == 2u || op == 3u) {
      rpos += len;
    }
  }
  *reflen = rpos - pos;
  return f;
}

// A read whose CIGAR is `H* S? (M|=|X|I|D|N)+ S? H*` with every length >= 1, consuming exactly l_seq query bases, is
// served to the device as its match segments: records that are each one gap-free run of aligned bases ("simple").
// Clipped and inserted bases are in no segment (they are never tallied); the read-level numbers the filter needs
// travel in every segment's record (layout.h).  Returns the number of segments (1..kMaxSegments) or 0 when the read
// keeps its CIGAR and takes the device's general path (anything else: P/B ops, clips elsewhere, several S at one
// end, zero-length ops, a query length that does not add up, absent or large NM, l_seq = 1024, negative pos, or more
// segments than kMaxSegments).
struct Segment { int32_t qoff; int64_t roff; int32_t len; };
int segment_plan(const uint32_t* cg, uint32_t nc, uint32_t l, int32_t nm, int64_t pos, Segment* segs, uint32_t* align_total) {
  if (l < 1 || l > kMaxSegField || nm < 0 || nm > kMaxSegField || pos < 0 || nc == 0) return 0;
  uint32_t k = 0;
  while (k < nc && (cg[k] & 15u) == 5u) { if ((cg[k] >> 4) == 0) return 0; ++k; }              // H*
  uint32_t lead = 0, trail = 0;
  if (k < nc && (cg[k] & 15u) == 4u) { lead = cg[k] >> 4; if (lead == 0) return 0; ++k; }        // S?
  uint32_t e = nc;
  while (e > k && (cg[e - 1] & 15u) == 5u) { if ((cg[e - 1] >> 4) == 0) return 0; --e; }         // H* at the end
  if (e > k && (cg[e - 1] & 15u) == 4u) { trail = cg[e - 1] >> 4; if (trail == 0) return 0; --e; }
  if (e <= k) return 0;
  // pysam's backward walk never inspects op 0: a trailing clip at index 0 cannot happen here (e > k >= 0 and the
  // ops in [k, e) are not clips), so lead/trail are exactly what getQueryStart/getQueryEnd return
  int64_t q = lead, r = 0;
  int n = 0;
  bool prev_match = false;
  for (uint32_t i = k; i < e; ++i) {
    const uint32_t op = cg[i] & 15u;
    const int64_t len = cg[i] >> 4;
    if (len == 0) return 0;
    if (op_match(op)) {
      if (prev_match) {
        segs[n - 1].len += (int32_t)len;          // "10=1X20=": one gap-free run
      } else {
        if (n == kMaxSegments) return 0;
        segs[n].qoff = (int32_t)q;
        segs[n].roff = r;
        segs[n].len = (int32_t)len;
        ++n;
      }
      q += len;
      r += len;
      prev_match = true;
    } else if (op == 1u) {        // I
      q += len;
      prev_match = false;
    } else if (op == 2u || op == 3u) {   // D N
      r += len;
      prev_match = false;
    } else {
      return 0;                   // S/H in the middle, P, B, unknown
    }
    if (q > (int64_t)l || r > 0x7FFFFFFFll) return 0;
  }
  if (n == 0 || q + trail != (int64_t)l) return 0;
  *align_total = l - lead - trail;
  return n;
}

// The device records of a segmentable read: its match segments, each cut where it crosses a tile boundary (tile_len > 0)
// and clipped to [0, contig_len) -- sites outside the contig are never counted.  A record then lies inside exactly
// one tile, so no segment is ever processed by two tiles or needs an edge mask on the device.  Returns the number of
// pieces, 0 when the read keeps its CIGAR after all (too many pieces, or its first segment has no base inside the
// contig).
int piece_plan(const Segment* segs, int k, int64_t pos, int64_t contig_len, int32_t tile_len, Segment* out) {
  int n = 0;
  for (int s = 0; s < k; ++s) {
    int64_t qo = segs[s].qoff, ro = segs[s].roff, len = segs[s].len;
    int64_t start = pos + ro;
    // clip to the contig
    if (start + len > contig_len) len = contig_len - start;
    if (len <= 0) {
      if (s == 0) return 0;   // the read's first record must exist (it counts the read): leave this one to the general path
      continue;
    }
    if (tile_len > 0) {
      while (len > 0) {
        const int64_t room = tile_len - (start % tile_len);
        const int64_t take = len < room ? len : room;
        if (n == kMaxPieces) return 0;
        out[n++] = Segment{(int32_t)qo, ro, (int32_t)take};
        qo += take; ro += take; start += take; len -= take;
      }
    } else {
      if (n == kMaxPieces) return 0;
      out[n++] = Segment{(int32_t)qo, ro, (int32_t)len};
    }
  }
  return n;
}

}  // namespace

int32_t pack_reads(const midas_snps_reads* r, const midas_snps_contigs* contigs, int32_t tile_len, ReadRec* rec,
                   uint8_t* blob, uint32_t* orig_index, uint32_t* key_out, int64_t blob_capacity, PackSummary* out,
                   char* err256) {
  if (!r || !out) return MIDAS_SNPS_ERR_INVALID_ARG;
  const int64_t n = r->n_reads;
  *out = PackSummary{};
  if (n < 0 || n > 2000000000LL) {
    set_err(err256, "n_reads %lld out of range", (long long)n);
    return n < 0 ? MIDAS_SNPS_ERR_INVALID_ARG : MIDAS_SNPS_ERR_UNSUPPORTED;
  }
  if (n > 0 && (!r->pos || !r->mapq || !r->nm || !r->l_seq || !r->seq_off || !r->qual_off || !r->cigar_off ||
                !r->seq4 || !r->qual || !r->cigar)) {
    set_err(err256, "NULL array in midas_snps_reads");
    return MIDAS_SNPS_ERR_INVALID_ARG;
  }

  // Tiles (only when the caller asked for the tile-ordered device layout): first tile id of every contig.
  const bool tiled = tile_len > 0 && contigs && contigs->n_contigs > 0;
  std::vector<int64_t> tile_base;
  if (tiled) {
    tile_base.assign((size_t)contigs->n_contigs + 1, 0);
    for (int32_t c = 0; c < contigs->n_contigs; ++c)
      tile_base[c + 1] = tile_base[c] + (contigs->length[c] + tile_len - 1) / tile_len;
  }

  // Pass 1: validate every read, decide how many device records it becomes (its match segments, or one record that
  // keeps the CIGAR).
  std::vector<uint8_t> nseg(n);       // 0: one general record; k >= 1: k segment records
  const bool have_contigs = contigs && contigs->n_contigs > 0;
  std::vector<int32_t> contig_of(have_contigs ? n : 0);
  std::atomic<int32_t> status{MIDAS_SNPS_OK};
  std::atomic<long long> bad_read{-1};
  const int nt = hw_threads();
  std::vector<int64_t> alg(nt, 0);
  std::vector<int32_t> maxl(nt, 0);
  auto fail = [&](int32_t st, int64_t i) {
    int32_t ok = MIDAS_SNPS_OK;
    if (status.compare_exchange_strong(ok, st)) bad_read = i;
  };
  parallel_ranges(n, [&](int t, int64_t lo, int64_t hi) {
    int64_t a = 0;
    int32_t ml = 0;
    int32_t c = 0;  // contig of read i (reads are grouped by contig)
    if (contigs && contigs->n_contigs > 0) {
      c = (int32_t)(std::upper_bound(contigs->read_begin, contigs->read_begin + contigs->n_contigs + 1, lo) -
                    contigs->read_begin) - 1;
      c = std::max(0, std::min(c, contigs->n_contigs - 1));
    }
    Segment segs[kMaxSegments], pieces[kMaxPieces];
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t l = r->l_seq[i];
      const int64_t nc = r->cigar_off[i + 1] - r->cigar_off[i];
      const int64_t sb = r->seq_off[i + 1] - r->seq_off[i];
      const int64_t qb = r->qual_off[i + 1] - r->qual_off[i];
      if (l < 0 || nc < 0 || r->cigar_off[i] < 0 || r->seq_off[i] < 0 || r->qual_off[i] < 0 ||
          sb < (l + 1) / 2 || qb < l) {
        fail(MIDAS_SNPS_ERR_BAD_LAYOUT, i);
        return;
      }
      if (l > kMaxLSeq || nc > kMaxField16 || r->nm[i] > kMaxField16) {
        fail(MIDAS_SNPS_ERR_UNSUPPORTED, i);
        return;
      }
      if (contigs && contigs->n_contigs > 0) {
        while (c + 1 < contigs->n_contigs && i >= contigs->read_begin[c + 1]) ++c;
        contig_of[i] = c;
      }
      uint32_t at = 0;
      const int k = segment_plan(r->cigar + r->cigar_off[i], (uint32_t)nc, (uint32_t)l, r->nm[i], r->pos[i], segs, &at);
      const int64_t clen1 = have_contigs ? contigs->length[c] : INT64_MAX;
      nseg[i] = (uint8_t)(k > 0 && r->pos[i] < clen1 ? piece_plan(segs, k, r->pos[i], clen1, tiled ? tile_len : 0, pieces) : 0);
      a += (l + 1) / 2 + l + 4 * nc + 16;
      ml = std::max<int32_t>(ml, (int32_t)l);
    }
    alg[t] += a;
    maxl[t] = std::max(maxl[t], ml);
  });
  if (status != MIDAS_SNPS_OK) {
    if (status == MIDAS_SNPS_ERR_UNSUPPORTED)
      set_err(err256, "read %lld: l_seq > %lld or n_cigar/NM > %lld is not supported", bad_read.load(),
              kMaxLSeq, kMaxField16);
    else
      set_err(err256, "read %lld: negative size or CSR offsets shorter than l_seq", bad_read.load());
    return status;
  }
  for (int t = 0; t < nt; ++t) {
    out->read_algorithmic_bytes += alg[t];
    out->max_l_seq = std::max(out->max_l_seq, maxl[t]);
  }
  out->lane_bases = lane_bases_for(out->max_l_seq);
  const uint32_t lane_bases = (uint32_t)out->lane_bases;
  // records in input order: read i owns records [first[i], first[i + 1])
  std::vector<int64_t> first(n + 1);
  first[0] = 0;
  for (int64_t i = 0; i < n; ++i) first[i + 1] = first[i] + (nseg[i] ? nseg[i] : 1);
  const int64_t m = first[n];
  if (m > 2000000000LL) {
    set_err(err256, "%lld device records exceed the supported range", (long long)m);
    return MIDAS_SNPS_ERR_UNSUPPORTED;
  }
  out->n_records = m;

  // Pass 1b: per record -- payload size, sort key (3 * owner tile + class: 0 = segment inside one tile, 1 = record with
  // a CIGAR inside one tile, 2 = record reaching into a later tile), index key for the device.
  std::vector<uint32_t> rec_read(m);
  std::vector<uint8_t> rec_seg(m);
  std::vector<uint32_t> bytes(m);
  std::vector<uint8_t> cflags(m);
  std::vector<uint8_t> phase(tiled ? m : 0);   // first site of the record modulo 8: its LDS bank phase (see below)
  std::vector<uint32_t> key(tiled ? m : 0);
  std::vector<uint32_t> tile_key_in(tiled && key_out ? m : 0);
  uint32_t* const tile_key = tile_key_in.empty() ? nullptr : tile_key_in.data();
  parallel_ranges(n, [&](int, int64_t lo, int64_t hi) {
    Segment segs[kMaxSegments], pieces[kMaxPieces];
    for (int64_t i = lo; i < hi; ++i) {
      const uint32_t l = (uint32_t)r->l_seq[i];
      const uint32_t nc = (uint32_t)(r->cigar_off[i + 1] - r->cigar_off[i]);
      const uint32_t* cg = r->cigar + r->cigar_off[i];
      const int64_t clen = have_contigs ? contigs->length[contig_of[i]] : INT64_MAX;
      const int64_t tb = tiled ? tile_base[contig_of[i]] : 0;
      auto set_keys = [&](int64_t j, int64_t start, int64_t reflen, bool simple) {
        if (!tiled) return;
        int64_t pc = start < 0 ? 0 : start;
        pc = pc > clen - 1 ? clen - 1 : pc;
        // same arithmetic as index_reads_kernel: first and last tile the record touches
        int64_t pe = start + (reflen > 0 ? reflen : 1) - 1;
        pe = pe < pc ? pc : (pe > clen - 1 ? clen - 1 : pe);
        const int64_t reach = pe / tile_len - pc / tile_len;
        const uint32_t cls = reach > 0 ? 2u : (simple ? 0u : 1u);
        key[j] = (uint32_t)(3 * (tb + pc / tile_len)) + cls;
        phase[j] = (uint8_t)(pc & 7);
        if (tile_key) tile_key[j] = (uint32_t)(((tb + pc / tile_len) << 7) | ((reach > 31 ? 31 : reach) << 2) | cls);
      };
      if (nseg[i] == 0) {
        const int64_t j = first[i];
        int64_t reflen = 0;
        cflags[j] = cigar_flags(cg, nc, l, r->pos[i], clen, &reflen);
        rec_read[j] = (uint32_t)i;
        rec_seg[j] = 0;
        bytes[j] = blob_bytes(l, nc, lane_bases);
        set_keys(j, r->pos[i], reflen, false);
      } else {
        uint32_t at = 0;
        const int k = piece_plan(segs, segment_plan(cg, nc, l, r->nm[i], r->pos[i], segs, &at), r->pos[i], clen,
                                 tiled ? tile_len : 0, pieces);
        for (int s = 0; s < k; ++s) {
          const int64_t j = first[i] + s;
          cflags[j] = kRecSimple;
          rec_read[j] = (uint32_t)i;
          rec_seg[j] = (uint8_t)s;
          bytes[j] = blob_bytes((uint32_t)pieces[s].len, 0u, lane_bases);
          set_keys(j, (int64_t)r->pos[i] + pieces[s].roff, pieces[s].len, true);
        }
      }
    }
  });
  int64_t total = 0;
  for (int64_t j = 0; j < m; ++j) total += bytes[j];
  out->blob_bytes = total;
  if ((uint64_t)total / 8 > 0xFFFFFFFFull) {
    set_err(err256, "packed payload %lld bytes exceeds the 32 GiB a batch can address", (long long)total);
    return MIDAS_SNPS_ERR_UNSUPPORTED;
  }
  if (!rec && !blob) return MIDAS_SNPS_OK;  // size query
  if (!rec || !blob || blob_capacity < total) {
    set_err(err256, "blob capacity %lld < %lld", (long long)blob_capacity, (long long)total);
    return MIDAS_SNPS_ERR_INVALID_ARG;
  }

  // Device order.  Within the window of every tile: the segments that stay inside the tile, then the records with a
  // CIGAR that stay inside it, then the records that reach into a later tile -- each group in input order.  A wave of
  // the pileup kernel then mostly works on records of one kind, and the straddlers a later tile needs are one
  // contiguous run at the end of the window.  A stable counting sort by (owner tile, class); records never leave
  // their contig because tiles do not span contigs.
  std::vector<int64_t> order(m);
  if (tiled) {
    std::vector<int64_t> start((size_t)(3 * tile_base.back()) + 1, 0);
    for (int64_t j = 0; j < m; ++j) start[key[j] + 1]++;
    for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
    for (int64_t j = 0; j < m; ++j) order[start[key[j]]++] = j;
    // Inside a tile's run of segment records the order is free (tallies commute).  Deal them so that consecutive
    // records -- the reads one wave tallies together -- start at different sites modulo 8: a read's lanes occupy five of
    // the eight 4-bank groups of the [site][A,C,G,T] tallies, starting at group (first site mod 8), and reads with the
    // same phase collide bank for bank.
    // (measured: 97.5 -> 95.5 us.)
    {
      std::vector<int64_t> bucket[8];
      int64_t d0 = 0;
      while (d0 < m) {
        const uint32_t k = key[order[d0]];
        int64_t d1 = d0 + 1;
        while (d1 < m && key[order[d1]] == k) ++d1;
        if (k % 3 == 0) {
          for (auto& b : bucket) b.clear();
          for (int64_t d = d0; d < d1; ++d) bucket[phase[order[d]]].push_back(order[d]);
          int64_t d = d0;
          for (size_t round = 0; d < d1; ++round)
            for (int ph = 0; ph < 8; ++ph)
              if (round < bucket[ph].size()) order[d++] = bucket[ph][round];
        }
        d0 = d1;
      }
    }
  } else {
    for (int64_t j = 0; j < m; ++j) order[j] = j;
  }
  // Pass 2: offsets (serial prefix sum in device order), then copy in parallel.
  std::atomic<int> any_high{0};
  std::vector<int64_t> off(m + 1);
  off[0] = 0;
  for (int64_t d = 0; d < m; ++d) off[d + 1] = off[d] + bytes[order[d]];
  parallel_ranges(m, [&](int, int64_t lo, int64_t hi) {
    Segment segs[kMaxSegments], pieces[kMaxPieces];
    for (int64_t d = lo; d < hi; ++d) {
      const int64_t j = order[d];
      const int64_t i = rec_read[j];
      if (orig_index) orig_index[d] = (uint32_t)i;
      if (key_out && tile_key) key_out[d] = tile_key[j];
      const uint32_t l = (uint32_t)r->l_seq[i];
      const uint32_t nc = (uint32_t)(r->cigar_off[i + 1] - r->cigar_off[i]);
      const uint32_t* cg = r->cigar + r->cigar_off[i];
      const uint8_t* q = r->qual + r->qual_off[i];
      const uint8_t* s4 = r->seq4 + r->seq_off[i];
      uint8_t* b = blob + off[d];
      memset(b, 0, bytes[j]);
      uint64_t qsum = 0;
      bool high = false;
      for (uint32_t x = 0; x < l; ++x) { qsum += q[x]; high |= q[x] > (uint8_t)kMaxPackedQual; }
      if (high && !(l > 0 && q[0] == 0xFF)) any_high.store(1, std::memory_order_relaxed);
      const uint32_t qmean = l > 0 ? (uint32_t)(qsum / l) : 0u;   // <= 255, over the WHOLE read (clips included)
      // the bases this record carries: the whole read, or one match segment of it
      uint32_t q0 = 0, len = l;
      int64_t pos = r->pos[i];
      uint32_t at = 0;
      int k = 0;
      if (cflags[j] & kRecSimple) {
        const int64_t clen = have_contigs ? contigs->length[contig_of[i]] : INT64_MAX;
        k = piece_plan(segs, segment_plan(cg, nc, l, r->nm[i], r->pos[i], segs, &at), r->pos[i], clen,
                       tiled ? tile_len : 0, pieces);
        const Segment& sg = pieces[rec_seg[j]];
        q0 = (uint32_t)sg.qoff;
        len = (uint32_t)sg.len;
        pos += sg.roff;
      }
      {
        // one byte per base (layout.h base_byte), 32 slots per lane chunk; 0 for a base that is not A/C/G/T, for the
        // padding slot of a 31-base lane and past the end of the record (the memset above)
        const uint32_t n_chunks = blob_chunks(len, lane_bases);
        for (uint32_t c = 0; c < n_chunks; ++c) {
          for (uint32_t slot = 0; slot < lane_bases; ++slot) {
            const uint32_t x = c * lane_bases + slot;        // base of the record
            if (x >= len) break;
            const uint32_t y = q0 + x;                        // base of the read
            const uint8_t code = kCallCode[(uint8_t)((s4[y >> 1] >> ((~y & 1u) * 4)) & 15u)];
            b[c * kChunk + slot] = code < 4 ? base_byte(q[y], code) : (uint8_t)0;
          }
        }
      }
      ReadRec rr;
      rr.pos = (int32_t)pos;
      rr.blob_off8 = (uint32_t)(off[d] >> 3);
      rr.l_seq = (uint16_t)(len | ((qmean & 31u) << kRecLBits));
      rr.mapq = r->mapq[i];
      const uint8_t qflag = (l > 0 && q[0] == 0xFF) ? kRecQualAbsent : 0;
      if (cflags[j] & kRecSimple) {
        // read-level numbers of the filter, in every segment (layout.h): l_seq, aligned length, NM, "first segment"
        rr.n_cigar = (uint16_t)(l | ((at >> 6) << 10) | ((rec_seg[j] == 0 ? 1u : 0u) << 14));
        rr.nm = (uint16_t)((uint32_t)r->nm[i] | ((at & 63u) << 10));
        (void)k;
      } else {
        memcpy(b + blob_cigar_off(l, lane_bases), cg, 4ull * nc);
        rr.n_cigar = (uint16_t)nc;
        rr.nm = r->nm[i] < 0 ? kNmAbsent : (uint16_t)r->nm[i];
      }
      rr.flags = (uint8_t)(cflags[j] | qflag | ((qmean >> 5) << 4));
      rec[d] = rr;
    }
  });
  out->has_high_qual = any_high.load();
  {
    ReadRec s;  // sentinel: where the payload ends
    memset(&s, 0, sizeof s);
    s.blob_off8 = (uint32_t)(off[m] >> 3);
    s.flags = kRecSentinel;
    rec[m] = s;
  }
  return MIDAS_SNPS_OK;
}



}  // namespace midas

extern "C" {

// pad_rule: MIDAS_SNPS_PAD_SPEC / MIDAS_SNPS_PAD_PYSAM (include/midas_snps.h, midas_snps_set_pad_rule)
int32_t midas_mirror_pack_reads(const midas_snps_reads* reads, const midas_snps_contigs* contigs, int32_t pad_rule, void* rec16, void* blob,
                                int64_t blob_capacity, int64_t* out_blob_bytes, int64_t* out_n_records, int32_t* out_max_l_seq, char* err256) {
  midas::g_pad_advances = pad_rule == MIDAS_SNPS_PAD_PYSAM ? 1 : 0;
  midas::PackSummary s;
  int32_t st = midas::pack_reads(reads, contigs, 0, reinterpret_cast<midas::ReadRec*>(rec16), reinterpret_cast<uint8_t*>(blob), nullptr, nullptr,
                                 blob_capacity, &s, err256);
  if (out_blob_bytes) *out_blob_bytes = s.blob_bytes;
  if (out_n_records) *out_n_records = s.n_records;
  if (out_max_l_seq) *out_max_l_seq = s.max_l_seq;
  return st;
}

// the same in the tile order of a batch (tile_sites = the library's tile: 2048): what batch_create's device packer must produce
int32_t midas_mirror_pack_reads_tiled(const midas_snps_reads* reads, const midas_snps_contigs* contigs, int32_t pad_rule, int32_t tile_sites,
                                      void* rec16, void* blob, int64_t blob_capacity, uint32_t* orig_index, uint32_t* key,
                                      int64_t* out_blob_bytes, int64_t* out_n_records, int32_t* out_max_l_seq, char* err256) {
  if (!contigs) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas::g_pad_advances = pad_rule == MIDAS_SNPS_PAD_PYSAM ? 1 : 0;
  midas::PackSummary s;
  int32_t st = midas::pack_reads(reads, contigs, tile_sites, reinterpret_cast<midas::ReadRec*>(rec16), reinterpret_cast<uint8_t*>(blob),
                                 orig_index, key, blob_capacity, &s, err256);
  if (out_blob_bytes) *out_blob_bytes = s.blob_bytes;
  if (out_n_records) *out_n_records = s.n_records;
  if (out_max_l_seq) *out_max_l_seq = s.max_l_seq;
  return st;
}

}  // extern "C"
