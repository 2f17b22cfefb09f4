// gfx950 (CDNA4) streaming pileup kernel of the MIDAS SNP path: whole tiles, no workgroup barrier.  Integer counting: no MFMA.
//
// Reference semantics implemented here (citations into /root/reference), the same as pileup_tiles.hip:
//   keep_read                       midas/run/snps.py:141-162
//   count_coverage call site        midas/run/snps.py:194-199  ([EXT] pysam: get_aligned_pairs(matches_only),
//                                   qual >= quality_threshold, only 'A','C','G','T' counted)
//   depth / covered / total_depth   midas/run/snps.py:204-213
//   str(rec.seq).upper()            midas/run/snps.py:62
//
// Why a second kernel.  pileup_tiles.hip walks a tile in three barrier-separated phases (stream + tally, write-out,
// re-arm); with ~46 wave-iterations of reads per tile and 8 waves, every tile ends with the waves waiting for the slowest
// of them twice, the iterations of a tile are rounded up to whole rounds of 8, and while a workgroup writes its tile out
// none of its waves streams (measured there: 18 of 96 us on configs[1] are barrier waits).  Here:
//   * ONE 1024-thread workgroup per CU owns a contiguous run of tiles (cut by the host so that every run costs the same)
//     and TWO 64 KiB tally buffers in LDS; non-empty tile number q of the run tallies into buffer q & 1.
//   * The wave-iterations ("units": 12 reads each) of all the run's tiles form ONE sequence, handed out by an LDS counter:
//     a wave always holds three of them (processing one, payload of the next and record of the one after in flight) and
//     draws a new one when it finishes one -- straight across tile boundaries, no rounding per tile, the two-deep
//     register prefetch pipeline never drains, and a wave that is busy writing a tile out simply draws fewer units.
//   * Fifteen waves stream; a wave that finishes a unit bumps the tile's counter in LDS (a returnless atomic).  The
//     SIXTEENTH wave only writes out: it walks the run's tiles in order, copies the tile's upper-cased alleles (and its
//     all-zero rows when no read touches it) while the others tally, then waits for the tile's counter to reach its number
//     of units, writes the tile out alone (64 x {ds_read_b128, re-zero, streaming store}) while the fifteen are already
//     tallying the next tile into the other buffer, and publishes "buffer free".  A wave about to tally tile q + 2 checks
//     that word (it almost never waits).  LDS executes a wave's operations in order, so the counter bump is behind the
//     tallies it counts and the "free" word behind the re-zeroing it announces: no barrier, no fence.  (Letting the wave
//     that finishes a tile write it out was measured first: it still holds two prefetched units of the next tile, which
//     then finish last as well -- a chain through every tile of the run.)
//   * What a wave needs to know about a tile (decoded read ranges, units before it, descriptor) is worked out by the whole
//     workgroup into a 64-byte record in LDS, 320 tiles at a time (a run is walked in segments of 320 tiles with one
//     workgroup barrier between segments -- a few milliseconds of work apart).
// Tiles that hold a coverage hot spot are still cut into parts and merged with atomics by pileup_tiles.hip<SPLIT>.
#include "pileup_common.h"

namespace midas {

using namespace dev;
using namespace pile;

namespace {

// developer timing experiments (tools/build_variant.sh -DMIDAS_STREAM_ABLATE=bits; results are wrong on purpose):
// 1 no tallies, 2 no write-out stores, 4 no chores, 8 no LDS traffic in the write-out
#ifndef MIDAS_STREAM_ABLATE
#define MIDAS_STREAM_ABLATE 0
#endif
constexpr int kAblate = MIDAS_STREAM_ABLATE;
constexpr int kStreamBlock = 1024;
constexpr int kStreamWaves = kStreamBlock / 64;
// Waves that only write tiles out, each a contiguous part of the tile's sites.  One is not enough: a wave may have at
// most 15 LDS operations in flight, every one of them waits its turn behind the tallies of the streaming waves, and a
// tile is 64 reads + 64 re-zeroing writes of 1 KiB -- one wave needs ~10 us per tile (measured), longer than the others
// need to fill the next one.
#ifndef MIDAS_STREAM_DRAIN_WAVES
#define MIDAS_STREAM_DRAIN_WAVES 2
#endif
constexpr int kDrainWaves = MIDAS_STREAM_DRAIN_WAVES;
constexpr int kWorkWaves = kStreamWaves - kDrainWaves;

constexpr int kInfoWords = 16;         // per tile: four 16-byte words, see tile_info below
constexpr int kLdsTiles = 320;         // tiles of a run segment: their info records live in LDS (20 KiB)

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int TILE_SHIFT>
__global__ __launch_bounds__(kStreamBlock, 4) void pileup_stream_kernel(PileupParams p) {
  constexpr int TILE = 1 << TILE_SHIFT;
  constexpr int NW = kChunk / 4;                 // quality words per lane
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][4 * TILE];
  __shared__ uint32_t s_cnt[2];                  // finished units of the tile in buffer b
  __shared__ uint32_t s_epoch[2];                // tiles written out of buffer b so far
  __shared__ uint32_t s_part[2];                 // parts of tiles written out of buffer b so far (kDrainWaves per tile)
  __shared__ uint32_t s_next;                    // next unit of the run to hand out
  __shared__ uint32_t s_units;                   // units in the run
  __shared__ uint32_t s_scan[2][kStreamWaves];   // prologue: per-wave totals of the two prefix sums
  // Per tile, 64 bytes:  w0 = {sb, ib, gb, ns}  w1 = {nsi, total, nit, ubase}  w2 = {q, start, len, species}
  //                      w3 = {site_base lo, hi, contig_len, split}
  // sb..total: the tile's three read ranges as one virtual stream (index_reads.hip); nit: its units of reads (then comes
  // one chore unit); ubase: units of the run before it; q: non-empty tiles of the run before it (buffer = q & 1).
  __shared__ __attribute__((aligned(16))) uint4 s_info[kLdsTiles * 4];
  extern __shared__ __attribute__((aligned(16))) int32_t s_tables[];   // [min_match table_len][min_align table_len]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const ConstWords c_wg = (ConstWords)(size_t)p.wg_begin;
  const int t_begin = (int)c_wg[blockIdx.x], t_end = (int)c_wg[blockIdx.x + 1];
  if (t_begin >= t_end) return;

  const int lpr = p.lanes_per_read;
  const int rpw = p.reads_per_wave;

  const int g = lane / lpr;
  const int c = lane - g * lpr;
  const bool lane_used = g < rpw;
  const int lane_bases = p.lane_bases;           // 31 or 32 (layout.h), uniform
  const int q0 = c * lane_bases;                 // first base of the lane; its payload sits in slot block c
  const uint4* recs = reinterpret_cast<const uint4*>(p.rec);
  const int bq = (int)base_threshold(p.baseq);   // a base counts iff its payload byte >= this (layout.h)
  const uint32_t lds_base0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0][0];

  // ---- once: LDS clean, filter tables in place -----------------------------------------------------------------------------
  {
    uint4* z = reinterpret_cast<uint4*>(&lds[0][0]);
    for (int i = tid; i < 2 * TILE; i += kStreamBlock) z[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < p.table_len; i += kStreamBlock) {
      s_tables[i] = p.filt->min_match[i];
      s_tables[p.table_len + i] = p.filt->min_align[i];
    }
  }
  // per-wave counters; they go to the species row when the wave moves on to another species or leaves
  uint32_t w_aligned = 0, w_mapped = 0;
  int w_species = -1;
  unsigned long long acc_cov = 0ull, acc_depth = 0ull;   // of the tiles this wave wrote out (per lane)
  int acc_species = -1;

  for (int seg_begin = t_begin; seg_begin < t_end; seg_begin += kLdsTiles) {
  const int seg_end = seg_begin + kLdsTiles < t_end ? seg_begin + kLdsTiles : t_end;
  // ---- segment prologue: the segment's tiles decoded into info records (every tally buffer is clean and free here) -----
  {
    if (tid < 2) { s_cnt[tid] = 0u; s_epoch[tid] = 0u; s_part[tid] = 0u; }
    if (tid == 0) s_next = 0u;
    const int t = seg_begin + tid;
    const bool in = tid < kLdsTiles && t < seg_end;
    int sb = 0, ib = 0, gb = 0, ns = 0, nsi = 0, total = 0, nit = 0, split = 0;
    Tile tl;
    if (in) {
      const uint32_t vs = p.rbinv[3 * t], vg = p.rbinv[3 * t + 1], vi = p.rbinv[3 * t + 2];
      const int se = (int)p.rend[3 * t], ge = (int)p.rend[3 * t + 1], ie = (int)p.rend[3 * t + 2];
      sb = vs ? p.n_reads - (int)vs : se;
      gb = vg ? p.n_reads - (int)vg : ge;
      ib = vi ? p.n_reads - (int)vi : ie;
      ns = se - sb;
      nsi = ns + (ie - ib);
      total = nsi + (ge - gb);
      split = p.tile_split ? (int)p.tile_split[t] : 0;
      nit = split ? 0 : (total + rpw - 1) / rpw;
      tl = p.tiles[t];
    }
    // exclusive prefix sums over the segment: units, non-empty tiles
    const uint32_t xu = in ? (uint32_t)nit : 0u, xq = (in && nit > 0) ? 1u : 0u;
    uint32_t su = xu, sq = xq;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t ou = __shfl_up(su, d), oq = __shfl_up(sq, d);
      if (lane >= d) { su += ou; sq += oq; }
    }
    if (lane == 63) { s_scan[0][wave] = su; s_scan[1][wave] = sq; }
    __syncthreads();
    uint32_t bu = 0u, bqn = 0u, tu = 0u;   // (units: a tile's wave-iterations of reads)
    for (int w = 0; w < kStreamWaves; ++w) {
      const uint32_t a = s_scan[0][w], b = s_scan[1][w];
      if (w < wave) { bu += a; bqn += b; }
      tu += a;
    }
    if (in) {
      const uint32_t ubase = bu + su - xu, q = bqn + sq - xq;
      s_info[4 * tid] = make_uint4((uint32_t)sb, (uint32_t)ib, (uint32_t)gb, (uint32_t)ns);
      s_info[4 * tid + 1] = make_uint4((uint32_t)nsi, (uint32_t)total, (uint32_t)nit, ubase);
      s_info[4 * tid + 2] = make_uint4(q, (uint32_t)tl.start, (uint32_t)tl.len, (uint32_t)tl.species);
      s_info[4 * tid + 3] = make_uint4((uint32_t)tl.site_base, (uint32_t)((unsigned long long)tl.site_base >> 32),
                                       (uint32_t)tl.contig_len, (uint32_t)split);
    }
    if (tid == 0) s_units = tu;
  }
  __syncthreads();   // between here and the end of the segment there is no workgroup barrier
  const uint32_t n_units = s_units;

  struct Payload {
    uint32_t qw[NW];  // 32 base bytes: (min(qual, 62) + 1) << 2 | code, 0 = never counts (layout.h)
  };
  // word k of tile t's info record
  auto info = [&](int t, int k) -> uint4 { return s_info[4 * (t - seg_begin) + k]; };
  // virtual stream position of a tile -> record index (n_reads = the sentinel record for positions past the end)
  auto read_at = [&](const uint4& w0, const uint4& w1, int v) -> int {
    const int sb = (int)w0.x, ib = (int)w0.y, gb = (int)w0.z, ns = (int)w0.w, nsi = (int)w1.x, total = (int)w1.y;
    int base = gb - nsi;
    base = v < nsi ? ib - ns : base;
    base = v < ns ? sb : base;
    return (lane_used && v < total) ? base + v : p.n_reads;
  };
  // A unit of the segment: iteration `it` of tile t (t == seg_end: none left).
  struct Pos { int t, it; };
  int cursor = seg_begin;        // tile of the last unit this wave drew (units come in increasing order)
  auto draw = [&]() -> Pos {
    uint32_t u = 0u;
    if (lane == 0) u = __hip_atomic_fetch_add(&s_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    u = (uint32_t)uni((int)u);
    Pos s;
    if (u >= n_units) { s.t = seg_end; s.it = 0; return s; }
    for (;;) {
      const uint4 w1 = info(cursor, 1);
      const uint32_t ubase = (uint32_t)uni((int)w1.w), nit = (uint32_t)uni((int)w1.z);
      if (u < ubase + nit) { s.t = cursor; s.it = (int)(u - ubase); return s; }
      ++cursor;
    }
  };
  auto rec_index = [&](const Pos& s) -> int {
    if (s.t >= seg_end) return p.n_reads;
    const uint4 w0 = info(s.t, 0), w1 = info(s.t, 1);
    return read_at(w0, w1, s.it * rpw + g);
  };
  auto fetch_payload = [&](const uint4& rv, Payload& d) {
    const int l = rec_l(rv);
    const uint8_t* bp = p.blob + (size_t)rec_off8(rv) * 8;
    if (q0 < l) {
      const u32x4_a8 qa = *reinterpret_cast<const u32x4_a8*>(bp + c * kChunk);
      const u32x4_a8 qb = *reinterpret_cast<const u32x4_a8*>(bp + c * kChunk + 16);
      d.qw[0] = qa.x; d.qw[1] = qa.y; d.qw[2] = qa.z; d.qw[3] = qa.w;
      d.qw[4] = qb.x; d.qw[5] = qb.y; d.qw[6] = qb.z; d.qw[7] = qb.w;
    }
  };

  auto flush_reads = [&]() {
    if (lane == 0 && w_species >= 0) {
      if (w_aligned) atomicAdd(&p.stats[(size_t)w_species * MIDAS_STATS + MIDAS_STAT_ALIGNED], (unsigned long long)w_aligned);
      if (w_mapped) atomicAdd(&p.stats[(size_t)w_species * MIDAS_STATS + MIDAS_STAT_MAPPED], (unsigned long long)w_mapped);
    }
    w_aligned = 0;
    w_mapped = 0;
  };
  auto flush_sites = [&]() {
    if (acc_species >= 0) {
      for (int d = 32; d >= 1; d >>= 1) {
        acc_cov += __shfl_down(acc_cov, d);
        acc_depth += __shfl_down(acc_depth, d);
      }
      if (lane == 0) {
        if (acc_cov) atomicAdd(&p.stats[(size_t)acc_species * MIDAS_STATS + MIDAS_STAT_COVERED], acc_cov);
        if (acc_depth) atomicAdd(&p.stats[(size_t)acc_species * MIDAS_STATS + MIDAS_STAT_DEPTH], acc_depth);
      }
    }
    acc_cov = 0ull;
    acc_depth = 0ull;
  };

  // ---- what needs no tallies: upper-cased alleles of the tile's sites [lo, hi), and their rows when no read touches it ---
  auto chores = [&](const uint4& w1, const uint4& w3, int lo, int hi) {
    if (w3.w) return;     // a split tile belongs to the parts kernel, alleles included
    const size_t site_base = (size_t)w3.x | ((size_t)w3.y << 32);
    if (p.out_allele) {
      const uint8_t* ref = p.ref + site_base + lo;
      uint8_t* al = p.out_allele + site_base + lo;
      const int len = hi - lo;
      constexpr int PER = TILE / (kDrainWaves * 64 * 16);      // 16-byte pieces per lane of a full part
      if (len == TILE / kDrainWaves) {     // every load goes out before the first result is looked at
        u32x4_a1 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = *reinterpret_cast<const u32x4_a1*>(ref + 16 * (lane + 64 * k));
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          u32x4_a1 o; o.x = upper4(v[k].x); o.y = upper4(v[k].y); o.z = upper4(v[k].z); o.w = upper4(v[k].w);
          __builtin_nontemporal_store(o, reinterpret_cast<u32x4_a1*>(al + 16 * (lane + 64 * k)));
        }
      } else {               // the last tile of a contig
        for (int i = 4 * lane; i < len; i += 256) {
          if (i + 4 <= len) {
            __builtin_nontemporal_store(upper4(*reinterpret_cast<const u32_a1*>(ref + i)), reinterpret_cast<u32_a1*>(al + i));
          } else {
            for (int j = i; j < len; ++j) {
              uint32_t ch = ref[j];
              if (ch >= 'a' && ch <= 'z') ch -= 32u;
              al[j] = (uint8_t)ch;
            }
          }
        }
      }
    }
    if (w1.z == 0u) {        // no read touches the tile: all-zero rows, nothing covered
      uint4* out = reinterpret_cast<uint4*>(p.out_counts) + site_base;
      u32x4_a8 zv; zv.x = 0u; zv.y = 0u; zv.z = 0u; zv.w = 0u;
      for (int i = lo + lane; i < hi; i += 64) __builtin_nontemporal_store(zv, reinterpret_cast<u32x4_a8*>(out + i));
    }
  };
  // ---- write the sites [lo, hi) of a finished tile out of its buffer, re-zeroing the buffer on the way -------------------
  auto drain = [&](const uint4& w2, const uint4& w3, int b, int lo, int hi) {
    const int species = uni((int)w2.w);
    if (acc_species != species) {
      flush_sites();
      acc_species = species;
    }
    uint4* lds4 = reinterpret_cast<uint4*>(&lds[b][0]);
    uint4* out = reinterpret_cast<uint4*>(p.out_counts) + ((size_t)w3.x | ((size_t)w3.y << 32));
    const int n = (kAblate & 8) ? 0 : hi;
    // The LDS serves requests in arrival order and the streaming waves keep its queue full of tallies: every read that
    // is waited for costs a whole queue delay, so the reads go out eight at a time.
    constexpr int B = 8;
    for (int i0 = lo; i0 < n; i0 += 64 * B) {
      uint4 v[B];
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const int i = i0 + 64 * k + lane;
        v[k] = i < n ? lds4[i] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const int i = i0 + 64 * k + lane;
        if (i < n) {
          lds4[i] = make_uint4(0u, 0u, 0u, 0u);
          u32x4_a8 nv; nv.x = v[k].x; nv.y = v[k].y; nv.z = v[k].z; nv.w = v[k].w;
          if (!(kAblate & 2)) __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_a8*>(out + i));
          const uint32_t d = v[k].x + v[k].y + v[k].z + v[k].w;
          acc_cov += d > 0u ? 1ull : 0ull;
          acc_depth += d;
        }
      }
    }
    // The wave that writes the tile's last part out re-arms the buffer: unit counter to zero, then "free" (LDS executes a
    // wave's operations in order: these words are behind the zeroes above; the other parts' zeroes are behind their bumps).
    if (lane == 0) {
      const uint32_t k = __hip_atomic_fetch_add(&s_part[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((k + 1u) % (uint32_t)kDrainWaves == 0u) {
        __hip_atomic_store(&s_cnt[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&s_epoch[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };

  if (wave >= kWorkWaves) {
    // ---- a write-out wave: tiles in order -- chores while the others tally, then its part of the tile as soon as the tile
    // is complete ---------------------------------------------------------------------------------------------------------
    const int dw = wave - kWorkWaves;
    for (int t = seg_begin; t < seg_end; ++t) {
      const uint4 iw1 = info(t, 1), iw2 = info(t, 2), iw3 = info(t, 3);
      const int n = uni((int)iw2.z);
      const int per = ((n + kDrainWaves * 64 - 1) / (kDrainWaves * 64)) * 64;
      const int lo = dw * per < n ? dw * per : n, hi = lo + per < n ? lo + per : n;
      if (!(kAblate & 4)) chores(iw1, iw3, lo, hi);
      const uint32_t nit = (uint32_t)uni((int)iw1.z);
      if (nit == 0u) continue;
      const int tq = uni((int)iw2.x);
      const int b = tq & 1;
      // the buffer's unit counter belongs to this tile only once the tile two back has been written out completely
      while (__hip_atomic_load(&s_epoch[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)(tq >> 1)) __builtin_amdgcn_s_sleep(1);
      while (__hip_atomic_load(&s_cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nit) __builtin_amdgcn_s_sleep(1);
      drain(iw2, iw3, b, lo, hi);
    }
  } else {
  // ---- the pipeline: records two units ahead, payload one unit ahead, in registers -------------------------------------
  Pos s0 = draw(), s1 = draw(), s2 = draw();
  uint4 rec_cur = recs[rec_index(s0)];
  uint4 rec_nxt = recs[rec_index(s1)];
  Payload cur;
  fetch_payload(rec_cur, cur);

  int tile_t = -1;              // the tile whose buffer this wave has been admitted to
  while (s0.t < seg_end) {
    const uint4 rec_nn = recs[rec_index(s2)];
    Payload nxt;
    fetch_payload(rec_nxt, nxt);

    const uint4 iw0 = info(s0.t, 0), iw1 = info(s0.t, 1), iw2 = info(s0.t, 2), iw3 = info(s0.t, 3);
    {
    const int tq = uni((int)iw2.x);
    const int buf = tq & 1;
    if (s0.t != tile_t) {       // first unit of this wave in the tile
      tile_t = s0.t;
      const int species = uni((int)iw2.w);
      if (species != w_species) {
        flush_reads();
        w_species = species;
      }
      // the buffer must have been written out by the tile two (non-empty) tiles back
      const uint32_t need = (uint32_t)(tq >> 1);
      while (__hip_atomic_load(&s_epoch[buf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(2);
    }
    const uint32_t lds_base = lds_base0 + (uint32_t)buf * (uint32_t)(16 * TILE);
    const int tile_start = (int)iw2.y;
    const int tile_len = (int)iw2.z;
    const int contig_len = (int)iw3.z;
    const int rg_ns = (int)iw0.w;
    const int vpos = s0.it * rpw + g;

    // ================= process (rec_cur, cur): the body of pileup_tiles.hip's stream loop ==========================
    bool fast = s0.it * rpw < rg_ns;
    if (fast) {
      const int fl = rec_l(rec_cur);
      const int frel = rec_pos(rec_cur) - tile_start;
      const bool fact = (rec_cur.w >> 31) == 0u;                       // not the sentinel
      const bool inside = frel >= 0 && frel + fl <= tile_len && fl > 0 && (rec_cur.w & ((uint32_t)kRecSimple << 24));
      fast = __ballot(fact && !inside) == 0ull;
      if (fast) {
        const int fnm = seg_nm(rec_cur);
        const int a_tot = seg_align_len(rec_cur);                      // aligned length of the whole read
        const int l_read = seg_read_l(rec_cur);                        // its l_seq
        const int min_match = s_tables[a_tot];                         // both <= max l_seq of the batch < table_len
        const int min_align = s_tables[p.table_len + l_read];
        const bool t_pid = a_tot - fnm < min_match;
        const bool t_noqual = (rec_cur.w & ((uint32_t)kRecQualAbsent << 24)) != 0u;
        const bool t_drop = (rec_qmean(rec_cur) < p.readq) | (rec_mapq(rec_cur) < p.mapq) | (a_tot < min_align);
        uint32_t err = t_noqual ? (uint32_t)E_NO_QUAL : 0u;            // same precedence as the general cascade
        err = t_pid ? 0u : err;
        err = fact ? err : 0u;
        const bool keep = fact & !(t_pid | t_noqual | t_drop);
        if (keep && q0 < fl) {
          uint32_t cd[NW];
#pragma unroll
          for (int w = 0; w < NW; ++w) cd[w] = (cur.qw[w] << 2) & 0x0C0C0C0Cu;   // code << 2: the counter's byte offset
          const uint32_t abase = ((uint32_t)(frel + q0) << 4) + lds_base;
          if (!(kAblate & 1)) tally_chunk(cur.qw, cd, (uint32_t)bq, abase, 1u);
        }
        const bool head = fact && c == 0 && seg_first(rec_cur);        // a read is counted by its first segment
        w_aligned += (uint32_t)__popcll(__ballot(head));
        w_mapped += (uint32_t)__popcll(__ballot(head && keep));
        if (head && err) atomicMin(p.err, ((unsigned long long)p.orig[read_at(iw0, iw1, vpos)] << 8) | err);
      }
    }
    if (!fast) {
      const int l = rec_l(rec_cur);
      const int pos = rec_pos(rec_cur);
      const uint32_t flags = rec_flags(rec_cur);
      bool act = (flags & kRecSentinel) == 0u;   // stream positions past the tile's reads fetched the sentinel
      const bool simple = (flags & kRecSimple) != 0u;   // a match segment: no CIGAR, read-level numbers in n / nm
      const int n = simple ? 1 : rec_n(rec_cur);
      // owner tile of a read = the tile holding its (clamped) start: it alone counts the read in the stats
      int cpos = pos < 0 ? 0 : pos;
      cpos = cpos > contig_len - 1 ? contig_len - 1 : cpos;
      const bool owner = act && cpos >= tile_start && cpos < tile_start + tile_len;
      const int rel = pos - tile_start;   // pos >= -1, tile_start >= 0: fits an int
      int rrel = (rel > (1 << 25) || rel < -(1 << 30)) ? (1 << 25) : rel;
      if (act && !owner && n == 1 && rrel + l <= 0) act = false;
      const bool has = act && q0 < l;

      // ---- soft-clip trimming ([EXT] pysam getQueryStart / getQueryEnd) ---------------------------
      int k0 = 0, lead_s = 0, trail_s = 0;
      const uint32_t* cig = nullptr;
      uint32_t cg0 = 0u, cg1 = 0u, cg2 = 0u, cg3 = 0u, cgl = 0u;   // first four CIGAR ops and the last one
      if (act && !simple) {
        cig = reinterpret_cast<const uint32_t*>(p.blob + (size_t)rec_off8(rec_cur) * 8 + blob_cigar_off((uint32_t)l, (uint32_t)lane_bases));
        if (n > 0) {
          const u32x4_a4 cv = *reinterpret_cast<const u32x4_a4*>(cig);   // may overhang into padding / next blob
          cg0 = cv.x; cg1 = cv.y; cg2 = cv.z; cg3 = cv.w;
          if (n > 4) cgl = cig[n - 1];
        }
        if (!(flags & kRecClipGeneric)) {
          if (n > 0 && (cg0 & 15u) == OP_S) { lead_s = (int)(cg0 >> 4); k0 = 1; }
          const uint32_t last = n > 4 ? cgl : (n == 2 ? cg1 : (n == 3 ? cg2 : cg3));
          if (n > 1 && (last & 15u) == OP_S) trail_s = (int)(last >> 4);
        } else {
          while (k0 < n) {
            const uint32_t v = cig[k0];
            const uint32_t op = v & 15u;
            if (op == OP_H) { ++k0; }
            else if (op == OP_S) { lead_s += (int)(v >> 4); ++k0; }
            else break;
          }
          for (int k = n - 1; k >= 1; --k) {   // index 0 is never inspected by pysam's backward walk
            const uint32_t v = cig[k];
            const uint32_t op = v & 15u;
            if (op == OP_H) continue;
            if (op == OP_S) trail_s += (int)(v >> 4); else break;
          }
        }
      }
      int align_len = (l - trail_s) - lead_s;
      align_len = align_len < 0 ? 0 : align_len;
      align_len = simple ? seg_align_len(rec_cur) : align_len;   // of the whole read
      const int l_read = simple ? seg_read_l(rec_cur) : l;
      const int min_match = s_tables[align_len < p.table_len ? align_len : 0];
      const int min_align = s_tables[p.table_len + (l_read < p.table_len ? l_read : 0)];
      const int nvalid = has ? (l - q0 < lane_bases ? l - q0 : lane_bases) : 0;

      // ---- keep_read (midas/run/snps.py:141-162), same order of evaluation ----------------------
      const int nm = simple ? seg_nm(rec_cur) : (int)rec_nm(rec_cur);
      const bool t_noseq = l == 0;
      const bool t_nonm = !simple && nm == (int)kNmAbsent;
      const bool t_zero = align_len == 0;
      const bool t_pid = align_len - nm < min_match;                                     // pid < mapid
      const bool t_noqual = (flags & kRecQualAbsent) != 0u;
      const bool t_drop = (rec_qmean(rec_cur) < p.readq) | (rec_mapq(rec_cur) < p.mapq) |
                          (align_len < min_align);                                       // readq, mapq, aln_cov
      const bool t_over = (flags & kRecOverrun) != 0u;   // kept, and its CIGAR reaches past SEQ inside the contig
      uint32_t err = t_over ? (uint32_t)E_CIGAR_OVERRUN : 0u;
      err = t_drop ? 0u : err;
      err = t_noqual ? (uint32_t)E_NO_QUAL : err;
      err = t_pid ? 0u : err;
      err = t_zero ? (uint32_t)E_ZERO_ALIGN : err;
      err = t_nonm ? (uint32_t)E_NO_NM : err;
      err = t_noseq ? (uint32_t)E_NO_SEQ : err;
      err = act ? err : 0u;
      const bool keep = act & !(t_noseq | t_nonm | t_zero | t_pid | t_noqual | t_drop | t_over);

      uint32_t cd[NW];   // byte offset of the base's counter inside its site (call code & 0xC)
      bool walking = keep && has;
      if (walking) {
#pragma unroll
        for (int w = 0; w < NW; ++w) cd[w] = (cur.qw[w] << 2) & 0x0C0C0C0Cu;   // code << 2: the counter's byte offset
      }

      // ---- CIGAR walk ([EXT] get_aligned_pairs(matches_only=True)): one match segment at a time -----
      int k = k0;
      int qpos = lead_s;
      const int q1 = q0 + nvalid;
      int jlo = 0, jhi = 0, loc0 = 0;
      auto next_segment = [&]() -> bool {
        while (k < n) {
          const uint32_t v = k < 4 ? (k == 0 ? cg0 : (k == 1 ? cg1 : (k == 2 ? cg2 : cg3))) : cig[k];
          ++k;
          const uint32_t op = v & 15u;
          const int len = (int)(v >> 4);
          const bool m = consumes_both(op);
          bool found = false;
          if (m) {
            const int lo = qpos > q0 ? qpos : q0;
            const int hi = (qpos + len) < q1 ? (qpos + len) : q1;
            found = lo < hi;
            if (found) { jlo = lo - q0; jhi = hi - q0; loc0 = rrel + (q0 - qpos); }
          }
          if (m || op == OP_I || op == OP_S) { qpos += len; qpos = qpos > (1 << 29) ? (1 << 29) : qpos; }
          if (m || op == OP_D || op == OP_N) { rrel += len; rrel = rrel > (1 << 29) ? (1 << 29) : rrel; }
          if (found) return true;   // H, P and anything else: no effect
        }
        return false;
      };
      if (walking) {
        if (simple) {
          jlo = 0; jhi = nvalid; loc0 = rrel + q0; k = n;
        } else {
          walking = next_segment();
        }
      }
      while (walking) {
        uint32_t q4[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) q4[w] = cur.qw[w];
        const int lo = jlo > -loc0 ? jlo : -loc0;
        const int hi = jhi < tile_len - loc0 ? jhi : tile_len - loc0;
        if (lo > 0 || hi < nvalid) {    // partial chunk (segment border or tile edge): zero the bytes outside
          const uint32_t below_hi = hi >= 32 ? 0xFFFFFFFFu : (hi <= 0 ? 0u : ((1u << hi) - 1u));
          const uint32_t below_lo = lo >= 32 ? 0xFFFFFFFFu : (lo <= 0 ? 0u : ((1u << lo) - 1u));
          const uint32_t jm = below_hi & ~below_lo;
#pragma unroll
          for (int w = 0; w < NW; ++w) q4[w] &= bits_to_bytes((jm >> (4 * w)) & 0xFu);
        }
        const uint32_t abase = ((uint32_t)loc0 << 4) + lds_base;
        if (!(kAblate & 1)) tally_chunk(q4, cd, (uint32_t)bq, abase, 1u);
        walking = (k < n) ? next_segment() : false;
      }

      const bool head = owner && c == 0 && (!simple || seg_first(rec_cur));   // a read is counted by its first segment
      w_aligned += (uint32_t)__popcll(__ballot(head));
      w_mapped += (uint32_t)__popcll(__ballot(head && keep));
      if (head && err)   // input-order index of the record
        atomicMin(p.err, ((unsigned long long)p.orig[read_at(iw0, iw1, vpos)] << 8) | err);
    }   // general path

    // ---- this unit is done: count it (LDS runs a wave's operations in order: the bump is behind the unit's tallies) ---
    if (lane == 0) (void)__hip_atomic_fetch_add(&s_cnt[buf], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    rec_cur = rec_nxt;
    rec_nxt = rec_nn;
    cur = nxt;
    s0 = s1;
    s1 = s2;
    s2 = draw();
  }
  }   // streaming waves
  if (seg_end >= t_end) {
    flush_reads();
    flush_sites();
  }
  __syncthreads();   // every tile of the segment is written out, its info records may be replaced
  }   // segments of the run
}

}  // namespace

hipError_t launch_pileup_stream(const PileupParams& p, hipStream_t stream) {
  const size_t dyn_lds = (size_t)p.table_len * 2 * sizeof(int32_t);
  static bool attr_set = false;
  if (!attr_set) {   // two 64 KiB tally buffers + tables: above the default dynamic-LDS limit
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pileup_stream_kernel<kTileShift>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(2 * (kMaxLSeq + 1) * sizeof(int32_t)));
    attr_set = true;
  }
  if (p.n_stream_wgs > 0)
    hipLaunchKernelGGL((pileup_stream_kernel<kTileShift>), dim3(p.n_stream_wgs), dim3(kStreamBlock), dyn_lds, stream, p);
  return hipGetLastError();
}

}  // namespace midas
