// gfx950 (CDNA4) pileup kernel of the MIDAS SNP path.  Integer counting: no MFMA.
//
// Reference semantics implemented here (citations into /root/reference):
//   keep_read                       midas/run/snps.py:141-162
//   count_coverage call site        midas/run/snps.py:194-199  ([EXT] pysam: get_aligned_pairs(matches_only),
//                                   qual >= quality_threshold, only 'A','C','G','T' counted)
//   depth / covered / total_depth   midas/run/snps.py:204-213
//   str(rec.seq).upper()            midas/run/snps.py:62
//
// Work decomposition.  The site space is cut into tiles of <= 2048 sites that never span contigs.  A
// persistent 256-thread workgroup (four per CU) takes tiles blockIdx, blockIdx + grid, then whatever a per-XCD counter hands it
// (a hot-spot tile comes as several parts, accumulated with global atomics by a second instantiation); a tile's tallies live in
// LDS as [site][A,C,G,T] u32, reads are streamed straight from the packed HBM arrays through a
// two-deep register prefetch pipeline, tallies are LDS atomics, and the tile is written out once,
// 16 B per site, fully coalesced.  The loads of the NEXT tile are issued before the write-out of the
// current one and the read-out re-zeroes LDS as it goes, so the memory-bound edges of a tile overlap
// the compute-bound middle of its neighbours instead of every workgroup marching through
// load -> compute -> store in lockstep (measured: those three phases used to simply add up).  The counts and
// alleles leave as non-temporal stores; the barriers between the phases wait for LDS traffic only.
//
// Lane mapping.  A lane owns 31 consecutive bases of one read (32 payload slots of one byte, the last one padding: layout.h
// says why 31, and when a batch uses all 32 instead): two 16-byte loads, quality and base code in the same byte.  A read of l_seq
// bases occupies ceil(l_seq/31) adjacent lanes (5 for 150 bp; `lanes_per_read` is fixed per batch from the longest read) and a wave works on
// floor(64 / lanes_per_read) reads at a time.  The read filter's numbers (aligned length, NM, floor of the mean
// quality) come with the record: the packer computed them once per read.
//
// Instruction diet.  The loop is instruction-issue and LDS bound long before it is HBM bound, so the per-base work is
// arranged as: (1) SWAR, four bases per instruction -- everything that decides WHETHER a base counts
// (not A/C/G/T, read tail, CIGAR segment, tile edge) is folded into the quality byte itself (a base that
// must not count gets quality 0); (2) per base -- one byte compare against baseq, one OR that forms the
// LDS address (site << 4 | call code), one predicated returnless ds_add.
#include "pileup_common.h"

namespace midas {

using namespace dev;
using namespace pile;

namespace {

// SPLIT = false: whole tiles, plain stores (the common case); SPLIT = true: the parts of split tiles (hot spots),
// launched separately so that the common case carries none of that code or its registers.
// Barrier between the phases of a tile.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: every wave would
// sit out the acknowledgement of its own write-out stores and the arrival of the next tile's prefetched records and
// payload before it may even wait for the others.  What the phases exchange lives in LDS only (tallies, counters), so
// the whole-tile kernel waits for LDS traffic alone; prefetched registers are guarded by the compiler's own counters.
// The parts kernel keeps the full fence (its global atomics are ordered against the tile's arrival ticket).
template <bool SPLIT>
__device__ __forceinline__ void tile_barrier() {
  if (SPLIT || (kDebug & 64)) __syncthreads();
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int TILE_SHIFT, bool SPLIT>
__global__ __launch_bounds__(kPileupBlock, 4) void pileup_tiles_kernel(PileupParams p) {
  constexpr int TILE = 1 << TILE_SHIFT;
  constexpr int NW = kChunk / 4;                 // quality words per lane
  constexpr int OUT_IT = TILE / kPileupBlock;    // read-out iterations per thread (8)
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * TILE];
  __shared__ unsigned long long s_stats[MIDAS_STATS];
  __shared__ uint32_t s_last_part;
  __shared__ uint32_t s_next_ticket;
  extern __shared__ __attribute__((aligned(16))) int32_t s_tables[];   // [min_match table_len][min_align table_len]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in an SGPR
  // Work items: one per tile, except that a tile holding very many reads (a coverage hot spot) comes as n_parts
  // items, each taking a slice of the tile's wave-iterations and adding its tallies to the output with atomics.
  const int w_end = SPLIT ? p.n_items : p.n_whole_items;      // items [0, n_whole) are whole tiles, the rest are parts
  // Items are handed out dynamically: the first two of a workgroup are blockIdx and blockIdx + grid, every further one
  // comes from a device counter (fetched two tiles ahead, so the atomic's round trip is never waited for).  With ~7
  // tiles per workgroup a static round-robin leaves the chip waiting for the workgroups that drew one tile more.
  // One counter per XCD (workgroups are dispatched round-robin over the 8 XCDs, so blockIdx % 8 names the XCD): it
  // hands out the items congruent to it modulo 8 and its cache line stays in that XCD's L2.  A single counter
  // bounces between the eight L2s and costs more than the balance gains (measured: 129 us against 108 us static).
  const int w_base = SPLIT ? p.n_whole_items : 0;
  const bool dynamic = !(kDebug & 16) && (gridDim.x % kSchedGroups) == 0;
  const int sched_group = (int)(blockIdx.x % kSchedGroups);
  uint32_t* const sched = p.split_ticket + p.n_tiles + (SPLIT ? kSchedWords : 0);   // [8 counters, 32 words apart][done]
  int w = w_base + (int)blockIdx.x;
  if (w >= w_end) return;          // (never: the grid is at most the number of items)
  int w_next = w + (int)gridDim.x;
  const ConstWords c_items = (ConstWords)(size_t)p.items;
  int t = (!SPLIT && p.n_whole_items == p.n_tiles) ? w : (int)c_items[4 * w];
  int part = SPLIT ? (int)c_items[4 * w + 1] : 0, nparts = SPLIT ? (int)c_items[4 * w + 2] : 1;

  {
    uint4* z = reinterpret_cast<uint4*>(lds);
    for (int i = tid; i < TILE; i += kPileupBlock) z[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < p.table_len; i += kPileupBlock) {
      s_tables[i] = p.filt->min_match[i];
      s_tables[p.table_len + i] = p.filt->min_align[i];
    }
    if (tid < MIDAS_STATS) s_stats[tid] = 0ull;
  }

  const int lpr = p.lanes_per_read;
  const int rpw = p.reads_per_wave;
  const int g = lane / lpr;
  const int c = lane - g * lpr;
  const bool lane_used = g < rpw;
  const int lane_bases = p.lane_bases;           // 31 or 32 (layout.h), uniform
  const int q0 = c * lane_bases;                 // first base of the lane; its payload sits in slot block c
  const int stride = (kPileupBlock / 64) * rpw;
  const uint4* recs = reinterpret_cast<const uint4*>(p.rec);
  const int bq = (int)base_threshold(p.baseq);   // a base counts iff its payload byte >= this (layout.h)
  // LDS byte address of the tally array (LDS pointers are 32-bit offsets on amdgcn)
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds;

  struct Payload {
    uint32_t qw[NW];  // 32 base bytes: (min(qual, 62) + 1) << 2 | code, 0 = never counts (layout.h)
  };
  // ---- two-deep prefetch: records two iterations ahead, payload one iteration ahead.  Nothing may be
  // computed from a loaded value here: any use would make the compiler drain the loads at once.
  // A tile's reads come as three index ranges (index_reads.hip): S = its simple reads that stay inside it,
  // I = reads of earlier tiles reaching in, G = every other read that starts in it.  They are dealt to the waves
  // as ONE virtual stream S, I, G: the leading wave-iterations are purely "simple", so the CIGAR-walk / clipping /
  // segment-mask code is branched over (not masked through) for ~85 % of the reads, there is a single mixed
  // iteration per tile, and the prefetch pipeline never restarts.
  const ConstWords c_tiles = (ConstWords)(size_t)p.tiles;
  const ConstWords c_rbinv = (ConstWords)(size_t)p.rbinv;
  const ConstWords c_rend = (ConstWords)(size_t)p.rend;
  auto load_ranges = [&](int tt) -> RawRanges {
    RawRanges w;
    w.vs = c_rbinv[3 * tt]; w.vg = c_rbinv[3 * tt + 1]; w.vi = c_rbinv[3 * tt + 2];
    w.se = c_rend[3 * tt]; w.ge = c_rend[3 * tt + 1]; w.ie = c_rend[3 * tt + 2];
    return w;
  };
  auto make_ranges = [&](const RawRanges& w) -> Ranges {
    Ranges q;
    const int se = (int)w.se, ge = (int)w.ge, ie = (int)w.ie;
    q.sb = w.vs ? p.n_reads - (int)w.vs : se;
    q.gb = w.vg ? p.n_reads - (int)w.vg : ge;
    q.ib = w.vi ? p.n_reads - (int)w.vi : ie;
    q.ns = se - q.sb;
    q.nsi = q.ns + (ie - q.ib);
    q.total = q.nsi + (ge - q.gb);
    return q;
  };
  // virtual stream position -> read index (n_reads = the sentinel record for positions past the end)
  auto read_at = [&](const Ranges& q, int v) -> int {
    // selects, not branches: base index of the range that stream position v falls into
    int base = q.gb - q.nsi;
    base = v < q.nsi ? q.ib - q.ns : base;
    base = v < q.ns ? q.sb : base;
    return (lane_used && v < q.total) ? base + v : p.n_reads;
  };
  // v = stream position of this lane's read
  auto fetch_rec = [&](const Ranges& q, int v) -> uint4 { return recs[read_at(q, v)]; };

  auto fetch_payload = [&](const uint4& rv, Payload& d) {
    // lanes without a chunk (the sentinel record has l_seq 0) load nothing and never look at their payload
    // registers: `has` guards every use
    const int l = rec_l(rv);
    const uint8_t* bp = p.blob + (size_t)rec_off8(rv) * 8;
    if (q0 < l) {
      const u32x4_a8 qa = *reinterpret_cast<const u32x4_a8*>(bp + c * kChunk);
      const u32x4_a8 qb = *reinterpret_cast<const u32x4_a8*>(bp + c * kChunk + 16);
      d.qw[0] = qa.x; d.qw[1] = qa.y; d.qw[2] = qa.z; d.qw[3] = qa.w;
      d.qw[4] = qb.x; d.qw[5] = qb.y; d.qw[6] = qb.z; d.qw[7] = qb.w;
    }
  };
  constexpr int NWAVES = kPileupBlock / 64;
  Tile tile = load_tile(c_tiles, t);
  Ranges rg = make_ranges(load_ranges(t));
  const int vstep = NWAVES * rpw;                 // stream positions between a wave's consecutive iterations
  // iterations [it_lo, it_hi) of the tile belong to this item (all of them unless the tile was split)
  auto first_iter = [&](const Ranges& q, int pt, int np) -> int { return (int)(((long long)((q.total + rpw - 1) / rpw) * pt) / np); };
  int it_lo = first_iter(rg, part, nparts), it_hi = first_iter(rg, part + 1, nparts);
  int v0 = (it_lo + wave) * rpw + g;              // this lane's stream position in the wave's first iteration
  uint4 rec_cur = fetch_rec(rg, v0);
  uint4 rec_nxt = fetch_rec(rg, v0 + vstep);
  Payload cur;
  fetch_payload(rec_cur, cur);
  __syncthreads();   // LDS zeroed, tables in place

  unsigned long long acc_cov = 0ull, acc_depth = 0ull;   // covered sites / depth of this thread's sites since the last flush
  for (;;) {
    const int tile_len = tile.len;
    const int tile_start = tile.start;
    uint32_t w_aligned = 0, w_mapped = 0;
    // the tile's reference letters are fetched now and written (upper-cased) with the tile's counts: loading them
    // in the write-out put a full HBM round trip on every wave's critical path, once per tile
    constexpr int REF_IT = TILE / (4 * kPileupBlock);
    uint32_t refw[REF_IT];
    if (p.out_allele) {
      const uint8_t* ref = p.ref + tile.site_base;
#pragma unroll
      for (int it = 0; it < REF_IT; ++it) {
        const int i = 4 * (tid + it * kPileupBlock);
        if (i + 4 <= tile_len) refw[it] = *reinterpret_cast<const u32_a1*>(ref + i);
      }
    }

    int vpos = v0;
    // The record/payload prefetch runs across the tile boundary: a wave's last two iterations fetch the records of its
    // first two iterations of the NEXT tile where the stream of this one has nothing left, so that nothing is loaded
    // -- or waited for -- between the stream loop and the write-out.  Only the two record indices are kept (the next
    // tile's tables are read here, early, and again after the loop, by then from the scalar cache).
    const int n_w = it_hi > it_lo + wave ? (it_hi - (it_lo + wave) + NWAVES - 1) / NWAVES : 0;
    const bool xt = !SPLIT && p.n_whole_items == p.n_tiles && w_next < w_end && n_w >= 2;
    int xidx0 = p.n_reads, xidx1 = p.n_reads;
    if (xt) {
      const Ranges xr = make_ranges(load_ranges(w_next));
      const int xv0 = (first_iter(xr, 0, 1) + wave) * rpw + g;
      xidx0 = read_at(xr, xv0);
      xidx1 = read_at(xr, xv0 + vstep);
    }
    for (int it = it_lo + wave; it < it_hi; it += NWAVES, vpos += vstep) {
      int nn_idx = read_at(rg, vpos + 2 * vstep);
      if (xt && it + 2 * NWAVES >= it_hi) nn_idx = (it + NWAVES < it_hi) ? xidx0 : xidx1;
      const uint4 rec_nn = recs[nn_idx];
      Payload nxt;
      fetch_payload(rec_nxt, nxt);

      // ================= process (rec_cur, cur) ===================================================
      // Fast path: every record of this wave-iteration comes from the tile's S range: gap-free match segments (the
      // packer resolves CIGARs into them, layout.h) that start AND end inside the tile -- ~96 % of the iterations at
      // 150 bp / 4096 sites.  Nothing of the CIGAR walk, clipping, tile-edge or ownership logic applies: sites = pos +
      // query index, the read-level numbers of the filter come out of the record.  The general code below is for
      // everything else: records reaching into the next tile, reads that keep their CIGAR (and any S record that does
      // not lie inside the tile after all -- checked, not assumed).
      // ... including the iteration that holds the last few S records: the stream positions behind them carry I / G
      // records (never "simple and inside": the ballot below sends the iteration down the general path) or, when the
      // tile has none, the sentinel.  (That last, partial iteration used to take the general path in every tile, and the
      // seven other waves waited for it at the barrier: -2 %.)
      bool fast = it * rpw < rg.ns && !(kDebug & 4);
      if (fast) {
        const int fl = rec_l(rec_cur);
        const int frel = rec_pos(rec_cur) - tile_start;
        const bool fact = (rec_cur.w >> 31) == 0u;                       // not the sentinel
        const bool inside = frel >= 0 && frel + fl <= tile_len && fl > 0 && (rec_cur.w & ((uint32_t)kRecSimple << 24));
        fast = __ballot(fact && !inside) == 0ull;
        if (fast) {
          // read-level numbers of the filter travel in the segment's record
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
            if (!(kDebug & 1)) tally_chunk(cur.qw, cd, (uint32_t)bq, abase, 1u);
          }
          const bool head = fact && c == 0 && seg_first(rec_cur);        // S records start in this tile: it owns them;
                                                                         // a read is counted by its first segment
          w_aligned += (uint32_t)__popcll(__ballot(head));
          w_mapped += (uint32_t)__popcll(__ballot(head && keep));
          if (head && err) atomicMin(p.err, ((unsigned long long)p.orig[read_at(rg, vpos)] << 8) | err);
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
      cpos = cpos > tile.contig_len - 1 ? tile.contig_len - 1 : cpos;
      const bool owner = act && cpos >= tile_start && cpos < tile_start + tile_len && !(tile.halo && pos < 0);
      // position of the read relative to the tile; a read that can never reach the tile is parked far right
      // (reference positions only grow along a CIGAR, so "far right" stays far right)
      const int rel = pos - tile_start;   // pos >= -1, tile_start >= 0: fits an int
      int rrel = (rel > (1 << 25) || rel < -(1 << 30)) ? (1 << 25) : rel;
      // reads that start in an earlier tile and provably end before this one: nothing to do here
      if (act && !owner && n == 1 && rrel + l <= 0) act = false;
      const bool has = act && q0 < l;

      // ---- soft-clip trimming ([EXT] pysam getQueryStart / getQueryEnd) ---------------------------
      int k0 = 0, lead_s = 0, trail_s = 0;
      const uint32_t* cig = nullptr;
      uint32_t cg0 = 0u, cg1 = 0u, cg2 = 0u, cg3 = 0u, cgl = 0u;   // first four CIGAR ops and the last one
      if (act && !simple) {
        cig = reinterpret_cast<const uint32_t*>(p.blob + (size_t)rec_off8(rec_cur) * 8 + blob_cigar_off((uint32_t)l, (uint32_t)lane_bases));
        if (n > 0) {
          // not prefetched (it would cost 10 VGPRs of the double-buffered payload): only the ~15 % mixed / general
          // wave-iterations come here, the pure-simple ones branch over all of this
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
      // exact integer form of the two fp64 ratio tests (tables built by the host with the reference's expressions)
      const int min_match = s_tables[align_len < p.table_len ? align_len : 0];
      const int min_align = s_tables[p.table_len + (l_read < p.table_len ? l_read : 0)];

      const int nvalid = has ? (l - q0 < lane_bases ? l - q0 : lane_bases) : 0;

      // ---- keep_read (midas/run/snps.py:141-162), same order of evaluation ----------------------
      // Every test is evaluated (selects, no branches: the cascade used to cost seven exec-mask branches per
      // iteration); the reference's order decides which outcome wins, lowest priority first.
      const int nm = simple ? seg_nm(rec_cur) : (int)rec_nm(rec_cur);
      const bool t_noseq = l == 0;
      const bool t_nonm = !simple && nm == (int)kNmAbsent;
      const bool t_zero = align_len == 0;
      const bool t_pid = align_len - nm < min_match;                                     // pid < mapid
      const bool t_noqual = (flags & kRecQualAbsent) != 0u;
      // np.mean(q) < readq  <=>  floor(sum(q) / l) < readq for an integer readq; the packer stored the quotient
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

      // ---- per-base call codes and validity, four bases per instruction ----------------------------
      // cur.qw: one byte per base, quality above the code; 0 = never counts (not A/C/G/T, padding, past the record)
      uint32_t cd[NW];   // byte offset of the base's counter inside its site (call code & 0xC)
      bool walking = keep && has && !(kDebug & 4);
      if (walking) {     // (cd is only ever read under `walking`)
#pragma unroll
        for (int w = 0; w < NW; ++w) cd[w] = (cur.qw[w] << 2) & 0x0C0C0C0Cu;   // code << 2: the counter's byte offset
      }

      // ---- CIGAR walk ([EXT] get_aligned_pairs(matches_only=True)): one match segment at a time -----
      // 32-bit saturating positions: a query position only matters below q1 <= 1024 and a tile-relative
      // reference position only below the tile length, and both only ever grow.
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
          if (m || op == OP_I || op == OP_S || (op == OP_P && p.pad_advances)) { qpos += len; qpos = qpos > (1 << 29) ? (1 << 29) : qpos; }
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
        // bases of the chunk that belong to this segment AND lie inside the tile: [lo, hi)
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
        if (!(kDebug & 1)) tally_chunk(q4, cd, (uint32_t)bq, abase, 1u);
        walking = (k < n) ? next_segment() : false;
      }

      // ---- per-species read counters: one ballot per wave -----------------------------------------
      const bool head = owner && c == 0 && (!simple || seg_first(rec_cur));   // a read is counted by its first segment
      const unsigned long long m_al = __ballot(head);
      const unsigned long long m_mp = __ballot(head && keep);
      w_aligned += (uint32_t)__popcll(m_al);
      w_mapped += (uint32_t)__popcll(m_mp);
      // (a read of the piece in front, midas_snps_contigs.origin: its own piece reports what keep_read raises, this one the
      // overrun its walk runs into here)
      const bool walk_err = act && tile.halo && pos < 0 && c == 0 && err == (uint32_t)E_CIGAR_OVERRUN;
      if ((head && err) || walk_err)   // input-order index of the record
        atomicMin(p.err, ((unsigned long long)p.orig[read_at(rg, vpos)] << 8) | err);
      }   // general path

      rec_cur = rec_nxt;
      rec_nxt = rec_nn;
      cur = nxt;
    }

    if (lane == 0) {
      if (w_aligned) atomicAdd(&s_stats[MIDAS_STAT_ALIGNED], (unsigned long long)w_aligned);
      if (w_mapped) atomicAdd(&s_stats[MIDAS_STAT_MAPPED], (unsigned long long)w_mapped);
    }
    // ---- next tile: its record loads go out before the barrier, its payload loads before this tile's stores ----
    const int wn = w_next;
    const bool more = wn < w_end;
    // whole-tile items are the identity list unless some tile of the batch was split: no dependent load then
    const bool ident = !SPLIT && p.n_whole_items == p.n_tiles;
    const int tn = more ? (ident ? wn : (int)c_items[4 * wn]) : t;   // scalar loads (constant address space)
    const int npart = (SPLIT && more) ? (int)c_items[4 * wn + 1] : (SPLIT ? part : 0);
    const int nnparts = (SPLIT && more) ? (int)c_items[4 * wn + 2] : (SPLIT ? nparts : 1);
    const Tile ntile = load_tile(c_tiles, tn);
    const Ranges nrg = make_ranges(load_ranges(tn));
    const int nit_lo = first_iter(nrg, npart, nnparts), nit_hi = first_iter(nrg, npart + 1, nnparts);
    const int nv0 = (nit_lo + wave) * rpw + g;
    if (more && !xt) {   // (with xt the pipeline already holds the next tile's first two records and first payload)
      rec_cur = fetch_rec(nrg, nv0);
      rec_nxt = fetch_rec(nrg, nv0 + vstep);
    }
    tile_barrier<SPLIT>();   // every tally of this tile is in LDS
    // the item after the next one: the counter is asked here, behind the next tile's record loads and ahead of its
    // payload loads, whose first use is a whole write-out away (loads and returning atomics come back in order: an
    // atomic issued ahead of the stream loop held back every load of the tile's first iterations)
    if (more && !xt) fetch_payload(rec_cur, cur);
    uint32_t ticket = 0;
    if (dynamic && more && tid == 0) ticket = atomicAdd(&sched[32 * sched_group], 1u);

    // ---- emit the tile: counts[site][A,C,G,T] (and re-zero LDS), covered/total-depth partials ----------
    unsigned long long covered = 0, depth_sum = 0;
    {
      // debug bit 8 (timing experiment only, results wrong): every tile's counts land on the first tile's sites, so
      // the stores stay in L2 and cost no HBM write bandwidth
      uint4* out = reinterpret_cast<uint4*>(p.out_counts) + ((kDebug & 8) ? 0 : tile.site_base);
      uint4* lds4 = reinterpret_cast<uint4*>(lds);
      const int lim = (kDebug & 2) ? 0 : tile_len;
      if constexpr (!SPLIT) {
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it) {
          const int i = tid + it * kPileupBlock;
          if (i < lim) {
            const uint4 v = lds4[i];
            lds4[i] = make_uint4(0u, 0u, 0u, 0u);
            {   // streaming store: the counts are never read again here, and a plain store keeps its lines in L2 at
                // the expense of the reads' (measured: 107.5 -> 103 us; non-temporal LOADS of the payload: 116 us)
              u32x4_a8 nv; nv.x = v.x; nv.y = v.y; nv.z = v.z; nv.w = v.w;
              __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_a8*>(out + i));
            }
            const uint32_t d = v.x + v.y + v.z + v.w;
            covered += d > 0u ? 1ull : 0ull;
            depth_sum += d;
          }
        }
      } else {
        // one slice of a split tile: the output was zeroed before the launch, every part adds to it; the part that
        // arrives last counts the covered sites from the finished counts (below)
        uint32_t* outw = p.out_counts + 4 * tile.site_base;
        for (int it = 0; it < OUT_IT; ++it) {
          const int i = tid + it * kPileupBlock;
          if (i < lim) {
            const uint4 v = lds4[i];
            lds4[i] = make_uint4(0u, 0u, 0u, 0u);
            if (v.x) atomicAdd(&outw[4 * (size_t)i + 0], v.x);
            if (v.y) atomicAdd(&outw[4 * (size_t)i + 1], v.y);
            if (v.z) atomicAdd(&outw[4 * (size_t)i + 2], v.z);
            if (v.w) atomicAdd(&outw[4 * (size_t)i + 3], v.w);
            depth_sum += (unsigned long long)v.x + v.y + v.z + v.w;
          }
        }
        __threadfence();   // this thread's additions are visible device-wide before the workgroup takes its ticket
      }
    }
    // ---- upper-cased ref allele, four sites per lane ---------------------------------------------------------
    if (p.out_allele && !(kDebug & 2) && part == 0) {
      const uint8_t* ref = p.ref + tile.site_base;
      uint8_t* al = p.out_allele + tile.site_base;
#pragma unroll
      for (int it = 0; it < REF_IT; ++it) {
        const int i = 4 * (tid + it * kPileupBlock);
        if (i + 4 <= tile_len) {
          __builtin_nontemporal_store(upper4(refw[it]), reinterpret_cast<u32_a1*>(al + i));   // streaming, like the counts
        } else {
          for (int j = i; j < tile_len; ++j) {
            uint32_t ch = ref[j];
            if (ch >= 'a' && ch <= 'z') ch -= 32u;
            al[j] = (uint8_t)ch;
          }
        }
      }
    }

    if constexpr (!SPLIT) {
      // covered sites / depth stay in the thread until the species row is flushed: the wave reduction and the LDS
      // atomics would sit between the write-out and the barrier of every tile
      acc_cov += covered;
      acc_depth += depth_sum;
    } else {
      for (int d = 32; d >= 1; d >>= 1) {
        covered += __shfl_down(covered, d);
        depth_sum += __shfl_down(depth_sum, d);
      }
      if (lane == 0) {
        if (covered) atomicAdd(&s_stats[MIDAS_STAT_COVERED], covered);
        if (depth_sum) atomicAdd(&s_stats[MIDAS_STAT_DEPTH], depth_sum);
      }
    }
    if (dynamic && more && tid == 0) s_next_ticket = ticket;
    // One barrier per tile after the write-out: tallies re-zeroed (and this tile's s_stats additions done) before
    // the next tile's waves touch LDS.  The workgroup's counters go to the species row only when the next tile
    // belongs to another species or there is no next tile: they are additive, so tiles of one species share them.
    tile_barrier<SPLIT>();
    if constexpr (SPLIT) {
      if (tid == 0) {
        const uint32_t ticket = atomicAdd(&p.split_ticket[t], 1u);
        s_last_part = ticket == (uint32_t)(nparts - 1) ? 1u : 0u;
        if (s_last_part) p.split_ticket[t] = 0u;   // every part has arrived: ready for the next run
      }
      __syncthreads();
      if (s_last_part) {   // all parts' additions are in: covered sites of the finished tile
        __threadfence();
        const uint32_t* outw = p.out_counts + 4 * tile.site_base;
        unsigned long long cov = 0;
        for (int i = tid; i < tile_len; i += kPileupBlock) {
          const uint32_t a = __hip_atomic_load(&outw[4 * (size_t)i + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t b = __hip_atomic_load(&outw[4 * (size_t)i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t cc = __hip_atomic_load(&outw[4 * (size_t)i + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t d = __hip_atomic_load(&outw[4 * (size_t)i + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          cov += (a | b | cc | d) ? 1ull : 0ull;
        }
        for (int d = 32; d >= 1; d >>= 1) cov += __shfl_down(cov, d);
        if (lane == 0 && cov) atomicAdd(&s_stats[MIDAS_STAT_COVERED], cov);
      }
      __syncthreads();
    }
    if (more) {
      const long long nn = (dynamic && !(kDebug & 32)) ? (long long)w_base + 2ll * (long long)gridDim.x + (long long)kSchedGroups * s_next_ticket + sched_group
                                                        : (long long)wn + (long long)gridDim.x;
      w_next = __builtin_amdgcn_readfirstlane((int)(nn < (long long)w_end ? nn : (long long)w_end));
    }
    const bool flush = !more || ntile.species != tile.species;   // workgroup-uniform
    if (flush) {
      if constexpr (!SPLIT) {
        for (int d = 32; d >= 1; d >>= 1) {
          acc_cov += __shfl_down(acc_cov, d);
          acc_depth += __shfl_down(acc_depth, d);
        }
        if (lane == 0) {
          if (acc_cov) atomicAdd(&s_stats[MIDAS_STAT_COVERED], acc_cov);
          if (acc_depth) atomicAdd(&s_stats[MIDAS_STAT_DEPTH], acc_depth);
        }
        acc_cov = 0ull;
        acc_depth = 0ull;
        tile_barrier<SPLIT>();
      }
      if (tid < MIDAS_STATS) {
        const unsigned long long v = s_stats[tid];
        if (v) atomicAdd(&p.stats[(size_t)tile.species * MIDAS_STATS + tid], v);
        s_stats[tid] = 0ull;
      }
      if (!more) {
        if (dynamic && tid == 0) {   // the last workgroup to leave rewinds the counters for the next launch (every
          // ticket this workgroup drew has come back and been used by now, so its arrival is ordered behind them; no
          // fence: an agent-scope fence here writes back and invalidates the XCD's whole L2, once per workgroup)
          if (atomicAdd(&sched[32 * kSchedGroups], 1u) == gridDim.x - 1u) {
            for (int k = 0; k <= kSchedGroups; ++k) sched[32 * k] = 0u;
          }
        }
        break;
      }
      tile_barrier<SPLIT>();   // s_stats reset before the next tile adds to it
    }
    w = wn;
    part = npart;
    nparts = nnparts;
    it_lo = nit_lo;
    it_hi = nit_hi;
    v0 = nv0;
    t = tn;
    tile = ntile;
    rg = nrg;
  }
}

}  // namespace

hipError_t launch_pileup_tiles(const PileupParams& p, hipStream_t stream, bool whole_tiles, bool parts) {
  const size_t dyn_lds = (size_t)p.table_len * 2 * sizeof(int32_t);
  if (whole_tiles && p.n_whole_items > 0) {
    const int grid = p.n_whole_items < p.grid_blocks ? p.n_whole_items : p.grid_blocks;
    hipLaunchKernelGGL((pileup_tiles_kernel<kTileShift, false>), dim3(grid), dim3(kPileupBlock), dyn_lds, stream, p);
  }
  const int n_parts = p.n_items - p.n_whole_items;
  if (parts && n_parts > 0) {   // hot spots: the parts of split tiles, merged with atomics
    const int grid = n_parts < p.grid_blocks ? n_parts : p.grid_blocks;
    hipLaunchKernelGGL((pileup_tiles_kernel<kTileShift, true>), dim3(grid), dim3(kPileupBlock), dyn_lds, stream, p);
  }
  return hipGetLastError();
}

}  // namespace midas
