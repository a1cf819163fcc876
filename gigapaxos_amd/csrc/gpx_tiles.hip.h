/*
 * gpx_tiles.hip.h — the TILED front end of the accept-reply call (round 6).
 *
 * Round 5's slotted front end gave every (bucket, scatter workgroup) pair a FIXED slot of 24 records: whole lines on
 * the way out, but the per-bucket kernel then fetched all three lines of every slot whatever they held (69 MB for
 * 24 MB of records at the headline shape), a run longer than its slot went to overflow lists (a stream sorted by
 * group: every vote - 2.3x the shuffled time), a kernel in between summed the count matrix, and the per-bucket
 * kernel's register rounds capped a call at 192 scatter workgroups (five replicas at 1 M groups: 306).  This form
 * has no slots:
 *
 *   k_scatter_tiles<NT, R4>  one workgroup counting-SORTS its tile of T = NT * 4 * R4 votes by bucket in LDS, as
 *                        8-byte records (offset in the tile | local group | escapes, and slot / max_cp / acceptor
 *                        relative to the batch's reference vote), and writes the sorted tile as ONE contiguous run
 *                        of A.recs (16-byte stores, nothing but whole lines) plus its column of A.off: where each
 *                        bucket's run starts inside the tile.  All six columns of a vote are loaded once, in one
 *                        phase (round 5 loaded the group column, sorted, then loaded the other five: two exposed
 *                        memory phases per workgroup with one workgroup per CU).  No histogram pass, no atomics on
 *                        shared words, no capacity anywhere: a run is as long as it is.
 *   k_bucket16<.., TILES> (gpx_ar16.hip.h) reads rows b and b + 1 of A.off - the run of every tile in its bucket;
 *                        their starts add up to the records BEFORE the bucket (its place in the staging areas: no
 *                        totals kernel), their lengths to its own - and then its records by a flat index -> (tile,
 *                        position) search over the prefix of the run lengths in LDS: a shuffled stream's 184 short
 *                        runs and a sorted stream's one long run cost the same per record.
 *
 * Records that do not fit the 8-byte form: another ballot than the reference's or an acceptor id beyond 16 bits
 * (ESC_B: the per-bucket kernel fetches ballot and acceptor from the caller's columns, as the 16-byte records'
 * escape path does), a slot or checkpoint that does not fit a byte next to the reference's (ESC_S: fetched from the
 * columns) - unless the TILE has many of those (groups that are not in lock-step: every deployment that is not a
 * benchmark), then the whole tile is WIDE: its slots and checkpoints follow the sort into A.ext as 8 more bytes per
 * vote and nothing is fetched by arrival index.  The reference vote is the majority of the batch's first 64 votes,
 * not vote 0 (round 5: an odd vote 0 sent all 3 M votes down the escape path).
 *
 * Taken by every accept-reply call that is one pass over the table with 16-byte aligned columns and at most
 * GPX_TL_MAXWG tiles (gpx_engine.hip: ar_tiles_call).
 */
#pragma once
#include "gpx_kernels.hip.h"

/* the bucket counters of a scatter workgroup: whole rounds of 4 or 8 per thread (tl_publish_starts) */
__host__ __device__ __forceinline__ int32_t tl_cnt_words(int32_t nbk, int32_t nt) {
  const int32_t per = (nbk + 1 + nt - 1) / nt; /* (+ 1: the entry behind the last bucket holds the tile's total) */
  return nt * (per <= 4 ? 4 : 8);
}
#ifndef GPX_TL_MAXWG
#define GPX_TL_MAXWG 1024
#endif /* tiles of a call at most: the per-bucket kernel keeps their run starts and prefix in LDS */
struct __attribute__((aligned(8))) Vote8 {
  uint32_t a; /* offset of the vote in its tile (14 bits) | local group << 14 (10 bits) | ESC_S << 30 | ESC_B << 31 */
  uint32_t b; /* slot - slot0 + 128 (8 bits) | (slot - 1 - max_cp + 128) << 8 (8 bits) | acceptor << 16 */
};
#define V8_ESC_S 0x40000000u /* slot / max_cp do not fit: from the caller's columns (or from A.ext in a wide tile) */
#define V8_ESC_B 0x80000000u /* ballot / acceptor do not fit: from the caller's columns */
#define TL_WIDE 0x8000u      /* in an A.off entry: this tile is wide */
struct TileArea {
  Vote8* recs;   /* [nwg][tile] every tile sorted by bucket */
  int2* ext;     /* [nwg][tile] {slot, max_cp} at the same positions, written by wide tiles only */
  uint16_t* off; /* [nbk / 4 + 1][nwg_pad][4] start of bucket b's run inside tile w (entry nbk: the tile's records) | TL_WIDE */
  int32_t* ref;  /* [4] the batch's reference vote: slot, ballot number, ballot coordinator (written by tile 0) */
  int32_t nwg, nwg_pad, tile;
  int32_t xcd_rows; /* per-bucket kernel: consecutive buckets on one XCD (they share lines of A.off and A.recs) */
};

/* the 16-byte record of a tile entry: {arrival index, slot, max_cp, local group | V16_ESC | acceptor << 16} */
__device__ __forceinline__ I4 tile_expand(const Vote8 v, const int2* __restrict__ ext_at, int32_t w, int32_t tile, int32_t slot0,
                                          const int32_t* __restrict__ slot_col, const int32_t* __restrict__ maxcp_col) {
  I4 r;
  r.x = w * tile + (int32_t)(v.a & 0x3fffu);
  const uint32_t lg = (v.a >> 14) & 0x3ffu;
  if (ext_at) {
    const int2 e = *ext_at;
    r.y = e.x;
    r.z = e.y;
  } else if (v.a & V8_ESC_S) {
    r.y = slot_col[r.x];
    r.z = maxcp_col[r.x];
  } else {
    r.y = slot0 + (int32_t)(v.b & 255u) - 128;
    r.z = r.y - 1 - ((int32_t)((v.b >> 8) & 255u) - 128);
  }
  r.w = (int32_t)(lg | ((v.a & V8_ESC_B) ? 0x4000u /* V16_ESC */ : (v.b & 0xffff0000u)));
  return r;
}

/* The value most of the wave's first `m` lanes hold (m <= 64), if more than half of them agree on one; lane 0's
 * otherwise.  Four candidates at most: an odd vote or three in front do not change the answer. */
__device__ __forceinline__ void wave_majority2(int32_t x, int32_t y, int32_t m, int32_t* ox, int32_t* oy) {
  const int lane = (int)__lane_id();
  const unsigned long long act = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
  unsigned long long tried = 0;
  int32_t cx = __shfl(x, 0, 64), cy = __shfl(y, 0, 64);
  const int32_t fx = cx, fy = cy;
  for (int t = 0; t < 4; t++) {
    const unsigned long long same = __ballot(x == cx && y == cy) & act;
    if (2 * __popcll(same) > __popcll(act)) {
      *ox = cx, *oy = cy;
      return;
    }
    tried |= same;
    const unsigned long long rest = act & ~tried;
    if (!rest) break;
    const int nl = __ffsll((long long)rest) - 1;
    cx = __shfl(x, nl, 64), cy = __shfl(y, nl, 64);
  }
  (void)lane;
  *ox = fx, *oy = fy;
}

/* A 16-byte store that is written THROUGH the XCD's L2 (sc1) instead of staying dirty in it: what a kernel leaves dirty
 * is written back at its end, at ~6 TB/s, before the next kernel starts (MI355X_MICROARCH.md, price list: "boundary") -
 * 24 MB of sorted records were 4 us of k_scatter_tiles behind its last workgroup's exit.  (Narrower sc1 stores cost a
 * fabric write each: only for 16-byte streams.) */
typedef int32_t gpx_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_through(I4* dst, const I4 v) {
#ifdef GPX_NO_SC1_STORES
  *dst = v;
#else
  gpx_i32x4 x;
  x.x = v.x, x.y = v.y, x.z = v.z, x.w = v.w;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(x) : "memory");
#endif
}

/* four consecutive entries of a column; FULL: the whole vector is inside the batch (one 16-byte load) */
template <bool FULL>
__device__ __forceinline__ I4 tiles_load4(const int32_t* __restrict__ col, int64_t i0, int32_t n, int32_t fill) {
  if (FULL) return *(const I4*)(col + i0);
  I4 r;
  r.x = i0 + 0 < n ? col[i0 + 0] : fill;
  r.y = i0 + 1 < n ? col[i0 + 1] : fill;
  r.z = i0 + 2 < n ? col[i0 + 2] : fill;
  r.w = i0 + 3 < n ? col[i0 + 3] : fill;
  return r;
}

/* block_exscan_n with the wave totals in caller-provided LDS: this kernel declares NO static LDS - the dynamic block
 * starts behind the static one at whatever offset that leaves (84 bytes in round 6's first build), and the sorted
 * records' 8-byte stores and 16-byte loads must be aligned (misaligned they cost three times the cycles) */
template <int NT>
__device__ __forceinline__ int32_t tl_block_exscan(int32_t v, int32_t* total, int32_t* wsum) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int32_t x = wave_incscan(v);
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  /* the wave totals (at most 16) scanned by every wave itself: one LDS read, one wave scan, two lane reads */
  const int32_t ps = wave_incscan(lane < NT / 64 ? wsum[lane] : 0);
  *total = __shfl(ps, NT / 64 - 1, 64);
  const int32_t base = __shfl(ps, wid > 0 ? wid - 1 : 0, 64);
  /* (no second barrier: the caller's own comes before anything writes wsum again) */
  return (wid > 0 ? base : 0) + x - v;
}
/* Exclusive scan of the bucket counts -> where each bucket's run starts in the sorted tile, and this tile's column of
 * A.off.  Four (or eight) consecutive buckets per thread, cnt[] padded to whole rounds of zeroed words, and the four
 * starts leave as ONE 8-byte store: A.off is [bucket / 4][tile][4].  (Round 6's first forms: a runtime count per thread,
 * every access tested, 64-bit addresses, and a 2-byte store per bucket - 1,955 scattered partial-line stores per
 * workgroup, 1.7 us of this phase's 3.5 by ablation.) */
struct __attribute__((aligned(8))) Off4 {
  uint16_t h[4];
};
template <int NT, int P>
__device__ __forceinline__ int32_t tl_publish_starts(int32_t* cnt, int32_t* s_wsum, const TileArea& A, int32_t w, int32_t nbk,
                                                     uint32_t wf) {
  static_assert(P % 4 == 0, "whole entries of A.off");
  const int32_t bq = (int32_t)threadIdx.x * P;
  int32_t v[P], s = 0;
#pragma unroll
  for (int q = 0; q < P; q++) {
    v[q] = cnt[bq + q];
    s += v[q];
  }
  int32_t tot;
  int32_t ex = tl_block_exscan<NT>(s, &tot, s_wsum);
#pragma unroll
  for (int q4 = 0; q4 < P; q4 += 4) {
    Off4 o;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      cnt[bq + q4 + q] = ex;
      o.h[q] = (uint16_t)((uint32_t)ex | wf); /* (beyond the last bucket: the tile's total - row nbk is read as "the end") */
      ex += v[q4 + q];
    }
#ifdef GPX_TL_TRACE /* ablation (trace build only, GPX_TL_ABLATE bit 1): the phase without its scattered stores */
    if (A.xcd_rows & 2) continue;
#endif
    if (bq + q4 <= nbk) ((Off4*)A.off)[(uint32_t)((bq + q4) >> 2) * (uint32_t)A.nwg_pad + (uint32_t)w] = o;
  }
  return tot;
}

#define TL_RK_MASK 0x3fffu /* rank of a vote inside its bucket's run of the tile; ESC_S / ESC_B above it (Vote8.a's bits) */
/* FULL: every vector of the tile lies inside the batch and the status column takes 4-byte stores - every workgroup but
 * the last. */
template <int NT, int R4, bool FULL>
__device__ __forceinline__ void scatter_tile(int32_t n, int32_t G, const DevScratch& X, const TileArea& A, int32_t w,
                                             const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
                                             const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot,
                                             const int32_t* __restrict__ acceptor, const int32_t* __restrict__ max_cp,
                                             uint8_t* __restrict__ status, int32_t* cnt, Vote8* recs, const int32_t* s_ref,
                                             int32_t* s_nesc, int32_t* s_wsum) {
  constexpr int T = NT * 4 * R4;
  constexpr int U = (R4 == 2) ? 2 : 1; /* vectors of all six columns in flight together (R4 = 4 with two spills) */
  const int32_t nbk = X.nbk;
  const int32_t slot0 = s_ref[0], b0n = s_ref[1], b0c = s_ref[2];
  const int32_t shift = X.shift, mask = X.gb - 1;
  const int64_t t0 = (int64_t)w * T + (int64_t)threadIdx.x * 4; /* this lane's first vector; the others NT * 4 apart */
  /* phase 1: every column of the tile, once.  Kept per vote: the group (-1: not in the table), the packed word, the
   * rank inside the bucket's run with the two escape bits. */
  int32_t gg[R4 * 4];
  uint32_t bw[R4 * 4], rk[R4 * 4];
  int32_t bad = 0, nesc = 0;
  int32_t chain = 0; /* always 0, but only known once the pair before has its ranks: see the end of the loop */
#pragma unroll
  for (int k0 = 0; k0 < R4; k0 += U) {
    I4 g4[U], s4[U], a4[U], m4[U], n4[U], c4[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i0 = t0 + (int64_t)(k0 + u) * NT * 4 + chain;
      g4[u] = tiles_load4<FULL>(gidx, i0, n, -1);
      s4[u] = tiles_load4<FULL>(slot, i0, n, 0);
      a4[u] = tiles_load4<FULL>(acceptor, i0, n, 0);
      m4[u] = tiles_load4<FULL>(max_cp, i0, n, 0);
      n4[u] = tiles_load4<FULL>(bnum, i0, n, 0);
      c4[u] = tiles_load4<FULL>(bcoord, i0, n, 0);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = k0 + u;
      const int64_t i0 = t0 + (int64_t)k * NT * 4;
      const int32_t g_[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w}, ss[4] = {s4[u].x, s4[u].y, s4[u].z, s4[u].w};
      const int32_t aa[4] = {a4[u].x, a4[u].y, a4[u].z, a4[u].w}, mm[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w};
      const int32_t nn[4] = {n4[u].x, n4[u].y, n4[u].z, n4[u].w}, cc[4] = {c4[u].x, c4[u].y, c4[u].z, c4[u].w};
      uint32_t stw = 0;
      /* (no branch around the LDS atomic: a vote outside the table adds 0 to counter 0, so that the four of a vector -
       * the eight of a pair - go out together instead of each waiting for the one before; round 6: the phase was a chain
       * of twelve LDS round trips per lane behind fifteen other waves' */
      int32_t av[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool in_batch = FULL || i0 + q < n;
        const bool ok = in_batch && (uint32_t)g_[q] < (uint32_t)G;
        av[q] = atomicAdd(&cnt[ok ? (g_[q] >> shift) : 0], ok ? 1 : 0);
        if (in_batch && !ok) {
          bad++;
          stw |= (uint32_t)GPX_S_NOGROUP << (8 * q); /* PaxosManager.java:1162-1194 */
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int e = k * 4 + q;
        const bool ok = (FULL || i0 + q < n) && (uint32_t)g_[q] < (uint32_t)G;
        const uint32_t dslot = (uint32_t)(ss[q] - slot0 + 128), dcp = (uint32_t)(ss[q] - 1 - mm[q] + 128);
        const bool es = dslot > 255u || dcp > 255u;
        const bool eb = nn[q] != b0n || cc[q] != b0c || (uint32_t)aa[q] > 0xffffu;
        gg[e] = ok ? g_[q] : -1;
        bw[e] = (es ? 0u : (dslot | (dcp << 8))) | (eb ? 0u : ((uint32_t)aa[q] << 16));
        rk[e] = (uint32_t)av[q] | (es ? V8_ESC_S : 0u) | (eb ? V8_ESC_B : 0u);
        nesc += (ok && es) ? 1 : 0;
      }
#ifdef GPX_TL_TRACE /* ablation bit 2: no status prefill */
      if (A.xcd_rows & 4) continue;
#endif
      if (status) { /* what k_hist does for the partition path (GPX_S_OK == 0) */
        if (FULL) {
          *(uint32_t*)(status + i0) = stw;
        } else {
          for (int q = 0; q < 4; q++)
            if (i0 + q < n) status[i0 + q] = (uint8_t)((stw >> (8 * q)) & 0xffu);
        }
      }
    }
    /* The next pair's loads stay behind this pair's ranks: the compiler otherwise hoists every load of the tile to the
     * top (branch-free code) and a 1024-thread workgroup's 128 registers spill (188-420 bytes of scratch per lane). */
    chain = (int32_t)rk[(k0 + U) * 4 - 1];
    asm volatile("v_and_b32 %0, 0, %0" : "+v"(chain));
  }
  if (bad) atomicAdd(&X.counters[2], (unsigned long long)bad);
  if (nesc) atomicAdd(s_nesc, nesc);
  if (w == 0 && threadIdx.x == 0) atomicAdd(&X.counters[0], (unsigned long long)n);
  __syncthreads();
  TL_STAMP(w, 1); /* columns loaded, votes counted */
  /* (three words per vote cross the barrier, as they are: the compiler otherwise carries phase 2's derived words along
   * as well - bucket, shifted local group, flags - and a 1024-thread workgroup's registers spill) */
#pragma unroll
  for (int e = 0; e < R4 * 4; e++) asm volatile("" : "+v"(gg[e]), "+v"(rk[e]), "+v"(bw[e]));
  /* a tile where more than one vote in 32 does not fit its byte (groups out of lock-step) carries slot and max_cp
   * of EVERY vote beside the sorted records */
  const bool wide = *s_nesc * 32 > T;
  const uint32_t wf = wide ? TL_WIDE : 0u;
  const int32_t per = (nbk + 1 + NT - 1) / NT;
  int32_t tot;
  if (per <= 4) tot = tl_publish_starts<NT, 4>(cnt, s_wsum, A, w, nbk, wf);
  else tot = tl_publish_starts<NT, 8>(cnt, s_wsum, A, w, nbk, wf);
  (void)per;
  __syncthreads();
  TL_STAMP(w, 2); /* scanned, A.off written */
  /* phase 2: the tile, sorted by bucket, as 8-byte records in LDS.  Every run start is requested before the first is
   * used (no branch between them); a vote outside the table writes nothing. */
  const uint32_t keep = wide ? V8_ESC_B : (V8_ESC_S | V8_ESC_B); /* (a wide tile's slots travel in A.ext: nothing to fetch) */
#pragma unroll
  for (int k = 0; k < R4; k++) { /* (a vector's four at a time: sixteen positions held at once spill) */
    int32_t pos[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pos[q] = cnt[max(gg[k * 4 + q], 0) >> shift];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int e = k * 4 + q;
      const int32_t g = gg[e];
      Vote8 r;
      r.a = (uint32_t)(k * NT * 4 + (int32_t)threadIdx.x * 4 + q) | ((uint32_t)(g & mask) << 14) | (rk[e] & keep);
      r.b = bw[e];
      if (g >= 0) recs[pos[q] + (int32_t)(rk[e] & TL_RK_MASK)] = r;
    }
  }
  __syncthreads();
  TL_STAMP(w, 3); /* sorted in LDS */
  /* the sorted tile leaves as it lies: 16 bytes per lane and step */
  {
    const I4* src = (const I4*)recs;
    I4* dst = (I4*)(A.recs + (int64_t)w * T);
    const int32_t nv = (tot + 1) >> 1; /* (an odd tail writes one stale record behind the tile's last: nobody reads it) */
    int32_t i = (int32_t)threadIdx.x;
    for (; i + NT < nv; i += 2 * NT) { /* two LDS reads in flight */
      const I4 x = src[i], y = src[i + NT];
      store16_through(dst + i, x);
      store16_through(dst + i + NT, y);
    }
    if (i < nv) store16_through(dst + i, src[i]);
  }
  if (wide) {
    /* (rare) slot and max_cp of every vote take the same road behind the records: the two columns once more (from L2),
     * sorted through the same LDS block, out as one run of A.ext.  (First form: 8-byte stores straight to the sorted
     * positions in A.ext - 3 M scattered stores, k_scatter_tiles 65 us instead of 30.) */
    __syncthreads();
    int2* ext = (int2*)recs;
#pragma unroll
    for (int k = 0; k < R4; k++) {
      const int64_t i0 = t0 + (int64_t)k * NT * 4;
      const I4 s4 = tiles_load4<FULL>(slot, i0, n, 0), m4 = tiles_load4<FULL>(max_cp, i0, n, 0);
      const int32_t ss[4] = {s4.x, s4.y, s4.z, s4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int e = k * 4 + q;
        if (gg[e] >= 0) ext[cnt[gg[e] >> shift] + (int32_t)(rk[e] & TL_RK_MASK)] = make_int2(ss[q], mm[q]);
      }
    }
    __syncthreads();
    const I4* src = (const I4*)recs;
    I4* dst = (I4*)(A.ext + (int64_t)w * T);
    const int32_t nv = (tot + 1) >> 1;
    for (int32_t i = (int32_t)threadIdx.x; i < nv; i += NT) store16_through(dst + i, src[i]);
  }
  TL_STAMP(w, 4); /* stores issued */
  TL_CLOCK(w, 7);
}

template <int NT, int R4>
__global__ __launch_bounds__(NT) void k_scatter_tiles(int32_t n, int32_t G, DevScratch X, TileArea A,
                                                      const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
                                                      const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot,
                                                      const int32_t* __restrict__ acceptor,
                                                      const int32_t* __restrict__ max_cp, uint8_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  constexpr int T = NT * 4 * R4;
  static_assert(T <= 16384, "14 bits of tile offset, 15 bits of run start");
  const int32_t nbk = X.nbk;
  /* dynamic LDS only (see tl_block_exscan): recs[T] | cnt[nbk] | reference vote | escapes | wave totals */
  Vote8* recs = (Vote8*)lds;
  int32_t* cnt = lds + 2 * T; /* [nbk] count -> exclusive base */
  int32_t* s_ref = cnt + tl_cnt_words(nbk, NT);
  int32_t* s_nesc_p = s_ref + 4;
  int32_t* s_wsum = s_ref + 8;
  const int32_t w = tile_of_block(A.nwg);
  if (w >= A.nwg) return;
  if (X.gate && *X.unsorted != X.epoch) return; /* a few sorted runs: k_ar_runs did it (gpx_runs.hip.h) */
  TL_STAMP(w, 0);
  TL_CLOCK(w, 6);
  for (int32_t b = threadIdx.x; b < tl_cnt_words(nbk, NT); b += NT) cnt[b] = 0;
  if (threadIdx.x < 64) { /* the batch's reference vote: what most of its first 64 votes carry (every tile finds the same) */
    const int32_t m = min(n, 64), i = min((int32_t)threadIdx.x, m - 1);
    int32_t rb, rc, rs, dummy;
    wave_majority2(bnum[i], bcoord[i], m, &rb, &rc);
    wave_majority2(slot[i], 0, m, &rs, &dummy);
    if (threadIdx.x == 0) {
      s_ref[0] = rs, s_ref[1] = rb, s_ref[2] = rc;
      *s_nesc_p = 0;
      if (w == 0) A.ref[0] = rs, A.ref[1] = rb, A.ref[2] = rc;
    }
  }
  __syncthreads();
  if ((int64_t)(w + 1) * T <= n && !((uintptr_t)status & 3))
    scatter_tile<NT, R4, true>(n, G, X, A, w, gidx, bnum, bcoord, slot, acceptor, max_cp, status, cnt, recs, s_ref, s_nesc_p, s_wsum);
  else
    scatter_tile<NT, R4, false>(n, G, X, A, w, gidx, bnum, bcoord, slot, acceptor, max_cp, status, cnt, recs, s_ref, s_nesc_p, s_wsum);
}
#define GPX_TL_LDS_BYTES(nbk, T, NT) ((size_t)tl_cnt_words(nbk, NT) * 4 + (size_t)(T) * sizeof(Vote8) + 128)
