/*
 * gpx_slots.hip.h — the SLOTTED front end of the shuffled accept-reply call (round 5).
 *
 * The partition front end (k_hist + k_scatter_ar16) writes every vote as ONE 16-byte record to an effectively random
 * address: 3 M stores that each cost a 32-byte sector (95 MB of WRITE_SIZE for 48 MB of records) at the chip's
 * random-access rate, ~50 us whatever is done around them, after a histogram pass whose 0.29 M returning atomics cost
 * another ~16 us.  Round 4's micro-benchmark (scripts/ubench/ubench_front16.hip, variant D) showed the form whose
 * writes reach HBM as whole 64-byte lines; this is that form inside the engine:
 *
 *   k_scatter_slots<T>   one workgroup SORTS T votes by bucket in LDS (counting sort on 8-byte records: offset in the
 *                        tile | local group | escape, and slot / max_cp / acceptor as bytes relative to vote 0's) and
 *                        writes the run of every bucket into a FIXED slot of GPX_SL_SLOT records that belongs to
 *                        (bucket, workgroup): no histogram pass, no reservation, no atomics on shared words; eight
 *                        lanes write a run, so it leaves as one to three full lines.  A count byte per (bucket,
 *                        workgroup) says how many records the slot holds; a run longer than the slot puts its tail
 *                        on the overflow list, in the workgroup's own segment of it (a stream like BASELINE's has
 *                        a record or two there per call - Poisson mean 8.4 against 24 -, a stream sorted by group
 *                        nearly all its votes: see SlotArea).  Also what k_hist did on the side: status prefill, the
 *                        vote and dropped counters.
 *   k_slot_totals        row sums of the count matrix (+ overflow per bucket) -> X.bucket_tot; the per-bucket kernel sums
 *                        what lies before its bucket and from there on everything - its regions in X.rec / X.perm / the
 *                        output staging, k_emit_dec16 - is what it was.
 *   k_bucket16<.., SLOTS> (gpx_ar16.hip.h) reads its records from its nwg slots (eight lanes per slot) instead of one
 *                        contiguous region: twice (count, then place - the second time from L2); a bucket too big for
 *                        the LDS staging first copies them into its X.rec region and goes on as before.
 *
 * Taken by a call that is one partition pass over 820 ... 4096 buckets with at most 192 scatter workgroups (3.1 M
 * votes over 1 M groups, 1.5 M over 500,000), 16-byte aligned columns, group sizes up to 5; GPX_AR_SLOTS=0 at engine
 * creation keeps the partition front end for every call.  Measured (BASELINE config #3, bench.py): 0.103 against 0.128 ms
 * per step - DESIGN.md 3.
 */
#pragma once
#include "gpx_kernels.hip.h"

#define GPX_SL_SLOT 24 /* records per (bucket, workgroup) slot: three 64-byte lines */
#define GPX_SL_MAXWG 192 /* scatter workgroups of a call at most (the engine routes larger calls elsewhere) */
#define GPX_SL_SCAN 2048 /* an overflow list up to this long is read whole by the buckets that have records on it */
struct __attribute__((aligned(8))) Vote8 {
  uint32_t a; /* offset of the vote in its workgroup's tile (14 bits) | local group << 14 (10 bits) | ESC << 31 */
  uint32_t b; /* slot - slot0 + 128 (8 bits) | (slot - 1 - max_cp + 128) << 8 (8 bits) | acceptor << 16 */
};
#define V8_ESC 0x80000000u
struct SlotArea {
  Vote8* slots;     /* [nbk][nwg][GPX_SL_SLOT] */
  uint8_t* cntm;    /* [nbk][nwg_pad] records in each slot (at most GPX_SL_SLOT: the rest overflowed) */
  /* The overflow list: ONE segment per scatter workgroup (reserved with one atomic, the records of a segment in bucket
   * order: the tails of the tile's sorted runs), so that a bucket finds its records with a binary search per segment
   * and a stream that is NOT shuffled - sorted by group, a few ascending runs: nearly every vote overflows - costs
   * in proportion to its votes.  (The first form appended record by record and every bucket scanned the whole list:
   * 10.5 ms per step on bench.py --sorted, the scatter 0.93 ms of it in the list's counter.) */
  I4* ovf_rec;      /* [max_batch] overflow records, as Vote16 */
  int32_t* ovf_bkt; /* [max_batch] their buckets (ascending inside a segment) */
  int32_t* ovf_seg; /* [2][GPX_SL_MAXWG] where workgroup w's segment starts; its length */
  int32_t* ovf_n;   /* [2] overflow records of the call (k_scatter_slots adds; k_slot_totals moves it to word 1 and clears) */
  int32_t* ovf_cnt; /* [nbk] overflow records per bucket (cleared by the per-bucket kernel) */
  int32_t nwg, nwg_pad, tile; /* scatter workgroups of this call, the count matrix's row stride, votes per workgroup */
};

/* the 16-byte record of a slot entry; w = the scatter workgroup that wrote it.  An escaped entry (another ballot than
 * the batch's, an acceptor id beyond 16 bits, a slot or checkpoint that does not fit a byte next to vote 0's) fetches
 * slot and max_cp from the caller's columns here and keeps ESC, so that the ballot and the acceptor are fetched where
 * the 16-byte records' escape path fetches them */
__device__ __forceinline__ I4 slot_expand(const Vote8 v, int32_t w, int32_t tile, int32_t slot0,
                                          const int32_t* __restrict__ slot_col, const int32_t* __restrict__ maxcp_col) {
  I4 r;
  r.x = w * tile + (int32_t)(v.a & 0x3fffu);
  const uint32_t lg = (v.a >> 14) & 0x3ffu;
  if (v.a & V8_ESC) {
    r.y = slot_col[r.x];
    r.z = maxcp_col[r.x];
    r.w = (int32_t)(lg | 0x4000u /* V16_ESC */);
  } else {
    r.y = slot0 + (int32_t)(v.b & 255u) - 128;
    r.z = r.y - 1 - ((int32_t)((v.b >> 8) & 255u) - 128);
    r.w = (int32_t)(lg | (v.b & 0xffff0000u));
  }
  return r;
}

/* four consecutive entries of a column; FULL: the whole vector is inside the batch (one 16-byte load) */
template <bool FULL>
__device__ __forceinline__ I4 slots_load4(const int32_t* __restrict__ col, int64_t i0, int32_t n, int32_t fill) {
  if (FULL) return *(const I4*)(col + i0);
  I4 r;
  r.x = i0 + 0 < n ? col[i0 + 0] : fill;
  r.y = i0 + 1 < n ? col[i0 + 1] : fill;
  r.z = i0 + 2 < n ? col[i0 + 2] : fill;
  r.w = i0 + 3 < n ? col[i0 + 3] : fill;
  return r;
}
/* FULL: every vector of the tile lies inside the batch and the status column takes 4-byte stores - every workgroup but
 * the last.  Then ALL of a phase's loads go out before the first is used (round 5, second form: the first had the
 * compiler wait for each of the four gidx vectors, and each of the four sets of columns, in turn - eight round trips
 * where two or three do - and read the gidx column twice). */
template <int T, bool FULL>
__device__ __forceinline__ void scatter_slots_tile(int32_t n, int32_t G, const DevScratch& X, const SlotArea& A, int32_t w,
                                                   const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
                                                   const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot,
                                                   const int32_t* __restrict__ acceptor, const int32_t* __restrict__ max_cp,
                                                   uint8_t* __restrict__ status, int32_t* cnt, Vote8* recs) {
  constexpr int R4 = T / (GPX_FBLOCK * 4);
  const int32_t nbk = X.nbk;
  const int32_t b0n = bnum[0], b0c = bcoord[0], slot0 = slot[0];
  const int32_t shift = X.shift, mask = X.gb - 1;
  const int64_t t0 = (int64_t)w * T + (int64_t)threadIdx.x * 4; /* this lane's first vector; the others GPX_FBLOCK * 4 apart */
  /* phase 1: the group indices (kept for phase 2; an index outside the table becomes -1), ranks inside the buckets */
  int32_t gg[R4 * 4], rk[R4 * 4];
  {
    I4 gq[R4];
#pragma unroll
    for (int k = 0; k < R4; k++) gq[k] = slots_load4<FULL>(gidx, t0 + (int64_t)k * GPX_FBLOCK * 4, n, -1);
    int32_t bad = 0;
#pragma unroll
    for (int k = 0; k < R4; k++) {
      const int64_t i0 = t0 + (int64_t)k * GPX_FBLOCK * 4;
      gg[k * 4 + 0] = gq[k].x, gg[k * 4 + 1] = gq[k].y, gg[k * 4 + 2] = gq[k].z, gg[k * 4 + 3] = gq[k].w;
      uint32_t stw = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        rk[k * 4 + q] = 0;
        if (FULL || i0 + q < n) {
          if ((uint32_t)gg[k * 4 + q] < (uint32_t)G) {
            rk[k * 4 + q] = atomicAdd(&cnt[gg[k * 4 + q] >> shift], 1);
          } else {
            gg[k * 4 + q] = -1;
            bad++;
            stw |= (uint32_t)GPX_S_NOGROUP << (8 * q); /* PaxosManager.java:1162-1194 */
          }
        } else {
          gg[k * 4 + q] = -1;
        }
      }
      if (status) { /* what k_hist does for the partition path (GPX_S_OK == 0) */
        if (FULL) {
          *(uint32_t*)(status + i0) = stw;
        } else {
          for (int q = 0; q < 4; q++)
            if (i0 + q < n) status[i0 + q] = (uint8_t)((stw >> (8 * q)) & 0xffu);
        }
      }
    }
    if (bad) atomicAdd(&X.counters[2], (unsigned long long)bad);
  }
  if (w == 0 && threadIdx.x == 0) atomicAdd(&X.counters[0], (unsigned long long)n);
  __syncthreads();
  /* exclusive scan of the counts -> local bases; this workgroup's column of the count matrix */
  const int32_t per = (nbk + GPX_FBLOCK - 1) / GPX_FBLOCK; /* <= 4: at most 4096 buckets */
  const int32_t bq = (int32_t)threadIdx.x * per;
  /* (two scans in one word: the records before a bucket in the low half, of those the ones beyond their slots - the
   * bucket's place in this workgroup's overflow segment - in the high half; a tile has at most 16,384 of either) */
  int32_t v[4], s = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    v[q] = (q < per && bq + q < nbk) ? cnt[bq + q] : 0;
    s += v[q] + (max(v[q] - GPX_SL_SLOT, 0) << 16);
  }
  int32_t tot2;
  int32_t ex = block_exscan_n<GPX_FBLOCK>(s, &tot2);
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (q < per && bq + q < nbk) {
      cnt[bq + q] = ex;
      ex += v[q] + (max(v[q] - GPX_SL_SLOT, 0) << 16);
      A.cntm[(int64_t)(bq + q) * A.nwg_pad + w] = (uint8_t)min(v[q], GPX_SL_SLOT);
    }
  const int32_t tot = tot2 & 0xffff, wovf = tot2 >> 16;
  __shared__ int32_t s_seg0;
  if (threadIdx.x == 0) { /* this workgroup's segment of the overflow list (empty on a shuffled stream) */
    const int32_t p0 = wovf ? atomicAdd(A.ovf_n, wovf) : 0;
    A.ovf_seg[w] = p0;
    A.ovf_seg[GPX_SL_MAXWG + w] = wovf;
    s_seg0 = p0;
  }
  __syncthreads();
  /* phase 2: the tile, sorted by bucket, as 8-byte records in LDS; two vectors' worth of columns in flight at a time */
#pragma unroll
  for (int k0 = 0; k0 < R4; k0 += 2) {
    I4 s4[2], a4[2], m4[2], n4[2], c4[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int64_t i0 = t0 + (int64_t)(k0 + u) * GPX_FBLOCK * 4;
      s4[u] = slots_load4<FULL>(slot, i0, n, 0);
      a4[u] = slots_load4<FULL>(acceptor, i0, n, 0);
      m4[u] = slots_load4<FULL>(max_cp, i0, n, 0);
      n4[u] = slots_load4<FULL>(bnum, i0, n, 0);
      c4[u] = slots_load4<FULL>(bcoord, i0, n, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int k = k0 + u;
      const int64_t i0 = t0 + (int64_t)k * GPX_FBLOCK * 4;
      const int32_t ss[4] = {s4[u].x, s4[u].y, s4[u].z, s4[u].w}, aa[4] = {a4[u].x, a4[u].y, a4[u].z, a4[u].w};
      const int32_t mm[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w}, nn[4] = {n4[u].x, n4[u].y, n4[u].z, n4[u].w};
      const int32_t cc[4] = {c4[u].x, c4[u].y, c4[u].z, c4[u].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int32_t g = gg[k * 4 + q];
        if (g < 0) continue;
        const uint32_t dslot = (uint32_t)(ss[q] - slot0 + 128), dcp = (uint32_t)(ss[q] - 1 - mm[q] + 128);
        const bool esc = nn[q] != b0n || cc[q] != b0c || (uint32_t)aa[q] > 0xffffu || dslot > 255u || dcp > 255u;
        Vote8 r;
        r.a = (uint32_t)(i0 + q - (int64_t)w * T) | ((uint32_t)(g & mask) << 14) | (esc ? V8_ESC : 0u);
        r.b = esc ? 0u : (dslot | (dcp << 8) | ((uint32_t)aa[q] << 16));
        recs[(cnt[g >> shift] & 0xffff) + rk[k * 4 + q]] = r;
      }
    }
  }
  __syncthreads();
  /* every bucket's run leaves in order: eight lanes per bucket, a 64-byte line per step */
  const int32_t seg0 = s_seg0;
  for (int32_t b = (int32_t)threadIdx.x >> 3; b < nbk; b += GPX_FBLOCK >> 3) {
    const int32_t base = cnt[b] & 0xffff;
    const int32_t c = (b + 1 < nbk ? cnt[b + 1] & 0xffff : tot) - base;
    Vote8* sl = A.slots + ((int64_t)b * A.nwg + w) * GPX_SL_SLOT;
    for (int32_t j = (int32_t)threadIdx.x & 7; j < c; j += 8) {
      if (j < GPX_SL_SLOT) {
        sl[j] = recs[base + j];
      } else { /* the slot is full: the tail of the run goes to this workgroup's segment of the overflow list */
        const int32_t p = seg0 + (cnt[b] >> 16) + j - GPX_SL_SLOT;
        A.ovf_rec[p] = slot_expand(recs[base + j], w, T, slot0, slot, max_cp);
        A.ovf_bkt[p] = b;
        if (j == GPX_SL_SLOT) atomicAdd(&A.ovf_cnt[b], c - GPX_SL_SLOT);
      }
    }
  }
}

template <int T>
__global__ __launch_bounds__(GPX_FBLOCK) void k_scatter_slots(int32_t n, int32_t G, DevScratch X, SlotArea A,
                                                             const int32_t* __restrict__ gidx,
                                                             const int32_t* __restrict__ bnum,
                                                             const int32_t* __restrict__ bcoord,
                                                             const int32_t* __restrict__ slot,
                                                             const int32_t* __restrict__ acceptor,
                                                             const int32_t* __restrict__ max_cp,
                                                             uint8_t* __restrict__ status) {
  extern __shared__ int32_t lds[];
  static_assert(T <= 16384 && T % (GPX_FBLOCK * 8) == 0, "14 bits of tile offset; pairs of vectors per lane");
  const int32_t nbk = X.nbk;
  int32_t* cnt = lds; /* [nbk] count -> exclusive base */
  Vote8* recs = (Vote8*)(lds + ((nbk + 3) & ~3));
  const int32_t w = tile_of_block(A.nwg);
  if (w >= A.nwg) return;
  if (X.gate && *X.unsorted != X.epoch) return; /* a few sorted runs: k_ar_runs did it (gpx_runs.hip.h) */
  for (int32_t b = threadIdx.x; b < nbk; b += GPX_FBLOCK) cnt[b] = 0;
  __syncthreads();
  if ((int64_t)(w + 1) * T <= n && !((uintptr_t)status & 3))
    scatter_slots_tile<T, true>(n, G, X, A, w, gidx, bnum, bcoord, slot, acceptor, max_cp, status, cnt, recs);
  else
    scatter_slots_tile<T, false>(n, G, X, A, w, gidx, bnum, bcoord, slot, acceptor, max_cp, status, cnt, recs);
}

/* row sums of the count matrix (+ the buckets' overflow) -> X.bucket_tot, and per GPX_SL_ROWS buckets -> X.tile_rel
 * (free in this mode: no k_hist ran): the per-bucket kernel adds up what lies before its bucket itself.  Sixteen lanes
 * per row, GPX_SL_ROWS rows per workgroup.  (Round 5's first form was ONE workgroup that also scanned the totals into
 * X.bucket_off: 29 us for 360 KB - every thread walked two rows word by word.) */
#define GPX_SL_ROWS 16
__global__ __launch_bounds__(256) void k_slot_totals(DevScratch X, SlotArea A) {
  if (X.gate && *X.unsorted != X.epoch) return;
  __shared__ int32_t s_row[GPX_SL_ROWS];
  const int32_t b = (int32_t)blockIdx.x * GPX_SL_ROWS + ((int32_t)threadIdx.x >> 4);
  const int32_t j = (int32_t)threadIdx.x & 15;
  uint32_t acc = 0; /* two 16-bit sums; a byte is at most GPX_SL_SLOT */
  if (b < X.nbk) {
    const uint32_t* row = (const uint32_t*)(A.cntm + (int64_t)b * A.nwg_pad);
    const int32_t full = A.nwg >> 2, rem = A.nwg & 3;
    for (int32_t t = j; t < full; t += 16) {
      const uint32_t x = row[t];
      acc += (x & 0x00ff00ffu) + ((x >> 8) & 0x00ff00ffu);
    }
    if (rem && j == (full & 15)) { /* bytes at and beyond nwg may be an earlier, larger call's */
      const uint32_t x = row[full] & ((1u << (8 * rem)) - 1u);
      acc += (x & 0x00ff00ffu) + ((x >> 8) & 0x00ff00ffu);
    }
  }
  int32_t v = (int32_t)((acc & 0xffffu) + (acc >> 16));
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, 16);
  if (j == 0) {
    if (b < X.nbk) {
      v += A.ovf_cnt[b];
      X.bucket_tot[b] = v;
    } else {
      v = 0;
    }
    s_row[threadIdx.x >> 4] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t t = 0;
    for (int q = 0; q < GPX_SL_ROWS; q++) t += s_row[q];
    X.tile_rel[blockIdx.x] = t;
    if (blockIdx.x == 0) {
      A.ovf_n[1] = A.ovf_n[0]; /* the list's length, for the per-bucket kernel */
      A.ovf_n[0] = 0;          /* ... and the next call's scatter starts an empty one */
    }
  }
}
/* records before bucket b and in it (every thread of the per-bucket workgroup calls it; gb threads) */
__device__ __forceinline__ int32_t slot_bucket_offset(const DevScratch& X, int32_t b, int32_t* nb) {
  int32_t s = 0;
  const int32_t wfull = b / GPX_SL_ROWS;
  for (int32_t t = (int32_t)threadIdx.x; t < wfull; t += (int32_t)blockDim.x) s += X.tile_rel[t];
  const int32_t r = wfull * GPX_SL_ROWS + (int32_t)threadIdx.x;
  if ((int32_t)threadIdx.x < GPX_SL_ROWS && r < b) s += X.bucket_tot[r];
  int32_t tot;
  block_exscan_rt(s, &tot);
  *nb = X.bucket_tot[b];
  return tot;
}
