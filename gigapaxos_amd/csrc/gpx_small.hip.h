/*
 * gpx_small.hip.h — accept-reply batches of at most 65,536 votes in ONE launch (round 2).
 *
 * The partition pipeline costs four launches over every bucket whatever the batch holds (~55 us per
 * call, measured); BASELINE config #2 (10 k groups, 30 k-vote batches) and every latency-bound caller
 * sit on that floor.  A batch this small is L2-resident, so nothing has to be partitioned through
 * HBM: workgroup w owns the groups [w * gw, (w + 1) * gw) and
 *   1. reads the whole gidx column twice (16-byte loads): counts its groups' votes, then - after a
 *      block scan - files the ARRIVAL INDEX of each vote (16 bits) in its group's LDS segment;
 *   2. one lane per group (lanes loop when gw > lanes): sorts its segment (= arrival order), replays
 *      the votes through apply_ar_group - the columns are read by index straight from L2 - exactly as
 *      PaxosInstanceStateMachine.handleBatchedAcceptReply would (PISM:1370-1419 -> PCS:597-683);
 *   3. outputs are parked per vote in scratch columns, counted per workgroup; a workgroup publishes
 *      its count and waits for the counts of the workgroups before it (they were dispatched earlier,
 *      so they are running or done: no deadlock), then writes its decisions at the right offset of
 *      the caller's columns - group-major, the output order contract of include/gpx.h.
 */
#pragma once
#include "gpx_ar16.hip.h"

#define GPX_SMALL_MAX_N 65536 /* arrival indices are filed as 16-bit words */
#define GPX_SMALL_MAX_GW 4096 /* groups per workgroup (two LDS words each) */
#define GPX_SMALL_MAX_WG 512
#define GPX_SMALL_NT 1024

struct SmallArgs {
  int32_t n, gw;             /* votes; groups per workgroup */
  const int32_t *gidx, *bnum, *bcoord, *slot, *acceptor, *max_cp;
  int32_t *d_gidx, *d_slot, *d_bnum, *d_bcoord, *d_median;
  uint8_t* d_kind;
  int32_t* n_out;
  uint8_t* status;
  Stage16 O;                    /* per-vote parking of outputs (indexed by arrival index) */
  unsigned long long* tickets;  /* [GPX_SMALL_MAX_WG] (epoch << 32) | outputs of the workgroup */
  uint32_t epoch;
};

/* one group's votes: 16-bit arrival indices in LDS, ascending after sort() */
struct SmallIter {
  uint16_t* seg;
  const int32_t *bnum, *bcoord, *slot, *acceptor, *max_cp;
  Stage16 O;
  int32_t my_bnum, my_bcoord;
  int32_t c, done, nout, cur;
  __device__ __forceinline__ bool next(Rec& out) {
    if (done >= c) return false;
    const int32_t ix = (int32_t)seg[done];
    cur = ix;
    out.idx = ix;
    out.a = slot[ix];
    out.b = acceptor[ix];
    out.c = max_cp[ix];
    out.bnum = bnum[ix];
    out.bcoord = bcoord[ix];
    done++;
    return true;
  }
  __device__ __forceinline__ void emit(int32_t slot_, int32_t x, int32_t y, int32_t z, int32_t kind) {
    O.slot()[cur] = slot_;
    O.bnum()[cur] = x;
    O.bcoord()[cur] = y;
    O.median()[cur] = z;
    O.kind()[cur] = (uint8_t)kind;
    seg[nout++] = (uint16_t)cur; /* entry nout <= done - 1: consumed */
  }
};

/* cooperative ascending bitonic sort of a[0 .. c) (16-bit keys in LDS), whole workgroup */
__device__ void sort_long_u16(uint16_t* a, uint32_t c) {
  uint32_t p2 = 1;
  while (p2 < c) p2 <<= 1;
  for (uint32_t k = 2; k <= p2; k <<= 1) {
    for (uint32_t t = threadIdx.x; t < c; t += blockDim.x) {
      const uint32_t q = t ^ (k - 1);
      if (q > t && q < c && a[t] > a[q]) {
        const uint16_t x = a[t];
        a[t] = a[q];
        a[q] = x;
      }
    }
    __syncthreads();
    for (uint32_t j = k >> 2; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < c; t += blockDim.x) {
        const uint32_t q = t ^ j;
        if (q > t && q < c && a[t] > a[q]) {
          const uint16_t x = a[t];
          a[t] = a[q];
          a[q] = x;
        }
      }
      __syncthreads();
    }
  }
}

template <int KMAX>
__global__ __launch_bounds__(GPX_SMALL_NT) void k_small_ar(DevState S, DevScratch X, SmallArgs A) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int32_t gw = A.gw, n = A.n;
  const int32_t w = (int32_t)blockIdx.x;
  const int32_t g0 = w * gw;
  int32_t* lcnt = lds;           /* [gw] votes of each of my groups; later: outputs */
  int32_t* lcur = lds + gw;      /* [gw] segment cursor / end */
  uint16_t* glist = (uint16_t*)(lds + 2 * gw);       /* [gw] my groups that received votes, ascending */
  uint16_t* seg = glist + ((gw + 7) & ~7);           /* [n] arrival indices, group-major */
  __shared__ int32_t s_any_long, s_tot;
  const int32_t nt = (int32_t)blockDim.x;
  for (int32_t l = threadIdx.x; l < gw; l += nt) lcnt[l] = 0;
  if (threadIdx.x == 0) s_any_long = 0;
  __syncthreads();
  /* pass 1: count.  The status of a vote is written by the workgroup that owns its group (pass 2:
   * GPX_S_OK, overwritten by the replay for a missing / stopped group - same workgroup, ordered by a
   * barrier); a vote whose index is out of range has no owner: workgroup 0 marks it */
  const bool vec = !((uintptr_t)A.gidx & 15);
  int32_t bad = 0;
  for (int32_t i0 = (int32_t)threadIdx.x * 4; i0 < n; i0 += nt * 4) {
    int32_t g[4];
    if (vec && i0 + 3 < n) {
      const I4 v = *(const I4*)(A.gidx + i0);
      g[0] = v.x, g[1] = v.y, g[2] = v.z, g[3] = v.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) g[q] = i0 + q < n ? A.gidx[i0 + q] : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (i0 + q >= n) continue;
      const uint32_t lg = (uint32_t)(g[q] - g0);
      const bool inr = (uint32_t)g[q] < (uint32_t)S.G;
      if (lg < (uint32_t)gw && inr) atomicAdd(&lcnt[lg], 1);
      if (w == 0 && !inr) { /* PaxosManager.java:1162-1194 */
        if (A.status) A.status[i0 + q] = GPX_S_NOGROUP;
        bad++;
      }
    }
  }
  if (w == 0) {
    if (bad) atomicAdd(&X.counters[2], (unsigned long long)bad);
    if (threadIdx.x == 0) atomicAdd(&X.counters[0], (unsigned long long)n);
  }
  __syncthreads();
  /* scan: thread t owns `per` consecutive groups */
  const int32_t per = (gw + nt - 1) / nt;
  const int32_t l0 = (int32_t)threadIdx.x * per;
  int32_t mine = 0;
  for (int32_t q = 0; q < per; q++)
    if (l0 + q < gw) mine += lcnt[l0 + q];
  int32_t tot;
  int32_t ex = block_exscan_n<GPX_SMALL_NT>(mine, &tot);
  for (int32_t q = 0; q < per; q++)
    if (l0 + q < gw) {
      lcur[l0 + q] = ex;
      ex += lcnt[l0 + q];
      if (lcnt[l0 + q] > 16) s_any_long = 1;
    }
  __syncthreads();
  /* pass 2: file the arrival indices */
  if (tot)
    for (int32_t i0 = (int32_t)threadIdx.x * 4; i0 < n; i0 += nt * 4) {
      int32_t g[4];
      if (vec && i0 + 3 < n) {
        const I4 v = *(const I4*)(A.gidx + i0);
        g[0] = v.x, g[1] = v.y, g[2] = v.z, g[3] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) g[q] = i0 + q < n ? A.gidx[i0 + q] : 0;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (i0 + q >= n) continue;
        const uint32_t lg = (uint32_t)(g[q] - g0);
        if (lg < (uint32_t)gw && (uint32_t)g[q] < (uint32_t)S.G) {
          seg[atomicAdd(&lcur[lg], 1)] = (uint16_t)(i0 + q);
          if (A.status) A.status[i0 + q] = GPX_S_OK;
        }
      }
    }
  __syncthreads();
  /* long segments: cooperative sort (a hot group is serial by contract; this only has to be right) */
  if (s_any_long) {
    for (int32_t l = 0; l < gw; l++) {
      const int32_t c = lcnt[l]; /* uniform */
      if (c > 16) sort_long_u16(seg + (lcur[l] - c), (uint32_t)c); /* lcur = segment end here */
    }
    __syncthreads(); /* the replay below rewrites lcnt / lcur of its own groups */
  }
  /* the groups that received votes, compacted in group order: a batch this small touches few of a
   * workgroup's groups, and every lane should replay one of them rather than walk its own empty ones */
  int32_t ne = 0;
  for (int32_t q = 0; q < per; q++)
    if (l0 + q < gw && lcnt[l0 + q] != 0) ne++;
  int32_t nne;
  int32_t nex = block_exscan_n<GPX_SMALL_NT>(ne, &nne);
  for (int32_t q = 0; q < per; q++)
    if (l0 + q < gw && lcnt[l0 + q] != 0) glist[nex++] = (uint16_t)(l0 + q);
  __syncthreads();
  /* replay: lane t owns the touched groups k0 .. k1 - 1 (usually one) */
  const int32_t chunk = (nne + nt - 1) / nt;
  const int32_t k0 = (int32_t)threadIdx.x * chunk, k1 = min(nne, k0 + chunk);
  int32_t my_out = 0;
  for (int32_t k = k0; k < k1; k++) {
    const int32_t l = (int32_t)glist[k];
    const int32_t c = lcnt[l];
    int32_t nout = 0;
    {
      const int32_t g = g0 + l;
      const int32_t start = lcur[l] - c; /* the cursor ran to the segment's end during pass 2 */
      lcur[l] = start;                   /* from here on: the segment's start (read again when the outputs leave) */
      uint16_t* sg = seg + start;
      if (c <= 16)
        for (int32_t i = 1; i < c; i++) { /* insertion sort, this lane only */
          const uint16_t x = sg[i];
          int32_t p = i - 1;
          while (p >= 0 && sg[p] > x) {
            sg[p + 1] = sg[p];
            p--;
          }
          sg[p + 1] = x;
        }
      SmallIter it;
      it.seg = sg;
      it.bnum = A.bnum;
      it.bcoord = A.bcoord;
      it.slot = A.slot;
      it.acceptor = A.acceptor;
      it.max_cp = A.max_cp;
      it.O = A.O;
      it.c = c;
      it.done = 0;
      it.nout = 0;
      it.cur = 0;
      CoordPre<KMAX> P;
      coord_preload<KMAX>(S, g, P);
      coord_preload_ring<KMAX>(S, g, P);
      apply_ar_group<KMAX>(S, X, g, it, A.status, P);
      nout = it.nout;
    }
    lcnt[l] = nout; /* the group's vote count is no longer needed: now its output count */
    my_out += nout;
  }
  int32_t wtot;
  int32_t oex = block_exscan_n<GPX_SMALL_NT>(my_out, &wtot);
  /* publish this workgroup's count, collect the counts of the workgroups before it */
  if (threadIdx.x == 0) {
    __hip_atomic_store(&A.tickets[w], ((unsigned long long)A.epoch << 32) | (uint32_t)wtot, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_AGENT);
    s_tot = 0;
  }
  __syncthreads();
  int32_t before = 0;
  for (int32_t t = threadIdx.x; t < w; t += nt) {
    unsigned long long v;
    do {
      v = __hip_atomic_load(&A.tickets[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    } while ((uint32_t)(v >> 32) != A.epoch);
    before += (int32_t)(uint32_t)v;
  }
  if (before) atomicAdd(&s_tot, before);
  __syncthreads();
  const int32_t base = s_tot;
  int32_t o = base + oex;
  for (int32_t k = k0; k < k1; k++) {
    const int32_t l = (int32_t)glist[k];
    const int32_t nout = lcnt[l];
    const uint16_t* sg = seg + lcur[l]; /* the outputs' votes are listed from the segment's first word on */
    for (int32_t q = 0; q < nout; q++) {
      const int32_t ix = (int32_t)sg[q];
      A.d_gidx[o] = g0 + l;
      A.d_slot[o] = A.O.slot()[ix];
      A.d_bnum[o] = A.O.bnum()[ix];
      A.d_bcoord[o] = A.O.bcoord()[ix];
      A.d_median[o] = A.O.median()[ix];
      A.d_kind[o] = A.O.kind()[ix];
      o++;
    }
  }
  if (w == (int32_t)gridDim.x - 1 && threadIdx.x == 0) {
    if (A.n_out) *A.n_out = base + wtot;
    atomicAdd(&X.counters[1], (unsigned long long)(base + wtot));
  }
}
