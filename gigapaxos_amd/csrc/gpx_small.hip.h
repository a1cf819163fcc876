/*
 * gpx_small.hip.h — a TINY accept-reply call (at most 1,024 votes, any order) in ONE launch of ONE workgroup (round 4).
 *
 * The reference's frames carry at most 2,048 slots (BatchedAcceptReply.java:27) and a coordinator that is not
 * saturated drains part of a frame per call (PaxosPacketBatcher.java:182-209).  Through the partition pipeline
 * (k_hist -> k_scatter_ar16 -> k_bucket_ar16 -> k_emit_dec16) such a call costs 24-28 us on one MI355X whatever it
 * holds (profiles/r04_tiny_calls_vs_pipeline.json): four launches that each wait for two or three dependent memory
 * round trips of 2-3 us - next to nothing is moved.  What a call cannot do without is TWO round trips: its votes,
 * then the state of the groups they name.  This kernel is that chain and nothing else (16-19 us per call):
 *
 *   load      thread t reads vote t - all six columns, coalesced, in flight together.
 *   regroup   one LANE per `width` = ceil(G / 1024) consecutive groups: count per lane (LDS atomics), scan,
 *             placement lane-major in LDS (structure of arrays, 16 bytes per vote as in gpx_ar16.hip.h; the first
 *             word is KEY = group offset inside the lane << 17 | arrival index, so one unsigned compare orders a
 *             lane's votes by (group, arrival)).  Up to 16 votes per lane: order = a nibble word in a register; up to
 *             96: ranked by the lane itself; more (a hot group): the workgroup's bitonic sort - the three regimes
 *             of bucket16_body.
 *   replay    the lanes publish their votes in (group, arrival) order and every GROUP gets a thread of its own (in
 *             a thin batch over a large table a lane's votes belong to different groups: replayed by the lane one
 *             after the other, every group's two round trips would queue up); a lane with more than 16 votes keeps
 *             them and replays them itself.  The replay is apply_ar_group (gpx_kernels.hip.h), unchanged:
 *             PISM.handleAcceptReply -> PaxosCoordinatorState.handleAcceptReplyMyBallot / HigherBallot
 *             (PaxosInstanceStateMachine.java:1248-1419, PaxosCoordinatorState.java:597-683); the ring entry of a
 *             group's first vote's slot is requested together with the group's state.  No status prefill pass:
 *             the thread that loads a vote marks it.
 *   outputs   parked in the vote's own LDS words, counted per sorted position, scanned, written straight into
 *             the caller's columns: grouped by gidx ascending, a group's entries in arrival order (include/gpx.h
 *             ORDER), and the count.
 *
 * Round 4 also built this for up to 131,072 votes on up to 128 workgroups (every workgroup scanning the whole
 * gidx column for the votes of its group range, tickets for the output base, passes and arrival-index windows for
 * skewed batches): bit-exact, and no faster than the pipeline - a tie up to 32,768 votes on a 1 M-group table,
 * slower beyond and on small tables.  The timelines (profiles/r04_sar_trace_1 .. 4.txt) say why: that kernel's chain
 * was scan -> gather -> state -> tickets, four to six round trips like the pipeline's, and beyond 32 k votes the
 * redundant scans and the random gathers cost more than launches do.  Kept as scripts/experiments/
 * r04_ar_small_multi_workgroup.patch; the numbers in profiles/r04_small_calls_one_launch_vs_pipeline.json.  One
 * workgroup with two votes per thread (2,048 votes) was measured too: a thread then replays two groups one after the
 * other, the second group's loads queue behind the first's stores, and a thin batch of 2,048 votes over a 1 M-group
 * table took 48 us against the pipeline's 30 (profiles/r04_tiny_calls_2048_vs_pipeline.json) - so: one vote per thread.
 *
 * Results are identical to the partition pipeline's (tests/test_small_ar_gpu.py: both paths against the oracle).
 */
#pragma once
#include "gpx_ar16.hip.h"

#define GPX_SAR_MAX_N 1024        /* votes per call on this path: one per thread */
#define GPX_SAR_MAX_G (1 << 24)   /* groups in the table: a lane's group offset stays below 2^14 (KEY's high bits) */
#define GPX_SAR_IDX_BITS 17
#define GPX_SAR_IDX_MASK ((1u << GPX_SAR_IDX_BITS) - 1u)
#define GPX_SAR_BLOCK 1024
#define GPX_SAR_VPT 1                               /* votes per thread */
#define GPX_SAR_CAP (GPX_SAR_BLOCK * GPX_SAR_VPT)
#define GPX_SAR_LDS_BYTES ((2 * GPX_SAR_BLOCK + 9 * GPX_SAR_CAP) * 4)
#define GPX_SAR_LONG 0xffffffffu /* gA: this sorted position belongs to a lane that replays its votes itself */

/* One group's votes in arrival order, with GroupIter's interface (next / emit) for apply_ar_group.  Ranks
 * [done, c) are the group's votes; rank r's position in the LDS arrays is
 *   perm: ranks[r] (the balanced replay: one thread per group, the lanes published their sorted positions),
 *   else: the low word of keys[r], the lane's sorted keys (key << 32 | position) in global scratch (a lane with
 *         more than 16 votes replays its own, group by group; its outputs' positions come back in keys[0 .. nout)). */
struct SmallArIter {
  int32_t* keyA;
  int32_t *slotA, *cpA;
  uint32_t* metaA;
  int32_t *xA, *yA;
  const int32_t* ranks;     /* perm: positions of this group's votes, by rank */
  unsigned long long* keys; /* !perm: the lane's sorted keys, by rank */
  VoteCols in;
  int32_t b0n, b0c;
  int32_t c, done, nout, relg;
  bool perm;
  uint32_t omask;
  uint32_t cur;
  __device__ __forceinline__ bool next(Rec& out) {
    if (done >= c) return false;
    const uint32_t p = perm ? (uint32_t)ranks[done] : (uint32_t)keys[done];
    const int32_t ix = (int32_t)((uint32_t)keyA[p] & GPX_SAR_IDX_MASK);
    const uint32_t meta = metaA[p];
    cur = p;
    out.idx = ix;
    out.a = slotA[p];
    out.c = cpA[p];
    if (meta & V16_ESC) { /* another ballot than the batch's common one, or a node id beyond 16 bits */
      out.b = in.acceptor[ix];
      out.bnum = in.bnum[ix];
      out.bcoord = in.bcoord[ix];
    } else {
      out.b = (int32_t)(meta >> 16);
      out.bnum = b0n;
      out.bcoord = b0c;
    }
    done++;
    return true;
  }
  /* output of the CURRENT vote, parked in the vote's own words; its key word now names the group (relative to the
   * pass's first group): the vote is consumed, nobody needs its arrival index any more */
  __device__ __forceinline__ void emit(int32_t slot, int32_t x, int32_t y, int32_t z, int32_t kind) {
    slotA[cur] = slot;
    cpA[cur] = z;
    metaA[cur] = (uint32_t)kind;
    xA[cur] = x;
    yA[cur] = y;
    keyA[cur] = relg;
    if (perm)
      omask |= 1u << (done - 1);
    else
      keys[nout++] = cur; /* entry nout <= done - 1: consumed */
  }
};

template <int KMAX>
__global__ __launch_bounds__(GPX_SAR_BLOCK) void k_ar_tiny(
    DevState S, DevScratch X, int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot, const int32_t* __restrict__ acceptor,
    const int32_t* __restrict__ max_cp, int32_t* __restrict__ d_gidx, int32_t* __restrict__ d_slot,
    int32_t* __restrict__ d_bnum, int32_t* __restrict__ d_bcoord, int32_t* __restrict__ d_median,
    uint8_t* __restrict__ d_kind, int32_t* __restrict__ n_out, uint8_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int32_t t = (int32_t)threadIdx.x;
  SAR_STAMP(0, 0);
  int32_t* lcnt = lds;
  int32_t* lcur = lds + GPX_SAR_BLOCK;
  int32_t* keyA = lcur + GPX_SAR_BLOCK;
  int32_t* slotA = keyA + GPX_SAR_CAP;
  int32_t* cpA = slotA + GPX_SAR_CAP;
  uint32_t* metaA = (uint32_t*)(cpA + GPX_SAR_CAP);
  int32_t* xA = (int32_t*)(metaA + GPX_SAR_CAP);
  int32_t* yA = xA + GPX_SAR_CAP;
  int32_t* permA = yA + GPX_SAR_CAP;               /* sorted position -> position in the arrays above */
  uint32_t* gA = (uint32_t*)(permA + GPX_SAR_CAP); /* sorted position -> group, or LONG */
  uint32_t* outA = gA + GPX_SAR_CAP;               /* sorted position of a group's first vote -> its outputs */
  const int32_t G = S.G;
  const uint32_t width = ((uint32_t)G + GPX_SAR_BLOCK - 1) / GPX_SAR_BLOCK; /* groups per lane */
  const VoteCols in{bnum, bcoord, acceptor};
  unsigned long long* keysG = X.perm;

  /* ---- load: the vote's six columns in flight together (the index is clamped: a thread without a vote loads
   * somebody's and drops it), then A count per lane, B scan, C placement lane-major ---- */
  lcnt[t] = 0;
  int32_t vl[GPX_SAR_VPT], vk[GPX_SAR_VPT], vs[GPX_SAR_VPT], vc[GPX_SAR_VPT];
  uint32_t vm[GPX_SAR_VPT];
  int32_t bad = 0;
  {
    const int32_t b0n = bnum[0], b0c = bcoord[0];
    int32_t vg[GPX_SAR_VPT], va[GPX_SAR_VPT], vbn[GPX_SAR_VPT], vbc[GPX_SAR_VPT];
#pragma unroll
    for (int j = 0; j < GPX_SAR_VPT; j++) {
      const int32_t i = min(j * GPX_SAR_BLOCK + t, n - 1);
      vg[j] = gidx[i], vs[j] = slot[i], vc[j] = max_cp[i], va[j] = acceptor[i], vbn[j] = bnum[i], vbc[j] = bcoord[i];
    }
    __syncthreads(); /* lcnt is zero everywhere */
#pragma unroll
    for (int j = 0; j < GPX_SAR_VPT; j++) {
      const int32_t i = j * GPX_SAR_BLOCK + t;
      vl[j] = -1;
      vk[j] = 0;
      vm[j] = 0;
      if (i < n) {
        if ((uint32_t)vg[j] >= (uint32_t)G) { /* PaxosManager.java:1162-1194 */
          bad++;
          if (status) status[i] = GPX_S_NOGROUP;
        } else {
          const uint32_t lb = (uint32_t)vg[j] / width;
          vl[j] = (int32_t)lb;
          vk[j] = (int32_t)((((uint32_t)vg[j] - lb * width) << GPX_SAR_IDX_BITS) | (uint32_t)i);
          const bool esc = vbn[j] != b0n || vbc[j] != b0c || (uint32_t)va[j] > 0xffffu;
          vm[j] = esc ? V16_ESC : ((uint32_t)va[j] << 16);
          atomicAdd(&lcnt[lb], 1);
          if (status) status[i] = GPX_S_OK; /* no prefill pass ran; apply_ar_group overwrites it for a vote it drops */
        }
      }
    }
    if (__any(bad != 0)) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) bad += __shfl_xor(bad, d, 64);
      if ((t & 63) == 0) atomicAdd(&X.counters[2], (unsigned long long)bad);
    }
  }
  __syncthreads();
  SAR_STAMP(0, 3); /* votes loaded */
  const int32_t c = lcnt[t];
  int32_t nb;
  const int32_t start = block_exscan_rt(c, &nb);
  lcur[t] = start;
  const int32_t any_long = __syncthreads_or(c > V16_NIB_MAX);
#pragma unroll
  for (int j = 0; j < GPX_SAR_VPT; j++) {
    if (vl[j] >= 0) {
      const int32_t p = atomicAdd(&lcur[vl[j]], 1);
      keyA[p] = vk[j];
      slotA[p] = vs[j];
      cpA[p] = vc[j];
      metaA[p] = vm[j];
    }
  }
  __syncthreads();
  SAR_STAMP(0, 4); /* placed */
  /* D: lanes with more than 16 votes: sorted keys in global scratch (bucket16_body's regimes) */
  if (any_long) {
    if (c > V16_NIB_MAX && c <= V16_LANE_SORT) {
      for (int32_t a = 0; a < c; a++) {
        const uint32_t ka = (uint32_t)keyA[start + a];
        int32_t r = 0;
        for (int32_t u = 0; u < c; u++) r += (uint32_t)keyA[start + u] < ka;
        keysG[start + r] = ((unsigned long long)ka << 32) | (uint32_t)(start + a);
      }
    } else if (c > V16_LANE_SORT) {
      for (int32_t a = 0; a < c; a++)
        keysG[start + a] = ((unsigned long long)(uint32_t)keyA[start + a] << 32) | (uint32_t)(start + a);
    }
    if (__syncthreads_or(c > V16_LANE_SORT)) {
      for (int32_t q = 0; q < GPX_SAR_BLOCK; q++) {
        const int32_t cq = lcnt[q]; /* uniform */
        if (cq > V16_LANE_SORT) sort_long_segment(keysG + (lcur[q] - cq), (uint32_t)cq);
      }
      __syncthreads();
    }
  }
  /* ---- E1: the lanes publish their votes in (group, arrival) order: sorted position -> position, group ---- */
  const uint32_t rel_lane = (uint32_t)t * width;
  if (c > 0) {
    if (c <= V16_NIB_MAX) {
      const unsigned long long order = arrival_order(keyA, start, c);
      for (int32_t r = 0; r < c; r++) {
        const int32_t p = start + (int32_t)((order >> (4 * r)) & 15ull);
        permA[start + r] = p;
        gA[start + r] = rel_lane + ((uint32_t)keyA[p] >> GPX_SAR_IDX_BITS);
        outA[start + r] = 0;
      }
    } else {
      for (int32_t r = 0; r < c; r++) {
        gA[start + r] = GPX_SAR_LONG;
        outA[start + r] = 0;
      }
    }
  }
  __syncthreads();
  /* ---- E2 (job 0: the group whose first vote has sorted position t) and E3 (then: the groups of this thread's own
   * lane, if it kept them) through ONE replay site ---- */
  {
    SmallArIter it;
    it.keyA = keyA;
    it.slotA = slotA;
    it.cpA = cpA;
    it.metaA = metaA;
    it.xA = xA;
    it.yA = yA;
    it.in = in;
    it.b0n = bnum[0];
    it.b0c = bcoord[0];
    it.nout = 0;
    it.cur = 0;
    const bool longlane = c > V16_NIB_MAX;
    int32_t job = 0, r = 0;
#pragma unroll 1
    for (;;) {
      int32_t q = 0, first;
      if (job < GPX_SAR_VPT) {
        q = job * GPX_SAR_BLOCK + t;
        job++;
        if (q >= nb) continue;
        const uint32_t gq = gA[q];
        if (gq == GPX_SAR_LONG || (q > 0 && gA[q - 1] == gq)) continue;
        int32_t q2 = q + 1;
        while (q2 < nb && gA[q2] == gq) q2++; /* (at most 16: one lane's votes) */
        it.perm = true;
        it.ranks = permA + q;
        it.keys = nullptr;
        it.done = 0;
        it.c = q2 - q;
        it.relg = (int32_t)gq;
        it.omask = 0;
        first = permA[q];
      } else if (longlane && r < c) {
        it.perm = false;
        it.ranks = nullptr;
        it.keys = keysG + start;
        const uint32_t gk = (uint32_t)keyA[(uint32_t)it.keys[r]] >> GPX_SAR_IDX_BITS;
        int32_t r2 = r + 1;
        while (r2 < c && ((uint32_t)keyA[(uint32_t)it.keys[r2]] >> GPX_SAR_IDX_BITS) == gk) r2++;
        it.done = r;
        it.c = r2;
        it.relg = (int32_t)(rel_lane + gk);
        first = (int32_t)(uint32_t)it.keys[r];
        r = r2;
      } else {
        break;
      }
      const int32_t g = it.relg;
      CoordPre<KMAX> P;
      coord_preload<KMAX>(S, g, P);
      coord_preload_ring_at<KMAX>(S, g, slotA[first], P);
      apply_ar_group<KMAX>(S, X, g, it, status, P);
      if (it.perm) outA[q] = it.omask; /* bit r: the group's r-th vote produced an output */
    }
    if (longlane) outA[start] = 0x80000000u | (uint32_t)it.nout; /* its outputs: positions keysG[start .. start + nout) */
  }
  SAR_STAMP(0, 5); /* thread 0 replayed */
  __syncthreads();
  /* ---- F: the outputs in sorted-position order = gidx ascending, a group's in arrival order, into the caller's
   * columns: thread t takes sorted position t ---- */
  uint32_t ow[GPX_SAR_VPT];
  int32_t nout = 0;
#pragma unroll
  for (int j = 0; j < GPX_SAR_VPT; j++) {
    const int32_t q = GPX_SAR_VPT * t + j;
    ow[j] = q < nb ? outA[q] : 0u;
    nout += (ow[j] & 0x80000000u) ? (int32_t)(ow[j] & 0xffffu) : __popc(ow[j]);
  }
  int32_t tout;
  int32_t o = block_exscan_rt(nout, &tout);
  SAR_STAMP(0, 6); /* every thread replayed */
  auto put = [&](uint32_t p) {
    d_gidx[o] = keyA[p]; /* (emit left the group there) */
    d_slot[o] = slotA[p];
    d_bnum[o] = xA[p];
    d_bcoord[o] = yA[p];
    d_median[o] = cpA[p];
    d_kind[o] = (uint8_t)metaA[p];
    o++;
  };
#pragma unroll
  for (int j = 0; j < GPX_SAR_VPT; j++) {
    const int32_t q = GPX_SAR_VPT * t + j;
    if (ow[j] & 0x80000000u) {
      const int32_t no = (int32_t)(ow[j] & 0xffffu);
      for (int32_t i = 0; i < no; i++) put((uint32_t)keysG[q + i]);
    } else {
      uint32_t om = ow[j];
      while (om) {
        const int d = __ffs((int)om) - 1;
        om &= om - 1;
        put((uint32_t)permA[q + d]);
      }
    }
  }
  if (t == 0) {
    if (n_out) *n_out = tout;
    atomicAdd(&X.counters[0], (unsigned long long)n);
    atomicAdd(&X.counters[1], (unsigned long long)tout);
  }
  SAR_STAMP(0, 8); /* outputs written */
}
