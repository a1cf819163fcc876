/*
 * gpx_small.hip.h — a SMALL accept-reply call in ONE launch, whatever the order of its votes (round 4).
 *
 * The reference's frames carry at most 2,048 slots (BatchedAcceptReply.java:27) and a coordinator drains what
 * a handful of acceptors sent since the last call: a few thousand to a few ten thousand votes, in no particular
 * order (PaxosPacketBatcher.java:182-209).  Through the partition pipeline (k_hist -> k_scatter_ar16 ->
 * k_bucket_ar16 -> k_emit_dec16) such a call is four DEPENDENT launches that each move next to nothing:
 * 42 us per 65,536 votes (profiles/r03_batch_sweep.json) - the launch floor, not the work.  Here the whole call
 * is one kernel of W <= 128 workgroups:
 *
 *   range     workgroup w owns the groups [w * RG, (w + 1) * RG), RG = ceil(G / W).
 *   collect   every workgroup reads the WHOLE gidx column (n <= 131,072 ints: it is L2-resident after the first
 *             reader; eight 16-byte loads in flight per lane) and keeps the arrival indices of the votes of its
 *             range in LDS; it also counts the votes of lower ranges - its slice of the key scratch.
 *   regroup   one LANE per `width` = ceil(RG / 1024) consecutive groups: count per lane (LDS atomics), scan,
 *             placement lane-major in LDS (structure of arrays, 16 bytes per vote as in gpx_ar16.hip.h; the
 *             first word is KEY = group offset inside the lane << 17 | arrival index, so one unsigned compare
 *             orders a lane's votes by (group, arrival)).  Up to 16 votes per lane: order = a nibble word in a
 *             register; up to 96: ranked by the lane itself; more (a hot group): the workgroup's bitonic sort -
 *             exactly the three regimes of bucket16_body.
 *   replay    the lane walks its votes group by group through apply_ar_group (gpx_kernels.hip.h), unchanged:
 *             PISM.handleAcceptReply -> PaxosCoordinatorState.handleAcceptReplyMyBallot / HigherBallot
 *             (PaxosInstanceStateMachine.java:1248-1419, PaxosCoordinatorState.java:597-683).  No status prefill
 *             pass: the workgroup that collects a vote marks it.
 *   outputs   parked in the vote's own LDS words; per-workgroup count -> an epoch-tagged ticket; every workgroup
 *             sums the tickets before its own (no chain: a ticket depends on nothing but its own workgroup) and
 *             writes its decisions straight into the caller's columns: grouped by gidx ascending, a group's
 *             entries in arrival order (include/gpx.h ORDER).  The last ticket writes *n_out.
 *
 * A range that holds more votes than the LDS stages (a skewed batch) is taken in PASSES: the workgroup narrows
 * the range and scans again (ascending sub-ranges, each tried twice as wide as the last that fitted: the output
 * order is kept); a single group with more votes than that is taken in windows of arrival indices (its state
 * lives in global memory between passes: the replay is sequential either way).  Slow, and only has to be correct -
 * the uniform batch is one pass per workgroup.
 *
 * Results are identical to the partition pipeline's (tests/test_small_ar_gpu.py: both paths against the oracle).
 */
#pragma once
#include "gpx_ar16.hip.h"

#define GPX_SAR_MAX_N 131072      /* votes per call on this path (arrival index: 17 bits of KEY) */
#define GPX_SAR_MAX_G (1 << 24)   /* groups in the table: a lane's group offset stays below 2^14 (KEY's high bits) */
#define GPX_SAR_IDX_BITS 17
#define GPX_SAR_IDX_MASK ((1u << GPX_SAR_IDX_BITS) - 1u)
#define GPX_SAR_BLOCK 1024
#define GPX_SAR_VPT 2                               /* votes a thread carries through the regrouping */
#define GPX_SAR_CAP (GPX_SAR_BLOCK * GPX_SAR_VPT)   /* votes staged per pass */
#define GPX_SAR_MAX_WG 128
#define GPX_SAR_ROUND (32 * GPX_SAR_BLOCK)          /* column entries one scan round covers: 32 per lane */
#define GPX_SAR_ROUNDS (GPX_SAR_MAX_N / GPX_SAR_ROUND)
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
__global__ __launch_bounds__(GPX_SAR_BLOCK) void k_ar_small(
    DevState S, DevScratch X, int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot, const int32_t* __restrict__ acceptor,
    const int32_t* __restrict__ max_cp, int32_t* __restrict__ d_gidx, int32_t* __restrict__ d_slot,
    int32_t* __restrict__ d_bnum, int32_t* __restrict__ d_bcoord, int32_t* __restrict__ d_median,
    uint8_t* __restrict__ d_kind, int32_t* __restrict__ n_out, uint8_t* __restrict__ status,
    unsigned long long* __restrict__ tickets, uint32_t epoch, int32_t W, int32_t gate) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  __shared__ int32_t s_lower, s_before;
  const int32_t t = (int32_t)threadIdx.x;
  /* The range is read off blockIdx: a workgroup waits (for the tickets of lower ranges) only after its own work,
   * and the at most 128 workgroups of a call are all resident at once (one per CU), whatever order they start in. */
  const int32_t w = (int32_t)blockIdx.x;
  SAR_STAMP(w, 0);
  /* launched behind the sorted-runs attempt (GPX_TRY_REPLY_RUNS): only a batch it gave up on is this kernel's */
  if (gate && *X.unsorted != X.epoch) return;
  int32_t* lcnt = lds;
  int32_t* lcur = lds + GPX_SAR_BLOCK;
  int32_t* keyA = lcur + GPX_SAR_BLOCK;
  int32_t* slotA = keyA + GPX_SAR_CAP;
  int32_t* cpA = slotA + GPX_SAR_CAP;
  uint32_t* metaA = (uint32_t*)(cpA + GPX_SAR_CAP);
  int32_t* xA = (int32_t*)(metaA + GPX_SAR_CAP);
  int32_t* yA = xA + GPX_SAR_CAP;
  int32_t* permA = yA + GPX_SAR_CAP;             /* sorted position -> position in the arrays above */
  uint32_t* gA = (uint32_t*)(permA + GPX_SAR_CAP); /* sorted position -> group (relative to the pass's first), or LONG */
  uint32_t* outA = gA + GPX_SAR_CAP;             /* sorted position of a group's first vote -> its outputs */
  int32_t* idxS = xA; /* the collected arrival indices: read before the placement, xA is written by the replay */
  const int32_t G = S.G;
  const int32_t RG = (int32_t)(((int64_t)G + W - 1) / W);
  const int32_t lo = (int32_t)min((int64_t)G, (int64_t)w * RG);
  const int32_t hi = (int32_t)min((int64_t)G, (int64_t)lo + RG);
  const int32_t b0n = bnum[0], b0c = bcoord[0];
  const VoteCols in{bnum, bcoord, acceptor};

  /* This workgroup's SLICE of the batch (arrival indices): votes outside the table are marked and counted there,
   * once per call (PaxosManager.java:1162-1194) - every vote lies in exactly one slice, whatever its group.  The
   * slice's first 2,048 entries are requested here and looked at behind the first scan (no round trip of their own) */
  const int32_t slice = (n + W - 1) / W;
  const int32_t sl0 = (int32_t)min((int64_t)n, (int64_t)w * slice), sl1 = (int32_t)min((int64_t)n, (int64_t)sl0 + slice);
  const int32_t sg0 = gidx[min(sl0 + t, n - 1)], sg1 = gidx[min(sl0 + GPX_SAR_BLOCK + t, n - 1)];
  bool slice_done = false;
  /* sum of the tickets before this workgroup's (each depends on its own workgroup only) */
  auto wait_earlier = [&]() -> int32_t {
    if (t == 0) s_before = 0;
    __syncthreads();
    int32_t before = 0;
    for (int32_t q = t; q < w; q += GPX_SAR_BLOCK) {
      unsigned long long v;
      do {
        v = __hip_atomic_load(&tickets[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      } while ((uint32_t)(v >> 32) != epoch);
      before += (int32_t)(uint32_t)v;
    }
    if (before) atomicAdd(&s_before, before);
    __syncthreads();
    return s_before;
  };

  int32_t running = 0; /* outputs of this workgroup's earlier passes */
  int32_t base = -1;   /* outputs of the workgroups before this one (once known) */
  bool published = false;
  int32_t cur = lo, ghi = hi, ilo = 0, ihi = n;
  int32_t span_hint = hi - lo; /* width of the next range to try (narrowed by a pass that overflowed) */
  while (cur < hi) {
    /* ---- collect: the votes of groups [cur, ghi) with arrival index in [ilo, ihi).  Straight-line per entry (a
     * compare and a mask bit: the first build branched per entry and spent 19 of its 66 us here,
     * profiles/r04_sar_trace_1.txt): thread t reads entries k * 1024 + t of every round of 32,768, bit k of the
     * round's mask word says "mine" ---- */
    if (t == 0) s_lower = 0;
    lcnt[t] = 0;
    __syncthreads();
    uint32_t* mkS = (uint32_t*)keyA; /* [rounds][1024] mask words (keyA and slotA are free until the placement) */
    int32_t lower = 0, mine = 0;
    {
      const uint32_t gspan = (uint32_t)(ghi - cur);
      const uint32_t ispan = (uint32_t)(ihi - ilo);
#pragma unroll 1
      for (int r = 0; r < GPX_SAR_ROUNDS; r++) {
        const int32_t r0 = r * GPX_SAR_ROUND;
        if (r0 >= n) break; /* uniform */
        uint32_t mk = 0;
        {
          /* 32 coalesced dword loads in flight, none of them behind a branch (an index behind the batch's end is
           * clamped for the load and the entry then counts as INT32_MIN: in no range, below no group) */
          int32_t gg[32];
#pragma unroll
          for (int k = 0; k < 32; k++) gg[k] = gidx[min(r0 + k * GPX_SAR_BLOCK + t, n - 1)];
#pragma unroll
          for (int k = 0; k < 32; k++) {
            const int32_t i = r0 + k * GPX_SAR_BLOCK + t;
            const int32_t g = i < n ? gg[k] : INT32_MIN;
            const bool inr = (uint32_t)(g - cur) < gspan;
            const bool inw = (uint32_t)(i - ilo) < ispan;
            mk |= (inr && inw) ? (1u << k) : 0u;
            /* the votes that come before this pass's in (group, arrival) order: its slice of the key scratch */
            lower += (((uint32_t)g < (uint32_t)cur) || (inr && i < ilo)) ? 1 : 0;
          }
        }
        mkS[r * GPX_SAR_BLOCK + t] = mk; /* (read back by this thread only) */
        mine += __popc(mk);
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lower += __shfl_xor(lower, d, 64);
    if ((t & 63) == 0 && lower) atomicAdd(&s_lower, lower);
    int32_t nb;
    int32_t ex0 = block_exscan_rt(mine, &nb); /* (its barriers also publish s_lower) */
    const int32_t boff = s_lower;
    SAR_STAMP(w, 2); /* column scanned */
    if (!slice_done) {
      slice_done = true;
      int32_t bad = 0;
      if (sl0 + t < sl1 && (uint32_t)sg0 >= (uint32_t)G) {
        bad++;
        if (status) status[sl0 + t] = GPX_S_NOGROUP;
      }
      if (sl0 + GPX_SAR_BLOCK + t < sl1 && (uint32_t)sg1 >= (uint32_t)G) {
        bad++;
        if (status) status[sl0 + GPX_SAR_BLOCK + t] = GPX_S_NOGROUP;
      }
      for (int32_t i = sl0 + 2 * GPX_SAR_BLOCK + t; i < sl1; i += GPX_SAR_BLOCK) { /* (a slice of more than 2,048 votes: a call capped at few workgroups) */
        if ((uint32_t)gidx[i] >= (uint32_t)G) {
          bad++;
          if (status) status[i] = GPX_S_NOGROUP;
        }
      }
      if (__any(bad != 0)) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) bad += __shfl_xor(bad, d, 64);
        if ((t & 63) == 0) atomicAdd(&X.counters[2], (unsigned long long)bad);
      }
    }
    if (nb > GPX_SAR_CAP) {
      /* more than one pass stages: narrow the range (aiming at half the capacity), or - one group alone - take
       * the next window of arrival indices */
      if (ghi - cur > 1) {
        ghi = cur + (int32_t)max((int64_t)1, (int64_t)(ghi - cur) * GPX_SAR_CAP / nb / 2);
        span_hint = ghi - cur;
      } else { /* (a window of GPX_SAR_CAP indices always fits: the estimate only ever shrinks towards it) */
        ihi = ilo + (int32_t)max((int64_t)GPX_SAR_CAP, (int64_t)(ihi - ilo) * GPX_SAR_CAP / nb / 2);
      }
      __syncthreads(); /* (thread 0 resets s_lower at the top of the next pass) */
      continue;
    }
    uint32_t mkr[GPX_SAR_ROUNDS]; /* (all four read before the first index is stored: idxS lies elsewhere, but keep it so) */
#pragma unroll
    for (int r = 0; r < GPX_SAR_ROUNDS; r++) mkr[r] = r * GPX_SAR_ROUND < n ? mkS[r * GPX_SAR_BLOCK + t] : 0u;
#pragma unroll
    for (int r = 0; r < GPX_SAR_ROUNDS; r++) {
      uint32_t m = mkr[r];
      while (m) {
        const int e = __ffs((int)m) - 1;
        m &= m - 1;
        idxS[ex0++] = r * GPX_SAR_ROUND + e * GPX_SAR_BLOCK + t;
      }
    }
    __syncthreads();
    const bool last = ghi == hi && ihi == n;
    const uint32_t width = ((uint32_t)(ghi - cur) + GPX_SAR_BLOCK - 1) / GPX_SAR_BLOCK; /* groups per lane */
    unsigned long long* keysG = X.perm + boff; /* [boff, boff + nb): nobody else's (header) */

    /* ---- regroup: A count per lane, B scan, C placement lane-major ---- */
    int32_t vl[GPX_SAR_VPT], vk[GPX_SAR_VPT], vs[GPX_SAR_VPT], vc[GPX_SAR_VPT];
    uint32_t vm[GPX_SAR_VPT];
    {
      /* every load of both votes in flight before the first is looked at: the indices are clamped (a lane without a
       * vote loads somebody's and drops it), so nothing here stands behind a branch */
      int32_t vi[GPX_SAR_VPT], vg[GPX_SAR_VPT], va[GPX_SAR_VPT], vbn[GPX_SAR_VPT], vbc[GPX_SAR_VPT];
#pragma unroll
      for (int j = 0; j < GPX_SAR_VPT; j++) vi[j] = idxS[min(j * GPX_SAR_BLOCK + t, max(nb - 1, 0))];
      if (nb == 0) vi[0] = vi[1] = 0; /* (uniform; idxS holds nothing) */
#pragma unroll
      for (int j = 0; j < GPX_SAR_VPT; j++) {
        const int32_t i = vi[j];
        vg[j] = gidx[i], vs[j] = slot[i], vc[j] = max_cp[i], va[j] = acceptor[i], vbn[j] = bnum[i], vbc[j] = bcoord[i];
      }
#pragma unroll
      for (int j = 0; j < GPX_SAR_VPT; j++) {
        vl[j] = -1;
        vk[j] = 0;
        vm[j] = 0;
        if (j * GPX_SAR_BLOCK + t < nb) {
          const int32_t i = vi[j];
          const uint32_t rel = (uint32_t)(vg[j] - cur);
          const uint32_t lb = rel / width;
          vl[j] = (int32_t)lb;
          vk[j] = (int32_t)(((rel - lb * width) << GPX_SAR_IDX_BITS) | (uint32_t)i);
          const bool esc = vbn[j] != b0n || vbc[j] != b0c || (uint32_t)va[j] > 0xffffu;
          vm[j] = esc ? V16_ESC : ((uint32_t)va[j] << 16);
          atomicAdd(&lcnt[lb], 1);
          if (status) status[i] = GPX_S_OK; /* no prefill pass ran; apply_ar_group overwrites it for a vote it drops */
        }
      }
    }
    __syncthreads();
    SAR_STAMP(w, 3); /* votes gathered */
    const int32_t c = lcnt[t];
    int32_t tot_;
    const int32_t start = block_exscan_rt(c, &tot_);
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
    SAR_STAMP(w, 4); /* placed */
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
    /* ---- E: replay.  A lane's range holds `width` groups, and in a thin batch over a large table most of a
     * lane's votes belong to DIFFERENT groups - replayed by the lane one after the other that was 30 of the first
     * build's 66 us (every group is two or three dependent round trips).  So the lanes only publish their votes in
     * (group, arrival) order - E1: sorted position -> position, group - and every group gets a thread of its own -
     * E2: thread t takes the groups whose first vote has sorted position t or t + 1024.  A lane with more than 16
     * votes (a hot group) keeps them and replays them itself (E3). ---- */
    const uint32_t rel_lane = (uint32_t)t * width;
    unsigned long long order = 0;
    if (c > 0) {
      if (c <= V16_NIB_MAX) {
        order = arrival_order(keyA, start, c);
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
    /* E2 (jobs 0 .. VPT - 1: the group whose first vote has sorted position job * 1024 + t) and E3 (then: the
     * groups of this thread's own lane, if it kept them) through ONE replay site */
    {
      SmallArIter it;
      it.keyA = keyA;
      it.slotA = slotA;
      it.cpA = cpA;
      it.metaA = metaA;
      it.xA = xA;
      it.yA = yA;
      it.in = in;
      it.b0n = b0n;
      it.b0c = b0c;
      it.nout = 0;
      it.cur = 0;
      const bool longlane = c > V16_NIB_MAX;
      int32_t job = 0, r = 0;
#pragma unroll 1
      for (;;) {
        int32_t g, q = 0, first;
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
        g = cur + it.relg;
        CoordPre<KMAX> P;
        coord_preload<KMAX>(S, g, P);
        coord_preload_ring_at<KMAX>(S, g, slotA[first], P);
        apply_ar_group<KMAX>(S, X, g, it, status, P);
        if (it.perm) outA[q] = it.omask; /* bit r: the group's r-th vote produced an output */
      }
      if (longlane) outA[start] = 0x80000000u | (uint32_t)it.nout; /* its outputs: positions keysG[start .. start + nout) */
    }
    SAR_STAMP(w, 5); /* thread 0 replayed */
    __syncthreads();
    /* ---- F: this pass's outputs in sorted-position order = gidx ascending, a group's in arrival order, into the
     * caller's columns: thread t takes sorted positions 2 t and 2 t + 1 ---- */
    uint32_t ow[GPX_SAR_VPT];
    int32_t nout = 0;
#pragma unroll
    for (int j = 0; j < GPX_SAR_VPT; j++) {
      const int32_t q = GPX_SAR_VPT * t + j;
      ow[j] = q < nb ? outA[q] : 0u;
      nout += (ow[j] & 0x80000000u) ? (int32_t)(ow[j] & 0xffffu) : __popc(ow[j]);
    }
    int32_t tout;
    const int32_t ex = block_exscan_rt(nout, &tout);
    SAR_STAMP(w, 6); /* every thread replayed */
    if (last) {
      if (t == 0)
        __hip_atomic_store(&tickets[w], ((unsigned long long)epoch << 32) | (uint32_t)(running + tout), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_AGENT);
      published = true;
    }
    if (tout > 0) {
      if (base < 0) base = wait_earlier();
      SAR_STAMP(w, 7); /* earlier tickets in */
      int64_t o = (int64_t)base + running + ex;
      auto put = [&](uint32_t p) {
        d_gidx[o] = cur + keyA[p]; /* (emit left the group there) */
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
          for (int32_t i = 0; i < no; i++) put((uint32_t)(X.perm + boff)[q + i]);
        } else {
          uint32_t om = ow[j];
          while (om) {
            const int d = __ffs((int)om) - 1;
            om &= om - 1;
            put((uint32_t)permA[q + d]);
          }
        }
      }
    }
    running += tout;
    __syncthreads(); /* the next pass stages over these words */
    SAR_STAMP(w, 8); /* outputs written */
    if (ihi < n) { /* the same group's next window of arrival indices */
      ilo = ihi;
      ihi = n;
    } else { /* the next range: twice as wide as the last one that fitted */
      cur = ghi;
      span_hint = (int32_t)min((int64_t)(hi - lo), 2 * (int64_t)span_hint);
      ghi = (int32_t)min((int64_t)hi, (int64_t)cur + span_hint);
      ilo = 0;
    }
  }
  if (!published && t == 0) /* an empty range (more workgroups than groups) */
    __hip_atomic_store(&tickets[w], ((unsigned long long)epoch << 32) | (uint32_t)running, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_AGENT);
  if (w == 0 && t == 0) atomicAdd(&X.counters[0], (unsigned long long)n);
  if (w == W - 1) { /* the call's count */
    if (base < 0) base = wait_earlier();
    if (t == 0) {
      if (n_out) *n_out = base + running;
      atomicAdd(&X.counters[1], (unsigned long long)(base + running));
    }
  }
}
