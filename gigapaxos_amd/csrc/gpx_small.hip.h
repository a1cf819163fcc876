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
 *   range     workgroup w (a drawn ticket, as in k_ac_small: a workgroup only ever waits for workgroups that
 *             have started) owns the groups [w * RG, (w + 1) * RG), RG = ceil(G / W).
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
#define GPX_SAR_LDS_BYTES ((2 * GPX_SAR_BLOCK + 6 * GPX_SAR_CAP) * 4)

/* One lane's votes in (group, arrival) order, with GroupIter's interface (next / emit) for apply_ar_group.  Ranks
 * [done, c) are the CURRENT group's votes; positions come from the nibble word (at most 16 votes in the lane) or
 * from the lane's sorted keys (key << 32 | position) in global scratch. */
struct SmallArIter {
  const int32_t* keyA;
  int32_t *slotA, *cpA;
  uint32_t* metaA;
  int32_t *xA, *yA;
  unsigned long long* keys;
  VoteCols in;
  int32_t b0n, b0c;
  int32_t start, c, done, nout;
  bool nib;
  unsigned long long order;
  uint32_t omask;
  uint32_t cur;
  __device__ __forceinline__ uint32_t pos(int32_t r) const {
    return nib ? (uint32_t)start + (uint32_t)((order >> (4 * r)) & 15ull) : (uint32_t)keys[r];
  }
  __device__ __forceinline__ bool next(Rec& out) {
    if (done >= c) return false;
    const uint32_t p = pos(done);
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
  /* output of the CURRENT vote, parked in the vote's own words (the key stays: it names the group) */
  __device__ __forceinline__ void emit(int32_t slot, int32_t x, int32_t y, int32_t z, int32_t kind) {
    slotA[cur] = slot;
    cpA[cur] = z;
    metaA[cur] = (uint32_t)kind;
    xA[cur] = x;
    yA[cur] = y;
    if (nib)
      omask |= 1u << (done - 1);
    else
      keys[nout] = cur; /* entry nout <= done - 1: consumed */
    nout++;
  }
};

template <int KMAX>
__global__ __launch_bounds__(GPX_SAR_BLOCK) void k_ar_small(
    DevState S, DevScratch X, int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot, const int32_t* __restrict__ acceptor,
    const int32_t* __restrict__ max_cp, int32_t* __restrict__ d_gidx, int32_t* __restrict__ d_slot,
    int32_t* __restrict__ d_bnum, int32_t* __restrict__ d_bcoord, int32_t* __restrict__ d_median,
    uint8_t* __restrict__ d_kind, int32_t* __restrict__ n_out, uint8_t* __restrict__ status,
    unsigned long long* __restrict__ tickets, uint32_t epoch, uint32_t* __restrict__ draw, uint32_t draw_base,
    int32_t W, int32_t gate) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  __shared__ int32_t s_w, s_nb, s_lower, s_before;
  const int32_t t = (int32_t)threadIdx.x;
  /* the range is DRAWN, not read off blockIdx (k_ac_small): always, so that the host's count of draws stays true */
  if (t == 0) s_w = (int32_t)(atomicAdd(draw, 1u) - draw_base);
  __syncthreads();
  const int32_t w = s_w;
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
  int32_t* idxS = xA; /* the collected arrival indices: read before the placement, xA is written by the replay */
  const int32_t G = S.G;
  const int32_t RG = (int32_t)(((int64_t)G + W - 1) / W);
  const int32_t lo = (int32_t)min((int64_t)G, (int64_t)w * RG);
  const int32_t hi = (int32_t)min((int64_t)G, (int64_t)lo + RG);
  const int32_t b0n = bnum[0], b0c = bcoord[0];
  const VoteCols in{bnum, bcoord, acceptor};
  const bool vec = !((uintptr_t)gidx & 15);

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
  bool published = false, first_scan = true;
  int32_t cur = lo, ghi = hi, ilo = 0, ihi = n;
  int32_t span_hint = hi - lo; /* width of the next range to try (narrowed by a pass that overflowed) */
  while (cur < hi) {
    /* ---- collect: the votes of groups [cur, ghi) with arrival index in [ilo, ihi) ---- */
    if (t == 0) {
      s_nb = 0;
      s_lower = 0;
    }
    lcnt[t] = 0;
    __syncthreads();
    {
      const uint32_t gspan = (uint32_t)(ghi - cur);
      const bool do_bad = w == 0 && first_scan; /* votes outside the table: marked and counted once per call */
      int32_t lower = 0, bad = 0;
      for (int32_t r0 = 0; r0 < n; r0 += 8 * 4 * GPX_SAR_BLOCK) {
        int32_t gg[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int32_t i0 = r0 + (k * GPX_SAR_BLOCK + t) * 4;
          if (vec && i0 + 3 < n) {
            const I4 v = *(const I4*)(gidx + i0);
            gg[k][0] = v.x, gg[k][1] = v.y, gg[k][2] = v.z, gg[k][3] = v.w;
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) gg[k][q] = i0 + q < n ? gidx[i0 + q] : 0;
          }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int32_t i0 = r0 + (k * GPX_SAR_BLOCK + t) * 4;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int32_t i = i0 + q;
            if (i >= n) continue;
            const int32_t g = gg[k][q];
            if ((uint32_t)(g - cur) < gspan) {
              if (i >= ilo && i < ihi) {
                const int32_t p = atomicAdd(&s_nb, 1);
                if (p < GPX_SAR_CAP) idxS[p] = i;
              } else if (i < ilo) {
                lower++;
              }
            } else if ((uint32_t)g < (uint32_t)cur) {
              lower++;
            } else if (do_bad && (uint32_t)g >= (uint32_t)G) {
              bad++;
              if (status) status[i] = GPX_S_NOGROUP; /* PaxosManager.java:1162-1194 */
            }
          }
        }
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        lower += __shfl_xor(lower, d, 64);
        bad += __shfl_xor(bad, d, 64);
      }
      if ((t & 63) == 0) {
        if (lower) atomicAdd(&s_lower, lower);
        if (bad) atomicAdd(&X.counters[2], (unsigned long long)bad);
      }
    }
    first_scan = false;
    __syncthreads();
    const int32_t nb = s_nb, boff = s_lower;
    __syncthreads(); /* (thread 0 resets the two words at the top of the next pass) */
    if (nb > GPX_SAR_CAP) {
      /* more than one pass stages: narrow the range (aiming at half the capacity), or - one group alone - take
       * the next GPX_SAR_CAP arrival indices */
      if (ghi - cur > 1) {
        ghi = cur + (int32_t)max((int64_t)1, (int64_t)(ghi - cur) * GPX_SAR_CAP / nb / 2);
        span_hint = ghi - cur;
      } else { /* (a window of GPX_SAR_CAP indices always fits: the estimate only ever shrinks towards it) */
        ihi = ilo + (int32_t)max((int64_t)GPX_SAR_CAP, (int64_t)(ihi - ilo) * GPX_SAR_CAP / nb / 2);
      }
      continue;
    }
    const bool last = ghi == hi && ihi == n;
    const uint32_t width = ((uint32_t)(ghi - cur) + GPX_SAR_BLOCK - 1) / GPX_SAR_BLOCK; /* groups per lane */
    unsigned long long* keysG = X.perm + boff; /* [boff, boff + nb): nobody else's (header) */

    /* ---- regroup: A count per lane, B scan, C placement lane-major ---- */
    int32_t vl[GPX_SAR_VPT], vk[GPX_SAR_VPT], vs[GPX_SAR_VPT], vc[GPX_SAR_VPT];
    uint32_t vm[GPX_SAR_VPT];
#pragma unroll
    for (int j = 0; j < GPX_SAR_VPT; j++) {
      const int32_t p = j * GPX_SAR_BLOCK + t;
      vl[j] = -1;
      vk[j] = vs[j] = vc[j] = 0;
      vm[j] = 0;
      if (p < nb) {
        const int32_t i = idxS[p];
        const int32_t g = gidx[i], sl = slot[i], cp = max_cp[i], ac = acceptor[i], bn = bnum[i], bc = bcoord[i];
        const uint32_t rel = (uint32_t)(g - cur);
        const uint32_t lb = rel / width;
        vl[j] = (int32_t)lb;
        vk[j] = (int32_t)(((rel - lb * width) << GPX_SAR_IDX_BITS) | (uint32_t)i);
        vs[j] = sl;
        vc[j] = cp;
        const bool esc = bn != b0n || bc != b0c || (uint32_t)ac > 0xffffu;
        vm[j] = esc ? V16_ESC : ((uint32_t)ac << 16);
        atomicAdd(&lcnt[lb], 1);
        if (status) status[i] = GPX_S_OK; /* no prefill pass ran; apply_ar_group overwrites it for a vote it drops */
      }
    }
    __syncthreads();
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
    /* ---- E: replay, one lane per `width` groups, group by group ---- */
    SmallArIter it;
    it.keyA = keyA;
    it.slotA = slotA;
    it.cpA = cpA;
    it.metaA = metaA;
    it.xA = xA;
    it.yA = yA;
    it.keys = keysG + start;
    it.in = in;
    it.b0n = b0n;
    it.b0c = b0c;
    it.start = start;
    it.c = 0;
    it.done = 0;
    it.nout = 0;
    it.nib = c <= V16_NIB_MAX;
    it.order = 0;
    it.omask = 0;
    it.cur = 0;
    const int32_t g_lane = cur + t * (int32_t)width; /* (only lanes with votes use it: those lie inside the range) */
    if (c > 0) {
      if (it.nib) it.order = arrival_order(keyA, start, c);
      int32_t r = 0;
      while (r < c) {
        const uint32_t gk = (uint32_t)keyA[it.pos(r)] >> GPX_SAR_IDX_BITS;
        int32_t r2 = r + 1;
        while (r2 < c && ((uint32_t)keyA[it.pos(r2)] >> GPX_SAR_IDX_BITS) == gk) r2++;
        const int32_t g = g_lane + (int32_t)gk;
        CoordPre<KMAX> P;
        coord_preload<KMAX>(S, g, P);
        coord_preload_ring<KMAX>(S, g, P);
        it.done = r;
        it.c = r2;
        apply_ar_group<KMAX>(S, X, g, it, status, P);
        r = r2;
      }
    }
    /* ---- F: this pass's outputs, lane-major = gidx ascending, into the caller's columns ---- */
    const int32_t nout = it.nout;
    int32_t tout;
    const int32_t ex = block_exscan_rt(nout, &tout);
    if (last) {
      if (t == 0)
        __hip_atomic_store(&tickets[w], ((unsigned long long)epoch << 32) | (uint32_t)(running + tout), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_AGENT);
      published = true;
    }
    if (tout > 0) {
      if (base < 0) base = wait_earlier();
      uint32_t om = it.omask;
      for (int32_t q = 0; q < nout; q++) {
        uint32_t p;
        if (it.nib) {
          const int d = __ffs((int)om) - 1; /* rank of the next vote with an output */
          om &= om - 1;
          p = (uint32_t)start + (uint32_t)((it.order >> (4 * d)) & 15ull);
        } else {
          p = (uint32_t)it.keys[q];
        }
        const int64_t o = (int64_t)base + running + ex + q;
        d_gidx[o] = g_lane + (int32_t)((uint32_t)keyA[p] >> GPX_SAR_IDX_BITS);
        d_slot[o] = slotA[p];
        d_bnum[o] = xA[p];
        d_bcoord[o] = yA[p];
        d_median[o] = cpA[p];
        d_kind[o] = (uint8_t)metaA[p];
      }
    }
    running += tout;
    __syncthreads(); /* the next pass stages over these words */
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
