/*
 * gpx_runs.hip.h — accept replies that arrive as a few SORTED RUNS (round 3): no partition.
 *
 * Inside the real pipeline the coordinator's vote batch is the concatenation of what each acceptor
 * sent, and every acceptor's replies leave gpx_accept_batch in the order of the ACCEPT batch, i.e.
 * grouped by group, groups ascending (include/gpx.h ORDER): the batch is K ascending runs, not a
 * shuffle.  The reference sees the same shape - PISM.handleBatchedAcceptReply walks one acceptor's
 * TreeMap per (group, ballot) (PaxosInstanceStateMachine.java:1370-1419).  Such a batch needs no
 * histogram, no scatter, no LDS regrouping:
 *
 *   k_runs_check    reads the gidx column once: in range, and at most GPX_RUNS_MAX - 1 descents -> the run
 *                   starts; anything else raises the call's epoch in *X.unsorted (the partition pipeline
 *                   takes the batch - or, under the GPX_ORDERED_REPLY_RUNS promise, it is refused whole).
 *   k_ar_runs       one lane per record; the lane of the FIRST record of a group (first run that holds
 *                   the group, first record there) owns the group: it finds the group's votes in the
 *                   later runs (same offset as in its own run if the runs are alike - the usual case -
 *                   otherwise a binary search) and replays them in ARRAY order - run 0's votes before run
 *                   1's: exactly arrival order - through apply_ar_group, i.e. PISM.handleAcceptReply ->
 *                   PaxosCoordinatorState.handleAcceptReplyMyBallot / HigherBallot (PCS:597-683) unchanged.
 *                   Consecutive lanes of run 0 own ascending groups: coalesced state accesses.  The q-th
 *                   output of a group is parked at the array index of the group's q-th vote.
 *   k_emit_dec_runs parked outputs -> dense columns in record order.  When every output was parked
 *                   inside run 0 (the usual case: every group has a vote in run 0) that IS the contract's
 *                   order, grouped by gidx ascending.
 *   k_merge_runs    otherwise (a group absent from run 0, or more outputs than votes in run 0): the
 *                   compacted outputs are up to GPX_RUNS_MAX ascending segments; every entry computes its
 *                   rank in their merge (binary searches) and moves there.  Returns at once when not needed.
 */
#pragma once
#include "gpx_ar16.hip.h"
#include "gpx_direct.hip.h"

#define GPX_RUNS_MAX 16 /* PC.MAX_GROUP_SIZE acceptors: PaxosConfig.java:532 */

/* per-call facts about the batch; two of them, used alternately: call N's k_runs_check clears the one
 * call N + 1 will use (its last user, call N - 1, has finished: stream order) */
struct RunsInfo {
  int32_t n_desc;                  /* descents found so far (atomic) */
  int32_t need_merge;              /* an output was parked outside run 0 */
  int32_t total;                   /* outputs of the call (k_emit_dec_runs) */
  int32_t pad;
  int32_t start[GPX_RUNS_MAX + 1]; /* start[0] = 0; the others in the order the atomics gave: sorted by the readers */
  int32_t seg_off[GPX_RUNS_MAX + 1]; /* compacted outputs parked before each run start (k_emit_dec_runs) */
};

struct RunsStage {
  Stage16 O;          /* parked outputs, by record index */
  uint32_t* tag;      /* [n] == epoch: record i holds a parked output */
  int32_t* chunk_cnt; /* [ceil(n / 1024)] parked outputs per chunk; zeroed by k_runs_check */
  Stage16 T;          /* compacted outputs awaiting the merge */
};

/* order check of a vote batch: at most GPX_RUNS_MAX ascending runs, every index in range */
__global__ __launch_bounds__(GPX_OC_BLOCK) void k_runs_check(int32_t n, const int32_t* __restrict__ gidx, int32_t G,
                                                          DevScratch X, uint8_t* __restrict__ status,
                                                          RunsInfo* __restrict__ info, RunsInfo* __restrict__ next_info,
                                                          int32_t* __restrict__ zero, int32_t nzero) {
  const int64_t i0 = ((int64_t)blockIdx.x * GPX_OC_BLOCK + threadIdx.x) * 8;
  if (zero && i0 / 8 < nzero) zero[i0 / 8] = 0; /* nzero <= ceil(n / 8): the grid covers it */
  if (blockIdx.x == 0 && threadIdx.x < (int)(sizeof(RunsInfo) / 4)) ((int32_t*)next_info)[threadIdx.x] = 0;
  bool bad = false;
  uint32_t desc = 0; /* bit q: gidx[i0 + q] > gidx[i0 + q + 1] */
  if (i0 < n) {
    int32_t g[9];
    const bool full = i0 + 8 < n;
    if (full && !((uintptr_t)gidx & 15)) {
      const I4 a = *(const I4*)(gidx + i0), b = *(const I4*)(gidx + i0 + 4);
      g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w;
      g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
      g[8] = gidx[i0 + 8];
    } else {
#pragma unroll
      for (int q = 0; q < 9; q++) g[q] = (i0 + q < n) ? gidx[i0 + q] : INT32_MAX;
    }
    unsigned long long stw = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (i0 + q < n) {
        const bool oob = (uint32_t)g[q] >= (uint32_t)G;
        bad |= oob;
        if (i0 + q + 1 < n && g[q] > g[q + 1]) desc |= 1u << q;
        if (oob) stw |= (unsigned long long)GPX_S_NOGROUP << (8 * q);
      }
    }
    if (!status) {
    } else if (full && !((uintptr_t)status & 7)) {
      *(unsigned long long*)(status + i0) = stw; /* GPX_S_OK == 0 */
    } else {
      for (int q = 0; q < 8; q++)
        if (i0 + q < n) status[i0 + q] = (uint8_t)(stw >> (8 * q));
    }
  }
  /* a shuffled batch has ~1000 descents per workgroup: judged here, without touching the shared counter */
  const int32_t nd = __syncthreads_count(desc != 0);
  bad = __syncthreads_or(bad) || nd > GPX_RUNS_MAX - 1;
  if (!bad && desc) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if ((desc >> q) & 1u) {
        const int32_t k = atomicAdd(&info->n_desc, 1);
        if (k < GPX_RUNS_MAX - 1)
          info->start[k + 1] = (int32_t)(i0 + q + 1);
        else
          bad = true;
      }
    }
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicMax(X.unsorted, X.epoch);
}

/* the run starts of the call, ascending, in LDS: rs[0 .. R], rs[R] = n; returns R (every thread).  The
 * workgroup must call it uniformly. */
__device__ __forceinline__ int32_t runs_load(const RunsInfo* __restrict__ info, int32_t n, int32_t* rs) {
  const int32_t nd = min(info->n_desc, GPX_RUNS_MAX - 1);
  if (threadIdx.x <= (unsigned)nd) { /* rank sort of at most 16 distinct values */
    const int32_t v = threadIdx.x == 0 ? 0 : info->start[threadIdx.x];
    int32_t r = 0;
    for (int32_t q = 1; q <= nd; q++) r += info->start[q] < v;
    rs[threadIdx.x == 0 ? 0 : r + 1] = v;
  }
  if (threadIdx.x == 0) rs[nd + 1] = n;
  __syncthreads();
  return nd + 1;
}

/* first record of group g in run [lo, hi) (ascending gidx), -1 if the run does not hold it; `hint` = where
 * it is if this run looks like the one the caller comes from */
__device__ __forceinline__ int32_t runs_find(const int32_t* __restrict__ gidx, int32_t lo, int32_t hi, int32_t g,
                                             int32_t hint) {
  if (lo >= hi) return -1;
  int32_t p = min(max(hint, lo), hi - 1);
  const int32_t v = gidx[p];
  if (v == g) {
    while (p > lo && gidx[p - 1] == g) p--;
    return p;
  }
  int32_t a = v < g ? p + 1 : lo, b = v < g ? hi : p; /* lower bound of g in [a, b) */
  while (a < b) {
    const int32_t m = a + ((b - a) >> 1);
    if (gidx[m] < g)
      a = m + 1;
    else
      b = m;
  }
  return (a < hi && gidx[a] == g) ? a : -1;
}

/* walks the votes of one group over the runs in array order */
struct RunsCursor {
  int32_t r, p, o; /* run, next position to look at, offset of the group's first vote in the last run that had one */
  __device__ __forceinline__ bool locate(const int32_t* __restrict__ gidx, const int32_t* rs, int32_t R, int32_t g) {
    for (;;) {
      if (p < rs[r + 1] && gidx[p] == g) return true;
      do {
        if (++r >= R) return false;
        p = runs_find(gidx, rs[r], rs[r + 1], g, rs[r] + o);
      } while (p < 0);
      o = p - rs[r];
    }
  }
};

struct RunsIter {
  const int32_t *gidx, *bnum, *bcoord, *slot, *acceptor, *maxcp;
  const int32_t* rs;
  int32_t R, g;
  RunsCursor rd, pk; /* read cursor; park cursor (the q-th output goes to the q-th vote's index) */
  RunsStage st;
  RunsInfo* info;
  uint32_t epoch;
  int32_t chunk, local;
  __device__ __forceinline__ bool next(Rec& out) {
    if (!rd.locate(gidx, rs, R, g)) return false;
    const int32_t i = rd.p++;
    out.idx = i;
    out.a = slot[i];
    out.b = acceptor[i];
    out.c = maxcp[i];
    out.bnum = bnum[i];
    out.bcoord = bcoord[i];
    return true;
  }
  __device__ __forceinline__ void emit(int32_t sl, int32_t x, int32_t y, int32_t z, int32_t kind) {
    pk.locate(gidx, rs, R, g); /* always there: outputs <= votes consumed */
    const int32_t i = pk.p++;
    st.O.slot()[i] = sl;
    st.O.bnum()[i] = x;
    st.O.bcoord()[i] = y;
    st.O.median()[i] = z;
    st.O.kind()[i] = (uint8_t)kind;
    st.tag[i] = epoch;
    if ((i >> GPX_DCHUNK_SHIFT) == chunk)
      local++;
    else
      atomicAdd(&st.chunk_cnt[i >> GPX_DCHUNK_SHIFT], 1);
    if (i >= rs[1]) info->need_merge = 1; /* parked outside run 0: the compaction alone does not give gidx order */
  }
};

/* The usual batch: the runs are ALIKE - every group of run 0 has exactly one vote in every run, at the same
 * offset - and the coordinator is in its steady state (SteadyGroup: all those votes answer one outstanding
 * slot at the current ballot).  A wave of run 0 whose 64 groups all look like that fetches everything it
 * needs up front (the three neighbours of the expected position in every run and the vote columns there:
 * independent loads, all in flight together) and replays from registers in a straight line; anything else
 * walks the runs with RunsIter through apply_ar_group. */
#define GPX_RUNS_FAST 5 /* runs held in registers (five replicas: BASELINE config #4) */

template <int KMAX>
__global__ __launch_bounds__(GPX_DCHUNK) void k_ar_runs(DevState S, DevScratch X, int32_t n,
                                                       const int32_t* __restrict__ gidx,
                                                       const int32_t* __restrict__ bnum,
                                                       const int32_t* __restrict__ bcoord,
                                                       const int32_t* __restrict__ slot,
                                                       const int32_t* __restrict__ acceptor,
                                                       const int32_t* __restrict__ maxcp,
                                                       uint8_t* __restrict__ status, RunsStage st,
                                                       RunsInfo* __restrict__ info, int32_t refuse) {
  __shared__ int32_t rs[GPX_RUNS_MAX + 2];
  __shared__ int32_t wsum[GPX_DCHUNK / 64];
  const int32_t i = (int32_t)blockIdx.x * GPX_DCHUNK + (int32_t)threadIdx.x;
  if (*X.unsorted == X.epoch) {
    /* not a few sorted runs: the partition pipeline launched behind does it - or, under the
     * GPX_ORDERED_REPLY_RUNS promise (no partition pipeline launched), the batch is refused whole */
    if (refuse && i < n && status) status[i] = GPX_S_UNORDERED;
    return;
  }
  const int32_t R = runs_load(info, n, rs);
  if (i == 0) atomicAdd(&X.counters[0], (unsigned long long)n);
  int32_t local = 0;
  const bool active = i < n;
  const int32_t g = active ? gidx[i] : 0;
  int32_t r = 0;
  for (int32_t q = 1; q < R; q++) r += rs[q] <= i;
  const int32_t o = i - rs[r];
  bool done = !active;
  CoordPre<KMAX> P;
  bool have_p = false;
  if (R <= GPX_RUNS_FAST) {
    int32_t sl[GPX_RUNS_FAST], ac[GPX_RUNS_FAST], cp[GPX_RUNS_FAST], bn[GPX_RUNS_FAST], bc[GPX_RUNS_FAST];
    bool ok = active && r == 0;
#pragma unroll
    for (int q = 0; q < GPX_RUNS_FAST; q++) {
      sl[q] = ac[q] = cp[q] = bn[q] = bc[q] = 0;
      if (q < R) {
        const int64_t pq = (int64_t)rs[q] + o;
        const bool inb = active && pq < rs[q + 1];
        const int32_t pc = inb ? (int32_t)pq : (active ? i : 0); /* a valid index whatever happens */
        const int32_t gq = gidx[pc];
        const int32_t gp = (inb && o > 0) ? gidx[pc - 1] : ~g;
        const int32_t gn = (inb && pq + 1 < rs[q + 1]) ? gidx[pc + 1] : ~g;
        ok = ok && inb && gq == g && gp != g && gn != g;
        sl[q] = slot[pc];
        ac[q] = acceptor[pc];
        cp[q] = maxcp[pc];
        bn[q] = bnum[pc];
        bc[q] = bcoord[pc];
      }
    }
    if (__all(!active || ok)) { /* every lane owns its group and knows where its votes are */
      if (active) {
        coord_preload<KMAX>(S, g, P);
        coord_preload_ring<KMAX>(S, g, P);
        have_p = true;
      }
      bool el = active && SteadyGroup<KMAX>::group_ok(P) && SteadyGroup<KMAX>::slot_ok(S, P, sl[0]);
#pragma unroll
      for (int q = 0; q < GPX_RUNS_FAST; q++)
        if (q < R) el = el && sl[q] == sl[0] && bn[q] == P.my_bnum && bc[q] == P.my_bcoord;
      if (__all(!active || el)) {
        if (active) {
          SteadyGroup<KMAX> sg;
          sg.init(S, g, P, sl[0]);
          bool dec = false;
          int32_t dmed = 0;
#pragma unroll
          for (int q = 0; q < GPX_RUNS_FAST; q++) {
            if (q < R) {
              int32_t med;
              if (sg.vote(ac[q], cp[q], &med)) {
                dec = true;
                dmed = med;
              }
            }
          }
          sg.finish(S, g, sl[0]);
          if (dec) { /* the group's first (only) output: parked at its first vote = this lane's record */
            st.O.slot()[i] = sl[0];
            st.O.bnum()[i] = P.my_bnum;
            st.O.bcoord()[i] = P.my_bcoord;
            st.O.median()[i] = dmed;
            st.O.kind()[i] = (uint8_t)GPX_D_DECISION;
            st.tag[i] = X.epoch;
            local = 1;
          }
        }
        done = true;
      }
    }
  }
  if (!done) {
    bool owner = i == rs[r] || gidx[i - 1] != g; /* first record of g in its run */
    for (int32_t q = 0; owner && q < r; q++) /* ... and no earlier run holds g */
      owner = runs_find(gidx, rs[q], rs[q + 1], g, rs[q] + o) < 0;
    if (owner) {
      RunsIter it;
      it.gidx = gidx;
      it.bnum = bnum;
      it.bcoord = bcoord;
      it.slot = slot;
      it.acceptor = acceptor;
      it.maxcp = maxcp;
      it.rs = rs;
      it.R = R;
      it.g = g;
      it.rd = RunsCursor{r, i, o};
      it.pk = it.rd;
      it.st = st;
      it.info = info;
      it.epoch = X.epoch;
      it.chunk = (int32_t)blockIdx.x;
      it.local = 0;
      if (!have_p) {
        coord_preload<KMAX>(S, g, P);
        coord_preload_ring<KMAX>(S, g, P);
      }
      apply_ar_group<KMAX>(S, X, g, it, status, P);
      local = it.local;
    }
  }
  /* this chunk's own parked outputs: one atomic per workgroup */
  int32_t x = local;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t tot = 0;
    for (int w = 0; w < GPX_DCHUNK / 64; w++) tot += wsum[w];
    if (tot) atomicAdd(&st.chunk_cnt[blockIdx.x], tot);
  }
}

/* parked outputs -> dense columns, chunk by chunk in record order: the caller's columns, or the merge's
 * input when an output was parked outside run 0 */
__global__ __launch_bounds__(GPX_DCHUNK) void k_emit_dec_runs(DevScratch X, int32_t n, const int32_t* __restrict__ gidx,
                                                             RunsStage st, RunsInfo* __restrict__ info,
                                                             int32_t* __restrict__ d_gidx, int32_t* __restrict__ d_slot,
                                                             int32_t* __restrict__ d_bnum, int32_t* __restrict__ d_bcoord,
                                                             int32_t* __restrict__ d_median, uint8_t* __restrict__ d_kind,
                                                             int32_t* total_out, unsigned long long* acc, int32_t refuse) {
  __shared__ int32_t rs[GPX_RUNS_MAX + 2];
  if (*X.unsorted == X.epoch) { /* the partition pipeline (k_emit_dec16) writes the outputs; refused: none */
    if (refuse && blockIdx.x == 0 && threadIdx.x == 0 && total_out) *total_out = 0;
    return;
  }
  const int32_t R = runs_load(info, n, rs);
  const int32_t w = (int32_t)blockIdx.x;
  int32_t before = 0;
  for (int32_t t = threadIdx.x; t < w; t += GPX_DCHUNK) before += st.chunk_cnt[t];
  int32_t pre;
  block_exscan_n<GPX_DCHUNK>(before, &pre);
  const int32_t i = w * GPX_DCHUNK + (int32_t)threadIdx.x;
  const bool have = i < n && st.tag[i] == X.epoch;
  int32_t tot;
  const int32_t ex = block_exscan_n<GPX_DCHUNK>(have ? 1 : 0, &tot);
  const bool merge = info->need_merge != 0;
  const int64_t o = (int64_t)pre + ex;
  if (i < n) /* outputs parked before each run start (the merge's segment bounds) */
    for (int32_t q = 1; q < R; q++)
      if (rs[q] == i) info->seg_off[q] = (int32_t)o;
  if (have) {
    if (!merge) {
      d_gidx[o] = gidx[i];
      d_slot[o] = st.O.slot()[i];
      d_bnum[o] = st.O.bnum()[i];
      d_bcoord[o] = st.O.bcoord()[i];
      d_median[o] = st.O.median()[i];
      d_kind[o] = st.O.kind()[i];
    } else {
      st.T.gidx()[o] = gidx[i];
      st.T.slot()[o] = st.O.slot()[i];
      st.T.bnum()[o] = st.O.bnum()[i];
      st.T.bcoord()[o] = st.O.bcoord()[i];
      st.T.median()[o] = st.O.median()[i];
      st.T.kind()[o] = st.O.kind()[i];
    }
  }
  if (w == (int32_t)gridDim.x - 1 && threadIdx.x == 0) {
    const int32_t total = pre + tot;
    info->total = total;
    info->seg_off[0] = 0;
    info->seg_off[R] = total;
    if (total_out) *total_out = total;
    if (acc) atomicAdd(acc, (unsigned long long)total);
  }
}

/* number of entries of the ascending segment a[lo, hi) that are < g (UPPER: <= g) */
template <bool UPPER>
__device__ __forceinline__ int32_t seg_bound(const int32_t* __restrict__ a, int32_t lo, int32_t hi, int32_t g) {
  int32_t x = lo, y = hi;
  while (x < y) {
    const int32_t m = x + ((y - x) >> 1);
    if (UPPER ? a[m] <= g : a[m] < g)
      x = m + 1;
    else
      y = m;
  }
  return x - lo;
}

/* the compacted outputs are R segments (one per run the outputs were parked in), each ascending by
 * (gidx, vote order); the contract's order is their merge, a group's entries of an earlier segment first */
__global__ __launch_bounds__(GPX_BLOCK) void k_merge_runs(DevScratch X, int32_t n, RunsStage st,
                                                         const RunsInfo* __restrict__ info,
                                                         int32_t* __restrict__ d_gidx, int32_t* __restrict__ d_slot,
                                                         int32_t* __restrict__ d_bnum, int32_t* __restrict__ d_bcoord,
                                                         int32_t* __restrict__ d_median, uint8_t* __restrict__ d_kind) {
  if (*X.unsorted == X.epoch || !info->need_merge) return;
  const int32_t R = min(info->n_desc, GPX_RUNS_MAX - 1) + 1;
  const int32_t total = info->total;
  const int32_t* tg = st.T.gidx();
  for (int32_t t = blockIdx.x * GPX_BLOCK + threadIdx.x; t < total; t += gridDim.x * GPX_BLOCK) {
    const int32_t g = tg[t];
    int32_t a = 0;
    for (int32_t q = 1; q < R; q++) a += info->seg_off[q] <= t;
    /* runs without parked outputs give empty segments (equal offsets): `a` is the last segment starting at or before t */
    int32_t rank = t - info->seg_off[a];
    for (int32_t b = 0; b < R; b++) {
      if (b == a) continue;
      const int32_t lo = info->seg_off[b], hi = info->seg_off[b + 1];
      rank += b < a ? seg_bound<true>(tg, lo, hi, g) : seg_bound<false>(tg, lo, hi, g);
    }
    d_gidx[rank] = g;
    d_slot[rank] = st.T.slot()[t];
    d_bnum[rank] = st.T.bnum()[t];
    d_bcoord[rank] = st.T.bcoord()[t];
    d_median[rank] = st.T.median()[t];
    d_kind[rank] = st.T.kind()[t];
  }
}
