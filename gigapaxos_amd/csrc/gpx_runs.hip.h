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
 *                   1's: exactly arrival order.  Consecutive lanes of run 0 own ascending groups: coalesced
 *                   state accesses.  The q-th output of a group is PARKED IN THE CALLER'S OUTPUT COLUMNS at
 *                   the array index of the group's q-th vote (the columns hold n entries: gpx.h).
 *                   REGULAR batch - every record of run 0 is a group in the coordinator's steady state
 *                   whose votes sit at the same offset in every run, and every one of them decided: the
 *                   decision of record i of run 0 is parked at index i, i.e. the columns already hold the
 *                   decisions dense and in gidx order.  Nothing is left to do.
 *   k_emit_dec_runs regular: publishes n_out and returns.  Otherwise: the parked outputs -> a dense
 *                   staging block in record order ...
 *   k_merge_runs    ... and from there back into the caller's columns: a plain copy when everything was
 *                   parked inside run 0 (record order IS gidx order), else (a group absent from run 0, or
 *                   more outputs than votes in run 0) the staged outputs are up to GPX_RUNS_MAX ascending
 *                   segments and every entry computes its rank in their merge (binary searches).
 */
#pragma once
#include "gpx_ar16.hip.h"
#include "gpx_direct.hip.h"
#include "gpx_one.hip.h"

#define GPX_RUNS_MAX 16 /* PC.MAX_GROUP_SIZE acceptors: PaxosConfig.java:532 */

/* per-call facts about the batch; two of them, used alternately: call N's k_runs_check clears the one
 * call N + 1 will use (its last user, call N - 1, has finished: stream order) */
struct RunsInfo {
  int32_t n_desc;                  /* descents found so far (atomic) */
  int32_t need_merge;              /* an output was parked outside run 0 */
  int32_t total;                   /* outputs of the call (k_emit_dec_runs) */
  int32_t general_used;            /* a group went through the general replay, or a straight-line one did not decide */
  int32_t pad[4];
  int32_t start[GPX_RUNS_MAX + 1]; /* start[0] = 0; the others in the order the atomics gave: sorted by the readers */
  int32_t seg_off[GPX_RUNS_MAX + 1]; /* staged outputs parked before each run start (k_emit_dec_runs) */
};

/* the caller's decision columns (also the parking area) */
struct DecCols {
  int32_t *gidx, *slot, *bnum, *bcoord, *median;
  uint8_t* kind;
  __device__ __forceinline__ void put(int64_t i, int32_t g, int32_t sl, int32_t bn, int32_t bc, int32_t md,
                                      int32_t kd) const {
    gidx[i] = g;
    slot[i] = sl;
    bnum[i] = bn;
    bcoord[i] = bc;
    median[i] = md;
    kind[i] = (uint8_t)kd;
  }
};

struct RunsStage {
  DecCols D;          /* the caller's columns: parked outputs by record index, then the final outputs */
  uint32_t* tag;      /* [n] == epoch: record i holds a parked output */
  int32_t* chunk_cnt; /* [ceil(n / 1024)] parked outputs per chunk; zeroed by k_runs_check */
  Stage16 T;          /* dense staging between k_emit_dec_runs and k_merge_runs (irregular batches only) */
};

/* order check of a vote batch: at most GPX_RUNS_MAX ascending runs, every index in range */
#define GPX_RC_ITEMS 16 /* records per lane */
__global__ __launch_bounds__(GPX_OC_BLOCK) void k_runs_check(int32_t n, const int32_t* __restrict__ gidx, int32_t G,
                                                          DevScratch X, uint8_t* __restrict__ status,
                                                          RunsInfo* __restrict__ info, RunsInfo* __restrict__ next_info,
                                                          int32_t* __restrict__ zero, int32_t nzero,
                                                          uint32_t* __restrict__ arrive, int32_t* __restrict__ n_out,
                                                          unsigned long long* __restrict__ acc, int32_t refuse) {
  const int64_t t = (int64_t)blockIdx.x * GPX_OC_BLOCK + threadIdx.x;
  const int64_t i0 = t * GPX_RC_ITEMS;
  for (int64_t z = t; z < nzero; z += (int64_t)gridDim.x * GPX_OC_BLOCK) zero[z] = 0;
  if (blockIdx.x == 0 && threadIdx.x < (int)(sizeof(RunsInfo) / 4)) ((int32_t*)next_info)[threadIdx.x] = 0;
  bool bad = false;
  uint32_t desc = 0; /* bit q: gidx[i0 + q] > gidx[i0 + q + 1] */
  if (i0 < n) {
    int32_t g[GPX_RC_ITEMS + 1];
    const bool full = i0 + GPX_RC_ITEMS < n;
    if (full && !((uintptr_t)gidx & 15)) {
#pragma unroll
      for (int v = 0; v < GPX_RC_ITEMS / 4; v++) {
        const I4 a = *(const I4*)(gidx + i0 + 4 * v);
        g[4 * v] = a.x;
        g[4 * v + 1] = a.y;
        g[4 * v + 2] = a.z;
        g[4 * v + 3] = a.w;
      }
      g[GPX_RC_ITEMS] = gidx[i0 + GPX_RC_ITEMS];
    } else {
#pragma unroll
      for (int q = 0; q <= GPX_RC_ITEMS; q++) g[q] = (i0 + q < n) ? gidx[i0 + q] : INT32_MAX;
    }
    unsigned long long stw0 = 0, stw1 = 0;
#pragma unroll
    for (int q = 0; q < GPX_RC_ITEMS; q++) {
      if (i0 + q < n) {
        const bool oob = (uint32_t)g[q] >= (uint32_t)G;
        bad |= oob;
        if (i0 + q + 1 < n && g[q] > g[q + 1]) desc |= 1u << q;
        if (oob) {
          if (q < 8)
            stw0 |= (unsigned long long)GPX_S_NOGROUP << (8 * q);
          else
            stw1 |= (unsigned long long)GPX_S_NOGROUP << (8 * (q - 8));
        }
      }
    }
    if (!status) {
    } else if (full && !((uintptr_t)status & 7)) {
      *(unsigned long long*)(status + i0) = stw0; /* GPX_S_OK == 0 */
      *(unsigned long long*)(status + i0 + 8) = stw1;
    } else {
      for (int q = 0; q < GPX_RC_ITEMS; q++)
        if (i0 + q < n) status[i0 + q] = (uint8_t)((q < 8 ? stw0 >> (8 * q) : stw1 >> (8 * (q - 8))) & 0xffu);
    }
  }
  /* a shuffled batch has thousands of descents per workgroup: judged here, without touching the shared counter */
  const int32_t nd = __syncthreads_count(desc != 0);
  bad = __syncthreads_or(bad) || nd > GPX_RUNS_MAX - 1;
  if (!bad && desc) {
#pragma unroll
    for (int q = 0; q < GPX_RC_ITEMS; q++) {
      if ((desc >> q) & 1u) {
        const int32_t k = atomicAdd(&info->n_desc, 1);
        if (k < GPX_RUNS_MAX - 1)
          __hip_atomic_store(&info->start[k + 1], (int32_t)(i0 + q + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          bad = true;
      }
    }
  }
  const bool wg_bad = __syncthreads_or(bad);
  const bool wg_desc = nd != 0; /* this workgroup recorded run starts (every thread knows: __syncthreads_count above) */
  if (threadIdx.x == 0) {
    if (wg_bad) atomicMax(X.unsorted, X.epoch);
    /* The LAST workgroup to finish publishes what the whole check found, so that the usual batch needs no other
     * kernel for its count: a REGULAR batch decides once per record of run 0 (k_ar_runs overwrites the count with -1
     * when the batch turns out otherwise), a refused one has no output.  Arrival counters: 16 workgroups share one,
     * 128 bytes apart (device-scope atomics on one cache line are serial at ~16 ns each: gpx_one.hip.h), then one
     * for the groups of 16.  A workgroup that recorded a run start fences first: the last one reads them all. */
    if (wg_desc || wg_bad) __threadfence(); /* ... and one that raised *X.unsorted */
    const int32_t nwg = (int32_t)gridDim.x, grp = (int32_t)blockIdx.x >> 4;
    const uint32_t size1 = (uint32_t)min(16, nwg - (grp << 4));
    bool last = false;
    if (atomicAdd(&arrive[32 * (1 + grp)], 1u) + 1u == size1) {
      __hip_atomic_store(&arrive[32 * (1 + grp)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (atomicAdd(&arrive[0], 1u) + 1u == (uint32_t)((nwg + 15) >> 4)) {
        __hip_atomic_store(&arrive[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = true;
      }
    }
    if (last && n_out) {
      __threadfence();
      const bool unsorted = __hip_atomic_load(X.unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X.epoch;
      if (!unsorted) {
        const int32_t nd = min(__hip_atomic_load(&info->n_desc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), GPX_RUNS_MAX - 1);
        int32_t len0 = n;
        for (int32_t q = 1; q <= nd; q++)
          len0 = min(len0, __hip_atomic_load(&info->start[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        *n_out = len0;
        info->total = len0; /* what was counted already (k_emit_dec_runs adds the difference for an irregular batch) */
        if (acc) atomicAdd(acc, (unsigned long long)len0);
      } else if (refuse) {
        *n_out = 0;
      }
    }
  }
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
/* length of run 0 without sorting: the smallest start (n if there is one run) */
__device__ __forceinline__ int32_t runs_len0(const RunsInfo* __restrict__ info, int32_t n) {
  const int32_t nd = min(info->n_desc, GPX_RUNS_MAX - 1);
  int32_t m = n;
  for (int32_t q = 1; q <= nd; q++) m = min(m, info->start[q]);
  return m;
}
/* REGULAR: only the straight-line path parked outputs, one per record of run 0 (header).  No counter is
 * needed for that (a counter every workgroup adds to is a chain of same-address atomics: it made the kernel
 * 2.5x slower): a record of run 0 is either a straight-line lane - which raises general_used when it does
 * not decide - or belongs to a group whose owner took the general replay, which raises it too. */
__device__ __forceinline__ bool runs_regular(const RunsInfo* __restrict__ info, int32_t n) {
  (void)n;
  return !info->general_used;
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

/* does run [lo, hi) hold group g at all? (the ownership test of the later runs' lanes: one load when the
 * runs are alike) */
__device__ __forceinline__ bool runs_has(const int32_t* __restrict__ gidx, int32_t lo, int32_t hi, int32_t g,
                                         int32_t hint) {
  if (lo >= hi) return false;
  const int32_t p = min(max(hint, lo), hi - 1);
  const int32_t v = gidx[p];
  if (v == g) return true;
  int32_t a = v < g ? p + 1 : lo, b = v < g ? hi : p;
  while (a < b) {
    const int32_t m = a + ((b - a) >> 1);
    if (gidx[m] < g)
      a = m + 1;
    else
      b = m;
  }
  return a < hi && gidx[a] == g;
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
  uint8_t* st_ok = nullptr;  /* small calls: no prefill pass ran - the replaying lane marks a vote OK before it judges it */
  bool count_chunks = true;  /* ... and nothing is counted per chunk (k_runs_count does it for the batch that needs it) */
  __device__ __forceinline__ bool next(Rec& out) {
    if (!rd.locate(gidx, rs, R, g)) return false;
    const int32_t i = rd.p++;
    if (st_ok) st_ok[i] = GPX_S_OK;
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
    st.D.put(i, g, sl, x, y, z, kind);
    st.tag[i] = epoch;
    if (!count_chunks) {
    } else if ((i >> GPX_DCHUNK_SHIFT) == chunk) {
      local++;
    } else {
      atomicAdd(&st.chunk_cnt[i >> GPX_DCHUNK_SHIFT], 1);
    }
    if (i >= rs[1]) info->need_merge = 1; /* parked outside run 0: record order alone is not gidx order */
  }
};

/* The usual group: exactly one vote in every run, at the same offset as in run 0, and the coordinator in its
 * steady state (SteadyGroup: those votes answer one outstanding slot at the current ballot).  A lane of run 0
 * fetches everything up front (the three neighbours of the expected position in every run and the vote
 * columns there: independent loads, all in flight together), then the state, and replays from registers in a
 * straight line; any other group walks the runs with RunsIter through apply_ar_group. */
#define GPX_RUNS_FAST 5 /* runs held in registers (five replicas: BASELINE config #4) */
/* 256-thread workgroups: with 1024 (the chunk) the kernel's 65 VGPRs leave room for ONE workgroup per CU -
 * four waves per SIMD, each a chain of three dependent load waves - and the kernel took 87 us per 3 M votes */
#define GPX_RBLOCK 256

/* SMALL = the call in ONE launch (round 4: at most 65,536 votes, the workgroups exchanging the verdict through a ticket
 * each; round 5: any grid the host knows to be resident - xchg_ctl, 131,072 votes for one engine on an MI355X - and
 * grid_exchange's sixteen arrival counters, gpx_one.hip.h): every workgroup judges its own 256 records - in range; a
 * descent = a run start, appended to info->start as k_runs_check does - raises *X.unsorted for a batch that is no few
 * runs in range, and meets the others once; then every workgroup reads the same verdict and the same run starts.  The
 * lane that replays a vote marks its status (no prefill pass), nothing is counted per chunk, and the LAST workgroup to
 * finish (two levels of arrival counters, 16 workgroups per counter, a cache line apart) publishes the count - or -1
 * for a batch that needs the compaction pass - and leaves the run starts in `info` for that pass.
 * EARLY (calls of at most 65,536 votes): the lane's group state is requested before the exchange.
 * !SMALL: k_runs_check has run before - larger batches, or a device shared by many streams. */
template <int KMAX, bool SMALL = false, bool EARLY = false>
__global__ __launch_bounds__(GPX_RBLOCK) void k_ar_runs(DevState S, DevScratch X, int32_t n,
                                                       const int32_t* __restrict__ gidx,
                                                       const int32_t* __restrict__ bnum,
                                                       const int32_t* __restrict__ bcoord,
                                                       const int32_t* __restrict__ slot,
                                                       const int32_t* __restrict__ acceptor,
                                                       const int32_t* __restrict__ maxcp,
                                                       uint8_t* __restrict__ status, RunsStage st,
                                                       RunsInfo* __restrict__ info, int32_t refuse,
                                                       int32_t* __restrict__ n_out, RunsInfo* __restrict__ next_info,
                                                       uint32_t* __restrict__ arrive, unsigned long long* __restrict__ acc,
                                                       GridXchg Q) {
  __shared__ int32_t rs[GPX_RUNS_MAX + 2];
  __shared__ int32_t wsum[GPX_RBLOCK / 64];
  const int32_t i = (int32_t)blockIdx.x * GPX_RBLOCK + (int32_t)threadIdx.x;
  int32_t R;
  CoordPre<KMAX> P;
  bool have_p = false;
#ifdef GPX_SAR_TRACE
  if (SMALL) SAR_STAMP(blockIdx.x, 0);
#endif
  if (SMALL) {
    /* EARLY (the small calls' build): the state of this chunk's groups is REQUESTED before the verdict is exchanged (as
     * k_ac_pers does): round 4's exchange was 10 of the small kernel's 11.5 us because nothing else was in flight
     * behind it (profiles/r04_sar_trace_6_tiny.txt).  Every lane asks for the state of its own record's group - the
     * lanes of run 0 will own those groups, the others have asked in vain (their lines are in L2 for the owner) */
    if (EARLY && i < n) {
      const int32_t g0 = gidx[i];
      if ((uint32_t)g0 < (uint32_t)S.G) {
        coord_preload<KMAX>(S, g0, P);
        coord_preload_ring<KMAX>(S, g0, P);
        have_p = true;
      }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(sizeof(RunsInfo) / 4)) ((int32_t*)next_info)[threadIdx.x] = 0;
    {
      bool bad = false, desc = false;
      if (i < n) {
        const int32_t gi = gidx[i];
        const int32_t gp = i > 0 ? gidx[i - 1] : INT32_MIN;
        bad = (uint32_t)gi >= (uint32_t)S.G;
        desc = gp > gi;
      }
      const int32_t nd_wg = __syncthreads_count(desc);
      bad = __syncthreads_or(bad) || nd_wg > GPX_RUNS_MAX - 1;
      if (!bad && desc) { /* (at most sixteen lanes of the whole batch) */
        const int32_t k = atomicAdd(&info->n_desc, 1);
        if (k < GPX_RUNS_MAX - 1) {
          __hip_atomic_store(&info->start[k + 1], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); /* performed before this workgroup's arrival (behind the barrier) */
        } else {
          bad = true;
        }
      }
      if (__syncthreads_or(bad) && threadIdx.x == 0) { /* rare: performed before the arrival */
        const uint32_t old = atomicMax(X.unsorted, X.epoch);
        asm volatile("" ::"v"(old));
      }
    }
    /* (the barrier above and grid_exchange's fence order this workgroup's run starts and verdict before its arrival) */
    if (grid_exchange(X, Q, ONE_NONE) == 0u) {
      /* somebody gave up waiting (nobody else writes a verdict here): apply nothing - and do NOT raise *X.unsorted: under
       * the hint the partition pipeline launched behind would take the whole batch, on top of whatever the workgroups
       * that got through have applied.  The call ends in GPX_EDEVICE. */
      if (refuse && i < n && status) status[i] = GPX_S_UNORDERED;
      return;
    }
    if (__hip_atomic_load(X.unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X.epoch) {
      /* not a few ascending runs in range: refused whole under the promise, else the partition pipeline (or the
       * one-launch kernel of small calls) launched behind takes the batch */
      if (refuse) {
        if (i < n && status) status[i] = GPX_S_UNORDERED;
        if (blockIdx.x == 0 && threadIdx.x == 0 && n_out) *n_out = 0;
      }
      return;
    }
    R = runs_load(info, n, rs);
#ifdef GPX_SAR_TRACE
    SAR_STAMP(blockIdx.x, 1); /* column judged, run starts sorted */
#endif
  } else {
    if (*X.unsorted == X.epoch) {
      /* not a few sorted runs: the partition pipeline launched behind does it - or, under the
       * GPX_ORDERED_REPLY_RUNS promise (no partition pipeline launched), the batch is refused whole */
      if (refuse && i < n && status) status[i] = GPX_S_UNORDERED;
      return; /* (k_runs_check's last workgroup wrote n_out = 0 for the refused batch) */
    }
    R = runs_load(info, n, rs);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&X.counters[0], (unsigned long long)n);
  const int32_t my_chunk = (int32_t)(((int64_t)blockIdx.x * GPX_RBLOCK) >> GPX_DCHUNK_SHIFT); /* (!SMALL: the only chunk) */
  int32_t local = 0;
  bool irregular = false; /* this lane saw why the columns are not dense as parked (travels with the arrival counters) */
  const bool active = i < n;
  const int32_t g = active ? gidx[i] : 0;
  int32_t r = 0;
  for (int32_t q = 1; q < R; q++) r += rs[q] <= i;
  const int32_t o = i - rs[r];
  const bool in0 = active && r == 0;
  bool done = !active;
  /* a lane of a later run (two thirds of a three-replica batch) usually finds its group at the same offset of
   * run 0 and has nothing to do: that word is fetched together with the lane's own, not after it */
  if (active && r > 0 && o < rs[1] && gidx[o] == g) done = true;
  /* the lanes of the later runs (two thirds of a three-replica batch) skip the speculative fetches: they
   * only test whether an earlier run holds their group */
  if (R <= GPX_RUNS_FAST && __any(in0)) {
    int32_t sl[GPX_RUNS_FAST], ac[GPX_RUNS_FAST], cp[GPX_RUNS_FAST], bn[GPX_RUNS_FAST], bc[GPX_RUNS_FAST];
    bool ok = in0;
#pragma unroll
    for (int q = 0; q < GPX_RUNS_FAST; q++) {
      sl[q] = ac[q] = cp[q] = bn[q] = bc[q] = 0;
      if (q < R && in0) {
        const int64_t pq = (int64_t)rs[q] + o;
        const bool inb = pq < rs[q + 1];
        const int32_t pc = inb ? (int32_t)pq : i; /* a valid index whatever happens */
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
    if (ok) { /* this lane owns its group and knows where its votes are */
      if (!have_p) {
        coord_preload<KMAX>(S, g, P);
        coord_preload_ring<KMAX>(S, g, P);
        have_p = true;
      }
      bool el = SteadyGroup<KMAX>::group_ok(P) && SteadyGroup<KMAX>::slot_ok(S, P, sl[0]);
#pragma unroll
      for (int q = 0; q < GPX_RUNS_FAST; q++)
        if (q < R) el = el && sl[q] == sl[0] && bn[q] == P.my_bnum && bc[q] == P.my_bcoord;
      if (el) {
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
        if (SMALL && status) {
#pragma unroll
          for (int q = 0; q < GPX_RUNS_FAST; q++)
            if (q < R) status[rs[q] + o] = GPX_S_OK; /* this lane's votes: one per run, at its own offset */
        }
        if (dec) { /* the group's first (only) output: parked at its first vote = this lane's record */
          st.D.put(i, g, sl[0], P.my_bnum, P.my_bcoord, dmed, GPX_D_DECISION);
          st.tag[i] = X.epoch;
          local += 1;
        } else {
          info->general_used = 1; /* a hole in run 0: the columns are not dense */
          irregular = true;
        }
        done = true;
      }
    }
  }
#ifdef GPX_SAR_TRACE
  if (SMALL) SAR_STAMP(blockIdx.x, 2); /* straight-line replay done (thread 0) */
#endif
  if (!done) {
    /* the owner of a group: no earlier run holds it (tested first: one load for a lane of an alike later
     * run), and it is the group's first record in its run */
    bool owner = true;
    for (int32_t q = 0; owner && q < r; q++) owner = !runs_has(gidx, rs[q], rs[q + 1], g, rs[q] + o);
    owner = owner && (i == rs[r] || gidx[i - 1] != g);
    if (owner) {
      info->general_used = 1;
      irregular = true;
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
      it.chunk = my_chunk;
      it.local = 0;
      if (SMALL) {
        it.st_ok = status;
        it.count_chunks = false;
      }
      if (!have_p) {
        coord_preload<KMAX>(S, g, P);
        coord_preload_ring<KMAX>(S, g, P);
      }
      apply_ar_group<KMAX>(S, X, g, it, status, P);
      local += it.local;
    }
  }
  /* this workgroup's own parked outputs: one atomic per workgroup */
  int32_t x = local;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  const bool wave_irregular = __any(irregular) != 0;
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = x | (wave_irregular ? (int32_t)0x40000000 : 0);
  __syncthreads();
#ifdef GPX_SAR_TRACE
  if (SMALL) SAR_STAMP(blockIdx.x, 3); /* every lane done */
#endif
  if (threadIdx.x == 0) {
    int32_t tot = 0;
    bool wg_irregular = false;
    for (int w = 0; w < GPX_RBLOCK / 64; w++) {
      tot += wsum[w] & 0x3fffffff;
      wg_irregular |= (wsum[w] & 0x40000000) != 0;
    }
    if (!SMALL) {
      if (tot) atomicAdd(&st.chunk_cnt[my_chunk], tot);
      /* a REGULAR batch is finished: k_runs_check's last workgroup wrote its count (one decision per record of run
       * 0, dense as parked).  A workgroup that saw otherwise says so: k_emit_dec_runs / k_merge_runs follow (at
       * once, or on gpx_compact_last_dev under GPX_LAZY_OUTPUTS) */
      if (wg_irregular && n_out) __hip_atomic_store(n_out, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const int32_t nwg = (int32_t)gridDim.x, grp = (int32_t)blockIdx.x >> 4;
      const uint32_t size1 = (uint32_t)min(16, nwg - (grp << 4));
      bool last = false, any = wg_irregular;
      const uint32_t old1 = atomicAdd(&arrive[32 * (1 + grp)], 1u + (wg_irregular ? 0x10000u : 0u));
      if ((old1 & 0xffffu) + 1u == size1) {
        any |= (old1 >> 16) != 0;
        __hip_atomic_store(&arrive[32 * (1 + grp)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t old0 = atomicAdd(&arrive[0], 1u + (any ? 0x10000u : 0u));
        if ((old0 & 0xffffu) + 1u == (uint32_t)((nwg + 15) >> 4)) {
          any |= (old0 >> 16) != 0;
          __hip_atomic_store(&arrive[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          last = true;
        }
      }
      if (last) { /* what the compaction pass reads (it runs behind this kernel, if at all) and the count */
        info->n_desc = R - 1;
        for (int32_t q = 1; q < R; q++) info->start[q] = rs[q];
        info->general_used = any ? 1 : 0;
        info->total = any ? 0 : rs[1];
        if (n_out) *n_out = any ? -1 : rs[1];
        if (!any && acc) atomicAdd(acc, (unsigned long long)rs[1]);
      }
    }
  }
}

/* small calls, irregular batches only: parked outputs per 1024-record chunk (k_ar_runs<.., SMALL> counts nothing) */
__global__ __launch_bounds__(GPX_DCHUNK) void k_runs_count(DevScratch X, int32_t n, RunsStage st, const RunsInfo* __restrict__ info) {
  if (!info->general_used || *X.unsorted == X.epoch) return;
  const int32_t i = (int32_t)blockIdx.x * GPX_DCHUNK + (int32_t)threadIdx.x;
  const int32_t c = __syncthreads_count(i < n && st.tag[i] == X.epoch);
  if (threadIdx.x == 0) st.chunk_cnt[blockIdx.x] = c;
}

/* regular batch: the caller's columns are final already - publish the count.  Otherwise: parked outputs ->
 * the dense staging block, chunk by chunk in record order */
__global__ __launch_bounds__(GPX_DCHUNK) void k_emit_dec_runs(DevScratch X, int32_t n, int32_t nchunks, RunsStage st,
                                                             RunsInfo* __restrict__ info, int32_t* total_out,
                                                             unsigned long long* acc, int32_t refuse,
                                                             int32_t published = 0) {
  __shared__ int32_t rs[GPX_RUNS_MAX + 2];
  const bool unsorted = *X.unsorted == X.epoch, regular = runs_regular(info, n); /* one round trip */
  if (published && regular && !unsorted) return; /* k_ar_runs' last workgroup has written the count */
  if (unsorted) { /* the partition pipeline (k_emit_dec16) writes the outputs; refused: none */
    if (refuse && blockIdx.x == 0 && threadIdx.x == 0 && total_out) *total_out = 0;
    return;
  }
  if (regular) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const int32_t total = runs_len0(info, n); /* one decision per record of run 0 */
      if (total_out) *total_out = total;
      if (acc) atomicAdd(acc, (unsigned long long)total);
    }
    return;
  }
  const bool merge = info->need_merge != 0;
  const int32_t R = runs_load(info, n, rs);
  for (int32_t w = (int32_t)blockIdx.x; w < nchunks; w += (int32_t)gridDim.x) { /* persistent workgroups: GPX_EMIT_GRID */
  /* nothing parked in this chunk (the chunks of the later runs, usually): nothing to place - only the last
   * chunk (the total) and, before a merge, the chunks that hold a run start (segment bounds) go on */
  if (!merge && st.chunk_cnt[w] == 0 && w != nchunks - 1) continue;
  int32_t before = 0;
  for (int32_t t = threadIdx.x; t < w; t += GPX_DCHUNK) before += st.chunk_cnt[t];
  int32_t pre;
  block_exscan_n<GPX_DCHUNK>(before, &pre);
  const int32_t i = w * GPX_DCHUNK + (int32_t)threadIdx.x;
  const bool have = i < n && st.tag[i] == X.epoch;
  int32_t tot;
  const int32_t ex = block_exscan_n<GPX_DCHUNK>(have ? 1 : 0, &tot);
  const int64_t o = (int64_t)pre + ex;
  if (i < n) /* outputs parked before each run start (the merge's segment bounds) */
    for (int32_t q = 1; q < R; q++)
      if (rs[q] == i) info->seg_off[q] = (int32_t)o;
  if (have) {
    st.T.gidx()[o] = st.D.gidx[i];
    st.T.slot()[o] = st.D.slot[i];
    st.T.bnum()[o] = st.D.bnum[i];
    st.T.bcoord()[o] = st.D.bcoord[i];
    st.T.median()[o] = st.D.median[i];
    st.T.kind()[o] = st.D.kind[i];
  }
  if (w == nchunks - 1 && threadIdx.x == 0) {
    const int32_t total = pre + tot;
    const int32_t counted = published ? info->total : 0; /* k_runs_check counted a regular batch's decisions already */
    info->total = total;
    info->seg_off[0] = 0;
    info->seg_off[R] = total;
    if (total_out) *total_out = total;
    if (acc) atomicAdd(acc, (unsigned long long)(long long)(total - counted));
  }
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

/* irregular batch: the staged outputs back into the caller's columns.  Parked inside run 0 only: record order
 * is gidx order - a copy.  Otherwise the staged outputs are R segments (one per run the outputs were parked
 * in), each ascending by (gidx, vote order); the contract's order is their merge, a group's entries of an
 * earlier segment first. */
__global__ __launch_bounds__(GPX_BLOCK) void k_merge_runs(DevScratch X, int32_t n, RunsStage st,
                                                         const RunsInfo* __restrict__ info) {
  const bool unsorted = *X.unsorted == X.epoch, regular = runs_regular(info, n); /* one round trip */
  if (unsorted || regular) return;
  const int32_t R = min(info->n_desc, GPX_RUNS_MAX - 1) + 1;
  const int32_t total = info->total;
  const bool merge = info->need_merge != 0;
  const int32_t* tg = st.T.gidx();
  for (int32_t t = blockIdx.x * GPX_BLOCK + threadIdx.x; t < total; t += gridDim.x * GPX_BLOCK) {
    const int32_t g = tg[t];
    int32_t rank = t;
    if (merge) {
      int32_t a = 0;
      for (int32_t q = 1; q < R; q++) a += info->seg_off[q] <= t;
      /* runs without parked outputs give empty segments (equal offsets): `a` is the last segment starting at or before t */
      rank = t - info->seg_off[a];
      for (int32_t b = 0; b < R; b++) {
        if (b == a) continue;
        const int32_t lo = info->seg_off[b], hi = info->seg_off[b + 1];
        rank += b < a ? seg_bound<true>(tg, lo, hi, g) : seg_bound<false>(tg, lo, hi, g);
      }
    }
    st.D.put(rank, g, st.T.slot()[t], st.T.bnum()[t], st.T.bcoord()[t], st.T.median()[t], st.T.kind()[t]);
  }
}
