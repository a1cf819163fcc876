/*
 * gpx_one.hip.h — ordered PROPOSE / ACCEPT / COMMIT batches in ONE launch (round 4).
 *
 * Under the gpx_engine_set_ordered_batches promise a batch of more than 65,536 records used to cost three or
 * four dependent launches: k_order_check (the verdict), the direct kernel (the work), k_emit_runs_direct and
 * k_copy_runs (which find nothing to do for a usual batch).  On this chip a dependent launch costs 5-6 us whatever
 * it does - the previous kernel's dirty lines leave the XCDs' L2s first - and they were 113 of the full round's
 * 279 us (profiles/r03_bench_full_round.json).  Here the verdict travels INSIDE the work kernel:
 *
 *   order      every workgroup checks its own 256 records against their left neighbours (words it loads anyway)
 *              and publishes the first violating index it saw; a decoupled look-back over per-workgroup words
 *              {epoch, state, first bad index} - MIN instead of the usual sum - tells it the first violation of
 *              the whole PREFIX before it.  A record is applied iff it lies before the batch's first violation:
 *              the run of a group never straddles that index (equal neighbours are no violation), and a second
 *              run of the same group can only start behind a descent, i.e. behind the first violation - so no
 *              two lanes ever own the same group, without a global verdict.  The words wait for the
 *              predecessors' first LOADS, not for their work: the group state is requested before the wait.
 *              Promise broken: records from the first violation on are refused (GPX_S_UNORDERED, outputs zero,
 *              no state change), the records before it are applied - include/gpx.h.
 *   workgroups are taken in blockIdx order.  A workgroup only waits for lower block indices, and the dispatcher
 *              hands those out first (per XCD, in order): the lowest unfinished workgroup is always resident and
 *              waits for nobody.  (A ticket per workgroup - the wire decode's way - is a chain of same-address
 *              atomics at 16 ns each: 3,907 of them are 62 us, twice this kernel.)  The wait is BOUNDED all the
 *              same: a workgroup that gives up declares the batch broken at its own first record, which keeps
 *              "a prefix is applied, the rest refused" - a lost tail, never a hang.
 *   outputs    execution runs are parked at their records' indices and tagged, as in gpx_direct.hip.h.  The last
 *              workgroup to finish (two-level arrival counters) knows whether the batch was REGULAR - ACCEPTs
 *              released no commit; every COMMIT executed exactly one run - and publishes n_runs: the usual case
 *              needs no compaction at all.  Otherwise it raises D.mark and writes n_runs = -1: the compaction
 *              kernels (k_one_count, k_emit_runs_direct, k_copy_runs) follow at once (the default) or when the
 *              caller asks for dense columns (GPX_LAZY_OUTPUTS, gpx_compact_last_dev).
 */
#pragma once
#include "gpx_direct.hip.h"

#define ONE_AGG 1ull
#define ONE_PRE 2ull
#define ONE_NONE 0xffffffffu
#define ONE_SPIN_LIMIT (1u << 18) /* polls of one look-back step before the workgroup gives up (~ a second) */

struct OneCtl {
  unsigned long long* ord; /* [workgroups] epoch << 34 | state << 32 | first violating index (ONE_NONE: none) */
  uint32_t* done1;         /* [workgroups / 64 + 1] arrivals per 64 workgroups | irregular ones << 16; zero between calls */
  uint32_t* done0;         /* [1] ... of the groups of 64 */
  uint32_t epoch;          /* 30 bits, never 0 (the words are cleared when it wraps) */
};

__device__ __forceinline__ unsigned long long one_word(uint32_t epoch, unsigned long long st, uint32_t v) {
  return ((unsigned long long)epoch << 34) | (st << 32) | (unsigned long long)v;
}

/* first violating index among the workgroups before `w` (ONE_NONE: the prefix keeps the order); called by one
 * whole wave.  The walk stops at the nearest workgroup that already knows its inclusive prefix. */
__device__ __forceinline__ uint32_t one_lookback_min(const unsigned long long* __restrict__ st, int32_t w,
                                                     uint32_t epoch, bool* timed_out) {
  const int32_t lane = (int32_t)(threadIdx.x & 63);
  uint32_t acc = ONE_NONE;
  uint32_t spins = 0;
  for (int32_t hi = w - 1; hi >= 0; hi -= 64) {
    const int32_t j = hi - lane; /* lane 0 = the nearest predecessor of this step */
    unsigned long long v = j >= 0 ? __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    bool need = j >= 0 && (uint32_t)(v >> 34) != epoch;
    unsigned long long pre_mask;
    for (;;) {
      pre_mask = __ballot(!need && j >= 0 && ((v >> 32) & 3ull) == ONE_PRE);
      const unsigned long long wait_mask = __ballot(need);
      if (pre_mask) {
        const unsigned long long nearer = (pre_mask & (0ull - pre_mask)) - 1ull;
        if ((wait_mask & nearer) == 0) break;
      } else if (wait_mask == 0) {
        break;
      }
      if (++spins > ONE_SPIN_LIMIT) {
        *timed_out = true;
        return acc;
      }
      __builtin_amdgcn_s_sleep(4);
      if (need) {
        v = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 34) == epoch) need = false;
      }
    }
    const int32_t first_pre = pre_mask ? (__ffsll((long long)pre_mask) - 1) : 64;
    uint32_t x = (j >= 0 && lane <= first_pre) ? (uint32_t)v : ONE_NONE;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t y = (uint32_t)__shfl_xor((int)x, d, 64);
      x = y < x ? y : x;
    }
    acc = x < acc ? x : acc;
    if (pre_mask) return acc;
  }
  return acc;
}

/* The batch's first violation as far as workgroup `w` can know it (its own records and everything before them):
 * `bad` = this lane's record breaks the order (out of range, or a descent into it).  Every thread of the
 * workgroup must call it; contains barriers.  s_bad: one shared word. */
__device__ __forceinline__ uint32_t one_prefix_verdict(const OneCtl& C, int32_t w, int32_t i, bool bad, uint32_t* s_bad) {
  if (threadIdx.x == 0) *s_bad = ONE_NONE;
  const bool any_bad = __syncthreads_or(bad); /* also orders the store above */
  if (any_bad) {
    if (bad) atomicMin(s_bad, (uint32_t)i);
    __syncthreads();
  }
  if (threadIdx.x < 64) {
    const uint32_t local = *s_bad;
    uint32_t incl = local;
    if (w > 0) {
      if (threadIdx.x == 0)
        __hip_atomic_store(&C.ord[w], one_word(C.epoch, ONE_AGG, local), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool timed_out = false;
      const uint32_t before = one_lookback_min(C.ord, w, C.epoch, &timed_out);
      incl = before < incl ? before : incl;
      if (timed_out) { /* gave up: the batch counts as broken from this workgroup's first record on */
        const uint32_t mine = (uint32_t)w * (uint32_t)blockDim.x;
        incl = mine < incl ? mine : incl;
      }
    }
    if (threadIdx.x == 0) {
      __hip_atomic_store(&C.ord[w], one_word(C.epoch, ONE_PRE, incl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_bad = incl;
    }
  }
  __syncthreads();
  return *s_bad;
}

/* Arrival of workgroup `w` of `nwg`; true in thread 0 of the LAST workgroup to arrive, with *any_irregular = some
 * workgroup reported `irregular`.  The counters are back at zero when it returns true.  Thread 0 only. */
__device__ __forceinline__ bool one_arrive(const OneCtl& C, int32_t w, int32_t nwg, bool irregular, bool* any_irregular) {
  const int32_t grp = w >> 6;
  const uint32_t size1 = (uint32_t)min(64, nwg - (grp << 6));
  const uint32_t old1 = atomicAdd(&C.done1[grp], 1u + (irregular ? 0x10000u : 0u));
  if ((old1 & 0xffffu) + 1u != size1) return false;
  const bool irr1 = irregular || (old1 >> 16) != 0;
  __hip_atomic_store(&C.done1[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t ngrp = (uint32_t)((nwg + 63) >> 6);
  const uint32_t old0 = atomicAdd(C.done0, 1u + (irr1 ? 0x10000u : 0u));
  if ((old0 & 0xffffu) + 1u != ngrp) return false;
  *any_irregular = irr1 || (old0 >> 16) != 0;
  __hip_atomic_store(C.done0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

template <bool COMMIT>
__global__ __launch_bounds__(GPX_DBLOCK) GPX_AC_ATTR void k_ac_one(
    DevState S, DevScratch X, OneCtl C, int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot, const int32_t* __restrict__ median,
    const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, uint8_t* __restrict__ status, DirectStage D,
    int32_t* __restrict__ n_runs) {
  __shared__ uint32_t s_bad;
  const int32_t w = (int32_t)blockIdx.x;
  const int32_t i = w * GPX_DBLOCK + (int32_t)threadIdx.x;
  /* wave 1 of loads: the neighbours in gidx and this record's columns */
  int32_t g = 0, g_prev = 0, g_next = 0, f_a = 0, f_b = 0, f_c = 0, f_bnum = 0, f_bcoord = 0;
  bool bad = false, head = false, runstart = false;
  if (i < n) {
    g = gidx[i];
    g_prev = i > 0 ? gidx[i - 1] : ~g;
    g_next = i + 1 < n ? gidx[i + 1] : ~g;
    f_a = slot[i], f_b = median[i], f_c = flags ? (int32_t)flags[i] : 0;
    f_bnum = bnum[i], f_bcoord = bcoord[i];
    const bool oob = (uint32_t)g >= (uint32_t)S.G;
    bad = oob || (i > 0 && g_prev > g);
    runstart = g_prev != g; /* first record of a run of equal gidx: this lane answers for the whole run */
    head = runstart && !oob;
  }
  /* wave 2, requested BEFORE the verdict is waited for: the group's acceptor state and the ring entry of this
   * record's slot (a head whose run turns out to lie behind the first violation has loaded them in vain) */
  AccPre P = acc_nopre();
  if (head) acc_preload(S, g, f_a, P);
  const uint32_t first_bad = one_prefix_verdict(C, w, i, bad, &s_bad);
  bool irregular = false;
  if (runstart) {
    if ((uint32_t)i >= first_bad) {
      /* refused: the promise was broken at or before this run.  The first violation is always a run start (an
       * index out of range starts a run of its own, a descent starts a new group), so "runs that start at or
       * behind it" are exactly "records at or behind it" - and a workgroup that gave up waiting names its own
       * first record, which may lie inside a run: that run belongs to its head, refused or applied as a whole */
      int32_t j = i;
      for (;;) {
        if (!COMMIT) {
          r_bnum[j] = 0;
          r_bcoord[j] = 0;
          r_maxcp[j] = 0;
          r_flags[j] = 0;
        }
        status[j] = GPX_S_UNORDERED;
        if (++j >= n || (j == i + 1 ? g_next : gidx[j]) != g) break;
      }
      irregular = true; /* no regular count for a batch that broke its promise */
    } else {
      RunIter it;
      it.gidx = gidx;
      it.bnum = bnum;
      it.bcoord = bcoord;
      it.slot = slot;
      it.median = median;
      it.flags = flags;
      it.D = D;
      it.epoch = X.epoch;
      it.n = n;
      it.i = i;
      it.g = g;
      it.cur = i;
      it.chunk = -1;
      it.local = 0;
      it.count_chunks = false; /* nothing is counted here: k_one_count does it for the rare batch that needs it */
      it.st = status;          /* no prefill pass ran: the replaying lane marks a record OK before it judges it */
      it.mark_local = true;
      it.inplace = COMMIT;
      it.have_first = true;
      it.f_a = f_a;
      it.f_b = f_b;
      it.f_c = f_c;
      it.f_bnum = f_bnum;
      it.f_bcoord = f_bcoord;
      it.head = i;
      it.g_next = g_next;
      if (COMMIT)
        apply_commit_group(S, X, g, it, status, P);
      else
        apply_accept_group(S, X, g, it, r_bnum, r_bcoord, r_maxcp, r_flags, status, nullptr, P);
      irregular = it.irregular || it.pend >= 0; /* pend: the replay stopped on a commit without a run */
    }
  }
  const bool wg_irregular = __syncthreads_or(irregular);
  if (threadIdx.x == 0) {
    bool any = false;
    if (one_arrive(C, w, (int32_t)gridDim.x, wg_irregular, &any)) {
      if (any) {
        D.mark[0] = X.epoch; /* the compaction kernels have work */
        if (n_runs) *n_runs = -1;
      } else if (n_runs) {
        *n_runs = COMMIT ? n : 0; /* one run per commit, each parked at its record's index: dense as they stand */
      }
    }
  }
}

/* irregular batches only (D.mark raised): parked runs per 1024-record chunk, for k_emit_runs_direct */
__global__ __launch_bounds__(GPX_DCHUNK) void k_one_count(DevScratch X, int32_t n, DirectStage D) {
  if (*D.mark != X.epoch) return;
  const int32_t i = (int32_t)blockIdx.x * GPX_DCHUNK + (int32_t)threadIdx.x;
  const int32_t c = __syncthreads_count(i < n && D.tag[i] == X.epoch);
  if (threadIdx.x == 0) D.chunk_cnt[blockIdx.x] = c;
}

/* PROPOSE: strictly ascending gidx (every group at most once), so every record is its group's only one */
template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_propose_one(
    DevState S, DevScratch X, OneCtl C, int32_t n, const int32_t* __restrict__ gidx,
    const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status,
    const int64_t* __restrict__ handle) {
  __shared__ uint32_t s_bad;
  const int32_t w = (int32_t)blockIdx.x;
  const int32_t i = w * GPX_BLOCK + (int32_t)threadIdx.x;
  int32_t g = 0;
  bool bad = false, live = false;
  if (i < n) {
    g = gidx[i];
    const int32_t g_prev = i > 0 ? gidx[i - 1] : -1;
    const bool oob = (uint32_t)g >= (uint32_t)S.G;
    bad = oob || (i > 0 && g_prev >= g);
    live = !oob;
  }
  ProposePre<KMAX> P;
  if (live) {
    propose_preload<KMAX>(S, g, P);
    propose_preload_ring<KMAX>(S, g, P);
  }
  const uint32_t first_bad = one_prefix_verdict(C, w, i, bad, &s_bad);
  if (i >= n) return;
  if ((uint32_t)i >= first_bad) {
    o_slot[i] = 0;
    o_bnum[i] = 0;
    o_bcoord[i] = 0;
    o_median[i] = 0;
    status[i] = GPX_S_UNORDERED;
    return;
  }
  OneRec it;
  it.idx = i;
  it.a = is_stop ? (int32_t)(is_stop[i] & 1) : 0;
  it.c = 1;
  it.done = 0;
  status[i] = GPX_S_OK; /* no prefill pass ran; apply_propose_group overwrites it for a record it refuses */
  apply_propose_group<KMAX, OneRec>(S, X, g, it, o_slot, o_bnum, o_bcoord, o_median, status, P, handle);
}
