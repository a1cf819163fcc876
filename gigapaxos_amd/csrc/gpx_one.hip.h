/*
 * gpx_one.hip.h — ordered PROPOSE / ACCEPT / COMMIT batches: a check kernel and ONE work kernel (round 4).
 *
 * Under the gpx_engine_set_ordered_batches promise a batch of more than 65,536 records used to cost three or
 * four dependent launches: k_order_check (the verdict), the direct kernel (the work), k_emit_runs_direct and
 * k_copy_runs (which find nothing to do for a usual batch).  On this chip a dependent launch costs 5-6 us whatever
 * it does - the previous kernel's dirty lines leave the XCDs' L2s first - and they were 113 of the full round's
 * 279 us by round 3's per-kernel profile (profiles/r03_bench_full_round.json).  Round 4's form is TWO launches:
 *
 *   order      k_one_check reads the gidx column (eight records per lane) and leaves the batch's FIRST VIOLATION -
 *              the first index that is out of range or lower than its predecessor (PROPOSE: not higher) - in one
 *              word tagged with the launch's epoch; a batch that keeps its promise costs it no atomic.  It also
 *              writes the REGULAR batch's output count.  The work kernel, launched behind it, reads that word and
 *              applies a run of equal gidx iff it starts before the first violation (which is always a run start:
 *              an index out of range starts a run of its own, a descent starts a new group): the records before
 *              the first violation are applied, the records from it on are refused (GPX_S_UNORDERED, outputs
 *              zero, no state change) - include/gpx.h.  A second run of a group can only start behind a descent,
 *              so no two lanes ever own the same group.  No status prefill pass: the lane that replays a record
 *              marks it.
 *   outputs    execution runs are parked at their records' indices and tagged, as in gpx_direct.hip.h.  The
 *              regular count - ACCEPTs release no commit: 0; every COMMIT executes exactly one run: n, dense as
 *              parked - is in place already; a workgroup that sees otherwise overwrites it with -1 and raises
 *              D.mark.  The usual batch needs no compaction and no other kernel; for the unusual one the
 *              compaction kernels (k_one_count, k_emit_runs_direct, k_copy_runs) follow at once (the default) or
 *              when the caller asks for dense columns (GPX_LAZY_OUTPUTS, gpx_compact_last_dev).
 *
 * A dependent launch that finds nothing to do costs ~2 us on this chip (not the 5-6 us the event-bracketed kernel
 * profile of round 3 suggested: bench_full_round.py with and without the idle compaction launches, profiles/
 * r04_full_round_*.json).  Fusing the VERDICT into the work kernel was built twice and measured slower than the
 * kernel boundary it replaces: (1) a decoupled look-back over per-workgroup words (MIN of first violations): 2,048
 * workgroups start together and the "prefix known" front moves 64 workgroups per atomic round trip - k_propose_one
 * 39 us against 18 + 6; (2) checker workgroups at the head of the grid publishing one verdict word the others wait
 * for: three memory-side round trips (checker word, collection, verdict) before the first store - an ACCEPT call 32.0
 * us against 29.1 for the three launches of round 3.  Also measured on the way: "last workgroup to finish" by arrival
 * counters - device-scope atomics on one cache line are serial at ~16 ns each whatever the address, and the
 * workgroups alive at one time share two or three lines of counters: 3,907 arrivals = 62 us per launch.
 * (profiles/r04_full_round_lookback_attempt.json, r04_full_round_checker_workgroups.json)
 */
#pragma once
#include "gpx_direct.hip.h"

#define ONE_NONE 0xffffffffu

struct OneCtl {
  unsigned long long* verdict; /* [1] epoch << 32 | (ONE_NONE - first violating index); another epoch: no violation */
  uint32_t epoch;              /* ascending, never 0: a word of an older launch never needs clearing */
};

/* the batch's first violation (ONE_NONE: none), as k_one_check left it */
__device__ __forceinline__ uint32_t one_first_bad(const OneCtl& C) {
  const unsigned long long v = *C.verdict;
  return (uint32_t)(v >> 32) == C.epoch ? ONE_NONE - (uint32_t)v : ONE_NONE;
}

/* The verdict: the first index that is out of range or lower (STRICT: not higher) than its predecessor.  Eight
 * records per lane (16-byte loads); a workgroup that finds a violation raises the word with one atomicMax -
 * (epoch, ONE_NONE - index): a newer launch beats an older word, a lower index a higher one - so a batch that keeps
 * its promise costs no atomic at all.  Thread 0 of the grid also writes the REGULAR batch's output count; the work
 * kernel overwrites it with -1 when the batch turns out otherwise (launched behind: the order is the stream's). */
template <bool STRICT>
__global__ __launch_bounds__(GPX_DBLOCK) void k_one_check(int32_t n, const int32_t* __restrict__ gidx, int32_t G, OneCtl C,
                                                         int32_t* __restrict__ count_out, int32_t regular_count) {
  __shared__ uint32_t s_bad;
  const int64_t i0 = ((int64_t)blockIdx.x * GPX_DBLOCK + threadIdx.x) * 8;
  if (blockIdx.x == 0 && threadIdx.x == 0 && count_out) *count_out = regular_count;
  if (threadIdx.x == 0) s_bad = ONE_NONE;
  uint32_t mine = ONE_NONE;
  if (i0 < n) {
    int32_t g[9];
    g[0] = i0 > 0 ? gidx[i0 - 1] : INT32_MIN;
    if (i0 + 7 < n && !((uintptr_t)gidx & 15)) {
      const I4 a = *(const I4*)(gidx + i0), b = *(const I4*)(gidx + i0 + 4);
      g[1] = a.x, g[2] = a.y, g[3] = a.z, g[4] = a.w, g[5] = b.x, g[6] = b.y, g[7] = b.z, g[8] = b.w;
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++) g[q + 1] = i0 + q < n ? gidx[i0 + q] : INT32_MAX;
    }
#pragma unroll
    for (int q = 7; q >= 0; q--)
      if (i0 + q < n && ((uint32_t)g[q + 1] >= (uint32_t)G || (i0 + q > 0 && (STRICT ? g[q] >= g[q + 1] : g[q] > g[q + 1]))))
        mine = (uint32_t)(i0 + q);
  }
  if (__syncthreads_or(mine != ONE_NONE)) { /* (also orders the store of s_bad above) */
    if (mine != ONE_NONE) atomicMin(&s_bad, mine);
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(C.verdict, ((unsigned long long)C.epoch << 32) | (unsigned long long)(ONE_NONE - s_bad));
  }
}

/* XCHG (round 4, batches of at most 65,536 records = at most 256 workgroups, all resident): no k_one_check launch -
 * every workgroup judges its OWN 256 records (it has loaded them and their predecessors anyway), publishes its first
 * violation in an epoch-tagged ticket (C.verdict + GPX_ONE_TICKETS + blockIdx), and reads all tickets: one exchange
 * between resident workgroups, with the group state already requested.  Workgroup 0 writes the regular count before
 * its ticket, so a workgroup that overwrites it with -1 does so after it. */
#define GPX_ONE_TICKETS 16
#define GPX_ONE_XCHG_MAX_N 65536
/* the exchange: this lane's record i violates the order (or not) -> the batch's first violation.  Every thread of every
 * workgroup of the grid must call it (barriers; at most GPX_DBLOCK workgroups).  count_out / regular_count: workgroup 0
 * writes the regular batch's count before its ticket goes out. */
__device__ __forceinline__ uint32_t one_exchange(const DevScratch& X, const OneCtl& C, bool viol, int32_t i,
                                                 int32_t* __restrict__ count_out, int32_t regular_count) {
  __shared__ uint32_t s_mine, s_all, s_gave_up;
  if (threadIdx.x == 0) s_mine = 0, s_all = 0, s_gave_up = 0;
  __syncthreads();
  /* the verdict's encoding: ONE_NONE - index, 0 = none; the batch's first violation is the MAX over everybody */
  if (viol) atomicMax(&s_mine, ONE_NONE - (uint32_t)i);
  __syncthreads();
  unsigned long long* const tick = C.verdict + GPX_ONE_TICKETS;
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0 && count_out) *count_out = regular_count;
    __hip_atomic_store(&tick[blockIdx.x], ((unsigned long long)C.epoch << 32) | s_mine, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x < gridDim.x) {
    unsigned long long v;
    XchgWait w;
    for (;;) {
      v = __hip_atomic_load(&tick[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(v >> 32) == C.epoch) break;
      if (w.tired()) { /* that workgroup never became resident (gpx_kernels.hip.h: XchgWait) */
        s_gave_up = 1;
        v = 0;
        break;
      }
    }
    if ((uint32_t)v) atomicMax(&s_all, (uint32_t)v);
  }
  __syncthreads();
  if (s_gave_up) { /* nothing of this workgroup's records is applied; the host refuses every later call */
    if (threadIdx.x == 0) xchg_abort(X);
    return 0u;
  }
  return ONE_NONE - s_all;
}
template <bool COMMIT, bool XCHG = false>
__global__ __launch_bounds__(GPX_DBLOCK) GPX_AC_ATTR void k_ac_one(
    DevState S, DevScratch X, OneCtl C, int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot, const int32_t* __restrict__ median,
    const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, uint8_t* __restrict__ status, DirectStage D,
    int32_t* __restrict__ n_runs, int32_t regular_count = 0) {
  const int32_t i = (int32_t)blockIdx.x * GPX_DBLOCK + (int32_t)threadIdx.x;
  uint32_t first_bad = XCHG ? ONE_NONE : one_first_bad(C);
  /* wave 1 of loads: the neighbours in gidx and this record's columns */
  int32_t g = 0, g_prev = 0, g_next = 0, f_a = 0, f_b = 0, f_c = 0, f_bnum = 0, f_bcoord = 0;
  bool head = false, runstart = false, oob = false;
  if (i < n) {
    g = gidx[i];
    g_prev = i > 0 ? gidx[i - 1] : ~g;
    g_next = i + 1 < n ? gidx[i + 1] : ~g;
    f_a = slot[i], f_b = median[i], f_c = flags ? (int32_t)flags[i] : 0;
    f_bnum = bnum[i], f_bcoord = bcoord[i];
    oob = (uint32_t)g >= (uint32_t)S.G;
    runstart = g_prev != g; /* first record of a run of equal gidx: this lane answers for the whole run */
    head = runstart && !oob;
  }
  /* wave 2: the group's acceptor state and the ring entry of this record's slot */
  AccPre P = acc_nopre();
  if (head) acc_preload(S, g, f_a, P); /* (not made to wait for the verdict word: a refused head has loaded in vain) */
  if (XCHG) first_bad = one_exchange(X, C, i < n && (oob || (i > 0 && g_prev > g)), i, n_runs, regular_count);
  bool irregular = false;
  if (runstart) {
    if ((uint32_t)i >= first_bad) {
      /* refused: the promise was broken at or before this run.  The first violation is always a run start (an
       * index out of range starts a run of its own, a descent starts a new group), so "runs that start at or
       * behind it" are exactly "records at or behind it" */
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
  /* a usual batch is finished: k_one_check wrote its count.  Any workgroup that saw otherwise says so */
  if (__syncthreads_or(irregular) && threadIdx.x == 0) {
    __hip_atomic_store(D.mark, X.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* the compaction kernels have work */
    if (n_runs) __hip_atomic_store(n_runs, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
template <int KMAX, bool XCHG = false>
__global__ __launch_bounds__(GPX_BLOCK) void k_propose_one(
    DevState S, DevScratch X, OneCtl C, int32_t n, const int32_t* __restrict__ gidx,
    const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status,
    const int64_t* __restrict__ handle) {
  const int32_t i = (int32_t)blockIdx.x * GPX_BLOCK + (int32_t)threadIdx.x;
  if (!XCHG && i >= n) return;
  uint32_t first_bad = XCHG ? ONE_NONE : one_first_bad(C);
  const int32_t g = i < n ? gidx[i] : -1;
  const int32_t g_prev = (XCHG && i > 0 && i < n) ? gidx[i - 1] : INT32_MIN;
  ProposePre<KMAX> P;
  if ((uint32_t)g < (uint32_t)S.G) { /* requested without waiting for the verdict */
    propose_preload<KMAX>(S, g, P);
    propose_preload_ring<KMAX>(S, g, P);
  }
  if (XCHG) { /* at most 65,536 requests: no k_one_check launch (one_exchange); strictly ascending, in range */
    first_bad = one_exchange(X, C, i < n && ((uint32_t)g >= (uint32_t)S.G || (i > 0 && g_prev >= g)), i, nullptr, 0);
    if (i >= n) return;
  }
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
