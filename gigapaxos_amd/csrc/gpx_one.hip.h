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

#define GPX_ONE_TICKETS 16 /* words of one_words in front of the arrival counters (the verdict word is the first) */
/* one lane's share of an ordered ACCEPT / COMMIT batch once the verdict is known: the lane of a run's first record
 * replays the whole run (or refuses it: the promise was broken at or before it).  Returns "the cheap answer does not
 * hold" (a run parked by an ACCEPT, a commit without a run, a refused run). */
template <bool COMMIT>
__device__ __forceinline__ bool ac_one_apply(const DevState& S, const DevScratch& X, int32_t n, int32_t i, uint32_t first_bad,
                                             bool runstart, int32_t g, int32_t g_next, int32_t f_a, int32_t f_b, int32_t f_c,
                                             int32_t f_bnum, int32_t f_bcoord, const AccPre& P,
                                             const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
                                             const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot,
                                             const int32_t* __restrict__ median, const uint8_t* __restrict__ flags,
                                             int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
                                             int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags,
                                             uint8_t* __restrict__ status, const DirectStage& D) {
  if (!runstart) return false;
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
    return true; /* no regular count for a batch that broke its promise */
  }
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
  return it.irregular || it.pend >= 0; /* pend: the replay stopped on a commit without a run */
}

template <bool COMMIT>
__global__ __launch_bounds__(GPX_DBLOCK) GPX_AC_ATTR void k_ac_one(
    DevState S, DevScratch X, OneCtl C, int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot, const int32_t* __restrict__ median,
    const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, uint8_t* __restrict__ status, DirectStage D,
    int32_t* __restrict__ n_runs) {
  const int32_t i = (int32_t)blockIdx.x * GPX_DBLOCK + (int32_t)threadIdx.x;
  const uint32_t first_bad = one_first_bad(C);
  /* wave 1 of loads: the neighbours in gidx and this record's columns */
  int32_t g = 0, g_prev = 0, g_next = 0, f_a = 0, f_b = 0, f_c = 0, f_bnum = 0, f_bcoord = 0;
  bool head = false, runstart = false;
  if (i < n) {
    g = gidx[i];
    g_prev = i > 0 ? gidx[i - 1] : ~g;
    g_next = i + 1 < n ? gidx[i + 1] : ~g;
    f_a = slot[i], f_b = median[i], f_c = flags ? (int32_t)flags[i] : 0;
    f_bnum = bnum[i], f_bcoord = bcoord[i];
    runstart = g_prev != g; /* first record of a run of equal gidx: this lane answers for the whole run */
    head = runstart && (uint32_t)g < (uint32_t)S.G;
  }
  /* wave 2: the group's acceptor state and the ring entry of this record's slot */
  AccPre P = acc_nopre();
  if (head) acc_preload(S, g, f_a, P); /* (not made to wait for the verdict word: a refused head has loaded in vain) */
  const bool irregular = ac_one_apply<COMMIT>(S, X, n, i, first_bad, runstart, g, g_next, f_a, f_b, f_c, f_bnum, f_bcoord, P,
                                              gidx, bnum, bcoord, slot, median, flags, r_bnum, r_bcoord, r_maxcp, r_flags,
                                              status, D);
  /* a usual batch is finished: k_one_check wrote its count.  Any workgroup that saw otherwise says so */
  if (__syncthreads_or(irregular) && threadIdx.x == 0) {
    __hip_atomic_store(D.mark, X.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* the compaction kernels have work */
    if (n_runs) __hip_atomic_store(n_runs, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

/* ---- ONE launch up to 2 workgroups per CU (round 5) ---------------------------------------------------------------
 * Round 4's one-launch form (every workgroup judges its own 256 records and all of them exchange the verdict through
 * 256 tickets) NEEDED its workgroups resident together and nothing enforced it.  Now the host launches this form only
 * with a grid that is resident whatever the kernel - at most 2 workgroups per CU, divided by the streams the device's
 * engines launch on and the processes sharing it (gpx_engine.hip: xchg_ctl): 512 workgroups = 131,072 records for one
 * engine on an MI355X - and the exchange is sixteen arrival counters 128 bytes apart instead of a ticket per
 * workgroup (device-scope atomics on one line are serial at ~16 ns each: 32 arrivals per line), polled by sixteen
 * lanes.  The counters are never reset: the grid is padded to a multiple of sixteen, every line gets the same number of
 * arrivals per launch, the host passes the value they reach.  Everything that crosses workgroups here is a device-
 * scope atomic (verdict word, arrivals, the count word); what must be performed before the arrival is a RETURNING
 * atomic whose result the arriving lane holds (no fence: a fence would also wait for the state preloads in flight),
 * the polls are relaxed and ONE acquire fence follows the last of them (an acquire per iteration invalidates the
 * caches every time round).
 * Larger batches keep the check kernel + the work kernel.  Two one-launch forms for them were built this round and
 * measured (docs/HISTORY.md 3 ii-c): resident workgroups LOOPING over chunks - the grid hipOccupancyMaxActiveBlocksPer
 * Multiprocessor promises does not all become resident: the waiters gave up after two seconds (profiles/
 * r05_pers_loop_gave_up.txt); one workgroup per chunk with the first 512 TO START (a ticket each, sixteen counters) as
 * judges and everybody polling the arrivals - correct at every size (all parity tests), no assumption about residency,
 * and 5-7 x slower: 166 us against 7 + 16 for a 1 M-record proposal call, 138 against 6 + 28 for an ACCEPT call
 * (profiles/r05_ticket_roles_one_launch.txt) - thousands of workgroups on sixteen lines. */
#define GPX_GX_LINES 16
struct GridXchg {
  uint32_t* arrive;            /* [GPX_GX_LINES * 32] arrival counters, one per 128-byte line, cumulative */
  unsigned long long* verdict; /* OneCtl's word: epoch << 32 | (ONE_NONE - first violating index) */
  uint32_t epoch;
  uint32_t arrive_target;      /* every arrival counter's value once all workgroups of THIS launch have arrived */
  uint32_t timeout_ms;         /* a poller gives up after this long (XchgWait) */
};
/* First thing in the kernel: workgroup 0 writes the REGULAR batch's count (a device-scope atomic store).  A workgroup
 * that finds the batch irregular overwrites it with -1 AFTER the exchange, i.e. after it has seen workgroup 0's arrival,
 * which leaves this wave microseconds behind this store, to the same address.  Nothing waits for the store: a fence
 * (or a returning atomic whose result is held until the arrival - built, and the compiler turns that into vmcnt(0)
 * too) would make this wave wait for its state preloads before it arrives, and everybody waits for the slowest
 * arrival. */
__device__ __forceinline__ void grid_begin(int32_t* __restrict__ count_out, int32_t regular_count) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && count_out)
    __hip_atomic_store(count_out, regular_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* Every thread of every workgroup of the grid calls it once.  wg_bad: this workgroup's first violation,
 * already reduced (ONE_NONE: none; thread 0's value counts).  Returns the batch's first violation once every workgroup
 * has arrived. */
__device__ __forceinline__ uint32_t grid_exchange(const DevScratch& X, const GridXchg& Q, uint32_t wg_bad) {
  __shared__ uint32_t s_first, s_gave_up;
  if (threadIdx.x == 0) {
    s_gave_up = 0;
    if (wg_bad != ONE_NONE) { /* rare: the verdict must have been performed before the arrival can be seen */
      const unsigned long long old =
          atomicMax(Q.verdict, ((unsigned long long)Q.epoch << 32) | (unsigned long long)(ONE_NONE - wg_bad));
      asm volatile("" ::"v"((uint32_t)old));
    }
    __hip_atomic_fetch_add(&Q.arrive[(blockIdx.x % GPX_GX_LINES) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads(); /* (s_gave_up cleared before a poller may raise it) */
  if (threadIdx.x < GPX_GX_LINES) {
    XchgWait w;
    while ((int32_t)(__hip_atomic_load(&Q.arrive[threadIdx.x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                     Q.arrive_target) < 0) {
      if (w.tired(Q.timeout_ms)) { /* a workgroup of the grid never became resident (gpx_kernels.hip.h: XchgWait) */
        s_gave_up = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* ONE cache invalidate per workgroup, after the last poll */
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_gave_up) {
      /* nobody who reads the verdict from now on applies anything: the workgroups that were not resident start when
       * this one has left, find every arrival in and would otherwise go ahead with their part of the batch */
      const unsigned long long old = atomicMax(Q.verdict, ((unsigned long long)Q.epoch << 32) | (unsigned long long)ONE_NONE);
      asm volatile("" ::"v"((uint32_t)old));
      xchg_abort(X);
    }
    const unsigned long long v = __hip_atomic_load(Q.verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_first = (uint32_t)(v >> 32) == Q.epoch ? ONE_NONE - (uint32_t)v : ONE_NONE;
    if (s_gave_up) s_first = 0u; /* nothing of this workgroup's records is applied */
  }
  __syncthreads();
  return s_first;
}
/* a workgroup's first violation from its lanes' (ONE_NONE: none); every thread must call it */
__device__ __forceinline__ uint32_t wg_first_bad(uint32_t mine) {
  __shared__ uint32_t s_bad;
  if (threadIdx.x == 0) s_bad = ONE_NONE;
  if (__syncthreads_or(mine != ONE_NONE)) { /* (also orders the store of s_bad above) */
    if (mine != ONE_NONE) atomicMin(&s_bad, mine);
    __syncthreads();
  }
  return s_bad;
}
template <bool COMMIT>
__global__ __launch_bounds__(GPX_DBLOCK) GPX_AC_ATTR void k_ac_pers(
    DevState S, DevScratch X, GridXchg Q, int32_t n, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ bnum, const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot,
    const int32_t* __restrict__ median, const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum,
    int32_t* __restrict__ r_bcoord, int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags,
    uint8_t* __restrict__ status, DirectStage D, int32_t* __restrict__ n_runs, int32_t regular_count) {
  grid_begin(n_runs, regular_count);
  /* this chunk's loads go out before anything else (waves 1 and 2 of k_ac_one) */
  const int32_t i = (int32_t)blockIdx.x * GPX_DBLOCK + (int32_t)threadIdx.x;
  int32_t g = 0, g_prev = 0, g_next = 0, f_a = 0, f_b = 0, f_c = 0, f_bnum = 0, f_bcoord = 0;
  bool runstart = false;
  uint32_t mine = ONE_NONE;
  AccPre P = acc_nopre();
  if (i < n) {
    g = gidx[i];
    g_prev = i > 0 ? gidx[i - 1] : ~g;
    g_next = i + 1 < n ? gidx[i + 1] : ~g;
    f_a = slot[i], f_b = median[i], f_c = flags ? (int32_t)flags[i] : 0;
    f_bnum = bnum[i], f_bcoord = bcoord[i];
    const bool oob = (uint32_t)g >= (uint32_t)S.G;
    runstart = g_prev != g;
    if (runstart && !oob) acc_preload(S, g, f_a, P);
    if (oob || (i > 0 && g_prev > g)) mine = (uint32_t)i;
  }
  const uint32_t first_bad = grid_exchange(X, Q, wg_first_bad(mine));
  const bool irregular = ac_one_apply<COMMIT>(S, X, n, i, first_bad, runstart, g, g_next, f_a, f_b, f_c, f_bnum, f_bcoord, P,
                                              gidx, bnum, bcoord, slot, median, flags, r_bnum, r_bcoord, r_maxcp, r_flags,
                                              status, D);
  /* a usual batch is finished: workgroup 0 wrote its count.  Any workgroup that saw otherwise says so */
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
template <int KMAX>
__device__ __forceinline__ void propose_one_apply(const DevState& S, const DevScratch& X, int32_t i, int32_t g, uint32_t first_bad,
                                                  const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot,
                                                  int32_t* __restrict__ o_bnum, int32_t* __restrict__ o_bcoord,
                                                  int32_t* __restrict__ o_median, uint8_t* __restrict__ status,
                                                  ProposePre<KMAX>& P, const int64_t* __restrict__ handle) {
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
template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_propose_one(
    DevState S, DevScratch X, OneCtl C, int32_t n, const int32_t* __restrict__ gidx,
    const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status,
    const int64_t* __restrict__ handle) {
  const int32_t i = (int32_t)blockIdx.x * GPX_BLOCK + (int32_t)threadIdx.x;
  if (i >= n) return;
  const uint32_t first_bad = one_first_bad(C);
  const int32_t g = gidx[i];
  ProposePre<KMAX> P;
  if ((uint32_t)g < (uint32_t)S.G) { /* requested without waiting for the verdict */
    propose_preload<KMAX>(S, g, P);
    propose_preload_ring<KMAX>(S, g, P);
  }
  propose_one_apply<KMAX>(S, X, i, g, first_bad, is_stop, o_slot, o_bnum, o_bcoord, o_median, status, P, handle);
}
/* ... in ONE launch: a grid that is resident for sure, the verdict exchanged at grid_exchange's counters (k_ac_pers) */
template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_propose_pers(
    DevState S, DevScratch X, GridXchg Q, int32_t n, const int32_t* __restrict__ gidx,
    const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status,
    const int64_t* __restrict__ handle) {
  const int32_t i = (int32_t)blockIdx.x * GPX_BLOCK + (int32_t)threadIdx.x;
  int32_t g = -1;
  uint32_t mine = ONE_NONE;
  ProposePre<KMAX> P;
  if (i < n) {
    g = gidx[i];
    const int32_t g_prev = i > 0 ? gidx[i - 1] : INT32_MIN;
    if ((uint32_t)g < (uint32_t)S.G) { /* requested without waiting for the verdict */
      propose_preload<KMAX>(S, g, P);
      propose_preload_ring<KMAX>(S, g, P);
    }
    if ((uint32_t)g >= (uint32_t)S.G || (i > 0 && g_prev >= g)) mine = (uint32_t)i; /* strictly ascending, in range */
  }
  const uint32_t first_bad = grid_exchange(X, Q, wg_first_bad(mine));
  if (i < n) propose_one_apply<KMAX>(S, X, i, g, first_bad, is_stop, o_slot, o_bnum, o_bcoord, o_median, status, P, handle);
}
