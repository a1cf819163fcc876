/*
 * gpx_direct.hip.h — ACCEPT and COMMIT batches that arrive grouped by group (round 2).
 *
 * Inside the pipeline a batch is usually the previous stage's output: ACCEPTs follow the proposal
 * batch (gidx ascending), commits are the decisions, which leave the accept-reply call grouped by
 * gidx ascending (include/gpx.h ORDER).  k_order_check<false> recognises such a batch on the device
 * (gidx in range and non-decreasing) and these kernels then apply it WITHOUT the bucket partition:
 * one lane per record; the lane of the first record of a group's run replays the whole run in array
 * order through the same per-group functions as the per-bucket kernels (apply_accept_group /
 * apply_commit_group: PISM.handleAccept PISM:1080-1166, handleBatchedCommit / handleCommittedRequest
 * PISM:1432-1528 -> extractExecuteAndCheckpoint PISM:1619-1701).  Consecutive lanes own ascending
 * groups, so the state accesses of a dense batch are coalesced.  Anything else falls back to the
 * partition path, which is launched behind these kernels and returns at once for an ordered batch
 * (device-side choice on *X.unsorted, like the proposal path).
 *
 * Compacted outputs (execution runs): a record produces at most one run; it is parked at the
 * record's own index (tag[i] == the call's epoch: record i holds one) and counted per 1024-record
 * chunk; k_emit_runs_direct compacts chunk by chunk - record order is already the output order (gidx
 * ascending, a group's entries in array order).
 *
 * Round 3: the kernel is bound by dependent round trips, not by bytes (profiles/
 * r03_pmc_full_round_ordered_before.txt: 92 MB of traffic for 81 MB algorithmic, SQ_WAIT_ANY 77 % of the
 * wave cycles), so its loads go out in three waves instead of one after the other: the lane's neighbours
 * in gidx and its own record columns; then the group's acceptor state together with the ring entry of the
 * record's slot (one 16-byte entry = accepted pvalue + both rings' flags, AccView); then whatever the
 * garbage collection looks at.  No per-record clearing store (an epoch tag says which records hold a
 * run), 256-thread workgroups (finer tail), and a commit that is executable at once is never written
 * to the committed ring and read back (acc_eec).
 */
#pragma once
#include "gpx_kernels.hip.h"

#define GPX_DCHUNK_SHIFT 10
#define GPX_DCHUNK (1 << GPX_DCHUNK_SHIFT)

#define GPX_DBLOCK 256 /* threads of a k_ac_direct workgroup: GPX_DCHUNK / GPX_DBLOCK of them per chunk */

struct DirectStage {
  int32_t *st_first, *st_count; /* [n] run parked at the record that produced it */
  uint32_t* tag;                /* [n] == the call's epoch: record i holds a parked run */
  int32_t* chunk_cnt;           /* [ceil(n / 1024)] runs per chunk; zeroed before the call */
  /* COMMIT batches park their runs in the CALLER'S columns (n entries each: gpx.h) at the record's own index:
   * when every record executed something - the steady state: one commit, one slot executed - the columns are
   * final as they stand, run i belongs to record i, and nothing is left to compact (k_emit_runs_direct<true>
   * sees that the chunk counts add up to n).  Otherwise the parked runs go through st_gidx / st_first /
   * st_count (dense staging) and k_copy_runs brings them back. */
  int32_t *x_gidx, *x_first, *x_count;
  int32_t* st_gidx;
  int32_t* st_total; /* [1] runs staged by k_emit_runs_direct<true> */
  /* [1] == the call's epoch: the cheap answer does not hold.  ACCEPT batches: some record parked a run (the
   * usual batch releases no commit: nothing to emit).  COMMIT batches: some record parked NO run (the usual
   * batch executes one slot per commit: the caller's columns are final as parked).  Plain stores of one
   * value, no atomics; the emit kernels of a usual batch read this word and return. */
  uint32_t* mark;
};

/* the run of records of one group in a gidx-ordered batch, with GroupIter's interface */
struct RunIter {
  const int32_t *gidx, *bnum, *bcoord, *slot, *median;
  const uint8_t* flags;
  DirectStage D;
  uint32_t epoch;
  int32_t n, i, g, cur, chunk, local;
  bool count_chunks = true; /* k_ac_direct counts runs per chunk; the single-launch kernel reads the parking */
  uint8_t* st = nullptr;    /* single-launch kernel: no status prefill pass ran - the lane that replays a
                             * record marks it OK before it judges it (a prefill by the record's own lane
                             * would race with the head's verdict, across waves and across workgroups) */
  /* the head's own record, fetched by the kernel ahead of the group state; g_next = gidx[head + 1]
   * (or ~g at the end of the batch): a run of one record never touches memory here */
  bool inplace = false; /* park runs in the caller's columns (k_ac_direct<COMMIT>) */
  /* k_ac_one: "the cheap answer does not hold" is collected per workgroup and travels with the arrival counters
   * instead of being stored to D.mark by every lane that notices */
  bool mark_local = false, irregular = false;
  int32_t pend = -1;    /* inplace: the record handed out last has not parked a run yet */
  bool have_first = false;
  int32_t f_a = 0, f_b = 0, f_c = 0, f_bnum = 0, f_bcoord = 0;
  int32_t head = -2, g_next = 0;
  __device__ __forceinline__ bool next(Rec& out) {
    if (inplace) {
      if (pend >= 0) { /* a commit that executed nothing: the columns have a hole */
        if (mark_local)
          irregular = true;
        else
          D.mark[0] = epoch;
      }
      pend = i;
    }
    if (have_first) {
      have_first = false;
      out.idx = i;
      out.a = f_a;
      out.b = f_b;
      out.c = f_c;
      out.bnum = f_bnum;
      out.bcoord = f_bcoord;
      cur = i;
      if (st) st[i] = GPX_S_OK;
      i++;
      return true;
    }
    if (i >= n || (i == head + 1 ? g_next : gidx[i]) != g) {
      pend = -1; /* no record handed out */
      return false;
    }
    cur = i;
    out.idx = i;
    out.a = slot[i];
    out.b = median[i];
    out.c = flags ? (int32_t)flags[i] : 0;
    out.bnum = bnum[i];
    out.bcoord = bcoord[i];
    if (st) st[i] = GPX_S_OK;
    i++;
    return true;
  }
  __device__ __forceinline__ void emit(int32_t, int32_t first_slot, int32_t count, int32_t, int32_t) {
    if (inplace) {
      D.x_gidx[cur] = g;
      D.x_first[cur] = first_slot;
      D.x_count[cur] = count;
      pend = -1;
    } else {
      D.st_first[cur] = first_slot;
      D.st_count[cur] = count; /* > 0 */
      if (mark_local)
        irregular = true;
      else
        D.mark[0] = epoch;
    }
    D.tag[cur] = epoch;
    if (!count_chunks) return;
    if ((cur >> GPX_DCHUNK_SHIFT) == chunk)
      local++;
    else
      atomicAdd(&D.chunk_cnt[cur >> GPX_DCHUNK_SHIFT], 1); /* a run reaching into the next chunk: rare */
  }
};

#ifdef GPX_AC_WAVES /* occupancy sweep builds (scripts/ac_occupancy_sweep.sh): never shipped */
#define GPX_AC_ATTR __attribute__((amdgpu_waves_per_eu(GPX_AC_WAVES)))
#else
#define GPX_AC_ATTR
#endif
template <bool COMMIT>
__global__ __launch_bounds__(GPX_DBLOCK) GPX_AC_ATTR void k_ac_direct(
    DevState S, DevScratch X, int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot, const int32_t* __restrict__ median,
    const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, uint8_t* __restrict__ status,
    DirectStage D, int32_t refuse) {
  const int32_t i = (int32_t)blockIdx.x * GPX_DBLOCK + (int32_t)threadIdx.x;
  if (*X.unsorted == X.epoch) {
    /* not ordered: the partition path does it - or, under the caller's GPX_ORDERED_* promise (no
     * partition path launched), the batch is refused whole */
    if (refuse && i < n) {
      if (!COMMIT) {
        r_bnum[i] = 0;
        r_bcoord[i] = 0;
        r_maxcp[i] = 0;
        r_flags[i] = 0;
      }
      status[i] = GPX_S_UNORDERED;
    }
    return;
  }
  int32_t local = 0;
  if (i < n) {
    /* wave 1 of loads: the neighbours in gidx and this record's columns (independent of each other) */
    const int32_t g = gidx[i];
    const int32_t g_prev = i > 0 ? gidx[i - 1] : ~g;
    const int32_t g_next = i + 1 < n ? gidx[i + 1] : ~g;
    const int32_t f_a = slot[i], f_b = median[i], f_c = flags ? (int32_t)flags[i] : 0;
    const int32_t f_bnum = bnum[i], f_bcoord = bcoord[i];
    if (g_prev != g) { /* head of its group's run */
      /* wave 2: the group's acceptor state and the ring entry of this record's slot */
      AccPre P;
      acc_preload(S, g, f_a, P);
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
      it.chunk = i >> GPX_DCHUNK_SHIFT;
      it.local = 0;
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
      if (it.pend >= 0) D.mark[0] = X.epoch; /* the replay stopped on a record without a run */
      local = it.local;
    }
  }
  /* this workgroup's runs inside its own chunk: one atomic per workgroup */
  __shared__ int32_t wsum[GPX_DBLOCK / 64];
  int32_t x = local;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t tot = 0;
    for (int w = 0; w < GPX_DBLOCK / 64; w++) tot += wsum[w];
    if (tot) atomicAdd(&D.chunk_cnt[((int64_t)blockIdx.x * GPX_DBLOCK) >> GPX_DCHUNK_SHIFT], tot);
  }
}

/* parked runs -> the caller's dense columns, chunk by chunk in record order.  INPLACE (commit batches): the
 * runs are parked in those very columns - all n records hold one: done; else -> dense staging. */
#define GPX_EMIT_GRID 256 /* persistent workgroups of the emit kernels: a usual batch has nothing for them to do */
template <bool INPLACE>
__global__ __launch_bounds__(GPX_DCHUNK) void k_emit_runs_direct(DevScratch X, int32_t n, int32_t nchunks,
                                                                const int32_t* __restrict__ gidx, DirectStage D,
                                                                int32_t* __restrict__ x_gidx,
                                                                int32_t* __restrict__ x_first,
                                                                int32_t* __restrict__ x_count,
                                                                int32_t* total_out, int32_t refuse,
                                                                int32_t published = 0) {
  const bool unsorted = *X.unsorted == X.epoch, marked = *D.mark == X.epoch; /* one round trip */
  if (published && !marked) return; /* k_ac_one's last workgroup has written the regular batch's count */
  if (unsorted) { /* the partition path (k_emit_runs) writes the outputs; refused: none */
    if (refuse && blockIdx.x == 0 && threadIdx.x == 0 && total_out) *total_out = 0;
    return;
  }
  if (!marked) { /* ACCEPT: no run anywhere.  COMMIT: one run per record, each at its record's index: final */
    if (blockIdx.x == 0 && threadIdx.x == 0 && total_out) *total_out = INPLACE ? n : 0;
    return;
  }
  for (int32_t w = (int32_t)blockIdx.x; w < nchunks; w += (int32_t)gridDim.x) {
    /* no run parked in this chunk: nothing to place (the last chunk publishes the total) */
    if (D.chunk_cnt[w] == 0 && w != nchunks - 1) continue;
    int32_t before = 0;
    for (int32_t t = threadIdx.x; t < w; t += GPX_DCHUNK) before += D.chunk_cnt[t];
    int32_t pre;
    block_exscan_n<GPX_DCHUNK>(before, &pre);
    const int32_t i = w * GPX_DCHUNK + (int32_t)threadIdx.x;
    const bool have = i < n && D.tag[i] == X.epoch;
    int32_t tot;
    const int32_t ex = block_exscan_n<GPX_DCHUNK>(have ? 1 : 0, &tot);
    if (have) {
      if (INPLACE) { /* source and destination are the same columns: through the staging block */
        D.st_gidx[pre + ex] = x_gidx[i];
        D.st_first[pre + ex] = x_first[i];
        D.st_count[pre + ex] = x_count[i];
      } else {
        x_gidx[pre + ex] = gidx[i];
        x_first[pre + ex] = D.st_first[i];
        x_count[pre + ex] = D.st_count[i];
      }
    }
    if (w == nchunks - 1 && threadIdx.x == 0) {
      if (total_out) *total_out = pre + tot;
      if (INPLACE) *D.st_total = pre + tot;
    }
  }
}

/* commit batches that were not one-run-per-record: the staged runs back into the caller's columns */
__global__ __launch_bounds__(GPX_BLOCK) void k_copy_runs(DevScratch X, DirectStage D, int32_t* __restrict__ x_gidx,
                                                        int32_t* __restrict__ x_first, int32_t* __restrict__ x_count) {
  const bool unsorted = *X.unsorted == X.epoch, marked = *D.mark == X.epoch;
  if (unsorted || !marked) return; /* k_emit_runs_direct<true> found the columns final */
  const int32_t m = *D.st_total;
  for (int32_t t = blockIdx.x * GPX_BLOCK + threadIdx.x; t < m; t += gridDim.x * GPX_BLOCK) {
    x_gidx[t] = D.st_gidx[t];
    x_first[t] = D.st_first[t];
    x_count[t] = D.st_count[t];
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* Small ordered batches in ONE launch.  A batch of at most 65,536 records is L2-resident: every
 * workgroup reads the whole gidx column itself to judge the order (so no k_order_check launch), applies
 * its 1024-record chunk like k_ac_direct / k_propose_direct, and places its execution runs behind those
 * of the chunks before it with a ticket per workgroup (so no k_emit_runs_direct launch either): chunks are
 * handed out by an atomic counter, so the chunks before a workgroup's own belong to workgroups that have
 * started - they are running or done, no deadlock whatever the dispatch order.  BASELINE
 * config #2's round (10 k groups: every kernel sits on the launch floor) goes from 24 launches to 12.
 * The verdict "not ordered" is published in *X.unsorted for the partition path launched behind it (or,
 * under the GPX_ORDERED_* promise, the batch is refused whole). */
#define GPX_SMALL_DIRECT_MAX_N 65536

template <bool COMMIT>
__global__ __launch_bounds__(GPX_DCHUNK) void k_ac_small(
    DevState S, DevScratch X, int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot, const int32_t* __restrict__ median,
    const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, uint8_t* __restrict__ status, DirectStage D,
    int32_t* __restrict__ x_gidx, int32_t* __restrict__ x_first, int32_t* __restrict__ x_count,
    int32_t* __restrict__ n_runs, unsigned long long* __restrict__ tickets, uint32_t epoch, int32_t refuse,
    uint32_t* __restrict__ draw, uint32_t draw_base) {
  __shared__ int32_t s_before;
  __shared__ int32_t s_w;
  /* the chunk is DRAWN, not read off blockIdx: a workgroup that waits below only ever waits for chunks
   * whose workgroups have already started - no assumption about the order workgroups are dispatched in */
  if (threadIdx.x == 0) s_w = (int32_t)(atomicAdd(draw, 1u) - draw_base);
  __syncthreads();
  const int32_t w = s_w;
  const int32_t i = w * GPX_DCHUNK + (int32_t)threadIdx.x;
  /* wave 1 of loads, as in k_ac_one: the neighbours in gidx and this record's columns; wave 2: the group's
   * acceptor state and the ring entry of this record's slot - both requested BEFORE the verdict is worked out (round 4:
   * until then the replay fetched them one dependent round trip after the other, behind the scan of the column:
   * five or six of the kernel's twelve round trips) */
  int32_t g = 0, g_prev = 0, g_next = 0, f_a = 0, f_b = 0, f_c = 0, f_bnum = 0, f_bcoord = 0;
  bool runstart = false;
  AccPre P = acc_nopre();
  if (i < n) {
    g = gidx[i];
    g_prev = i > 0 ? gidx[i - 1] : ~g;
    g_next = i + 1 < n ? gidx[i + 1] : ~g;
    f_a = slot[i], f_b = median[i], f_c = flags ? (int32_t)flags[i] : 0;
    f_bnum = bnum[i], f_bcoord = bcoord[i];
    runstart = g_prev != g; /* first record of its group's run: answers for the whole run */
    if (runstart && (uint32_t)g < (uint32_t)S.G) acc_preload(S, g, f_a, P); /* (a refused head has loaded in vain) */
  }
  const uint32_t first_bad = small_batch_first_bad<false>(n, gidx, S.G);
  if (first_bad != 0xffffffffu && !refuse) { /* no promise: the partition path launched behind takes the whole batch */
    if (w == 0 && threadIdx.x == 0) atomicMax(X.unsorted, X.epoch);
    if (i < n) status[i] = (uint32_t)g < (uint32_t)S.G ? GPX_S_OK : GPX_S_NOGROUP; /* what k_order_check leaves */
    return;
  }
  int32_t have = 0, first = 0, count = 0;
  if (runstart) {
    if ((uint32_t)i >= first_bad) {
      /* GPX_ORDERED_* broken: runs from the first violation on are refused (the first violation is always a
       * run start: gpx_one.hip.h) */
      for (int32_t j = i; j < n && (j == i || gidx[j] == g); j++) {
        if (!COMMIT) {
          r_bnum[j] = 0;
          r_bcoord[j] = 0;
          r_maxcp[j] = 0;
          r_flags[j] = 0;
        }
        status[j] = GPX_S_UNORDERED;
      }
    } else { /* replays the run in array order */
      RunIter it;
      it.gidx = gidx;
      it.bnum = bnum;
      it.bcoord = bcoord;
      it.slot = slot;
      it.median = median;
      it.flags = flags;
      it.D = D;
      it.n = n;
      it.i = i;
      it.g = g;
      it.cur = i;
      it.epoch = X.epoch;
      it.chunk = -1; /* every run is parked at its record and counted below */
      it.local = 0;
      it.count_chunks = false;
      it.st = status;
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
    }
  }
  /* the runs parked at THIS chunk's records: by this workgroup's heads, or by a head of an earlier
   * chunk whose run reaches in here.  Two rounds of flags, neither a chain: (1) "my heads are done"
   * (depends on nothing), (2) "runs parked in my chunk" (depends on round 1 of the chunks before
   * me).  A workgroup waits for ALL earlier flags of a round, never for a flag that itself waits on
   * the same round - so the depth is 2 global round trips whatever the number of chunks. */
  unsigned long long* const done = tickets + GPX_SMALL_DIRECT_MAX_N / GPX_DCHUNK;
  if (threadIdx.x == 0) s_before = 0;
  __syncthreads(); /* this chunk's heads have parked */
  if (threadIdx.x == 0)
    __hip_atomic_store(&done[w], (unsigned long long)epoch << 32, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  for (int32_t t = threadIdx.x; t < w; t += GPX_DCHUNK) {
    unsigned long long v;
    do {
      v = __hip_atomic_load(&done[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    } while ((uint32_t)(v >> 32) != epoch);
  }
  __syncthreads(); /* every head that can park a run in this chunk has */
  if (i < n) {
    have = __hip_atomic_load(&D.tag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X.epoch;
    count = have ? __hip_atomic_load(&D.st_count[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    first = have ? __hip_atomic_load(&D.st_first[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  }
  int32_t tot;
  const int32_t ex = block_exscan_n<GPX_DCHUNK>(have, &tot);
  if (threadIdx.x == 0)
    __hip_atomic_store(&tickets[w], ((unsigned long long)epoch << 32) | (uint32_t)tot, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_AGENT);
  int32_t before = 0;
  for (int32_t t = threadIdx.x; t < w; t += GPX_DCHUNK) {
    unsigned long long v;
    do {
      v = __hip_atomic_load(&tickets[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    } while ((uint32_t)(v >> 32) != epoch);
    before += (int32_t)(uint32_t)v;
  }
  if (before) atomicAdd(&s_before, before);
  __syncthreads();
  const int32_t base = s_before;
  if (have) {
    x_gidx[base + ex] = gidx[i];
    x_first[base + ex] = first;
    x_count[base + ex] = count;
  }
  if (w == (int32_t)gridDim.x - 1 && threadIdx.x == 0 && n_runs) *n_runs = base + tot;
}
