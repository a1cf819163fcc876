/*
 * gpx_kernels.hip.h — CDNA4 (gfx950) kernels of the batched consensus engine.
 *
 * Integer / indexing work only: HBM-bound, no MFMA.  Wave = 64 lanes everywhere.
 *
 * Pipeline of every batch call (docs/HISTORY.md §3).  Groups are binned into BUCKETS of GB = 2^shift
 * consecutive group indices; a batch is partitioned by bucket in one pass, then ONE WORKGROUP PER
 * BUCKET regroups its records by group in LDS and ONE LANE PER GROUP replays them in arrival order:
 *
 *   k_hist          one workgroup per `hsub` tiles of 4096 records: LDS histograms over buckets; one
 *                   returning atomic per touched bucket reserves the tiles' slices of the bucket
 *                   regions (tile_rel); also the common per-record status and, for proposals, whether
 *                   the gidx column is strictly ascending
 *   k_scatter_*     per tile: every workgroup scans the bucket totals itself, then packs record i into
 *                   a 32-byte Rec and writes it to its bucket's region (LDS cursor per bucket; lane
 *                   pairs write one record per request) — bucket-contiguous, any order inside
 *   k_bucket_*      per bucket: LDS count per local group -> scan -> keys grouped by group; long
 *                   segments (> 16) get a cooperative arrival-order sort; then one lane per group loads
 *                   its SoA state (coalesced: consecutive lanes own consecutive groups; prefetched at
 *                   kernel entry for dense batches), replays the group's records in ARRIVAL ORDER
 *                   exactly as PaxosInstanceStateMachine.handlePaxosMessage would, once per record;
 *                   dense per-record outputs go to the record's arrival index, compacted outputs
 *                   (decisions / exec runs / batches) leave as one run of 32-byte rows per bucket
 *   k_emit_*        per bucket: rows -> the caller's dense SoA columns, buckets in order
 *   k_propose_direct  proposals whose gidx column is strictly ascending need no regrouping
 *
 * Each device function cites the reference method it implements (paths relative to
 * /root/reference/src/edu/umass/cs/gigapaxos/).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpx.h"

/* timeline build (-DGPX_SAR_TRACE, scripts/ubench/sar_trace.sh; never shipped): wall-clock stamps of thread 0 of
 * every workgroup of the last small call */
#ifdef GPX_SAR_TRACE
__device__ unsigned long long* g_sar_trace = nullptr;
#define SAR_STAMP(wg, k)                                                                             \
  do {                                                                                               \
    if (threadIdx.x == 0 && g_sar_trace) g_sar_trace[(size_t)(wg) * 16 + (k)] = wall_clock64();        \
  } while (0)
#else
#define SAR_STAMP(wg, k) do { } while (0)
#endif

/* ... and of the tiled accept-reply call's two kernels (-DGPX_TL_TRACE, scripts/ubench/tiles_trace.sh): row = scatter
 * workgroup w, or 4096 + bucket; eight stamps a row */
#ifdef GPX_TL_TRACE
__device__ unsigned long long* g_tl_trace = nullptr;
#define TL_TRACE_ROWS 8192
#define TL_STAMP(row, k)                                                                             \
  do {                                                                                               \
    if (threadIdx.x == 0 && g_tl_trace) g_tl_trace[(size_t)(row) * 8 + (k)] = wall_clock64();          \
  } while (0)
#define TL_CLOCK(row, k) /* the shader clock beside the 100 MHz wall clock: what the CUs really ran at */ \
  do {                                                                                               \
    if (threadIdx.x == 0 && g_tl_trace) g_tl_trace[(size_t)(row) * 8 + (k)] = clock64();               \
  } while (0)
#else
#define TL_STAMP(row, k) do { } while (0)
#define TL_CLOCK(row, k) do { } while (0)
#endif

/* Plain aggregates for packed records and ring entries (not HIP's int4: plain structs give the
 * same dwordx4 accesses and keep the code independent of the vector-type accessor proxies). */
struct __attribute__((aligned(16))) I4 {
  int32_t x, y, z, w;
};
__host__ __device__ __forceinline__ I4 mk4(int32_t x, int32_t y, int32_t z, int32_t w) {
  I4 r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = w;
  return r;
}
/* one batch record after partitioning: exactly one 32-byte sector.
 *   accept-reply: a=slot b=acceptor c=max_cp        accept/commit: a=slot b=median_cp c=flags
 *   propose:      a=is_stop                          idx = arrival index, lg = gidx & (GB-1) */
struct __attribute__((aligned(32))) Rec {
  int32_t idx, lg, a, b, c, bnum, bcoord, pad;
};
/* one per-record output (decision / preempt / exec run), written as one full 32-byte sector at
 * the record's arrival index and gathered by the ordered compaction:
 *   accept-reply: x=bnum y=bcoord z=median_cp, slot=slot    exec runs: x=first y=count */
struct __attribute__((aligned(32))) Out {
  int32_t gidx, slot, x, y, z, kind, pad0, pad1;
};
__device__ __forceinline__ Out mk_out(int32_t gidx, int32_t slot, int32_t x, int32_t y, int32_t z,
                                      int32_t kind) {
  Out o;
  o.gidx = gidx;
  o.slot = slot;
  o.x = x;
  o.y = y;
  o.z = z;
  o.kind = kind;
  o.pad0 = 0;
  o.pad1 = 0;
  return o;
}

#define GPX_BLOCK 256   /* threads of a per-bucket / lifecycle workgroup */
#ifndef GPX_FBLOCK
#define GPX_FBLOCK 1024 /* threads of a streaming (histogram / scatter / compaction) workgroup */
#endif
#ifndef GPX_TILE
#define GPX_TILE 4096 /* records per scatter workgroup */
#endif
#define GPX_HSUB_MAX 8 /* most scatter tiles one histogram workgroup covers (LDS: 16 KiB each) */
/* Two tile sizes on purpose.  k_scatter_* is bound by what ONE CU can push out, so its tiles must
 * spread evenly over the 256 CUs x 2 resident workgroups: 3 M records in 8192-record tiles are 366
 * workgroups - 110 CUs carry two, 146 carry one, and the kernel lasts as long as the loaded ones
 * (88 us); in 4096-record tiles it is 66 us.  k_hist pays one returning atomic per (workgroup,
 * touched bucket) on only nbk addresses, so IT wants few, big tiles: one histogram workgroup covers
 * `hsub` consecutive scatter tiles - chosen per call so that there are about 245 histogram
 * workgroups, one per CU: 3 tiles = 12,288 records for 3 M votes - keeps one LDS histogram per
 * sub-tile, reserves the bucket slices of all of them with one atomic and hands
 * each sub-tile its share. */
#define GPX_TILE_ITEMS (GPX_TILE / GPX_FBLOCK)
#define GPX_TILE_VECS (GPX_TILE_ITEMS / 4)
#define GPX_SCAN_ITEMS 8 /* records per thread in the output-flag kernels (one 8-byte load) */
#define GPX_SCAN_TILE (GPX_FBLOCK * GPX_SCAN_ITEMS)
#define GPX_SMALL_SEG 16  /* segments up to this long are ordered by per-lane min-scan */
#define GPX_MIN_SHIFT 8   /* >= 256 groups per bucket */
#define GPX_MAX_BUCKETS 4096

/* group flag word */
#define GF_EXISTS 1u
#define GF_STOPPED 2u  /* PaxosAcceptor.STATES.STOPPED */
#define GF_HASCOORD 4u /* PaxosInstanceStateMachine.coordinator != null */
#define GF_PREPARING 8u /* ... and !coordinator.isActive(): running for coordinator (view change) */
#define GF_K(f) (((f) >> 8) & 0xffu)

/* proposal ring entry: bits 0..15 = responded mask (WaitforUtility.responded) */
#define PR_PRESENT 0x10000u
#define PR_STOP 0x20000u
/* accepted / committed ring flags */
#define RF_PRESENT 1
#define RF_STOP 2
#define RF_HASVALUE 4

struct DevState {
  int32_t G, kmax, W, my_id;
  uint32_t flags;
  uint32_t* g_flags;
  int32_t* g_version;
  int32_t *a_slot, *a_bnum, *a_bcoord, *a_gc;     /* PaxosAcceptor.java:94-99 */
  int32_t *c_bnum, *c_bcoord, *c_next, *c_pcount; /* PaxosCoordinatorState.java:69-105 */
  int32_t* members;                               /* [kmax][G] */
  int32_t* node_slots;                            /* [kmax][G] nodeSlotNumbers */
  uint32_t* p_ring;                               /* [W][G] myProposals */
  I4* acc_ring;                                   /* [W][G] acceptedProposals {slot, bnum, bcoord, flags}: flags bits
                                                   * 0-7 = this entry's RF_*, bits 8-15 = the RF_* of com_ring's
                                                   * entry at the same index (AccView) */
  I4* com_ring;                                   /* [W][G] committedRequests {slot,bnum,bcoord,median} */
  /* view change, coordinator side; allocated by the first gpx_election_begin (null before: no
   * group can be GF_PREPARING then) */
  uint32_t* c_wait;   /* [G] waitforMyBallot: members heard from */
  I4* co_ring;        /* [W][G] carryoverProposals {slot, bnum, bcoord, CO_PRESENT | GPX_PV_*} */
  int64_t* co_handle; /* [W][G] the caller's key of a carried-over request value */
  int64_t* p_handle;  /* [W][G] ... and of a pre-active proposal (myProposals while not active) */
};
#define CO_PRESENT 0x100

struct DevScratch {
  int32_t shift, nbk, gb; /* bucket = (gidx - g_base) >> shift; nbk buckets of gb = 1 << shift groups */
  int32_t g_base, g_end;  /* groups this pass partitions: [g_base, g_end), g_base a multiple of gb; the
                           * whole table except for accept-reply calls over more than 4 M groups, which
                           * run one pass per 4 M-group range (records of other ranges are skipped) */
  int32_t lds_recs;       /* a bucket with at most this many records is regrouped entirely in LDS */
  int32_t* bucket_tot;    /* [nbk] records per bucket of the current batch (k_hist; re-zeroed by k_bucket_*) */
  int32_t* tile_rel;      /* [ntiles][nbk] start of each tile's slice inside each bucket region (k_hist) */
  int32_t* bucket_off;    /* [nbk + 1] exclusive prefix of bucket_tot (written by tile 0 of k_scatter_*) */
  Rec* rec;               /* [n] bucket-partitioned records */
  int32_t* rank2;         /* [n] rank of a record among its group's records (LDS atomic order) */
  unsigned long long* perm; /* [n] sort keys (arrival idx << 32 | position) of buckets too big for LDS */
  Out* o_rec;             /* [n] compacted outputs of bucket b as a dense run at o_rec[bucket_off[b] ..] */
  int32_t* bucket_nout;   /* [nbk] number of compacted outputs of each bucket */
  unsigned long long* counters; /* [3] votes, outputs (decisions + preempts), dropped */
  uint32_t* unsorted;           /* [1] = `epoch` iff the current batch's gidx column is NOT strictly
                                 * ascending and in range (k_hist); stale values mean "sorted" */
  uint32_t epoch;               /* call number, > 0 */
  int32_t gate;                 /* accept-reply partition kernels launched BEHIND the sorted-runs path
                                 * (gpx_runs.hip.h): they run only if k_runs_check raised *unsorted */
  uint32_t* xabort;             /* [1] host-mapped word: a workgroup of an exchange kernel gave up waiting for the
                                 * others' tickets (xchg_wait): the engine refuses every later call */
};

/* Workgroups that wait for each other (grid_exchange, gpx_one.hip.h) wait only for workgroups that have started (or,
 * in a small grid, for a grid the host knows to be resident: xchg_ctl in gpx_engine.hip).  This is the backstop for
 * what neither covers - a wedged device, another process holding every CU while a small grid starts: a waiter gives up
 * after ten seconds of the 100 MHz wall clock (GPX_XCHG_TIMEOUT_MS), leaves the call's epoch in the host-mapped word AND the
 * verdict "first violation at record 0" in the exchange's verdict word, so that every workgroup that reads the verdict
 * afterwards - the ones that were not resident and start once the quitters have left - applies NOTHING either.  (Not a
 * transaction: a workgroup that saw every arrival an instant before another one's clock ran out has applied its records.
 * The engine is unusable from then on, and the call during which it happened says so itself: GPX_EDEVICE.) */
#define GPX_XCHG_TIMEOUT_MS 10000u /* default; GPX_XCHG_TIMEOUT_MS at engine creation (tests: a few ms) */
struct XchgWait {
  uint32_t spins = 0;
  unsigned long long t0 = 0;
  /* one more failed poll; true = give up (limit_ms of the 100 MHz wall clock have passed since the first 1024 polls) */
  __device__ __forceinline__ bool tired(uint32_t limit_ms) {
    if ((++spins & 1023u) != 0) return false;
    const unsigned long long now = wall_clock64();
    if (!t0) {
      t0 = now;
      return false;
    }
    return now - t0 > (unsigned long long)limit_ms * 100000ull;
  }
};
__device__ __forceinline__ void xchg_abort(const DevScratch& X) {
  if (X.xabort) __hip_atomic_store(X.xabort, X.epoch ? X.epoch : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* Java int subtraction (wraps) */
__device__ __forceinline__ int32_t jsub(int32_t a, int32_t b) {
  return (int32_t)((uint32_t)a - (uint32_t)b);
}
/* paxosutil/Ballot.java:60-73 */
__device__ __forceinline__ int32_t ballot_cmp(int32_t n1, int32_t c1, int32_t n2, int32_t c2) {
  return (n1 != n2) ? jsub(n1, n2) : jsub(c1, c2);
}

/* ------------------------------------------------------------------------- */
/* block-wide exclusive scan of one int per thread (256 threads = 4 waves)     */
/* Inclusive scan over the 64 lanes of a wave with DPP moves (one VALU op per step): four row_shr steps
 * scan each row of 16 lanes, row_bcast:15 carries a row's total into the next row (rows 1 and 3), row_bcast:31
 * carries lane 31's into rows 2 and 3.  __shfl_up compiles to ds_bpermute + waitcnt + select per step:
 * the scans were a sixth of k_bucket16's instructions. */
__device__ __forceinline__ int32_t wave_incscan(int32_t v) {
#ifdef GPX_SCAN_SHFL /* the portable form, kept for comparison builds */
  const int lane = threadIdx.x & 63;
  int32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  return x;
#else
  int32_t x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); /* row_shr:1, lanes without a source get 0 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); /* row_shr:2 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); /* row_shr:4 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); /* row_shr:8 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1 and 3 */
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); /* row_bcast:31 into rows 2 and 3 */
  return x;
#endif
}
/* the same with max over non-negative values (lanes without a source contribute 0) */
__device__ __forceinline__ int32_t wave_incscan_max(int32_t v) {
  int32_t x = v;
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
  return x;
}
template <int NT>
__device__ __forceinline__ int32_t block_exscan_n(int32_t v, int32_t* total) {
  __shared__ int32_t wsum[NT / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int32_t x = wave_incscan(v);
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  int32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    int32_t s = wsum[w];
    if (w < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}
__device__ __forceinline__ int32_t block_exscan(int32_t v, int32_t* total) {
  return block_exscan_n<GPX_BLOCK>(v, total);
}

/* ------------------------------------------------------------------------- */
/* front end: bucket partition without a per-tile histogram matrix               */
/*   k_hist      per tile: LDS histogram -> one global atomicAdd per touched bucket into bucket_tot
 *   k_scatter_* per tile: LDS histogram again, every workgroup scans bucket_tot itself (nbk <= 4096
 *               ints from L2), reserves its slice of each bucket region with ONE returning global
 *               atomic per touched bucket (cursor), then writes its records there.  The order
 *               inside a bucket region is whatever the atomics gave: every record carries its
 *               arrival index and the per-bucket kernels restore arrival order per group.
 * Streaming workgroups are 1024 threads x 8 records: the chip holds only a few hundred tiles, so
 * the loads of a tile must all be in flight at once (16 waves per tile, 2 x 16-byte loads per
 * column per lane). */

/* Workgroup -> tile.  Block b runs on XCD b % 8 (observed dispatch order; only speed depends on
 * it): each XCD gets a contiguous run of tiles. */
__device__ __forceinline__ int32_t tile_of_block(int32_t ntiles) {
  const int32_t per = (ntiles + 7) >> 3;
  return (int32_t)(blockIdx.x & 7) * per + (int32_t)(blockIdx.x >> 3);
}

/* LDS bucket counters under sorted input: when every active lane of the wave addresses the SAME
 * bucket (a batch the host already grouped by group, or a proposal batch in gidx order), one lane
 * adds the whole wave's count instead of 64 serialised LDS atomics on one address.  The check is
 * wave-uniform; shuffled input takes the per-lane path. */
__device__ __forceinline__ void bucket_count(int32_t* lds, int32_t b) {
  const unsigned long long act = __ballot(1);
  const int32_t b0 = __builtin_amdgcn_readfirstlane(b);
  if (__ballot(b == b0) == act) {
    if ((int)__lane_id() == __ffsll((long long)act) - 1) atomicAdd(&lds[b0], (int32_t)__popcll(act));
  } else {
    atomicAdd(&lds[b], 1);
  }
}
/* same for the cursors: returns this lane's position */
__device__ __forceinline__ int32_t bucket_take(int32_t* lds, int32_t b) {
  const unsigned long long act = __ballot(1);
  const int32_t b0 = __builtin_amdgcn_readfirstlane(b);
  if (__ballot(b == b0) == act) {
    const int lane = (int)__lane_id();
    const int leader = __ffsll((long long)act) - 1;
    const int32_t rank = (int32_t)__popcll(act & ((1ull << lane) - 1ull));
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(&lds[b0], (int32_t)__popcll(act));
    return __shfl(base, leader, 64) + rank;
  }
  return atomicAdd(&lds[b], 1);
}

/* LDS histogram of one tile over buckets; returns the number of out-of-range gidx seen by this
 * lane.  VEC: gidx is 16-byte aligned -> 4 consecutive records per lane per load. */
template <bool VEC>
__device__ __forceinline__ int32_t tile_histogram(int32_t n, int64_t base,
                                                  const int32_t* __restrict__ gidx, int32_t G,
                                                  int32_t shift, int32_t* lds, int32_t g_base, int32_t g_end) {
  int32_t bad = 0;
  const uint32_t span = (uint32_t)(g_end - g_base);
  auto one = [&](int32_t g) {
    if ((uint32_t)(g - g_base) < span)
      bucket_count(lds, (g - g_base) >> shift);
    else if ((uint32_t)g >= (uint32_t)G)
      bad++;
  };
  if (VEC) {
#pragma unroll
    for (int j = 0; j < GPX_TILE_VECS; j++) {
      const int64_t i0 = base + (int64_t)(j * GPX_FBLOCK + threadIdx.x) * 4;
      if (i0 + 3 < n) {
        const I4 g4 = *(const I4*)(gidx + i0);
        const int32_t gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int q = 0; q < 4; q++) one(gg[q]);
      } else {
        for (int q = 0; q < 4; q++)
          if (i0 + q < n) one(gidx[i0 + q]);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < GPX_TILE_ITEMS; j++) {
      const int64_t i = base + j * GPX_FBLOCK + threadIdx.x;
      if (i < n) one(gidx[i]);
    }
  }
  return bad;
}

/* per-tile bucket histogram -> bucket_tot; also writes the common per-record status (coalesced)
 * so the apply kernels only touch the rare non-OK ones */
template <bool VEC>
__global__ __launch_bounds__(GPX_FBLOCK) void k_hist(int32_t n, int32_t ntiles,
                                                    const int32_t* __restrict__ gidx, int32_t G,
                                                    DevScratch X, uint8_t* __restrict__ status,
                                                    int32_t is_votes, int32_t check_order,
                                                    int32_t hsub) {
  extern __shared__ int32_t lds[]; /* [hsub][nbk] */
  const int32_t nsuper = (ntiles + hsub - 1) / hsub;
  const int32_t st = tile_of_block(nsuper);
  if (st >= nsuper) return;
  /* check_order == 2: fallback pass of a proposal batch - k_propose_check has already judged the
   * order; a strictly ascending batch is applied by k_propose_direct and needs no partition */
  if (check_order == 2 && *X.unsorted != X.epoch) return;
  for (int32_t b = threadIdx.x; b < hsub * X.nbk; b += GPX_FBLOCK) lds[b] = 0;
  __syncthreads();
  const int64_t base = (int64_t)st * hsub * GPX_TILE;
  const int64_t end = base + (int64_t)hsub * GPX_TILE;
  int32_t bad = 0;
  for (int sub = 0; sub < hsub; sub++)
    bad += tile_histogram<VEC>(n, base + (int64_t)sub * GPX_TILE, gidx, G, X.shift, lds + sub * X.nbk, X.g_base, X.g_end);
  if (check_order == 1) {
    /* strictly ascending, in-range gidx = every group at most once: such a batch needs no
     * regrouping (k_propose_direct); anything else marks the call's epoch in *X.unsorted */
    bool out_of_order = bad != 0;
    for (int64_t i = base + threadIdx.x; i < end && i + 1 < n; i += GPX_FBLOCK)
      out_of_order |= gidx[i] >= gidx[i + 1];
    if (__syncthreads_or(out_of_order) && threadIdx.x == 0) atomicMax(X.unsorted, X.epoch);
  }
  /* status: 8 consecutive records per lane */
  if (status)
  for (int64_t i0 = base + (int64_t)threadIdx.x * 8; i0 < end; i0 += GPX_FBLOCK * 8) {
    if (i0 + 7 < n && !((uintptr_t)status & 7)) {
      {
        unsigned long long stw = 0;
        for (int q = 0; q < 8; q++)
          if ((uint32_t)gidx[i0 + q] >= (uint32_t)G) stw |= (unsigned long long)GPX_S_NOGROUP << (8 * q);
        *(unsigned long long*)(status + i0) = stw; /* GPX_S_OK == 0; PaxosManager.java:1162-1194 */
      }
    } else {
      for (int q = 0; q < 8; q++) {
        const int64_t i = i0 + q;
        if (i < n) status[i] = ((uint32_t)gidx[i] < (uint32_t)G) ? GPX_S_OK : GPX_S_NOGROUP;
      }
    }
  }
  /* is_votes < 0: a regrouping pass over records an earlier call already counted */
  if (bad && is_votes >= 0) atomicAdd(&X.counters[2], (unsigned long long)bad);
  if (st == 0 && threadIdx.x == 0 && is_votes > 0) atomicAdd(&X.counters[0], (unsigned long long)n);
  __syncthreads();
  /* reserve the slices of this workgroup's sub-tiles in every bucket region: one returning atomic
   * per touched bucket for all of them; sub-tile `sub` starts behind its predecessors' records.
   * The order of the slices inside a region is whatever the atomics give (records carry their
   * arrival index; the per-bucket kernels restore arrival order per group). */
  for (int32_t b = threadIdx.x; b < X.nbk; b += GPX_FBLOCK) {
    int32_t tot = 0;
    for (int sub = 0; sub < hsub; sub++) tot += lds[sub * X.nbk + b];
    int32_t rel = tot ? atomicAdd(&X.bucket_tot[b], tot) : 0;
    for (int sub = 0; sub < hsub; sub++) {
      const int32_t tile = st * hsub + sub;
      if (tile < ntiles) X.tile_rel[(int64_t)tile * X.nbk + b] = rel;
      rel += lds[sub * X.nbk + b];
    }
  }
}

/* Leaves in lds[b] the position where this tile's next record of bucket b goes: every workgroup
 * scans the bucket totals itself (nbk <= 4096 ints from L2: thread t owns `per` <= 4 consecutive
 * buckets) and adds the slice start k_hist reserved for it. */
__device__ __forceinline__ void scatter_init(const DevScratch& X, int32_t tile, int32_t* lds) {
  const int32_t per = (X.nbk + GPX_FBLOCK - 1) / GPX_FBLOCK;
  const int32_t b0 = threadIdx.x * per;
  const int32_t* rel = X.tile_rel + (int64_t)tile * X.nbk;
  int32_t v[GPX_MAX_BUCKETS / GPX_FBLOCK], rl[GPX_MAX_BUCKETS / GPX_FBLOCK];
  int32_t s = 0;
#pragma unroll
  for (int q = 0; q < GPX_MAX_BUCKETS / GPX_FBLOCK; q++) {
    const bool on = q < per && b0 + q < X.nbk;
    v[q] = on ? X.bucket_tot[b0 + q] : 0;
    rl[q] = on ? rel[b0 + q] : 0;
    s += v[q];
  }
  int32_t tot;
  int32_t ex = block_exscan_n<GPX_FBLOCK>(s, &tot);
#pragma unroll
  for (int q = 0; q < GPX_MAX_BUCKETS / GPX_FBLOCK; q++) {
    const int32_t b = b0 + q;
    if (q < per && b < X.nbk) {
      lds[b] = ex + rl[q];
      if (tile == 0) X.bucket_off[b] = ex;
      ex += v[q];
    }
  }
  if (tile == 0 && threadIdx.x == 0) X.bucket_off[X.nbk] = tot;
  __syncthreads();
}

/* value of the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2], no LDS traffic */
__device__ __forceinline__ int32_t lane_xor1(int32_t v) {
  return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
}

/* Writes this lane's 32-byte record r to base[pos] (pos < 0: nothing to write) together with its
 * neighbour lane: a lane can store 16 bytes per instruction, so a record written by ONE lane is two
 * requests of 16 bytes each whatever its address - k_scatter_ar issued 6 M of them for 3 M votes
 * and was bound by that, not by HBM (storing the records coalesced at their arrival index took
 * just as long).  Here the even lane writes the low halves and the odd lane the high halves of
 * both lanes' records: each store instruction then has lane pairs on 32 contiguous bytes, which
 * leave the CU as one request. */
__device__ __forceinline__ void store_rec_pair(void* __restrict__ base32, int32_t pos, const I4 lo,
                                               const I4 hi) {
  I4* base = (I4*)base32; /* element pos = the two I4 at base[2 * pos], base[2 * pos + 1] */
  const int lane = (int)__lane_id();
  const unsigned long long act = __ballot(1);
  if (!((act >> (lane ^ 1)) & 1ull)) { /* neighbour not here (tail of the batch): write it alone */
    if (pos >= 0) {
      base[2 * (int64_t)pos] = lo;
      base[2 * (int64_t)pos + 1] = hi;
    }
    return;
  }
  const bool odd = (lane & 1) != 0;
  /* the half the neighbour writes for me goes over; its half for me comes back */
  const int32_t sx = odd ? lo.x : hi.x, sy = odd ? lo.y : hi.y, sz = odd ? lo.z : hi.z,
                sw = odd ? lo.w : hi.w;
  const int32_t rx = lane_xor1(sx), ry = lane_xor1(sy), rz = lane_xor1(sz), rw = lane_xor1(sw);
  const int32_t pos_n = lane_xor1(pos);
  const int32_t pos_even = odd ? pos_n : pos, pos_odd = odd ? pos : pos_n;
  const int32_t half = odd ? 1 : 0;
  if (pos_even >= 0) /* the even lane's record: even lane writes its lo, odd lane the hi it received */
    base[2 * (int64_t)pos_even + half] = odd ? mk4(rx, ry, rz, rw) : lo;
  if (pos_odd >= 0) /* the odd lane's record: even lane writes the lo it received, odd lane its hi */
    base[2 * (int64_t)pos_odd + half] = odd ? hi : mk4(rx, ry, rz, rw);
}

__device__ __forceinline__ void put_rec(const DevScratch& X, int32_t* lds, int32_t G, int32_t mask,
                                        int64_t i, int32_t g, int32_t a, int32_t b, int32_t c,
                                        int32_t bnum, int32_t bcoord) {
  int32_t pos = -1;
  if ((uint32_t)g < (uint32_t)G) pos = bucket_take(lds, g >> X.shift);
  /* Rec = {idx, lg, a, b | c, bnum, bcoord, pad}
   * (non-temporal stores measured: 2.6x slower - the L2 merges neighbouring records of a slice
   * before they reach HBM) */
  store_rec_pair(X.rec, pos, mk4((int32_t)i, g & mask, a, b), mk4(c, bnum, bcoord, 0));
}

/* accepts / commits (and every other call whose record is {a, b, c, ballot}); for accepts also
 * zeroes the dense reply columns of dropped records.  VEC: the int columns are 16-byte aligned and
 * `flags` 4-byte aligned -> 4 consecutive records per lane per load. */
template <bool VEC>
__global__ __launch_bounds__(GPX_FBLOCK) void k_scatter_ac(
    int32_t n, int32_t ntiles, int32_t G, DevScratch X, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ bnum, const int32_t* __restrict__ bcoord,
    const int32_t* __restrict__ slot, const int32_t* __restrict__ median_cp,
    const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, int32_t only_unsorted) {
  extern __shared__ int32_t lds[];
  const int32_t tile = tile_of_block(ntiles);
  if (tile >= ntiles) return;
  if (only_unsorted && *X.unsorted != X.epoch) return; /* ordered batch: applied directly (gpx_direct.hip.h) */
  scatter_init(X, tile, lds);
  const int64_t base = (int64_t)tile * GPX_TILE;
  const int32_t mask = X.gb - 1;
  auto one = [&](int64_t i, int32_t g, int32_t a, int32_t b, int32_t c, int32_t bn, int32_t bc) {
    put_rec(X, lds, G, mask, i, g, a, b, c, bn, bc);
    if ((uint32_t)g >= (uint32_t)G && r_bnum) {
      r_bnum[i] = 0;
      r_bcoord[i] = 0;
      r_maxcp[i] = 0;
      r_flags[i] = 0;
    }
  };
  if (VEC) {
#pragma unroll
    for (int j = 0; j < GPX_TILE_VECS; j++) {
      const int64_t i0 = base + (int64_t)(j * GPX_FBLOCK + threadIdx.x) * 4;
      if (i0 + 3 < n) {
        const I4 g4 = *(const I4*)(gidx + i0), s4 = *(const I4*)(slot + i0);
        const I4 m4 = *(const I4*)(median_cp + i0);
        const I4 n4 = *(const I4*)(bnum + i0), c4 = *(const I4*)(bcoord + i0);
        const uint32_t f4 = flags ? *(const uint32_t*)(flags + i0) : 0u;
        one(i0 + 0, g4.x, s4.x, m4.x, (int32_t)(f4 & 0xffu), n4.x, c4.x);
        one(i0 + 1, g4.y, s4.y, m4.y, (int32_t)((f4 >> 8) & 0xffu), n4.y, c4.y);
        one(i0 + 2, g4.z, s4.z, m4.z, (int32_t)((f4 >> 16) & 0xffu), n4.z, c4.z);
        one(i0 + 3, g4.w, s4.w, m4.w, (int32_t)(f4 >> 24), n4.w, c4.w);
      } else {
        for (int q = 0; q < 4; q++) {
          const int64_t i = i0 + q;
          if (i < n) one(i, gidx[i], slot[i], median_cp[i], flags ? (int32_t)flags[i] : 0, bnum[i], bcoord[i]);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < GPX_TILE_ITEMS; j++) {
      const int64_t i = base + j * GPX_FBLOCK + threadIdx.x;
      if (i < n) one(i, gidx[i], slot[i], median_cp[i], flags ? (int32_t)flags[i] : 0, bnum[i], bcoord[i]);
    }
  }
}

/* proposals */
__global__ __launch_bounds__(GPX_FBLOCK) void k_scatter_pr(
    int32_t n, int32_t ntiles, int32_t G, DevScratch X, const int32_t* __restrict__ gidx,
    const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median) {
  extern __shared__ int32_t lds[];
  const int32_t tile = tile_of_block(ntiles);
  if (tile >= ntiles) return;
  if (*X.unsorted != X.epoch) return; /* strictly ascending batch: k_propose_direct handles it */
  scatter_init(X, tile, lds);
  const int64_t base = (int64_t)tile * GPX_TILE;
  const int32_t mask = X.gb - 1;
#pragma unroll
  for (int j = 0; j < GPX_TILE_ITEMS; j++) {
    const int64_t i = base + j * GPX_FBLOCK + threadIdx.x;
    if (i < n) {
      const int32_t g = gidx[i];
      put_rec(X, lds, G, mask, i, g, is_stop ? (int32_t)(is_stop[i] & 1) : 0, 0, 0, 0, 0);
      if ((uint32_t)g >= (uint32_t)G) {
        o_slot[i] = 0;
        o_bnum[i] = 0;
        o_bcoord[i] = 0;
        o_median[i] = 0;
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* per-bucket regrouping (phases A-D of every k_bucket_* kernel)                */
/* A bucket with at most X.lds_recs records (the normal case: ~K records per group) is staged in
 * LDS whole: 32-byte records + one 8-byte sort key each (arrival idx << 32 | bucket-relative
 * record position j).  Bigger buckets (a hot group) keep records and keys in global memory.
 * (Measured on MI355X: re-reading single records from L2 in the apply phase instead of staging
 * them doubles the kernel's time.) */

__device__ __forceinline__ void cmpxchg_asc(unsigned long long* a, uint32_t lo, uint32_t hi) {
  unsigned long long x = a[lo], y = a[hi];
  if (x > y) {
    a[lo] = y;
    a[hi] = x;
  }
}

/* Sorts the keys a[0 .. c) of ONE long segment ascending (= by arrival index), cooperatively by
 * the whole workgroup.  All-ascending bitonic network (first stage of every merge compares t with
 * its mirror t ^ (k-1), the rest with t ^ j), so positions >= c behave as +inf simply by being
 * skipped.  A single hot group is inherently serial under the per-group ordering contract (like
 * the Java monitor); this only has to be correct.  `a` may point to LDS or to global memory. */
__device__ void sort_long_segment(unsigned long long* a, uint32_t c) {
  uint32_t p2 = 1;
  while (p2 < c) p2 <<= 1;
  for (uint32_t k = 2; k <= p2; k <<= 1) {
    for (uint32_t t = threadIdx.x; t < c; t += blockDim.x) {
      const uint32_t q = t ^ (k - 1);
      if (q > t && q < c) cmpxchg_asc(a, t, q);
    }
    __syncthreads();
    for (uint32_t j = k >> 2; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < c; t += blockDim.x) {
        const uint32_t q = t ^ j;
        if (q > t && q < c) cmpxchg_asc(a, t, q);
      }
      __syncthreads();
    }
  }
}

/* block-wide exclusive scan for a runtime block size (<= 1024 threads): the wave totals (at most 16)
 * are scanned by every wave itself - one LDS read, one wave scan, two lane reads - instead of a loop
 * over a runtime number of waves */
__device__ __forceinline__ int32_t block_exscan_rt(int32_t v, int32_t* total) {
  __shared__ int32_t wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (int)(blockDim.x >> 6);
  const int32_t x = wave_incscan(v);
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  const int32_t ps = wave_incscan(lane < nw ? wsum[lane] : 0);
  const int32_t tot = __shfl(ps, nw - 1, 64);
  const int32_t base = __shfl(ps, wid > 0 ? wid - 1 : 0, 64);
  __syncthreads();
  *total = tot;
  return (wid > 0 ? base : 0) + x - v;
}

/* ... of v, and the block's sum of u, behind the same two barriers */
__device__ __forceinline__ int32_t block_exscan_and_sum_rt(int32_t v, int32_t u, int32_t* vtotal, int32_t* utotal) {
  __shared__ int32_t wsum2[32];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (int)(blockDim.x >> 6);
  const int32_t x = wave_incscan(v), y = wave_incscan(u);
  if (lane == 63) {
    wsum2[wid] = x;
    wsum2[16 + wid] = y;
  }
  __syncthreads();
  const int32_t ps = wave_incscan(lane < nw ? wsum2[lane] : 0);
  const int32_t pu = wave_incscan(lane < nw ? wsum2[16 + lane] : 0);
  const int32_t tot = __shfl(ps, nw - 1, 64);
  const int32_t base = __shfl(ps, wid > 0 ? wid - 1 : 0, 64);
  *utotal = __shfl(pu, nw - 1, 64);
  __syncthreads();
  *vtotal = tot;
  return (wid > 0 ? base : 0) + x - v;
}

/* What one workgroup sees of its bucket after regrouping: the bucket's records (bucket-relative
 * index j) and keys: the records of local group lg are keys[loff[lg] .. loff[lg] + lcnt[lg]),
 * key = arrival idx << 32 | j.  Both in LDS or both in global memory (see above).
 * Only the five payload words of a record are staged (the arrival index lives in the key, the
 * local group is implied by the key segment): word f of record j is pay[j * rs + f * fs] —
 * structure-of-arrays in LDS (rs = 1, fs = lds_recs: staging writes and sequential reads are
 * bank-conflict free), the Rec itself in global memory (rs = 8, fs = 1, pay = &rec[0].a).
 * 28 bytes of LDS per record instead of 40: one more workgroup per CU, and these kernels are
 * latency-bound (measured: 2 / 3 / 4 workgroups per CU -> 100 / 78 / 67 us for k_bucket_ar). */
struct BucketView {
  int32_t* pay;
  int32_t rs, fs;
  unsigned long long* keys;
  int32_t* lcnt;
  int32_t* loff;
};

#define GPX_PAY_WORDS 5
#define GPX_BUCKET_ITEMS 8 /* max LDS-staged records per thread (X.lds_recs <= 8 * threads) */
/* dynamic LDS of every k_bucket_* kernel: lcnt[gb] | loff[gb] | keys[lds_recs] | pay[5][lds_recs] */
#define GPX_BUCKET_LDS_BYTES(gb, lds_recs) \
  ((size_t)(gb) * 8 + (size_t)(lds_recs) * (8 + 4 * GPX_PAY_WORDS))

/* Returns false (whole workgroup) when the bucket received no record.  after_a() runs once the
 * loads of phase A have been issued and consumed: the place for loads that depend on state fetched
 * at kernel entry (they overlap phases B-D). */
template <class F>
__device__ __forceinline__ bool bucket_prepare(const DevScratch& X, int32_t* lds, BucketView* bv,
                                               F after_a) {
  const int32_t b = blockIdx.x;
  const int32_t boff = X.bucket_off[b];
  const int32_t nb = X.bucket_off[b + 1] - boff;
  if (threadIdx.x == 0) {
    X.bucket_tot[b] = 0; /* ready for the next batch's k_hist */
    if (nb == 0) X.bucket_nout[b] = 0;
  }
  if (nb == 0) return false;
  const int32_t gb = X.gb;
  const int32_t nt = (int32_t)blockDim.x;
  int32_t* lcnt = lds;
  int32_t* loff = lds + gb;
  unsigned long long* keysL = (unsigned long long*)(lds + 2 * gb);
  int32_t* payL = (int32_t*)(keysL + X.lds_recs);
  const bool in_lds = nb <= X.lds_recs;
  Rec* recG = X.rec + boff;
  bv->lcnt = lcnt;
  bv->loff = loff;
  bv->pay = in_lds ? payL : &recG[0].a;
  bv->rs = in_lds ? 1 : (int32_t)(sizeof(Rec) / 4);
  bv->fs = in_lds ? X.lds_recs : 1;
  bv->keys = in_lds ? keysL : (X.perm + boff);
  for (int32_t l = threadIdx.x; l < gb; l += nt) lcnt[l] = 0;
  __syncthreads();
  /* A: stage + count per local group (LDS atomics); remember each record's rank */
  int32_t ix[GPX_BUCKET_ITEMS], lr[GPX_BUCKET_ITEMS]; /* lr = lg << 16 | rank */
  if (in_lds) {
#pragma unroll
    for (int m = 0; m < GPX_BUCKET_ITEMS; m++) {
      const int32_t j = m * nt + threadIdx.x;
      ix[m] = 0;
      lr[m] = 0;
      if (j < nb) {
        const Rec r = recG[j]; /* coalesced 32 B per lane */
        const int32_t L = X.lds_recs;
        payL[j] = r.a;
        payL[L + j] = r.b;
        payL[2 * L + j] = r.c;
        payL[3 * L + j] = r.bnum;
        payL[4 * L + j] = r.bcoord;
        ix[m] = r.idx;
        lr[m] = (r.lg << 16) | atomicAdd(&lcnt[r.lg], 1);
      }
    }
  } else {
    for (int32_t j = threadIdx.x; j < nb; j += nt)
      X.rank2[boff + j] = atomicAdd(&lcnt[recG[j].lg], 1);
  }
  __syncthreads();
  after_a();
  /* B: exclusive scan lcnt -> loff; thread t owns gb / threads consecutive groups */
  const int32_t per = gb / nt;
  int32_t s = 0;
  for (int32_t q = 0; q < per; q++) s += lcnt[threadIdx.x * per + q];
  int32_t tot;
  int32_t ex = block_exscan_rt(s, &tot);
  int32_t any_long = 0;
  for (int32_t q = 0; q < per; q++) {
    const int32_t l = threadIdx.x * per + q;
    loff[l] = ex;
    ex += lcnt[l];
    any_long |= lcnt[l] > GPX_SMALL_SEG;
  }
  any_long = __syncthreads_or(any_long);
  /* C: keys, grouped by local group */
  if (in_lds) {
#pragma unroll
    for (int m = 0; m < GPX_BUCKET_ITEMS; m++) {
      const int32_t j = m * nt + threadIdx.x;
      if (j < nb)
        keysL[loff[lr[m] >> 16] + (lr[m] & 0xffff)] =
            ((unsigned long long)(uint32_t)ix[m] << 32) | (uint32_t)j;
    }
  } else {
    for (int32_t j = threadIdx.x; j < nb; j += nt) {
      const int2 h = *(const int2*)&recG[j]; /* {idx, lg} */
      X.perm[boff + loff[h.y] + X.rank2[boff + j]] = ((unsigned long long)(uint32_t)h.x << 32) | (uint32_t)j;
    }
  }
  __syncthreads();
  /* D: arrival-order sort of long segments (rare) */
  if (any_long) {
    for (int32_t l = 0; l < gb; l++) {
      const int32_t c = lcnt[l]; /* uniform across the workgroup */
      if (c > GPX_SMALL_SEG) sort_long_segment(bv->keys + loff[l], (uint32_t)c);
    }
  }
  return true;
}

/* Iterates one group's records in arrival order.  c <= 4 (the normal case): keys sorted in
 * registers; longer segments are sorted in place first (4 < c <= GPX_SMALL_SEG by this lane,
 * beyond that cooperatively in bucket_prepare), then read sequentially.
 * emit() parks an output of the CURRENT record in that record's slot and lists the slot in
 * keys[nout]: entry nout <= done - 1 has always been consumed already. */
struct GroupIter {
  int32_t* pay;             /* bucket payload base (BucketView) */
  int32_t rs, fs;
  unsigned long long* keys; /* segment base */
  int32_t c, done, nout;
  uint32_t cur;
  unsigned long long k0, k1, k2, k3;
  __device__ __forceinline__ void init(const BucketView& bv, int32_t l, int32_t n) {
    pay = bv.pay;
    rs = bv.rs;
    fs = bv.fs;
    keys = bv.keys + bv.loff[l];
    c = n;
    done = 0;
    nout = 0;
    cur = 0;
    k0 = k1 = k2 = k3 = 0;
    if (n <= 4) {
      const unsigned long long inf = ~0ull;
      k0 = keys[0];
      k1 = (n > 1) ? keys[1] : inf;
      k2 = (n > 2) ? keys[2] : inf;
      k3 = (n > 3) ? keys[3] : inf;
#define GPX_CSWAP(x, y)              \
  if (x > y) {                       \
    const unsigned long long t_ = x; \
    x = y;                           \
    y = t_;                          \
  }
      GPX_CSWAP(k0, k1)
      GPX_CSWAP(k2, k3)
      GPX_CSWAP(k0, k2)
      GPX_CSWAP(k1, k3)
      GPX_CSWAP(k1, k2)
#undef GPX_CSWAP
    } else if (n <= GPX_SMALL_SEG) {
      for (int32_t i = 1; i < n; i++) { /* insertion sort, this lane only */
        const unsigned long long x = keys[i];
        int32_t p = i - 1;
        while (p >= 0 && keys[p] > x) {
          keys[p + 1] = keys[p];
          p--;
        }
        keys[p + 1] = x;
      }
    }
  }
  /* next record in arrival order; false when exhausted */
  __device__ __forceinline__ bool next(Rec& out) {
    if (done >= c) return false;
    unsigned long long k;
    if (c <= 4)
      k = done == 0 ? k0 : (done == 1 ? k1 : (done == 2 ? k2 : k3));
    else
      k = keys[done];
    cur = (uint32_t)k;
    const int32_t* p = pay + (int64_t)cur * rs;
    out.idx = (int32_t)(k >> 32);
    out.a = p[0];
    out.b = p[fs];
    out.c = p[2 * fs];
    out.bnum = p[3 * fs];
    out.bcoord = p[4 * fs];
    done++;
    return true;
  }
  /* parks {slot, x, y, z, kind} of an output of the CURRENT record in that record's payload words
   * (already consumed) and lists the record in keys[nout]: entry nout <= done - 1 has always been
   * consumed already */
  __device__ __forceinline__ void emit(int32_t slot, int32_t x, int32_t y, int32_t z, int32_t kind) {
    int32_t* p = pay + (int64_t)cur * rs;
    p[0] = slot;
    p[fs] = x;
    p[2 * fs] = y;
    p[3 * fs] = z;
    p[4 * fs] = kind;
    keys[nout++] = cur;
  }
  /* same, parked in the slot of an EARLIER record of this group (its payload is consumed too) */
  __device__ __forceinline__ void emit_at(uint32_t where, int32_t slot, int32_t x, int32_t y, int32_t z,
                                          int32_t kind) {
    int32_t* p = pay + (int64_t)where * rs;
    p[0] = slot;
    p[fs] = x;
    p[2 * fs] = y;
    p[3 * fs] = z;
    p[4 * fs] = kind;
    keys[nout++] = where;
  }
};

/* PaxosCoordinatorState.getMedianMinus (PaxosCoordinatorState.java:867-875): element of rank
 * (k even ? k/2-1 : k/2) in signed ascending order.  Rank selection, ties by index. */
template <int KMAX>
__device__ __forceinline__ int32_t median_minus(const int32_t (&ns)[KMAX], int32_t k) {
  /* the VALUE at a rank does not depend on how ties are ordered: three members (the usual group) are
   * one v_med3_i32 instead of nine ranked compares */
  if (KMAX >= 3 && k == 3) {
    const int32_t lo = ns[0] < ns[1] ? ns[0] : ns[1], hi = ns[0] < ns[1] ? ns[1] : ns[0];
    const int32_t m = hi < ns[2] ? hi : ns[2];
    return lo > m ? lo : m;
  }
  const int32_t target = (k % 2 == 0) ? (k / 2 - 1) : (k / 2);
  int32_t res = 0;
#pragma unroll
  for (int j = 0; j < KMAX; j++) {
    if (j < k) {
      int32_t r = 0;
#pragma unroll
      for (int i = 0; i < KMAX; i++)
        if (i < k) r += (ns[i] < ns[j]) || (ns[i] == ns[j] && i < j);
      if (r == target) res = ns[j];
    }
  }
  return res;
}

/* ------------------------------------------------------------------------- */
/* compacted outputs (decisions / preempts / exec runs)                         */
/* Output order contract (include/gpx.h): grouped by gidx ascending; the entries of one group in
 * the array order of the records that produced them.  That is the order the per-bucket kernels
 * produce for free (one lane per group, groups of a bucket consecutive, buckets consecutive), and
 * the shape the reference's next stage wants anyway: PaxosPacketBatcher keys outgoing decisions
 * by paxosID (PaxosPacketBatcher.java:121-156).
 *
 * During the replay a lane parks each output in the LDS (or scratch) slot of the record that
 * produced it and lists the slot in its group's key segment (GroupIter::emit).  bucket_emit then
 * writes the bucket's outputs, group-major, as one dense run of 32-byte rows at o_rec[boff ..]
 * (outputs <= records, so the bucket's own record range always has room) and their count to
 * bucket_nout[b]; k_emit_* turns the per-bucket runs into the caller's dense SoA columns. */

/* all lanes of the workgroup; the number of outputs of each group is already stored in bv.lcnt[l]
 * (overwriting the record count, which is no longer needed).
 * (Measured and rejected: letting this kernel place its outputs in the caller's columns itself -
 * offset from a decoupled look-back over the buckets' counts, or from one returning atomic per
 * workgroup - costs the per-bucket kernel exactly the 13-14 us k_emit_* takes: the kernel is
 * latency-bound and every extra dependent global round trip per workgroup shows up in full.) */
__device__ __forceinline__ void bucket_emit(const DevScratch& X, const BucketView& bv) {
  __syncthreads();
  const int32_t b = blockIdx.x;
  const int32_t boff = X.bucket_off[b];
  const int32_t nt = (int32_t)blockDim.x;
  const int32_t per = X.gb / nt;
  int32_t s = 0;
  for (int32_t q = 0; q < per; q++) s += bv.lcnt[threadIdx.x * per + q];
  int32_t tot;
  int32_t ex = block_exscan_rt(s, &tot);
  Out* dst = X.o_rec + boff;
  const int32_t g0 = b << X.shift;
  for (int32_t q = 0; q < per; q++) {
    const int32_t l = threadIdx.x * per + q;
    const int32_t d = bv.lcnt[l];
    const unsigned long long* kk = bv.keys + bv.loff[l];
    /* rows are 32 bytes: written by lane pairs (store_rec_pair), all lanes of the wave in step */
    for (int32_t t = 0; __any(t < d); t++) {
      int32_t pos = -1;
      I4 lo = mk4(0, 0, 0, 0), hi = mk4(0, 0, 0, 0);
      if (t < d) {
        const int32_t* p = bv.pay + (int64_t)(uint32_t)kk[t] * bv.rs;
        pos = ex + t;
        lo = mk4(g0 + l, p[0], p[bv.fs], p[2 * bv.fs]);   /* Out = {gidx, slot, x, y | z, kind, 0, 0} */
        hi = mk4(p[3 * bv.fs], p[4 * bv.fs], 0, 0);
      }
      store_rec_pair(dst, pos, lo, hi);
    }
    ex += d;
  }
  if (threadIdx.x == 0) X.bucket_nout[b] = tot;
}

/* Number of outputs of the buckets before this one (every workgroup sums the counts itself: at
 * most GPX_MAX_BUCKETS ints from L2); the last bucket publishes the total. */
__device__ __forceinline__ int32_t emit_base(const DevScratch& X, int32_t* total_out,
                                             unsigned long long* acc) {
  const int32_t b = blockIdx.x;
  int32_t before = 0;
  for (int32_t t = threadIdx.x; t < b; t += GPX_BLOCK) before += X.bucket_nout[t];
  int32_t pre;
  block_exscan(before, &pre);
  if (b == (int32_t)gridDim.x - 1 && threadIdx.x == 0) {
    const int32_t tot = pre + X.bucket_nout[b];
    if (total_out) *total_out = tot;
    if (acc) atomicAdd(acc, (unsigned long long)tot);
  }
  return pre;
}

/* decisions: d_* columns */
__global__ __launch_bounds__(GPX_BLOCK) void k_emit_dec(
    DevScratch X, int32_t* __restrict__ d_gidx, int32_t* __restrict__ d_slot,
    int32_t* __restrict__ d_bnum, int32_t* __restrict__ d_bcoord, int32_t* __restrict__ d_median,
    uint8_t* __restrict__ d_kind, int32_t* total_out, unsigned long long* acc) {
  const int32_t out0 = emit_base(X, total_out, acc);
  const int32_t nd = X.bucket_nout[blockIdx.x];
  const Out* src = X.o_rec + X.bucket_off[blockIdx.x];
  for (int32_t t = threadIdx.x; t < nd; t += GPX_BLOCK) {
    const Out r = src[t];
    d_gidx[out0 + t] = r.gidx;
    d_slot[out0 + t] = r.slot;
    d_bnum[out0 + t] = r.x;
    d_bcoord[out0 + t] = r.y;
    d_median[out0 + t] = r.z;
    d_kind[out0 + t] = (uint8_t)r.kind;
  }
}

/* exec runs: (gidx, first, count) */
__global__ __launch_bounds__(GPX_BLOCK) void k_emit_runs(DevScratch X, int32_t* __restrict__ x_gidx,
                                                        int32_t* __restrict__ x_first,
                                                        int32_t* __restrict__ x_count,
                                                        int32_t* total_out) {
  if (*X.unsorted != X.epoch) return; /* ordered batch: k_emit_runs_direct wrote the outputs */
  const int32_t out0 = emit_base(X, total_out, nullptr);
  const int32_t nd = X.bucket_nout[blockIdx.x];
  const Out* src = X.o_rec + X.bucket_off[blockIdx.x];
  for (int32_t t = threadIdx.x; t < nd; t += GPX_BLOCK) {
    const Out r = src[t];
    x_gidx[out0 + t] = r.gidx;
    x_first[out0 + t] = r.x;
    x_count[out0 + t] = r.y;
  }
}

/* ------------------------------------------------------------------------- */
/* coordinator side                                                             */
/* Coordinator-side state of one group, fetched by the lane that owns it.  In a dense batch the
 * loads are issued at kernel entry, before the bucket's records are staged, so that their latency
 * hides behind phases A-D of bucket_prepare (these kernels are latency-bound: each dependent
 * round trip to HBM costs a few microseconds per workgroup). */
template <int KMAX>
struct CoordPre {
  uint32_t gf;
  int32_t my_bnum, my_bcoord, next, pcount;
  int32_t mem[KMAX], ns[KMAX];
  uint32_t pe; /* speculative: myProposals entry of slot pe_slot = next - 1 (the usual outstanding slot), or of the
                * group's first vote's slot (coord_preload_ring_at) */
  int32_t pe_slot;
  bool have_pe;
};
template <int KMAX>
__device__ __forceinline__ void coord_preload(const DevState& S, int32_t g, CoordPre<KMAX>& P) {
  const int32_t G = S.G;
  P.gf = S.g_flags[g];
  P.my_bnum = S.c_bnum[g];
  P.my_bcoord = S.c_bcoord[g];
  P.next = S.c_next[g];
  P.pcount = S.c_pcount[g];
#pragma unroll
  for (int j = 0; j < KMAX; j++) { /* rows >= k of a group hold 0 (k_group_create) */
    P.mem[j] = (j < S.kmax) ? S.members[(int64_t)j * G + g] : 0;
    P.ns[j] = (j < S.kmax) ? S.node_slots[(int64_t)j * G + g] : 0;
  }
  P.pe = 0;
  P.pe_slot = 0;
  P.have_pe = false;
}
template <int KMAX>
__device__ __forceinline__ void coord_preload_ring(const DevState& S, int32_t g, CoordPre<KMAX>& P) {
  P.pe_slot = jsub(P.next, 1);
  P.pe = S.p_ring[(int64_t)(P.pe_slot & (S.W - 1)) * S.G + g];
  P.have_pe = true;
}
/* the ring entry of a slot the caller already knows (its first vote's): requested TOGETHER with coord_preload's
 * loads, not behind `next` - one round trip instead of two (the entry of any slot value lies inside the ring) */
template <int KMAX>
__device__ __forceinline__ void coord_preload_ring_at(const DevState& S, int32_t g, int32_t slot, CoordPre<KMAX>& P) {
  P.pe_slot = slot;
  P.pe = S.p_ring[(int64_t)(slot & (S.W - 1)) * S.G + g];
  P.have_pe = true;
}

/* PaxosInstanceStateMachine.handleAcceptReply (PISM:1248-1364) ->              */
/* PaxosCoordinator.handleAcceptReply (PaxosCoordinator.java:210-250) ->        */
/* PaxosCoordinatorState.handleAcceptReplyMyBallot / HigherBallot (:597-683)    */
template <int KMAX, class IT>
__device__ __forceinline__ void apply_ar_group(const DevState& S, const DevScratch& X, int32_t g,
                                               IT& it, uint8_t* __restrict__ status,
                                               const CoordPre<KMAX>& P) {
  const int32_t G = S.G;
  const uint32_t gf = P.gf;
  if (!(gf & GF_EXISTS) || (gf & GF_STOPPED)) {
    /* PaxosManager.java:1162-1194 / PaxosInstanceStateMachine.java:456-460: dropped */
    const uint8_t st = (gf & GF_EXISTS) ? GPX_S_STOPPED : GPX_S_NOGROUP;
    Rec r;
    unsigned long long nd = 0;
    while (it.next(r)) {
      if (status) status[r.idx] = st;
      nd++;
    }
    atomicAdd(&X.counters[2], nd); /* rare path */
    return;
  }
  const int32_t k = (int32_t)GF_K(gf);
  bool has_coord = (gf & GF_HASCOORD) != 0;
  const bool preparing = (gf & GF_PREPARING) != 0;
  const int32_t my_bnum = P.my_bnum, my_bcoord = P.my_bcoord;
  const int32_t next = P.next;
  int32_t pcount = P.pcount;
  const int32_t pcount0 = pcount;
  const int32_t Wm = S.W - 1;
  int32_t mem[KMAX], ns[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; j++) {
    mem[j] = (j < k) ? P.mem[j] : 0;
    ns[j] = (j < k) ? P.ns[j] : 0;
  }
  bool ns_dirty = false;
  /* one-entry write-back register cache over this group's myProposals ring: the K votes of one
   * slot (the normal case) cost one load and one store instead of K dependent round trips */
  int64_t pc_off = P.have_pe ? ((int64_t)(P.pe_slot & Wm) * G + g) : -1;
  uint32_t pc_val = P.pe;
  bool pc_dirty = false;
  auto pr_load = [&](int64_t off) -> uint32_t {
    if (off != pc_off) {
      if (pc_dirty) S.p_ring[pc_off] = pc_val;
      pc_val = S.p_ring[off];
      pc_off = off;
      pc_dirty = false;
    }
    return pc_val;
  };
  auto pr_store = [&](uint32_t v) {
    pc_val = v;
    pc_dirty = true;
  };
  Rec r;
  while (it.next(r)) {
    const int32_t slot = r.a, acc = r.b, maxcp = r.c;
    if (!has_coord) continue; /* PaxosCoordinator.java:196-198: c == null -> null */
    const int32_t cmp = ballot_cmp(r.bnum, r.bcoord, my_bnum, my_bcoord);
    const int32_t d = jsub(next, slot); /* slot in myProposals' window iff 1 <= d <= W */
    const bool inwin = (d >= 1) && (d <= S.W) && !preparing; /* !isActive() -> null (:212-213) */
    if (cmp > 0) {
      /* handleAcceptReplyHigherBallot :661-675 */
      if (inwin) {
        const uint32_t e = pr_load((int64_t)(slot & Wm) * G + g);
        if (e & PR_PRESENT) {
          pr_store(0);
          pcount--;
          it.emit(slot, my_bnum, my_bcoord, -1, GPX_D_PREEMPTED); /* preempt(): median stays -1 */
        }
      }
      /* nullifyCoordinatorIfPreemptedFully, PISM:1361-1364: active or not */
      if (pcount == 0) has_coord = false;
    } else if (cmp == 0 && !preparing) {
      /* handleAcceptReplyMyBallot :597-640; recordSlotNumber :809-825 (plain <) */
      int32_t midx = -1;
#pragma unroll
      for (int q = 0; q < KMAX; q++) {
        if (q < k && mem[q] == acc) {
          midx = q; /* WaitforUtility.getIndex: last match */
          if (ns[q] < maxcp) {
            ns[q] = maxcp;
            ns_dirty = true;
          }
        }
      }
      if (inwin) {
        uint32_t e = pr_load((int64_t)(slot & Wm) * G + g);
        if (e & PR_PRESENT) {
          if (midx >= 0) e |= (1u << midx); /* updateHeardFrom :51-62 */
          if (__popc(e & 0xffffu) > k / 2) { /* heardFromMajority :64-68 */
            pr_store(0);
            pcount--;
            it.emit(slot, my_bnum, my_bcoord, median_minus<KMAX>(ns, k), GPX_D_DECISION);
          } else {
            pr_store(e);
          }
        }
      }
    }
    /* cmp < 0: reply to a lower ballot, ignored (PaxosCoordinator.java:241-247) */
  }
  if (pc_dirty) S.p_ring[pc_off] = pc_val;
  if (ns_dirty) {
#pragma unroll
    for (int q = 0; q < KMAX; q++)
      if (q < k) S.node_slots[(int64_t)q * G + g] = ns[q];
  }
  if (pcount != pcount0) S.c_pcount[g] = pcount;
  if (!has_coord && (gf & GF_HASCOORD)) S.g_flags[g] = gf & ~(GF_HASCOORD | GF_PREPARING);
}

/* The coordinator's STEADY STATE as straight-line code: every vote of the group answers ONE outstanding slot
 * s0 at the group's current ballot, the coordinator exists, is active and its acceptor is not stopped.  Then
 * only the cmp == 0 branch of apply_ar_group can run (handleAcceptReplyMyBallot, PCS:597-640): member bit,
 * nodeSlotNumbers max (recordSlotNumber, plain <, PCS:809-825), majority test, getMedianMinus on the deciding
 * vote; later votes of the slot only move nodeSlotNumbers (PCS:607 runs before the pstate == null test).
 * Shared by the per-bucket kernel's wave-uniform fast path and the sorted-runs kernel. */
template <int KMAX>
struct SteadyGroup {
  int32_t mem[KMAX], ns[KMAX];
  int32_t k, pcount, pcount0;
  uint32_t e, e0;
  bool ns_dirty;
  /* eligibility of the GROUP (the caller checks its votes: one slot s0, the coordinator's ballot) */
  static __device__ __forceinline__ bool group_ok(const CoordPre<KMAX>& P) {
    return (P.gf & (GF_EXISTS | GF_STOPPED | GF_HASCOORD | GF_PREPARING)) == (GF_EXISTS | GF_HASCOORD);
  }
  static __device__ __forceinline__ bool slot_ok(const DevState& S, const CoordPre<KMAX>& P, int32_t s0) {
    const int32_t d = jsub(P.next, s0);
    return d >= 1 && d <= S.W;
  }
  __device__ __forceinline__ void init(const DevState& S, int32_t g, const CoordPre<KMAX>& P, int32_t s0) {
    k = (int32_t)GF_K(P.gf);
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
      mem[j] = (j < k) ? P.mem[j] : 0;
      ns[j] = (j < k) ? P.ns[j] : 0;
    }
    const int64_t off = (int64_t)(s0 & (S.W - 1)) * S.G + g;
    e0 = (P.have_pe && s0 == jsub(P.next, 1)) ? P.pe : S.p_ring[off];
    e = e0;
    pcount = pcount0 = P.pcount;
    ns_dirty = false;
  }
  /* one vote; true when it completes the majority: *median = the decision's medianCheckpointedSlot */
  __device__ __forceinline__ bool vote(int32_t acc, int32_t maxcp, int32_t* median) {
    int32_t midx = -1;
#pragma unroll
    for (int q = 0; q < KMAX; q++) {
      if (q < k && mem[q] == acc) {
        midx = q; /* WaitforUtility.getIndex: last match */
        if (ns[q] < maxcp) { /* recordSlotNumber :809-825 (plain <) */
          ns[q] = maxcp;
          ns_dirty = true;
        }
      }
    }
    if (e & PR_PRESENT) {
      if (midx >= 0) e |= (1u << midx);   /* updateHeardFrom :51-62 */
      if (__popc(e & 0xffffu) > k / 2) { /* heardFromMajority :64-68 */
        *median = median_minus<KMAX>(ns, k);
        e = 0;
        pcount--;
        return true;
      }
    }
    return false;
  }
  __device__ __forceinline__ void finish(const DevState& S, int32_t g, int32_t s0) {
    if (e != e0) S.p_ring[(int64_t)(s0 & (S.W - 1)) * S.G + g] = e;
    if (ns_dirty) {
#pragma unroll
      for (int q = 0; q < KMAX; q++)
        if (q < k) S.node_slots[(int64_t)q * S.G + g] = ns[q];
    }
    if (pcount != pcount0) S.c_pcount[g] = pcount;
  }
};

/* ------------------------------------------------------------------------- */
/* acceptor helpers (one lane owns the group: plain loads/stores)               */

struct AccState {
  int32_t slot, bnum, bcoord, gc;
  bool stopped;
};

/* One lane's view of its group's accepted ring.  Entry w = {slot, bnum, bcoord, flags}: the flag word carries
 * BOTH flag bytes of ring index w - bits 0-7 of acceptedProposals' entry (RF_PRESENT | RF_STOP), bits 8-15 of
 * committedRequests' entry (RF_PRESENT | RF_STOP | RF_HASVALUE; its other four words live in com_ring) - so
 * one 16-byte load answers "what did I accept for this slot" and "is a commit waiting there" together (round
 * 2 kept two separate byte arrays: two more loads, and byte stores, per record).  Two entries are cached in
 * registers, stores go straight through: the record's own slot and the one the garbage collection or the
 * execution loop looks at stay at hand instead of costing a dependent round trip each time. */
#define AF_MASK 0xffu
#define CF_SHIFT 8
#define CF_MASK 0xff00u
struct AccView {
  I4* ring;
  int64_t G;
  int32_t g;
  /* two cached entries as plain scalars, updated with selects only: anything that makes the compiler
   * take the address of a cached word (an if / else between the two copies) sends the whole view to
   * scratch memory */
  int32_t w0, x0, y0, z0, f0; /* ring index (-1 = none) and the entry's four words */
  int32_t w1, x1, y1, z1, f1;
  __device__ __forceinline__ void init(const DevState& S, int32_t g_) {
    ring = S.acc_ring;
    G = S.G;
    g = g_;
    w0 = w1 = -1;
    x0 = y0 = z0 = f0 = 0;
    x1 = y1 = z1 = f1 = 0;
  }
  /* an entry the caller fetched ahead of time */
  __device__ __forceinline__ void preset(int32_t w, const I4& v) {
    w0 = w;
    x0 = v.x;
    y0 = v.y;
    z0 = v.z;
    f0 = v.w;
  }
  __device__ __forceinline__ void put(int32_t w, const I4& v) {
    const bool in0 = w == w0 || w0 < 0;
    w0 = in0 ? w : w0;
    x0 = in0 ? v.x : x0;
    y0 = in0 ? v.y : y0;
    z0 = in0 ? v.z : z0;
    f0 = in0 ? v.w : f0;
    w1 = in0 ? w1 : w;
    x1 = in0 ? x1 : v.x;
    y1 = in0 ? y1 : v.y;
    z1 = in0 ? z1 : v.z;
    f1 = in0 ? f1 : v.w;
  }
  __device__ __forceinline__ I4 rd(int32_t w) {
    const bool h0 = w == w0, h1 = w == w1;
    I4 v = mk4(h0 ? x0 : x1, h0 ? y0 : y1, h0 ? z0 : z1, h0 ? f0 : f1);
    if (!(h0 || h1)) {
      v = ring[(int64_t)w * G + g];
      put(w, v);
    }
    return v;
  }
  __device__ __forceinline__ void wr(int32_t w, const I4& v) {
    ring[(int64_t)w * G + g] = v;
    put(w, v);
  }
  /* only the flag word (the entry must have been read: it is cached) */
  __device__ __forceinline__ void wr_flags(int32_t w, int32_t f) {
    ((int32_t*)&ring[(int64_t)w * G + g])[3] = f;
    f0 = (w == w0) ? f : f0;
    f1 = (w == w1 && w != w0) ? f : f1;
  }
};

/* PaxosAcceptor.garbageCollectAccepted (PaxosAcceptor.java:476-494) and, at its end, garbageCollectDecisions
 * (:496-506).  In ordinary operation garbageCollectDecisions finds nothing: committedRequests only holds slots
 * >= _slot (put only if slot - _slot >= 0, :341; removed on execution) and it drops slots "before" gcSlot <=
 * _slot - 1.  But both methods compare by SUBTRACTION, and a gcSlot that lies nearly half the int range behind
 * _slot - a stale medianCheckpointedSlot of a retransmitted ACCEPT after the slots have crossed
 * Integer.MAX_VALUE (median 0 against slots near -2^31) - makes `gcSlot - key > 0` true for keys AHEAD of _slot:
 * the Java then drops committed (and accepted) slots it still needs.  Reproduced, not repaired (`far` below;
 * found on the GPU by the whole round across the wrap against the Java reading, round 4). */
__device__ __forceinline__ void acc_gc(const DevState& S, AccView& V, AccState& a, int32_t gcSlot) {
  if (jsub(a.slot, gcSlot) <= 0) gcSlot = jsub(a.slot, 1);
  const int32_t delta = jsub(gcSlot, a.gc);
  /* key - gcSlot (key in [_slot - W, _slot + W)) may overflow: the narrow walks below do not hold */
  const bool far = jsub(a.slot, gcSlot) > INT32_MAX - 2 * S.W;
  if (delta > 0) {
    const int32_t Wm = S.W - 1;
    if (delta >= S.W || far) {
      for (int32_t w = 0; w < S.W; w++) {
        const I4 e = V.rd(w);
        if ((e.w & RF_PRESENT) && jsub(e.x, gcSlot) <= 0) V.wr_flags(w, e.w & ~(int32_t)AF_MASK);
      }
    } else {
      /* live accepted slots are all > a.gc: only (a.gc, gcSlot] can die */
      for (int32_t s = (int32_t)((uint32_t)a.gc + 1u);; s = (int32_t)((uint32_t)s + 1u)) {
        const I4 e = V.rd(s & Wm);
        if ((e.w & RF_PRESENT) && e.x == s) V.wr_flags(s & Wm, e.w & ~(int32_t)AF_MASK);
        if (s == gcSlot) break;
      }
    }
    a.gc = gcSlot;
  }
  if (far && jsub(gcSlot, a.slot) < 0) { /* garbageCollectDecisions(gcSlot): "can only GC executed decisions" */
    for (int32_t w = 0; w < S.W; w++) {
      const I4 e = V.rd(w);
      if ((((uint32_t)e.w >> CF_SHIFT) & RF_PRESENT) && jsub(gcSlot, S.com_ring[(int64_t)w * S.G + V.g].x) > 0)
        V.wr_flags(w, e.w & ~(int32_t)CF_MASK);
    }
  }
}

struct Dec {
  int32_t bnum, bcoord, slot, median;
  bool has_value, stop;
};

/* PaxosAcceptor.reconstructDecision (PaxosAcceptor.java:369-385) for a slot whose ring index holds a
 * commit (the caller has tested the commit's RF_PRESENT bit).
 * Written branch-free on purpose.  The natural nested-if form (return early per failed test,
 * assign *out inside the two succeeding branches) was MISCOMPILED by hipcc (ROCm 7.2, -O3,
 * gfx950) once inlined into the accept kernel: after CFG structurization the median of the
 * "placeholder + matching accept" path was replaced by the failing paths' value (an undefined
 * register, or 0 when *out was pre-zeroed) — the isolated function's LLVM IR was correct, the
 * kernel's ISA was not.  Found by the parity fuzz; the select form below has no merge to get
 * wrong.  The loads are always in bounds (same ring index for both rings). */
__device__ __forceinline__ bool acc_reconstruct(const DevState& S, AccView& V, int32_t g, int32_t slot,
                                                Dec* out) {
  const int32_t w = slot & (S.W - 1);
  const I4 ar = V.rd(w);
  const uint32_t cf = ((uint32_t)ar.w >> CF_SHIFT) & 0xffu;
  const uint32_t af = (uint32_t)ar.w & AF_MASK;
  const I4 cr = S.com_ring[(int64_t)w * S.G + g];
  const bool committed = (cf & RF_PRESENT) && cr.x == slot;
  const bool hasv = (cf & RF_HASVALUE) != 0;
  const bool acc_ok = (af & RF_PRESENT) && ar.x == slot && ballot_cmp(ar.y, ar.z, cr.y, cr.z) == 0;
  Dec d;
  d.bnum = hasv ? cr.y : ar.y;
  d.bcoord = hasv ? cr.z : ar.z;
  d.slot = slot;
  d.median = cr.w; /* the decision keeps the COMMIT's medianCheckpointedSlot (:379-382) */
  d.has_value = true;
  d.stop = ((hasv ? cf : af) & RF_STOP) != 0;
  *out = d;
  return committed && (hasv || acc_ok);
}

/* PaxosInstanceStateMachine.extractExecuteAndCheckpoint (PISM:1619-1701) around
 * PaxosAcceptor.putAndRemoveNextExecutable (PaxosAcceptor.java:325-366) and executed (:462-474).
 * Returns the number of slots executed in order starting at the entry value of a.slot. */
__device__ __forceinline__ int32_t acc_eec(const DevState& S, AccView& V, int32_t g, AccState& a,
                                           const Dec& d) {
  int32_t count = 0;
  const int32_t Wm = S.W - 1;
  const bool from_disk = (S.flags & GPX_F_ACCEPTS_FROM_DISK) != 0;
  while (!a.stopped) {
    acc_gc(S, V, a, d.median);
    /* the usual commit - a decision WITH its value for exactly the next slot - is put into
     * committedRequests and taken out again by the same call (:343-353): nothing needs storing */
    bool direct = false;
    if (jsub(d.slot, a.slot) >= 0) {
      /* don't overwrite an existing decision that has a value (:343-346) */
      const int32_t wd = d.slot & Wm;
      const I4 e = V.rd(wd);
      const uint32_t cf = ((uint32_t)e.w >> CF_SHIFT) & 0xffu;
      bool same = false;
      if (cf & RF_PRESENT) same = S.com_ring[(int64_t)wd * S.G + g].x == d.slot;
      if (!same || !(cf & RF_HASVALUE)) {
        if (d.slot == a.slot && d.has_value) {
          direct = true;
          if (cf) V.wr_flags(wd, e.w & ~(int32_t)CF_MASK); /* what it replaces goes with it */
        } else {
          S.com_ring[(int64_t)wd * S.G + g] = mk4(d.slot, d.bnum, d.bcoord, d.median);
          const uint32_t nf = RF_PRESENT | (d.has_value ? RF_HASVALUE : 0u) | (d.stop ? RF_STOP : 0u);
          V.wr_flags(wd, (int32_t)(((uint32_t)e.w & ~CF_MASK) | (nf << CF_SHIFT)));
        }
      }
    }
    Dec nx = d;
    const int32_t wx = a.slot & Wm;
    if (!direct) {
      /* nothing committed in the next slot's ring entry (the usual end of the loop) */
      const I4 e2 = V.rd(wx);
      if (!(((uint32_t)e2.w >> CF_SHIFT) & RF_PRESENT)) break;
      nx = Dec{0, 0, 0, 0, false, false};
      if (!acc_reconstruct(S, V, g, a.slot, &nx)) break;
      /* committedRequests.remove(_slot) */
      V.wr_flags(wx, V.rd(wx).w & ~(int32_t)CF_MASK);
    }
    /* executed(slot, isStop) */
    a.slot = (int32_t)((uint32_t)a.slot + 1u);
    if (nx.stop) a.stopped = true;
    if (a.stopped)
      for (int32_t w = 0; w < S.W; w++) { /* committedRequests.clear() (:468-469) */
        const I4 e = V.rd(w);
        if ((uint32_t)e.w & CF_MASK) V.wr_flags(w, e.w & ~(int32_t)CF_MASK);
      }
    if (from_disk) {
      /* acceptedProposals.remove(nextExecutable.slot) (:357-359) */
      const I4 e = V.rd(wx);
      if ((e.w & RF_PRESENT) && e.x == nx.slot) V.wr_flags(wx, e.w & ~(int32_t)AF_MASK);
    }
    count++;
    if (nx.stop) break;
  }
  return count;
}

/* acceptor state of one group, fetched ahead of the replay by kernels that know their group early
 * (k_ac_direct: the loads go out together instead of one dependent round trip after the other) */
struct AccPre {
  uint32_t gf;
  int32_t slot, bnum, bcoord, gc;
  int32_t w;  /* ring index of the first record's slot; -2 = nothing was fetched ahead (acc_begin loads) */
  int32_t ex, ey, ez, ew; /* that ring entry */
};
__device__ __forceinline__ AccPre acc_nopre() {
  AccPre P;
  P.gf = 0;
  P.slot = P.bnum = P.bcoord = P.gc = 0;
  P.w = -2;
  P.ex = P.ey = P.ez = P.ew = 0;
  return P;
}
__device__ __forceinline__ void acc_preload(const DevState& S, int32_t g, int32_t first_slot, AccPre& P) {
  P.gf = S.g_flags[g];
  P.slot = S.a_slot[g];
  P.bnum = S.a_bnum[g];
  P.bcoord = S.a_bcoord[g];
  P.gc = S.a_gc[g];
  P.w = first_slot & (S.W - 1);
  const I4 e = S.acc_ring[(int64_t)P.w * S.G + g];
  P.ex = e.x;
  P.ey = e.y;
  P.ez = e.z;
  P.ew = e.w;
}
__device__ __forceinline__ void acc_begin(const DevState& S, int32_t g, const AccPre& pre, uint32_t& gf,
                                          AccState& a, AccView& V) {
  V.init(S, g);
  a.slot = a.bnum = a.bcoord = a.gc = 0;
  a.stopped = false;
  if (pre.w != -2) {
    gf = pre.gf;
    if (gf & GF_EXISTS) {
      a.slot = pre.slot;
      a.bnum = pre.bnum;
      a.bcoord = pre.bcoord;
      a.gc = pre.gc;
      a.stopped = (gf & GF_STOPPED) != 0;
      V.preset(pre.w, mk4(pre.ex, pre.ey, pre.ez, pre.ew));
    }
  } else {
    gf = S.g_flags[g];
    if (gf & GF_EXISTS) {
      a.slot = S.a_slot[g];
      a.bnum = S.a_bnum[g];
      a.bcoord = S.a_bcoord[g];
      a.gc = S.a_gc[g];
      a.stopped = (gf & GF_STOPPED) != 0;
    }
  }
}
__device__ __forceinline__ void acc_store(const DevState& S, int32_t g, uint32_t gf,
                                          const AccState& a, const AccState& a0) {
  /* only what changed: every store instruction is a request the CU has to get out */
  if (a.slot != a0.slot) S.a_slot[g] = a.slot;
  if (a.bnum != a0.bnum) S.a_bnum[g] = a.bnum;
  if (a.bcoord != a0.bcoord) S.a_bcoord[g] = a.bcoord;
  if (a.gc != a0.gc) S.a_gc[g] = a.gc;
  if (a.stopped && !(gf & GF_STOPPED)) S.g_flags[g] = gf | GF_STOPPED;
}

/* PaxosInstanceStateMachine.handleAccept (PISM:1080-1166) */
/* the ACCEPT_REPLY of record ix: four dense columns - or, when r_packed is set, ONE 16-byte row
 * {bnum, bcoord, maxcp, flags} (k_unpack_replies turns the rows into the columns afterwards): in an
 * unordered batch ix is a random position and four 4-byte stores per record cost four random accesses */
__device__ __forceinline__ void put_reply(int32_t ix, int32_t bn, int32_t bc, int32_t mcp, int32_t fl,
                                          int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
                                          int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags,
                                          I4* __restrict__ r_packed) {
  if (r_packed) {
    r_packed[ix] = mk4(bn, bc, mcp, fl);
  } else {
    r_bnum[ix] = bn;
    r_bcoord[ix] = bc;
    r_maxcp[ix] = mcp;
    r_flags[ix] = (uint8_t)fl;
  }
}
template <class IT>
__device__ __forceinline__ void apply_accept_group(
    const DevState& S, const DevScratch& X, int32_t g, IT& it, int32_t* __restrict__ r_bnum,
    int32_t* __restrict__ r_bcoord, int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags,
    uint8_t* __restrict__ status, I4* __restrict__ r_packed = nullptr, const AccPre pre = acc_nopre()) {
  uint32_t gf;
  AccState a;
  AccView V;
  acc_begin(S, g, pre, gf, a, V);
  const bool exists = (gf & GF_EXISTS) != 0;
  const AccState a0 = a;
  const int32_t Wm = S.W - 1;
  unsigned long long n_drop = 0;
  Rec r;
  while (it.next(r)) {
    const int32_t slot = r.a, median = r.b, ix = r.idx;
    const bool stop = (r.c & GPX_A_STOP) != 0;
    if (!exists || a.stopped) {
      put_reply(ix, 0, 0, 0, 0, r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
      status[ix] = exists ? GPX_S_STOPPED : GPX_S_NOGROUP;
      n_drop++;
      continue;
    }
    const int32_t w = slot & Wm;
    /* PValuePacket prev = paxosState.getAccept(accept.slot)  (:1122, before accepting) */
    const I4 ar = V.rd(w);
    const bool live = (ar.w & RF_PRESENT) != 0;
    const bool have_prev = live && ar.x == slot;
    /* PaxosAcceptor.acceptAndUpdateBallot (PaxosAcceptor.java:302-322) */
    const bool ballot_ok = ballot_cmp(r.bnum, r.bcoord, a.bnum, a.bcoord) >= 0;
    const bool will_store = ballot_ok && jsub(slot, a.gc) > 0;
    if (will_store && live && ar.x != slot) {
      put_reply(ix, 0, 0, 0, 0, r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
      status[ix] = GPX_S_WINDOW; /* ring slot held by another live accepted slot */
      n_drop++;
      continue;
    }
    if (ballot_ok) {
      a.bnum = r.bnum;
      a.bcoord = r.bcoord;
      if (will_store)
        V.wr(w, mk4(slot, r.bnum, r.bcoord,
                    (int32_t)(((uint32_t)ar.w & CF_MASK) | RF_PRESENT | (stop ? (uint32_t)RF_STOP : 0u))));
    }
    acc_gc(S, V, a, median);
    /* reply (myID, ballot, slot, getSlot()-1)  (:1139-1143) */
    const int32_t rep_bnum = a.bnum, rep_bcoord = a.bcoord, rep_maxcp = jsub(a.slot, 1);
    /* toLog (:1146-1149) */
    const bool to_log = ballot_cmp(r.bnum, r.bcoord, a.bnum, a.bcoord) >= 0 &&
                        jsub(slot, a.gc) > 0 &&
                        (!have_prev || ballot_cmp(ar.y, ar.z, r.bnum, r.bcoord) < 0);
    put_reply(ix, rep_bnum, rep_bcoord, rep_maxcp, (to_log ? GPX_R_TOLOG : 0) | (will_store ? GPX_R_STORED : 0),
              r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
    /* status[ix] stays GPX_S_OK (prefilled by k_hist) */
    /* might release some meta-commits (:1158-1161) */
    Dec rd = Dec{0, 0, 0, 0, false, false};
    if ((((uint32_t)V.rd(w).w >> CF_SHIFT) & RF_PRESENT) && acc_reconstruct(S, V, g, slot, &rd)) {
      const int32_t first = a.slot;
      const int32_t cnt_exec = acc_eec(S, V, g, a, rd);
      if (cnt_exec > 0) {
        it.emit(0, first, cnt_exec, 0, 1);
      }
    }
  }
  if (exists) acc_store(S, g, gf, a, a0);
  if (n_drop) atomicAdd(&X.counters[2], n_drop);
}

__global__ __launch_bounds__(1024) void k_bucket_accept(
    DevState S, DevScratch X, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, uint8_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  BucketView bv;
  if (*X.unsorted != X.epoch) return; /* ordered batch: k_ac_direct did it; nothing was partitioned */
  if (!bucket_prepare(X, lds, &bv, []() {})) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += (int32_t)blockDim.x) {
    const int32_t c = bv.lcnt[l];
    int32_t nout = 0;
    if (c != 0 && g0 + l < S.G) {
      GroupIter it;
      it.init(bv, l, c);
      apply_accept_group(S, X, g0 + l, it, r_bnum, r_bcoord, r_maxcp, r_flags, status);
      nout = it.nout;
    }
    bv.lcnt[l] = nout; /* the group's record count is no longer needed: now its output count */
  }
  bucket_emit(X, bv);
}

/* PaxosInstanceStateMachine.handleBatchedCommit (PISM:1480-1528) per slot and
 * handleCommittedRequest (:1432-1478) for full decisions */
template <class IT>
__device__ __forceinline__ void apply_commit_group(const DevState& S, const DevScratch& X,
                                                   int32_t g, IT& it,
                                                   uint8_t* __restrict__ status, const AccPre pre = acc_nopre()) {
  uint32_t gf;
  AccState a;
  AccView V;
  acc_begin(S, g, pre, gf, a, V);
  const bool exists = (gf & GF_EXISTS) != 0;
  const AccState a0 = a;
  const int32_t Wm = S.W - 1;
  unsigned long long n_drop = 0;
  Rec r;
  while (it.next(r)) {
    const int32_t slot = r.a, median = r.b, kind = r.c, ix = r.idx;
    if (!exists || a.stopped) {
      status[ix] = exists ? GPX_S_STOPPED : GPX_S_NOGROUP;
      n_drop++;
      continue;
    }
    if (jsub(slot, a.slot) >= S.W) {
      status[ix] = GPX_S_WINDOW; /* further ahead than the committed window */
      n_drop++;
      continue;
    }
    Dec d = Dec{0, 0, 0, 0, false, false};
    if (kind & GPX_C_HASVALUE) {
      d = Dec{r.bnum, r.bcoord, slot, median, true, (kind & GPX_C_STOP) != 0};
    } else {
      /* accept != null && accept.ballot.equals(batchedCommit.ballot) (:1492) */
      const I4 ar = V.rd(slot & Wm);
      if ((ar.w & RF_PRESENT) && ar.x == slot && ballot_cmp(ar.y, ar.z, r.bnum, r.bcoord) == 0)
        d = Dec{ar.y, ar.z, slot, median, true, (ar.w & RF_STOP) != 0};
      else
        d = Dec{r.bnum, r.bcoord, slot, median, false, false}; /* placeholder (:1510-1520) */
    }
    const int32_t first = a.slot;
    const int32_t cnt_exec = acc_eec(S, V, g, a, d);
    if (cnt_exec > 0) {
      it.emit(0, first, cnt_exec, 0, 1);
    }
  }
  if (exists) acc_store(S, g, gf, a, a0);
  if (n_drop) atomicAdd(&X.counters[2], n_drop);
}

__global__ __launch_bounds__(1024) void k_bucket_commit(DevState S, DevScratch X,
                                                        uint8_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  BucketView bv;
  if (*X.unsorted != X.epoch) return; /* ordered batch: k_ac_direct did it; nothing was partitioned */
  if (!bucket_prepare(X, lds, &bv, []() {})) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += (int32_t)blockDim.x) {
    const int32_t c = bv.lcnt[l];
    int32_t nout = 0;
    if (c != 0 && g0 + l < S.G) {
      GroupIter it;
      it.init(bv, l, c);
      apply_commit_group(S, X, g0 + l, it, status);
      nout = it.nout;
    }
    bv.lcnt[l] = nout; /* the group's record count is no longer needed: now its output count */
  }
  bucket_emit(X, bv);
}

/* PaxosInstanceStateMachine.handleProposal (PISM:818-888) ->
 * PaxosCoordinatorState.propose (:233-263) + initCommander (:841-851) */
template <int KMAX>
struct ProposePre {
  uint32_t gf;
  int32_t a_bnum, a_bcoord, my_bnum, my_bcoord, next, pcount;
  int32_t ns[KMAX];
  uint32_t pe_prev, pe_cur; /* myProposals entries of slots next - 1 and next */
};
template <int KMAX>
__device__ __forceinline__ void propose_preload(const DevState& S, int32_t g, ProposePre<KMAX>& P) {
  const int32_t G = S.G;
  P.gf = S.g_flags[g];
  P.a_bnum = S.a_bnum[g];
  P.a_bcoord = S.a_bcoord[g];
  P.my_bnum = S.c_bnum[g];
  P.my_bcoord = S.c_bcoord[g];
  P.next = S.c_next[g];
  P.pcount = S.c_pcount[g];
#pragma unroll
  for (int q = 0; q < KMAX; q++) P.ns[q] = (q < S.kmax) ? S.node_slots[(int64_t)q * G + g] : 0;
}
template <int KMAX>
__device__ __forceinline__ void propose_preload_ring(const DevState& S, int32_t g,
                                                     ProposePre<KMAX>& P) {
  const int32_t Wm = S.W - 1;
  P.pe_prev = S.p_ring[(int64_t)(jsub(P.next, 1) & Wm) * S.G + g];
  P.pe_cur = S.p_ring[(int64_t)(P.next & Wm) * S.G + g];
}

/* the one record of a group in a strictly ascending batch, with GroupIter's reading interface */
struct OneRec {
  int32_t idx, a, c, done;
  __device__ __forceinline__ bool next(Rec& out) {
    if (done) return false;
    out.idx = idx;
    out.a = a;
    done = 1;
    return true;
  }
};

template <int KMAX, class IT>
__device__ __forceinline__ void apply_propose_group(
    const DevState& S, const DevScratch& X, int32_t g, IT& it, int32_t* __restrict__ o_slot,
    int32_t* __restrict__ o_bnum, int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median,
    uint8_t* __restrict__ status, const ProposePre<KMAX>& P, const int64_t* __restrict__ handle) {
  const int32_t G = S.G;
  const uint32_t gf = P.gf;
  /* not active: the proposal is kept (pre-active) but no ACCEPT goes out (PCS:254-261) */
  const bool preparing = (gf & GF_PREPARING) != 0;
  const bool exists = (gf & GF_EXISTS) != 0, stopped = (gf & GF_STOPPED) != 0;
  const int32_t k = (int32_t)GF_K(gf);
  const int32_t a_bnum = exists ? P.a_bnum : 0, a_bcoord = exists ? P.a_bcoord : 0;
  const int32_t my_bnum = exists ? P.my_bnum : 0, my_bcoord = exists ? P.my_bcoord : 0;
  /* PaxosCoordinator.exists(coordinator, paxosState.getBallot()) (PISM:825-826) */
  const bool coord_ok = exists && (gf & GF_HASCOORD) &&
                        ballot_cmp(my_bnum, my_bcoord, a_bnum, a_bcoord) >= 0;
  int32_t next = coord_ok ? P.next : 0;
  const int32_t next0 = next;
  int32_t pcount = coord_ok ? P.pcount : 0;
  int32_t ns[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; q++) ns[q] = (coord_ok && q < k) ? P.ns[q] : 0;
  const int32_t median = coord_ok ? median_minus<KMAX>(ns, k) : 0;
  const int32_t Wm = S.W - 1;
  /* entry of the slot before `next` and of `next` itself: after a successful proposal the entry
   * just written IS the previous one of the following record, so only the new current entry is
   * loaded */
  uint32_t pe_prev = P.pe_prev, pe_cur = P.pe_cur;
  unsigned long long n_drop = 0;
  Rec r;
  while (it.next(r)) {
    const int32_t ix = r.idx;
    const bool stop = r.a != 0;
    /* every record writes its four output words exactly once (a store instruction is a request
     * the CU has to get out: pre-zeroing them cost k_propose_direct four extra stores per record) */
    int32_t st = GPX_S_OK, ob = 0, oc = 0;
    if (!exists || stopped) {
      st = exists ? GPX_S_STOPPED : GPX_S_NOGROUP;
      n_drop++;
    } else if (!coord_ok) {
      /* unicast to paxosState.getBallotCoord() (PISM:854-860) */
      ob = a_bnum;
      oc = a_bcoord;
      st = GPX_S_FORWARD;
    } else if ((pe_prev & PR_PRESENT) && (pe_prev & PR_STOP)) {
      /* no point enqueuing anything after stop (PaxosCoordinatorState.java:235-239) */
      st = GPX_S_REFUSED;
    } else if (pe_cur & PR_PRESENT) {
      st = GPX_S_WINDOW; /* slot next-W still outstanding */
      n_drop++;
    }
    if (st != GPX_S_OK) {
      o_slot[ix] = 0;
      o_bnum[ix] = ob;
      o_bcoord[ix] = oc;
      o_median[ix] = 0;
      status[ix] = (uint8_t)st;
      continue;
    }
    const uint32_t e = PR_PRESENT | (stop ? PR_STOP : 0u);
    S.p_ring[(int64_t)(next & Wm) * G + g] = e;
    pcount++;
    o_slot[ix] = next;
    o_bnum[ix] = my_bnum;
    o_bcoord[ix] = my_bcoord;
    o_median[ix] = preparing ? 0 : median; /* getMajorityCommittedSlot: nodeSlots unchanged by propose */
    if (preparing) {
      S.p_handle[(int64_t)(next & Wm) * G + g] = handle ? handle[ix] : 0;
      status[ix] = GPX_S_PREACTIVE; /* over k_hist's GPX_S_OK */
    }
    next = (int32_t)((uint32_t)next + 1u);
    pe_prev = e;
    if (it.done < it.c) pe_cur = S.p_ring[(int64_t)(next & Wm) * G + g];
  }
  if (coord_ok && next != next0) {
    S.c_next[g] = next;
    S.c_pcount[g] = pcount;
  }
  if (n_drop) atomicAdd(&X.counters[2], n_drop);
}

template <int KMAX>
__global__ __launch_bounds__(1024) void k_bucket_propose(
    DevState S, DevScratch X, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status,
    const int64_t* __restrict__ handle) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  BucketView bv;
  if (*X.unsorted != X.epoch) { /* handled by k_propose_direct; leave the counts ready for the next call */
    if (threadIdx.x == 0) X.bucket_tot[blockIdx.x] = 0;
    return;
  }
  const int32_t g0 = blockIdx.x << X.shift;
  const int32_t nb_ = X.bucket_off[blockIdx.x + 1] - X.bucket_off[blockIdx.x];
  const bool pre = (int32_t)blockDim.x == X.gb && 2 * nb_ >= X.gb;
  const int32_t gme = g0 + (int32_t)threadIdx.x;
  ProposePre<KMAX> P;
  if (pre && gme < S.G) propose_preload<KMAX>(S, gme, P);
  if (!bucket_prepare(X, lds, &bv, [&]() {
        if (pre && gme < S.G) propose_preload_ring<KMAX>(S, gme, P);
      }))
    return;
  for (int32_t l = threadIdx.x; l < X.gb; l += (int32_t)blockDim.x) {
    const int32_t c = bv.lcnt[l];
    if (c == 0 || g0 + l >= S.G) continue;
    GroupIter it;
    it.init(bv, l, c);
    if (!pre) {
      propose_preload<KMAX>(S, g0 + l, P);
      propose_preload_ring<KMAX>(S, g0 + l, P);
    }
    apply_propose_group<KMAX, GroupIter>(S, X, g0 + l, it, o_slot, o_bnum, o_bcoord, o_median, status, P, handle);
  }
}


/* Order check of a batch: is the gidx column in range and strictly ascending (STRICT: every group at
 * most once - proposals) or non-decreasing (the records of a group adjacent and groups ascending -
 * what the previous stage of the pipeline emits: decisions leave grouped by gidx, ACCEPTs follow the
 * proposal batch)?  Such a batch needs no partition.  Reads the column once (16-byte loads when
 * aligned), pre-fills the status column and raises the call's epoch in *X.unsorted otherwise. */
#define GPX_OC_BLOCK 256
template <bool STRICT>
__global__ __launch_bounds__(GPX_OC_BLOCK) void k_order_check(int32_t n, const int32_t* __restrict__ gidx,
                                                           int32_t G, DevScratch X,
                                                           uint8_t* __restrict__ status,
                                                           int32_t* __restrict__ zero, int32_t nzero) {
  const int64_t i0 = ((int64_t)blockIdx.x * GPX_OC_BLOCK + threadIdx.x) * 8;
  if (zero && i0 / 8 < nzero) zero[i0 / 8] = 0; /* nzero <= ceil(n / 8): the grid covers it */
  bool bad = false;
  if (i0 < n) {
    int32_t g[9];
    const bool full = i0 + 8 < n; /* all eight and the successor of the last one exist */
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
        bad |= oob || (i0 + q + 1 < n && (STRICT ? g[q] >= g[q + 1] : g[q] > g[q + 1]));
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
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicMax(X.unsorted, X.epoch);
}

/* Proposal batch whose gidx column is strictly ascending (every group at most once - what
 * RequestBatcher produces: one batched request per group per dequeue, RequestBatcher.java:79-81):
 * no regrouping needed, one lane per record applies it straight to its group.  Consecutive
 * records address ascending groups, so the state accesses are coalesced when the batch is dense. */
template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_propose_direct(
    DevState S, DevScratch X, int32_t n, const int32_t* __restrict__ gidx,
    const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status,
    const int64_t* __restrict__ handle, int32_t refuse) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (*X.unsorted == X.epoch) {
    /* not strictly ascending: the partition path applies it - or, under the caller's
     * GPX_ORDERED_PROPOSE promise (no partition path launched), the batch is refused whole */
    if (refuse && i < n) {
      o_slot[i] = 0;
      o_bnum[i] = 0;
      o_bcoord[i] = 0;
      o_median[i] = 0;
      status[i] = GPX_S_UNORDERED;
    }
    return;
  }
  if (i >= n) return;
  const int32_t g = gidx[i]; /* in range: k_order_check marks out-of-range batches unordered */
  ProposePre<KMAX> P;
  propose_preload<KMAX>(S, g, P);
  propose_preload_ring<KMAX>(S, g, P);
  OneRec it;
  it.idx = i;
  it.a = is_stop ? (int32_t)(is_stop[i] & 1) : 0;
  it.c = 1;
  it.done = 0;
  apply_propose_group<KMAX, OneRec>(S, X, g, it, o_slot, o_bnum, o_bcoord, o_median, status, P, handle);
}

/* Proposal batch of at most 65,536 requests in ONE launch: every workgroup judges the order of the whole
 * (L2-resident) gidx column itself, then applies its records like k_propose_direct - no k_order_check
 * launch.  Not strictly ascending: *X.unsorted is raised for the partition path launched behind it, or
 * (GPX_ORDERED_PROPOSE promise) the batch is refused whole. */
/* is the whole gidx column in range and ascending?  Every lane of every workgroup ends up with the same
 * verdict (the column is read by all of them: at most 256 KB, L2-resident) */
/* first index that breaks the order - out of range, or a descent (STRICT: a non-ascent) INTO it - or 0xffffffff */
template <bool STRICT>
__device__ __forceinline__ uint32_t small_batch_first_bad(int32_t n, const int32_t* __restrict__ gidx, int32_t G) {
  __shared__ uint32_t s_first_bad;
  if (threadIdx.x == 0) s_first_bad = 0xffffffffu;
  uint32_t mine = 0xffffffffu;
  /* sixteen entries and their predecessors in flight per lane and round, none of them behind a branch (an index
   * behind the batch's end is clamped for the load): a column of 16,384 entries is one round trip for a workgroup of
   * 1,024 lanes (round 4; before: four entries per lane and round, three rounds for 10,000 entries) */
  const int32_t B = (int32_t)blockDim.x, t = (int32_t)threadIdx.x;
  for (int32_t r0 = 0; r0 < n; r0 += 16 * B) {
    int32_t gg[16], gp[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int32_t ic = min(r0 + k * B + t, n - 1);
      gg[k] = gidx[ic];
      gp[k] = gidx[max(ic - 1, 0)];
    }
#pragma unroll
    for (int k = 15; k >= 0; k--) {
      const int32_t i = r0 + k * B + t;
      const bool viol = i < n && ((uint32_t)gg[k] >= (uint32_t)G || (i > 0 && (STRICT ? gp[k] >= gg[k] : gp[k] > gg[k])));
      mine = viol ? min(mine, (uint32_t)i) : mine;
    }
  }
  if (__syncthreads_or(mine != 0xffffffffu)) {
    if (mine != 0xffffffffu) atomicMin(&s_first_bad, mine);
    __syncthreads();
  }
  return s_first_bad;
}
template <bool STRICT>
__device__ __forceinline__ bool small_batch_ordered(int32_t n, const int32_t* __restrict__ gidx, int32_t G) {
  return small_batch_first_bad<STRICT>(n, gidx, G) == 0xffffffffu;
}

template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_propose_small(
    DevState S, DevScratch X, int32_t n, const int32_t* __restrict__ gidx,
    const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status,
    const int64_t* __restrict__ handle, int32_t refuse) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  /* the record's group state is requested before the verdict is worked out, as in k_propose_one (a refused record
   * has loaded in vain) */
  const int32_t g = i < n ? gidx[i] : -1;
  ProposePre<KMAX> P;
  if ((uint32_t)g < (uint32_t)S.G) {
    propose_preload<KMAX>(S, g, P);
    propose_preload_ring<KMAX>(S, g, P);
  }
  const uint32_t first_bad = small_batch_first_bad<true>(n, gidx, S.G);
  if (first_bad != 0xffffffffu && !refuse) { /* no promise: the partition path launched behind takes the whole batch */
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(X.unsorted, X.epoch);
    if (i < n) status[i] = (uint32_t)g < (uint32_t)S.G ? GPX_S_OK : GPX_S_NOGROUP; /* what k_order_check leaves */
    return;
  }
  if (i >= n) return;
  if ((uint32_t)i >= first_bad) { /* GPX_ORDERED_PROPOSE broken: refused from the first violation on (gpx.h) */
    o_slot[i] = 0;
    o_bnum[i] = 0;
    o_bcoord[i] = 0;
    o_median[i] = 0;
    status[i] = GPX_S_UNORDERED;
    return;
  }
  status[i] = GPX_S_OK;
  OneRec it;
  it.idx = i;
  it.a = is_stop ? (int32_t)(is_stop[i] & 1) : 0;
  it.c = 1;
  it.done = 0;
  apply_propose_group<KMAX, OneRec>(S, X, g, it, o_slot, o_bnum, o_bcoord, o_median, status, P, handle);
}

/* ------------------------------------------------------------------------- */
/* view change, acceptor side                                                   */
/* PISM.handlePrepare (PISM:900-1006) -> PaxosAcceptor.handlePrepare (PaxosAcceptor.java:239-273).
 * Record payload (k_scatter_ac): a = firstUndecidedSlot, bnum / bcoord = the prepare's ballot. */
struct PrepOut {
  int32_t n;
  int32_t *r_bnum, *r_bcoord, *r_gc;
  uint8_t* r_flags;
  unsigned long long* p_mask;
  int32_t *p_slot, *p_bnum, *p_bcoord; /* [W][n] */
  uint8_t* status;
};
__global__ __launch_bounds__(1024) void k_bucket_prepare(DevState S, DevScratch X, PrepOut O) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  BucketView bv;
  if (!bucket_prepare(X, lds, &bv, []() {})) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += (int32_t)blockDim.x) {
    const int32_t c = bv.lcnt[l];
    const int32_t g = g0 + l;
    if (c == 0 || g >= S.G) continue;
    GroupIter it;
    it.init(bv, l, c);
    const uint32_t gf = S.g_flags[g];
    const bool exists = (gf & GF_EXISTS) != 0, stopped = (gf & GF_STOPPED) != 0;
    int32_t bn = exists ? S.a_bnum[g] : 0, bc = exists ? S.a_bcoord[g] : 0;
    const int32_t bn0 = bn, bc0 = bc;
    const int32_t gc = exists ? S.a_gc[g] : 0;
    unsigned long long n_drop = 0;
    Rec r;
    while (it.next(r)) {
      const int32_t ix = r.idx, first = r.a;
      O.r_bnum[ix] = 0;
      O.r_bcoord[ix] = 0;
      O.r_gc[ix] = 0;
      O.r_flags[ix] = 0;
      O.p_mask[ix] = 0;
      if (!exists || stopped) { /* isStopped() -> null (PaxosAcceptor.java:241-242) */
        O.status[ix] = exists ? GPX_S_STOPPED : GPX_S_NOGROUP;
        n_drop++;
        continue;
      }
      const int32_t pn = bn, pc = bc; /* prevBallot (PISM:903) */
      if (ballot_cmp(r.bnum, r.bcoord, bn, bc) > 0) { /* strictly greater: adopt (:246-252) */
        bn = r.bnum;
        bc = r.bcoord;
      }
      const bool nack = ballot_cmp(bn, bc, r.bnum, r.bcoord) > 0;
      unsigned long long mask = 0;
      if (!nack) {
        /* pruneAcceptedProposals: keep slot - firstUndecidedSlot >= 0 (:283-293) */
        for (int32_t w = 0; w < S.W; w++) {
          const int64_t o = (int64_t)w * S.G + g;
          const I4 a = S.acc_ring[o];
          if (!(a.w & RF_PRESENT)) continue;
          if (jsub(a.x, first) < 0) continue;
          mask |= 1ull << w;
          const int64_t q = (int64_t)w * O.n + ix;
          O.p_slot[q] = a.x;
          O.p_bnum[q] = a.y;
          O.p_bcoord[q] = a.z;
        }
      }
      O.r_bnum[ix] = bn;
      O.r_bcoord[ix] = bc;
      /* getMaxGCSlotFirstUndecidedSlot (:275-280) */
      const int32_t fm1 = jsub(first, 1);
      O.r_gc[ix] = jsub(gc, fm1) < 0 ? fm1 : gc;
      O.r_flags[ix] = (uint8_t)((nack ? GPX_P_NACK : 0) | (ballot_cmp(pn, pc, bn, bc) < 0 ? GPX_P_TOLOG : 0));
      O.p_mask[ix] = mask;
      /* status stays GPX_S_OK (k_hist) */
    }
    if (exists && (bn != bn0 || bc != bc0)) {
      S.a_bnum[g] = bn;
      S.a_bcoord[g] = bc;
    }
    if (n_drop) atomicAdd(&X.counters[2], n_drop);
  }
}

/* ------------------------------------------------------------------------- */
/* RequestBatcher (RequestBatcher.java:111-239)                                  */
/* The batcher keeps one FIFO per paxosID; every dequeue takes the head of a queue and latches the
 * following requests of that group onto it while the byte and batch-size limits hold
 * (dequeueImpl :163-239).  For a whole burst that is a regrouping by group plus, per group, a greedy
 * split of the FIFO into consecutive batches - one lane per group again.  Record payload (via
 * k_scatter_ac): a = lengthEstimate, b = batchSize() + 1, c = stop flag.  Dense output
 * leader[i] = the request whose batch request i was latched onto (itself for a batch head);
 * compacted rows, group-major: one per batch = one proposal. */
__global__ __launch_bounds__(1024) void k_bucket_reqbatch(DevState S, DevScratch X, int32_t max_bytes,
                                                          int32_t max_size,
                                                          int32_t* __restrict__ leader) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  BucketView bv;
  if (!bucket_prepare(X, lds, &bv, []() {})) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += (int32_t)blockDim.x) {
    const int32_t c = bv.lcnt[l];
    int32_t nout = 0;
    if (c != 0 && g0 + l < S.G) {
      GroupIter it;
      it.init(bv, l, c);
      int32_t head = -1, count = 0, stop = 0;
      uint32_t head_cur = 0;
      long long bytes = 0, size = 0;
      Rec r;
      while (it.next(r)) {
        const long long est = r.a, w = r.b;
        bool fresh = head < 0;
        if (!fresh) {
          /* ((totalByteLength += next.lengthEstimate()) > limit) ||
           * ((totalBatchSize += next.batchSize() + 1) > MAX_BATCH_SIZE) -> break   (:205-211) */
          if (bytes + est > max_bytes || size + w > max_size) {
            it.emit_at(head_cur, head, count, (int32_t)bytes, (int32_t)size, stop);
            fresh = true;
          }
        }
        if (fresh) { /* RequestPacket first = reqPktIter.next()  (:175-177) */
          head = r.idx;
          head_cur = it.cur;
          count = 1;
          bytes = est;
          size = w;
          stop = r.c & 1;
        } else { /* batch.add(next) -> first.latchToBatch(...)  (:213-219) */
          count++;
          bytes += est;
          size += w;
          stop |= r.c & 1;
        }
        leader[r.idx] = head;
      }
      if (head >= 0) it.emit_at(head_cur, head, count, (int32_t)bytes, (int32_t)size, stop);
      nout = it.nout;
    }
    bv.lcnt[l] = nout;
  }
  bucket_emit(X, bv);
}

/* dense prefill of the leader column for records the partition drops (gidx out of range) */
__global__ __launch_bounds__(GPX_BLOCK) void k_fill_i32(int32_t n, int32_t v, int32_t* __restrict__ a) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i < n) a[i] = v;
}

/* ------------------------------------------------------------------------- */
/* PISM.checkRunForCoordinator's decision (PISM:2090-2176), one lane per group    */
#define GPX_MAX_NODE_LIST 16
struct NodeLists {
  int32_t n_down, n_long;
  int32_t down[GPX_MAX_NODE_LIST], longdead[GPX_MAX_NODE_LIST];
};
__global__ __launch_bounds__(GPX_BLOCK) void k_election_scan(DevState S, int32_t n,
                                                            const int32_t* __restrict__ gidx,
                                                            NodeLists L, int32_t force,
                                                            uint8_t* __restrict__ run,
                                                            int32_t* __restrict__ p_bnum,
                                                            int32_t* __restrict__ p_first,
                                                            uint8_t* __restrict__ status) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx ? gidx[i] : i;
  run[i] = GPX_RUN_NO;
  p_bnum[i] = 0;
  p_first[i] = 0;
  if ((uint32_t)g >= (uint32_t)S.G || !(S.g_flags[g] & GF_EXISTS)) {
    status[i] = GPX_S_NOGROUP;
    return;
  }
  status[i] = GPX_S_OK;
  const uint32_t gf = S.g_flags[g];
  const int32_t k = (int32_t)GF_K(gf);
  const int32_t bn = S.a_bnum[g], bc = S.a_bcoord[g]; /* curBallot = paxosState.getBallot() */
  /* PaxosCoordinator.exists(coordinator, curBallot): coordinator != null && its ballot >= curBallot */
  const bool have = (gf & GF_HASCOORD) && ballot_cmp(S.c_bnum[g], S.c_bcoord[g], bn, bc) >= 0;
  bool down = false, longdead = false;
  for (int32_t q = 0; q < L.n_down; q++) down |= L.down[q] == bc;
  for (int32_t q = 0; q < L.n_long; q++) longdead |= L.longdead[q] == bc;
  /* getNextCoordinator: the member after the coordinator, members ascending, wrapping (:2231-2240) */
  int32_t next = INT32_MIN;
  bool member = false;
  for (int32_t q = 0; q < k; q++)
    if (S.members[(int64_t)q * S.G + g] == bc) {
      next = S.members[(int64_t)((q + 1) % k) * S.G + g];
      member = true;
    }
  int32_t why = GPX_RUN_NO;
  if (!have) {
    if (bc == S.my_id)
      why = GPX_RUN_MINE;
    else if (down && member && next == S.my_id)
      why = GPX_RUN_NEXT;
    else if (down && longdead)
      why = GPX_RUN_LONGDEAD;
  }
  if (why == GPX_RUN_NO && force) why = GPX_RUN_FORCED;
  if (why != GPX_RUN_NO) {
    run[i] = (uint8_t)why;
    p_bnum[i] = (int32_t)((uint32_t)bn + 1u); /* new Ballot(curBallot.ballotNumber + 1, myID) */
    p_first[i] = S.a_slot[g];                 /* new PreparePacket(newBallot, paxosState.getSlot()) */
  }
}

/* ------------------------------------------------------------------------- */
/* gap detection (PaxosAcceptor.getMissingCommittedSlots / getMaxCommittedSlot,  */
/* PaxosAcceptor.java:405-438; PISM.shouldSync, PISM:2341-2364)                  */
/* One lane per listed group.  The committed window holds slots in [_slot, _slot + W), so the
 * missing set fits a 64-bit mask relative to _slot. */
__global__ __launch_bounds__(GPX_BLOCK) void k_gap_scan(DevState S, int32_t n,
                                                       const int32_t* __restrict__ gidx,
                                                       int32_t threshold, int32_t sync_mode,
                                                       int32_t size_limit,
                                                       int32_t* __restrict__ first_slot,
                                                       int32_t* __restrict__ max_committed,
                                                       unsigned long long* __restrict__ missing,
                                                       uint8_t* __restrict__ should_sync,
                                                       uint8_t* __restrict__ status) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx[i];
  first_slot[i] = 0;
  max_committed[i] = 0;
  missing[i] = 0;
  should_sync[i] = 0;
  if ((uint32_t)g >= (uint32_t)S.G || !(S.g_flags[g] & GF_EXISTS)) {
    status[i] = GPX_S_NOGROUP;
    return;
  }
  const uint32_t gf = S.g_flags[g];
  const int32_t slot = S.a_slot[g];
  first_slot[i] = slot;
  const bool stopped = (gf & GF_STOPPED) != 0;
  /* getMaxCommittedSlot (PaxosAcceptor.java:425-438): stopped or empty -> getSlot() - 1; else
   * committedRequests.lastKey() - the largest key in SIGNED order (a TreeMap<Integer, ..>) - and only when that key
   * is Integer.MAX_VALUE itself does the Java walk the keys with the wraparound compare.  Keys on both sides of the
   * wrap without MAX_VALUE among them therefore give the key BEFORE the wrap: reproduced, not repaired. */
  int32_t maxc = jsub(slot, 1);
  if (!stopped) {
    bool any = false;
    int32_t last = INT32_MIN, wrapmax = maxc;
    for (int32_t w = 0; w < S.W; w++) {
      const int64_t o = (int64_t)w * S.G + g;
      if (!(((uint32_t)S.acc_ring[o].w >> CF_SHIFT) & RF_PRESENT)) continue;
      const int32_t key = S.com_ring[o].x;
      any = true;
      last = key > last ? key : last;
    }
    if (any && last == INT32_MAX) {
      /* for (int i : keySet()) if (i - maxSlot > 0) maxSlot = i;  keys ascending in signed order: with at most W
       * keys inside one window of W slots the walk ends at the wraparound-aware maximum whatever the order */
      for (int32_t w = 0; w < S.W; w++) {
        const int64_t o = (int64_t)w * S.G + g;
        if ((((uint32_t)S.acc_ring[o].w >> CF_SHIFT) & RF_PRESENT) && jsub(S.com_ring[o].x, wrapmax) > 0) wrapmax = S.com_ring[o].x;
      }
      last = wrapmax;
    }
    if (any) maxc = last;
  }
  max_committed[i] = maxc;
  /* shouldSync (PISM:2341-2364), DISABLE_SYNC_DECISIONS = false */
  const int32_t gap = jsub(maxc, slot);
  const bool nontrivial = gap >= threshold / 100;       /* NONTRIVIAL_GAP_FACTOR */
  const bool small_thr = threshold <= 1;                /* INITIAL_SYNC_THRESHOLD */
  const bool sync = (gap >= threshold) || ((slot == 0 || slot == 1) && (nontrivial || small_thr)) ||
                    (nontrivial && sync_mode == GPX_SYNC_TO_PAUSE) || sync_mode == GPX_SYNC_FORCE;
  should_sync[i] = sync ? 1 : 0;
  status[i] = stopped ? GPX_S_STOPPED : GPX_S_OK;
  if (stopped) return; /* getMissingCommittedSlots returns null */
  /* missing: no commit, or a meta-commit without its accept (:414-420) */
  unsigned long long m = 0;
  const int32_t limit = (int32_t)((uint32_t)slot + (uint32_t)size_limit);
  const int32_t Wm = S.W - 1;
  int32_t j = 0;
  for (int32_t s = slot; jsub(s, maxc) < 0 && jsub(s, limit) < 0 && j < 64;
       s = (int32_t)((uint32_t)s + 1u), j++) {
    const int64_t o = (int64_t)(s & Wm) * S.G + g;
    const I4 ae = S.acc_ring[o];
    const uint32_t cf = ((uint32_t)ae.w >> CF_SHIFT) & 0xffu;
    const bool have = (cf & RF_PRESENT) && S.com_ring[o].x == s;
    const bool acc = (ae.w & RF_PRESENT) && ae.x == s;
    if (!have || (!(cf & RF_HASVALUE) && !acc)) m |= 1ull << j;
  }
  missing[i] = m;
}

/* ------------------------------------------------------------------------- */
/* lifecycle                                                                    */
/* name rows and table entries of the wire codec (gpx_wire.hip.h: NM_* / NameEnt, checked there): a named
 * group's (exists, version) live in its row AND in the table entry the frames' lookups read */
#define GPX_NAME_ROW_STRIDE 160
#define GPX_NAME_ROW_EXISTS 5
#define GPX_NAME_ROW_VERSION 8
#define GPX_NAME_ROW_SLOT 144
#define GPX_NAME_BUCKET_BYTES 128 /* four entries: 32 bytes of keys, four payloads of 24 bytes */
#define GPX_NAME_BUCKET_KEYS 32
#define GPX_NAME_PAYLOAD_BYTES 24
struct NameCopies {
  uint8_t* rows; /* null: no wire codec in use */
  uint8_t* tab;
  __device__ __forceinline__ void set(int32_t g, bool exists, bool set_version, int32_t version) const {
    if (!rows) return;
    uint8_t* nr = rows + (int64_t)g * GPX_NAME_ROW_STRIDE;
    if (set_version) *(int32_t*)(nr + GPX_NAME_ROW_VERSION) = version;
    nr[GPX_NAME_ROW_EXISTS] = exists ? 1 : 0;
    if (nr[4] != 0) { /* the name is bound: its table entry carries the copies too */
      const int32_t s = *(const int32_t*)(nr + GPX_NAME_ROW_SLOT);
      if (s >= 0) {
        uint8_t* e = tab + (int64_t)(s >> 2) * GPX_NAME_BUCKET_BYTES + GPX_NAME_BUCKET_KEYS + (s & 3) * GPX_NAME_PAYLOAD_BYTES;
        if (set_version) *(int32_t*)(e + 4) = version;
        e[1] = exists ? 1 : 0; /* payload = {length | exists << 8, version, name[4]} */
      }
    }
  }
};

/* PaxosInstanceStateMachine.hotRestore (PISM:677-690), PaxosAcceptor.hotRestore
 * (PaxosAcceptor.java:128-134), PaxosCoordinator.hotRestore (PaxosCoordinator.java:122-131) */
__global__ __launch_bounds__(GPX_BLOCK) void k_group_create(DevState S, int32_t n,
                                                           const int32_t* __restrict__ gidx,
                                                           const int32_t* __restrict__ members,
                                                           const uint8_t* __restrict__ kk,
                                                           const gpx_hri* __restrict__ rows,
                                                           uint8_t* __restrict__ status,
                                                           NameCopies names) {
  int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx[i];
  const int32_t k = kk[i];
  if ((uint32_t)g >= (uint32_t)S.G || k < 1 || k > S.kmax) {
    status[i] = GPX_S_NOGROUP;
    return;
  }
  if (S.g_flags[g] & GF_EXISTS) {
    status[i] = GPX_S_EXISTS;
    return;
  }
  const gpx_hri r = rows[i];
  const bool coord = r.has_coord && r.coord_bcoord == S.my_id;
  S.g_version[g] = r.version;
  S.a_slot[g] = r.acc_slot;
  S.a_bnum[g] = r.acc_bnum;
  S.a_bcoord[g] = r.acc_bcoord;
  S.a_gc[g] = r.acc_gc_slot;
  S.c_bnum[g] = coord ? r.coord_bnum : 0;
  S.c_bcoord[g] = coord ? r.coord_bcoord : 0;
  S.c_next[g] = coord ? r.next_proposal_slot : 0;
  S.c_pcount[g] = 0;
  for (int32_t j = 0; j < S.kmax; j++) {
    S.members[(int64_t)j * S.G + g] = (j < k) ? members[(int64_t)i * S.kmax + j] : 0;
    S.node_slots[(int64_t)j * S.G + g] = (coord && j < k) ? r.node_slots[j] : 0;
  }
  for (int32_t w = 0; w < S.W; w++) {
    const int64_t o = (int64_t)w * S.G + g;
    S.p_ring[o] = 0;
    S.acc_ring[o] = mk4(0, 0, 0, 0);
  }
  S.g_flags[g] = GF_EXISTS | (coord ? GF_HASCOORD : 0u) | ((uint32_t)k << 8);
  names.set(g, true, true, r.version); /* the wire codec's copies of (exists, version) (gpx_wire.hip.h) */
  status[i] = GPX_S_OK;
}

/* HotRestoreInfo of a live group (PISM.tryPause :2011-2020) */
__device__ __forceinline__ void fill_hri_dev(const DevState& S, int32_t g, uint32_t gf,
                                             gpx_hri* out) {
  gpx_hri r;
  r.version = S.g_version[g];
  r.acc_slot = S.a_slot[g];
  r.acc_bnum = S.a_bnum[g];
  r.acc_bcoord = S.a_bcoord[g];
  r.acc_gc_slot = S.a_gc[g];
  /* getNextProposalSlotIfActive / getNodeSlots / getBallot: only of an ACTIVE coordinator
   * (PaxosCoordinator.java:375-402) */
  const bool coord = (gf & GF_HASCOORD) != 0 && !(gf & GF_PREPARING);
  const int32_t k = (int32_t)GF_K(gf);
  r.has_coord = coord ? 1 : 0;
  r.coord_bnum = coord ? S.c_bnum[g] : 0;
  r.coord_bcoord = coord ? S.c_bcoord[g] : 0;
  r.next_proposal_slot = coord ? S.c_next[g] : -1;
  for (int32_t j = 0; j < GPX_KMAX_LIMIT; j++)
    r.node_slots[j] = (coord && j < k && j < S.kmax) ? S.node_slots[(int64_t)j * S.G + g] : 0;
  *out = r;
}

/* mode: 0 pause (tryPause, only if caught up), 1 kill, 2 snapshot (read only) */
__global__ __launch_bounds__(GPX_BLOCK) void k_group_retire(DevState S, int32_t n,
                                                           const int32_t* __restrict__ gidx,
                                                           int32_t mode, gpx_hri* __restrict__ rows,
                                                           uint8_t* __restrict__ status,
                                                           NameCopies names) {
  int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx[i];
  gpx_hri zero = {};
  if (rows) rows[i] = zero;
  if ((uint32_t)g >= (uint32_t)S.G || !(S.g_flags[g] & GF_EXISTS)) {
    if (status) status[i] = GPX_S_NOGROUP;
    return;
  }
  const uint32_t gf = S.g_flags[g];
  if (mode == GPX_RETIRE_PAUSE) {
    /* PaxosAcceptor.caughtUp (PaxosAcceptor.java:451-459) && PaxosCoordinator.caughtUp */
    bool caught = true;
    const bool from_disk = (S.flags & GPX_F_ACCEPTS_FROM_DISK) != 0;
    for (int32_t w = 0; w < S.W; w++) {
      const int64_t o = (int64_t)w * S.G + g;
      const int32_t fl = S.acc_ring[o].w;
      if (((uint32_t)fl >> CF_SHIFT) & RF_PRESENT) caught = false;
      if (!from_disk && (fl & RF_PRESENT)) caught = false;
    }
    if ((gf & GF_HASCOORD) && S.c_pcount[g] != 0) caught = false;
    if (!caught) {
      if (status) status[i] = GPX_S_BUSY;
      return;
    }
  }
  if (rows) fill_hri_dev(S, g, gf, &rows[i]);
  if (mode != 2) {
    S.g_flags[g] = 0;
    names.set(g, false, false, 0);
  }
  if (status) status[i] = GPX_S_OK;
}
