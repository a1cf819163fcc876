/*
 * gpx_kernels.hip.h — CDNA4 (gfx950) kernels of the batched consensus engine.
 *
 * Integer / indexing work only: HBM-bound, no MFMA.  Wave = 64 lanes everywhere.
 *
 * Pipeline of every batch call.  Groups are binned into BUCKETS of
 * GB = 2^shift consecutive group indices; a batch is partitioned by bucket with LDS histograms
 * (no global atomics anywhere on the data path), then ONE WORKGROUP PER BUCKET regroups its
 * records by group in LDS-counted order and ONE LANE PER GROUP replays them in arrival order:
 *
 *   k_hist          one workgroup per 4096-record tile: LDS histogram over buckets -> tile_hist
 *   k_colscan       per bucket: exclusive scan of tile_hist down the tiles (in chunks)
 *   k_bucket_offs   per bucket: scan of the chunk sums; exclusive scan over buckets -> bucket_off
 *   k_scatter_*     per tile: record i is packed into a 32-byte Rec and written to its bucket's
 *                   region (LDS cursor per bucket) — bucket-contiguous, any order inside
 *   k_bucket_*      per bucket: LDS count per local group -> scan -> perm (records of one group
 *                   contiguous); long segments (> 16) get a cooperative arrival-order sort;
 *                   then one lane per group loads its SoA state (coalesced: consecutive lanes own
 *                   consecutive groups), replays the group's records in ARRIVAL ORDER exactly as
 *                   PaxosInstanceStateMachine.handlePaxosMessage would, once per record, and
 *                   writes per-record dense outputs at the record's arrival index
 *   compact         scan of the per-record output flags + LDS-staged ordered gather, so
 *                   decisions / exec runs leave in the arrival order of the record that made them
 *
 * Each device function cites the reference method it implements (paths relative to
 * /root/reference/src/edu/umass/cs/gigapaxos/).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpx.h"

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
  int32_t a, b, c, idx, bnum, bcoord, lg, pad;
};

#define GPX_BLOCK 256
#define GPX_TILE 4096 /* records per k_hist / k_scatter workgroup */
#define GPX_TILE_ITEMS (GPX_TILE / GPX_BLOCK)
#define GPX_SCAN_ITEMS 8 /* items per thread in the output-flag scan kernels */
#define GPX_SCAN_TILE (GPX_BLOCK * GPX_SCAN_ITEMS)
#define GPX_SMALL_SEG 16  /* segments up to this long are ordered by per-lane min-scan */
#define GPX_MIN_SHIFT 8   /* >= 256 groups per bucket */
#define GPX_LDS_RECS 1024 /* a bucket with at most this many records is regrouped entirely in LDS */
#define GPX_MAX_BUCKETS 4096

/* group flag word */
#define GF_EXISTS 1u
#define GF_STOPPED 2u  /* PaxosAcceptor.STATES.STOPPED */
#define GF_HASCOORD 4u /* PaxosInstanceStateMachine.coordinator != null */
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
  I4* acc_ring;                                   /* [W][G] acceptedProposals {slot,bnum,bcoord,-} */
  uint8_t* acc_flags;                             /* [W][G] */
  I4* com_ring;                                   /* [W][G] committedRequests {slot,bnum,bcoord,median} */
  uint8_t* com_flags;                             /* [W][G] */
};

struct DevScratch {
  int32_t shift, nbk, gb; /* bucket = gidx >> shift; nbk buckets of gb = 1 << shift groups */
  int32_t* tile_hist;     /* [ntiles][nbk] per-tile bucket histogram -> in-chunk exclusive prefix */
  int32_t* chunk_part;    /* [nchunks][nbk] chunk sums -> exclusive prefix over chunks */
  int32_t* bucket_off;    /* [nbk + 1] */
  Rec* rec;               /* [n] bucket-partitioned records */
  int32_t* rank2;         /* [n] rank of a record among its group's records (LDS atomic order) */
  int32_t* perm;          /* [n] per bucket: record positions grouped by local group */
  unsigned long long* ord; /* [n] sort keys of long segments */
  uint8_t* o_kind;        /* [n] per-record output flag (0 = none) */
  I4* o_rec;              /* [n] per-record output payload */
  int32_t* blocksum;      /* output-flag scan partials */
  unsigned long long* counters; /* [3] votes, outputs (decisions + preempts), dropped */
};

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
__device__ __forceinline__ int32_t block_exscan(int32_t v, int32_t* total) {
  __shared__ int32_t wsum[GPX_BLOCK / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  int32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < GPX_BLOCK / 64; w++) {
    int32_t s = wsum[w];
    if (w < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

/* ------------------------------------------------------------------------- */
/* front end                                                                    */

/* per-tile bucket histogram in LDS; also resets the per-record output flag and writes the
 * common per-record status (coalesced) so the apply kernels only touch the rare non-OK ones */
__global__ __launch_bounds__(GPX_BLOCK) void k_hist(int32_t n, const int32_t* __restrict__ gidx,
                                                   int32_t G, DevScratch X,
                                                   uint8_t* __restrict__ status,
                                                   int32_t is_votes) {
  extern __shared__ int32_t lds[];
  for (int32_t b = threadIdx.x; b < X.nbk; b += GPX_BLOCK) lds[b] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * GPX_TILE;
  int32_t bad = 0;
#pragma unroll
  for (int j = 0; j < GPX_TILE_ITEMS; j++) {
    const int64_t i = base + j * GPX_BLOCK + threadIdx.x;
    if (i < n) {
      const int32_t g = gidx[i];
      X.o_kind[i] = 0;
      if ((uint32_t)g < (uint32_t)G) {
        atomicAdd(&lds[g >> X.shift], 1);
        if (status) status[i] = GPX_S_OK;
      } else {
        if (status) status[i] = GPX_S_NOGROUP; /* PaxosManager.java:1162-1194: no such instance */
        bad++;
      }
    }
  }
  if (bad) atomicAdd(&X.counters[2], (unsigned long long)bad);
  if (blockIdx.x == 0 && threadIdx.x == 0 && is_votes) atomicAdd(&X.counters[0], (unsigned long long)n);
  __syncthreads();
  int32_t* out = X.tile_hist + (int64_t)blockIdx.x * X.nbk;
  for (int32_t b = threadIdx.x; b < X.nbk; b += GPX_BLOCK) out[b] = lds[b];
}

/* exclusive scan of tile_hist down the tiles of one chunk, per bucket; chunk sum -> chunk_part */
__global__ __launch_bounds__(GPX_BLOCK) void k_colscan(DevScratch X, int32_t ntiles, int32_t tc) {
  const int32_t b = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (b >= X.nbk) return;
  const int32_t t0 = blockIdx.y * tc;
  const int32_t t1 = min(t0 + tc, ntiles);
  int32_t run = 0;
  for (int32_t t = t0; t < t1; t++) {
    int32_t* p = X.tile_hist + (int64_t)t * X.nbk + b;
    const int32_t v = *p;
    *p = run;
    run += v;
  }
  X.chunk_part[(int64_t)blockIdx.y * X.nbk + b] = run;
}

/* per bucket: exclusive scan of the chunk sums (parallel over buckets; loads issued in batches of
 * 8 so the loop is not one memory latency per chunk); bucket total -> bucket_off[b] (unscanned) */
__global__ __launch_bounds__(GPX_BLOCK) void k_chunkscan(DevScratch X, int32_t nchunks) {
  const int32_t b = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (b >= X.nbk) return;
  int32_t run = 0;
  for (int32_t c0 = 0; c0 < nchunks; c0 += 8) {
    int32_t v[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
      v[q] = (c0 + q < nchunks) ? X.chunk_part[(int64_t)(c0 + q) * X.nbk + b] : 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (c0 + q < nchunks) X.chunk_part[(int64_t)(c0 + q) * X.nbk + b] = run;
      run += v[q];
    }
  }
  X.bucket_off[b] = run;
}

/* one workgroup: exclusive scan of the bucket totals in place; thread t owns a run of
 * consecutive buckets, so a single block scan covers all of them (nbk <= 256 * 64) */
__global__ __launch_bounds__(GPX_BLOCK) void k_bucketscan(DevScratch X) {
  const int32_t per = (X.nbk + GPX_BLOCK - 1) / GPX_BLOCK;
  const int32_t b0 = threadIdx.x * per;
  int32_t v[64];
  int32_t s = 0;
#pragma unroll
  for (int q = 0; q < 64; q++) {
    if (q < per) {
      v[q] = (b0 + q < X.nbk) ? X.bucket_off[b0 + q] : 0;
      s += v[q];
    }
  }
  int32_t tot;
  int32_t ex = block_exscan(s, &tot);
#pragma unroll
  for (int q = 0; q < 64; q++) {
    if (q < per) {
      if (b0 + q < X.nbk) X.bucket_off[b0 + q] = ex;
      ex += v[q];
    }
  }
  if (threadIdx.x == 0) X.bucket_off[X.nbk] = tot;
}

/* LDS cursor per bucket for this tile = bucket_off + chunk prefix + in-chunk tile prefix */
__device__ __forceinline__ void scatter_init(const DevScratch& X, int32_t* lds, int32_t tc) {
  const int32_t* th = X.tile_hist + (int64_t)blockIdx.x * X.nbk;
  const int32_t* cp = X.chunk_part + (int64_t)(blockIdx.x / tc) * X.nbk;
  for (int32_t b = threadIdx.x; b < X.nbk; b += GPX_BLOCK) lds[b] = X.bucket_off[b] + cp[b] + th[b];
  __syncthreads();
}

/* accept-reply votes */
__global__ __launch_bounds__(GPX_BLOCK) void k_scatter_ar(
    int32_t n, int32_t G, int32_t tc, DevScratch X, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ bnum, const int32_t* __restrict__ bcoord,
    const int32_t* __restrict__ slot, const int32_t* __restrict__ acceptor,
    const int32_t* __restrict__ max_cp) {
  extern __shared__ int32_t lds[];
  scatter_init(X, lds, tc);
  const int64_t base = (int64_t)blockIdx.x * GPX_TILE;
  const int32_t mask = X.gb - 1;
#pragma unroll 4
  for (int j = 0; j < GPX_TILE_ITEMS; j++) {
    const int64_t i = base + j * GPX_BLOCK + threadIdx.x;
    if (i < n) {
      const int32_t g = gidx[i];
      if ((uint32_t)g < (uint32_t)G) {
        const int32_t pos = atomicAdd(&lds[g >> X.shift], 1);
        Rec r;
        r.a = slot[i];
        r.b = acceptor[i];
        r.c = max_cp[i];
        r.idx = (int32_t)i;
        r.bnum = bnum[i];
        r.bcoord = bcoord[i];
        r.lg = g & mask;
        r.pad = 0;
        X.rec[pos] = r;
      }
    }
  }
}

/* accepts / commits; for accepts also zeroes the dense reply columns of dropped records */
__global__ __launch_bounds__(GPX_BLOCK) void k_scatter_ac(
    int32_t n, int32_t G, int32_t tc, DevScratch X, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ bnum, const int32_t* __restrict__ bcoord,
    const int32_t* __restrict__ slot, const int32_t* __restrict__ median_cp,
    const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags) {
  extern __shared__ int32_t lds[];
  scatter_init(X, lds, tc);
  const int64_t base = (int64_t)blockIdx.x * GPX_TILE;
  const int32_t mask = X.gb - 1;
#pragma unroll 4
  for (int j = 0; j < GPX_TILE_ITEMS; j++) {
    const int64_t i = base + j * GPX_BLOCK + threadIdx.x;
    if (i < n) {
      const int32_t g = gidx[i];
      if ((uint32_t)g < (uint32_t)G) {
        const int32_t pos = atomicAdd(&lds[g >> X.shift], 1);
        Rec r;
        r.a = slot[i];
        r.b = median_cp[i];
        r.c = flags ? (int32_t)flags[i] : 0;
        r.idx = (int32_t)i;
        r.bnum = bnum[i];
        r.bcoord = bcoord[i];
        r.lg = g & mask;
        r.pad = 0;
        X.rec[pos] = r;
      } else if (r_bnum) {
        r_bnum[i] = 0;
        r_bcoord[i] = 0;
        r_maxcp[i] = 0;
        r_flags[i] = 0;
      }
    }
  }
}

/* proposals */
__global__ __launch_bounds__(GPX_BLOCK) void k_scatter_pr(
    int32_t n, int32_t G, int32_t tc, DevScratch X, const int32_t* __restrict__ gidx,
    const uint8_t* __restrict__ is_stop, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median) {
  extern __shared__ int32_t lds[];
  scatter_init(X, lds, tc);
  const int64_t base = (int64_t)blockIdx.x * GPX_TILE;
  const int32_t mask = X.gb - 1;
#pragma unroll 4
  for (int j = 0; j < GPX_TILE_ITEMS; j++) {
    const int64_t i = base + j * GPX_BLOCK + threadIdx.x;
    if (i < n) {
      const int32_t g = gidx[i];
      if ((uint32_t)g < (uint32_t)G) {
        const int32_t pos = atomicAdd(&lds[g >> X.shift], 1);
        Rec r;
        r.a = is_stop ? (int32_t)(is_stop[i] & 1) : 0;
        r.b = 0;
        r.c = 0;
        r.idx = (int32_t)i;
        r.bnum = 0;
        r.bcoord = 0;
        r.lg = g & mask;
        r.pad = 0;
        X.rec[pos] = r;
      } else {
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

__device__ __forceinline__ void cmpxchg_asc(unsigned long long* a, uint32_t lo, uint32_t hi) {
  unsigned long long x = a[lo], y = a[hi];
  if (x > y) {
    a[lo] = y;
    a[hi] = x;
  }
}

/* Sorts perm[0 .. c) of ONE long segment by arrival index, cooperatively by the whole workgroup,
 * through 64-bit keys (idx << 32 | j) in global scratch `a`.  All-ascending bitonic network
 * (first stage of every merge compares t with its mirror t ^ (k-1), the rest with t ^ j), so
 * positions >= c behave as +inf simply by being skipped.  A single hot group is inherently serial
 * under the per-group ordering contract (like the Java monitor); this only has to be correct.
 * rec / perm may point to LDS or to global memory. */
__device__ void sort_long_segment(const Rec* rec, int32_t* perm, unsigned long long* a, uint32_t c) {
  for (uint32_t t = threadIdx.x; t < c; t += GPX_BLOCK) {
    const int32_t j = perm[t];
    a[t] = ((unsigned long long)(uint32_t)rec[j].idx << 32) | (uint32_t)j;
  }
  __syncthreads();
  uint32_t p2 = 1;
  while (p2 < c) p2 <<= 1;
  for (uint32_t k = 2; k <= p2; k <<= 1) {
    for (uint32_t t = threadIdx.x; t < c; t += GPX_BLOCK) {
      const uint32_t q = t ^ (k - 1);
      if (q > t && q < c) cmpxchg_asc(a, t, q);
    }
    __syncthreads();
    for (uint32_t j = k >> 2; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < c; t += GPX_BLOCK) {
        const uint32_t q = t ^ j;
        if (q > t && q < c) cmpxchg_asc(a, t, q);
      }
      __syncthreads();
    }
  }
  for (uint32_t t = threadIdx.x; t < c; t += GPX_BLOCK) perm[t] = (int32_t)(uint32_t)(a[t] & 0xffffffffull);
  __syncthreads();
}

/* What one workgroup sees of its bucket after regrouping: records (bucket-relative index j) and
 * perm (records of local group lg at perm[loff[lg] .. loff[lg] + lcnt[lg])).  Both live in LDS
 * when the bucket has at most GPX_LDS_RECS records (the normal case: ~K records per group), else
 * in global scratch. */
struct BucketView {
  const Rec* rec;
  int32_t* perm;
  int32_t* lcnt;
  int32_t* loff;
};

/* dynamic LDS of every k_bucket_* kernel: lcnt[gb] | loff[gb] | Rec[GPX_LDS_RECS] | perm[GPX_LDS_RECS] */
#define GPX_BUCKET_LDS_BYTES(gb) ((size_t)(gb) * 8 + (size_t)GPX_LDS_RECS * (sizeof(Rec) + 4))

/* Returns false (whole workgroup) when the bucket received no record. */
__device__ __forceinline__ bool bucket_prepare(const DevScratch& X, int32_t* lds, BucketView* bv) {
  const int32_t b = blockIdx.x;
  const int32_t boff = X.bucket_off[b];
  const int32_t nb = X.bucket_off[b + 1] - boff;
  if (nb == 0) return false;
  const int32_t gb = X.gb;
  int32_t* lcnt = lds;
  int32_t* loff = lds + gb;
  Rec* recL = (Rec*)(lds + 2 * gb);
  int32_t* permL = (int32_t*)(recL + GPX_LDS_RECS);
  const bool in_lds = nb <= GPX_LDS_RECS;
  const Rec* recG = X.rec + boff;
  bv->lcnt = lcnt;
  bv->loff = loff;
  bv->rec = in_lds ? (const Rec*)recL : recG;
  bv->perm = in_lds ? permL : (X.perm + boff);
  for (int32_t l = threadIdx.x; l < gb; l += GPX_BLOCK) lcnt[l] = 0;
  __syncthreads();
  /* A: count per local group (LDS atomics); remember each record's rank */
  int32_t rk[GPX_LDS_RECS / GPX_BLOCK];
  if (in_lds) {
#pragma unroll
    for (int m = 0; m < GPX_LDS_RECS / GPX_BLOCK; m++) {
      const int32_t j = m * GPX_BLOCK + threadIdx.x;
      rk[m] = 0;
      if (j < nb) {
        const Rec r = recG[j]; /* coalesced 32 B per lane */
        recL[j] = r;
        rk[m] = atomicAdd(&lcnt[r.lg], 1);
      }
    }
  } else {
    for (int32_t j = threadIdx.x; j < nb; j += GPX_BLOCK)
      X.rank2[boff + j] = atomicAdd(&lcnt[recG[j].lg], 1);
  }
  __syncthreads();
  /* B: exclusive scan lcnt -> loff; thread t owns gb/256 consecutive groups */
  const int32_t per = gb / GPX_BLOCK;
  int32_t s = 0;
  for (int32_t q = 0; q < per; q++) s += lcnt[threadIdx.x * per + q];
  int32_t tot;
  int32_t ex = block_exscan(s, &tot);
  int32_t any_long = 0;
  for (int32_t q = 0; q < per; q++) {
    const int32_t l = threadIdx.x * per + q;
    loff[l] = ex;
    ex += lcnt[l];
    any_long |= lcnt[l] > GPX_SMALL_SEG;
  }
  any_long = __syncthreads_or(any_long);
  /* C: perm */
  if (in_lds) {
#pragma unroll
    for (int m = 0; m < GPX_LDS_RECS / GPX_BLOCK; m++) {
      const int32_t j = m * GPX_BLOCK + threadIdx.x;
      if (j < nb) permL[loff[recL[j].lg] + rk[m]] = j;
    }
  } else {
    for (int32_t j = threadIdx.x; j < nb; j += GPX_BLOCK)
      X.perm[boff + loff[recG[j].lg] + X.rank2[boff + j]] = j;
  }
  __syncthreads();
  /* D: arrival-order sort of long segments (rare) */
  if (any_long) {
    for (int32_t l = 0; l < gb; l++) {
      const int32_t c = lcnt[l]; /* uniform across the workgroup */
      if (c > GPX_SMALL_SEG)
        sort_long_segment(bv->rec, bv->perm + loff[l], X.ord + boff + loff[l], (uint32_t)c);
    }
  }
  return true;
}

/* Iterates one group's records in arrival order.  c <= 4: (idx, j) pairs sorted in registers;
 * c <= GPX_SMALL_SEG: repeated min-scan; longer: perm is already in arrival order. */
struct GroupIter {
  const Rec* rec;      /* bucket base */
  const int32_t* perm; /* segment base */
  int32_t c, done, last;
  int32_t j0, j1, j2, j3;
  __device__ __forceinline__ void init(const Rec* r, const int32_t* p, int32_t n) {
    rec = r;
    perm = p;
    c = n;
    done = 0;
    last = -1;
    j0 = j1 = j2 = j3 = 0;
    if (n <= 4) {
      int32_t i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff, i3 = 0x7fffffff;
      j0 = perm[0];
      i0 = rec[j0].idx;
      if (n > 1) {
        j1 = perm[1];
        i1 = rec[j1].idx;
      }
      if (n > 2) {
        j2 = perm[2];
        i2 = rec[j2].idx;
      }
      if (n > 3) {
        j3 = perm[3];
        i3 = rec[j3].idx;
      }
#define GPX_CSWAP(ia, ja, ib, jb) \
  if (ia > ib) {                  \
    int32_t t_ = ia;              \
    ia = ib;                      \
    ib = t_;                      \
    t_ = ja;                      \
    ja = jb;                      \
    jb = t_;                      \
  }
      GPX_CSWAP(i0, j0, i1, j1)
      GPX_CSWAP(i2, j2, i3, j3)
      GPX_CSWAP(i0, j0, i2, j2)
      GPX_CSWAP(i1, j1, i3, j3)
      GPX_CSWAP(i1, j1, i2, j2)
#undef GPX_CSWAP
    }
  }
  /* bucket-relative record position of the next record, or -1 */
  __device__ __forceinline__ int32_t next() {
    if (done >= c) return -1;
    int32_t j;
    if (c <= 4) {
      j = done == 0 ? j0 : (done == 1 ? j1 : (done == 2 ? j2 : j3));
    } else if (c <= GPX_SMALL_SEG) {
      int32_t best = 0x7fffffff, bj = -1;
      for (int32_t t = 0; t < c; t++) {
        const int32_t jj = perm[t];
        const int32_t ix = rec[jj].idx;
        if (ix > last && ix < best) {
          best = ix;
          bj = jj;
        }
      }
      last = best;
      j = bj;
    } else {
      j = perm[done];
    }
    done++;
    return j;
  }
};

/* PaxosCoordinatorState.getMedianMinus (PaxosCoordinatorState.java:867-875): element of rank
 * (k even ? k/2-1 : k/2) in signed ascending order.  Rank selection, ties by index. */
template <int KMAX>
__device__ __forceinline__ int32_t median_minus(const int32_t (&ns)[KMAX], int32_t k) {
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
/* coordinator side                                                             */
/* PaxosInstanceStateMachine.handleAcceptReply (PISM:1248-1364) ->              */
/* PaxosCoordinator.handleAcceptReply (PaxosCoordinator.java:210-250) ->        */
/* PaxosCoordinatorState.handleAcceptReplyMyBallot / HigherBallot (:597-683)    */
template <int KMAX>
__device__ __forceinline__ void apply_ar_group(const DevState& S, const DevScratch& X, int32_t g,
                                               GroupIter& it, uint8_t* __restrict__ status) {
  const int32_t G = S.G;
  const uint32_t gf = S.g_flags[g];
  if (!(gf & GF_EXISTS) || (gf & GF_STOPPED)) {
    /* PaxosManager.java:1162-1194 / PaxosInstanceStateMachine.java:456-460: dropped */
    const uint8_t st = (gf & GF_EXISTS) ? GPX_S_STOPPED : GPX_S_NOGROUP;
    for (int32_t j = it.next(); j >= 0; j = it.next())
      if (status) status[it.rec[j].idx] = st;
    atomicAdd(&X.counters[2], (unsigned long long)it.c); /* rare path */
    return;
  }
  const int32_t k = (int32_t)GF_K(gf);
  bool has_coord = (gf & GF_HASCOORD) != 0;
  const int32_t my_bnum = S.c_bnum[g], my_bcoord = S.c_bcoord[g];
  const int32_t next = S.c_next[g];
  int32_t pcount = S.c_pcount[g];
  const int32_t pcount0 = pcount;
  const int32_t Wm = S.W - 1;
  int32_t mem[KMAX], ns[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; j++) {
    mem[j] = (j < k) ? S.members[(int64_t)j * G + g] : 0;
    ns[j] = (j < k) ? S.node_slots[(int64_t)j * G + g] : 0;
  }
  bool ns_dirty = false;
  for (int32_t j = it.next(); j >= 0; j = it.next()) {
    const Rec r = it.rec[j];
    const int32_t slot = r.a, acc = r.b, maxcp = r.c, ix = r.idx;
    if (!has_coord) continue; /* PaxosCoordinator.java:196-198: c == null -> null */
    const int32_t cmp = ballot_cmp(r.bnum, r.bcoord, my_bnum, my_bcoord);
    const int32_t d = jsub(next, slot); /* slot in myProposals' window iff 1 <= d <= W */
    const bool inwin = (d >= 1) && (d <= S.W);
    if (cmp > 0) {
      /* handleAcceptReplyHigherBallot :661-675 */
      if (inwin) {
        uint32_t* pe = &S.p_ring[(int64_t)(slot & Wm) * G + g];
        const uint32_t e = *pe;
        if (e & PR_PRESENT) {
          *pe = 0;
          pcount--;
          X.o_rec[ix] = mk4(my_bnum, my_bcoord, -1, 0); /* preempt(): median stays -1 */
          X.o_kind[ix] = GPX_D_PREEMPTED;
        }
      }
      /* nullifyCoordinatorIfPreemptedFully, PISM:1361-1364 */
      if (pcount == 0) has_coord = false;
    } else if (cmp == 0) {
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
        uint32_t* pe = &S.p_ring[(int64_t)(slot & Wm) * G + g];
        uint32_t e = *pe;
        if (e & PR_PRESENT) {
          if (midx >= 0) e |= (1u << midx); /* updateHeardFrom :51-62 */
          if (__popc(e & 0xffffu) > k / 2) { /* heardFromMajority :64-68 */
            *pe = 0;
            pcount--;
            X.o_rec[ix] = mk4(my_bnum, my_bcoord, median_minus<KMAX>(ns, k), 0);
            X.o_kind[ix] = GPX_D_DECISION;
          } else {
            *pe = e;
          }
        }
      }
    }
    /* cmp < 0: reply to a lower ballot, ignored (PaxosCoordinator.java:241-247) */
  }
  if (ns_dirty) {
#pragma unroll
    for (int q = 0; q < KMAX; q++)
      if (q < k) S.node_slots[(int64_t)q * G + g] = ns[q];
  }
  if (pcount != pcount0) S.c_pcount[g] = pcount;
  if (!has_coord && (gf & GF_HASCOORD)) S.g_flags[g] = gf & ~GF_HASCOORD;
}

template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_bucket_ar(DevState S, DevScratch X,
                                                        uint8_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(32))) int32_t lds[];
  BucketView bv;
  if (!bucket_prepare(X, lds, &bv)) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += GPX_BLOCK) {
    const int32_t c = bv.lcnt[l];
    if (c == 0 || g0 + l >= S.G) continue;
    GroupIter it;
    it.init(bv.rec, bv.perm + bv.loff[l], c);
    apply_ar_group<KMAX>(S, X, g0 + l, it, status);
  }
}

/* ------------------------------------------------------------------------- */
/* acceptor helpers (one lane owns the group: plain loads/stores)               */

struct AccState {
  int32_t slot, bnum, bcoord, gc;
  bool stopped;
};

/* PaxosAcceptor.garbageCollectAccepted (PaxosAcceptor.java:476-494).
 * garbageCollectDecisions (:496-506) can never find anything: committedRequests only ever
 * holds slots >= _slot (put only if slot - _slot >= 0, :341; removed on execution) and it
 * only drops slots < gcSlot <= _slot - 1. */
__device__ __forceinline__ void acc_gc(const DevState& S, int32_t g, AccState& a, int32_t gcSlot) {
  if (jsub(a.slot, gcSlot) <= 0) gcSlot = jsub(a.slot, 1);
  const int32_t delta = jsub(gcSlot, a.gc);
  if (delta > 0) {
    const int32_t Wm = S.W - 1;
    if (delta >= S.W) {
      for (int32_t w = 0; w < S.W; w++) {
        const int64_t o = (int64_t)w * S.G + g;
        if ((S.acc_flags[o] & RF_PRESENT) && jsub(S.acc_ring[o].x, gcSlot) <= 0) S.acc_flags[o] = 0;
      }
    } else {
      /* live accepted slots are all > a.gc: only (a.gc, gcSlot] can die */
      for (int32_t s = (int32_t)((uint32_t)a.gc + 1u);; s = (int32_t)((uint32_t)s + 1u)) {
        const int64_t o = (int64_t)(s & Wm) * S.G + g;
        if ((S.acc_flags[o] & RF_PRESENT) && S.acc_ring[o].x == s) S.acc_flags[o] = 0;
        if (s == gcSlot) break;
      }
    }
    a.gc = gcSlot;
  }
}

struct Dec {
  int32_t bnum, bcoord, slot, median;
  bool has_value, stop;
};

/* PaxosAcceptor.reconstructDecision (PaxosAcceptor.java:369-385).
 * Written branch-free on purpose.  The natural nested-if form (return early per failed test,
 * assign *out inside the two succeeding branches) was MISCOMPILED by hipcc (ROCm 7.2, -O3,
 * gfx950) once inlined into the accept kernel: after CFG structurization the median of the
 * "placeholder + matching accept" path was replaced by the failing paths' value (an undefined
 * register, or 0 when *out was pre-zeroed) — the isolated function's LLVM IR was correct, the
 * kernel's ISA was not.  Found by the parity fuzz; the select form below has no merge to get
 * wrong.  All four loads are always in bounds (same ring index for both rings). */
__device__ __forceinline__ bool acc_reconstruct(const DevState& S, int32_t g, int32_t slot,
                                                Dec* out) {
  const int64_t o = (int64_t)(slot & (S.W - 1)) * S.G + g;
  const uint32_t cf = S.com_flags[o];
  const uint32_t af = S.acc_flags[o];
  const I4 cr = S.com_ring[o];
  const I4 ar = S.acc_ring[o];
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
__device__ __forceinline__ int32_t acc_eec(const DevState& S, int32_t g, AccState& a,
                                           const Dec& d) {
  int32_t count = 0;
  const int32_t Wm = S.W - 1;
  const bool from_disk = (S.flags & GPX_F_ACCEPTS_FROM_DISK) != 0;
  while (!a.stopped) {
    acc_gc(S, g, a, d.median);
    if (jsub(d.slot, a.slot) >= 0) {
      /* don't overwrite an existing decision that has a value (:343-346) */
      const int64_t o = (int64_t)(d.slot & Wm) * S.G + g;
      const uint8_t cf = S.com_flags[o];
      const bool same = (cf & RF_PRESENT) && S.com_ring[o].x == d.slot;
      if (!same || !(cf & RF_HASVALUE)) {
        S.com_ring[o] = mk4(d.slot, d.bnum, d.bcoord, d.median);
        S.com_flags[o] =
            (uint8_t)(RF_PRESENT | (d.has_value ? RF_HASVALUE : 0) | (d.stop ? RF_STOP : 0));
      }
    }
    Dec nx = Dec{0, 0, 0, 0, false, false};
    if (!acc_reconstruct(S, g, a.slot, &nx)) break;
    /* committedRequests.remove(_slot); executed(slot, isStop) */
    const int64_t o0 = (int64_t)(a.slot & Wm) * S.G + g;
    S.com_flags[o0] = 0;
    a.slot = (int32_t)((uint32_t)a.slot + 1u);
    if (nx.stop) a.stopped = true;
    if (a.stopped)
      for (int32_t w = 0; w < S.W; w++) S.com_flags[(int64_t)w * S.G + g] = 0;
    if (from_disk) {
      /* acceptedProposals.remove(nextExecutable.slot) (:357-359) */
      if ((S.acc_flags[o0] & RF_PRESENT) && S.acc_ring[o0].x == nx.slot) S.acc_flags[o0] = 0;
    }
    count++;
    if (nx.stop) break;
  }
  return count;
}

__device__ __forceinline__ void acc_load(const DevState& S, int32_t g, uint32_t gf, AccState& a) {
  a.slot = S.a_slot[g];
  a.bnum = S.a_bnum[g];
  a.bcoord = S.a_bcoord[g];
  a.gc = S.a_gc[g];
  a.stopped = (gf & GF_STOPPED) != 0;
}
__device__ __forceinline__ void acc_store(const DevState& S, int32_t g, uint32_t gf,
                                          const AccState& a) {
  S.a_slot[g] = a.slot;
  S.a_bnum[g] = a.bnum;
  S.a_bcoord[g] = a.bcoord;
  S.a_gc[g] = a.gc;
  if (a.stopped && !(gf & GF_STOPPED)) S.g_flags[g] = gf | GF_STOPPED;
}

/* PaxosInstanceStateMachine.handleAccept (PISM:1080-1166) */
__device__ __forceinline__ void apply_accept_group(
    const DevState& S, const DevScratch& X, int32_t g, GroupIter& it, int32_t* __restrict__ r_bnum,
    int32_t* __restrict__ r_bcoord, int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags,
    uint8_t* __restrict__ status) {
  const uint32_t gf = S.g_flags[g];
  AccState a;
  a.slot = a.bnum = a.bcoord = a.gc = 0;
  a.stopped = false;
  const bool exists = (gf & GF_EXISTS) != 0;
  if (exists) acc_load(S, g, gf, a);
  const int32_t Wm = S.W - 1;
  unsigned long long n_drop = 0;
  for (int32_t j = it.next(); j >= 0; j = it.next()) {
    const Rec r = it.rec[j];
    const int32_t slot = r.a, median = r.b, ix = r.idx;
    const bool stop = (r.c & GPX_A_STOP) != 0;
    r_bnum[ix] = 0;
    r_bcoord[ix] = 0;
    r_maxcp[ix] = 0;
    r_flags[ix] = 0;
    if (!exists || a.stopped) {
      status[ix] = exists ? GPX_S_STOPPED : GPX_S_NOGROUP;
      n_drop++;
      continue;
    }
    const int64_t o = (int64_t)(slot & Wm) * S.G + g;
    /* PValuePacket prev = paxosState.getAccept(accept.slot)  (:1122, before accepting) */
    const uint8_t af = S.acc_flags[o];
    const I4 ar = S.acc_ring[o];
    const bool live = (af & RF_PRESENT) != 0;
    const bool have_prev = live && ar.x == slot;
    /* PaxosAcceptor.acceptAndUpdateBallot (PaxosAcceptor.java:302-322) */
    const bool ballot_ok = ballot_cmp(r.bnum, r.bcoord, a.bnum, a.bcoord) >= 0;
    const bool will_store = ballot_ok && jsub(slot, a.gc) > 0;
    if (will_store && live && ar.x != slot) {
      status[ix] = GPX_S_WINDOW; /* ring slot held by another live accepted slot */
      n_drop++;
      continue;
    }
    if (ballot_ok) {
      a.bnum = r.bnum;
      a.bcoord = r.bcoord;
      if (will_store) {
        S.acc_ring[o] = mk4(slot, r.bnum, r.bcoord, 0);
        S.acc_flags[o] = (uint8_t)(RF_PRESENT | (stop ? RF_STOP : 0));
      }
    }
    acc_gc(S, g, a, median);
    /* reply (myID, ballot, slot, getSlot()-1)  (:1139-1143) */
    r_bnum[ix] = a.bnum;
    r_bcoord[ix] = a.bcoord;
    r_maxcp[ix] = jsub(a.slot, 1);
    /* toLog (:1146-1149) */
    const bool to_log = ballot_cmp(r.bnum, r.bcoord, a.bnum, a.bcoord) >= 0 &&
                        jsub(slot, a.gc) > 0 &&
                        (!have_prev || ballot_cmp(ar.y, ar.z, r.bnum, r.bcoord) < 0);
    r_flags[ix] = (uint8_t)((to_log ? GPX_R_TOLOG : 0) | (will_store ? GPX_R_STORED : 0));
    /* status[ix] stays GPX_S_OK (prefilled by k_hist) */
    /* might release some meta-commits (:1158-1161) */
    Dec rd = Dec{0, 0, 0, 0, false, false};
    if (acc_reconstruct(S, g, slot, &rd)) {
      const int32_t first = a.slot;
      const int32_t cnt_exec = acc_eec(S, g, a, rd);
      if (cnt_exec > 0) {
        X.o_rec[ix] = mk4(first, cnt_exec, 0, 0);
        X.o_kind[ix] = 1;
      }
    }
  }
  if (exists) acc_store(S, g, gf, a);
  if (n_drop) atomicAdd(&X.counters[2], n_drop);
}

__global__ __launch_bounds__(GPX_BLOCK) void k_bucket_accept(
    DevState S, DevScratch X, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, uint8_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(32))) int32_t lds[];
  BucketView bv;
  if (!bucket_prepare(X, lds, &bv)) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += GPX_BLOCK) {
    const int32_t c = bv.lcnt[l];
    if (c == 0 || g0 + l >= S.G) continue;
    GroupIter it;
    it.init(bv.rec, bv.perm + bv.loff[l], c);
    apply_accept_group(S, X, g0 + l, it, r_bnum, r_bcoord, r_maxcp, r_flags, status);
  }
}

/* PaxosInstanceStateMachine.handleBatchedCommit (PISM:1480-1528) per slot and
 * handleCommittedRequest (:1432-1478) for full decisions */
__device__ __forceinline__ void apply_commit_group(const DevState& S, const DevScratch& X,
                                                   int32_t g, GroupIter& it,
                                                   uint8_t* __restrict__ status) {
  const uint32_t gf = S.g_flags[g];
  AccState a;
  a.slot = a.bnum = a.bcoord = a.gc = 0;
  a.stopped = false;
  const bool exists = (gf & GF_EXISTS) != 0;
  if (exists) acc_load(S, g, gf, a);
  const int32_t Wm = S.W - 1;
  unsigned long long n_drop = 0;
  for (int32_t j = it.next(); j >= 0; j = it.next()) {
    const Rec r = it.rec[j];
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
      const int64_t o = (int64_t)(slot & Wm) * S.G + g;
      const uint8_t af = S.acc_flags[o];
      const I4 ar = S.acc_ring[o];
      if ((af & RF_PRESENT) && ar.x == slot && ballot_cmp(ar.y, ar.z, r.bnum, r.bcoord) == 0)
        d = Dec{ar.y, ar.z, slot, median, true, (af & RF_STOP) != 0};
      else
        d = Dec{r.bnum, r.bcoord, slot, median, false, false}; /* placeholder (:1510-1520) */
    }
    const int32_t first = a.slot;
    const int32_t cnt_exec = acc_eec(S, g, a, d);
    if (cnt_exec > 0) {
      X.o_rec[ix] = mk4(first, cnt_exec, 0, 0);
      X.o_kind[ix] = 1;
    }
  }
  if (exists) acc_store(S, g, gf, a);
  if (n_drop) atomicAdd(&X.counters[2], n_drop);
}

__global__ __launch_bounds__(GPX_BLOCK) void k_bucket_commit(DevState S, DevScratch X,
                                                            uint8_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(32))) int32_t lds[];
  BucketView bv;
  if (!bucket_prepare(X, lds, &bv)) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += GPX_BLOCK) {
    const int32_t c = bv.lcnt[l];
    if (c == 0 || g0 + l >= S.G) continue;
    GroupIter it;
    it.init(bv.rec, bv.perm + bv.loff[l], c);
    apply_commit_group(S, X, g0 + l, it, status);
  }
}

/* PaxosInstanceStateMachine.handleProposal (PISM:818-888) ->
 * PaxosCoordinatorState.propose (:233-263) + initCommander (:841-851) */
template <int KMAX>
__device__ __forceinline__ void apply_propose_group(
    const DevState& S, const DevScratch& X, int32_t g, GroupIter& it, int32_t* __restrict__ o_slot,
    int32_t* __restrict__ o_bnum, int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median,
    uint8_t* __restrict__ status) {
  const int32_t G = S.G;
  const uint32_t gf = S.g_flags[g];
  const bool exists = (gf & GF_EXISTS) != 0, stopped = (gf & GF_STOPPED) != 0;
  const int32_t k = (int32_t)GF_K(gf);
  const int32_t a_bnum = exists ? S.a_bnum[g] : 0, a_bcoord = exists ? S.a_bcoord[g] : 0;
  const int32_t my_bnum = exists ? S.c_bnum[g] : 0, my_bcoord = exists ? S.c_bcoord[g] : 0;
  /* PaxosCoordinator.exists(coordinator, paxosState.getBallot()) (PISM:825-826) */
  const bool coord_ok = exists && (gf & GF_HASCOORD) &&
                        ballot_cmp(my_bnum, my_bcoord, a_bnum, a_bcoord) >= 0;
  int32_t next = coord_ok ? S.c_next[g] : 0;
  int32_t pcount = coord_ok ? S.c_pcount[g] : 0;
  int32_t ns[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; q++) ns[q] = (coord_ok && q < k) ? S.node_slots[(int64_t)q * G + g] : 0;
  const int32_t median = coord_ok ? median_minus<KMAX>(ns, k) : 0;
  const int32_t Wm = S.W - 1;
  unsigned long long n_drop = 0;
  for (int32_t j = it.next(); j >= 0; j = it.next()) {
    const Rec r = it.rec[j];
    const int32_t ix = r.idx;
    const bool stop = r.a != 0;
    o_slot[ix] = 0;
    o_bnum[ix] = 0;
    o_bcoord[ix] = 0;
    o_median[ix] = 0;
    if (!exists || stopped) {
      status[ix] = exists ? GPX_S_STOPPED : GPX_S_NOGROUP;
      n_drop++;
      continue;
    }
    if (!coord_ok) {
      /* unicast to paxosState.getBallotCoord() (PISM:854-860) */
      o_bnum[ix] = a_bnum;
      o_bcoord[ix] = a_bcoord;
      status[ix] = GPX_S_FORWARD;
      continue;
    }
    /* no point enqueuing anything after stop (PaxosCoordinatorState.java:235-239) */
    const uint32_t pe_prev = S.p_ring[(int64_t)(jsub(next, 1) & Wm) * G + g];
    if ((pe_prev & PR_PRESENT) && (pe_prev & PR_STOP)) {
      status[ix] = GPX_S_REFUSED;
      continue;
    }
    uint32_t* pe = &S.p_ring[(int64_t)(next & Wm) * G + g];
    if (*pe & PR_PRESENT) {
      status[ix] = GPX_S_WINDOW; /* slot next-W still outstanding */
      n_drop++;
      continue;
    }
    *pe = PR_PRESENT | (stop ? PR_STOP : 0u);
    pcount++;
    o_slot[ix] = next;
    o_bnum[ix] = my_bnum;
    o_bcoord[ix] = my_bcoord;
    o_median[ix] = median; /* getMajorityCommittedSlot: nodeSlots unchanged by propose */
    next = (int32_t)((uint32_t)next + 1u);
  }
  if (coord_ok) {
    S.c_next[g] = next;
    S.c_pcount[g] = pcount;
  }
  if (n_drop) atomicAdd(&X.counters[2], n_drop);
}

template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_bucket_propose(
    DevState S, DevScratch X, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(32))) int32_t lds[];
  BucketView bv;
  if (!bucket_prepare(X, lds, &bv)) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += GPX_BLOCK) {
    const int32_t c = bv.lcnt[l];
    if (c == 0 || g0 + l >= S.G) continue;
    GroupIter it;
    it.init(bv.rec, bv.perm + bv.loff[l], c);
    apply_propose_group<KMAX>(S, X, g0 + l, it, o_slot, o_bnum, o_bcoord, o_median, status);
  }
}

/* ------------------------------------------------------------------------- */
/* ordered compaction of the per-record outputs                                 */

/* phase 1: per-tile count of flagged records */
__global__ __launch_bounds__(GPX_BLOCK) void k_flag_reduce(const uint8_t* __restrict__ o_kind,
                                                          int32_t n, int32_t* blocksum) {
  const int64_t base = (int64_t)blockIdx.x * GPX_SCAN_TILE;
  int32_t s = 0;
#pragma unroll
  for (int j = 0; j < GPX_SCAN_ITEMS; j++) {
    int64_t i = base + j * GPX_BLOCK + threadIdx.x;
    if (i < n) s += o_kind[i] != 0;
  }
  int32_t tot;
  block_exscan(s, &tot);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

/* phase 2: one block turns the tile sums into exclusive prefixes; total -> *total_out */
__global__ __launch_bounds__(GPX_BLOCK) void k_scan_top(int32_t* blocksum, int32_t nb,
                                                       int32_t* total_out,
                                                       unsigned long long* acc) {
  __shared__ int32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int32_t start = 0; start < nb; start += GPX_BLOCK) {
    int32_t i = start + threadIdx.x;
    int32_t v = (i < nb) ? blocksum[i] : 0;
    int32_t tot;
    int32_t ex = block_exscan(v, &tot);
    int32_t carry = carry_s;
    if (i < nb) blocksum[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (total_out) *total_out = carry_s;
    if (acc) atomicAdd(acc, (unsigned long long)carry_s);
  }
}

/* phase 3: the tile's flagged record indices go to LDS in arrival order (blocked scan: thread t
 * owns GPX_SCAN_ITEMS consecutive records), then thread t gathers the t-th flagged record and
 * writes output row base+t, so every output column is written as a dense coalesced run. */
__device__ __forceinline__ int32_t compact_tile_list(int32_t n, const uint8_t* __restrict__ o_kind,
                                                     int32_t* lds_idx, uint8_t* lds_kind) {
  const int64_t base = (int64_t)blockIdx.x * GPX_SCAN_TILE + (int64_t)threadIdx.x * GPX_SCAN_ITEMS;
  uint8_t kd[GPX_SCAN_ITEMS];
  int32_t s = 0;
#pragma unroll
  for (int j = 0; j < GPX_SCAN_ITEMS; j++) {
    int64_t i = base + j;
    kd[j] = (i < n) ? o_kind[i] : 0;
    s += kd[j] != 0;
  }
  int32_t tot;
  int32_t ex = block_exscan(s, &tot);
#pragma unroll
  for (int j = 0; j < GPX_SCAN_ITEMS; j++) {
    if (kd[j]) {
      lds_idx[ex] = (int32_t)(base + j);
      lds_kind[ex] = kd[j];
      ex++;
    }
  }
  __syncthreads();
  return tot;
}

/* decisions: d_* columns; gidx/slot are re-read from the input columns */
__global__ __launch_bounds__(GPX_BLOCK) void k_compact_dec(
    int32_t n, const uint8_t* __restrict__ o_kind, const I4* __restrict__ o_rec,
    const int32_t* __restrict__ blocksum, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ slot, int32_t* __restrict__ d_gidx, int32_t* __restrict__ d_slot,
    int32_t* __restrict__ d_bnum, int32_t* __restrict__ d_bcoord, int32_t* __restrict__ d_median,
    uint8_t* __restrict__ d_kind) {
  __shared__ int32_t lds_idx[GPX_SCAN_TILE];
  __shared__ uint8_t lds_kind[GPX_SCAN_TILE];
  const int32_t tot = compact_tile_list(n, o_kind, lds_idx, lds_kind);
  const int32_t out0 = blocksum[blockIdx.x];
  for (int32_t t = threadIdx.x; t < tot; t += GPX_BLOCK) {
    const int32_t i = lds_idx[t];
    const I4 r = o_rec[i];
    d_gidx[out0 + t] = gidx[i];
    d_slot[out0 + t] = slot[i];
    d_bnum[out0 + t] = r.x;
    d_bcoord[out0 + t] = r.y;
    d_median[out0 + t] = r.z;
    d_kind[out0 + t] = lds_kind[t];
  }
}

/* exec runs: (gidx, first, count) */
__global__ __launch_bounds__(GPX_BLOCK) void k_compact_runs(
    int32_t n, const uint8_t* __restrict__ o_kind, const I4* __restrict__ o_rec,
    const int32_t* __restrict__ blocksum, const int32_t* __restrict__ gidx,
    int32_t* __restrict__ x_gidx, int32_t* __restrict__ x_first, int32_t* __restrict__ x_count) {
  __shared__ int32_t lds_idx[GPX_SCAN_TILE];
  __shared__ uint8_t lds_kind[GPX_SCAN_TILE];
  const int32_t tot = compact_tile_list(n, o_kind, lds_idx, lds_kind);
  const int32_t out0 = blocksum[blockIdx.x];
  for (int32_t t = threadIdx.x; t < tot; t += GPX_BLOCK) {
    const int32_t i = lds_idx[t];
    const I4 r = o_rec[i];
    x_gidx[out0 + t] = gidx[i];
    x_first[out0 + t] = r.x;
    x_count[out0 + t] = r.y;
  }
}

/* ------------------------------------------------------------------------- */
/* lifecycle                                                                    */

/* PaxosInstanceStateMachine.hotRestore (PISM:677-690), PaxosAcceptor.hotRestore
 * (PaxosAcceptor.java:128-134), PaxosCoordinator.hotRestore (PaxosCoordinator.java:122-131) */
__global__ __launch_bounds__(GPX_BLOCK) void k_group_create(DevState S, int32_t n,
                                                           const int32_t* __restrict__ gidx,
                                                           const int32_t* __restrict__ members,
                                                           const uint8_t* __restrict__ kk,
                                                           const gpx_hri* __restrict__ rows,
                                                           uint8_t* __restrict__ status) {
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
    S.acc_flags[o] = 0;
    S.com_flags[o] = 0;
  }
  S.g_flags[g] = GF_EXISTS | (coord ? GF_HASCOORD : 0u) | ((uint32_t)k << 8);
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
  const bool coord = (gf & GF_HASCOORD) != 0;
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
                                                           uint8_t* __restrict__ status) {
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
      if (S.com_flags[o] & RF_PRESENT) caught = false;
      if (!from_disk && (S.acc_flags[o] & RF_PRESENT)) caught = false;
    }
    if ((gf & GF_HASCOORD) && S.c_pcount[g] != 0) caught = false;
    if (!caught) {
      if (status) status[i] = GPX_S_BUSY;
      return;
    }
  }
  if (rows) fill_hri_dev(S, g, gf, &rows[i]);
  if (mode != 2) S.g_flags[g] = 0;
  if (status) status[i] = GPX_S_OK;
}
