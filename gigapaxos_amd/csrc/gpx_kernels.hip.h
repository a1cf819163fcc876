/*
 * gpx_kernels.hip.h — CDNA4 (gfx950) kernels of the batched consensus engine.
 *
 * Integer / indexing work only: HBM-bound, no MFMA.  Wave = 64 lanes everywhere.
 *
 * Pipeline of every batch call:
 *   k_count   : one lane per record; rank[i] = atomicAdd(&cnt[gidx[i]], 1)
 *   scan      : exclusive scan of cnt[0..G) -> offs (3 small kernels)
 *   k_fill_*  : record i is packed and written to its group's segment
 *               seg[offs[g] + rank[i]]   (group-contiguous, order inside a segment
 *               is the atomic order, fixed up below)
 *   k_biglist / k_sort_big : segments longer than SMALL_SEG get an arrival-order
 *               permutation from a workgroup-wide bitonic sort
 *   k_apply_* : ONE LANE PER GROUP sweeps groups in gidx order (coalesced SoA state),
 *               replays that group's records in ARRIVAL ORDER exactly as the Java
 *               state machine would (PaxosInstanceStateMachine.handlePaxosMessage once
 *               per record), and writes per-record dense outputs
 *   compact   : scan of the per-record output flags + ordered gather, so decisions /
 *               exec runs leave in the arrival order of the record that produced them
 *
 * Each device function cites the reference method it implements (paths relative to
 * /root/reference/src/edu/umass/cs/gigapaxos/).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpx.h"

/* Plain 16-/8-byte aggregates for packed records and ring entries.  (Deliberately NOT HIP's
 * I4/I2 = HIP_vector_type: with ROCm 7.2 -O3 a `.w` read through its accessor proxy came out
 * undefined on one path of acc_reconstruct; plain structs give the same dwordx4 accesses.) */
struct __attribute__((aligned(16))) I4 {
  int32_t x, y, z, w;
};
struct __attribute__((aligned(8))) I2 {
  int32_t x, y;
};
__host__ __device__ __forceinline__ I4 mk4(int32_t x, int32_t y, int32_t z, int32_t w) {
  I4 r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = w;
  return r;
}
__host__ __device__ __forceinline__ I2 mk2(int32_t x, int32_t y) {
  I2 r;
  r.x = x;
  r.y = y;
  return r;
}

#define GPX_BLOCK 256
#define GPX_SCAN_ITEMS 8 /* items per thread in the scan kernels */
#define GPX_SCAN_TILE (GPX_BLOCK * GPX_SCAN_ITEMS)
#define GPX_SMALL_SEG 16 /* segments up to this long are ordered by per-lane min-scan */
#define GPX_SORT_LDS_MAX 4096 /* bitonic in LDS up to this many records */

/* group flag word */
#define GF_EXISTS 1u
#define GF_STOPPED 2u  /* PaxosAcceptor.STATES.STOPPED */
#define GF_HASCOORD 4u /* PaxosInstanceStateMachine.coordinator != null */
#define GF_K(f) (((f) >> 8) & 0xffu)

/* proposal ring entry: bits 0..15 = responded mask (WaitforUtility.responded), */
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
  int32_t *a_slot, *a_bnum, *a_bcoord, *a_gc;       /* PaxosAcceptor.java:94-99 */
  int32_t *c_bnum, *c_bcoord, *c_next, *c_pcount;   /* PaxosCoordinatorState.java:69-105 */
  int32_t* members;                                 /* [kmax][G] */
  int32_t* node_slots;                              /* [kmax][G] nodeSlotNumbers */
  uint32_t* p_ring;                                 /* [W][G] myProposals */
  I4* acc_ring;                                   /* [W][G] acceptedProposals {slot,bnum,bcoord,-} */
  uint8_t* acc_flags;                               /* [W][G] */
  I4* com_ring;                                   /* [W][G] committedRequests {slot,bnum,bcoord,median} */
  uint8_t* com_flags;                               /* [W][G] */
};

struct DevScratch {
  int32_t* cnt;      /* [G] per-group record count of the current batch (zero between calls) */
  int32_t* offs;     /* [G] exclusive scan of cnt */
  int32_t* rank;     /* [n] */
  I4* seg_a;       /* [n] packed record, .w = arrival index */
  I2* seg_b;       /* [n] */
  uint8_t* o_kind;   /* [n] per-record output flag (0 = none) */
  I4* o_rec;       /* [n] per-record output payload */
  int32_t* blocksum; /* scan partials */
  int32_t* biglist;  /* [1 + cap] count + gidx of long segments */
  unsigned long long* ord; /* [n] (arrival idx << 32 | pos) for long segments, sorted */
  unsigned long long* counters; /* [3] votes, decisions, dropped */
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

/* scan phase 1: per-tile sums.  MODE 0: int32 input; MODE 1: uint8 flags (!=0 -> 1) */
template <int MODE>
__global__ __launch_bounds__(GPX_BLOCK) void k_scan_reduce(const void* in, int32_t n,
                                                          int32_t* blocksum) {
  const int64_t base = (int64_t)blockIdx.x * GPX_SCAN_TILE;
  int32_t s = 0;
#pragma unroll
  for (int j = 0; j < GPX_SCAN_ITEMS; j++) {
    int64_t i = base + j * GPX_BLOCK + threadIdx.x;
    if (i < n) s += (MODE == 0) ? ((const int32_t*)in)[i] : (((const uint8_t*)in)[i] != 0);
  }
  int32_t tot;
  block_exscan(s, &tot);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

/* scan phase 2: one block turns the tile sums into exclusive prefixes; total -> *total_out */
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

/* scan phase 3 for the group counts: offs[g] = exclusive prefix.  Thread t owns
 * GPX_SCAN_ITEMS CONSECUTIVE items so a running sum gives the in-tile prefix. */
__global__ __launch_bounds__(GPX_BLOCK) void k_scan_down_offs(const int32_t* cnt, int32_t n,
                                                             const int32_t* blocksum,
                                                             int32_t* offs, int32_t* biglist,
                                                             int32_t cap) {
  const int64_t base = (int64_t)blockIdx.x * GPX_SCAN_TILE + (int64_t)threadIdx.x * GPX_SCAN_ITEMS;
  int32_t v[GPX_SCAN_ITEMS];
  int32_t s = 0;
#pragma unroll
  for (int j = 0; j < GPX_SCAN_ITEMS; j++) {
    int64_t i = base + j;
    v[j] = (i < n) ? cnt[i] : 0;
    s += v[j];
  }
  int32_t tot;
  int32_t ex = block_exscan(s, &tot) + blocksum[blockIdx.x];
#pragma unroll
  for (int j = 0; j < GPX_SCAN_ITEMS; j++) {
    int64_t i = base + j;
    if (i < n) offs[i] = ex;
    ex += v[j];
    if (v[j] > GPX_SMALL_SEG) { /* long segment: needs k_sort_big */
      int32_t q = atomicAdd(&biglist[0], 1);
      if (q < cap) biglist[1 + q] = (int32_t)i;
    }
  }
}

/* NOTE: k_scan_reduce sums a tile in strided order, k_scan_down_* in blocked order: both
 * cover the same GPX_SCAN_TILE items of tile blockIdx.x, so the tile sums agree. */

/* ------------------------------------------------------------------------- */
/* front end: count + rank.  Also clears the per-record output flag and writes  */
/* the NOGROUP status for out-of-range gidx.                                    */
__global__ __launch_bounds__(GPX_BLOCK) void k_count(int32_t n, const int32_t* __restrict__ gidx,
                                                    int32_t G, int32_t* cnt,
                                                    int32_t* __restrict__ rank,
                                                    uint8_t* __restrict__ o_kind,
                                                    uint8_t* __restrict__ status,
                                                    unsigned long long* counters,
                                                    int32_t* biglist, int32_t is_votes) {
  int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  if (i == 0) {
    biglist[0] = 0; /* consumed by the previous batch's k_sort_big / k_apply (stream order) */
    if (is_votes) atomicAdd(&counters[0], (unsigned long long)n);
  }
  int32_t g = gidx[i];
  o_kind[i] = 0;
  if ((uint32_t)g >= (uint32_t)G) {
    rank[i] = -1;
    if (status) status[i] = GPX_S_NOGROUP;
    atomicAdd(&counters[2], 1ull);
    return;
  }
  /* the common per-record status is written here, coalesced; k_apply_* only overwrites the
   * rare non-OK ones (a per-vote status byte scattered by arrival index from k_apply_ar cost a
   * 32 B sector write per vote: 96 MB per 3 M-vote batch in the round-1 profile) */
  if (status && is_votes) status[i] = GPX_S_OK;
  rank[i] = atomicAdd(&cnt[g], 1);
}

/* fill: accept-reply votes.  seg_a = {slot, acceptor, max_cp, idx}, seg_b = {bnum, bcoord} */
__global__ __launch_bounds__(GPX_BLOCK) void k_fill_ar(
    int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot,
    const int32_t* __restrict__ acceptor, const int32_t* __restrict__ max_cp,
    const int32_t* __restrict__ rank, const int32_t* __restrict__ offs, I4* __restrict__ seg_a,
    I2* __restrict__ seg_b) {
  int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  int32_t r = rank[i];
  if (r < 0) return;
  int32_t pos = offs[gidx[i]] + r;
  seg_a[pos] = mk4(slot[i], acceptor[i], max_cp[i], i);
  seg_b[pos] = mk2(bnum[i], bcoord[i]);
}

/* fill: accepts / commits.  seg_a = {slot, median_cp, flags, idx}, seg_b = {bnum, bcoord} */
__global__ __launch_bounds__(GPX_BLOCK) void k_fill_ac(
    int32_t n, const int32_t* __restrict__ gidx, const int32_t* __restrict__ bnum,
    const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot,
    const int32_t* __restrict__ median_cp, const uint8_t* __restrict__ flags,
    const int32_t* __restrict__ rank, const int32_t* __restrict__ offs, I4* __restrict__ seg_a,
    I2* __restrict__ seg_b, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags) {
  int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  int32_t r = rank[i];
  if (r < 0) {
    if (r_bnum) { /* accept: the dense reply columns of a dropped record read as zero */
      r_bnum[i] = 0;
      r_bcoord[i] = 0;
      r_maxcp[i] = 0;
      r_flags[i] = 0;
    }
    return;
  }
  int32_t pos = offs[gidx[i]] + r;
  seg_a[pos] = mk4(slot[i], median_cp[i], flags ? (int32_t)flags[i] : 0, i);
  seg_b[pos] = mk2(bnum[i], bcoord[i]);
}

/* fill: proposals.  seg_a = {is_stop, 0, 0, idx} */
__global__ __launch_bounds__(GPX_BLOCK) void k_fill_pr(int32_t n, const int32_t* __restrict__ gidx,
                                                      const uint8_t* __restrict__ is_stop,
                                                      const int32_t* __restrict__ rank,
                                                      const int32_t* __restrict__ offs,
                                                      I4* __restrict__ seg_a,
                                                      int32_t* __restrict__ o_slot,
                                                      int32_t* __restrict__ o_bnum,
                                                      int32_t* __restrict__ o_bcoord,
                                                      int32_t* __restrict__ o_median) {
  int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  int32_t r = rank[i];
  if (r < 0) {
    o_slot[i] = 0;
    o_bnum[i] = 0;
    o_bcoord[i] = 0;
    o_median[i] = 0;
    return;
  }
  int32_t pos = offs[gidx[i]] + r;
  seg_a[pos] = mk4(is_stop ? (int32_t)(is_stop[i] & 1) : 0, 0, 0, i);
}

/* ------------------------------------------------------------------------- */
/* long segments: list them, then sort (arrival idx, pos) per segment           */
/* one workgroup per long segment: ord[base + j] = key of the j-th record in arrival order,
 * key = (idx << 32) | pos.  Bitonic network in its all-ascending form (first stage of every
 * merge compares t with its mirror t ^ (k-1), the rest with t ^ j): positions >= c behave as
 * +inf simply by being skipped.  In LDS when the segment fits, else in global memory (correct
 * but slow: one hot group is serial under the per-group ordering contract anyway). */
__device__ __forceinline__ void cmpxchg_asc(unsigned long long* a, uint32_t lo, uint32_t hi) {
  unsigned long long x = a[lo], y = a[hi];
  if (x > y) {
    a[lo] = y;
    a[hi] = x;
  }
}
__global__ __launch_bounds__(GPX_BLOCK) void k_sort_big(const int32_t* __restrict__ biglist,
                                                       const int32_t* __restrict__ cnt,
                                                       const int32_t* __restrict__ offs,
                                                       const I4* __restrict__ seg_a,
                                                       unsigned long long* ord) {
  __shared__ unsigned long long lds[GPX_SORT_LDS_MAX];
  const int32_t nbig = biglist[0];
  for (int32_t b = blockIdx.x; b < nbig; b += gridDim.x) {
    const int32_t g = biglist[1 + b];
    const uint32_t c = (uint32_t)cnt[g];
    const int32_t base = offs[g];
    uint32_t p2 = 1;
    while (p2 < c) p2 <<= 1;
    const bool in_lds = c <= GPX_SORT_LDS_MAX;
    unsigned long long* a = in_lds ? lds : (ord + base);
    for (uint32_t j = threadIdx.x; j < c; j += GPX_BLOCK)
      a[j] = ((unsigned long long)(uint32_t)seg_a[base + j].w << 32) | (uint32_t)(base + j);
    __syncthreads();
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
    if (in_lds)
      for (uint32_t j = threadIdx.x; j < c; j += GPX_BLOCK) ord[base + j] = lds[j];
    __syncthreads();
  }
}

/* Iterates one group's records in arrival order.  Short segments: repeated min-scan over
 * the arrival indices (c <= GPX_SMALL_SEG, typically 1..5).  Long: the sorted `ord`. */
struct SegIter {
  const I4* seg_a;
  const unsigned long long* ord;
  int32_t base, c, done;
  int32_t last; /* last arrival idx consumed */
  __device__ __forceinline__ void init(const I4* a, const unsigned long long* o, int32_t b,
                                       int32_t n) {
    seg_a = a;
    ord = o;
    base = b;
    c = n;
    done = 0;
    last = -1;
  }
  /* returns position in seg arrays of the next record, or -1 */
  __device__ __forceinline__ int32_t next() {
    if (done >= c) return -1;
    int32_t pos;
    if (c == 1) {
      pos = base;
    } else if (c <= GPX_SMALL_SEG) {
      int32_t best = 0x7fffffff, bp = -1;
      for (int32_t j = 0; j < c; j++) {
        int32_t ix = seg_a[base + j].w;
        if (ix > last && ix < best) {
          best = ix;
          bp = base + j;
        }
      }
      last = best;
      pos = bp;
    } else {
      pos = (int32_t)(uint32_t)(ord[base + done] & 0xffffffffull);
    }
    done++;
    return pos;
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
/* k_apply_ar: the coordinator side — one lane per group.                       */
/* PaxosInstanceStateMachine.handleAcceptReply (PISM:1248-1364) ->              */
/* PaxosCoordinator.handleAcceptReply (PaxosCoordinator.java:210-250) ->        */
/* PaxosCoordinatorState.handleAcceptReplyMyBallot / HigherBallot (:597-683)    */
template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_apply_ar(DevState S, DevScratch X,
                                                       uint8_t* __restrict__ status) {
  const int32_t g = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (g >= S.G) return;
  const int32_t c = X.cnt[g];
  if (c == 0) return;
  X.cnt[g] = 0;
  const int32_t base = X.offs[g];
  const int32_t G = S.G;
  uint32_t gf = S.g_flags[g];
  SegIter it;
  it.init(X.seg_a, X.ord, base, c);
  if (!(gf & GF_EXISTS) || (gf & GF_STOPPED)) {
    /* PaxosManager.java:1162-1194 / PaxosInstanceStateMachine.java:456-460: dropped */
    const uint8_t st = (gf & GF_EXISTS) ? GPX_S_STOPPED : GPX_S_NOGROUP;
    for (int32_t j = 0; j < c; j++) {
      int32_t ix = X.seg_a[base + j].w;
      if (status) status[ix] = st;
    }
    atomicAdd(&X.counters[2], (unsigned long long)c); /* rare path */
    return;
  }
  const int32_t k = (int32_t)GF_K(gf);
  bool has_coord = (gf & GF_HASCOORD) != 0;
  const int32_t my_bnum = S.c_bnum[g], my_bcoord = S.c_bcoord[g];
  const int32_t next = S.c_next[g];
  int32_t pcount = S.c_pcount[g];
  const int32_t Wm = S.W - 1;
  int32_t mem[KMAX], ns[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; j++) {
    mem[j] = (j < k) ? S.members[(int64_t)j * G + g] : 0;
    ns[j] = (j < k) ? S.node_slots[(int64_t)j * G + g] : 0;
  }
  bool ns_dirty = false;
  for (int32_t pos = it.next(); pos >= 0; pos = it.next()) {
    const I4 ra = X.seg_a[pos];
    const I2 rb = X.seg_b[pos];
    const int32_t slot = ra.x, acc = ra.y, maxcp = ra.z, ix = ra.w;
    if (!has_coord) continue; /* PaxosCoordinator.java:196-198: c == null -> null */
    const int32_t cmp = ballot_cmp(rb.x, rb.y, my_bnum, my_bcoord);
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
      for (int j = 0; j < KMAX; j++) {
        if (j < k && mem[j] == acc) {
          midx = j; /* WaitforUtility.getIndex: last match */
          if (ns[j] < maxcp) {
            ns[j] = maxcp;
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
    for (int j = 0; j < KMAX; j++)
      if (j < k) S.node_slots[(int64_t)j * G + g] = ns[j];
  }
  S.c_pcount[g] = pcount;
  if (!has_coord && (gf & GF_HASCOORD)) S.g_flags[g] = gf & ~GF_HASCOORD;
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
 * gfx950) once inlined into k_apply_accept: after CFG structurization the median of the
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

/* k_apply_accept: PaxosInstanceStateMachine.handleAccept (PISM:1080-1166) */
__global__ __launch_bounds__(GPX_BLOCK) void k_apply_accept(
    DevState S, DevScratch X, int32_t* __restrict__ r_bnum, int32_t* __restrict__ r_bcoord,
    int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags, uint8_t* __restrict__ status) {
  const int32_t g = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (g >= S.G) return;
  const int32_t c = X.cnt[g];
  if (c == 0) return;
  X.cnt[g] = 0;
  const int32_t base = X.offs[g];
  const uint32_t gf = S.g_flags[g];
  SegIter it;
  it.init(X.seg_a, X.ord, base, c);
  AccState a;
  a.stopped = false;
  const bool exists = (gf & GF_EXISTS) != 0;
  if (exists) acc_load(S, g, gf, a);
  const int32_t Wm = S.W - 1;
  unsigned long long n_drop = 0;
  for (int32_t pos = it.next(); pos >= 0; pos = it.next()) {
    const I4 ra = X.seg_a[pos];
    const I2 rb = X.seg_b[pos];
    const int32_t slot = ra.x, median = ra.y, ix = ra.w;
    const bool stop = (ra.z & GPX_A_STOP) != 0;
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
    const bool ballot_ok = ballot_cmp(rb.x, rb.y, a.bnum, a.bcoord) >= 0;
    const bool will_store = ballot_ok && jsub(slot, a.gc) > 0;
    if (will_store && live && ar.x != slot) {
      status[ix] = GPX_S_WINDOW; /* ring slot held by another live accepted slot */
      n_drop++;
      continue;
    }
    if (ballot_ok) {
      a.bnum = rb.x;
      a.bcoord = rb.y;
      if (will_store) {
        S.acc_ring[o] = mk4(slot, rb.x, rb.y, 0);
        S.acc_flags[o] = (uint8_t)(RF_PRESENT | (stop ? RF_STOP : 0));
      }
    }
    acc_gc(S, g, a, median);
    /* reply (myID, ballot, slot, getSlot()-1)  (:1139-1143) */
    r_bnum[ix] = a.bnum;
    r_bcoord[ix] = a.bcoord;
    r_maxcp[ix] = jsub(a.slot, 1);
    /* toLog (:1146-1149) */
    const bool to_log = ballot_cmp(rb.x, rb.y, a.bnum, a.bcoord) >= 0 && jsub(slot, a.gc) > 0 &&
                        (!have_prev || ballot_cmp(ar.y, ar.z, rb.x, rb.y) < 0);
    r_flags[ix] = (uint8_t)((to_log ? GPX_R_TOLOG : 0) | (will_store ? GPX_R_STORED : 0));
    status[ix] = GPX_S_OK;
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

/* k_apply_commit: PaxosInstanceStateMachine.handleBatchedCommit (PISM:1480-1528) per slot and
 * handleCommittedRequest (:1432-1478) for full decisions */
__global__ __launch_bounds__(GPX_BLOCK) void k_apply_commit(DevState S, DevScratch X,
                                                           uint8_t* __restrict__ status) {
  const int32_t g = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (g >= S.G) return;
  const int32_t c = X.cnt[g];
  if (c == 0) return;
  X.cnt[g] = 0;
  const int32_t base = X.offs[g];
  const uint32_t gf = S.g_flags[g];
  SegIter it;
  it.init(X.seg_a, X.ord, base, c);
  AccState a;
  a.stopped = false;
  const bool exists = (gf & GF_EXISTS) != 0;
  if (exists) acc_load(S, g, gf, a);
  const int32_t Wm = S.W - 1;
  unsigned long long n_drop = 0;
  for (int32_t pos = it.next(); pos >= 0; pos = it.next()) {
    const I4 ra = X.seg_a[pos];
    const I2 rb = X.seg_b[pos];
    const int32_t slot = ra.x, median = ra.y, kind = ra.z, ix = ra.w;
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
    status[ix] = GPX_S_OK;
    Dec d = Dec{0, 0, 0, 0, false, false};
    if (kind & GPX_C_HASVALUE) {
      d = Dec{rb.x, rb.y, slot, median, true, (kind & GPX_C_STOP) != 0};
    } else {
      /* accept != null && accept.ballot.equals(batchedCommit.ballot) (:1492) */
      const int64_t o = (int64_t)(slot & Wm) * S.G + g;
      const uint8_t af = S.acc_flags[o];
      const I4 ar = S.acc_ring[o];
      if ((af & RF_PRESENT) && ar.x == slot && ballot_cmp(ar.y, ar.z, rb.x, rb.y) == 0)
        d = Dec{ar.y, ar.z, slot, median, true, (af & RF_STOP) != 0};
      else
        d = Dec{rb.x, rb.y, slot, median, false, false}; /* placeholder (:1510-1520) */
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

/* k_apply_propose: PaxosInstanceStateMachine.handleProposal (PISM:818-888) ->
 * PaxosCoordinatorState.propose (:233-263) + initCommander (:841-851) */
template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_apply_propose(
    DevState S, DevScratch X, int32_t* __restrict__ o_slot, int32_t* __restrict__ o_bnum,
    int32_t* __restrict__ o_bcoord, int32_t* __restrict__ o_median, uint8_t* __restrict__ status) {
  const int32_t g = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (g >= S.G) return;
  const int32_t c = X.cnt[g];
  if (c == 0) return;
  X.cnt[g] = 0;
  const int32_t base = X.offs[g];
  const int32_t G = S.G;
  const uint32_t gf = S.g_flags[g];
  SegIter it;
  it.init(X.seg_a, X.ord, base, c);
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
  for (int j = 0; j < KMAX; j++) ns[j] = (coord_ok && j < k) ? S.node_slots[(int64_t)j * G + g] : 0;
  const int32_t median = coord_ok ? median_minus<KMAX>(ns, k) : 0;
  const int32_t Wm = S.W - 1;
  unsigned long long n_drop = 0;
  for (int32_t pos = it.next(); pos >= 0; pos = it.next()) {
    const I4 ra = X.seg_a[pos];
    const int32_t ix = ra.w;
    const bool stop = ra.x != 0;
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
    status[ix] = GPX_S_OK;
    next = (int32_t)((uint32_t)next + 1u);
  }
  if (coord_ok) {
    S.c_next[g] = next;
    S.c_pcount[g] = pcount;
  }
  if (n_drop) atomicAdd(&X.counters[2], n_drop);
}

/* ------------------------------------------------------------------------- */
/* ordered compaction of the per-record outputs (phase 3 of the flag scan)      */

/* Both compaction kernels first build the tile's list of flagged record indices in LDS (in
 * arrival order, via the same blocked scan as phase 1), then thread t gathers the t-th flagged
 * record and writes output row base+t: every output column is written as a dense coalesced run
 * (the direct form, each thread storing its own ex++ rows, cost 5x sector write amplification in
 * the round-1 profile). */
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
