/*
 * gpx_ar16.hip.h — the accept-reply back end on 16-byte vote records (round 2).
 *
 * Same pipeline shape as gpx_kernels.hip.h (k_hist -> scatter -> one workgroup per bucket, one
 * lane per group -> emit), with half the bytes through the partition:
 *
 *   Vote16 {idx, slot, max_cp, meta}   meta = local group (14 bits) | ESC | acceptor << 16
 *
 * A vote's ballot is almost always the batch's common ballot (the ballot of record 0) and node ids
 * almost always fit 16 bits: such a vote travels without ballot and with the acceptor inlined.
 * Anything else sets ESC and the per-bucket kernel re-reads (bnum, bcoord, acceptor) of that vote
 * from the caller's columns by arrival index - the columns are valid for the whole call - so the
 * compact form loses nothing.
 *
 *   k_scatter_ar16   per tile: one 16-byte store per vote into its bucket region
 *   k_bucket_ar16    per bucket: count per group (LDS atomics) -> scan -> votes placed GROUP-MAJOR
 *                    in LDS (structure of arrays): lane l's votes are consecutive words, so the
 *                    replay reads LDS at a near-constant stride across lanes instead of at random
 *                    slots.  The arrival order of a group's <= 16 votes is a 64-bit nibble word in
 *                    a register; longer segments are sorted cooperatively through global scratch.
 *                    The replay itself is apply_ar_group (gpx_kernels.hip.h), unchanged: PISM.
 *                    handleAcceptReply -> PaxosCoordinatorState.handleAcceptReplyMyBallot /
 *                    HigherBallot (PaxosInstanceStateMachine.java:1248-1419, PCS:597-683).
 *                    Outputs are staged as six dense COLUMNS per bucket (21 bytes per decision).
 *   k_emit_dec16     per bucket: the staged columns -> the caller's columns, buckets in order.
 */
#pragma once
#include "gpx_kernels.hip.h"
#include "gpx_tiles.hip.h"

struct __attribute__((aligned(16))) Vote16 {
  int32_t idx, slot, maxcp;
  uint32_t meta;
};
#define V16_LG_MASK 0x3fffu
#define V16_ESC 0x4000u
#define V16_MAX_SHIFT 10 /* one lane per group, at most 1024 lanes */
/* dynamic LDS the tiled per-bucket kernels may take: the CU's 160 KB less their static block (run starts and prefix of
 * up to 1024 tiles 6.1 KB, scan scratch) */
#define GPX_TL_BUCKET_DYN_MAX (152 * 1024)

/* staged outputs of the per-bucket kernel: six columns over the record index space (outputs of
 * bucket b at [bucket_off[b], bucket_off[b] + bucket_nout[b])) */
struct Stage16 { /* one block of six columns n apart (five of int32, one of bytes): ONE pointer in the kernel's
                  * argument registers instead of six - the per-bucket kernel is short of SGPRs */
  int32_t* base;
  int64_t n;
  __host__ __device__ __forceinline__ int32_t* gidx() const { return base; }
  __host__ __device__ __forceinline__ int32_t* slot() const { return base + n; }
  __host__ __device__ __forceinline__ int32_t* bnum() const { return base + 2 * n; }
  __host__ __device__ __forceinline__ int32_t* bcoord() const { return base + 3 * n; }
  __host__ __device__ __forceinline__ int32_t* median() const { return base + 4 * n; }
  __host__ __device__ __forceinline__ uint8_t* kind() const { return (uint8_t*)(base + 5 * n); }
};
/* the caller's vote columns the ESC path reads, and the batch's common ballot = ballot of vote 0 */
struct VoteCols {
  const int32_t *bnum, *bcoord, *acceptor;
  const int32_t *slot, *maxcp; /* the tiled front end's escape path (gpx_tiles.hip.h) */
};
/* The caller's output columns, for the per-bucket kernel of the tiled front end to write IN PLACE (round 6).  A bucket's
 * place among the outputs is only known once every bucket before it has replayed - but in a coordinator's steady state
 * every D votes make one output (D = the replicas that answer), so bucket b with `boff` records before it and `nb` of its
 * own PREDICTS the span [boff / D, (boff + nb) / D) (the spans of all buckets telescope to a dense column).  A bucket whose
 * output count IS its span writes there as well as to the staging; one whose count differs raises the call's epoch in
 * `A.ref[3]`, and k_emit_dec16 then compacts the staging as before.  No bucket differing = the outputs are in place and
 * k_emit_dec16 only publishes the total: the 63 MB staging round trip is a 21 MB second store.
 * D is the engine's (a kernel argument, with its reciprocal): k_emit_dec16 leaves votes / outputs of a call that had to be
 * compacted in a host-mapped word, the host reads that word - however stale: it is only a prediction - at its next call. */
struct PlaceCols {
  int32_t *gidx, *slot, *bnum, *bcoord, *median;
  uint8_t* kind;
  int32_t div;  /* D; 0: staging only (GPX_AR_INPLACE=0; the partition front end) */
  uint32_t mul; /* floor(2^32 / D) + 1: x / D == umulhi(x, mul) for x < 2^26, D <= 64 (D == 1: x itself) */
  __device__ __forceinline__ int32_t quot(int32_t x) const { return div == 1 ? x : (int32_t)__umulhi((uint32_t)x, mul); }
};
#define GPX_IP_LEARN_WORD 8 /* of the host-mapped block X.xabort points at */

__device__ __forceinline__ void put_vote16(const DevScratch& X, int32_t* lds, int32_t G, int32_t mask,
                                           int32_t b0n, int32_t b0c, int64_t i, int32_t g, int32_t slot,
                                           int32_t acc, int32_t maxcp, int32_t bn, int32_t bc) {
  /* outside this pass's group range: another pass takes it, or (outside the table) the status already
   * says GPX_S_NOGROUP (k_hist) */
  if ((uint32_t)(g - X.g_base) >= (uint32_t)(X.g_end - X.g_base)) return;
  (void)G;
  const int32_t pos = bucket_take(lds, (g - X.g_base) >> X.shift);
  const bool esc = bn != b0n || bc != b0c || (uint32_t)acc > 0xffffu;
  Vote16 v;
  v.idx = (int32_t)i;
  v.slot = slot;
  v.maxcp = maxcp;
  v.meta = (uint32_t)(g & mask) | (esc ? V16_ESC : ((uint32_t)acc << 16));
  ((Vote16*)X.rec)[pos] = v; /* one 16-byte request per vote */
}

template <bool VEC>
__global__ __launch_bounds__(GPX_FBLOCK) void k_scatter_ar16(
    int32_t n, int32_t ntiles, int32_t G, DevScratch X, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ bnum, const int32_t* __restrict__ bcoord,
    const int32_t* __restrict__ slot, const int32_t* __restrict__ acceptor,
    const int32_t* __restrict__ max_cp) {
  extern __shared__ int32_t lds[];
  const int32_t tile = tile_of_block(ntiles);
  if (tile >= ntiles) return;
  if (X.gate && *X.unsorted != X.epoch) return; /* a few sorted runs: k_ar_runs did it (gpx_runs.hip.h) */
  const int32_t b0n = bnum[0], b0c = bcoord[0];
  scatter_init(X, tile, lds);
  const int64_t base = (int64_t)tile * GPX_TILE;
  const int32_t mask = X.gb - 1;
  if (VEC) {
#pragma unroll
    for (int j = 0; j < GPX_TILE_VECS; j++) {
      const int64_t i0 = base + (int64_t)(j * GPX_FBLOCK + threadIdx.x) * 4;
      if (i0 + 3 < n) {
        const I4 g4 = *(const I4*)(gidx + i0), s4 = *(const I4*)(slot + i0);
        const I4 a4 = *(const I4*)(acceptor + i0), m4 = *(const I4*)(max_cp + i0);
        const I4 n4 = *(const I4*)(bnum + i0), c4 = *(const I4*)(bcoord + i0);
        put_vote16(X, lds, G, mask, b0n, b0c, i0 + 0, g4.x, s4.x, a4.x, m4.x, n4.x, c4.x);
        put_vote16(X, lds, G, mask, b0n, b0c, i0 + 1, g4.y, s4.y, a4.y, m4.y, n4.y, c4.y);
        put_vote16(X, lds, G, mask, b0n, b0c, i0 + 2, g4.z, s4.z, a4.z, m4.z, n4.z, c4.z);
        put_vote16(X, lds, G, mask, b0n, b0c, i0 + 3, g4.w, s4.w, a4.w, m4.w, n4.w, c4.w);
      } else {
        for (int q = 0; q < 4; q++) {
          const int64_t i = i0 + q;
          if (i < n)
            put_vote16(X, lds, G, mask, b0n, b0c, i, gidx[i], slot[i], acceptor[i], max_cp[i], bnum[i],
                       bcoord[i]);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < GPX_TILE_ITEMS; j++) {
      const int64_t i = base + j * GPX_FBLOCK + threadIdx.x;
      if (i < n)
        put_vote16(X, lds, G, mask, b0n, b0c, i, gidx[i], slot[i], acceptor[i], max_cp[i], bnum[i],
                   bcoord[i]);
    }
  }
}

/* ACCEPT / COMMIT records in the same 16 bytes: {idx, slot, median_cp, local group | ESC | flags << 16};
 * the dense reply columns of dropped ACCEPTs are zeroed here, like k_scatter_ac does */
__device__ __forceinline__ void put_ac16(const DevScratch& X, int32_t* lds, int32_t G, int32_t mask, int32_t b0n,
                                         int32_t b0c, int64_t i, int32_t g, int32_t slot, int32_t median,
                                         int32_t flags, int32_t bn, int32_t bc, int32_t* __restrict__ r_bnum,
                                         int32_t* __restrict__ r_bcoord, int32_t* __restrict__ r_maxcp,
                                         uint8_t* __restrict__ r_flags, I4* __restrict__ r_packed) {
  if ((uint32_t)g >= (uint32_t)G) {
    if (r_packed) {
      r_packed[i] = mk4(0, 0, 0, 0);
    } else if (r_bnum) {
      r_bnum[i] = 0;
      r_bcoord[i] = 0;
      r_maxcp[i] = 0;
      r_flags[i] = 0;
    }
    return;
  }
  const int32_t pos = bucket_take(lds, g >> X.shift);
  Vote16 v;
  v.idx = (int32_t)i;
  v.slot = slot;
  v.maxcp = median;
  v.meta = (uint32_t)(g & mask) | ((bn != b0n || bc != b0c) ? V16_ESC : 0u) | ((uint32_t)(flags & 0xff) << 16);
  ((Vote16*)X.rec)[pos] = v;
}
template <bool VEC>
__global__ __launch_bounds__(GPX_FBLOCK) void k_scatter_ac16(
    int32_t n, int32_t ntiles, int32_t G, DevScratch X, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ bnum, const int32_t* __restrict__ bcoord, const int32_t* __restrict__ slot,
    const int32_t* __restrict__ median_cp, const uint8_t* __restrict__ flags, int32_t* __restrict__ r_bnum,
    int32_t* __restrict__ r_bcoord, int32_t* __restrict__ r_maxcp, uint8_t* __restrict__ r_flags,
    I4* __restrict__ r_packed) {
  extern __shared__ int32_t lds[];
  const int32_t tile = tile_of_block(ntiles);
  if (tile >= ntiles) return;
  if (*X.unsorted != X.epoch) return; /* ordered batch: applied directly (gpx_direct.hip.h) */
  const int32_t b0n = bnum[0], b0c = bcoord[0];
  scatter_init(X, tile, lds);
  const int64_t base = (int64_t)tile * GPX_TILE;
  const int32_t mask = X.gb - 1;
  if (VEC) {
#pragma unroll
    for (int j = 0; j < GPX_TILE_VECS; j++) {
      const int64_t i0 = base + (int64_t)(j * GPX_FBLOCK + threadIdx.x) * 4;
      if (i0 + 3 < n) {
        const I4 g4 = *(const I4*)(gidx + i0), s4 = *(const I4*)(slot + i0), m4 = *(const I4*)(median_cp + i0);
        const I4 n4 = *(const I4*)(bnum + i0), c4 = *(const I4*)(bcoord + i0);
        const uint32_t f4 = flags ? *(const uint32_t*)(flags + i0) : 0u;
        put_ac16(X, lds, G, mask, b0n, b0c, i0 + 0, g4.x, s4.x, m4.x, (int32_t)(f4 & 0xffu), n4.x, c4.x, r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
        put_ac16(X, lds, G, mask, b0n, b0c, i0 + 1, g4.y, s4.y, m4.y, (int32_t)((f4 >> 8) & 0xffu), n4.y, c4.y, r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
        put_ac16(X, lds, G, mask, b0n, b0c, i0 + 2, g4.z, s4.z, m4.z, (int32_t)((f4 >> 16) & 0xffu), n4.z, c4.z, r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
        put_ac16(X, lds, G, mask, b0n, b0c, i0 + 3, g4.w, s4.w, m4.w, (int32_t)(f4 >> 24), n4.w, c4.w, r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
      } else {
        for (int q = 0; q < 4; q++) {
          const int64_t i = i0 + q;
          if (i < n)
            put_ac16(X, lds, G, mask, b0n, b0c, i, gidx[i], slot[i], median_cp[i], flags ? (int32_t)flags[i] : 0,
                     bnum[i], bcoord[i], r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < GPX_TILE_ITEMS; j++) {
      const int64_t i = base + j * GPX_FBLOCK + threadIdx.x;
      if (i < n)
        put_ac16(X, lds, G, mask, b0n, b0c, i, gidx[i], slot[i], median_cp[i], flags ? (int32_t)flags[i] : 0,
                 bnum[i], bcoord[i], r_bnum, r_bcoord, r_maxcp, r_flags, r_packed);
    }
  }
}

/* dynamic LDS of k_bucket_ar16: lcnt[gb] | lcur[gb] | idx[L] | slot[L] | maxcp[L] | meta[L] */
#define GPX_BUCKET16_LDS_BYTES(gb, lds_recs) ((size_t)(gb) * 8 + (size_t)(lds_recs) * 16)
#define V16_NIB_MAX 16 /* a group's arrival order fits a 64-bit nibble word up to this many votes */
#define V16_LANE_SORT 96 /* longer LDS-staged segments, up to this many votes, are ordered by their own lane */

/* One group's votes in arrival order, with GroupIter's interface (next / emit / c) so that
 * apply_ar_group replays them unchanged.
 *   LDSM: the bucket is staged in LDS, structure of arrays, this group's votes at words
 *         [start, start + c); otherwise the votes are the Vote16 records in global memory (a bucket
 *         beyond the LDS staging capacity) and `start` indexes the bucket's key array.
 *   c <= 16 (and LDSM): arrival order = nibble word `order` (nibble d = position of the vote with
 *         arrival rank d); a vote that produced an output is remembered in `omask` (bit d) and the
 *         output parked in that vote's own words.
 *   otherwise: keys (arrival idx << 32 | position) sorted ascending in global scratch; the
 *         positions of votes with outputs are listed in keys[0 .. nout). */
template <bool LDSM, bool AC = false>
struct VoteIter {
  const int32_t* idxA;  /* LDS arrays (LDSM) */
  int32_t *slotA, *cpA;
  uint32_t* metaA;
  Vote16* recG;              /* global records of the bucket (!LDSM) */
  unsigned long long* keys;  /* sorted keys of this group (long / global mode) */
  VoteCols in;
  int32_t b0n, b0c;
  int32_t start, c, done, nout;
  int32_t stride; /* LDSM + nib: the group's vote of nibble t sits at start + t * stride (1: packed group-major; the
                   * bucket's lane count: one row per vote rank, gpx_ar16.hip.h "slotted placement") */
  bool nib;
  unsigned long long order;
  uint32_t omask;
  uint32_t cur;
  __device__ __forceinline__ bool next(Rec& out) {
    if (done >= c) return false;
    uint32_t p;
    int32_t ix, sl, cp;
    uint32_t meta;
    if (LDSM) {
      if (nib)
        p = (uint32_t)start + (uint32_t)stride * (uint32_t)((order >> (4 * done)) & 15ull);
      else
        p = (uint32_t)keys[done];
      ix = idxA[p];
      sl = slotA[p];
      cp = cpA[p];
      meta = metaA[p];
    } else {
      p = (uint32_t)keys[done];
      const Vote16 v = recG[p];
      ix = v.idx;
      sl = v.slot;
      cp = v.maxcp;
      meta = v.meta;
    }
    cur = p;
    out.idx = ix;
    out.a = sl;
    if (AC) { /* ACCEPT / COMMIT record: {slot, median_cp, flags byte}; ESC = another ballot */
      out.b = cp;
      out.c = (int32_t)((meta >> 16) & 0xffu);
      const bool esc = (meta & V16_ESC) != 0;
      out.bnum = esc ? in.bnum[ix] : b0n;
      out.bcoord = esc ? in.bcoord[ix] : b0c;
    } else {
      out.c = cp;
      if (meta & V16_ESC) {
        out.b = in.acceptor[ix];
        out.bnum = in.bnum[ix];
        out.bcoord = in.bcoord[ix];
      } else {
        out.b = (int32_t)(meta >> 16);
        out.bnum = b0n;
        out.bcoord = b0c;
      }
    }
    done++;
    return true;
  }
  /* output of the CURRENT vote: (slot, median, kind) parked in the vote's own words; the ballot
   * of a decision / preemption is the coordinator's own (x, y), constant per group per call */
  __device__ __forceinline__ void emit(int32_t slot, int32_t x, int32_t y, int32_t z, int32_t kind) {
    if (AC) { /* an execution run (first, count): apply_accept_group / apply_commit_group emit(0, first, count, 0, 1) */
      slot = x;
      z = y;
    }
    if (LDSM) {
      slotA[cur] = slot;
      cpA[cur] = z;
      metaA[cur] = (uint32_t)kind;
      if (nib)
        omask |= 1u << (done - 1);
      else
        keys[nout] = cur; /* entry nout <= done - 1: consumed */
    } else {
      recG[cur].slot = slot;
      recG[cur].maxcp = z;
      recG[cur].meta = (uint32_t)kind;
      keys[nout] = cur;
    }
    nout++;
  }
  /* q-th output of the group (after the replay) */
  __device__ __forceinline__ void output(int32_t q, int32_t* slot, int32_t* median, int32_t* kind,
                                         uint32_t* om) const {
    uint32_t p;
    if (LDSM && nib) {
      const int d = __ffs((int)*om) - 1; /* arrival rank of the next vote with an output */
      *om &= *om - 1;
      p = (uint32_t)start + (uint32_t)stride * (uint32_t)((order >> (4 * d)) & 15ull);
    } else {
      p = (uint32_t)keys[q];
    }
    if (LDSM) {
      *slot = slotA[p];
      *median = cpA[p];
      *kind = (int32_t)metaA[p];
    } else {
      *slot = recG[p].slot;
      *median = recG[p].maxcp;
      *kind = (int32_t)recG[p].meta;
    }
  }
};

/* arrival order of c <= 16 votes at idxA[start ..): nibble d = position of the vote with rank d */
__device__ __forceinline__ unsigned long long arrival_order(const int32_t* idxA, int32_t start, int32_t c, int32_t stride = 1) {
  unsigned long long order = 0;
  if (c <= 4) {
    const uint32_t inf = 0xffffffffu;
    const uint32_t i0 = (uint32_t)idxA[start];
    const uint32_t i1 = c > 1 ? (uint32_t)idxA[start + stride] : inf;
    const uint32_t i2 = c > 2 ? (uint32_t)idxA[start + 2 * stride] : inf;
    const uint32_t i3 = c > 3 ? (uint32_t)idxA[start + 3 * stride] : inf;
    /* arrival indices are distinct; absent entries (inf) rank last and are never read */
    const uint32_t r0 = (i1 < i0) + (i2 < i0) + (i3 < i0);
    const uint32_t r1 = (i0 < i1) + (i2 < i1) + (i3 < i1);
    const uint32_t r2 = (i0 < i2) + (i1 < i2) + (i3 < i2);
    const uint32_t r3 = 6u - r0 - r1 - r2;
    const uint32_t o = (1u << (4 * r1)) | (2u << (4 * r2)) | (3u << (4 * r3)); /* 0 << (4 * r0) */
    order = o & 0xffffu;
  } else {
    for (int32_t t = 0; t < c; t++) {
      const uint32_t it = (uint32_t)idxA[start + t * stride];
      int32_t r = 0;
      for (int32_t u = 0; u < c; u++) r += (uint32_t)idxA[start + u * stride] < it;
      order |= (unsigned long long)t << (4 * r);
    }
  }
  return order;
}

/* (Forcing 8 or 7 waves per SIMD through amdgpu_waves_per_eu - 64 VGPRs and 60 bytes of spills - is slower:
 * 53.8 / 51.6 us against 47-49 us when it was measured; the kernel is issue-bound, docs/HISTORY.md section 7.) */
#ifdef GPX_AR16_WAVES
#define GPX_AR16_ATTR __attribute__((amdgpu_waves_per_eu(GPX_AR16_WAVES, 8)))
#else
#define GPX_AR16_ATTR
#endif
/* dense per-record outputs of the ACCEPT call (gpx_accept_batch: the ACCEPT_REPLY columns) */
struct AcceptOut {
  int32_t *r_bnum, *r_bcoord, *r_maxcp;
  uint8_t* r_flags;
  I4* r_packed; /* [n] scratch rows: set for the partition path (records at random arrival indices) */
};
#define B16_AR 0     /* accept replies at the coordinator */
#define B16_ACCEPT 1 /* ACCEPTs at an acceptor */
#define B16_COMMIT 2 /* commits at every replica */
/* The replay of ONE group whose votes are staged in LDS (the group's lane calls it): arrival order, then the straight
 * line when the wave's groups all allow it, else apply_ar_group / apply_accept_group / apply_commit_group.  Shared by the
 * per-bucket kernels (bucket16_body) and the pipelined one (k_bucket_ar16_tiles_pipe). */
template <int OP, int KMAX, class IT>
__device__ __forceinline__ void ar16_replay_group(const DevState& S, const DevScratch& X, int32_t g, IT& it, CoordPre<KMAX>& P,
                                                  const VoteCols& in, const AcceptOut& R, uint8_t* __restrict__ status,
                                                  const int32_t* idxA, int32_t* slotA, int32_t* cpA, uint32_t* metaA,
                                                  int32_t start, int32_t pstride, int32_t c) {
  auto replay = [&](auto& it_) {
    if (OP == B16_AR)
      apply_ar_group<KMAX>(S, X, g, it_, status, P);
    else if (OP == B16_ACCEPT)
      apply_accept_group(S, X, g, it_, R.r_bnum, R.r_bcoord, R.r_maxcp, R.r_flags, status, R.r_packed);
    else
      apply_commit_group(S, X, g, it_, status);
  };
  if (it.nib) it.order = arrival_order(idxA, start, c, pstride);
  /* Steady state of a coordinator: every vote of the group answers ONE outstanding slot at the
   * group's current ballot (which is also the batch's common ballot).  When that holds for every
   * group of the wave, the replay is a straight line per vote - member bit, nodeSlotNumbers max,
   * majority test - with exactly the effects of apply_ar_group's cmp == 0 branch
   * (handleAcceptReplyMyBallot, PCS:597-640); votes of a lower ballot are stepped over, as the
   * reference ignores them.  One group that needs anything else (a HIGHER ballot, a second slot, no
   * coordinator, a view change in progress, more than eight votes) sends the whole wave down the
   * general path. */
  bool fast = false;
  int32_t s0 = 0;
  uint32_t skip = 0; /* votes of a LOWER ballot: ignored (PaxosCoordinator.java:241-247), whatever their slot */
  int32_t nvote = c; /* votes the straight-line replay walks */
  bool esc = false;  /* this group has an escaped vote */
  if (OP == B16_AR) {
    /* a group that coordinates nothing (preempted, or never the coordinator) ignores every vote
     * (PaxosCoordinator.java:196-198: c == null) */
    const bool idle = (P.gf & (GF_EXISTS | GF_STOPPED | GF_HASCOORD)) == GF_EXISTS;
    bool el = c <= 8 && (P.gf & (GF_EXISTS | GF_STOPPED | GF_HASCOORD | GF_PREPARING)) == (GF_EXISTS | GF_HASCOORD) &&
              it.b0n == P.my_bnum && it.b0c == P.my_bcoord;
    if (idle) {
      el = true;
      nvote = 0;
    } else if (el) {
      bool have = false;
      for (int32_t i = 0; i < c; i++) {
        const uint32_t p = (uint32_t)start + (uint32_t)pstride * (uint32_t)((it.order >> (4 * i)) & 15ull);
        if (metaA[p] & V16_ESC) { /* another ballot than the batch's, or a node id beyond 16 bits */
          esc = true;
          const int32_t ix = idxA[p];
          const int32_t cmp = ballot_cmp(in.bnum[ix], in.bcoord[ix], P.my_bnum, P.my_bcoord);
          if (cmp < 0) {
            skip |= 1u << i;
            continue;
          }
          el = el && cmp == 0; /* a higher ballot preempts: general path */
        }
        const int32_t sl = slotA[p];
        if (!have) s0 = sl;
        have = true;
        el = el && sl == s0;
      }
      if (have) {
        const int32_t d = jsub(P.next, s0);
        el = el && d >= 1 && d <= S.W;
      }
    }
    fast = __all(el);
  }
  if (fast) {
    if (OP == B16_AR) {
      const int32_t G = S.G, k = (int32_t)GF_K(P.gf);
      int32_t mem[KMAX], ns[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; j++) {
        mem[j] = (j < k) ? P.mem[j] : 0;
        ns[j] = (j < k) ? P.ns[j] : 0;
      }
      const int64_t off = (int64_t)(s0 & (S.W - 1)) * G + g;
      const uint32_t e0 = (P.have_pe && s0 == jsub(P.next, 1)) ? P.pe : S.p_ring[off];
      uint32_t e = e0;
      int32_t pcount = P.pcount;
      bool ns_dirty = false;
      auto vote = [&](int32_t i, uint32_t p, int32_t acc, int32_t maxcp) {
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
          if (midx >= 0) e |= (1u << midx);       /* updateHeardFrom :51-62 */
          if (__popc(e & 0xffffu) > k / 2) {     /* heardFromMajority :64-68 */
            slotA[p] = s0;                       /* the decision, parked in the vote's own words */
            cpA[p] = median_minus<KMAX>(ns, k);
            metaA[p] = (uint32_t)GPX_D_DECISION;
            it.omask |= 1u << i;
            it.nout++;
            e = 0;
            pcount--;
          }
        }
      };
      if (!__any(esc)) { /* the wave holds no escaped vote at all: nothing to step over, acceptors in the records */
        for (int32_t i = 0; i < nvote; i++) {
          const uint32_t p = (uint32_t)start + (uint32_t)pstride * (uint32_t)((it.order >> (4 * i)) & 15ull);
          vote(i, p, (int32_t)(metaA[p] >> 16), cpA[p]);
        }
      } else {
        for (int32_t i = 0; i < nvote; i++) {
          if ((skip >> i) & 1u) continue;
          const uint32_t p = (uint32_t)start + (uint32_t)pstride * (uint32_t)((it.order >> (4 * i)) & 15ull);
          const uint32_t meta = metaA[p];
          vote(i, p, (meta & V16_ESC) ? in.acceptor[idxA[p]] : (int32_t)(meta >> 16), cpA[p]);
        }
      }
      if (e != e0) S.p_ring[off] = e;
      if (ns_dirty) {
#pragma unroll
        for (int q = 0; q < KMAX; q++)
          if (q < k) S.node_slots[(int64_t)q * G + g] = ns[q];
      }
      if (pcount != P.pcount) S.c_pcount[g] = pcount;
    }
  } else {
    replay(it);
  }
}

/* the tiled kernels' LDS beside the staging: run starts (| TL_WIDE) and the exclusive prefix of the run lengths of the
 * bucket in hand - declared by the KERNEL (the pipelined one shares them with the body it calls for unusual buckets) */
struct TileLds {
  int32_t* pre; /* [GPX_TL_MAXWG + 1] */
  uint16_t* st; /* [GPX_TL_MAXWG] */
};
#define GPX_TILE_LDS_DECL(TILES_)                                   \
  __shared__ int32_t s_pre_[(TILES_) ? GPX_TL_MAXWG + 1 : 1];       \
  __shared__ uint16_t s_st_[(TILES_) ? GPX_TL_MAXWG : 1];           \
  const TileLds TL{s_pre_, s_st_}
/* b_in < 0: the workgroup's own bucket (blockIdx); else the bucket to do (k_bucket_ar16_tiles_pipe) */
template <int OP, int KMAX, bool TILES = false>
__device__ __forceinline__ void bucket16_body(const DevState& S, const DevScratch& X, const Stage16& O, const VoteCols& in,
                                              const AcceptOut& R, uint8_t* __restrict__ status, const TileLds& TL,
                                              const TileArea& A = TileArea{}, int32_t b_in = -1,
                                              const PlaceCols& IP = PlaceCols{}) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  constexpr bool AC = OP != B16_AR;
  /* ordered batch: k_ac_direct did it (a few sorted runs of votes: k_ar_runs); nothing was partitioned */
  if ((AC || X.gate) && *X.unsorted != X.epoch) return;
  int32_t b = b_in >= 0 ? b_in : (int32_t)blockIdx.x;
  if (TILES && A.xcd_rows && b_in < 0) { /* consecutive buckets on one XCD (block i runs on XCD i % 8): they share lines of A.off / A.recs */
    b = tile_of_block(X.nbk);
    if (b >= X.nbk) return;
  }
  const int32_t gb = X.gb; /* == blockDim.x: one lane per group */
  const int32_t l = (int32_t)threadIdx.x;
  int32_t boff, nb;
  /* TILES (gpx_tiles.hip.h): the run of every tile in this bucket - where it starts in the tile (| TL_WIDE) and the
   * exclusive prefix of the run lengths, what turns a record's index in the bucket into (tile, position) */
  int32_t* const s_pre = TL.pre;
  uint16_t* const s_st = TL.st;
  if (TILES) TL_STAMP(4096 + b, 0);
  if (TILES) { /* no k_hist, no scanned offsets: rows b and b + 1 of A.off say everything */
    /* A.off is [bucket / 4][tile][4]: this bucket's start and the next one's - the same 8-byte entry three times in four */
    const unsigned long long* r0 = (const unsigned long long*)A.off + (uint32_t)(b >> 2) * (uint32_t)A.nwg_pad;
    const unsigned long long* r1 = (const unsigned long long*)A.off + (uint32_t)((b + 1) >> 2) * (uint32_t)A.nwg_pad;
    const int32_t per = (A.nwg + gb - 1) >> X.shift; /* (gb == 1 << X.shift) <= 4: at most 1024 tiles, at least 256 lanes */
    int32_t cw[4], ssum = 0, csum = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int32_t w = l * per + q;
      cw[q] = 0;
      if (q < per && w < A.nwg) {
        const unsigned long long e0 = r0[w];
        const unsigned long long e1 = ((b & 3) == 3) ? r1[w] : e0;
        const uint16_t o0 = (uint16_t)(e0 >> (16 * (b & 3))), o1 = (uint16_t)(e1 >> (16 * ((b + 1) & 3)));
        const int32_t st = (int32_t)(o0 & 0x7fffu);
        cw[q] = (int32_t)(o1 & 0x7fffu) - st;
        s_st[w] = o0;
        ssum += st;
        csum += cw[q];
      }
    }
    /* (the starts add up to the records of the buckets before this one) */
    int32_t ex = block_exscan_and_sum_rt(csum, ssum, &nb, &boff);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int32_t w = l * per + q;
      if (q < per && w < A.nwg) {
        s_pre[w] = ex;
        ex += cw[q];
      }
    }
    if (l == 0) {
      s_pre[A.nwg] = nb;
      X.bucket_off[b] = boff; /* where k_emit_dec16 finds this bucket's staged outputs */
      if (nb == 0) X.bucket_nout[b] = 0;
    }
  } else {
    boff = X.bucket_off[b];
    nb = X.bucket_off[b + 1] - boff;
    if (threadIdx.x == 0) {
      X.bucket_tot[b] = 0; /* ready for the next batch's k_hist */
      if (nb == 0) X.bucket_nout[b] = 0;
    }
  }
  if (nb == 0) return;
  const int32_t g = X.g_base + (b << X.shift) + l;
  /* coordinator state of a dense bucket: issued now, consumed after the regrouping.  (Requested one round trip earlier,
   * beside the rows of A.off, when the host knows the call is dense: measured, no change - profiles/
   * r06_bucket_kernel_attempts.txt.) */
#ifdef GPX_B16_NOPRELOAD /* tuning build: state fetched after the regrouping (fewer live registers) */
  const bool pre = false;
#else
  const bool pre = !AC && 2 * nb >= gb;
#endif
  CoordPre<KMAX> P;
  P.have_pe = false;
  P.my_bnum = P.my_bcoord = 0;
  if (pre && g < X.g_end) coord_preload<KMAX>(S, g, P);
  const int32_t L = X.lds_recs;
  int32_t* lcnt = lds;
  int32_t* lcur = lds + gb;
  int32_t* idxA = lds + 2 * gb;
  int32_t* slotA = idxA + L;
  int32_t* cpA = slotA + L;
  uint32_t* metaA = (uint32_t*)(cpA + L);
  const bool in_lds = nb <= L;
  Vote16* recG = (Vote16*)X.rec + boff;
  unsigned long long* keysG = X.perm + boff;
  lcnt[l] = 0;
  /* the batch's reference vote (gpx_tiles.hip.h): ballot of the 8-byte records, base of their slot bytes */
  const int32_t slot0 = TILES ? A.ref[0] : 0;
  const int32_t ref_bn = TILES ? A.ref[1] : in.bnum[0], ref_bc = TILES ? A.ref[2] : in.bcoord[0];
  /* in-place outputs (PlaceCols): this bucket's predicted span.  (The divisor as a kernel argument with its reciprocal:
   * fetched from device memory and divided by at run time it cost the kernel 1.8 us - profiles/r06_in_place_outputs.txt.) */
  int32_t ip_pb = 0, ip_span = -1;
  if (TILES && !AC && IP.div) {
    ip_pb = IP.quot(boff);
    ip_span = IP.quot(boff + nb) - ip_pb;
  }
  int32_t srch = 0; /* steps of the search over s_pre */
  if (TILES)
    while ((1 << srch) < A.nwg) srch++;
  /* record j of the bucket (TILES): the largest tile w with s_pre[w] <= j holds it at s_st[w] + j - s_pre[w] */
  auto tile_find = [&](int32_t j) -> int32_t {
    int32_t lo = 0, hi = A.nwg;
    for (int32_t t = 0; t < srch; t++) { /* s_pre[lo] <= j < s_pre[hi]; neighbours: mid == lo, nothing moves */
      const int32_t mid = (lo + hi) >> 1;
      const bool up = s_pre[mid] <= j;
      lo = up ? mid : lo;
      hi = up ? hi : mid;
    }
    return lo;
  };
  /* the same from a tile that holds an EARLIER record (s_pre[nwg] = nb ends the walk; empty runs are stepped over) */
  auto tile_adv = [&](int32_t j, int32_t lo) -> int32_t {
    while (s_pre[lo + 1] <= j) lo++;
    return lo;
  };
  /* where record j sits in A.recs, given its tile (32-bit: the area holds max_batch votes + a tile; | TL_WIDE << 16 in
   * *wide).  Split from the load and from the expansion so that a lane's loads go out TOGETHER: the expansion of one record
   * waits for its data, and with the next record's address behind it the round trips came one after the other. */
  auto tile_pos = [&](int32_t j, int32_t lo, uint32_t* wide) -> uint32_t {
    const uint32_t st = s_st[lo];
    *wide = st & TL_WIDE;
    return __umul24((uint32_t)lo, (uint32_t)A.tile) + (st & 0x7fffu) + (uint32_t)(j - s_pre[lo]);
  };
  auto tile_exp = [&](const Vote8 raw, uint32_t p, uint32_t wide, int32_t lo) -> Vote16 {
    const I4 x = tile_expand(raw, wide ? A.ext + p : (const int2*)nullptr, lo, A.tile, slot0, in.slot, in.maxcp);
    Vote16 v;
    v.idx = x.x, v.slot = x.y, v.maxcp = x.z, v.meta = (uint32_t)x.w;
    return v;
  };
  auto tile_at = [&](int32_t j, int32_t lo) -> Vote16 {
    uint32_t wide;
    const uint32_t p = tile_pos(j, lo, &wide);
    return tile_exp(A.recs[p], p, wide, lo);
  };
  auto tile_rec = [&](int32_t j) -> Vote16 { return tile_at(j, tile_find(j)); };
  if (TILES) __syncthreads(); /* s_pre / s_st */
  if (TILES) TL_STAMP(4096 + b, 1); /* rows of A.off read and scanned */
#ifdef GPX_ABL_STOP /* ablation builds (wrong results): the kernel up to one of its phases */
  if (TILES && GPX_ABL_STOP == 1) return;
#endif
  if (TILES && !in_lds) {
    /* too many records for the LDS staging (a skewed stream): copy them into this bucket's region of X.rec and go on as
     * the partition path does (the count pass below reads that region) */
    for (int32_t j = l; j < nb; j += gb) recG[j] = tile_rec(j);
    __threadfence_block();
  }
  __syncthreads();
  /* The first four votes of a lane stay in registers between its two passes (count, placement) - or its ONE pass: */
  Vote16 r0, r1, r2, r3;
  r0.meta = r1.meta = r2.meta = r3.meta = 0;
  r0.idx = r1.idx = r2.idx = r3.idx = 0;
  r0.slot = r1.slot = r2.slot = r3.slot = 0;
  r0.maxcp = r1.maxcp = r2.maxcp = r3.maxcp = 0;
  const bool from_tiles = TILES && in_lds; /* (beyond the LDS staging the records were just copied to recG) */
  /* A lane's records.  From the tiles: CONSECUTIVE ones, j = l * pl + t - they mostly sit in one run, so the run of the
   * first is searched for and the others walk on from it (a search per record was 8 LDS round trips and ~60 instructions
   * each, in a kernel bound by the instructions it issues: profiles/r06_bucket_kernel_attempts.txt).  From X.rec (the
   * partition path; a bucket beyond the LDS staging): j = l + t * gb, coalesced 16-byte records.  The first four stay in
   * registers, the fifth and later ones are fetched again by a second pass.  The order in which a workgroup takes its
   * records is free: ranks come from LDS atomics, the replay orders a group's votes by arrival index. */
  const int32_t pl = from_tiles ? (nb + gb - 1) >> X.shift : 0; /* records per lane */
  const int32_t jb = from_tiles ? l * pl : l, jstep = from_tiles ? 1 : gb;
  const int32_t jend = from_tiles ? min(jb + pl, nb) : nb;
  const bool v0 = jb < jend, v1 = jb + jstep < jend, v2 = jb + 2 * jstep < jend, v3 = jb + 3 * jstep < jend;
  int32_t lo3 = 0; /* tile of the lane's last record in a register */
  if (from_tiles) {
    int32_t l0 = 0, l1 = 0, l2 = 0;
    uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    Vote8 q0{}, q1{}, q2{}, q3{};
    /* the four places, the four loads ... */
    if (v0) lo3 = l0 = tile_find(jb), p0 = tile_pos(jb, l0, &w0), q0 = A.recs[p0];
    if (v1) lo3 = l1 = tile_adv(jb + 1, l0), p1 = tile_pos(jb + 1, l1, &w1), q1 = A.recs[p1];
    if (v2) lo3 = l2 = tile_adv(jb + 2, l1), p2 = tile_pos(jb + 2, l2, &w2), q2 = A.recs[p2];
    if (v3) lo3 = tile_adv(jb + 3, l2), p3 = tile_pos(jb + 3, lo3, &w3), q3 = A.recs[p3];
    /* ... then what they brought */
    if (v0) r0 = tile_exp(q0, p0, w0, l0);
    if (v1) r1 = tile_exp(q1, p1, w1, l1);
    if (v2) r2 = tile_exp(q2, p2, w2, l2);
    if (v3) r3 = tile_exp(q3, p3, w3, lo3);
  } else {
    if (v0) r0 = recG[jb];
    if (v1) r1 = recG[jb + jstep];
    if (v2) r2 = recG[jb + 2 * jstep];
    if (v3) r3 = recG[jb + 3 * jstep];
  }
  const int32_t tail0 = jb + 4 * jstep; /* first record of the lane that is not in a register */
  /* calls f on the lane's records beyond the first four */
  auto for_tail = [&](auto&& f) {
    if (from_tiles) {
      int32_t lo = lo3;
      for (int32_t j = tail0; j < jend; j++) {
        lo = tile_adv(j, lo);
        f(tile_at(j, lo));
      }
    } else {
      for (int32_t j = tail0; j < jend; j += jstep) f(recG[j]);
    }
  };
  /* SLOTTED PLACEMENT (round 6; accept replies behind the tiled front end): a group's vote of rank t goes straight to
   * row t of the staging arrays - [t * gb + group], KSLOT rows - at the rank its one LDS atomic returns: no count
   * pass, no scan, no second atomic.  (The kernel is bound by the instructions it issues, 929 vector instructions
   * per wave for 192 votes - profiles/r06_pmc_ar_loop.txt -, not by bytes.)  A group with more votes than rows - several
   * rounds in one call, a hot group - sends the whole bucket through the two passes below instead. */
  constexpr int KSLOT = KMAX <= 4 ? 4 : 8;
  bool slotted = false;
  if (TILES && !AC && in_lds && KSLOT * gb <= L) { /* (uniform over the workgroup) */
    bool over = false;
    auto put_row = [&](const Vote16& v) {
      const int32_t lg = (int32_t)(v.meta & V16_LG_MASK);
      const int32_t t = atomicAdd(&lcnt[lg], 1);
      if (t < KSLOT) {
        const int32_t p = t * gb + lg;
        idxA[p] = v.idx;
        slotA[p] = v.slot;
        cpA[p] = v.maxcp;
        metaA[p] = v.meta;
      } else {
        over = true;
      }
    };
    if (v0) put_row(r0);
    if (v1) put_row(r1);
    if (v2) put_row(r2);
    if (v3) put_row(r3);
    for_tail([&](const Vote16& v) { put_row(v); });
    slotted = !__syncthreads_or(over);
    if (!slotted) { /* start over, the general way */
      lcnt[l] = 0;
      __syncthreads();
    }
  }
  int32_t c, start, any_long = 0;
  if (slotted) {
    c = lcnt[l];
    start = l;
    if (TILES) TL_STAMP(4096 + b, 3);
#ifdef GPX_ABL_STOP
    if (TILES && GPX_ABL_STOP == 2) return;
#endif
    if (pre && g < X.g_end) coord_preload_ring<KMAX>(S, g, P);
  } else {
  /* A: votes per group */
  {
    if (v0) atomicAdd(&lcnt[r0.meta & V16_LG_MASK], 1);
    if (v1) atomicAdd(&lcnt[r1.meta & V16_LG_MASK], 1);
    if (v2) atomicAdd(&lcnt[r2.meta & V16_LG_MASK], 1);
    if (v3) atomicAdd(&lcnt[r3.meta & V16_LG_MASK], 1);
    for_tail([&](const Vote16& v) { atomicAdd(&lcnt[v.meta & V16_LG_MASK], 1); });
  }
  __syncthreads();
  if (TILES) TL_STAMP(4096 + b, 2); /* records fetched and counted */
  if (pre && g < X.g_end) coord_preload_ring<KMAX>(S, g, P);
  /* B: exclusive scan of the counts */
  c = lcnt[l];
  int32_t tot_;
  start = block_exscan_rt(c, &tot_);
  lcur[l] = start;
  any_long = __syncthreads_or(c > V16_NIB_MAX || !in_lds);
  /* C: placement.  LDS: the vote itself, group-major; global mode: a key per vote. */
  if (in_lds) {
    auto place = [&](const Vote16& v) {
      const int32_t p = atomicAdd(&lcur[v.meta & V16_LG_MASK], 1);
      idxA[p] = v.idx;
      slotA[p] = v.slot;
      cpA[p] = v.maxcp;
      metaA[p] = v.meta;
    };
    if (v0) place(r0);
    if (v1) place(r1);
    if (v2) place(r2);
    if (v3) place(r3);
    /* (a lane's records beyond its first four: read a second time, from L2) */
    for_tail([&](const Vote16& v) { place(v); });
  } else {
    for (int32_t j = l; j < nb; j += gb) {
      const Vote16 v = recG[j];
      const int32_t p = atomicAdd(&lcur[v.meta & V16_LG_MASK], 1);
      keysG[p] = ((unsigned long long)(uint32_t)v.idx << 32) | (uint32_t)j;
    }
  }
  __syncthreads();
  if (TILES) TL_STAMP(4096 + b, 3); /* placed group-major */
  }
  const int32_t pstride = slotted ? gb : 1; /* a group's vote of rank t: start + t * pstride */
  /* D: segments that do not fit the nibble word (or a bucket in global mode): sorted keys in global
   * scratch.  A segment of up to V16_LANE_SORT votes staged in LDS is ordered by ITS OWN lane (rank by
   * counting over the arrival indices in LDS: a call that brings many rounds of votes at once has 17+
   * votes in most groups, and taking those groups one at a time through the cooperative sort below was
   * the 13x cliff at 2^24 votes per call of round 2's batch sweep); only a really hot group - serial by
   * contract, like the Java monitor - is sorted by the whole workgroup. */
  if (any_long) {
    if (in_lds && c > V16_NIB_MAX && c <= V16_LANE_SORT) {
      for (int32_t t = 0; t < c; t++) {
        const uint32_t it_ = (uint32_t)idxA[start + t];
        int32_t r = 0;
        for (int32_t u = 0; u < c; u++) r += (uint32_t)idxA[start + u] < it_;
        keysG[start + r] = ((unsigned long long)it_ << 32) | (uint32_t)(start + t);
      }
    } else if (in_lds && c > V16_LANE_SORT) {
      for (int32_t t = 0; t < c; t++)
        keysG[start + t] = ((unsigned long long)(uint32_t)idxA[start + t] << 32) | (uint32_t)(start + t);
    } else if (!in_lds && c > 1 && c <= V16_NIB_MAX) { /* short segment in global mode: this lane */
      unsigned long long* a = keysG + start;
      for (int32_t i = 1; i < c; i++) {
        const unsigned long long x = a[i];
        int32_t p = i - 1;
        while (p >= 0 && a[p] > x) {
          a[p + 1] = a[p];
          p--;
        }
        a[p + 1] = x;
      }
    }
    const int32_t coop = __syncthreads_or(in_lds ? c > V16_LANE_SORT : c > V16_NIB_MAX);
    if (coop) {
      const int32_t lim = in_lds ? V16_LANE_SORT : V16_NIB_MAX;
      for (int32_t q = 0; q < gb; q++) {
        const int32_t cq = lcnt[q]; /* uniform */
        if (cq > lim) sort_long_segment(keysG + (lcur[q] - cq), (uint32_t)cq);
      }
      __syncthreads();
    }
  }
  /* E: replay, one lane per group */
  int32_t nout = 0;
  uint32_t omask = 0;
  const bool live = c != 0 && g < X.g_end;
  if (!AC && live && !pre) coord_preload<KMAX>(S, g, P);
  auto replay = [&](auto& it) {
    if (OP == B16_AR)
      apply_ar_group<KMAX>(S, X, g, it, status, P);
    else if (OP == B16_ACCEPT)
      apply_accept_group(S, X, g, it, R.r_bnum, R.r_bcoord, R.r_maxcp, R.r_flags, status, R.r_packed);
    else
      apply_commit_group(S, X, g, it, status);
  };
  /* a bucket's outputs, group-major, as columns: decisions (six columns) or execution runs (gidx,
   * first slot, count) */
  auto put = [&](int64_t o, int32_t a, int32_t z, int32_t kd) {
#ifdef GPX_STAGE_NT /* tuning build: the staging is not read again when the outputs are in place */
    if (TILES && !AC) {
      __builtin_nontemporal_store(g, &O.gidx()[o]);
      __builtin_nontemporal_store(a, &O.slot()[o]);
      __builtin_nontemporal_store(z, &O.median()[o]);
      __builtin_nontemporal_store(P.my_bnum, &O.bnum()[o]);
      __builtin_nontemporal_store(P.my_bcoord, &O.bcoord()[o]);
      __builtin_nontemporal_store((uint8_t)kd, &O.kind()[o]);
      return;
    }
#endif
    O.gidx()[o] = g;
    O.slot()[o] = a;
    O.median()[o] = z;
    if (!AC) {
      /* (in-place form: the ballot columns are not staged - a compacting k_emit_dec16 reads the group's own, which no
       * accept-reply call changes) */
      if (!(TILES && IP.div)) {
        O.bnum()[o] = P.my_bnum;
        O.bcoord()[o] = P.my_bcoord;
      }
      O.kind()[o] = (uint8_t)kd;
    }
  };
  /* the tiled front end's in-place outputs (PlaceCols): this bucket's predicted first output, or -1 - its count is not
   * the predicted one (the call's outputs get compacted from the staging: k_emit_dec16) or the form is off */
  auto in_place = [&](int32_t tout) -> int32_t {
    if (!TILES || AC || !IP.div) return -1;
    if (tout == ip_span) return ip_pb;
    if (l == 0) A.ref[3] = (int32_t)X.epoch;
    return -1;
  };
  auto put_in_place = [&](int32_t o, int32_t a, int32_t z, int32_t kd) {
    IP.gidx[o] = g;
    IP.slot[o] = a;
    IP.median[o] = z;
    IP.bnum[o] = P.my_bnum;
    IP.bcoord[o] = P.my_bcoord;
    IP.kind[o] = (uint8_t)kd;
  };
  if (in_lds) {
    VoteIter<true, AC> it;
    it.idxA = idxA;
    it.slotA = slotA;
    it.cpA = cpA;
    it.metaA = metaA;
    it.recG = recG;
    it.keys = keysG + start;
    it.in = in;
    it.b0n = ref_bn;
    it.b0c = ref_bc;
    it.start = start;
    it.stride = pstride;
    it.c = c;
    it.done = 0;
    it.nout = 0;
    it.nib = c <= V16_NIB_MAX;
    it.order = 0;
    it.omask = 0;
    it.cur = 0;
    if (live) ar16_replay_group<OP, KMAX>(S, X, g, it, P, in, R, status, idxA, slotA, cpA, metaA, start, pstride, c);
    nout = it.nout;
    omask = it.omask;
    if (TILES) TL_STAMP(4096 + b, 4); /* thread 0 replayed */
#ifdef GPX_ABL_STOP
    if (TILES && GPX_ABL_STOP == 3) return;
#endif
    /* F: the bucket's outputs, group-major, as columns */
    int32_t tout;
    const int32_t ex = block_exscan_rt(nout, &tout);
    const int32_t pb = in_place(tout);
    for (int32_t q = 0; q < nout; q++) {
      int32_t sl, md, kd;
      it.output(q, &sl, &md, &kd, &omask);
      put((int64_t)boff + ex + q, sl, md, kd);
      if (pb >= 0) put_in_place(pb + ex + q, sl, md, kd);
    }
    if (l == 0) X.bucket_nout[b] = tout;
    if (TILES) TL_STAMP(4096 + b, 5); /* outputs staged */
  } else {
    VoteIter<false, AC> it;
    it.idxA = idxA;
    it.slotA = slotA;
    it.cpA = cpA;
    it.metaA = metaA;
    it.recG = recG;
    it.keys = keysG + start;
    it.in = in;
    it.b0n = ref_bn;
    it.b0c = ref_bc;
    it.start = start;
    it.stride = 1;
    it.c = c;
    it.done = 0;
    it.nout = 0;
    it.nib = false;
    it.order = 0;
    it.omask = 0;
    it.cur = 0;
    if (live) replay(it);
    nout = it.nout;
    int32_t tout;
    const int32_t ex = block_exscan_rt(nout, &tout);
    const int32_t pb = in_place(tout);
    for (int32_t q = 0; q < nout; q++) {
      int32_t sl, md, kd;
      it.output(q, &sl, &md, &kd, &omask);
      put((int64_t)boff + ex + q, sl, md, kd);
      if (pb >= 0) put_in_place(pb + ex + q, sl, md, kd);
    }
    if (l == 0) X.bucket_nout[b] = tout;
  }
}

template <int OP, int KMAX>
__global__ __launch_bounds__(1024) GPX_AR16_ATTR void k_bucket16(DevState S, DevScratch X, Stage16 O, VoteCols in,
                                                   AcceptOut R, uint8_t* __restrict__ status) {
  GPX_TILE_LDS_DECL(false);
  bucket16_body<OP, KMAX>(S, X, O, in, R, status, TL);
}
/* accept replies behind the tiled front end (gpx_tiles.hip.h).
 *
 * The kernel's ~40 argument pointers do not fit the 102 SGPRs: taken by value, the compiler loads them all at entry and
 * parks what does not fit in the lanes of a VGPR - 562 v_readlane + 122 v_writelane of the k4 kernel's 3,068 vector
 * instructions (static; scripts/ubench/isa_mix.py), two v_readlane in front of every use of a parked pointer.  The blocks
 * that are only needed here and there - the caller's vote columns (the escape path), the scratch descriptor, the in-place
 * output columns (the last phase) - are therefore read from the kernarg segment WHERE THEY ARE USED (scalar loads from the
 * constant cache; the by-value parameters stay in the signature for the launch and are otherwise dead): 76 v_readlane,
 * 2,429 vector instructions, k4 40.9 -> 38.7 us, k5 76.5 -> 75.5 (profiles/r06_lazy_kernargs.txt).  The state block and the
 * tile area taken the same way only add scalar loads (GPX_LAZY bits 16 / 32: measured, no gain). */
struct TilesKernArgs { /* the argument block of the k_bucket_ar16_tiles kernels, as the kernarg segment lays it out */
  DevState S;
  DevScratch X;
  Stage16 O;
  VoteCols in;
  uint8_t* status;
  TileArea A;
  PlaceCols IP;
};
/* every argument is 8-byte aligned and a multiple of 8 bytes long: the segment packs them exactly as this struct does
 * (the mix stream's escapes and the in-place outputs of every parity test go through these offsets) */
static_assert(alignof(DevState) == 8 && alignof(DevScratch) == 8 && alignof(Stage16) == 8 && alignof(VoteCols) == 8 &&
                  alignof(TileArea) == 8 && alignof(PlaceCols) == 8,
              "kernarg layout");
static_assert(sizeof(DevState) % 8 == 0 && sizeof(DevScratch) % 8 == 0 && sizeof(Stage16) % 8 == 0 && sizeof(VoteCols) % 8 == 0 &&
                  sizeof(TileArea) % 8 == 0 && sizeof(PlaceCols) % 8 == 0,
              "kernarg layout");
#ifndef GPX_LAZY
#define GPX_LAZY 7 /* 1: the vote columns, 2: the scratch descriptor, 4: the in-place columns (8 status, 16 state, 32 tile area) */
#endif
template <bool LAZY, class T>
__device__ __forceinline__ const T& lazy_pick(const T& from_kernarg, const T& by_value) {
  if constexpr (LAZY) return from_kernarg;
  else return by_value;
}
#define GPX_TILES_KERNEL_BODY(KM)                                                                                          \
  GPX_TILE_LDS_DECL(true);                                                                                                 \
  const TilesKernArgs* ka = (const TilesKernArgs*)__builtin_amdgcn_kernarg_segment_ptr();                                  \
  bucket16_body<B16_AR, KM, true>(lazy_pick<(GPX_LAZY & 16) != 0>(ka->S, S), lazy_pick<(GPX_LAZY & 2) != 0>(ka->X, X), O,  \
                                  lazy_pick<(GPX_LAZY & 1) != 0>(ka->in, in), AcceptOut{},                                 \
                                  (GPX_LAZY & 8) ? ka->status : status, TL, lazy_pick<(GPX_LAZY & 32) != 0>(ka->A, A), -1, \
                                  lazy_pick<(GPX_LAZY & 4) != 0>(ka->IP, IP))
template <int KMAX>
__global__ __launch_bounds__(1024) GPX_AR16_ATTR void k_bucket_ar16_tiles(DevState S, DevScratch X, Stage16 O, VoteCols in,
                                                                         uint8_t* __restrict__ status, TileArea A, PlaceCols IP) {
  GPX_TILES_KERNEL_BODY(KMAX);
}
/* K <= 4 and five replicas held to 6 waves like k_bucket_ar16_k5 */
#ifndef GPX_K4_WAVES
#define GPX_K4_WAVES 6
#endif
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(GPX_K4_WAVES, 8))) void k_bucket_ar16_tiles_k4(
    DevState S, DevScratch X, Stage16 O, VoteCols in, uint8_t* __restrict__ status, TileArea A, PlaceCols IP) {
  GPX_TILES_KERNEL_BODY(4);
}
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_bucket_ar16_tiles_k5(
    DevState S, DevScratch X, Stage16 O, VoteCols in, uint8_t* __restrict__ status, TileArea A, PlaceCols IP) {
  GPX_TILES_KERNEL_BODY(5);
}
/* Five replicas (BASELINE config #4): the KMAX = 5 body needs 82 VGPRs left to itself - two over the step
 * to 5 waves per SIMD = two workgroups per CU instead of three; held to 6 waves it gives up two registers
 * to scratch instead. */
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_bucket_ar16_k5(
    DevState S, DevScratch X, Stage16 O, VoteCols in, AcceptOut R, uint8_t* __restrict__ status) {
  GPX_TILE_LDS_DECL(false);
  bucket16_body<B16_AR, 5>(S, X, O, in, R, status, TL);
}

/* staged columns -> the caller's columns, buckets in order */
__global__ __launch_bounds__(GPX_BLOCK) void k_emit_dec16(
    DevScratch X, Stage16 O, int32_t* __restrict__ d_gidx, int32_t* __restrict__ d_slot,
    int32_t* __restrict__ d_bnum, int32_t* __restrict__ d_bcoord, int32_t* __restrict__ d_median,
    uint8_t* __restrict__ d_kind, int32_t* total_out, unsigned long long* acc, const int32_t* base_in,
    int32_t* chain_out, int32_t* in_place = nullptr, int32_t nvotes = 0, const int32_t* __restrict__ c_bnum = nullptr,
    const int32_t* __restrict__ c_bcoord = nullptr) {
  /* base_in: outputs of the passes before this one (accept-reply calls over more than 4 M groups run
   * one pass per group range, ranges ascending: the concatenation is still grouped by gidx ascending) */
  if (X.gate && *X.unsorted != X.epoch) return; /* k_emit_dec_runs wrote the outputs (gpx_runs.hip.h) */
  /* in_place = A.ref of the tiled front end whose per-bucket kernel wrote the caller's columns itself (PlaceCols): word 3
   * holds the call's epoch iff some bucket's count was not the predicted one */
  const bool placed = in_place && in_place[3] != (int32_t)X.epoch;
  if (placed && blockIdx.x != gridDim.x - 1) return; /* (the last workgroup still publishes the total) */
  const int32_t base0 = base_in ? *base_in : 0;
  const int32_t b = blockIdx.x;
  int32_t before = 0;
  for (int32_t t = threadIdx.x; t < b; t += GPX_BLOCK) before += X.bucket_nout[t];
  int32_t pre;
  block_exscan(before, &pre);
  const int32_t out0 = base0 + pre;
  if (b == (int32_t)gridDim.x - 1 && threadIdx.x == 0) {
    const int32_t tot = pre + X.bucket_nout[b];
    if (total_out) *total_out = base0 + tot;
    if (chain_out) *chain_out = base0 + tot; /* never the word base_in points at: later workgroups still read that */
    if (acc) atomicAdd(acc, (unsigned long long)tot);
    /* a call that had to be compacted teaches the next one its votes per output */
    if (in_place && !placed && tot > 0 && nvotes % tot == 0 && nvotes / tot <= 64 && X.xabort)
      __hip_atomic_store(X.xabort + GPX_IP_LEARN_WORD, (uint32_t)(nvotes / tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (in_place) in_place[placed ? 5 : 6]++; /* gpx_engine_path_counters */
  }
  if (placed) return;
  const int32_t nd = X.bucket_nout[blockIdx.x];
  const int64_t src = X.bucket_off[blockIdx.x];
  if (threadIdx.x == 0) X.bucket_tot[blockIdx.x] = 0; /* (the slotted front end leaves its totals there: ready for a k_hist) */
  for (int32_t t = threadIdx.x; t < nd; t += GPX_BLOCK) {
    const int32_t g = O.gidx()[src + t];
    d_gidx[out0 + t] = g;
    d_slot[out0 + t] = O.slot()[src + t];
    /* c_bnum: the in-place form staged no ballots - the coordinator's own (PlaceCols) */
    d_bnum[out0 + t] = c_bnum ? c_bnum[g] : O.bnum()[src + t];
    d_bcoord[out0 + t] = c_bnum ? c_bcoord[g] : O.bcoord()[src + t];
    d_median[out0 + t] = O.median()[src + t];
    d_kind[out0 + t] = O.kind()[src + t];
  }
}

/* staged execution runs -> the caller's columns, buckets in order (partition path of ACCEPT / COMMIT) */
__global__ __launch_bounds__(GPX_BLOCK) void k_emit_runs16(DevScratch X, Stage16 O, int32_t* __restrict__ x_gidx,
                                                          int32_t* __restrict__ x_first,
                                                          int32_t* __restrict__ x_count, int32_t* total_out) {
  if (*X.unsorted != X.epoch) return; /* ordered batch: k_emit_runs_direct wrote the outputs */
  const int32_t out0 = emit_base(X, total_out, nullptr);
  const int32_t nd = X.bucket_nout[blockIdx.x];
  const int64_t src = X.bucket_off[blockIdx.x];
  for (int32_t t = threadIdx.x; t < nd; t += GPX_BLOCK) {
    x_gidx[out0 + t] = O.gidx()[src + t];
    x_first[out0 + t] = O.slot()[src + t];
    x_count[out0 + t] = O.median()[src + t];
  }
}

/* packed ACCEPT_REPLY rows -> the caller's four columns (streaming; the partition path of an
 * unordered ACCEPT batch wrote one 16-byte row per record at its arrival index) */
__global__ __launch_bounds__(GPX_BLOCK) void k_unpack_replies(DevScratch X, int32_t n, const I4* __restrict__ rows,
                                                             int32_t* __restrict__ r_bnum,
                                                             int32_t* __restrict__ r_bcoord,
                                                             int32_t* __restrict__ r_maxcp,
                                                             uint8_t* __restrict__ r_flags) {
  if (*X.unsorted != X.epoch) return; /* ordered batch: k_ac_direct wrote the columns itself */
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const I4 r = rows[i];
  r_bnum[i] = r.x;
  r_bcoord[i] = r.y;
  r_maxcp[i] = r.z;
  r_flags[i] = (uint8_t)r.w;
}
