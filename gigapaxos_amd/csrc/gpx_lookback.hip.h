/*
 * gpx_lookback.hip.h — decoupled look-back over per-workgroup words {epoch | state | count} (a single-pass prefix sum
 * across the workgroups of one launch).  Written for the one-launch wire decode (gpx_wire.hip.h, round 3); since round 5
 * the per-bucket kernel of the slotted accept-reply call places its outputs with it too (gpx_ar16.hip.h).
 */
#pragma once
#include "gpx_kernels.hip.h"

#define WL_AGG 1ull
#define WL_PRE 2ull
#define WL_VAL_MASK ((1ull << 38) - 1)
__device__ __forceinline__ unsigned long long wl_word(uint32_t epoch, unsigned long long st, uint32_t v) {
  return ((unsigned long long)epoch << 40) | (st << 38) | (unsigned long long)v;
}
/* exclusive prefix of class words before `tile`; called by one whole wave.  Polling is what this
 * costs: ~6,000 waves spinning on 64 words each flood the L2 request path (measured: four words per
 * lane made the kernel 35 % slower), so a wave first waits on ONE word - its nearest predecessor's,
 * published last of all it needs in the usual case - and only then reads 64 at a time. */
#ifndef GPX_WL_WIDE
#define GPX_WL_WIDE 1 /* words per lane and round trip (4 = 256 tiles per step: measured slower) */
#endif
__device__ __forceinline__ unsigned long long wl_lookback64(const unsigned long long* __restrict__ st, int32_t tile,
                                                 uint32_t epoch) {
  const int32_t lane = (int32_t)(threadIdx.x & 63);
  for (;;) { /* the nearest predecessor has parsed */
    const unsigned long long v = __hip_atomic_load(&st[tile - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(v >> 40) == epoch) break;
    __builtin_amdgcn_s_sleep(8);
  }
  unsigned long long excl = 0;
  for (int32_t hi0 = tile - 1; hi0 >= 0; hi0 -= 64 * GPX_WL_WIDE) {
    /* GPX_WL_WIDE x 64 words requested together (the older ones are published long since: one round trip for
     * 256 tiles; the walk's rate - tiles per round trip - is what bounds the whole kernel) */
    unsigned long long vv[GPX_WL_WIDE];
#pragma unroll
    for (int k = 0; k < GPX_WL_WIDE; k++) {
      const int32_t j = hi0 - 64 * k - lane;
      vv[k] = j >= 0 ? __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
#pragma unroll
    for (int k = 0; k < GPX_WL_WIDE; k++) {
      const int32_t hi = hi0 - 64 * k;
      if (hi < 0) break;
      const int32_t j = hi - lane; /* lane 0 = the nearest predecessor of this sub-step */
      unsigned long long v = vv[k];
      bool need = j >= 0 && (uint32_t)(v >> 40) != epoch;
      unsigned long long pre_mask;
      for (;;) {
        /* the walk stops at the nearest tile with a PREFIX: only lanes nearer than it must be valid */
        pre_mask = __ballot(!need && j >= 0 && ((v >> 38) & 3ull) == WL_PRE);
        const unsigned long long wait_mask = __ballot(need);
        if (pre_mask) {
          const unsigned long long nearer = (pre_mask & (0ull - pre_mask)) - 1ull;
          if ((wait_mask & nearer) == 0) break;
        } else if (wait_mask == 0) {
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        if (need) {
          v = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(v >> 40) == epoch) need = false;
        }
      }
      const int32_t first_pre = pre_mask ? (__ffsll((long long)pre_mask) - 1) : 64;
      unsigned long long x = (j >= 0 && lane <= first_pre) ? (v & WL_VAL_MASK) : 0ull;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, d, 64);
        const uint32_t hi32 = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), d, 64);
        x += ((unsigned long long)hi32 << 32) | lo;
      }
      excl += x;
      if (pre_mask) return excl;
    }
  }
  return excl;
}

__device__ __forceinline__ uint32_t wl_lookback(const unsigned long long* __restrict__ st, int32_t tile,
                                                 uint32_t epoch) {
  return (uint32_t)wl_lookback64(st, tile, epoch);
}
