/*
 * gpx_route.hip.h — hash-sharding of a batch across the GPUs of a node on the device.
 *
 * Groups are independent (PaxosManager.java:3170-3171): GPU s holds the groups with
 * fmix32(gidx) % n_shards == s (murmur3 finaliser) and no collective carries protocol data.  A node
 * whose batcher sees one mixed stream bins it by device with these kernels instead of on the host:
 * a STABLE partition of the batch's columns by shard (record order inside a shard is the batch's
 * order: the per-group ordering contract survives), the group index rewritten to the shard-local
 * dense index through a resident table.
 *   k_route_count    per tile of 4096 records: records per shard
 *   k_route_offsets  one workgroup: tile x shard counts -> start of every tile's slice in every shard
 *   k_route_scatter  per tile: stable placement (wave ballots in record order), all columns
 */
#pragma once
#include "gpx_kernels.hip.h"

#define GPX_ROUTE_MAX_SHARDS 16
#define GPX_ROUTE_MAX_COLS 8
#define GPX_ROUTE_TILE 4096
#define GPX_ROUTE_NT 1024

__device__ __forceinline__ uint32_t fmix32_dev(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
/* out-of-range indices stay on shard 0 (dropped there with GPX_S_NOGROUP), like ShardMap.route */
__device__ __forceinline__ int32_t shard_of_dev(int32_t g, int32_t G, int32_t ns) {
  return (uint32_t)g < (uint32_t)G ? (int32_t)(fmix32_dev((uint32_t)g) % (uint32_t)ns) : 0;
}

struct RouteCols {
  const int32_t* in[GPX_ROUTE_MAX_COLS]; /* in[0] = global gidx */
  int32_t* out[GPX_ROUTE_MAX_COLS];
  int32_t ncols;
};

__global__ __launch_bounds__(GPX_ROUTE_NT) void k_route_count(int32_t n, const int32_t* __restrict__ gidx,
                                                             int32_t G, int32_t ns, int32_t* __restrict__ tile_cnt) {
  __shared__ int32_t cnt[GPX_ROUTE_MAX_SHARDS];
  if (threadIdx.x < GPX_ROUTE_MAX_SHARDS) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * GPX_ROUTE_TILE + (int64_t)threadIdx.x * 4;
  int32_t sh[4];
#pragma unroll
  for (int q = 0; q < 4; q++) sh[q] = i0 + q < n ? shard_of_dev(gidx[i0 + q], G, ns) : -1;
  for (int32_t s = 0; s < ns; s++) {
    const int32_t mine = (sh[0] == s) + (sh[1] == s) + (sh[2] == s) + (sh[3] == s);
    int32_t x = mine;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&cnt[s], x);
  }
  __syncthreads();
  if ((int32_t)threadIdx.x < ns) tile_cnt[(int64_t)blockIdx.x * ns + threadIdx.x] = cnt[threadIdx.x];
}

/* one workgroup: per shard a block scan over the tiles (thread t owns a run of consecutive tiles) */
__global__ __launch_bounds__(GPX_ROUTE_NT) void k_route_offsets(int32_t ntiles, int32_t ns,
                                                               int32_t* __restrict__ tile_cnt,
                                                               int32_t* __restrict__ shard_off) {
  __shared__ int32_t tot[GPX_ROUTE_MAX_SHARDS + 1];
  const int32_t per = (ntiles + GPX_ROUTE_NT - 1) / GPX_ROUTE_NT;
  const int32_t t0 = (int32_t)threadIdx.x * per;
  for (int32_t s = 0; s < ns; s++) {
    int32_t mine = 0;
    for (int32_t q = 0; q < per; q++)
      if (t0 + q < ntiles) mine += tile_cnt[(int64_t)(t0 + q) * ns + s];
    int32_t total;
    int32_t run = block_exscan_n<GPX_ROUTE_NT>(mine, &total);
    for (int32_t q = 0; q < per; q++)
      if (t0 + q < ntiles) {
        const int64_t o = (int64_t)(t0 + q) * ns + s;
        const int32_t c = tile_cnt[o];
        tile_cnt[o] = run; /* start of the tile's slice inside shard s */
        run += c;
      }
    if (threadIdx.x == 0) tot[s] = total;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t o = 0;
    for (int32_t q = 0; q < ns; q++) {
      shard_off[q] = o;
      o += tot[q];
    }
    shard_off[ns] = o;
  }
}

__global__ __launch_bounds__(GPX_ROUTE_NT) void k_route_scatter(int32_t n, int32_t G, int32_t ns, RouteCols C,
                                                               const int32_t* __restrict__ g2l,
                                                               const int32_t* __restrict__ tile_off,
                                                               const int32_t* __restrict__ shard_off) {
  __shared__ int32_t wcnt[GPX_ROUTE_NT / 64][GPX_ROUTE_MAX_SHARDS];
  const int32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * GPX_ROUTE_TILE + (int64_t)threadIdx.x * 4;
  int32_t g[4], sh[4], rank[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    g[q] = i0 + q < n ? C.in[0][i0 + q] : -1;
    sh[q] = i0 + q < n ? shard_of_dev(g[q], G, ns) : -1;
    rank[q] = 0;
  }
  /* rank of a record among the records of its shard in this tile, in record order: records of
   * earlier waves + of earlier lanes of this wave + earlier records of this lane */
  for (int32_t s = 0; s < ns; s++) {
    const int32_t mine = (sh[0] == s) + (sh[1] == s) + (sh[2] == s) + (sh[3] == s);
    int32_t inc = mine; /* inclusive scan over lanes */
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t y = __shfl_up(inc, d, 64);
      if (lane >= d) inc += y;
    }
    if (lane == 63) wcnt[wid][s] = inc;
    int32_t before = inc - mine;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (sh[q] == s) rank[q] = before++;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (sh[q] < 0) continue;
    int32_t base = shard_off[sh[q]] + tile_off[(int64_t)blockIdx.x * ns + sh[q]];
    for (int32_t w = 0; w < wid; w++) base += wcnt[w][sh[q]];
    const int64_t o = (int64_t)base + rank[q];
    /* the shard-local dense index; -1 stays -1 (and any out-of-range index becomes -1) */
    C.out[0][o] = (uint32_t)g[q] < (uint32_t)G ? (g2l ? g2l[g[q]] : g[q]) : -1;
    for (int32_t k = 1; k < C.ncols; k++) C.out[k][o] = C.in[k][i0 + q];
  }
}
