// Microbenchmark (round 3): a partition-free front end for SHUFFLED votes - "mailboxes".
// Every vote draws its rank in its group's mailbox with ONE returning atomicAdd on cnt[g] and stores its
// 16-byte record at box[g * CAP + rank]; a second kernel (one lane per group) would read the boxes.
// Replaces k_hist + k_scatter_ar16 (+ the LDS regrouping of k_bucket_ar16) IF 3 M returning atomics on 1 M
// distinct addresses are cheap.  Measured here:
//   M1  the atomics alone (gidx in, rank out to a dense column)
//   M2  the whole filing kernel: six columns in, atomic, 16-byte store at the drawn position
//   M3  M2 without the atomic (rank = a pure function of the vote: what the stores alone cost)
//   M4  the reading side: one lane per group, cnt + CAP records (64-byte stride), sum out
//   M5  M2 with NON-returning atomics + rank from the vote (what the return trip costs)
//   hipcc --offload-arch=gfx950 -O3 -o ubench_mailbox ubench_mailbox.hip && ./ubench_mailbox
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct __attribute__((aligned(16))) I4 { int32_t x, y, z, w; };
struct __attribute__((aligned(16))) V16 { int32_t idx, slot, cp; uint32_t meta; };
#define NT 256
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
// a shuffled round: a host-made random permutation of the n = G * K (group, member) pairs
__global__ void k_setup(int n, int K, const int* __restrict__ perm, int* gidx, int* c1, int* c2, int* c3, int* c4, int* c5) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int j = perm[i]; gidx[i] = j / K; c1[i] = 0; c2[i] = 100; c3[i] = 7; c4[i] = 100 + j % K; c5[i] = 6; }
}
static int* make_perm(int n) { std::vector<int> p(n); for (int i = 0; i < n; i++) p[i] = i;
  uint64_t x = 0x9E3779B97F4A7C15ull; for (int i = n - 1; i > 0; i--) { x ^= x >> 12; x ^= x << 25; x ^= x >> 27; const uint64_t r = x * 2685821657736338717ull;
    std::swap(p[i], p[(int)(r % (uint64_t)(i + 1))]); }
  int* d; if (hipMalloc(&d, (size_t)n * 4) != hipSuccess) return nullptr; hipMemcpy(d, p.data(), (size_t)n * 4, hipMemcpyHostToDevice); return d; }
__global__ __launch_bounds__(NT) void k_m1(int n, const int* __restrict__ gidx, int* cnt, int* rank_out) {
  const long i0 = ((long)blockIdx.x * NT + threadIdx.x) * 4;
  if (i0 + 3 < n) { const I4 g = *(const I4*)(gidx + i0); I4 r;
    r.x = atomicAdd(&cnt[g.x], 1); r.y = atomicAdd(&cnt[g.y], 1); r.z = atomicAdd(&cnt[g.z], 1); r.w = atomicAdd(&cnt[g.w], 1);
    *(I4*)(rank_out + i0) = r; }
}
template <int MODE, int CAP>
__global__ __launch_bounds__(NT) void k_file(int n, const int* __restrict__ gidx, const int* __restrict__ bn, const int* __restrict__ bc,
                                             const int* __restrict__ slot, const int* __restrict__ acc, const int* __restrict__ cp,
                                             int* cnt, V16* box, int* ovf) {
  const long i0 = ((long)blockIdx.x * NT + threadIdx.x) * 4;
  if (i0 + 3 >= n) return;
  const I4 g = *(const I4*)(gidx + i0), b = *(const I4*)(bn + i0), c = *(const I4*)(bc + i0), s = *(const I4*)(slot + i0),
           a = *(const I4*)(acc + i0), p = *(const I4*)(cp + i0);
  const int gg[4] = {g.x, g.y, g.z, g.w}, ss[4] = {s.x, s.y, s.z, s.w}, aa[4] = {a.x, a.y, a.z, a.w}, pp[4] = {p.x, p.y, p.z, p.w},
            bb[4] = {b.x, b.y, b.z, b.w}, cc[4] = {c.x, c.y, c.z, c.w};
  int r[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (MODE == 0) r[q] = atomicAdd(&cnt[gg[q]], 1);
    else if (MODE == 1) r[q] = aa[q] - 100;
    else { __hip_atomic_fetch_add(&cnt[gg[q]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r[q] = aa[q] - 100; }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const bool esc = bb[q] != 0 || cc[q] != 100;
    if (r[q] < CAP) { V16 v; v.idx = (int)(i0 + q); v.slot = ss[q]; v.cp = pp[q]; v.meta = (uint32_t)aa[q] << 16 | (esc ? 0x8000u : 0u);
      box[(long)gg[q] * CAP + r[q]] = v; }
    else *ovf = 1;
  }
}
template <int CAP>
__global__ __launch_bounds__(NT) void k_read(int G, int* cnt, const V16* __restrict__ box, int* out) {
  const int g = blockIdx.x * NT + threadIdx.x;
  if (g >= G) return;
  const int c = cnt[g]; int acc = 0;
#pragma unroll
  for (int q = 0; q < CAP; q++) if (q < c) { const V16 v = box[(long)g * CAP + q]; acc += v.idx ^ v.slot ^ v.cp ^ (int)v.meta; }
  cnt[g] = 0; out[g] = acc;
}
// launch floor against grid barrier: NK dependent near-empty launches vs ONE launch with NK - 1 barriers
__global__ __launch_bounds__(NT) void k_touch(int* x, int G) { const int g = blockIdx.x * NT + threadIdx.x; if (g < G) x[g] += 1; }
__global__ __launch_bounds__(NT) void k_touch_barrier(int* x, int G, int nk, unsigned* bar, unsigned epoch_base) {
  const int g = blockIdx.x * NT + threadIdx.x;
  for (int k = 0; k < nk; k++) {
    if (g < G) x[g] += 1;
    if (k + 1 < nk) { // relaxed agent-scope counter, one arrival per workgroup, all workgroups co-resident
      __syncthreads();
      if (threadIdx.x == 0) { const unsigned target = (epoch_base + (unsigned)k + 1) * gridDim.x;
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1); }
      __syncthreads();
    }
  }
}
int main() {
  const int G = 1000000;
  for (int K : {3, 5}) {
    const int n = G * K;
    int *gidx, *c1, *c2, *c3, *c4, *c5, *cnt, *rk, *ovf; V16* box;
    CK(hipMalloc(&gidx, (size_t)n * 4)); CK(hipMalloc(&c1, (size_t)n * 4)); CK(hipMalloc(&c2, (size_t)n * 4)); CK(hipMalloc(&c3, (size_t)n * 4));
    CK(hipMalloc(&c4, (size_t)n * 4)); CK(hipMalloc(&c5, (size_t)n * 4)); CK(hipMalloc(&cnt, (size_t)G * 4)); CK(hipMalloc(&rk, (size_t)n * 4));
    CK(hipMalloc(&ovf, 4)); CK(hipMalloc(&box, (size_t)G * 8 * 16));
    { int* perm = make_perm(n); hipLaunchKernelGGL(k_setup, dim3((n + 255) / 256), dim3(256), 0, 0, n, K, perm, gidx, c1, c2, c3, c4, c5); CK(hipDeviceSynchronize()); hipFree(perm); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nb = (n / 4 + NT - 1) / NT, gb = (G + NT - 1) / NT;
    auto timeit = [&](const char* name, auto fn) { float best = 1e9f, sum = 0;
      for (int it = 0; it < 12; it++) { hipMemsetAsync(cnt, 0, (size_t)G * 4, 0); hipMemsetAsync(ovf, 0, 4, 0); hipEventRecord(e0, 0); fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) { best = std::min(best, ms); sum += ms; } }
      int o; hipMemcpy(&o, ovf, 4, hipMemcpyDeviceToHost);
      printf("K=%d %-58s best %7.1f us  avg %7.1f us  ovf %d\n", K, name, best * 1000, sum / 10 * 1000, o); };
    timeit("M1 atomics alone (returning, 1 per vote)", [&] { hipLaunchKernelGGL(k_m1, dim3(nb), dim3(NT), 0, 0, n, gidx, cnt, rk); });
    if (K == 3) {
      timeit("M2 file: columns in + atomic + 16 B store (CAP 4)", [&] { hipLaunchKernelGGL((k_file<0, 4>), dim3(nb), dim3(NT), 0, 0, n, gidx, c1, c2, c3, c4, c5, cnt, box, ovf); });
      timeit("M3 file without the atomic (CAP 4)", [&] { hipLaunchKernelGGL((k_file<1, 4>), dim3(nb), dim3(NT), 0, 0, n, gidx, c1, c2, c3, c4, c5, cnt, box, ovf); });
      timeit("M5 file, non-returning atomic (CAP 4)", [&] { hipLaunchKernelGGL((k_file<2, 4>), dim3(nb), dim3(NT), 0, 0, n, gidx, c1, c2, c3, c4, c5, cnt, box, ovf); });
      timeit("M2+M4 file + read side (CAP 4)", [&] { hipLaunchKernelGGL((k_file<0, 4>), dim3(nb), dim3(NT), 0, 0, n, gidx, c1, c2, c3, c4, c5, cnt, box, ovf);
        hipLaunchKernelGGL((k_read<4>), dim3(gb), dim3(NT), 0, 0, G, cnt, box, rk); });
    }
    timeit("M2 file: columns in + atomic + 16 B store (CAP 8)", [&] { hipLaunchKernelGGL((k_file<0, 8>), dim3(nb), dim3(NT), 0, 0, n, gidx, c1, c2, c3, c4, c5, cnt, box, ovf); });
    timeit("M3 file without the atomic (CAP 8)", [&] { hipLaunchKernelGGL((k_file<1, 8>), dim3(nb), dim3(NT), 0, 0, n, gidx, c1, c2, c3, c4, c5, cnt, box, ovf); });
    timeit("M2+M4 file + read side (CAP 8)", [&] { hipLaunchKernelGGL((k_file<0, 8>), dim3(nb), dim3(NT), 0, 0, n, gidx, c1, c2, c3, c4, c5, cnt, box, ovf);
      hipLaunchKernelGGL((k_read<8>), dim3(gb), dim3(NT), 0, 0, G, cnt, box, rk); });
    // the same at a 125 k-group shard (config #4 split eight ways)
    hipFree(gidx); hipFree(c1); hipFree(c2); hipFree(c3); hipFree(c4); hipFree(c5); hipFree(cnt); hipFree(rk); hipFree(ovf); hipFree(box);
  }
  {
    const int G2 = 125000, K = 5, n = G2 * K;
    int *gidx, *c1, *c2, *c3, *c4, *c5, *cnt, *rk, *ovf; V16* box;
    CK(hipMalloc(&gidx, (size_t)n * 4)); CK(hipMalloc(&c1, (size_t)n * 4)); CK(hipMalloc(&c2, (size_t)n * 4)); CK(hipMalloc(&c3, (size_t)n * 4));
    CK(hipMalloc(&c4, (size_t)n * 4)); CK(hipMalloc(&c5, (size_t)n * 4)); CK(hipMalloc(&cnt, (size_t)G2 * 4)); CK(hipMalloc(&rk, (size_t)n * 4));
    CK(hipMalloc(&ovf, 4)); CK(hipMalloc(&box, (size_t)G2 * 8 * 16));
    { int* perm = make_perm(n); hipLaunchKernelGGL(k_setup, dim3((n + 255) / 256), dim3(256), 0, 0, n, K, perm, gidx, c1, c2, c3, c4, c5); CK(hipDeviceSynchronize()); hipFree(perm); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nb = (n / 4 + NT - 1) / NT, gb = (G2 + NT - 1) / NT;
    float best = 1e9f;
    for (int it = 0; it < 12; it++) { hipEventRecord(e0, 0);
      hipLaunchKernelGGL((k_file<0, 8>), dim3(nb), dim3(NT), 0, 0, n, gidx, c1, c2, c3, c4, c5, cnt, box, ovf);
      hipLaunchKernelGGL((k_read<8>), dim3(gb), dim3(NT), 0, 0, G2, cnt, box, rk);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
    printf("125 k groups K=5: file + read, two launches: best %7.1f us\n", best * 1000);
    unsigned* bar; CK(hipMalloc(&bar, 4)); hipMemset(bar, 0, 4);
    for (int nk : {1, 2, 4, 6}) {
      float b1 = 1e9f, b2 = 1e9f; unsigned ep = 0;
      for (int it = 0; it < 12; it++) { float ms;
        hipEventRecord(e0, 0); for (int k = 0; k < nk; k++) hipLaunchKernelGGL(k_touch, dim3(gb), dim3(NT), 0, 0, rk, G2);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); b1 = std::min(b1, ms);
        hipMemsetAsync(bar, 0, 4, 0);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k_touch_barrier, dim3(gb), dim3(NT), 0, 0, rk, G2, nk, bar, ep);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); b2 = std::min(b2, ms); }
      printf("125 k lanes (%d workgroups): %d dependent launches %6.1f us | one launch with %d grid barriers %6.1f us\n", gb, nk, b1 * 1000, nk - 1, b2 * 1000);
    }
  }
  return 0;
}
