// What an LDS access costs one CU when the addresses are random (the counting sort of k_scatter_tiles: a returning atomic
// per vote on ~2000 counters, then one 8-byte store per vote at its sorted position), against the same accesses in lane
// order.  One 1024-thread workgroup per CU, every wave issues OPS accesses; cycles by clock64 around the loop of wave 0.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_lds ubench_lds.hip && ./ubench_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define OPS 64
#define WORDS (24 * 1024) /* 96 KB */
struct __attribute__((aligned(8))) I2 { uint32_t x, y; };
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k_lds(long long* out, int counters) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < WORDS; i += NT) lds[i] = 0;
  __syncthreads();
  uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 977u + 12345u;
  uint32_t acc = 0;
  const long long t0 = clock64();
#pragma unroll 8
  for (int i = 0; i < OPS; i++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t r = h >> 8;
    if (MODE == 0) ((I2*)lds)[(r % (WORDS / 2))] = I2{r, acc};                               // random 8-byte stores
    if (MODE == 1) ((I2*)lds)[(threadIdx.x + i * NT) % (WORDS / 2)] = I2{r, acc};            // 8-byte stores in lane order
    if (MODE == 2) acc += lds[r % counters];                                                 // random 4-byte loads (dependent sum only)
    if (MODE == 3) acc += atomicAdd(&lds[r % counters], 1u);                                 // random returning atomics
    if (MODE == 4) atomicAdd(&lds[r % counters], 1u);                                        // random atomics, no return
    if (MODE == 5) acc += lds[(threadIdx.x + i * NT) % WORDS];                               // 4-byte loads in lane order
    if (MODE == 7) *(I2*)((char*)lds + 4 + 8 * (r % (WORDS / 2 - 1))) = I2{r, acc};            // random 8-byte stores, 4 bytes off alignment
    if (MODE == 8) { const uint4 v = *(const uint4*)((char*)lds + 4 + 16 * ((threadIdx.x + i * NT) % (WORDS / 4 - 1))); acc += v.x ^ v.y ^ v.z ^ v.w; }  // 16-byte loads in lane order, 4 bytes off
    if (MODE == 9) { const uint4 v = *(const uint4*)((char*)lds + 16 * ((threadIdx.x + i * NT) % (WORDS / 4 - 1))); acc += v.x ^ v.y ^ v.z ^ v.w; }      // ... aligned
    if (MODE == 6) { const uint32_t p = lds[r % counters]; ((I2*)lds)[(p + r) % (WORDS / 2)] = I2{r, acc}; }  // load, then store at a dependent place
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
template <int MODE, int NT>
static void run(const char* what, long long* d_out, int counters) {
  const int grid = 256;
  long long h[256];
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lds<MODE, NT>), dim3(grid), dim3(NT), WORDS * 4, 0, d_out, counters);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lds<MODE, NT>), dim3(grid), dim3(NT), WORDS * 4, 0, d_out, counters);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
  double s = 0;
  for (int i = 0; i < grid; i++) s += (double)h[i];
  const double cyc = s / grid;
  const double waveops = (double)OPS * (NT / 64);
  printf("%-62s NT %4d  %8.0f cycles per workgroup  = %6.1f cycles per wave-instruction, %5.2f lanes per cycle\n", what, NT, cyc,
         cyc / waveops, 64.0 * waveops / cyc);
}
int main() {
  long long* d_out;
  CK(hipMalloc(&d_out, 256 * sizeof(long long)));
  CK(hipFuncSetAttribute((const void*)k_lds<0, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, WORDS * 4));
#define ATTR(M, N) CK(hipFuncSetAttribute((const void*)k_lds<M, N>, hipFuncAttributeMaxDynamicSharedMemorySize, WORDS * 4))
  ATTR(7, 1024); ATTR(8, 1024); ATTR(9, 1024);
  ATTR(1, 1024); ATTR(2, 1024); ATTR(3, 1024); ATTR(4, 1024); ATTR(5, 1024); ATTR(6, 1024); ATTR(0, 256); ATTR(3, 256); ATTR(6, 256);
  run<0, 1024>("8-byte stores at random", d_out, 2048);
  run<7, 1024>("8-byte stores at random, the block 4 bytes off alignment", d_out, 2048);
  run<1, 1024>("8-byte stores in lane order", d_out, 2048);
  run<9, 1024>("16-byte loads in lane order", d_out, 2048);
  run<8, 1024>("16-byte loads in lane order, the block 4 bytes off alignment", d_out, 2048);
  run<2, 1024>("4-byte loads at random over 2048 words", d_out, 2048);
  run<5, 1024>("4-byte loads in lane order", d_out, 2048);
  run<3, 1024>("returning atomics at random over 2048 counters", d_out, 2048);
  run<3, 1024>("returning atomics at random over 256 counters", d_out, 256);
  run<4, 1024>("atomics without return over 2048 counters", d_out, 2048);
  run<6, 1024>("4-byte load at random, then an 8-byte store at a dependent place", d_out, 2048);
  run<0, 256>("8-byte stores at random", d_out, 2048);
  run<3, 256>("returning atomics at random over 2048 counters", d_out, 2048);
  run<6, 256>("4-byte load at random, then an 8-byte store at a dependent place", d_out, 2048);
  return 0;
}
