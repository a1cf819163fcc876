// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS engine's access patterns (VERDICT r5 item 3;
// MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other
// access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel moves a byte count known in advance, over buffers larger than the 256 MB Infinity Cache; the program
// prints the true counts as JSON; scripts/counter_calibration.py divides the counters of the rocprofv3 passes by them.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_counters ubench_counters.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -- ./ubench_counters > true_bytes.json   (and --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct __attribute__((aligned(16))) I4 { int32_t x, y, z, w; };
struct __attribute__((aligned(8))) I2 { int32_t x, y; };
#define NT 512

// (0) the guide's calibrated case: 16 bytes per lane, streaming
__global__ __launch_bounds__(NT) void k_cal_read16(const I4* __restrict__ in, int64_t n16, int32_t* sink) {
  int32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n16; i += (int64_t)gridDim.x * NT) { const I4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x7fffffff) *sink = acc;
}
__global__ __launch_bounds__(NT) void k_cal_write16(I4* __restrict__ out, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n16; i += (int64_t)gridDim.x * NT) out[i] = I4{(int32_t)i, 1, 2, 3};
}
// (i) 8-byte entries, eight lanes to a 64-byte line, the lines in scattered order (the per-bucket kernel reading the
// runs of sorted tiles; round 5's slot lines): every line exactly once
__global__ __launch_bounds__(NT) void k_cal_read8_lines(const I2* __restrict__ in, int64_t nlines_log2, int32_t* sink) {
  const int64_t nl = (int64_t)1 << nlines_log2;
  int32_t acc = 0;
  for (int64_t q = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 3; q < nl; q += ((int64_t)gridDim.x * NT) >> 3) {
    const int64_t line = (q * 0x9E3779B1ll + 12345) & (nl - 1);  // odd multiplier: a permutation of the lines
    const I2 v = in[line * 8 + (threadIdx.x & 7)];
    acc += v.x ^ v.y;
  }
  if (acc == 0x7fffffff) *sink = acc;
}
__global__ __launch_bounds__(NT) void k_cal_write8_lines(I2* __restrict__ out, int64_t nlines_log2) {
  const int64_t nl = (int64_t)1 << nlines_log2;
  for (int64_t q = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 3; q < nl; q += ((int64_t)gridDim.x * NT) >> 3) {
    const int64_t line = (q * 0x9E3779B1ll + 12345) & (nl - 1);
    out[line * 8 + (threadIdx.x & 7)] = I2{(int32_t)q, 7};
  }
}
// (ii) 4-byte state columns by group: one lane per group, a workgroup per 512 consecutive groups, twelve columns
// (what coord_preload reads per group at K = 3), workgroups in dispatch order
__global__ __launch_bounds__(NT) void k_cal_read4_cols(const int32_t* __restrict__ base, int64_t G, int32_t ncols, int32_t* sink) {
  int32_t acc = 0;
  for (int64_t g = (int64_t)blockIdx.x * NT + threadIdx.x; g < G; g += (int64_t)gridDim.x * NT)
    for (int c = 0; c < ncols; c++) acc ^= base[(int64_t)c * G + g];
  if (acc == 0x7fffffff) *sink = acc;
}
__global__ __launch_bounds__(NT) void k_cal_write4_cols(int32_t* __restrict__ base, int64_t G, int32_t ncols) {
  for (int64_t g = (int64_t)blockIdx.x * NT + threadIdx.x; g < G; g += (int64_t)gridDim.x * NT)
    for (int c = 0; c < ncols; c++) base[(int64_t)c * G + g] = (int32_t)g + c;
}
// (iii) rows of 1-byte counts (round 5) / 2-byte run starts (round 6), one lane per entry, dense
__global__ __launch_bounds__(NT) void k_cal_read1_rows(const uint8_t* __restrict__ in, int64_t n, int32_t* sink) {
  int32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) acc += in[i];
  if (acc == 0x7fffffff) *sink = acc;
}
__global__ __launch_bounds__(NT) void k_cal_read2_rows(const uint16_t* __restrict__ in, int64_t n, int32_t* sink) {
  int32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) acc += in[i];
  if (acc == 0x7fffffff) *sink = acc;
}
// (iv) a COLUMN of a 2-byte matrix per workgroup (k_scatter_tiles writing A.off): entry [b][w] for every row b, rows
// `pad` entries apart - partial lines, every line shared by 32 workgroups
__global__ __launch_bounds__(NT) void k_cal_write2_column(uint16_t* __restrict__ out, int32_t rows, int32_t pad) {
  const int32_t w = blockIdx.x;
  for (int32_t b = threadIdx.x; b < rows; b += NT) out[(int64_t)b * pad + w] = (uint16_t)b;
}
// (v) 4-byte gathers at random (the escape path: a vote's fields by arrival index)
__global__ __launch_bounds__(NT) void k_cal_gather4(const int32_t* __restrict__ in, int64_t n_log2, int64_t count, int32_t* sink) {
  const int64_t n = (int64_t)1 << n_log2;
  int32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < count; i += (int64_t)gridDim.x * NT) {
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    acc ^= in[(int64_t)(h & (uint64_t)(n - 1))];
  }
  if (acc == 0x7fffffff) *sink = acc;
}

int main() {
  const int64_t B = (int64_t)1 << 30;  // 1 GiB per buffer: four times the Infinity Cache
  char *a = nullptr, *b = nullptr;
  int32_t* sink = nullptr;
  CK(hipMalloc(&a, B));
  CK(hipMalloc(&b, B));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, B));
  CK(hipMemset(b, 1, B));
  CK(hipDeviceSynchronize());
  const int grid = 256 * 8;
  const int reps = 3;
  const int64_t half = B / 2;
  const int64_t G = 16 << 20;               // 16 M groups x 12 columns x 4 B = 768 MB
  const int32_t rows = 4096, pad = 1024;    // 4096 x 1024 x 2 B = 8 MB matrix, 1024 column writers
  const int64_t gathers = (int64_t)64 << 20;
  for (int r = 0; r < reps; r++) {
    // each pair alternates the two buffers so that nothing a kernel reads was left in a cache by its predecessor
    hipLaunchKernelGGL(k_cal_read16, dim3(grid), dim3(NT), 0, 0, (const I4*)a, half / 16, sink);
    hipLaunchKernelGGL(k_cal_write16, dim3(grid), dim3(NT), 0, 0, (I4*)b, half / 16);
    hipLaunchKernelGGL(k_cal_read8_lines, dim3(grid), dim3(NT), 0, 0, (const I2*)(a + half), (int64_t)23, sink);  // 2^23 lines = 512 MB
    hipLaunchKernelGGL(k_cal_write8_lines, dim3(grid), dim3(NT), 0, 0, (I2*)(b + half), (int64_t)23);
    hipLaunchKernelGGL(k_cal_read4_cols, dim3(grid), dim3(NT), 0, 0, (const int32_t*)a, G, 12, sink);
    hipLaunchKernelGGL(k_cal_write4_cols, dim3(grid), dim3(NT), 0, 0, (int32_t*)b, G, 12);
    hipLaunchKernelGGL(k_cal_read1_rows, dim3(grid), dim3(NT), 0, 0, (const uint8_t*)a, half, sink);
    hipLaunchKernelGGL(k_cal_read2_rows, dim3(grid), dim3(NT), 0, 0, (const uint16_t*)(a + half), half / 2, sink);
    hipLaunchKernelGGL(k_cal_write2_column, dim3(pad), dim3(NT), 0, 0, (uint16_t*)b, rows, pad);
    hipLaunchKernelGGL(k_cal_gather4, dim3(grid), dim3(NT), 0, 0, (const int32_t*)a, (int64_t)28, gathers, sink);
    CK(hipDeviceSynchronize());
  }
  printf("{\n");
  printf(" \"k_cal_read16\": {\"read\": %lld, \"write\": 0, \"what\": \"16 B per lane, streaming (the guide's calibrated case)\"},\n", (long long)half);
  printf(" \"k_cal_write16\": {\"read\": 0, \"write\": %lld, \"what\": \"16 B per lane, streaming stores\"},\n", (long long)half);
  printf(" \"k_cal_read8_lines\": {\"read\": %lld, \"write\": 0, \"what\": \"8 B per lane, eight lanes to a 64-byte line, lines in scattered order\"},\n", (long long)((int64_t)64 << 23));
  printf(" \"k_cal_write8_lines\": {\"read\": 0, \"write\": %lld, \"what\": \"the same, stores\"},\n", (long long)((int64_t)64 << 23));
  printf(" \"k_cal_read4_cols\": {\"read\": %lld, \"write\": 0, \"what\": \"4 B per lane, twelve state columns by group (coalesced 256 B per wave and column)\"},\n", (long long)(G * 12 * 4));
  printf(" \"k_cal_write4_cols\": {\"read\": 0, \"write\": %lld, \"what\": \"the same, stores\"},\n", (long long)(G * 12 * 4));
  printf(" \"k_cal_read1_rows\": {\"read\": %lld, \"write\": 0, \"what\": \"1 B per lane, dense rows\"},\n", (long long)half);
  printf(" \"k_cal_read2_rows\": {\"read\": %lld, \"write\": 0, \"what\": \"2 B per lane, dense rows\"},\n", (long long)half);
  printf(" \"k_cal_write2_column\": {\"read\": 0, \"write\": %lld, \"lines_touched_bytes\": %lld, \"what\": \"2-byte entries, one matrix column per workgroup (partial lines shared by 32 workgroups)\"},\n",
         (long long)((int64_t)rows * pad * 2), (long long)((int64_t)rows * pad * 2));
  printf(" \"k_cal_gather4\": {\"read\": %lld, \"sectors32_bytes\": %lld, \"lines64_bytes\": %lld, \"write\": 0, \"what\": \"4-byte gathers at random over 1 GiB\"}\n",
         (long long)(gathers * 4), (long long)(gathers * 32), (long long)(gathers * 64));
  printf("}\n");
  return 0;
}
