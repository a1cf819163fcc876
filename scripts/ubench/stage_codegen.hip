#include <hip/hip_runtime.h>
#include <stdint.h>
#define NCH 7
template <int BLOCK, int VAR>
__device__ __forceinline__ void stage(uint32_t* lds, const uint4* src16, int32_t c_lo, int32_t c_hi) {
  uint4* dst16 = (uint4*)lds;
  for (int32_t c0 = c_lo; c0 < c_hi; c0 += NCH * BLOCK) {
    uint4 v[NCH];
    if (VAR == 0) {
#pragma unroll
      for (int k = 0; k < NCH; k++) {
        const int32_t c = c0 + k * BLOCK + (int32_t)threadIdx.x;
        if (c < c_hi) v[k] = src16[c];
      }
    } else if (VAR == 1) {   /* zero-initialised */
#pragma unroll
      for (int k = 0; k < NCH; k++) {
        const int32_t c = c0 + k * BLOCK + (int32_t)threadIdx.x;
        v[k] = make_uint4(0, 0, 0, 0);
        if (c < c_hi) v[k] = src16[c];
      }
    } else {                 /* clamped index: unconditional loads */
#pragma unroll
      for (int k = 0; k < NCH; k++) {
        const int32_t c = c0 + k * BLOCK + (int32_t)threadIdx.x;
        v[k] = src16[c < c_hi ? c : c_hi - 1];
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; k++) {
      const int32_t c = c0 + k * BLOCK + (int32_t)threadIdx.x;
      if (c < c_hi) dst16[c] = v[k];
    }
  }
}
template <int VAR>
__global__ __launch_bounds__(512) void k(const uint4* src, int32_t c_lo, int32_t c_hi, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[12288 + 8];
  stage<512, VAR>(lds, src, c_lo, c_hi);
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = lds[(threadIdx.x * 7) % 12288];
}
template __global__ void k<0>(const uint4*, int32_t, int32_t, uint32_t*);
template __global__ void k<1>(const uint4*, int32_t, int32_t, uint32_t*);
template __global__ void k<2>(const uint4*, int32_t, int32_t, uint32_t*);
