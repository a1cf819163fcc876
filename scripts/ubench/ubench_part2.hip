// Microbenchmark 2: tile partition (LDS-atomic cursor) with the real column count, and a 2-pass (64-way) variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e_),__LINE__); return 1;}}while(0)
struct __attribute__((aligned(32))) Rec { int32_t w[8]; };
struct __attribute__((aligned(16))) I4 { int32_t x,y,z,w; };
__device__ __forceinline__ uint32_t mix(uint32_t h){h^=h>>16;h*=0x85ebca6bu;h^=h>>13;h*=0xc2b2ae35u;h^=h>>16;return h;}
__global__ void k_setup(int n, int G, int* gidx){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) gidx[i]=mix(i*2654435761u)%G; }

// SoA (gidx + NCOL columns) -> AoS rec, bucket = gidx >> shift, cursor from tilecur matrix
template<int NT,int ITEMS,int NCOL>
__global__ __launch_bounds__(NT) void k_part_soa(int n, int nb, int shift, const int* __restrict__ gidx, const int* __restrict__ c0,const int* __restrict__ c1,const int* __restrict__ c2,const int* __restrict__ c3,const int* __restrict__ c4, const int* __restrict__ tilecur, Rec* out){
  extern __shared__ int lds[];
  const int* tc = tilecur + (size_t)blockIdx.x*nb;
  for(int b=threadIdx.x;b<nb;b+=NT) lds[b]=tc[b];
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  #pragma unroll
  for(int j=0;j<ITEMS/4;j++){
    size_t i0=base+((size_t)j*NT+threadIdx.x)*4;
    if(i0+3<(size_t)n){
      I4 g=*(const I4*)(gidx+i0); I4 a={0,0,0,0},b={0,0,0,0},c={0,0,0,0},d={0,0,0,0},e={0,0,0,0};
      if(NCOL>0) a=*(const I4*)(c0+i0); if(NCOL>1) b=*(const I4*)(c1+i0); if(NCOL>2) c=*(const I4*)(c2+i0); if(NCOL>3) d=*(const I4*)(c3+i0); if(NCOL>4) e=*(const I4*)(c4+i0);
      int gg[4]={g.x,g.y,g.z,g.w}; int aa[4]={a.x,a.y,a.z,a.w}; int bb[4]={b.x,b.y,b.z,b.w}; int cc[4]={c.x,c.y,c.z,c.w}; int dd[4]={d.x,d.y,d.z,d.w}; int ee[4]={e.x,e.y,e.z,e.w};
      #pragma unroll
      for(int q=0;q<4;q++){ int p=atomicAdd(&lds[gg[q]>>shift],1); Rec r; r.w[0]=(int)i0+q; r.w[1]=gg[q]; r.w[2]=aa[q]; r.w[3]=bb[q]; r.w[4]=cc[q]; r.w[5]=dd[q]; r.w[6]=ee[q]; r.w[7]=0; out[p]=r; }
    }
  }
}
// AoS -> AoS, bucket = (rec.w[1] >> shift) (second pass)
template<int NT,int ITEMS>
__global__ __launch_bounds__(NT) void k_part_aos(int n, int nb, int shift, int mask, const Rec* __restrict__ in, const int* __restrict__ tilecur, Rec* out){
  extern __shared__ int lds[];
  const int* tc = tilecur + (size_t)blockIdx.x*nb;
  for(int b=threadIdx.x;b<nb;b+=NT) lds[b]=tc[b];
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  #pragma unroll
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; if(i<(size_t)n){ Rec r=in[i]; int p=atomicAdd(&lds[(r.w[1]>>shift)&mask],1); out[p]=r; } }
}
template<int NT,int ITEMS>
__global__ __launch_bounds__(NT) void k_tilehist(int n, int nb, int shift, const int* __restrict__ gidx, int* tilecnt){
  extern __shared__ int lds[];
  for(int b=threadIdx.x;b<nb;b+=NT) lds[b]=0;
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; if(i<(size_t)n) atomicAdd(&lds[gidx[i]>>shift],1); }
  __syncthreads();
  for(int b=threadIdx.x;b<nb;b+=NT) tilecnt[(size_t)blockIdx.x*nb+b]=lds[b];
}
template<int NT,int ITEMS>
__global__ __launch_bounds__(NT) void k_tilehist_aos(int n, int nb, int shift, const Rec* __restrict__ in, int* tilecnt){
  extern __shared__ int lds[];
  for(int b=threadIdx.x;b<nb;b+=NT) lds[b]=0;
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; if(i<(size_t)n) atomicAdd(&lds[in[i].w[1]>>shift],1); }
  __syncthreads();
  for(int b=threadIdx.x;b<nb;b+=NT) tilecnt[(size_t)blockIdx.x*nb+b]=lds[b];
}
static void colscan(std::vector<int>& tc,int nt,int nb){ long long run=0; // bucket-major global exclusive prefix
  std::vector<long long> tot(nb,0); for(int t=0;t<nt;t++) for(int b=0;b<nb;b++) tot[b]+=tc[(size_t)t*nb+b];
  std::vector<long long> off(nb); for(int b=0;b<nb;b++){off[b]=run;run+=tot[b];}
  for(int b=0;b<nb;b++){ long long r=off[b]; for(int t=0;t<nt;t++){ int c=tc[(size_t)t*nb+b]; tc[(size_t)t*nb+b]=(int)r; r+=c; } } }
int main(){
  const int n=3000000, G=1000000;
  int *gidx,*col[5]; Rec *a,*b; CK(hipMalloc(&gidx,n*4)); for(int k=0;k<5;k++){ CK(hipMalloc(&col[k],n*4)); CK(hipMemset(col[k],k+1,n*4)); }
  CK(hipMalloc(&a,(size_t)n*32)); CK(hipMalloc(&b,(size_t)n*32));
  k_setup<<<(n+255)/256,256>>>(n,G,gidx); CK(hipDeviceSynchronize());
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); char* flushbuf; CK(hipMalloc(&flushbuf,(size_t)1<<30));
  auto timeit=[&](const char* name, auto f){ f(); hipDeviceSynchronize(); float best=1e9; for(int r=0;r<5;r++){ hipMemsetAsync(flushbuf,r,(size_t)1<<30); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); best=std::min(best,ms);} printf("%-64s %8.1f us\n",name,best*1e3); };
#define RUN_SOA(NT,IT,NCOL,SHIFT,LABEL) { int nb=((G-1)>>SHIFT)+1; int nt=(n+NT*IT-1)/(NT*IT); int* tcnt; CK(hipMalloc(&tcnt,(size_t)nt*nb*4)); \
    k_tilehist<NT,IT><<<nt,NT,nb*4>>>(n,nb,SHIFT,gidx,tcnt); std::vector<int> tc((size_t)nt*nb); CK(hipMemcpy(tc.data(),tcnt,tc.size()*4,hipMemcpyDeviceToHost)); colscan(tc,nt,nb); CK(hipMemcpy(tcnt,tc.data(),tc.size()*4,hipMemcpyHostToDevice)); \
    char nm[128]; snprintf(nm,128,"%s nb=%d %dthr x%d ncol=%d",LABEL,nb,NT,IT,NCOL); timeit(nm,[&]{ k_part_soa<NT,IT,NCOL><<<nt,NT,nb*4>>>(n,nb,SHIFT,gidx,col[0],col[1],col[2],col[3],col[4],tcnt,a); }); hipFree(tcnt); }
  RUN_SOA(1024,8,0,8,"SoA->AoS")
  RUN_SOA(1024,8,5,8,"SoA->AoS")
  RUN_SOA(256,16,5,8,"SoA->AoS")
  RUN_SOA(512,8,5,8,"SoA->AoS")
  RUN_SOA(1024,8,5,10,"SoA->AoS")
  RUN_SOA(1024,8,5,12,"SoA->AoS")
  RUN_SOA(1024,8,5,14,"SoA->AoS (pass 1 of 2: 62 coarse buckets)")
  RUN_SOA(256,16,5,14,"SoA->AoS (pass 1 of 2: 62 coarse buckets)")
  RUN_SOA(512,8,5,14,"SoA->AoS (pass 1 of 2: 62 coarse buckets)")
  // pass 2: a is now coarse-partitioned (shift 14). split by fine bucket (shift 8) -> b
  { const int NT=1024,IT=8; int nb=((G-1)>>8)+1; int nt=(n+NT*IT-1)/(NT*IT); int* tcnt; CK(hipMalloc(&tcnt,(size_t)nt*nb*4));
    k_tilehist_aos<NT,IT><<<nt,NT,nb*4>>>(n,nb,8,a,tcnt); std::vector<int> tc((size_t)nt*nb); CK(hipMemcpy(tc.data(),tcnt,tc.size()*4,hipMemcpyDeviceToHost)); colscan(tc,nt,nb); CK(hipMemcpy(tcnt,tc.data(),tc.size()*4,hipMemcpyHostToDevice));
    timeit("AoS->AoS pass 2 of 2 (fine, 3907 cursors/tile) 1024thr x8",[&]{ k_part_aos<NT,IT><<<nt,NT,nb*4>>>(n,nb,8,0xffffff,a,tcnt,b); }); hipFree(tcnt); }
  { const int NT=256,IT=16; int nb=((G-1)>>8)+1; int nt=(n+NT*IT-1)/(NT*IT); int* tcnt; CK(hipMalloc(&tcnt,(size_t)nt*nb*4));
    k_tilehist_aos<NT,IT><<<nt,NT,nb*4>>>(n,nb,8,a,tcnt); std::vector<int> tc((size_t)nt*nb); CK(hipMemcpy(tc.data(),tcnt,tc.size()*4,hipMemcpyDeviceToHost)); colscan(tc,nt,nb); CK(hipMemcpy(tcnt,tc.data(),tc.size()*4,hipMemcpyHostToDevice));
    timeit("AoS->AoS pass 2 of 2 (fine, 3907 cursors/tile) 256thr x16",[&]{ k_part_aos<NT,IT><<<nt,NT,nb*4>>>(n,nb,8,0xffffff,a,tcnt,b); }); hipFree(tcnt); }
  return 0;
}
