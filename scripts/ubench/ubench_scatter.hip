// Microbenchmark: what does the MI355X give for 32-byte record scatter into NB bucket frontiers?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e_),__LINE__); return 1;}}while(0)
struct __attribute__((aligned(32))) Rec { int32_t w[8]; };
struct __attribute__((aligned(16))) I4 { int32_t x,y,z,w; };
__device__ __forceinline__ uint32_t mix(uint32_t h){h^=h>>16;h*=0x85ebca6bu;h^=h>>13;h*=0xc2b2ae35u;h^=h>>16;return h;}

__global__ void k_setup(int n, int nb, int* cnt, int* bkt){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n){int b=mix(i*2654435761u)%nb; bkt[i]=b; atomicAdd(&cnt[b],1);} }
__global__ void k_setup_sorted(int n, int nb, int* cnt, int* bkt){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n){int b=(int)((long long)i*nb/n); bkt[i]=b; atomicAdd(&cnt[b],1);} }
__global__ void k_pos(int n, const int* bkt, const int* off, int* cur, int* pos){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n){int b=bkt[i]; pos[i]=off[b]+atomicAdd(&cur[b],1);} }
// tile-ordered positions: emulate what a tile-wise partition produces (records of tile t for bucket b contiguous, tiles in order)
// timed kernels
__global__ void k_write_pos(int n, const int* __restrict__ pos, Rec* out){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n){ Rec r; for(int k=0;k<8;k++) r.w[k]=i+k; out[pos[i]]=r; } }
__global__ void k_write_pos4(int n, const int* __restrict__ pos, Rec* out){ int i0=(blockIdx.x*blockDim.x+threadIdx.x)*4; if(i0+3<n){ I4 p=*(const I4*)(pos+i0); int pp[4]={p.x,p.y,p.z,p.w}; for(int q=0;q<4;q++){ Rec r; for(int k=0;k<8;k++) r.w[k]=i0+q+k; out[pp[q]]=r;} } }
__global__ void k_read_pos(int n, const int* __restrict__ pos, const Rec* in, int* sink){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n){ Rec r=in[pos[i]]; int s=0; for(int k=0;k<8;k++) s+=r.w[k]; if(s==0x12345678) sink[0]=s; } }
__global__ void k_copy(int n, const Rec* in, Rec* out){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) out[i]=in[i]; }
// LDS-atomic partition kernel (cursor precomputed per tile): the real thing minus init
template<int NT,int ITEMS>
__global__ __launch_bounds__(NT) void k_part(int n, int nb, const int* __restrict__ bkt, const int* __restrict__ tilecur, Rec* out){
  extern __shared__ int lds[];
  const int* tc = tilecur + (size_t)blockIdx.x*nb;
  for(int b=threadIdx.x;b<nb;b+=NT) lds[b]=tc[b];
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  #pragma unroll
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; if(i<(size_t)n){ int b=bkt[i]; int p=atomicAdd(&lds[b],1); Rec r; for(int k=0;k<8;k++) r.w[k]=(int)i+k; out[p]=r; } }
}
template<int NT,int ITEMS>
__global__ __launch_bounds__(NT) void k_tilehist(int n, int nb, const int* __restrict__ bkt, int* tilecnt){
  extern __shared__ int lds[];
  for(int b=threadIdx.x;b<nb;b+=NT) lds[b]=0;
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; if(i<(size_t)n) atomicAdd(&lds[bkt[i]],1); }
  __syncthreads();
  for(int b=threadIdx.x;b<nb;b+=NT) tilecnt[(size_t)blockIdx.x*nb+b]=lds[b];
}
int main(){
  const int n=3000000;
  int *bkt,*cnt,*off,*cur,*pos,*sink; Rec *a,*b;
  CK(hipMalloc(&bkt,n*4)); CK(hipMalloc(&pos,n*4)); CK(hipMalloc(&a,(size_t)n*32)); CK(hipMalloc(&b,(size_t)n*32)); CK(hipMalloc(&sink,64));
  CK(hipMalloc(&cnt,65536*4)); CK(hipMalloc(&off,65536*4)); CK(hipMalloc(&cur,65536*4));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); char* flushbuf; CK(hipMalloc(&flushbuf,(size_t)1<<30));
  auto timeit=[&](const char* name, auto f){ f(); hipDeviceSynchronize(); float best=1e9; for(int r=0;r<5;r++){ hipMemsetAsync(flushbuf,r,(size_t)1<<30); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); best=std::min(best,ms);} printf("%-44s %8.1f us\n",name,best*1e3); };
  timeit("copy 3M x 32B (96MB r + 96MB w)",[&]{ k_copy<<<(n+255)/256,256>>>(n,a,b); });
  int nbs[]={256,977,1954,3907,15625,0};
  for(int mode=0;mode<2;mode++) for(int t=0;nbs[t];t++){
    int nb=nbs[t];
    CK(hipMemset(cnt,0,65536*4)); CK(hipMemset(cur,0,65536*4));
    if(mode==0) k_setup<<<(n+255)/256,256>>>(n,nb,cnt,bkt); else k_setup_sorted<<<(n+255)/256,256>>>(n,nb,cnt,bkt);
    std::vector<int> h(nb),o(nb); CK(hipMemcpy(h.data(),cnt,nb*4,hipMemcpyDeviceToHost)); int s=0; for(int i=0;i<nb;i++){o[i]=s;s+=h[i];} CK(hipMemcpy(off,o.data(),nb*4,hipMemcpyHostToDevice));
    k_pos<<<(n+255)/256,256>>>(n,bkt,off,cur,pos); CK(hipDeviceSynchronize());
    char nm[128];
    snprintf(nm,128,"%s nb=%5d write 32B @pos[i]",mode?"sorted":"random",nb); timeit(nm,[&]{ k_write_pos<<<(n+255)/256,256>>>(n,pos,b); });
    snprintf(nm,128,"%s nb=%5d write 32B @pos[i] x4/lane",mode?"sorted":"random",nb); timeit(nm,[&]{ k_write_pos4<<<(n/4+255)/256,256>>>(n,pos,b); });
    snprintf(nm,128,"%s nb=%5d read  32B @pos[i]",mode?"sorted":"random",nb); timeit(nm,[&]{ k_read_pos<<<(n+255)/256,256>>>(n,pos,b,sink); });
    if(nb<=3907){
      // tile partition with LDS atomics, two geometries
      {
        const int NT=256,IT=16; int nt=(n+NT*IT-1)/(NT*IT); int *tcnt; CK(hipMalloc(&tcnt,(size_t)nt*nb*4));
        k_tilehist<NT,IT><<<nt,NT,nb*4>>>(n,nb,bkt,tcnt);
        std::vector<int> tc((size_t)nt*nb); CK(hipMemcpy(tc.data(),tcnt,tc.size()*4,hipMemcpyDeviceToHost));
        std::vector<int> run(o); for(int b2=0;b2<nb;b2++){int r=o[b2]; for(int t2=0;t2<nt;t2++){int c=tc[(size_t)t2*nb+b2]; tc[(size_t)t2*nb+b2]=r; r+=c;}}
        CK(hipMemcpy(tcnt,tc.data(),tc.size()*4,hipMemcpyHostToDevice));
        snprintf(nm,128,"%s nb=%5d LDS-atomic partition 256thr x16",mode?"sorted":"random",nb); timeit(nm,[&]{ k_part<NT,IT><<<nt,NT,nb*4>>>(n,nb,bkt,tcnt,b); });
        hipFree(tcnt);
      }
      {
        const int NT=1024,IT=8; int nt=(n+NT*IT-1)/(NT*IT); int *tcnt; CK(hipMalloc(&tcnt,(size_t)nt*nb*4));
        k_tilehist<NT,IT><<<nt,NT,nb*4>>>(n,nb,bkt,tcnt);
        std::vector<int> tc((size_t)nt*nb); CK(hipMemcpy(tc.data(),tcnt,tc.size()*4,hipMemcpyDeviceToHost));
        for(int b2=0;b2<nb;b2++){int r=o[b2]; for(int t2=0;t2<nt;t2++){int c=tc[(size_t)t2*nb+b2]; tc[(size_t)t2*nb+b2]=r; r+=c;}}
        CK(hipMemcpy(tcnt,tc.data(),tc.size()*4,hipMemcpyHostToDevice));
        snprintf(nm,128,"%s nb=%5d LDS-atomic partition 1024thr x8",mode?"sorted":"random",nb); timeit(nm,[&]{ k_part<NT,IT><<<nt,NT,nb*4>>>(n,nb,bkt,tcnt,b); });
        hipFree(tcnt);
      }
    }
  }
  return 0;
}
