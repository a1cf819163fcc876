// Microbenchmark: does reordering a tile's records by bucket in LDS before writing them (so that
// the records of one (tile, bucket) pair leave as ONE contiguous run written by adjacent lanes)
// beat the direct LDS-cursor scatter?  3 M records, 6 int32 input columns -> 32-byte records.
//   direct   : k_part      (what k_scatter_ar does today)
//   staged   : k_part_lds  (tile staged in LDS as full 32-byte records, T = NT*ITEMS)
//   gather   : k_part_perm (LDS holds only a permutation; columns re-gathered from L2)
// hipcc --offload-arch=gfx950 -O3 -o ubench_runs ubench_runs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e_),__LINE__); return 1;}}while(0)
struct __attribute__((aligned(32))) Rec { int32_t idx, lg, a, b, c, bnum, bcoord, pad; };
__device__ __forceinline__ uint32_t mix(uint32_t h){h^=h>>16;h*=0x85ebca6bu;h^=h>>13;h*=0xc2b2ae35u;h^=h>>16;return h;}

__global__ void k_setup(int n, int G, int* g){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) g[i]=mix(i*2654435761u+12345u)%G; }

template<int NT,int ITEMS>
__global__ __launch_bounds__(NT) void k_tilehist(int n, int nb, int shift, const int* __restrict__ g, int* tilecnt){
  extern __shared__ int lds[];
  for(int b=threadIdx.x;b<nb;b+=NT) lds[b]=0;
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; if(i<(size_t)n) atomicAdd(&lds[g[i]>>shift],1); }
  __syncthreads();
  for(int b=threadIdx.x;b<nb;b+=NT) tilecnt[(size_t)blockIdx.x*nb+b]=lds[b];
}

// direct: LDS cursor per bucket, record written straight to its global position
template<int NT,int ITEMS>
__global__ __launch_bounds__(NT) void k_part(int n, int nb, int shift, const int* __restrict__ g, const int* __restrict__ c1,
    const int* __restrict__ c2, const int* __restrict__ c3, const int* __restrict__ c4, const int* __restrict__ c5,
    const int* __restrict__ tilecur, Rec* out){
  extern __shared__ int lds[];
  const int* tc = tilecur + (size_t)blockIdx.x*nb;
  for(int b=threadIdx.x;b<nb;b+=NT) lds[b]=tc[b];
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  #pragma unroll
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; if(i<(size_t)n){ int gg=g[i]; int p=atomicAdd(&lds[gg>>shift],1);
    Rec r; r.idx=(int)i; r.lg=gg&((1<<shift)-1); r.a=c1[i]; r.b=c2[i]; r.c=c3[i]; r.bnum=c4[i]; r.bcoord=c5[i]; r.pad=0; out[p]=r; } }
}

// staged: count (rank per record), scan, stage the full records in LDS bucket-major with their global
// destination, then stream the staged tile out: adjacent lanes -> adjacent destinations inside a run
template<int NT,int ITEMS,int NBP>
__global__ __launch_bounds__(NT) void k_part_lds(int n, int nb, int shift, const int* __restrict__ g, const int* __restrict__ c1,
    const int* __restrict__ c2, const int* __restrict__ c3, const int* __restrict__ c4, const int* __restrict__ c5,
    const int* __restrict__ tilecur, Rec* out){
  extern __shared__ __attribute__((aligned(32))) int lds[];
  int* cnt = lds;            // [nbp]
  int* loc = lds + NBP;     // [nbp]
  Rec* stage = (Rec*)(lds + 2*NBP);
  __shared__ int wsum[NT/64];
  for(int b=threadIdx.x;b<NBP;b+=NT) cnt[b]=0;
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  int gg[ITEMS], rk[ITEMS];
  #pragma unroll
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; gg[j]=-1; rk[j]=0; if(i<(size_t)n){ gg[j]=g[i]; rk[j]=atomicAdd(&cnt[gg[j]>>shift],1);} }
  __syncthreads();
  // exclusive scan of cnt[0..4096) : thread t owns 4096/NT consecutive buckets
  { const int per=NBP/NT; int s=0; int v[NBP/NT];
    #pragma unroll
    for(int q=0;q<per;q++){ v[q]=cnt[threadIdx.x*per+q]; s+=v[q]; }
    int x=s; const int lane=threadIdx.x&63, wid=threadIdx.x>>6;
    #pragma unroll
    for(int d=1;d<64;d<<=1){ int y=__shfl_up(x,d,64); if(lane>=d) x+=y; }
    if(lane==63) wsum[wid]=x;
    __syncthreads();
    int bs=0; for(int w=0;w<wid;w++) bs+=wsum[w];
    int ex=bs+x-s;
    #pragma unroll
    for(int q=0;q<per;q++){ loc[threadIdx.x*per+q]=ex; ex+=v[q]; }
  }
  __syncthreads();
  const int* tc = tilecur + (size_t)blockIdx.x*nb;
  #pragma unroll
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; if(gg[j]>=0){ int b=gg[j]>>shift;
    Rec r; r.idx=(int)i; r.lg=gg[j]&((1<<shift)-1); r.a=c1[i]; r.b=c2[i]; r.c=c3[i]; r.bnum=c4[i]; r.bcoord=c5[i]; r.pad=tc[b]+rk[j];
    stage[loc[b]+rk[j]]=r; } }
  __syncthreads();
  const int tot = min((int)(NT*ITEMS), (int)((size_t)n-base));
  for(int p=threadIdx.x;p<tot;p+=NT){ Rec r=stage[p]; out[r.pad]=r; }
}

// gather: LDS holds (source index, destination) per sorted position; payload columns are re-read
template<int NT,int ITEMS,int NBP>
__global__ __launch_bounds__(NT) void k_part_perm(int n, int nb, int shift, const int* __restrict__ g, const int* __restrict__ c1,
    const int* __restrict__ c2, const int* __restrict__ c3, const int* __restrict__ c4, const int* __restrict__ c5,
    const int* __restrict__ tilecur, Rec* out){
  extern __shared__ __attribute__((aligned(32))) int lds[];
  int* cnt = lds; int* loc = lds + NBP; int2* perm=(int2*)(lds+2*NBP);
  __shared__ int wsum[NT/64];
  for(int b=threadIdx.x;b<NBP;b+=NT) cnt[b]=0;
  __syncthreads();
  size_t base=(size_t)blockIdx.x*NT*ITEMS;
  int gg[ITEMS], rk[ITEMS];
  #pragma unroll
  for(int j=0;j<ITEMS;j++){ size_t i=base+(size_t)j*NT+threadIdx.x; gg[j]=-1; rk[j]=0; if(i<(size_t)n){ gg[j]=g[i]; rk[j]=atomicAdd(&cnt[gg[j]>>shift],1);} }
  __syncthreads();
  { const int per=NBP/NT; int s=0; int v[NBP/NT];
    #pragma unroll
    for(int q=0;q<per;q++){ v[q]=cnt[threadIdx.x*per+q]; s+=v[q]; }
    int x=s; const int lane=threadIdx.x&63, wid=threadIdx.x>>6;
    #pragma unroll
    for(int d=1;d<64;d<<=1){ int y=__shfl_up(x,d,64); if(lane>=d) x+=y; }
    if(lane==63) wsum[wid]=x;
    __syncthreads();
    int bs=0; for(int w=0;w<wid;w++) bs+=wsum[w];
    int ex=bs+x-s;
    #pragma unroll
    for(int q=0;q<per;q++){ loc[threadIdx.x*per+q]=ex; ex+=v[q]; }
  }
  __syncthreads();
  const int* tc = tilecur + (size_t)blockIdx.x*nb;
  #pragma unroll
  for(int j=0;j<ITEMS;j++){ if(gg[j]>=0){ int b=gg[j]>>shift; perm[loc[b]+rk[j]]=make_int2(j*NT+threadIdx.x, tc[b]+rk[j]); } }
  __syncthreads();
  const int tot = min((int)(NT*ITEMS), (int)((size_t)n-base));
  for(int p=threadIdx.x;p<tot;p+=NT){ int2 pd=perm[p]; size_t i=base+pd.x; int g0=g[i];
    Rec r; r.idx=(int)i; r.lg=g0&((1<<shift)-1); r.a=c1[i]; r.b=c2[i]; r.c=c3[i]; r.bnum=c4[i]; r.bcoord=c5[i]; r.pad=0; out[pd.y]=r; }
}

template<int NT,int ITEMS,int WHICH,int NBP>
int run(const char* label, int n, int G, int shift, const int* g, int* const* cols, Rec* out, hipEvent_t e0, hipEvent_t e1, char* flushbuf){
  const int nb=(G+(1<<shift)-1)>>shift; const int T=NT*ITEMS; const int nt=(n+T-1)/T;
  int* tcnt; CK(hipMalloc(&tcnt,(size_t)nt*nb*4));
  k_tilehist<NT,ITEMS><<<nt,NT,nb*4>>>(n,nb,shift,g,tcnt);
  std::vector<int> tc((size_t)nt*nb); CK(hipMemcpy(tc.data(),tcnt,tc.size()*4,hipMemcpyDeviceToHost));
  { int r=0; for(int b=0;b<nb;b++) for(int t=0;t<nt;t++){ int c=tc[(size_t)t*nb+b]; tc[(size_t)t*nb+b]=r; r+=c; } }
  CK(hipMemcpy(tcnt,tc.data(),tc.size()*4,hipMemcpyHostToDevice));
  size_t lds = WHICH==0 ? (size_t)nb*4 : (WHICH==1 ? 2*NBP*4+(size_t)T*32 : 2*NBP*4+(size_t)T*8);
  auto f=[&]{
    if(WHICH==0) k_part<NT,ITEMS><<<nt,NT,lds>>>(n,nb,shift,g,cols[0],cols[1],cols[2],cols[3],cols[4],tcnt,out);
    if(WHICH==1) k_part_lds<NT,ITEMS,NBP><<<nt,NT,lds>>>(n,nb,shift,g,cols[0],cols[1],cols[2],cols[3],cols[4],tcnt,out);
    if(WHICH==2) k_part_perm<NT,ITEMS,NBP><<<nt,NT,lds>>>(n,nb,shift,g,cols[0],cols[1],cols[2],cols[3],cols[4],tcnt,out);
  };
  if(WHICH==1) CK(hipFuncSetAttribute((const void*)k_part_lds<NT,ITEMS,NBP>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)lds));
  if(WHICH==2) CK(hipFuncSetAttribute((const void*)k_part_perm<NT,ITEMS,NBP>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)lds));
  f(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
  float best=1e9;
  for(int r=0;r<5;r++){ hipMemsetAsync(flushbuf,r,(size_t)1<<30); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); best=std::min(best,ms); }
  // checksum: every destination written exactly once
  printf("%-10s NT=%4d T=%5d gb=%5d nb=%5d run~%.1f recs  %8.1f us\n",label,NT,T,1<<shift,nb,(double)T/nb,best*1e3);
  hipFree(tcnt);
  return 0;
}

int main(){
  const int n=3000000, G=1000000;
  int* g; int* cols[5]; Rec* out; char* flushbuf;
  CK(hipMalloc(&g,n*4)); for(int c=0;c<5;c++){ CK(hipMalloc(&cols[c],n*4)); CK(hipMemset(cols[c],c,n*4)); }
  CK(hipMalloc(&out,(size_t)n*32)); CK(hipMalloc(&flushbuf,(size_t)1<<30));
  k_setup<<<(n+255)/256,256>>>(n,G,g); CK(hipDeviceSynchronize());
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {
    const int shift=8;
    run<1024,8,0,4096>("direct",n,G,shift,g,cols,out,e0,e1,flushbuf);
    run<1024,3,1,4096>("staged",n,G,shift,g,cols,out,e0,e1,flushbuf);
    run<1024,8,2,4096>("gather",n,G,shift,g,cols,out,e0,e1,flushbuf);
  }
  for(int shift : {10,11,12}){
    run<1024,8,0,1024>("direct",n,G,shift,g,cols,out,e0,e1,flushbuf);
    run<1024,4,0,1024>("direct",n,G,shift,g,cols,out,e0,e1,flushbuf);
    run<1024,4,1,1024>("staged",n,G,shift,g,cols,out,e0,e1,flushbuf);
    run<1024,2,1,1024>("staged",n,G,shift,g,cols,out,e0,e1,flushbuf);
    run<512,4,1,1024>("staged",n,G,shift,g,cols,out,e0,e1,flushbuf);
    run<1024,8,2,1024>("gather",n,G,shift,g,cols,out,e0,e1,flushbuf);
    run<1024,16,2,1024>("gather",n,G,shift,g,cols,out,e0,e1,flushbuf);
  }
  return 0;
}
