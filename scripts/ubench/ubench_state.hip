// Microbenchmark 4: per-group state access of the per-bucket kernels, one lane per group, 1 M groups.
//   soa   : 12 loads from 12 int32 columns + 5 stores to 5 columns (today's DevState for K = 3)
//   aos16 : flags(4) + coord I4 + members I4 + node_slots I4 + p_ring(4) loads; p_ring(4) + node_slots I4 +
//           coord I4 stores (4 members / node slots per 16-byte entry)
// Same bytes useful, different number of memory instructions per lane.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e_),__LINE__); return 1;}}while(0)
struct __attribute__((aligned(16))) I4 { int32_t x,y,z,w; };
struct Soa { int32_t *f,*bn,*bc,*nx,*pc,*m0,*m1,*m2,*n0,*n1,*n2,*pr; };
struct Aos { int32_t* f; I4 *co,*me,*ns; int32_t* pr; };

__global__ __launch_bounds__(256) void k_soa(int G, Soa S, int W, int salt){
  int g=blockIdx.x*256+threadIdx.x; if(g>=G) return;
  int f=S.f[g], bn=S.bn[g], bc=S.bc[g], nx=S.nx[g], pc=S.pc[g];
  int m0=S.m0[g], m1=S.m1[g], m2=S.m2[g], n0=S.n0[g], n1=S.n1[g], n2=S.n2[g];
  int pe=S.pr[(size_t)((nx-1)&(W-1))*G+g];
  int v=f+bn+bc+m0+m1+m2+pe+salt;
  S.pr[(size_t)((nx-1)&(W-1))*G+g]=v;
  S.n0[g]=n0+v; S.n1[g]=n1+1; S.n2[g]=n2+2; S.pc[g]=pc+1;
}
__global__ __launch_bounds__(256) void k_aos(int G, Aos S, int W, int salt){
  int g=blockIdx.x*256+threadIdx.x; if(g>=G) return;
  int f=S.f[g]; I4 co=S.co[g]; I4 me=S.me[g]; I4 ns=S.ns[g];
  int pe=S.pr[(size_t)((co.z-1)&(W-1))*G+g];
  int v=f+co.x+co.y+me.x+me.y+me.z+pe+salt;
  S.pr[(size_t)((co.z-1)&(W-1))*G+g]=v;
  ns.x+=v; ns.y+=1; ns.z+=2; S.ns[g]=ns;
  co.w+=1; S.co[g]=co;
}
// the same with the three scalars of the bucket kernel's LDS replay in between is not modelled here.
int main(){
  const int G=1000000, W=8, IT=50;
  std::vector<int32_t*> cols; Soa S; int32_t** sp=(int32_t**)&S;
  for(int q=0;q<11;q++){ CK(hipMalloc(&sp[q],(size_t)G*4)); CK(hipMemset(sp[q],0,(size_t)G*4)); }
  CK(hipMalloc(&S.pr,(size_t)G*W*4)); CK(hipMemset(S.pr,0,(size_t)G*W*4));
  Aos A; CK(hipMalloc(&A.f,(size_t)G*4)); CK(hipMemset(A.f,0,(size_t)G*4));
  CK(hipMalloc(&A.co,(size_t)G*16)); CK(hipMemset(A.co,0,(size_t)G*16));
  CK(hipMalloc(&A.me,(size_t)G*16)); CK(hipMemset(A.me,0,(size_t)G*16));
  CK(hipMalloc(&A.ns,(size_t)G*16)); CK(hipMemset(A.ns,0,(size_t)G*16));
  CK(hipMalloc(&A.pr,(size_t)G*W*4)); CK(hipMemset(A.pr,0,(size_t)G*W*4));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid=(G+255)/256;
  for(int rep=0;rep<2;rep++){
    for(int i=0;i<3;i++) hipLaunchKernelGGL(k_soa,dim3(grid),dim3(256),0,0,G,S,W,i);
    CK(hipEventRecord(e0)); for(int i=0;i<IT;i++) hipLaunchKernelGGL(k_soa,dim3(grid),dim3(256),0,0,G,S,W,i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("soa   12 ld + 5 st : %7.2f us\n", ms*1000/IT);
    for(int i=0;i<3;i++) hipLaunchKernelGGL(k_aos,dim3(grid),dim3(256),0,0,G,A,W,i);
    CK(hipEventRecord(e0)); for(int i=0;i<IT;i++) hipLaunchKernelGGL(k_aos,dim3(grid),dim3(256),0,0,G,A,W,i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms,e0,e1)); printf("aos16  5 ld + 3 st : %7.2f us\n", ms*1000/IT);
  }
  return 0;
}
