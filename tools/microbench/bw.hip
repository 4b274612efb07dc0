#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
// pattern: items of 4 components, each [ld] contiguous. wave handles one item per iteration.
template<typename T> __global__ void k_write(T* X, int nitems, size_t ldv) {   // ldv in units of T
  const int lane=threadIdx.x, wave=threadIdx.y; const size_t b=(size_t)blockIdx.y*64+lane;
  for (int it=blockIdx.x*16+wave; it<min(nitems,(int)(blockIdx.x*16+16)); it+=4) {
    T* q=X+(size_t)it*4*ldv+b; T v; 
    if constexpr (sizeof(T)==8) v=(T)it; else { v.x=it; v.y=it+1; }
    q[0]=v; q[ldv]=v; q[2*ldv]=v; q[3*ldv]=v; }
}
template<typename T> __global__ void k_copy(const T* A, T* X, int nitems, size_t ldv) {
  const int lane=threadIdx.x, wave=threadIdx.y; const size_t b=(size_t)blockIdx.y*64+lane;
  for (int it=blockIdx.x*16+wave; it<min(nitems,(int)(blockIdx.x*16+16)); it+=4) {
    const T* p=A+(size_t)it*4*ldv+b; T* q=X+(size_t)it*4*ldv+b;
    T a=p[0],b1=p[ldv],c=p[2*ldv],d=p[3*ldv]; q[0]=a; q[ldv]=b1; q[2*ldv]=c; q[3*ldv]=d; }
}
// gather-sum: each item sums T terms from pseudo-random other items (like LU terms), 3 blocks per term
template<typename T,int U> __global__ void k_gather(const T* A, T* X, const int* idx, int nitems, int terms, size_t ldv) {
  const int lane=threadIdx.x, wave=__builtin_amdgcn_readfirstlane(threadIdx.y); const size_t b=(size_t)blockIdx.y*64+lane;
  for (int it=blockIdx.x*16+wave; it<min(nitems,(int)(blockIdx.x*16+16)); it+=4) {
    T a0{},a1{},a2{},a3{};
    for (int t=0;t<terms;t+=U) {
      T v[U][4];
      #pragma unroll
      for(int u=0;u<U;u++){ const int j=__builtin_amdgcn_readfirstlane(idx[(size_t)it*terms+t+u]); const T* p=A+(size_t)j*4*ldv+b; v[u][0]=p[0];v[u][1]=p[ldv];v[u][2]=p[2*ldv];v[u][3]=p[3*ldv]; }
      #pragma unroll
      for(int u=0;u<U;u++){ a0+=v[u][0]; a1+=v[u][1]; a2+=v[u][2]; a3+=v[u][3]; }
    }
    T* q=X+(size_t)it*4*ldv+b; q[0]=a0;q[ldv]=a1;q[2*ldv]=a2;q[3*ldv]=a3; }
}
template<typename F> float timeit(F f,int reps){ hipEvent_t e0,e1; hipEventCreate(&e0);hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); for(int i=0;i<reps;i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); return ms/reps; }
int main(){
  const int nitems=34434, B=512; const size_t bytes=(size_t)nitems*4*B*8;
  double *A,*X; CK(hipMalloc(&A,bytes)); CK(hipMalloc(&X,bytes)); CK(hipMemset(A,0,bytes));
  const int terms=8; std::vector<int> idx((size_t)nitems*terms); unsigned s=1; for(auto& v:idx){ s=s*1664525u+1013904223u; v=(int)((s>>8)%nitems);} 
  // make gathers semi-local like LU (nearby items)
  for(size_t i=0;i<idx.size();i++){ int it=i/terms; idx[i]=std::max(0,std::min(nitems-1,it-(int)(idx[i]%2000))); }
  int* didx; CK(hipMalloc(&didx,idx.size()*4)); CK(hipMemcpy(didx,idx.data(),idx.size()*4,hipMemcpyHostToDevice));
  dim3 blk(64,4);
  { dim3 g((nitems+15)/16,B/64); float ms=timeit([&]{hipLaunchKernelGGL(k_write<double>,g,blk,0,0,X,nitems,(size_t)B);},20); printf("write  8B/lane: %.3f ms %.0f GB/s\n",ms,bytes/ms/1e6);}
  { dim3 g((nitems+15)/16,B/128); float ms=timeit([&]{hipLaunchKernelGGL(k_write<double2>,g,blk,0,0,(double2*)X,nitems,(size_t)B/2);},20); printf("write 16B/lane: %.3f ms %.0f GB/s\n",ms,bytes/ms/1e6);}
  { dim3 g((nitems+15)/16,B/64); float ms=timeit([&]{hipLaunchKernelGGL(k_copy<double>,g,blk,0,0,A,X,nitems,(size_t)B);},20); printf("copy   8B/lane: %.3f ms %.0f GB/s (r+w)\n",ms,2*bytes/ms/1e6);}
  { dim3 g((nitems+15)/16,B/128); float ms=timeit([&]{hipLaunchKernelGGL(k_copy<double2>,g,blk,0,0,(const double2*)A,(double2*)X,nitems,(size_t)B/2);},20); printf("copy  16B/lane: %.3f ms %.0f GB/s (r+w)\n",ms,2*bytes/ms/1e6);}
  { dim3 g((nitems+15)/16,B/64); float ms=timeit([&]{hipLaunchKernelGGL((k_gather<double,4>),g,blk,0,0,A,X,didx,nitems,terms,(size_t)B);},10); printf("gather  8B/lane U4: %.3f ms %.0f GB/s (reads)\n",ms,(double)bytes*terms/ms/1e6);}
  { dim3 g((nitems+15)/16,B/128); float ms=timeit([&]{hipLaunchKernelGGL((k_gather<double2,4>),g,blk,0,0,(const double2*)A,(double2*)X,didx,nitems,terms,(size_t)B/2);},10); printf("gather 16B/lane U4: %.3f ms %.0f GB/s (reads)\n",ms,(double)bytes*terms/ms/1e6);}
  { dim3 g((nitems+15)/16,B/128); float ms=timeit([&]{hipLaunchKernelGGL((k_gather<double2,2>),g,blk,0,0,(const double2*)A,(double2*)X,didx,nitems,terms,(size_t)B/2);},10); printf("gather 16B/lane U2: %.3f ms %.0f GB/s (reads)\n",ms,(double)bytes*terms/ms/1e6);}
  return 0; }
