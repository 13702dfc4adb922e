// random-gather microbenchmark: how many independent random B-byte reads per second can MI355X serve?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x){ x ^= x>>33; x*=0xff51afd7ed558ccdull; x^=x>>33; x*=0xc4ceb9fe1a85ec53ull; x^=x>>33; return x;}
template<int BYTES, int ILP>
__global__ void gather(const uint4* __restrict__ tab, u64 nslots /* of BYTES */, u64 n_per_thread, u64* out){
  u64 tid = (u64)blockIdx.x*blockDim.x+threadIdx.x;
  u64 acc=0;
  for(u64 it=0; it<n_per_thread; it+=ILP){
    uint4 v[ILP][BYTES/16];
    #pragma unroll
    for(int j=0;j<ILP;++j){
      u64 h = mix(tid*0x9E3779B97F4A7C15ull + it + j);
      u64 s = __umul64hi(h, nslots);
      const uint4* p = tab + s*(BYTES/16);
      #pragma unroll
      for(int q=0;q<BYTES/16;++q) v[j][q]=p[q];
    }
    #pragma unroll
    for(int j=0;j<ILP;++j)
      #pragma unroll
      for(int q=0;q<BYTES/16;++q) acc += v[j][q].x ^ v[j][q].w;
  }
  if(acc==0x1234567) out[0]=acc;
}
template<int BYTES,int ILP> void run(const uint4* tab, u64 bytes, u64* out, const char* name){
  u64 nslots = bytes/BYTES; int blocks=256*8, threads=256; u64 npt=256;
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  gather<BYTES,ILP><<<blocks,threads>>>(tab,nslots,npt,out); hipDeviceSynchronize();
  hipEventRecord(a); for(int r=0;r<3;++r) gather<BYTES,ILP><<<blocks,threads>>>(tab,nslots,npt,out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); ms/=3; double n=(double)blocks*threads*npt;
  printf("%-28s table %.1f GiB: %.2f G loads/s, useful %.1f GB/s\n",name,bytes/1073741824.0,n/ms/1e6,n*BYTES/ms/1e6);
}
int main(){
  for (u64 gib : {1ull, 16ull, 64ull}) {
    u64 bytes=gib<<30; uint4* tab; if(hipMalloc(&tab,bytes)!=hipSuccess){printf("alloc fail\n");return 1;} hipMemset(tab,1,bytes); u64* out; hipMalloc(&out,8);
    run<16,1>(tab,bytes,out,"16B ilp1"); run<16,4>(tab,bytes,out,"16B ilp4");
    run<32,1>(tab,bytes,out,"32B ilp1"); run<32,4>(tab,bytes,out,"32B ilp4");
    run<64,1>(tab,bytes,out,"64B ilp1"); run<64,2>(tab,bytes,out,"64B ilp2");
    run<128,1>(tab,bytes,out,"128B ilp1");
    hipFree(tab); hipFree(out);
  }
  return 0; }
