#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ float fast_rcp(float x){ float r=__builtin_amdgcn_rcpf(x); float e=__builtin_fmaf(-x,r,1.0f); r=__builtin_fmaf(e,r,r); e=__builtin_fmaf(-x,r,1.0f); r=__builtin_fmaf(e,r,r); return r; }
__device__ __forceinline__ float fast_rcp1(float x){ float r=__builtin_amdgcn_rcpf(x); float e=__builtin_fmaf(-x,r,1.0f); r=__builtin_fmaf(e,r,r); return r; }
__global__ void chk(int expo, unsigned long long* bad, unsigned long long* bad1){
  unsigned m = blockIdx.x*blockDim.x+threadIdx.x; if(m>=(1u<<23)) return;
  unsigned u = ((unsigned)(127+expo)<<23)|m; float x=__uint_as_float(u);
  float a=__fdiv_rn(1.0f,x), b=fast_rcp(x), c=fast_rcp1(x);
  if(__float_as_uint(a)!=__float_as_uint(b)) atomicAdd(bad,1ull);
  if(__float_as_uint(a)!=__float_as_uint(c)) atomicAdd(bad1,1ull);
}
int main(){ unsigned long long *d; hipMalloc(&d,16); int es[]={0,1,2,10,33,60,100,120,125,126,-1,-10,-100,-126};
 for(int e: es){ hipMemset(d,0,16); chk<<<(1<<23)/256,256>>>(e,d,d+1); unsigned long long h[2]; hipMemcpy(h,d,16,hipMemcpyDeviceToHost); printf("expo %d: mismatches 2NR=%llu 1NR=%llu\n",e,h[0],h[1]); } }
