// Experiment (round 2): are v_pk_minimum3_f16 / v_pk_maximum3_f16 (gfx950) exact on u8 values held as f16 denormal bit
// patterns, and what is their issue rate next to v_pk_min_i16?
//   hipcc --offload-arch=gfx950 -O3 -o m3 m3.hip && ./m3
// Result on MI355X: exhaustive 256^3 triples x both halves: 0 mismatches; 565 / 575 / 588 G wave-instructions/s for
// v_pk_min_i16 / v_pk_minimum3_f16 / v_pk_maximum3_f16 (the 614 G/s issue peak): full rate.  k_fast.hip relies on it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t pkmin3(uint32_t a, uint32_t b, uint32_t c){ uint32_t r; asm("v_pk_minimum3_f16 %0, %1, %2, %3":"=v"(r):"v"(a),"v"(b),"v"(c)); return r;}
__device__ __forceinline__ uint32_t pkmax3(uint32_t a, uint32_t b, uint32_t c){ uint32_t r; asm("v_pk_maximum3_f16 %0, %1, %2, %3":"=v"(r):"v"(a),"v"(b),"v"(c)); return r;}
__device__ __forceinline__ uint32_t pkmin2(uint32_t a, uint32_t b){ uint32_t r; asm("v_pk_min_i16 %0, %1, %2":"=v"(r):"v"(a),"v"(b)); return r;}
__global__ void k(uint32_t* bad){
  uint32_t a = blockIdx.x, b = threadIdx.x;
  for (uint32_t c = 0; c < 256; c++){
    uint32_t A = a | ((255-b)<<16), B = b | (c<<16), C = c | (a<<16);
    uint32_t mn = pkmin3(A,B,C), mx = pkmax3(A,B,C);
    uint32_t lo = min(a,min(b,c)), hi = min(255-b, min(c,a));
    uint32_t lo2 = max(a,max(b,c)), hi2 = max(255-b, max(c,a));
    if (mn != (lo | (hi<<16))) atomicAdd(&bad[0],1);
    if (mx != (lo2 | (hi2<<16))) atomicAdd(&bad[1],1);
  }
}
template<int MODE> __global__ void rate(uint32_t* out, int iters){
  uint32_t x[8];
  for (int i=0;i<8;i++) x[i] = threadIdx.x*2654435761u + i*977u;
  uint32_t y = threadIdx.x & 0x00FF00FF, z = (threadIdx.x*3) & 0x00FF00FF;
  for (int i=0;i<8;i++) x[i] &= 0x00FF00FF;
  for (int it=0; it<iters; it++){
#pragma unroll
    for (int i=0;i<8;i++){
      if (MODE==0) x[i] = pkmin2(x[i], y);
      else if (MODE==1) x[i] = pkmin3(x[i], y, z);
      else x[i] = pkmax3(x[i], y, z);
    }
  }
  uint32_t s=0; for (int i=0;i<8;i++) s^=x[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
int main(){
  uint32_t* d; (void)hipMalloc(&d,8); (void)hipMemset(d,0,8); k<<<256,256>>>(d); uint32_t h[2]; (void)hipMemcpy(h,d,8,hipMemcpyDeviceToHost);
  printf("exhaustive 256^3 x2 halves: bad min %u max %u\n",h[0],h[1]);
  uint32_t* o; (void)hipMalloc(&o, 4*256*1024*8);
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode=0; mode<3; mode++){
    float best=1e9;
    for (int rep=0; rep<3; rep++){
      (void)hipEventRecord(e0);
      if (mode==0) rate<0><<<256*8,1024>>>(o, 4096); else if (mode==1) rate<1><<<256*8,1024>>>(o,4096); else rate<2><<<256*8,1024>>>(o,4096);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1); if (ms<best) best=ms;
    }
    double waveinst = 256.0*8*16*4096*8; // waves * iters * 8
    printf("mode %d: %.3f ms, %.2f G wave-inst/s\n", mode, best, waveinst/best/1e6);
  }
  return h[0]||h[1];
}
