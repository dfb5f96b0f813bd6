// Experiment (round 3): can the 7x7 Gaussian blur's column pass run on the f16 matrix cores with EXACT integer results?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
// Questions:
//  (1) v_mfma_f32_32x32x16_f16 with A = bytes zero-extended to 16 bit (f16 DENORMALS, value v * 2^-24): flushed or honoured?
//      exact f32 accumulation of sum k_i * v_i (taps up to 55*256, |sums| < 2^24 * 2^-24 around a -0.5 bias)?
//  (2) v_cvt_pk_u8_f32: rounding rule (nearest-even?) and saturation.
//  (3) which A byte slot of v_mfma_i32_32x32x32_i8 multiplies with which B byte slot.
//  (4) issue rates (G wave-instructions/s) of the VALU instructions the streaming kernels are made of.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i4v __attribute__((ext_vector_type(4)));

// (1) D[m][n] = sum_k A[m][k] * B[k][n] - 0.5 with A raw u16 patterns (denormals), B f16 taps
__global__ void k_f16(const uint16_t* A, const _Float16* B, float* D) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  s8 a; h8 b;
  for (int j = 0; j < 8; j++) { a[j] = (short)A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
  f16v c; for (int i = 0; i < 16; i++) c[i] = -0.5f;
  f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), b, c, 0, 0, 0);
  for (int i = 0; i < 16; i++) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = d[i];
}
// (2)
__global__ void k_cvt(const float* x, uint32_t* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t r; asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(x[i]), "v"(0xAABBCCDDu)); y[i] = r; }
}
// (3) one-hot A slot (h0,b0) in row 0; B slot (h,b) of every column holds 1 + 16 h + b
__global__ void k_i8slot(int* out) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  for (int h0 = 0; h0 < 2; h0++) for (int b0 = 0; b0 < 16; b0++) {
    uint8_t a[16] = {0}, b[16];
    if (r == 0 && h == h0) a[b0] = 1;
    for (int j = 0; j < 16; j++) b[j] = (uint8_t)(1 + 16 * h + j);
    i4v av, bv; __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
    i16v c = {0};
    i16v d = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    if (l == 0) out[h0 * 16 + b0] = d[0];  // D[0][0]
  }
}
// (4) issue rates
#define RATE_KERNEL(NAME, ASM)                                                            \
  __global__ __launch_bounds__(1024) void NAME(uint32_t* out, int iters) {                \
    uint32_t x[8], y = threadIdx.x * 2654435761u | 1u, z = threadIdx.x * 40503u + 7u;     \
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 977u + i;                            \
    for (int it = 0; it < iters; it++) {                                                  \
      _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(x[i]) : "v"(y), "v"(z)); \
    }                                                                                     \
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= x[i];                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                       \
  }
RATE_KERNEL(r_add_u32, "v_add_u32 %0, %0, %1")
RATE_KERNEL(r_and_b32, "v_and_b32 %0, %0, %1")
RATE_KERNEL(r_xor_b32, "v_xor_b32 %0, %0, %1")
RATE_KERNEL(r_min_u32, "v_min_u32 %0, %0, %1")
RATE_KERNEL(r_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
RATE_KERNEL(r_add3, "v_add3_u32 %0, %0, %1, %2")
RATE_KERNEL(r_perm, "v_perm_b32 %0, %0, %1, %2")
RATE_KERNEL(r_bfe, "v_bfe_u32 %0, %0, 3, 8")
RATE_KERNEL(r_alignbit, "v_alignbit_b32 %0, %0, %1, 8")
RATE_KERNEL(r_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
RATE_KERNEL(r_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
RATE_KERNEL(r_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
RATE_KERNEL(r_dot2_u32_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
RATE_KERNEL(r_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
RATE_KERNEL(r_sad_u8, "v_sad_u8 %0, %1, %2, %0")
RATE_KERNEL(r_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
RATE_KERNEL(r_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
RATE_KERNEL(r_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
RATE_KERNEL(r_fma_f32, "v_fma_f32 %0, %0, %1, %2")
RATE_KERNEL(r_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
RATE_KERNEL(r_cvt_f32_ubyte, "v_cvt_f32_ubyte0 %0, %0")
RATE_KERNEL(r_mov_dpp, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
RATE_KERNEL(r_med3_u32, "v_med3_u32 %0, %0, %1, %2")
RATE_KERNEL(r_max3_u32, "v_max3_u32 %0, %0, %1, %2")
RATE_KERNEL(r_pk_max3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
RATE_KERNEL(r_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
RATE_KERNEL(r_lshlrev, "v_lshlrev_b32 %0, 1, %0")
RATE_KERNEL(r_pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 1, %0")
RATE_KERNEL(r_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
RATE_KERNEL(r_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
RATE_KERNEL(r_pk_add_f32x, "v_cvt_pk_bf16_f32 %0, %0, %1")
RATE_KERNEL(r_permlane32_swap, "v_permlane32_swap_b32 %0, %1")

typedef void (*rate_fn)(uint32_t*, int);
struct RateEnt { const char* name; rate_fn fn; };

int main() {
  // (1)
  {
    std::vector<uint16_t> A(32 * 16); std::vector<_Float16> B(16 * 32); std::vector<float> D(1024);
    int bad = 0, total = 0; double maxerr = 0;
    uint16_t* dA; _Float16* dB; float* dD;
    (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dD, 4096);
    const int taps[7] = {18, 34, 49, 55, 49, 34, 18};
    for (int trial = 0; trial < 200; trial++) {
      srand(trial);
      for (auto& v : A) v = (trial == 0) ? 255 : (trial == 1 ? 0 : (uint16_t)(rand() & 255));
      for (int k = 0; k < 16; k++) for (int n = 0; n < 32; n++) {
        int t = taps[(k + n + trial) % 7] * ((k & 1) ? 256 : 1);   // mix of k and 256 k taps, dense (worst-case sums)
        if (trial >= 100) t = ((k + n) % 3 == 0) ? 0 : t;
        B[k * 32 + n] = (_Float16)(float)t;
      }
      (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
      (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
      k_f16<<<1, 64>>>(dA, dB, dD);
      (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
      for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) {
        long long s = 0;
        for (int k = 0; k < 16; k++) s += (long long)A[m * 16 + k] * (long long)(float)B[k * 32 + n];
        double want = (double)s / 16777216.0 - 0.5;
        double err = fabs((double)D[m * 32 + n] - want);
        if (err > maxerr) maxerr = err;
        if (err != 0.0) bad++;
        total++;
      }
    }
    printf("(1) f16 denormal-input MFMA: %d / %d inexact, max abs err %.3g (trial 0 sum = 255*sum taps)\n", bad, total, maxerr);
  }
  // (2)
  {
    std::vector<float> x; 
    for (int k = -2; k < 260; k++) for (int f = -2; f <= 2; f++) x.push_back((float)k + 0.5f + f * (1.0f / 65536.0f));
    for (int k = 0; k < 256; k++) { x.push_back((float)k); x.push_back((float)k + 0.25f); x.push_back((float)k + 0.75f); }
    x.push_back(-100.f); x.push_back(1e9f); x.push_back(-0.5f); x.push_back(255.5f); x.push_back(255.49f);
    float* dx; uint32_t* dy; (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dy, x.size() * 4);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    k_cvt<<<(x.size() + 255) / 256, 256>>>(dx, dy, (int)x.size());
    std::vector<uint32_t> y(x.size()); (void)hipMemcpy(y.data(), dy, x.size() * 4, hipMemcpyDeviceToHost);
    int bad_rne = 0, bad_pack = 0, bad_trunc = 0, bad_halfup = 0;
    for (size_t i = 0; i < x.size(); i++) {
      double v = x[i]; 
      double rne = nearbyint(v); if (rne < 0) rne = 0; if (rne > 255) rne = 255;
      double tr = floor(v); if (tr < 0) tr = 0; if (tr > 255) tr = 255;
      double hu = floor(v + 0.5); if (hu < 0) hu = 0; if (hu > 255) hu = 255;
      uint32_t got = (y[i] >> 8) & 255;
      if ((y[i] & 0xFFFF00FFu) != 0xAABB00DDu) bad_pack++;
      if (got != (uint32_t)rne) bad_rne++;
      if (got != (uint32_t)tr) bad_trunc++;
      if (got != (uint32_t)hu) bad_halfup++;
    }
    printf("(2) v_cvt_pk_u8_f32 over %zu values: mismatches vs nearest-even %d, vs truncation %d, vs half-up %d; byte insertion wrong %d\n",
           x.size(), bad_rne, bad_trunc, bad_halfup, bad_pack);
    for (float t : {0.5f, 1.5f, 2.5f, 254.5f, 255.5f, -0.5f, 300.f, 2.4999f, 2.50002f}) {
      for (size_t i = 0; i < x.size(); i++) if (x[i] == t) { printf("    cvt(%g) = %u\n", t, (y[i] >> 8) & 255); break; }
    }
  }
  // (3)
  {
    int* d; (void)hipMalloc(&d, 32 * 4); k_i8slot<<<1, 64>>>(d); int h[32]; (void)hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    int same = 1; for (int i = 0; i < 32; i++) if (h[i] != 1 + i) same = 0;
    printf("(3) i8 32x32x32: A slot (h,b) multiplies B slot (h,b): %s  [", same ? "yes" : "NO");
    for (int i = 0; i < 32; i++) printf("%d ", h[i]); printf("]\n");
  }
  // (4)
  {
    RateEnt ents[] = {
#define E(n) {#n, n}
      E(r_add_u32), E(r_and_b32), E(r_xor_b32), E(r_min_u32), E(r_lshl_add), E(r_add3), E(r_perm), E(r_bfe), E(r_alignbit),
      E(r_pk_add_u16), E(r_pk_mad_u16), E(r_pk_min_u16), E(r_dot2_u32_u16), E(r_dot4_u32_u8), E(r_sad_u8), E(r_mad_u32_u24),
      E(r_mul_hi_u32_u24), E(r_mul_lo_u32), E(r_fma_f32), E(r_cvt_pk_u8), E(r_cvt_f32_ubyte), E(r_mov_dpp), E(r_med3_u32),
      E(r_max3_u32), E(r_pk_max3_f16), E(r_cndmask), E(r_lshlrev), E(r_pk_lshrrev_b16), E(r_pk_mul_lo_u16), E(r_pk_fma_f16),
      E(r_pk_add_f32x), E(r_permlane32_swap)};
    uint32_t* o; (void)hipMalloc(&o, 4 * 256 * 8 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("(4) issue rate, G wave-instructions/s (chip: 1024 SIMDs x 2.4 GHz = 2458 G SIMD-cycles/s); waves per SIMD = 1, 2, 4\n");
    for (auto& e : ents) {
      printf("    %-22s", e.name);
      for (int threads : {256, 512, 1024}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
          (void)hipEventRecord(e0);
          hipLaunchKernelGGL(e.fn, dim3(256 * 4), dim3(threads), 0, 0, o, 2048);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double wi = 256.0 * 4 * (threads / 64) * 2048.0 * 8;
        printf("  %8.1f", wi / best / 1e6);
      }
      printf("\n");
    }
  }
  return 0;
}
