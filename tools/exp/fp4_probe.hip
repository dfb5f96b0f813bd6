// fp4_probe.hip -- can the Hamming matcher's key  8192 d - 2^20 + row  come out of v_mfma_scale_f32_32x32x64_f8f6f4 with
// FP4 (E2M1) operands?  Bits become +-4 (codes 0x6 / 0xE), both block scales are 2^4 (E8M0 131), so a product is +-4096 as
// with the int8 +-64 operands; 256 bits are FOUR instructions of K = 64 (int8: eight of K = 32, at the same 32 cycles
// each), and the row index rides in the C operand.  Checks every key of 64 random 32 x 32 tiles against popcounts and times
// both instruction streams.     hipcc --offload-arch=gfx950 -O3 -o fp4_probe fp4_probe.hip && ./fp4_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 8 descriptor bits -> 8 FP4 codes in a dword (bit i -> nibble i): `set` / `clear` are the 4-bit codes
__device__ inline uint32_t expand8(uint32_t b, bool set_is_plus) {
  uint32_t x = b & 0xFFu;
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  // +4 = 0x6, -4 = 0xE: the sign is bit 3 of the nibble
  return set_is_plus ? (0xEEEEEEEEu ^ (x << 3)) : (0x66666666u | (x << 3));
}

__device__ inline v8i expand32(uint32_t bits, bool set_is_plus) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < 4; k++) r[k] = (int)expand8(bits >> (8 * k), set_is_plus);
  return r;
}

// one wave: keys[train row][query] of a 32 x 32 tile (descriptors: 8 dwords each)
__global__ void tile_keys(const uint32_t* __restrict__ train, const uint32_t* __restrict__ query, float* __restrict__ keys) {
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  v16f acc;
  for (int i = 0; i < 16; i++) acc[i] = (float)((i & 3) + 8 * (i >> 2) + 4 * h);  // the row index, through C
  const int scale = 131;  // E8M0: 2^(131 - 127) = 16
  for (int s = 0; s < 4; s++) {
    const v8i a = expand32(train[c * 8 + 2 * s + h], true);    // lane (c, h): bits [32 h, 32 h + 32) of the step's 64
    const v8i b = expand32(query[c * 8 + 2 * s + h], false);
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, scale, 0, scale);
  }
  for (int i = 0; i < 16; i++) keys[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + c] = acc[i];
}

template <bool FP4>
__global__ void issue_rate(uint64_t* cycles, float* sink, int iters) {
  v16f accf = {0};
  v16i acci = {0};
  v8i a8 = {0x66666666, 0x6E6E6E6E, 0x66EE66EE, 0x6666EEEE, 0, 0, 0, 0}, b8 = a8;
  v4i a4 = {0x40404040, 0x40C040C0, 0x4040C0C0, 0x40404040}, b4 = a4;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (FP4) {
      for (int s = 0; s < 4; s++) accf = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, accf, 4, 4, 0, 127, 0, 127);
    } else {
      for (int s = 0; s < 8; s++) acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, acci, 0, 0, 0);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = FP4 ? accf[0] : (float)acci[0];
}

int main() {
  std::mt19937 rng(7);
  const int T = 64;
  std::vector<uint32_t> tr(T * 32 * 8), qu(T * 32 * 8);
  for (auto& v : tr) v = rng();
  for (auto& v : qu) v = rng();
  for (int i = 0; i < 8; i++) qu[i] = tr[i], qu[8 + i] = ~tr[8 + i];  // distance 0 and 256
  uint32_t *dt, *dq;
  float* dk;
  hipMalloc(&dt, tr.size() * 4);
  hipMalloc(&dq, qu.size() * 4);
  hipMalloc(&dk, 32 * 32 * 4);
  hipMemcpy(dt, tr.data(), tr.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dq, qu.data(), qu.size() * 4, hipMemcpyHostToDevice);
  long bad = 0, checked = 0;
  std::vector<float> keys(32 * 32);
  for (int t = 0; t < T; t++) {
    hipLaunchKernelGGL(tile_keys, dim3(1), dim3(64), 0, 0, dt + t * 256, dq + t * 256, dk);
    hipMemcpy(keys.data(), dk, keys.size() * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < 32; r++)
      for (int c = 0; c < 32; c++) {
        int d = 0;
        for (int w = 0; w < 8; w++) d += __builtin_popcount(tr[t * 256 + r * 8 + w] ^ qu[t * 256 + c * 8 + w]);
        const float want = 8192.0f * d - 1048576.0f + r;
        checked++;
        if (keys[r * 32 + c] != want) {
          if (bad++ < 5) std::printf("tile %d row %d col %d: d %d, key %.1f, want %.1f\n", t, r, c, d, keys[r * 32 + c], want);
        }
      }
  }
  std::printf("keys: %ld checked, %ld wrong\n", checked, bad);
  uint64_t* dc;
  float* ds;
  hipMalloc(&dc, 8 * 1024);
  hipMalloc(&ds, 1024 * 64 * 4);
  for (int fp4 = 0; fp4 < 2; fp4++) {
    const int iters = 20000;
    for (int rep = 0; rep < 2; rep++) {
      if (fp4) hipLaunchKernelGGL(issue_rate<true>, dim3(1024), dim3(64), 0, 0, dc, ds, iters);
      else hipLaunchKernelGGL(issue_rate<false>, dim3(1024), dim3(64), 0, 0, dc, ds, iters);
      hipDeviceSynchronize();
    }
    uint64_t c0 = 0;
    hipMemcpy(&c0, dc, 8, hipMemcpyDeviceToHost);
    std::printf("%s: %.1f cycles (s_memtime units, 100 MHz * ...) per 256-bit 32x32 tile, %d MFMAs per tile\n", fp4 ? "fp4 32x32x64" : "i8 32x32x32",
                (double)c0 / iters, fp4 ? 4 : 8);
  }
  // wall-clock rate: one wave per SIMD on the whole chip
  for (int fp4 = 0; fp4 < 2; fp4++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 100000;
    hipEventRecord(e0, 0);
    if (fp4) hipLaunchKernelGGL(issue_rate<true>, dim3(1024), dim3(64), 0, 0, dc, ds, iters);
    else hipLaunchKernelGGL(issue_rate<false>, dim3(1024), dim3(64), 0, 0, dc, ds, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("%s: %.3f ms for %d tiles per wave on 1024 waves -> %.2f G tile-waves/s, %.1f ns per tile\n", fp4 ? "fp4" : "i8 ", ms, iters,
                1024.0 * iters / ms / 1e6, ms * 1e6 / iters);
  }
  return bad != 0;
}
