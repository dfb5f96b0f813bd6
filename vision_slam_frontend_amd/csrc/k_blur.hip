// k_blur.hip -- K7: GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of every pyramid level (CV_8UC1), the
// image the rBRIEF tests sample.  ORB_Impl::detectAndCompute (features2d/orb.cpp) blurs each level in place
// before computeOrbDescriptors; reached from slam_frontend.cc:274.
//
// Restates the 8-bit fixed-point separable path of imgproc/smooth.cpp + filter.cpp: kernel
// round(256 * gauss) = [18 34 49 55 49 34 18] per axis (sum 257, not renormalised); row pass R = sum k_i p_i
// (fits 16 bit); column pass N = sum k_j R_j; result = N / 65536 rounded half-to-even on the columns OpenCV's
// SSE2 SymmColumnVec_32s8u handles ([0, w - w%4)) and half-up on the scalar tail, saturated to 255.
//
// Streaming "march" kernel, no LDS and no barriers: a wave owns a band of 248 columns (62 lanes x 4 pixels; lanes
// 0 and 63 only carry the 4-pixel halo) and walks down a strip of rows.  Per row each lane issues ONE coalesced
// 32-bit load, gets its neighbours' dwords with two DPP wave shifts, does the horizontal pass for its 4 pixels and
// pushes the 4 sums into a 7-row register window; the vertical pass reads that window and the lane stores one
// 32-bit word.  HBM traffic is the compulsory P read + P write (plus 6 halo rows per 64-row strip).
#include "vsf_internal.h"

namespace {

constexpr int kBandCols = VSF_BLUR_BAND_COLS;   // output columns per wave
constexpr int kStripRows = VSF_BLUR_STRIP_ROWS; // output rows per wave

struct BlurArgs {
  const VsfLevel* levels;
  const uint32_t* units;  // level << 24 | band << 16 | strip
  int nunits;
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;
  uint8_t* blur;
  uint32_t pyr_bytes;
  int k0, k1, k2, k3;  // fixed-point kernel taps (k[3-i] == k[3+i])
};

struct R4 {
  int a, b, c, d;
};

__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - 2 - p;
  return p < 0 ? 0 : (p >= len ? len - 1 : p);  // clamp only reachable for len < 4 (one reflection suffices otherwise)
}

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {  // lane i <- lane i-1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) {  // lane i <- lane i+1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false);
}

// Loads the lane's 4 pixels of one row (columns c0..c0+3), reflecting columns outside [0, w).
__device__ __forceinline__ uint32_t load_row_dword(const uint8_t* __restrict__ rowp, int c0, int w, bool interior) {
  if (interior) return *reinterpret_cast<const uint32_t*>(rowp + c0);
  uint32_t v = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) v |= (uint32_t)rowp[reflect101(c0 + j, w)] << (8 * j);
  return v;
}

__device__ __forceinline__ R4 row_pass(uint32_t dp, uint32_t d, uint32_t dn, int k0, int k1, int k2, int k3) {
  const int b1 = (dp >> 8) & 255, b2 = (dp >> 16) & 255, b3 = dp >> 24;
  const int b4 = d & 255, b5 = (d >> 8) & 255, b6 = (d >> 16) & 255, b7 = d >> 24;
  const int b8 = dn & 255, b9 = (dn >> 8) & 255, b10 = (dn >> 16) & 255;
  R4 r;
  r.a = k0 * (b1 + b7) + k1 * (b2 + b6) + k2 * (b3 + b5) + k3 * b4;
  r.b = k0 * (b2 + b8) + k1 * (b3 + b7) + k2 * (b4 + b6) + k3 * b5;
  r.c = k0 * (b3 + b9) + k1 * (b4 + b8) + k2 * (b5 + b7) + k3 * b6;
  r.d = k0 * (b4 + b10) + k1 * (b5 + b9) + k2 * (b6 + b8) + k3 * b7;
  return r;
}

__device__ __forceinline__ int round_px(int n, bool half_even) {
  int v;
  if (half_even) {
    v = n >> 16;
    const int rem = n & 0xFFFF;
    v += (rem > 0x8000) | ((rem == 0x8000) & (v & 1));
  } else {
    v = (n + 0x8000) >> 16;
  }
  return min(v, 255);
}

#define VSF_COL(f) (k0 * (w0.f + w6.f) + k1 * (w1.f + w5.f) + k2 * (w2.f + w4.f) + k3 * w3.f)

__global__ __launch_bounds__(256) void blur_march_kernel(BlurArgs a) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= a.nunits) return;  // wave-uniform
  const uint32_t ud = a.units[unit];
  const int level = (int)(ud >> 24), band = (int)((ud >> 16) & 0xFF), strip = (int)(ud & 0xFFFF);
  const VsfLevel L = a.levels[level];
  const int image = blockIdx.y;
  const uint8_t* src;
  int pitch;
  if (level == 0) {
    src = a.img0 + (size_t)image * a.img0_stride;
    pitch = a.img0_pitch;
  } else {
    src = a.pyr + (size_t)image * a.pyr_bytes + L.offset;
    pitch = L.pitch;
  }
  uint8_t* dst = a.blur + (size_t)image * a.pyr_bytes + L.offset;
  const int k0 = a.k0, k1 = a.k1, k2 = a.k2, k3 = a.k3;
  const int w = L.w, h = L.h;
  const int c0 = band * kBandCols - 4 + 4 * lane;  // first column of this lane's dword
  const bool interior = c0 >= 0 && c0 + 3 < w;
  const bool writer = lane >= 1 && lane <= 62 && c0 < w;
  const int ys = strip * kStripRows, ye = min(ys + kStripRows, h);
  const bool he0 = c0 + 0 < L.blur_vec_end, he1 = c0 + 1 < L.blur_vec_end, he2 = c0 + 2 < L.blur_vec_end,
             he3 = c0 + 3 < L.blur_vec_end;

  R4 w0, w1, w2, w3, w4, w5, w6;
  auto fetch = [&](int y) -> R4 {
    const uint8_t* rowp = src + (size_t)reflect101(y, h) * pitch;
    const uint32_t d = load_row_dword(rowp, c0, w, interior);
    return row_pass(wave_shr1(d), d, wave_shl1(d), k0, k1, k2, k3);
  };
  w1 = fetch(ys - 3);
  w2 = fetch(ys - 2);
  w3 = fetch(ys - 1);
  w4 = fetch(ys);
  w5 = fetch(ys + 1);
  w6 = fetch(ys + 2);
  for (int y = ys; y < ye; y++) {
    w0 = w1;
    w1 = w2;
    w2 = w3;
    w3 = w4;
    w4 = w5;
    w5 = w6;
    w6 = fetch(y + 3);
    if (writer) {
      const uint32_t o = (uint32_t)round_px(VSF_COL(a), he0) | ((uint32_t)round_px(VSF_COL(b), he1) << 8) |
                         ((uint32_t)round_px(VSF_COL(c), he2) << 16) | ((uint32_t)round_px(VSF_COL(d), he3) << 24);
      *reinterpret_cast<uint32_t*>(dst + (size_t)y * L.pitch + c0) = o;
    }
  }
}

}  // namespace

void vsf_launch_blur(const VsfDev& d, const VsfGeom& g, const VsfImages& im, const uint32_t* d_units, int nunits,
                     const int k[4], hipStream_t s) {
  BlurArgs a;
  a.levels = d.levels;
  a.units = d_units;
  a.nunits = nunits;
  a.img0 = im.base;
  a.img0_stride = im.image_stride;
  a.img0_pitch = (int)im.row_stride;
  a.pyr = d.pyr;
  a.blur = d.blur;
  a.pyr_bytes = g.pyr_bytes;
  a.k0 = k[0];
  a.k1 = k[1];
  a.k2 = k[2];
  a.k3 = k[3];
  hipLaunchKernelGGL(blur_march_kernel, dim3((nunits + 3) / 4, im.n), dim3(256), 0, s, a);
}
