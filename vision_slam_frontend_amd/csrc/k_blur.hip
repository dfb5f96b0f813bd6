// k_blur.hip -- K7: GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of every pyramid level (CV_8UC1), the
// image the rBRIEF tests sample.  ORB_Impl::detectAndCompute (features2d/orb.cpp) blurs each level in place
// before computeOrbDescriptors; reached from slam_frontend.cc:274.
//
// Restates the 8-bit fixed-point separable path of imgproc/smooth.cpp + filter.cpp: kernel
// round(256 * gauss) = [18 34 49 55 49 34 18] per axis (sum 257, not renormalised); row pass R = sum k_i p_i
// (fits 16 bit); column pass N = sum k_j R_j; result = N / 65536 rounded half-to-even on the columns OpenCV's
// SSE2 SymmColumnVec_32s8u handles ([0, w - w%4)) and half-up on the scalar tail, saturated to 255.
// One launch covers all levels and images through a tile table; a workgroup stages a (64+6) x (16+6) input
// tile in LDS, runs the row pass into a 16-bit LDS tile and the column pass from it.
#include "vsf_internal.h"

namespace {

constexpr int kTW = 64, kTH = 16, kR = 3;
constexpr int kInW = kTW + 2 * kR + 2;  // 72: padded row of the input tile

struct BlurArgs {
  const VsfLevel* levels;
  const uint32_t* tiles;  // level << 24 | ty << 12 | tx
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;
  uint8_t* blur;
  uint32_t pyr_bytes;
  int k0, k1, k2, k3;  // fixed-point kernel taps (k[3-i] == k[3+i])
};

__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - 2 - p;
  return p < 0 ? 0 : (p >= len ? len - 1 : p);  // clamp only reachable for len < 4 (one reflection suffices otherwise)
}

__global__ __launch_bounds__(256) void blur_tile_kernel(BlurArgs a) {
  __shared__ uint8_t in[(kTH + 2 * kR) * kInW];
  __shared__ uint16_t mid[(kTH + 2 * kR) * kTW];
  const int tid = threadIdx.x;
  const uint32_t td = a.tiles[blockIdx.x];
  const int level = (int)(td >> 24), ty = (int)((td >> 12) & 0xFFF), tx = (int)(td & 0xFFF);
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
  const int x0 = tx * kTW, y0 = ty * kTH;
  // input tile with reflected borders
  for (int i = tid; i < (kTH + 2 * kR) * (kTW + 2 * kR); i += 256) {
    const int r = i / (kTW + 2 * kR), c = i - r * (kTW + 2 * kR);
    const int sy = reflect101(y0 + r - kR, L.h), sx = reflect101(x0 + c - kR, L.w);
    in[r * kInW + c] = src[(size_t)sy * pitch + sx];
  }
  __syncthreads();
  // row pass
  for (int i = tid; i < (kTH + 2 * kR) * kTW; i += 256) {
    const int r = i >> 6, c = i & 63;
    const uint8_t* p = in + r * kInW + c;
    const int s = a.k0 * (p[0] + p[6]) + a.k1 * (p[1] + p[5]) + a.k2 * (p[2] + p[4]) + a.k3 * p[3];
    mid[i] = (uint16_t)s;
  }
  __syncthreads();
  // column pass: 4 rows x 64 columns per step
  uint8_t* dst = a.blur + (size_t)image * a.pyr_bytes + L.offset;
  for (int i = tid; i < kTH * kTW; i += 256) {
    const int r = i >> 6, c = i & 63;
    const int x = x0 + c, y = y0 + r;
    if (x < L.w && y < L.h) {
      const uint16_t* q = mid + r * kTW + c;
      const int n = a.k0 * ((int)q[0] + (int)q[6 * kTW]) + a.k1 * ((int)q[kTW] + (int)q[5 * kTW]) +
                    a.k2 * ((int)q[2 * kTW] + (int)q[4 * kTW]) + a.k3 * (int)q[3 * kTW];
      int v;
      if (x < L.blur_vec_end) {
        v = n >> 16;
        const int rem = n & 0xFFFF;
        v += (rem > 0x8000) | ((rem == 0x8000) & (v & 1));
      } else {
        v = (n + 0x8000) >> 16;
      }
      dst[(size_t)y * L.pitch + x] = (uint8_t)min(v, 255);
    }
  }
}

}  // namespace

void vsf_launch_blur(const VsfDev& d, const VsfGeom& g, const VsfImages& im, const uint32_t* d_tiles,
                           int ntiles, const int k[4], hipStream_t s) {
  BlurArgs a;
  a.levels = d.levels;
  a.tiles = d_tiles;
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
  hipLaunchKernelGGL(blur_tile_kernel, dim3(ntiles, im.n), dim3(256), 0, s, a);
}
