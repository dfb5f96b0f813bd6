// k_pyramid.hip -- K1: ORB scale pyramid, level l = resize(level l-1, INTER_LINEAR), CV_8UC1.
//
// Restates cv::resize's 8-bit bilinear path (imgproc/imgwarp.cpp: HResizeLinear<uchar,int,short,2048> +
// VResizeLinear<..., FixedPtCast<int,uchar,22>>) as reached from ORB_Impl::detectAndCompute
// (features2d/orb.cpp), i.e. from slam_frontend.cc:274.  The coefficient tables (xofs/ialpha, yofs/ibeta) are
// built once per context on the host exactly as cv::resize builds them, so the kernel is integer-only.
// One thread produces four horizontally adjacent output pixels (one 32-bit store).
#include "vsf_internal.h"

namespace {

__global__ __launch_bounds__(256) void resize_level_kernel(const uint8_t* __restrict__ src_base, size_t src_img_stride,
                                                           int src_pitch, uint8_t* __restrict__ dst_base,
                                                           size_t dst_img_stride, int dst_pitch, int dw, int dh,
                                                           const VsfTap* __restrict__ xt,
                                                           const VsfTap* __restrict__ yt) {
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x4 >= dw || y >= dh) return;
  const uint8_t* S = src_base + (size_t)blockIdx.z * src_img_stride;
  const VsfTap ty = yt[y];
  const uint8_t* S0 = S + (size_t)ty.i0 * src_pitch;
  const uint8_t* S1 = S + (size_t)ty.i1 * src_pitch;
  const int b0 = ty.c0, b1 = ty.c1;
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int x = x4 + k;
    if (x < dw) {
      const VsfTap tx = xt[x];
      const int r0 = S0[tx.i0] * tx.c0 + S0[tx.i1] * tx.c1;
      const int r1 = S1[tx.i0] * tx.c0 + S1[tx.i1] * tx.c1;
      const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 255) << (8 * k);
    }
  }
  uint8_t* D = dst_base + (size_t)blockIdx.z * dst_img_stride + (size_t)y * dst_pitch + x4;
  *reinterpret_cast<uint32_t*>(D) = out;
}

}  // namespace

void vsf_launch_pyramid(const VsfDev& d, const VsfGeom& g, const VsfLevel* h_levels, const VsfImages& im,
                        hipStream_t s) {
  for (int l = 1; l < g.nlevels; l++) {
    const VsfLevel& L = h_levels[l];
    const VsfLevel& P = h_levels[l - 1];
    const uint8_t* src = (l == 1) ? im.base : d.pyr + P.offset;
    const size_t src_img_stride = (l == 1) ? im.image_stride : (size_t)g.pyr_bytes;
    const int src_pitch = (l == 1) ? (int)im.row_stride : P.pitch;
    dim3 block(64, 4, 1);
    dim3 grid((L.w + 255) / 256, (L.h + 3) / 4, im.n);
    hipLaunchKernelGGL(resize_level_kernel, grid, block, 0, s, src, src_img_stride, src_pitch, d.pyr + L.offset,
                       (size_t)g.pyr_bytes, L.pitch, L.w, L.h, d.xtaps + L.xtab, d.ytaps + L.ytab);
  }
}
