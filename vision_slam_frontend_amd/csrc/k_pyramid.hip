// k_pyramid.hip -- K1: ORB scale pyramid, level l = resize(level l-1, INTER_LINEAR), CV_8UC1.
//
// Restates cv::resize's 8-bit bilinear path (imgproc/imgwarp.cpp: HResizeLinear<uchar,int,short,2048> +
// VResizeLinear<..., FixedPtCast<int,uchar,22>>) as reached from ORB_Impl::detectAndCompute
// (features2d/orb.cpp), i.e. from slam_frontend.cc:274.  The coefficient tables (xofs/ialpha, yofs/ibeta) are
// built once per context on the host exactly as cv::resize builds them, so the kernel is integer-only.
//
// resize_strip_kernel<R> (no LDS, no barriers): a wave owns 256 output columns (lane = 4 adjacent pixels) x a strip of
// R = 8 or 16 output rows.  The lane's four x-taps are loop-invariant: four v_perm_b32 byte selectors and four packed
// weight pairs, evaluated IN the kernel with cv::resize's own double / float steps (no table load ahead of the row
// loads); the y taps are evaluated once per wave (lane r <-> row r, v_readlane -> scalar addresses and weights).  At
// scale 1.04 the R output rows touch at most R + 2 consecutive source rows (checked per level on the host,
// VsfLevel::resize_rows), so every source row goes through the horizontal pass ONCE (one 8-byte load + 4 x (v_perm +
// v_dot2_u32_u16)) and an output row picks its two with a wave-uniform branch; the vertical pass is v_mul_hi_u32_u24 on
// pre-shifted weights.  One launch per level (the level chain is a true dependency), issued as two chains of half
// batches on two streams.  pyramid_image_kernel: when the batch fills the chip, the 26 one-band levels (w <= 256) are
// built by ONE launch, a 1024-thread workgroup per image with the levels ping-ponged through LDS.
// pyramid_slab_kernel: a batch of one to four images (a frame) walks chains of levels per launch, the last level of a
// chain cut into slabs whose workgroups never wait for each other (border rows are computed twice).
// resize_march_kernel is the general fallback for levels that fail the R + 2 check.
#pragma clang fp contract(off)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "vsf_internal.h"

namespace {


typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(1))) U8B {
  uint32_t lo, hi;
};

struct ResizeArgs {
  const uint8_t* src;
  size_t src_img_stride;
  int src_pitch, sw, sh;
  uint8_t* dst;
  size_t dst_img_stride;
  int dst_pitch, dw, dh;
  double scale_x, scale_y;  // cv::resize: 1. / ((double)dw / sw), 1. / ((double)dh / sh)
  int nstrips;
};

template <int kStripRows>
__global__ __launch_bounds__(256) void resize_march_kernel(ResizeArgs a) {
  const int lane = threadIdx.x & 63;
  const int strip = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), band = blockIdx.z;
  if (strip >= a.nstrips) return;  // wave-uniform
  const int x4 = band * 256 + lane * 4;
  const bool active = x4 < a.dw;
  // cv::resize's coefficient tables (xofs / ialpha, yofs / ibeta) evaluated in place with the same double / float
  // steps as the host (vsf_api.hip build_taps; no FMA contraction in this file): no table load sits in front of the
  // source-row loads, the wave's dependency chain is  rows -> arithmetic -> store.
  auto xtap = [&](int dx) -> VsfTap {
    float fx = (float)((dx + 0.5) * a.scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= a.sw - 1) fx = 0, sx = a.sw - 1;
    VsfTap t;
    t.i0 = (uint16_t)sx;
    t.i1 = (uint16_t)min(sx + 1, a.sw - 1);
    t.c0 = (int16_t)__float2int_rn((1.f - fx) * 2048);  // saturate_cast<short>: fx in [0, 1), no clamp can trigger
    t.c1 = (int16_t)__float2int_rn(fx * 2048);
    return t;
  };
  auto ytap = [&](int dy) -> VsfTap {
    float fy = (float)((dy + 0.5) * a.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    VsfTap t;
    t.i0 = (uint16_t)min(max(sy, 0), a.sh - 1);
    t.i1 = (uint16_t)min(max(sy + 1, 0), a.sh - 1);
    t.c0 = (int16_t)__float2int_rn((1.f - fy) * 2048);
    t.c1 = (int16_t)__float2int_rn(fy * 2048);
    return t;
  };
  // loop-invariant x taps of this lane's 4 pixels
  const VsfTap t0 = xtap(min(x4 + 0, a.dw - 1)), t1 = xtap(min(x4 + 1, a.dw - 1)), t2 = xtap(min(x4 + 2, a.dw - 1)),
               t3 = xtap(min(x4 + 3, a.dw - 1));
  const uint32_t base = (uint32_t)min((int)t0.i0, a.sw - 8);  // 8-byte window [base, base+8) covers all eight taps
  auto selector = [&](const VsfTap& t) -> uint32_t {
    return (t.i0 - base) | 0x0C000C00u | ((t.i1 - base) << 16);
  };
  auto weights = [](const VsfTap& t) -> uint32_t { return (uint32_t)(uint16_t)t.c0 | ((uint32_t)(uint16_t)t.c1 << 16); };
  const uint32_t s0 = selector(t0), s1 = selector(t1), s2 = selector(t2), s3 = selector(t3);
  const uint32_t q0 = weights(t0), q1 = weights(t1), q2 = weights(t2), q3 = weights(t3);
  const uint8_t* S = a.src + (size_t)blockIdx.y * a.src_img_stride;  // wave-uniform; the lane adds `base`
  uint8_t* D = a.dst + (size_t)blockIdx.y * a.dst_img_stride;

  struct H4 {
    uint32_t a, b, c, d;
  };
  auto hpass = [&](const U8B& v) -> H4 {
    H4 h;
    h.a = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, __builtin_amdgcn_perm(v.hi, v.lo, s0)),
                                 __builtin_bit_cast(v2u16, q0), 0u, false);
    h.b = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, __builtin_amdgcn_perm(v.hi, v.lo, s1)),
                                 __builtin_bit_cast(v2u16, q1), 0u, false);
    h.c = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, __builtin_amdgcn_perm(v.hi, v.lo, s2)),
                                 __builtin_bit_cast(v2u16, q2), 0u, false);
    h.d = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, __builtin_amdgcn_perm(v.hi, v.lo, s3)),
                                 __builtin_bit_cast(v2u16, q3), 0u, false);
    return h;
  };
  // The y taps are wave-uniform: lane r evaluates output row ys + r once, v_readlane hands the result to the scalar
  // unit, and row addresses / weights live in SGPRs from there on (one tap evaluation per wave instead of one per
  // row, address arithmetic off the vector ALU).
  const int ys = strip * kStripRows;
  const VsfTap tyl = ytap(min(ys + (lane & (kStripRows - 1)), a.dh - 1));
  const uint32_t ty_rows = (uint32_t)tyl.i0 | ((uint32_t)tyl.i1 << 16);
  const uint32_t ty_wts = (uint32_t)(uint16_t)tyl.c0 | ((uint32_t)(uint16_t)tyl.c1 << 16);  // both in [0, 2048]
  // Every output row issues its two source-row loads unconditionally (rows shared with the neighbouring output row
  // hit L1): no loop-carried state, so all 2 * kStripRows loads of the strip are in flight together.
  uint32_t wts[kStripRows];
  U8B v0[kStripRows], v1[kStripRows];
#pragma unroll
  for (int r = 0; r < kStripRows; r++) {
    const uint32_t rows = __builtin_amdgcn_readlane(ty_rows, r);
    wts[r] = __builtin_amdgcn_readlane(ty_wts, r);
    const uint8_t* r0 = S + (size_t)((rows & 0xFFFFu) * (uint32_t)a.src_pitch);  // scalar
    const uint8_t* r1 = S + (size_t)((rows >> 16) * (uint32_t)a.src_pitch);
    v0[r] = *reinterpret_cast<const U8B*>(r0 + base);
    v1[r] = *reinterpret_cast<const U8B*>(r1 + base);
  }
#pragma unroll
  for (int r = 0; r < kStripRows; r++) {
    const H4 h0 = hpass(v0[r]), h1 = hpass(v1[r]);
    // VResizeLinear:  ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.   b * (S >> 4) >> 16 is the high
    // word of the 24 x 24-bit product (b << 12) * (S & ~15) (b <= 2^11, S <= 255 * 2048 < 2^19): one full-rate
    // v_mul_hi_u32_u24 instead of shift + 32-bit multiply + shift.
    const uint32_t b0 = (wts[r] & 0xFFFFu) << 12, b1 = (wts[r] >> 16) << 12;  // scalar
    auto mulhi24 = [](uint32_t x, uint32_t y) -> uint32_t {
      return (uint32_t)(((uint64_t)(x & 0xFFFFFFu) * (uint64_t)(y & 0xFFFFFFu)) >> 32);
    };
    auto vpass = [&](uint32_t u0, uint32_t u1) -> uint32_t {
      return (mulhi24(b0, u0 & 0xFFFFF0u) + mulhi24(b1, u1 & 0xFFFFF0u) + 2u) >> 2;  // <= 255
    };
    const uint32_t out = vpass(h0.a, h1.a) | (vpass(h0.b, h1.b) << 8) | (vpass(h0.c, h1.c) << 16) |
                         (vpass(h0.d, h1.d) << 24);
    uint8_t* drow = D + (size_t)((uint32_t)min(ys + r, a.dh - 1) * (uint32_t)a.dst_pitch);  // scalar; a row past the
    if (active) *reinterpret_cast<uint32_t*>(drow + (uint32_t)x4) = out;  // last one rewrites the last row's values
  }
}

// Shared-row variant (levels whose y taps advance by 1 or 2 source rows per output row, VsfLevel::resize_rows): the
// strip's R output rows touch at most R + 2 consecutive source rows, so each source row is loaded and pushed through
// the horizontal pass ONCE (R + 2 row passes instead of 2 R) and an output row picks its two entries with a
// wave-uniform branch.  The kernel is VALU-bound (two pyramid chains overlap), so instruction count is time.
// The strip computation in three steps so that a caller can keep one band's x taps across strips and have the next
// strip's source rows in flight while the current one is computed (pyramid_image_kernel).
struct StripX {      // per (level, band, lane): x taps as byte selectors + weight pairs, source window, output column
  uint32_t s0, s1, s2, s3, q0, q1, q2, q3, base;
  int x4;
  bool active;
};
template <int R>
struct StripRows {   // per strip: the R + 2 source-row windows and the lane-distributed y taps
  U8B v[R + 2];
  uint32_t ty_i0, ty_wts;
};

__device__ __forceinline__ StripX strip_setup(const ResizeArgs& a, int band) {
  const int lane = threadIdx.x & 63;
  StripX c;
  c.x4 = band * 256 + lane * 4;
  c.active = c.x4 < a.dw;
  auto xtap = [&](int dx) -> VsfTap {
    float fx = (float)((dx + 0.5) * a.scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= a.sw - 1) fx = 0, sx = a.sw - 1;
    VsfTap t;
    t.i0 = (uint16_t)sx;
    t.i1 = (uint16_t)min(sx + 1, a.sw - 1);
    t.c0 = (int16_t)__float2int_rn((1.f - fx) * 2048);
    t.c1 = (int16_t)__float2int_rn(fx * 2048);
    return t;
  };
  const VsfTap t0 = xtap(min(c.x4 + 0, a.dw - 1)), t1 = xtap(min(c.x4 + 1, a.dw - 1)),
               t2 = xtap(min(c.x4 + 2, a.dw - 1)), t3 = xtap(min(c.x4 + 3, a.dw - 1));
  c.base = (uint32_t)min((int)t0.i0, a.sw - 8);
  const uint32_t base = c.base;
  auto selector = [&](const VsfTap& t) -> uint32_t {
    return (t.i0 - base) | 0x0C000C00u | ((t.i1 - base) << 16);
  };
  auto weights = [](const VsfTap& t) -> uint32_t { return (uint32_t)(uint16_t)t.c0 | ((uint32_t)(uint16_t)t.c1 << 16); };
  c.s0 = selector(t0), c.s1 = selector(t1), c.s2 = selector(t2), c.s3 = selector(t3);
  c.q0 = weights(t0), c.q1 = weights(t1), c.q2 = weights(t2), c.q3 = weights(t3);
  return c;
}

// LDS images of a level are reached through LDS-typed pointers where the compiler cannot see that for itself (an
// address computed from a kernel argument): a generic pointer costs a flat instruction per access.
using lds_u8p = __attribute__((address_space(3))) uint8_t*;
using lds_cu8p = const __attribute__((address_space(3))) uint8_t*;
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
__device__ __forceinline__ uint32_t ld32(lds_cu8p p) { return *(const __attribute__((address_space(3))) uint32_t*)p; }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { *reinterpret_cast<uint32_t*>(p) = v; }
__device__ __forceinline__ void st32(lds_u8p p, uint32_t v) { *(__attribute__((address_space(3))) uint32_t*)p = v; }

// cv::resize's yofs for output row y (the upper tap's source row, clamped like the table the host builds)
__device__ __forceinline__ int ytap_row(const ResizeArgs& a, int y) {
  const int dy = min(y, a.dh - 1);
  const float fy = (float)((dy + 0.5) * a.scale_y - 0.5);
  return min(max((int)floorf(fy), 0), a.sh - 1);
}

// ys: first output row of the strip (any row: the R + 2 window holds for every start, VsfLevel::resize_any8, not only for
// multiples of R)
template <int R, bool ALIGNED = false, class SP = const uint8_t*>
__device__ __forceinline__ void strip_issue_from(const ResizeArgs& a, const StripX& c, SP S, int ys,
                                                 StripRows<R>& o) {  // S = the source level of this image (HBM or LDS)
  const int lane = threadIdx.x & 63;
  // y taps: lane r evaluates output row ys + r (only row index i0 and the two weights are needed)
  {
    const int dy = min(ys + (lane & (R - 1)), a.dh - 1);
    float fy = (float)((dy + 0.5) * a.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    o.ty_i0 = (uint32_t)min(max(sy, 0), a.sh - 1);
    o.ty_wts = (uint32_t)__float2int_rn((1.f - fy) * 2048) | ((uint32_t)__float2int_rn(fy * 2048) << 16);
  }
  const uint32_t first = __builtin_amdgcn_readlane(o.ty_i0, 0);
#pragma unroll
  for (int k = 0; k < R + 2; k++) {
    const SP row = S + (min(first + (uint32_t)k, (uint32_t)(a.sh - 1)) * (uint32_t)a.src_pitch);  // scalar
    if constexpr (ALIGNED) {
      // LDS source: an unaligned 8-byte read is split by the hardware and stalls the LDS queue; three aligned dwords and
      // two v_alignbyte give the same window (levels are padded to their 64-byte pitch, so the third dword exists)
      const uint32_t b0 = c.base & ~3u;
      // (the third dword is clamped into the row: when it would start past the pitch none of its bytes is needed)
      const uint32_t d0 = ld32(row + b0), d1 = ld32(row + b0 + 4u), d2 = ld32(row + min(b0 + 8u, (uint32_t)a.src_pitch - 4u));
      o.v[k].lo = __builtin_amdgcn_alignbyte(d1, d0, c.base & 3u);
      o.v[k].hi = __builtin_amdgcn_alignbyte(d2, d1, c.base & 3u);
    } else {
      o.v[k] = *reinterpret_cast<const U8B*>((const uint8_t*)row + c.base);
    }
  }
}

template <int R>
__device__ __forceinline__ void strip_issue(const ResizeArgs& a, const StripX& c, int image, int strip, StripRows<R>& o) {
  strip_issue_from<R>(a, c, a.src + (size_t)image * a.src_img_stride, strip * R, o);
}

// Output rows [ys, min(ys + R, yend)).  lcopy != nullptr: the rows are also written to an LDS image of the level (same
// pitch as in HBM; lcopy points at where the level's row 0 would be)
template <int R, class LP = uint8_t*>
__device__ __forceinline__ void strip_finish(const ResizeArgs& a, const StripX& c, int image, int ys, int yend,
                                             const StripRows<R>& in, LP lcopy = nullptr, bool to_lds = false) {
  uint8_t* D = a.dst + (size_t)image * a.dst_img_stride;
  const uint32_t first = __builtin_amdgcn_readlane(in.ty_i0, 0);
  struct H4 {
    uint32_t a, b, c, d;
  };
  H4 H[R + 2];  // horizontal sums with the low 4 bits cleared (VResizeLinear uses S >> 4)
#pragma unroll
  for (int k = 0; k < R + 2; k++) {
    auto hsum = [&](uint32_t sel, uint32_t q) -> uint32_t {
      return __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, __builtin_amdgcn_perm(in.v[k].hi, in.v[k].lo, sel)),
                                    __builtin_bit_cast(v2u16, q), 0u, false) & 0xFFFFF0u;
    };
    H[k].a = hsum(c.s0, c.q0);
    H[k].b = hsum(c.s1, c.q1);
    H[k].c = hsum(c.s2, c.q2);
    H[k].d = hsum(c.s3, c.q3);
  }
  auto mulhi24 = [](uint32_t x, uint32_t y) -> uint32_t {
    return (uint32_t)(((uint64_t)(x & 0xFFFFFFu) * (uint64_t)(y & 0xFFFFFFu)) >> 32);
  };
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (ys + r >= yend) break;  // wave-uniform
    const uint32_t wts = __builtin_amdgcn_readlane(in.ty_wts, r);
    const uint32_t b0 = (wts & 0xFFFFu) << 12, b1 = (wts >> 16) << 12;  // scalar, <= 2^23
    const bool skip = __builtin_amdgcn_readlane(in.ty_i0, r) != first + (uint32_t)r;  // then it is first + r + 1
    // ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 per pixel; the four 10-bit sums are shifted as
    // two packed pairs and their low bytes gathered with one v_perm
    auto vrow = [&](const H4& h0, const H4& h1) -> uint32_t {
      const uint32_t ta = mulhi24(b0, h0.a) + mulhi24(b1, h1.a) + 2u, tb = mulhi24(b0, h0.b) + mulhi24(b1, h1.b) + 2u;
      const uint32_t tc = mulhi24(b0, h0.c) + mulhi24(b1, h1.c) + 2u, td = mulhi24(b0, h0.d) + mulhi24(b1, h1.d) + 2u;
      const v2u16 lo = __builtin_bit_cast(v2u16, ta | (tb << 16)) >> (v2u16){2, 2};
      const v2u16 hi = __builtin_bit_cast(v2u16, tc | (td << 16)) >> (v2u16){2, 2};
      return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
    };
    uint8_t* drow = D + (size_t)((uint32_t)(ys + r) * (uint32_t)a.dst_pitch);  // scalar
    const LP lrow = lcopy + (uint32_t)(ys + r) * (uint32_t)a.dst_pitch;
    if (skip) {
      const uint32_t out = vrow(H[r + 1], H[r + 2]);
      if (c.active) *reinterpret_cast<uint32_t*>(drow + (uint32_t)c.x4) = out;
      if (to_lds && c.active) st32(lrow + (uint32_t)c.x4, out);
      asm volatile("" ::: "memory");  // keeps the two arms distinct (no select of the eight operands)
    } else {
      const uint32_t out = vrow(H[r], H[r + 1]);
      if (c.active) *reinterpret_cast<uint32_t*>(drow + (uint32_t)c.x4) = out;
      if (to_lds && c.active) st32(lrow + (uint32_t)c.x4, out);
    }
  }
}

template <int R>
__device__ __forceinline__ void resize_strip_unit(const ResizeArgs& a, int image, int strip, int band) {
  const StripX c = strip_setup(a, band);
  StripRows<R> rows;
  strip_issue<R>(a, c, image, strip, rows);
  strip_finish<R>(a, c, image, strip * R, a.dh, rows);
}

template <int R>
__global__ __launch_bounds__(256) void resize_strip_kernel(ResizeArgs a) {
  const int strip = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (strip >= a.nstrips) return;  // wave-uniform
  resize_strip_unit<R>(a, blockIdx.y, strip, blockIdx.z);
}

// Image-major tail of the pyramid for large batches: the levels that are one band wide (w <= 256; 26 of the 49 at
// 640x480) are a chain of ~8 us launch-to-launch latencies when launched one by one.  Here ONE launch walks them: a
// 1024-thread workgroup per image; each level is produced into HBM (for the other stages) AND into one of two LDS
// images, from which the next level is read -- between levels there is one workgroup barrier and no memory round trip.
// 16 waves share a level's strips; the x taps (the same for every wave: one band) are evaluated by wave 0 for the NEXT
// level while the others finish the current one.  (Walking ALL levels this way was no faster than the two chains of
// launches: the large levels are throughput-bound and want the whole chip per level.)  Used when the batch fills the
// CUs (vsf_launch_pyramid); every level of the tail must qualify for the shared-row strips and fit kTailLdsBytes.
constexpr int kTailLdsBytes = 61440;

struct PyramidArgs {
  const VsfLevel* levels;
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  uint8_t* pyr;
  uint32_t pyr_bytes;
  int l_begin, nlevels;  // levels [l_begin, nlevels), l_begin >= 2
};

__device__ __forceinline__ ResizeArgs tail_level_args(const PyramidArgs& p, int l) {
  const VsfLevel L = p.levels[l];
  const VsfLevel P = p.levels[l - 1];
  ResizeArgs a;
  a.src = p.pyr + P.offset;
  a.src_img_stride = (size_t)p.pyr_bytes;
  a.src_pitch = P.pitch;
  a.sw = P.w;
  a.sh = P.h;
  a.dst = p.pyr + L.offset;
  a.dst_img_stride = (size_t)p.pyr_bytes;
  a.dst_pitch = L.pitch;
  a.dw = L.w;
  a.dh = L.h;
  a.scale_x = __builtin_bit_cast(double, ((unsigned long long)L.rscale_x[1] << 32) | L.rscale_x[0]);
  a.scale_y = __builtin_bit_cast(double, ((unsigned long long)L.rscale_y[1] << 32) | L.rscale_y[0]);
  a.nstrips = (L.h + 7) / 8;
  return a;
}

__global__ __launch_bounds__(1024) void pyramid_image_kernel(PyramidArgs p) {
  __shared__ __attribute__((aligned(16))) uint8_t lvl[2][kTailLdsBytes];
  __shared__ uint32_t xs[2][10][64];  // StripX of a level, per lane (double buffered)
  const int image = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto publish = [&](int l) {  // wave 0: the level's x taps -> xs[l & 1]
    const ResizeArgs a = tail_level_args(p, l);
    const StripX c = strip_setup(a, 0);
    uint32_t(*o)[64] = xs[l & 1];
    o[0][lane] = c.s0, o[1][lane] = c.s1, o[2][lane] = c.s2, o[3][lane] = c.s3;
    o[4][lane] = c.q0, o[5][lane] = c.q1, o[6][lane] = c.q2, o[7][lane] = c.q3;
    o[8][lane] = c.base;
  };
  if (wave == 0) publish(p.l_begin);
  __syncthreads();
  for (int l = p.l_begin; l < p.nlevels; l++) {
    const ResizeArgs a = tail_level_args(p, l);
    StripX c;
    {
      const uint32_t(*o)[64] = xs[l & 1];
      c.s0 = o[0][lane], c.s1 = o[1][lane], c.s2 = o[2][lane], c.s3 = o[3][lane];
      c.q0 = o[4][lane], c.q1 = o[5][lane], c.q2 = o[6][lane], c.q3 = o[7][lane];
      c.base = o[8][lane];
      c.x4 = lane * 4;
      c.active = c.x4 < a.dw;
    }
    const bool from_lds = l > p.l_begin;
    uint8_t* lcopy = (l + 1 < p.nlevels) ? lvl[(l - p.l_begin) & 1] : nullptr;
    const uint8_t* lsrc = lvl[(l - p.l_begin + 1) & 1];
    const uint8_t* gsrc = a.src + (size_t)image * a.src_img_stride;
    for (int strip = wave; strip < a.nstrips; strip += 16) {
      StripRows<8> rows;
      if (from_lds)  // (workgroup-uniform; two inlined copies so that the LDS one reads with ds_read_b64)
        strip_issue_from<8, true>(a, c, lsrc, strip * 8, rows);
      else
        strip_issue_from<8>(a, c, gsrc, strip * 8, rows);
      strip_finish<8>(a, c, image, strip * 8, a.dh, rows, lcopy, lcopy != nullptr);
    }
    if (wave == 0 && l + 1 < p.nlevels) publish(l + 1);
    __syncthreads();  // (waits for this wave's LDS writes; the HBM copy is not read in this kernel)
  }
}

// The whole level chain for up to 16 images (vsf_observe_stereo, the host-pointer calls, small batches): there the 48 dependent launches
// are nothing but latency (~6.7 us each against ~1.5 us of work).  A launch of this kernel walks a CHAIN of levels
// [la, lb); the last level's rows are cut into `nslabs` slabs, one 1024-thread workgroup each, and a workgroup computes,
// level by level, exactly the rows its slab of the last level descends from -- a few rows more than its share on the
// earlier levels, which its neighbours compute as well (the same values, written twice) -- so that no workgroup ever
// waits for another.  Levels pass from one to the next through two LDS buffers (and go to HBM for the other stages); the
// chain's first level is read from HBM.  Row ranges follow cv::resize's own yofs; 8-row strips start at any row.
constexpr int kSlabMaxLevels = 32;
constexpr int kSlabBands = 3;                                    // levels up to 768 columns
constexpr size_t kSlabTapBytes = sizeof(uint32_t) * 2 * kSlabBands * 9 * 64;
constexpr size_t kSlabFixedBytes = kSlabTapBytes + 2 * kSlabMaxLevels * sizeof(int) + (kSlabMaxLevels + 1) * sizeof(VsfLevel);

struct SlabArgs {
  PyramidArgs p;    // (l_begin / nlevels unused)
  int la, lb;       // levels [la, lb), la >= 1, lb - la <= kSlabMaxLevels
  int nslabs;
  uint32_t cap;     // bytes of one LDS level buffer
  int32_t* status;  // bit 0 is raised when a slab does not fit `cap` (a host-side sizing error)
};

// (L, P: levels l and l - 1 -- the kernel keeps the chain's entries in LDS: a scalar load per level and wave from the
// table in HBM sat at the head of every level's dependency chain)
__device__ __forceinline__ ResizeArgs slab_level_args(const PyramidArgs& p, int l, const VsfLevel& L, const VsfLevel& P) {
  ResizeArgs a;
  if (l >= 2) {
    a.src = p.pyr + P.offset;
    a.src_img_stride = (size_t)p.pyr_bytes;
    a.src_pitch = P.pitch;
  } else {
    a.src = p.img0;
    a.src_img_stride = p.img0_stride;
    a.src_pitch = p.img0_pitch;
  }
  a.sw = P.w;
  a.sh = P.h;
  a.dst = p.pyr + L.offset;
  a.dst_img_stride = (size_t)p.pyr_bytes;
  a.dst_pitch = L.pitch;
  a.dw = L.w;
  a.dh = L.h;
  a.scale_x = __builtin_bit_cast(double, ((unsigned long long)L.rscale_x[1] << 32) | L.rscale_x[0]);
  a.scale_y = __builtin_bit_cast(double, ((unsigned long long)L.rscale_y[1] << 32) | L.rscale_y[0]);
  a.nstrips = (L.h + 7) / 8;
  return a;
}

__global__ __launch_bounds__(1024) void pyramid_slab_kernel(SlabArgs q) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const lds_u8p buf0 = (lds_u8p)smem, buf1 = buf0 + q.cap;
  uint32_t(*xs)[kSlabBands][9][64] = reinterpret_cast<uint32_t(*)[kSlabBands][9][64]>(smem + 2 * (size_t)q.cap);
  int* s_lo = reinterpret_cast<int*>(smem + 2 * (size_t)q.cap + kSlabTapBytes);
  int* s_hi = s_lo + kSlabMaxLevels;
  VsfLevel* s_lev = reinterpret_cast<VsfLevel*>(s_hi + kSlabMaxLevels);  // levels la - 1 .. lb - 1
  const int image = blockIdx.y, slab = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nl = q.lb - q.la;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(q.p.levels + (q.la - 1));
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_lev);
    for (int i = threadIdx.x; i < (nl + 1) * (int)(sizeof(VsfLevel) / 4); i += 1024) dst[i] = src[i];
  }
  __syncthreads();
  auto level_args = [&](int l) { return slab_level_args(q.p, l, s_lev[l - q.la + 1], s_lev[l - q.la]); };
  if (threadIdx.x == 0) {  // the slab's rows [lo, hi) of every level of the chain, from the last level backwards
    const int hz = s_lev[nl].h;
    int lo = (int)((long)hz * slab / q.nslabs), hi = (int)((long)hz * (slab + 1) / q.nslabs);
    bool fits = true;
    s_lo[nl - 1] = lo;
    s_hi[nl - 1] = hi;
    for (int l = q.lb - 1; l > q.la; l--) {
      const ResizeArgs a = level_args(l);
      if (hi > lo) {
        const int f = ytap_row(a, lo), g = ytap_row(a, hi - 1);
        lo = f;
        hi = min(g + 1, a.sh - 1) + 1;
        fits = fits && (size_t)(hi - lo) * (size_t)a.src_pitch <= (size_t)q.cap;
      }
      s_lo[l - 1 - q.la] = lo;
      s_hi[l - 1 - q.la] = hi;
    }
    if (!fits) {
      atomicOr(q.status, 1);
      s_hi[nl - 1] = s_lo[nl - 1];  // (nothing is computed)
    }
  }
  // the LAST waves evaluate the next level's x taps (one band each, -> xs[l & 1]) while the first ones work on the
  // current level's strips: a level's taps cost about as much as a strip
  auto publish = [&](int l) {
    const ResizeArgs a = level_args(l);
    const int band = 15 - wave;
    if (band < ((a.dw + 255) >> 8)) {
      const StripX c = strip_setup(a, band);
      uint32_t(*o)[64] = xs[l & 1][band];
      o[0][lane] = c.s0, o[1][lane] = c.s1, o[2][lane] = c.s2, o[3][lane] = c.s3;
      o[4][lane] = c.q0, o[5][lane] = c.q1, o[6][lane] = c.q2, o[7][lane] = c.q3;
      o[8][lane] = c.base;
    }
  };
  if (wave >= 16 - kSlabBands) publish(q.la);
  __syncthreads();
  if (s_hi[nl - 1] <= s_lo[nl - 1]) return;  // (workgroup-uniform) more slabs than rows, or the sizing error
  for (int l = q.la; l < q.lb; l++) {
    const ResizeArgs a = level_args(l);
    const int j = l - q.la;
    if (wave >= 16 - kSlabBands && l + 1 < q.lb) publish(l + 1);
    const int ylo = s_lo[j], yhi = s_hi[j];
    const int nb = (a.dw + 255) >> 8;
    const bool from_lds = j > 0;
    // (LDS images are addressed as if they began at the level's row 0)
    const bool to_lds = l + 1 < q.lb;
    const lds_u8p lcopy = ((j & 1) ? buf1 : buf0) - (uint32_t)ylo * (uint32_t)a.dst_pitch;
    const lds_cu8p lsrc = ((j & 1) ? buf0 : buf1) - (uint32_t)(from_lds ? s_lo[j - 1] : 0) * (uint32_t)a.src_pitch;
    const uint8_t* gsrc = a.src + (size_t)image * a.src_img_stride;
    // A strip is one wave's serial instruction stream (~2.5 us for 8 rows): when 8-row strips would leave half the
    // waves idle the level is cut into 4-row strips instead
    auto run = [&](auto rows_tag) {
      constexpr int R = decltype(rows_tag)::value;
      const int nunits = ((yhi - ylo + R - 1) / R) * nb;
      for (int u = wave; u < nunits; u += 16) {
        const int st = u / nb, band = u - st * nb;
        StripX c;
        {
          const uint32_t(*o)[64] = xs[l & 1][band];
          c.s0 = o[0][lane], c.s1 = o[1][lane], c.s2 = o[2][lane], c.s3 = o[3][lane];
          c.q0 = o[4][lane], c.q1 = o[5][lane], c.q2 = o[6][lane], c.q3 = o[7][lane];
          c.base = o[8][lane];
          c.x4 = band * 256 + lane * 4;
          c.active = c.x4 < a.dw;
        }
        const int ys = ylo + st * R;
        StripRows<R> rows;
        if (from_lds)  // (workgroup-uniform; two inlined copies so that the LDS one reads with ds_read_b32)
          strip_issue_from<R, true, lds_cu8p>(a, c, lsrc, ys, rows);
        else
          strip_issue_from<R>(a, c, gsrc, ys, rows);
        strip_finish<R, lds_u8p>(a, c, image, ys, yhi, rows, lcopy, to_lds);
      }
    };
    if (((yhi - ylo + 1) >> 1) * nb <= 16 - kSlabBands)  // (the last waves are busy with the next level's taps)
      run(std::integral_constant<int, 2>{});
    else if (((yhi - ylo + 3) >> 2) * nb <= 16 - kSlabBands)
      run(std::integral_constant<int, 4>{});
    else
      run(std::integral_constant<int, 8>{});
    __syncthreads();  // (waits for this wave's LDS writes; the HBM copy is not read in this kernel)
  }
}

// Sizes a chain [la, lb) for `nslabs` slabs: bytes of the largest LDS level image a workgroup keeps (levels la .. lb - 2),
// from the bound rows(l - 1) <= floor(rows(l) * scale_y) + 4 (two taps per row and the float rounding of yofs).
size_t slab_chain_bytes(const VsfLevel* lv, int la, int lb, int nslabs) {
  long rows = (lv[lb - 1].h + nslabs - 1) / nslabs + 1;
  size_t need = 0;
  for (int l = lb - 1; l > la; l--) {
    const double sy = 1. / ((double)lv[l].h / lv[l - 1].h);
    rows = std::min<long>((long)std::floor((double)rows * sy) + 4, lv[l - 1].h);
    need = std::max(need, (size_t)rows * (size_t)lv[l - 1].pitch);
  }
  return need;
}

}  // namespace

hipError_t vsf_prepare_pyramid_kernels(int lds_limit) {
  if (lds_limit < 160 * 1024 - 2048) return hipSuccess;  // (the slab kernel is then never launched)
  return hipFuncSetAttribute(reinterpret_cast<const void*>(pyramid_slab_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024 - 2048);
}

void vsf_launch_pyramid(const VsfDev& d, const VsfGeom& g, const VsfLevel* h_levels, const VsfImages& im,
                        hipStream_t s, const VsfSideStream* side) {
  // Level l depends on level l - 1 of the same image only, so the chain of 49 small dependent launches is issued once
  // per half of the batch, on two streams, interleaved: the small levels are bound by the latency of a launch's
  // dependency chain (~5 us each), not by throughput, and two chains run in the time of one.
  // (two chains: with four the host's launch rate, ~4 us per launch, becomes the limit: 0.56 -> 0.94 ms measured)
  // A batch that fills the CUs with one workgroup per image (last round at least three quarters full) hands its
  // one-band levels to the image-major tail kernel; `side` doubles as the permission (the cross-call prefetch on the aux
  // stream keeps the plain chain).
  // A batch of a frame or two (vsf_observe_stereo, the host-pointer calls) is bound by the LATENCY of the level chain,
  // not by throughput: it takes pyramid_slab_kernel for every level (VSF_OPT_PYRAMID_CHAIN 0 keeps the launches + tail kernel;
  // VSF_OPT_PYRAMID_CHAIN / _ROWS: levels per launch and rows per slab, for experiments).
  int l_tail = g.nlevels;
  // images: 2 -> 105 us (335 as launches), 8 -> 137 (335), 16 -> 212 (348), 32 -> 386 (341)
  const int few_max = d.tune ? d.tune->pyramid_few : 16;
  const bool few = side && im.n <= few_max;
  if (few && g.nlevels > 1) {
    bool ok = true;
    for (int l = 1; l < g.nlevels; l++) ok = ok && h_levels[l].resize_any8 && h_levels[l].w <= 256 * kSlabBands;
    // (the slab kernel's dynamic LDS exceeds the default limit: vsf_prepare_pyramid_kernels raised it at vsf_create; a
    // device that cannot give a workgroup that much keeps the per-level launches)
    const int chain_env = d.tune ? d.tune->pyramid_chain : 8;
    const int lds_limit = d.tune ? d.tune->lds_limit : 160 * 1024;
    if (ok && chain_env > 0 && lds_limit >= 160 * 1024 - 2048) {
      const size_t fixed = kSlabFixedBytes;
      const size_t budget = 144 * 1024;
      for (int la = 1; la < g.nlevels;) {
        int lb = std::min({la + chain_env, la + kSlabMaxLevels, g.nlevels});
        const int rows_env = d.tune ? std::max(1, d.tune->pyramid_rows) : 6;
        int nslabs = std::max(1, std::min(64, h_levels[lb - 1].h / rows_env));
        size_t need = slab_chain_bytes(h_levels, la, lb, nslabs);
        while (2 * need + fixed > budget && (nslabs < 64 || lb > la + 1)) {  // thinner slabs, then a shorter chain
          if (nslabs < 64)
            nslabs = std::min(64, nslabs * 2);
          else
            --lb;
          need = slab_chain_bytes(h_levels, la, lb, nslabs);
        }
        SlabArgs q;
        q.p.levels = d.levels;
        q.p.img0 = im.base;
        q.p.img0_stride = im.image_stride;
        q.p.img0_pitch = (int)im.row_stride;
        q.p.pyr = d.pyr;
        q.p.pyr_bytes = g.pyr_bytes;
        q.p.l_begin = la;
        q.p.nlevels = lb;
        q.la = la;
        q.lb = lb;
        q.nslabs = nslabs;
        q.cap = (uint32_t)((need + 255) & ~(size_t)255);
        q.status = d.status;
        hipLaunchKernelGGL(pyramid_slab_kernel, dim3(nslabs, im.n), dim3(1024), 2 * (size_t)q.cap + fixed, s, q);
        la = lb;
      }
      return;
    }
  }
  if (side && (im.n >= 64 || few || (d.tune && d.tune->pyramid_tail_min > 0 && im.n >= d.tune->pyramid_tail_min))) {
    static int ncu = 0;
    if (ncu == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
      if (ncu <= 0) ncu = 256;
    }
    const int rounds = (im.n + ncu - 1) / ncu;
    const int tail_min = d.tune ? d.tune->pyramid_tail_min : 0;
    if (few || 4 * im.n >= 3 * rounds * ncu || (tail_min > 0 && im.n >= tail_min)) {
      while (l_tail > 2 && h_levels[l_tail - 1].w <= 256 && h_levels[l_tail - 1].resize_rows >= 8 &&
             h_levels[l_tail - 1].pitch * h_levels[l_tail - 1].h + 16 <= kTailLdsBytes)
        --l_tail;
      if (g.nlevels - l_tail < 4) l_tail = g.nlevels;  // not worth a launch
    }
  }
  const int nchains = (side && side->n > 0 && im.n >= 2) ? 2 : 1;
  hipStream_t st[VSF_SIDE_STREAMS + 1] = {s};
  for (int c = 1; c < nchains; c++) st[c] = side->stream[c - 1];
  if (nchains > 1) {
    vsf_note(hipEventRecord(side->fork, s));
    for (int c = 1; c < nchains; c++) vsf_note(hipStreamWaitEvent(st[c], side->fork, 0));
  }
  for (int l = 1; l < l_tail; l++) {
    const VsfLevel& L = h_levels[l];
    const VsfLevel& P = h_levels[l - 1];
    for (int c = 0; c < nchains; c++) {  // interleaved issue: the chains advance together
      const int i0 = (int)((long)im.n * c / nchains), n = (int)((long)im.n * (c + 1) / nchains) - i0;
      ResizeArgs a;
      a.src = (l == 1) ? im.base + (size_t)i0 * im.image_stride : d.pyr + (size_t)i0 * g.pyr_bytes + P.offset;
      a.src_img_stride = (l == 1) ? im.image_stride : (size_t)g.pyr_bytes;
      a.src_pitch = (l == 1) ? (int)im.row_stride : P.pitch;
      a.sw = P.w;
      a.sh = P.h;
      a.dst = d.pyr + (size_t)i0 * g.pyr_bytes + L.offset;
      a.dst_img_stride = (size_t)g.pyr_bytes;
      a.dst_pitch = L.pitch;
      a.dw = L.w;
      a.dh = L.h;
      a.scale_x = 1. / ((double)L.w / P.w);
      a.scale_y = 1. / ((double)L.h / P.h);
      const int nbands = (L.w + 255) / 256;
      // rows per wave: more bytes in flight per wave on the large levels, more waves on the small ones
      const bool large = (long)L.w * L.h * n >= 4000000;
      const dim3 block(256);
      if (L.resize_rows >= 8) {
        const int R = (large && L.resize_rows >= 16) ? 16 : 8;
        a.nstrips = (L.h + R - 1) / R;
        const dim3 grid((a.nstrips + 3) / 4, n, nbands);
        if (R == 16)
          hipLaunchKernelGGL(resize_strip_kernel<16>, grid, block, 0, st[c], a);
        else
          hipLaunchKernelGGL(resize_strip_kernel<8>, grid, block, 0, st[c], a);
        continue;
      }
      const int rows = large ? 8 : 4;
      a.nstrips = (L.h + rows - 1) / rows;
      if (large)
        hipLaunchKernelGGL(resize_march_kernel<8>, dim3((a.nstrips + 3) / 4, n, nbands), dim3(256), 0, st[c], a);
      else
        hipLaunchKernelGGL(resize_march_kernel<4>, dim3((a.nstrips + 3) / 4, n, nbands), dim3(256), 0, st[c], a);
    }
  }
  for (int c = 1; c < nchains; c++) {
    vsf_note(hipEventRecord(side->join[c - 1], st[c]));
    vsf_note(hipStreamWaitEvent(s, side->join[c - 1], 0));
  }
  if (l_tail < g.nlevels) {
    PyramidArgs p;
    p.levels = d.levels;
    p.img0 = im.base;
    p.img0_stride = im.image_stride;
    p.img0_pitch = (int)im.row_stride;
    p.pyr = d.pyr;
    p.pyr_bytes = g.pyr_bytes;
    p.l_begin = l_tail;
    p.nlevels = g.nlevels;
    hipLaunchKernelGGL(pyramid_image_kernel, dim3(im.n), dim3(1024), 0, s, p);
  }
}
