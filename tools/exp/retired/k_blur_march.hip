// k_blur_march.hip -- RETIRED from the product in round 6 (it was VSF_OPT_BLUR_MARCH): round 2's vector-ALU Gaussian blur,
// bit-identical to the matrix-core kernel of csrc/k_blur.hip and slower (1.5 against 0.93 ms per 512 images; records:
// profiles/r03).  Kept for reference; it compiled as part of csrc/k_blur.hip (same includes, same BlurArgs launcher).
// Streaming "march" kernel, no LDS and no barriers: a wave owns a band of 248 columns (62 lanes x 4 pixels; lanes
// 0 and 63 only carry the 4-pixel halo) and walks down a strip of rows; per row each lane issues ONE coalesced
// 32-bit load and ONE 32-bit store.  The two passes commute (exact integer sums, one rounding at the end), so the
// kernel runs the COLUMN pass first, on bytes widened to packed 16 bit (two pixels per v_pk_add/v_pk_mad_u16; the
// 7-tap sum is <= 255 * 257 = 65535, it just fits), over a 7-row register window, and then the ROW pass on those
// 16-bit sums with v_dot2_u32_u16 (two taps per instruction, 32-bit accumulator): a pixel's seven taps are four
// dot2 on the lane's own and its neighbours' packed pairs (4 DPP wave shifts per row).
// HBM traffic is the compulsory P read + P write (plus 6 halo rows per 64-row strip).
#include "../../../vision_slam_frontend_amd/csrc/vsf_internal.h"

namespace {

__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - 2 - p;
  return p < 0 ? 0 : (p >= len ? len - 1 : p);  // clamp only reachable for len < 4 (one reflection suffices otherwise)
}
constexpr int kBandCols = VSF_BLUR_BAND_COLS;   // output columns per wave
constexpr int kStripRows = VSF_BLUR_STRIP_ROWS; // output rows per wave

struct BlurArgs {
  const VsfLevel* levels;
  const uint32_t* units;  // level << 24 | band << 16 | two strips per wave << 15 | strip
  int nunits;
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;
  uint8_t* blur;
  uint32_t pyr_bytes;
  int k0, k1, k2, k3;  // fixed-point kernel taps (k[3-i] == k[3+i])
};

typedef unsigned short v2u __attribute__((ext_vector_type(2)));
typedef short v2s_ __attribute__((ext_vector_type(2)));

struct Px4 {  // one row's 4 pixels as two packed pairs of 16-bit values: lo = (x, x+1), hi = (x+2, x+3)
  v2u lo, hi;
};


__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {  // lane i <- lane i-1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) {  // lane i <- lane i+1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
}

__device__ __forceinline__ Px4 widen(uint32_t d) {
  Px4 r;
  r.lo = __builtin_bit_cast(v2u, __builtin_amdgcn_perm(0u, d, 0x0C010C00u));
  r.hi = __builtin_bit_cast(v2u, __builtin_amdgcn_perm(0u, d, 0x0C030C02u));
  return r;
}

__device__ __forceinline__ uint32_t dot2(v2u a, uint32_t kpair, uint32_t acc) {
  return __builtin_amdgcn_udot2(a, __builtin_bit_cast(v2u, kpair), acc, false);
}

__device__ __forceinline__ v2u dpp_shr1(v2u v) { return __builtin_bit_cast(v2u, wave_shr1(__builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ v2u dpp_shl1(v2u v) { return __builtin_bit_cast(v2u, wave_shl1(__builtin_bit_cast(uint32_t, v))); }

// N / 65536 with OpenCV's rounding, left in bits 16..23 (saturated): tie_up = 1 rounds half-up (scalar tail),
// tie_up = 0 rounds half-to-even (SSE2 cvtps2dq columns).
__device__ __forceinline__ uint32_t round_fix16(uint32_t n, uint32_t tie_up) {
  const uint32_t b = ((n >> 16) & 1u) | tie_up;
  return min(n + 0x7FFFu + b, 0x00FFFFFFu);
}

// HALF = false: the wave is one unit, a band of 248 columns x a strip of 64 rows.
// HALF = true : the level's LAST band is narrow (<= 120 columns = 30 lanes + 2 halo lanes) and the two 32-lane halves of
//               the wave take it in two consecutive strips (the second half idles when the level ends first): the
//               pyramid's widths leave such a remainder on most levels, 66 % -> 75 % of the lanes carry pixels.  The
//               wave shifts of the row pass cross the halves only into halo lanes, whose sums nobody uses.
template <bool HALF>
__device__ __forceinline__ void blur_march_body(const BlurArgs& a, uint32_t ud, int image) {
  const int lane = threadIdx.x & 63;
  const int hl = HALF ? (lane & 31) : lane;  // lane inside its cell
  const bool upper = HALF && lane >= 32;    // the second strip's half
  const int level = (int)(ud >> 24), band = (int)((ud >> 16) & 0xFF), strip = (int)(ud & 0x7FFF);
  const VsfLevel L = a.levels[level];
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
  const uint32_t k0 = (uint32_t)a.k0, k1 = (uint32_t)a.k1, k2 = (uint32_t)a.k2, k3 = (uint32_t)a.k3;
  const v2u K0 = {(unsigned short)k0, (unsigned short)k0}, K1 = {(unsigned short)k1, (unsigned short)k1},
            K2 = {(unsigned short)k2, (unsigned short)k2}, K3 = {(unsigned short)k3, (unsigned short)k3};
  // tap pairs (low half, high half) of the row pass
  const uint32_t p_0k0 = k0 << 16, p_k0_0 = k0, p_k1k2 = k1 | (k2 << 16), p_k3k2 = k3 | (k2 << 16),
                 p_k1k0 = k1 | (k0 << 16), p_k0k1 = k0 | (k1 << 16), p_k2k3 = k2 | (k3 << 16),
                 p_k2k1 = k2 | (k1 << 16);
  const int w = L.w, h = L.h;
  const int c0 = band * kBandCols - 4 + 4 * hl;  // first column of this lane's dword
  // BORDER_REFLECT_101 columns without a divergent slow path: every column a lane can need lies in an aligned
  // 8-byte window [a0, a0 + 8) of the row (a0 = c0 inside the image, 0 left of it, (w - 4) & ~3 at the right edge;
  // level widths are >= 8), so a lane loads that window's two dwords and picks its 4 bytes with one v_perm_b32.
  const bool interior = c0 >= 0 && c0 + 3 < w;
  const int a0 = interior ? c0 : (c0 < 0 ? 0 : ((w - 4) & ~3));
  const int a1 = min(a0 + 4, pitch - 4);
  uint32_t bsel = 0x03020100u;
  if (!interior) {
    bsel = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int sidx = reflect101(c0 + j, w) - a0;  // columns beyond w + 2 are never used: any in-window byte will do
      sidx = min(max(sidx, 0), 7);
      if (sidx >= 4 && a1 != a0 + 4) sidx = 3;  // (window clipped at the end of the row: only reachable there)
      bsel |= (uint32_t)sidx << (8 * j);
    }
  }
  const uint32_t tile_col = ((uint32_t)(max(c0, 0) >> 5) << 7) + (uint32_t)(max(c0, 0) & 31);
  // buffer resources over the source level and the blurred level (raw buffers, 32-bit data format; reads past the end
  // return 0, writes past the end are dropped)
  const __amdgpu_buffer_rsrc_t src_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, pitch * L.h, 0x00020000);
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, L.pitch * ((L.h + 7) & ~7), 0x00020000);
  const bool all_interior = __all(interior);
  const bool writer = hl >= 1 && hl <= (HALF ? 30 : 62) && c0 < w;
  const int ys = strip * kStripRows, ye = min(ys + kStripRows, h);  // (the first strip: never the shorter one)
  const uint32_t t0 = c0 + 0 < L.blur_vec_end ? 0u : 1u, t1 = c0 + 1 < L.blur_vec_end ? 0u : 1u,
                 t2 = c0 + 2 < L.blur_vec_end ? 0u : 1u, t3 = c0 + 3 < L.blur_vec_end ? 0u : 1u;

  // A row is requested two steps before the step that needs it (the raw dwords wait in registers): with the load
  // issued right in front of its use the kernel depended on eight resident waves per SIMD to cover the latency.
  struct Raw {
    uint32_t d0, d1;
  };
  auto request = [&](int y) -> Raw {
    // buffer load: lane offset in a VGPR, row offset in an SGPR -- no vector address arithmetic per row
    const uint32_t row_off = (uint32_t)reflect101(y, h) * (uint32_t)pitch;  // scalar
    Raw r;
    if constexpr (HALF) {  // two rows, one per half: the row offset joins the lane offset
      const uint32_t row_up = (uint32_t)reflect101(y + kStripRows, h) * (uint32_t)pitch;  // scalar
      const uint32_t ro = upper ? row_up : row_off;
      r.d0 = __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, a0 + ro, 0, 0);
      r.d1 = all_interior ? 0u : __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, a1 + ro, 0, 0);  // wave-uniform
    } else {
      r.d0 = __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, a0, row_off, 0);
      r.d1 = all_interior ? 0u : __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, a1, row_off, 0);  // wave-uniform
    }
    return r;
  };
  auto unpack = [&](const Raw& r) -> Px4 {
    if (all_interior) return widen(r.d0);  // wave-uniform
    return widen(__builtin_amdgcn_perm(r.d1, r.d0, bsel));
  };
  auto fetch = [&](int y) -> Px4 { return unpack(request(y)); };
  // One output row: column pass over the 7-row window (r0 = row y-3 ... r6 = row y+3), then the row pass.
  auto emit = [&](int y, const Px4& r0, const Px4& r1, const Px4& r2, const Px4& r3, const Px4& r4, const Px4& r5,
                  const Px4& r6) {
    const v2u clo = K3 * r3.lo + (K2 * (r2.lo + r4.lo) + (K1 * (r1.lo + r5.lo) + K0 * (r0.lo + r6.lo)));
    const v2u chi = K3 * r3.hi + (K2 * (r2.hi + r4.hi) + (K1 * (r1.hi + r5.hi) + K0 * (r0.hi + r6.hi)));
    const v2u llo = dpp_shr1(clo), lhi = dpp_shr1(chi), rlo = dpp_shl1(clo), rhi = dpp_shl1(chi);
    // pixel x = c0 + j sees C[x-3 .. x+3]; lanes hold C as (lo.x, lo.y, hi.x, hi.y) = columns c0 .. c0+3
    const uint32_t n0 = dot2(chi, p_k1k0, dot2(clo, p_k3k2, dot2(lhi, p_k1k2, dot2(llo, p_0k0, 0u))));
    const uint32_t n1 = dot2(rlo, p_k0_0, dot2(chi, p_k2k1, dot2(clo, p_k2k3, dot2(lhi, p_k0k1, 0u))));
    const uint32_t n2 = dot2(rlo, p_k1k0, dot2(chi, p_k3k2, dot2(clo, p_k1k2, dot2(lhi, p_0k0, 0u))));
    const uint32_t n3 = dot2(rhi, p_k0_0, dot2(rlo, p_k2k1, dot2(chi, p_k2k3, dot2(clo, p_k0k1, 0u))));
    if (writer && (!HALF || !upper || y + kStripRows < h)) {
      const uint32_t v0 = round_fix16(n0, t0), v1 = round_fix16(n1, t1), v2 = round_fix16(n2, t2),
                     v3 = round_fix16(n3, t3);
      // byte 2 of each value -> bytes 0..3
      const uint32_t lo2 = __builtin_amdgcn_perm(v1, v0, 0x0C0C0602u), hi2 = __builtin_amdgcn_perm(v3, v2, 0x06020C0Cu);
      // tiled store (VSF_BLUR_TILE_OFFSET): the row part is scalar, the lane part loop-invariant (c0 % 4 == 0)
      const uint32_t drow = (uint32_t)(y >> 2) * (uint32_t)(L.pitch * 4) + (uint32_t)((y & 3) << 5);
      if constexpr (HALF) {
        const int yu = y + kStripRows;
        const uint32_t drow_up = (uint32_t)(yu >> 2) * (uint32_t)(L.pitch * 4) + (uint32_t)((yu & 3) << 5);
        __builtin_amdgcn_raw_buffer_store_b32(lo2 | hi2, dst_rsrc, tile_col + (upper ? drow_up : drow), 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(lo2 | hi2, dst_rsrc, tile_col, drow, 0);
      }
    }
  };

  Px4 W0, W1, W2, W3, W4, W5, W6;
  W0 = fetch(ys - 3);
  W1 = fetch(ys - 2);
  W2 = fetch(ys - 1);
  W3 = fetch(ys);
  W4 = fetch(ys + 1);
  W5 = fetch(ys + 2);
  // The window rotates through seven register sets: row y uses (Wq .. Wq+6 mod 7) and refills the oldest one from the
  // raw row requested two steps earlier.
  Raw ra = request(ys + 3), rb = request(ys + 4);
#define VSF_BLUR_STEP(j, w0, w1, w2, w3, w4, w5, w6) \
  if (y + (j) >= ye) break;                           \
  w6 = unpack(ra);                                    \
  ra = rb;                                            \
  rb = request(y + (j) + 5);                          \
  emit(y + (j), w0, w1, w2, w3, w4, w5, w6);
  for (int y = ys;; y += 7) {
    VSF_BLUR_STEP(0, W0, W1, W2, W3, W4, W5, W6)
    VSF_BLUR_STEP(1, W1, W2, W3, W4, W5, W6, W0)
    VSF_BLUR_STEP(2, W2, W3, W4, W5, W6, W0, W1)
    VSF_BLUR_STEP(3, W3, W4, W5, W6, W0, W1, W2)
    VSF_BLUR_STEP(4, W4, W5, W6, W0, W1, W2, W3)
    VSF_BLUR_STEP(5, W5, W6, W0, W1, W2, W3, W4)
    VSF_BLUR_STEP(6, W6, W0, W1, W2, W3, W4, W5)
  }
#undef VSF_BLUR_STEP
}

__global__ __launch_bounds__(256) void blur_march_kernel(BlurArgs a) {
  // (readfirstlane: the unit and everything derived from it -- level, strip, row addresses -- stays on the scalar unit)
  const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (unit >= a.nunits) return;  // wave-uniform
  const uint32_t ud = a.units[unit];
  if (ud & 0x8000u)  // (wave-uniform) two strips of a narrow last band
    blur_march_body<true>(a, ud, blockIdx.y);
  else
    blur_march_body<false>(a, ud, blockIdx.y);
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

