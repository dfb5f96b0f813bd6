// k_ingest.hip -- SURVEY.md section 8(f) row f4, the part behind cv::imdecode: DecodeImage's
//   cv::cvtColor(COLOR_BayerBG2BGR) + cv::cvtColor(COLOR_BGR2GRAY)      (slam_frontend_main.cc:101-106)
// fused into one pass over a batch of mosaics resident in HBM (the BGR image is never materialised).
//
// imgproc/demosaicing.cpp Bayer2RGB_<uchar> is bilinear: with the 3x3 neighbourhood of an interior pixel,
//   cross = (N + S + W + E + 2) >> 2,  diag = (NW + NE + SW + SE + 2) >> 2,  hor = (W + E + 1) >> 1,  ver = (N + S + 1) >> 1,
// a BayerBG mosaic has blue at (odd x, odd y), red at (even, even), and
//   blue / red site:  own colour = centre, G = cross, the other colour = diag;
//   green site:       G = centre, the colour of the row's sites = hor, the other = ver;
// the one-pixel frame copies its inner neighbour (columns first, then rows).  color.cpp RGB2Gray<uchar>:
//   gray = (1868 B + 9617 G + 4899 R + 8192) >> 14.
// 1 byte read + 1 byte written per pixel: a lane owns 4 adjacent columns and walks 16 rows with a three-row register
// window of aligned dwords (own + both neighbours), evaluates the four sums for two pixels per packed 16-bit op and
// weights them per site with v_dot2_u32_u16; the frame columns are patched on the packed result.
#include "vsf_internal.h"

namespace {

typedef unsigned short v2u __attribute__((ext_vector_type(2)));

struct BayerArgs {
  const uint8_t* src;
  size_t src_image_stride;
  int src_pitch;
  uint8_t* dst;
  size_t dst_image_stride;
  int dst_pitch;
  int w, h;
};

// gray of interior pixel (x, y), 1 <= x <= w - 2, 1 <= y <= h - 2 (scalar; only for a last column whose source column
// belongs to another wave, i.e. widths of the form 256 k + 1)
__device__ __forceinline__ uint32_t gray_at(const uint8_t* img, int pitch, int x, int y) {
  const uint8_t* c = img + (size_t)y * pitch + x;
  const uint8_t* u = c - pitch;
  const uint8_t* d = c + pitch;
  const int cross = (u[0] + d[0] + c[-1] + c[1] + 2) >> 2, diag = (u[-1] + u[1] + d[-1] + d[1] + 2) >> 2;
  const int hor = (c[-1] + c[1] + 1) >> 1, ver = (u[0] + d[0] + 1) >> 1, cen = c[0];
  const bool ex = x & 1, ey = y & 1;
  const int G = ex == ey ? cross : cen;
  const int B = ey ? (ex ? cen : hor) : (ex ? ver : diag);
  const int R = ey ? (ex ? diag : ver) : (ex ? hor : cen);
  return (uint32_t)(1868 * B + 9617 * G + 4899 * R + 8192) >> 14;
}

__device__ __forceinline__ v2u pick(uint32_t hi, uint32_t lo, uint32_t sel) {
  return __builtin_bit_cast(v2u, __builtin_amdgcn_perm(hi, lo, sel));
}

constexpr int kStripRows = 16;  // output rows per wave

__global__ __launch_bounds__(256) void bayer_bg_gray_kernel(BayerArgs a) {
  // wave = 256 columns (4 per lane) x 16 rows, walked top to bottom with a three-row register window: a new row costs
  // three dword loads per lane (requested one row ahead) instead of nine, and a wave lives long enough to cover the
  // memory latency (one short-lived wave per row ran at a quarter of this rate).
  const int strip = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6)), image = blockIdx.z;
  const int x4 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
  const int y0 = strip * kStripRows, y1 = min(y0 + kStripRows, a.h);
  if (x4 >= a.w || y0 >= a.h) return;
  const uint8_t* img = a.src + (size_t)image * a.src_image_stride;
  uint8_t* out0 = a.dst + (size_t)image * a.dst_image_stride + x4;
  if (a.w < 3 || a.h < 3) {  // OpenCV's loops leave such images zero
    for (int y = y0; y < y1; y++)
      for (int j = 0; j < 4 && x4 + j < a.w; j++) out0[(size_t)y * a.dst_pitch + j] = 0;
    return;
  }
  // Frame columns: column 0 copies column 1 and column w - 1 copies column w - 2 -- both are fixed up on the packed
  // result (a byte move inside the lane, or one DPP hop from the left neighbour), so every lane takes the same path.
  // Columns >= w of the last dword are computed from whatever lies in the row padding and land in the output row's
  // padding (dst_row_stride >= (width + 3) & ~3).
  const int jr = (a.w - 1) - x4;  // position of the last column inside this lane's dword (0..3 when it is here)
  const bool lone_right = jr == 0 && (threadIdx.x & 63) == 0;  // ... and its source column is in another wave
  // interior lane: bytes x4 - 1 .. x4 + 4 of a source row from the aligned dwords p | d | n (buffer loads: lane offset in
  // a VGPR, row offset in an SGPR)
  const __amdgpu_buffer_rsrc_t src_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, a.src_pitch * a.h, 0x00020000);
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.dst + (size_t)image * a.dst_image_stride, 0, a.dst_pitch * a.h, 0x00020000);
  struct Row {
    uint32_t p, d, n;
  };
  auto load = [&](int r) -> Row {  // source row r, clamped into the image (the clamped rows feed frame rows only)
    const uint32_t row_off = (uint32_t)min(max(r, 0), a.h - 1) * (uint32_t)a.src_pitch;  // scalar
    return Row{__builtin_amdgcn_raw_buffer_load_b32(src_rsrc, (uint32_t)x4 - 4u, row_off, 0),
               __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, (uint32_t)x4, row_off, 0),
               __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, (uint32_t)min(x4 + 4, a.src_pitch - 4), row_off, 0)};
  };
  // 16-bit pairs (byte k, byte k + 1) of the 12-byte run p d n: on (d, p): bytes 3, 4;  on (n, d): bytes 0..4 of d | n
  const uint32_t s3 = 0x0C040C03u, s4 = 0x0C010C00u, s5 = 0x0C020C01u, s6 = 0x0C030C02u, s7 = 0x0C040C03u;
  const v2u one = {1, 1}, two = {2, 2};
  auto dot = [](v2u q, uint32_t wpair, uint32_t acc) -> uint32_t {
    return __builtin_amdgcn_udot2(q, __builtin_bit_cast(v2u, wpair), acc, false);
  };
  // One output row.  A pair of pixels is (even x, odd x) in the (low, high) halves of every packed quantity; a pixel's
  // gray is three weighted quantities of its half, so three v_dot2_u32_u16 with the weight in that half and zero in the
  // other give it without unpacking.  Which quantity carries which weight depends on the row parity only (scalar):
  //   odd row : even x = green (G cen, B hor, R ver),  odd x = blue site (B cen, G cross, R diag)
  //   even row: even x = red site (R cen, G cross, B diag),  odd x = green (G cen, R hor, B ver)
  auto yc_of = [&](int y) -> int { return min(max(y, 1), a.h - 2); };
  auto emit = [&](int y, const Row& U, const Row& C, const Row& D, bool ey) {
    const uint32_t kB = 1868u, kG = 9617u, kR = 4899u;
    // weights of (cen, cross, diag, hor, ver) for the even-x pixel (low half) ...
    const uint32_t e_cen = ey ? kG : kR, e_cross = ey ? 0u : kG, e_diag = ey ? 0u : kB, e_hor = ey ? kB : 0u,
                   e_ver = ey ? kR : 0u;
    // ... and for the odd-x pixel (high half)
    const uint32_t o_cen = (ey ? kB : kG) << 16, o_cross = (ey ? kG : 0u) << 16, o_diag = (ey ? kR : 0u) << 16,
                   o_hor = (ey ? 0u : kR) << 16, o_ver = (ey ? 0u : kB) << 16;
    uint32_t gpair[2] = {0u, 0u};
#pragma unroll
    for (int half = 0; half < 2; half++) {
      // pixel pair (x4 + 2 half, x4 + 2 half + 1): left = columns - 1, centre, right = columns + 1
      v2u ul, uc, ur, cl, cc, cr, dl, dc, dr;
      if (half == 0) {
        ul = pick(U.d, U.p, s3), uc = pick(U.n, U.d, s4), ur = pick(U.n, U.d, s5);
        cl = pick(C.d, C.p, s3), cc = pick(C.n, C.d, s4), cr = pick(C.n, C.d, s5);
        dl = pick(D.d, D.p, s3), dc = pick(D.n, D.d, s4), dr = pick(D.n, D.d, s5);
      } else {
        ul = pick(U.n, U.d, s5), uc = pick(U.n, U.d, s6), ur = pick(U.n, U.d, s7);
        cl = pick(C.n, C.d, s5), cc = pick(C.n, C.d, s6), cr = pick(C.n, C.d, s7);
        dl = pick(D.n, D.d, s5), dc = pick(D.n, D.d, s6), dr = pick(D.n, D.d, s7);
      }
      const v2u ns = uc + dc, we = cl + cr;
      const v2u cross = (ns + we + two) >> two, diag = (ul + ur + dl + dr + two) >> two;
      const v2u hor = (we + one) >> one, ver = (ns + one) >> one;
      // (a zero weight makes its term vanish; the row parity is uniform, so the unused products cost no branch)
      const uint32_t g0 = dot(cc, e_cen, dot(cross, e_cross, dot(diag, e_diag, dot(hor, e_hor, dot(ver, e_ver, 8192u)))));
      const uint32_t g1 = dot(cc, o_cen, dot(cross, o_cross, dot(diag, o_diag, dot(hor, o_hor, dot(ver, o_ver, 8192u)))));
      gpair[half] = (g0 >> 14) | ((g1 >> 14) << 8);
    }
    uint32_t g = gpair[0] | (gpair[1] << 16);
    const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x138, 0xF, 0xF, false);  // lane - 1's dword
    if (x4 == 0) g = (g & 0xFFFFFF00u) | ((g >> 8) & 0xFFu);                      // column 0 <- column 1
    if (jr == 0) g = (g & 0xFFFFFF00u) | (lone_right ? gray_at(img, a.src_pitch, a.w - 2, yc_of(y)) : (left >> 24));
    if (jr >= 1 && jr <= 3) {                                                     // column w - 1 <- column w - 2
      const uint32_t sh = 8u * (uint32_t)jr;
      g = (g & ~(0xFFu << sh)) | (((g >> (sh - 8u)) & 0xFFu) << sh);
    }
    __builtin_amdgcn_raw_buffer_store_b32(g, dst_rsrc, (uint32_t)x4, (uint32_t)y * (uint32_t)a.dst_pitch, 0);
  };
  // output row y is computed at yc = clamp(y, 1, h - 2) from source rows yc - 1 .. yc + 1.  Inside the image
  // yc advances with y, so the window slides; the first and the last row of the image repeat their neighbour's window.
  int yc = min(max(y0, 1), a.h - 2);
  Row U = load(yc - 1), C = load(yc), D = load(yc + 1);
  for (int y = y0; y < y1; y++) {
    const int ycn = min(max(y + 1, 1), a.h - 2);  // next output row's centre row (wave-uniform)
    Row N = D;
    if (ycn != yc) N = load(ycn + 1);  // requested before this row's arithmetic
    emit(y, U, C, D, yc & 1);
    if (ycn != yc) {
      U = C;
      C = D;
      D = N;
      yc = ycn;
    }
  }
}

}  // namespace

void vsf_launch_bayer_bg_gray(const uint8_t* d_src, int n, int w, int h, size_t src_image_stride, int src_pitch,
                              uint8_t* d_dst, size_t dst_image_stride, int dst_pitch, hipStream_t s) {
  BayerArgs a{d_src, src_image_stride, src_pitch, d_dst, dst_image_stride, dst_pitch, w, h};
  const int nstrips = (h + kStripRows - 1) / kStripRows;
  hipLaunchKernelGGL(bayer_bg_gray_kernel, dim3((w + 255) / 256, (nstrips + 3) / 4, n), dim3(256), 0, s, a);
}
