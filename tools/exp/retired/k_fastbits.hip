// k_fastbits.hip -- K2 for batches, second form: the FAST-9/16 segment test on bit planes, the score only where the test
// says "corner".
//
// Restates cv::FAST_t<16> / cornerScore<16> (features2d/fast.cpp, fast_score.cpp) as ORB's computeKeyPoints calls them
// (slam_frontend.cc:274, parameters :205-213) and writes exactly what k_fast.hip's march kernel writes: per cell
// (248-column band x 32-row strip) a candidate segment in raster order + the start of every row (vsf_gather.h).
//
// k_fast.hip computes the score of every pixel (72 packed min / max per pixel pair) and is at the vector ALU's issue
// ceiling.  Here a lane holds 32 adjacent pixels of a row as eight bit planes (vsf_bitslice.h): comparing a row with
// the centre row +- t is eight v_bitop3_b32 (full issue rate) for 32 pixels, and with the forward scheme only 16 such
// comparisons are made per row.  What the test finds -- a few percent of the pixels in a photograph, 18 % in the noisy
// synthetic scenes of the bench -- is compacted over the wave and scored two corners per lane (the polarity is known,
// so one packed min / max tree serves two corners), from the raw rows kept in LDS; scores go into an LDS byte map, the
// strict 3x3 suppression reads it back, survivors are appended in raster order.
//
// Work: an item = (level, band, chain of <= 4 strips); its lanes-per-image nl = ceil((band columns + 8) / 32) <= 8, and a
// wave marches 64 / nl IMAGES through the same item side by side, so geometry and control flow are wave-uniform and
// only the image base differs per lane group.  One wave per workgroup (its LDS: 8 raw rows, 3 score rows, a list).
#include <algorithm>
#include <vector>

#include "vsf_bitslice.h"  // (beside this file)
#include "../../../vision_slam_frontend_amd/csrc/vsf_internal.h"

namespace {

using namespace vsf_bs;

typedef short v2s __attribute__((ext_vector_type(2)));
constexpr int SR = VSF_FAST_STRIP_ROWS;
constexpr int ROWB = 2048;        // bytes of one wave row (64 lanes x 32 pixels)
#ifndef VSF_FB_RAW_SLOTS
#define VSF_FB_RAW_SLOTS 8
#endif
constexpr int RAW_SLOTS = VSF_FB_RAW_SLOTS;  // raw rows kept in LDS (7 needed: the circle of the centre row)
constexpr int LIST_CAP = 512;     // corners compacted per round

struct FastBitsArgs {
  const VsfLevel* levels;
  const uint2* items;     // x: level << 24 | band << 16 | first strip;  y: strips << 8 | lanes per image
  int nunits;
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;
  uint32_t pyr_bytes;
  uint32_t* cand;
  uint32_t cand_entries;
  uint16_t* rowstart;
  int threshold;
  int nimages;
};

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {  // lane i <- lane i-1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) {  // lane i <- lane i+1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
}
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}
template <class T>
__device__ __forceinline__ T uniform_copy(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dwords");
  T out;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(p);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (size_t i = 0; i < sizeof(T) / 4; i++) dst[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)src[i]);
  return out;
}

__device__ __forceinline__ v2s as_v2s(uint32_t v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ v2s vmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s vmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
// three-input packed minimum / maximum of values 0..255 in 16-bit halves (f16 denormals order like the integers: k_fast.hip)
__device__ __forceinline__ v2s vmin3(v2s a, v2s b, v2s c) {
  v2s r;
  asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ v2s vmax3(v2s a, v2s b, v2s c) {
  v2s r;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// bits [lo, hi) of a 32-bit word (any integers)
__device__ __forceinline__ uint32_t range_mask(int lo, int hi) {
  lo = max(lo, 0);
  hi = min(hi, 32);
  if (hi <= lo) return 0u;
  const uint32_t upto_hi = hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u);
  return upto_hi & ~((1u << lo) - 1u);
}

// max over the 16 arcs of the minimum over the arc's 9 values, per 16-bit half (k_fast.hip score_from_circle, one polarity)
__device__ __forceinline__ v2s arc_max_min(const v2s (&d)[16]) {
  v2s E[8], F[8], ta[8];
#pragma unroll
  for (int j = 0; j < 8; j++) E[j] = vmin(d[2 * j + 1], d[(2 * j + 2) & 15]);
#pragma unroll
  for (int j = 0; j < 8; j++) F[j] = vmin(E[j], E[(j + 1) & 7]);
#pragma unroll
  for (int j = 0; j < 8; j++) ta[j] = vmin3(F[j], F[(j + 2) & 7], vmax(d[2 * j], d[(2 * j + 9) & 15]));
  return vmax(vmax3(vmax3(ta[0], ta[1], ta[2]), vmax3(ta[3], ta[4], ta[5]), ta[6]), ta[7]);
}

#ifndef VSF_FB_WAVES
#define VSF_FB_WAVES 1
#endif
constexpr int FB_WAVES = VSF_FB_WAVES;  // waves per workgroup; each works alone (its own item, its own LDS): they only share a CU
constexpr int FB_LDS_PER_WAVE = RAW_SLOTS * ROWB + 3 * ROWB + LIST_CAP * 2 + 64 * 4;

// (between the lanes of ONE wave LDS operations take effect in program order: what is needed is that the compiler keeps
// that order)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64 * FB_WAVES) void fast_bits_kernel(FastBitsArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fb_lds[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* const my = fb_lds + wave * FB_LDS_PER_WAVE;
  uint8_t* const raw = my;
  uint8_t* const smap = my + RAW_SLOTS * ROWB;
  uint16_t* const list = reinterpret_cast<uint16_t*>(my + RAW_SLOTS * ROWB + 3 * ROWB);
  uint32_t* const kmask = reinterpret_cast<uint32_t*>(my + RAW_SLOTS * ROWB + 3 * ROWB + LIST_CAP * 2);

  const int lane = threadIdx.x & 63;
  const uint2 it = uniform_copy(a.items + blockIdx.x);
  const int level = (int)(it.x >> 24), band = (int)((it.x >> 16) & 0xFF), strip0 = (int)(it.x & 0xFFFF);
  const int nstr = (int)(it.y >> 8), nl = (int)(it.y & 0xFF);
  const int G = 64 / nl;
  const int image0 = ((int)blockIdx.y * FB_WAVES + wave) * G;
  if (image0 >= a.nimages) return;  // wave-uniform
  const VsfLevel L = uniform_copy(a.levels + level);
  const int t = a.threshold;

  const int g = lane / nl, j = lane - g * nl;
  const bool active = g < G && image0 + g < a.nimages;
  const int image = image0 + g;
  const int bx0 = L.fast_a0 + VSF_FAST_BAND_COLS * band;
  const int c0 = bx0 - 4 + 32 * j;  // column of this lane's bit 0
  const int ys0 = L.y_lo + strip0 * SR;
  const int ys_end = min(ys0 + nstr * SR, L.y_hi);
  const int nrows = ys_end - ys0;
  const int hrow = L.h;

  // pixels that may carry a score (FAST's 3-pixel rim, one column beyond the band for the suppression) / be emitted
  const int sx_lo = max(max(L.x_lo - 1, 3), bx0 - 1), sx_hi = min(min(L.x_hi + 1, L.w - 3), bx0 + VSF_FAST_BAND_COLS + 1);
  const int ex_lo = max(L.x_lo, bx0), ex_hi = min(L.x_hi, bx0 + VSF_FAST_BAND_COLS);
  const uint32_t smask = active ? range_mask(sx_lo - c0, sx_hi - c0) : 0u;
  const uint32_t emask = active ? range_mask(ex_lo - c0, ex_hi - c0) : 0u;

  // source rows through a buffer descriptor that starts at this wave's first image: lane offset = image in the wave +
  // column (a VGPR), row offset a scalar; anything outside the images' bytes reads as 0
  const uint8_t* src;
  uint32_t img_stride, pitch;
  if (level == 0) {
    src = a.img0 + (size_t)image0 * a.img0_stride;
    img_stride = (uint32_t)a.img0_stride;
    pitch = (uint32_t)a.img0_pitch;
  } else {
    src = a.pyr + (size_t)image0 * a.pyr_bytes + L.offset;
    img_stride = a.pyr_bytes;
    pitch = (uint32_t)L.pitch;
  }
  const int nimg_here = min(G, a.nimages - image0);
  const uint32_t src_bytes = (uint32_t)(nimg_here - 1) * img_stride + pitch * (uint32_t)hrow;
  const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, src_bytes, 0x00020000);
  const uint32_t col_off = active ? (uint32_t)g * img_stride + (uint32_t)c0 : 0xFFFFFF00u;
  auto load_row = [&](int ri, uint32_t(&w)[8]) {  // row index from ys0 - 4
    const int y = min(max(ys0 - 4 + ri, 0), hrow - 1);
    const uint32_t ro = (uint32_t)y * pitch;
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, col_off + 4u * k, ro, 0);
  };

  // per-cell outputs of this lane's image
  const int unit_first = strip0 * L.nbands + band;
  uint32_t* seg = a.cand + (size_t)(active ? image : 0) * a.cand_entries + L.cand_offset + (size_t)unit_first * L.seg_cap;
  uint16_t* rs = a.rowstart + ((size_t)(active ? image : 0) * a.nunits + L.unit0 + unit_first) * VSF_FAST_RS_STRIDE;
  const int seg_cap = L.seg_cap;
  const bool gfirst = active && j == 0;
  const int lane_gfirst = g * nl, lane_glast = g * nl + nl - 1;
  if (gfirst) rs[0] = 0;
  int gcount = 0;       // candidates of the current cell so far (same in all lanes of a group)
  int cell_row0 = ys0;  // first image row of the current cell

  uint32_t tm[8];
  threshold_masks(t, tm);
  // (the masks must live in vector registers: as scalar operands they halve the issue rate of every instruction using them)
#pragma unroll
  for (int i = 0; i < 8; i++) asm volatile("" : "+v"(tm[i]));

  uint32_t rows[4][8];
  FastHistory hst;
  hst.clear();
  uint32_t wN[8], wNN[8];
  {
    uint32_t w0[8], w1[8], w2[8];
    load_row(0, w0), load_row(1, w1), load_row(2, w2), load_row(3, wN), load_row(4, wNN);
    transpose_row(w0, rows[1]), transpose_row(w1, rows[2]), transpose_row(w2, rows[3]);
    uint4* r0 = reinterpret_cast<uint4*>(raw + 0 * ROWB + lane * 32);
    r0[0] = make_uint4(w0[0], w0[1], w0[2], w0[3]), r0[1] = make_uint4(w0[4], w0[5], w0[6], w0[7]);
    uint4* r1 = reinterpret_cast<uint4*>(raw + 1 * ROWB + lane * 32);
    r1[0] = make_uint4(w1[0], w1[1], w1[2], w1[3]), r1[1] = make_uint4(w1[4], w1[5], w1[6], w1[7]);
    uint4* r2 = reinterpret_cast<uint4*>(raw + 2 * ROWB + lane * 32);
    r2[0] = make_uint4(w2[0], w2[1], w2[2], w2[3]), r2[1] = make_uint4(w2[4], w2[5], w2[6], w2[7]);
  }
  kmask[lane] = 0u;
  {
    uint4* z = reinterpret_cast<uint4*>(smap + lane * 96);  // 64 x 96 = the three score rows
#pragma unroll
    for (int i = 0; i < 6; i++) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  uint32_t c_prev = 0u;  // corners of the previous centre row that may be emitted

  // Step q: centre row y = ys0 - 4 + q.  Its comparisons with rows y .. y + 3 are made; from q = 3 on (y = ys0 - 1) the
  // corner mask of row y is complete and its corners are scored; the suppression and emission of row y - 1 follow.
  const int nsteps = nrows + 5;  // y = ys0 - 4 .. ys_end
  for (int q = 0; q < nsteps; q++) {
    const int y = ys0 - 4 + q;
    // ---- the row that enters the window: row index q + 3
#pragma unroll
    for (int b = 0; b < 8; b++) rows[0][b] = rows[1][b], rows[1][b] = rows[2][b], rows[2][b] = rows[3][b];
    transpose_row(wN, rows[3]);
    {
      uint4* r = reinterpret_cast<uint4*>(raw + ((q + 3) & (RAW_SLOTS - 1)) * ROWB + lane * 32);
      r[0] = make_uint4(wN[0], wN[1], wN[2], wN[3]), r[1] = make_uint4(wN[4], wN[5], wN[6], wN[7]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) wN[k] = wNN[k];
    load_row(q + 5, wNN);

    // ---- dense: forward comparisons of the centre row, corner masks
    uint32_t hi[8], lo[8], hi_p[8], hi_n[8], lo_p[8], lo_n[8];
    saturating_add_sub(rows[0], tm, hi, lo);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      hi_p[b] = wave_shr1(hi[b]), hi_n[b] = wave_shl1(hi[b]);
      lo_p[b] = wave_shr1(lo[b]), lo_n[b] = wave_shl1(lo[b]);
    }
    uint32_t mb[8], md[8], mb_p[8], mb_n[8], md_p[8], md_n[8];
    forward_masks(rows, hi, hi_p, hi_n, lo, lo_p, lo_n, mb, md);
#pragma unroll
    for (int f = 0; f < 8; f++) {
      // (the neighbour a mask needs depends on the direction it moves back: FWD dx > 0 reads the lane to the right)
      const bool right = f == 1 || f == 3 || f == 5 || f == 7, left = f == 2 || f == 4 || f == 6;
      mb_p[f] = left ? wave_shr1(mb[f]) : 0u, md_p[f] = left ? wave_shr1(md[f]) : 0u;
      mb_n[f] = right ? wave_shl1(mb[f]) : 0u, md_n[f] = right ? wave_shl1(md[f]) : 0u;
    }
    uint32_t br, dk;
    corner_masks(mb, mb_p, mb_n, md, md_p, md_n, hst, &br, &dk);
    hst.push(mb, md);

    const bool score_row = q >= 3 && y >= 3 && y < hrow - 3;  // (y <= ys_end by the loop bound)
    const uint32_t C = score_row ? (br | dk) & smask : 0u;

    // ---- sparse 1: scores of the corners of row y into score row q % 3
    uint8_t* s_dn = smap + (q % 3) * ROWB;          // row y
    uint8_t* s_mid = smap + ((q + 2) % 3) * ROWB;   // row y - 1
    uint8_t* s_up = smap + ((q + 1) % 3) * ROWB;    // row y - 2
    {
      uint4* z = reinterpret_cast<uint4*>(s_dn + lane * 32);
      z[0] = make_uint4(0u, 0u, 0u, 0u), z[1] = make_uint4(0u, 0u, 0u, 0u);
    }
    {
      const int cnt = __popc(C);
      const int incl = wave_incl_scan(cnt);
      const int total = __builtin_amdgcn_readlane(incl, 63);
      if (total > 0) {  // wave-uniform
        uint32_t m = C;
        int idx = incl - cnt;
        // LDS byte offsets of the seven raw rows y - 3 .. y + 3, less the 3 columns the window starts to the left
        int rowoff[7];
#pragma unroll
        for (int dy = -3; dy <= 3; dy++) rowoff[dy + 3] = ((q + dy) & (RAW_SLOTS - 1)) * ROWB - 3;
        for (int R = 0; R < total; R += LIST_CAP) {
          while (true) {
            const bool act = m != 0u && idx < R + LIST_CAP;
            if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
            if (act) {
              const int b = __builtin_ctz(m);
              // entry = position in the wave row | polarity << 11 (1 = brighter)
              list[idx - R] = (uint16_t)((lane << 5) | b | (((br >> b) & 1u) << 11));
              m &= m - 1u;
              ++idx;
            }
          }
          wave_sync();
          const int nround = min(LIST_CAP, total - R);
          for (int p = 0; p < nround; p += 128) {
            const bool v0 = p + lane < nround, v1 = p + 64 + lane < nround;
            const uint32_t e0 = v0 ? list[p + lane] : 8u, e1 = v1 ? list[p + 64 + lane] : 8u;
            const int P0 = (int)(e0 & 2047u), P1 = (int)(e1 & 2047u);
            // dark corners work on 255 - pixel: the same maximum of minima
            const uint32_t flip = ((e0 >> 11) & 1u ? 0u : 0xFFu) | ((e1 >> 11) & 1u ? 0u : 0x00FF0000u);
            auto px = [&](int dy, int dx) -> v2s {
              const uint32_t x0 = raw[rowoff[dy + 3] + 3 + dx + P0], x1 = raw[rowoff[dy + 3] + 3 + dx + P1];
              return as_v2s((x0 | (x1 << 16)) ^ flip);
            };
            const v2s d[16] = {px(3, 0),  px(3, 1),   px(2, 2),   px(1, 3),   px(0, 3),  px(-1, 3), px(-2, 2), px(-3, 1),
                               px(-3, 0), px(-3, -1), px(-2, -2), px(-1, -3), px(0, -3), px(1, -3), px(2, -2), px(3, -1)};
            const v2s v = px(0, 0);
            const v2s one = {1, 1};
            const uint32_t sc = as_u32(arc_max_min(d) - v - one);  // cornerScore: max(A - v, v - Bm) - 1
            if (v0) s_dn[P0] = (uint8_t)(sc & 0xFFu);
            if (v1) s_dn[P1] = (uint8_t)((sc >> 16) & 0xFFu);
          }
          wave_sync();
        }
      }
    }

    // ---- sparse 2: strict 3x3 suppression of row y - 1, emission in raster order
    const int ye = y - 1;
    if (ye >= ys0) {  // wave-uniform (ye < ys_end by the loop bound)
      const uint32_t N = c_prev;
      const int cnt = __popc(N);
      const int incl = wave_incl_scan(cnt);
      const int total = __builtin_amdgcn_readlane(incl, 63);
      if (total > 0) {  // wave-uniform
        uint32_t m = N;
        int idx = incl - cnt;
        for (int R = 0; R < total; R += LIST_CAP) {
          while (true) {
            const bool act = m != 0u && idx < R + LIST_CAP;
            if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
            if (act) {
              const int b = __builtin_ctz(m);
              list[idx - R] = (uint16_t)((lane << 5) | b);
              m &= m - 1u;
              ++idx;
            }
          }
          wave_sync();
          const int nround = min(LIST_CAP, total - R);
          for (int p = 0; p < nround; p += 64) {
            const bool v0 = p + lane < nround;
            const int P = v0 ? (int)list[p + lane] : 1;
            const uint32_t own = s_mid[P];
            const uint32_t n0 = s_up[P - 1], n1 = s_up[P], n2 = s_up[P + 1], n3 = s_mid[P - 1], n4 = s_mid[P + 1];
            const uint32_t n5 = s_dn[P - 1], n6 = s_dn[P], n7 = s_dn[P + 1];
            const uint32_t nb = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
            if (v0 && own > nb) atomicOr(&kmask[P >> 5], 1u << (P & 31));
          }
          wave_sync();
        }
        const uint32_t K = kmask[lane];
        kmask[lane] = 0u;
        const int kc = __popc(K);
        const int kincl = wave_incl_scan(kc);
        const int before_group = __shfl(kincl - kc, lane_gfirst, 64);
        const int gtot = __shfl(kincl, lane_glast, 64) - before_group;
        int pos = gcount + (kincl - kc) - before_group;
        uint32_t km = K;
        const uint32_t yx = ((uint32_t)ye << 12) + (uint32_t)c0;
        while (true) {
          if (__builtin_amdgcn_ballot_w64(km != 0u) == 0ull) break;
          if (km != 0u) {
            const int b = __builtin_ctz(km);
            const uint32_t sc = s_mid[lane * 32 + b];
            if (pos < seg_cap) seg[pos] = (sc << 24) + yx + (uint32_t)b;
            ++pos;
            km &= km - 1u;
          }
        }
        gcount += gtot;
      }
      // row starts of the cell; the cell's total in its last slot; then the next cell of the chain
      const int r = ye - cell_row0;
      const int cell_rows = min(SR, L.y_hi - cell_row0);
      if (gfirst) rs[r + 1] = (uint16_t)min(gcount, seg_cap);
      if (r + 1 == cell_rows) {  // wave-uniform
        if (gfirst && cell_rows < SR) rs[SR] = (uint16_t)min(gcount, seg_cap);
        gcount = 0;
        cell_row0 += SR;
        seg += (size_t)L.nbands * seg_cap;
        rs += (size_t)L.nbands * VSF_FAST_RS_STRIDE;
        if (gfirst && cell_row0 < ys_end) rs[0] = 0;
      }
    }
    c_prev = (y >= ys0 && y < ys_end) ? (C & emask) : 0u;
  }
}

}  // namespace

// Items for the geometry `levels` (host): chains of up to `chain` strips per (level, band), longest first.
// Returns false when the geometry does not fit this kernel (a band starting left of column 4, or too many lanes).
bool vsf_fast_bits_items(const VsfLevel* levels, int nlevels, int chain, std::vector<uint2>* items) {
  struct Item {
    uint2 v;
    int rows;
  };
  std::vector<Item> all;
  for (int l = 0; l < nlevels; l++) {
    const VsfLevel& L = levels[l];
    if (L.nbands <= 0 || L.nstrips <= 0) continue;
    if (L.fast_a0 < 4 || L.nbands > 255 || L.nstrips > 0xFFFF || l > 255) return false;
    for (int b = 0; b < L.nbands; b++) {
      const int bx0 = L.fast_a0 + VSF_FAST_BAND_COLS * b;
      const int bw = std::min(VSF_FAST_BAND_COLS, L.x_hi - bx0);
      const int nl = (bw + 8 + 31) / 32;
      if (nl < 1 || nl > 8) return false;
      for (int s = 0; s < L.nstrips; s += chain) {
        const int ns = std::min(chain, L.nstrips - s);
        Item it;
        it.v.x = ((uint32_t)l << 24) | ((uint32_t)b << 16) | (uint32_t)s;
        it.v.y = ((uint32_t)ns << 8) | (uint32_t)nl;
        it.rows = std::min(ns * VSF_FAST_STRIP_ROWS, L.y_hi - (L.y_lo + s * VSF_FAST_STRIP_ROWS));
        all.push_back(it);
      }
    }
  }
  std::stable_sort(all.begin(), all.end(), [](const Item& x, const Item& y) { return x.rows > y.rows; });
  items->clear();
  for (const Item& it : all) items->push_back(it.v);
  return true;
}

void vsf_launch_fast_bits(const VsfDev& d, const VsfGeom& g, const VsfImages& im, const uint2* d_items, int nitems,
                          int threshold, hipStream_t s) {
  if (nitems <= 0 || im.n <= 0) return;
  FastBitsArgs a;
  a.levels = d.levels;
  a.items = d_items;
  a.nunits = g.nunits;
  a.img0 = im.base;
  a.img0_stride = im.image_stride;
  a.img0_pitch = (int)im.row_stride;
  a.pyr = d.pyr;
  a.pyr_bytes = g.pyr_bytes;
  a.cand = d.cand;
  a.cand_entries = g.cand_entries;
  a.rowstart = d.rowstart;
  a.threshold = threshold;
  a.nimages = im.n;
  // (dynamic LDS beyond the default limit needs the attribute -- only builds with several waves per workgroup get there; set
  // per launch: it is a property of the function on the CURRENT device, and contexts of several devices may share a process)
  if (FB_WAVES * FB_LDS_PER_WAVE > 64 * 1024)
    vsf_note(hipFuncSetAttribute(reinterpret_cast<const void*>(fast_bits_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 FB_WAVES * FB_LDS_PER_WAVE));
  hipLaunchKernelGGL(fast_bits_kernel, dim3(nitems, ((im.n + 7) / 8 + FB_WAVES - 1) / FB_WAVES), dim3(64 * FB_WAVES), FB_WAVES * FB_LDS_PER_WAVE, s, a);
}
