// vsf_geometry.hip -- everything OpenCV 3.2 computes once per image size on the host (layer sizes, layer scales, per-level
// feature budgets, resize coefficient tables, the fixed-point Gaussian kernel; features2d/orb.cpp, imgproc/imgwarp.cpp,
// imgproc/smooth.cpp), plus the work-unit lists and constant matrix-core operands of the kernels.  Built at vsf_create.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "vsf_ctx.h"

using namespace vsfi;

namespace {

// ---- OpenCV scalar helpers (core/fast_math.hpp) ----
inline int cvRoundD(double v) { return (int)std::nearbyint(v); }
inline int cvRoundF(float v) { return (int)std::nearbyintf(v); }
inline int cvFloorD(double v) {
  const int i = cvRoundD(v);
  return i - ((float)(v - i) < 0);
}
inline int cvCeilD(double v) {
  const int i = cvRoundD(v);
  return i + ((float)(i - v) < 0);
}
inline int16_t satShort(float v) { return (int16_t)std::min(std::max(cvRoundF(v), -32768), 32767); }

}  // namespace

namespace vsfi {

// cv::resize(INTER_LINEAR, 8u) coefficient tables for one level (source sw x sh -> dw x dh).
void build_taps(int sw, int sh, int dw, int dh, std::vector<VsfTap>* xt, std::vector<VsfTap>* yt) {
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloorD(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= sw - 1) fx = 0, sx = sw - 1;  // (also the dx >= xmax single-tap case: weight 2048 on S[sx])
    VsfTap t;
    t.i0 = (uint16_t)sx;
    t.i1 = (uint16_t)std::min(sx + 1, sw - 1);
    t.c0 = satShort((1.f - fx) * 2048);
    t.c1 = satShort(fx * 2048);
    xt->push_back(t);
  }
  while (xt->size() % 4) xt->push_back(VsfTap{0, 0, 0, 0});
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = cvFloorD(fy);
    fy -= sy;
    VsfTap t;
    t.i0 = (uint16_t)std::min(std::max(sy, 0), sh - 1);
    t.i1 = (uint16_t)std::min(std::max(sy + 1, 0), sh - 1);
    t.c0 = satShort((1.f - fy) * 2048);
    t.c1 = satShort(fy * 2048);
    yt->push_back(t);
  }
}

// ORB umax table (features2d/orb.cpp computeKeyPoints) for the sanity check of the device constant.
std::vector<int> orb_umax(int patch_size) {
  const int half = patch_size / 2;
  std::vector<int> umax(half + 2, 0);
  int v, v0;
  const int vmax = cvFloorD(half * std::sqrt(2.f) / 2 + 1);
  const int vmin = cvCeilD(half * std::sqrt(2.f) / 2);
  for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt((double)half * half - v * v));
  for (v = half, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
  return umax;
}

inline int reflect101_host(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - 2 - p;
  return p < 0 ? 0 : (p >= len ? len - 1 : p);
}

uint16_t f16_bits_of_small_int(int v) {  // exact binary16 encoding of an integer 0 <= v < 2048 * 32
  if (v == 0) return 0;
  int e = 0;
  while ((v >> (e + 1)) != 0) e++;           // v = 1.m * 2^e
  const int mant = e <= 10 ? (v << (10 - e)) & 0x3FF : (v >> (e - 10)) & 0x3FF;  // (exact: callers pass <= 11 significant bits)
  return (uint16_t)(((e + 15) << 10) | mant);
}

// Constant operands of the matrix-core blur (k_blur.hip).
//  pass 1, per level and 64-column band: four 32 x 32 int8 bands of the row filter, {L0, R0, L1, R1}: tile t = 2 band + j
//  takes operand Lj with the image columns [32 t - 16, 32 t + 16) and Rj with [32 t + 16, 32 t + 48); lane (n, h) holds, in
//  byte s, the weight of image column (operand's first column) + 16 h + s for output column 32 t + n -- BORDER_REFLECT_101
//  folded in: a reflected column's tap is added to the weight of the column it reflects onto.
//  pass 2, once: four 16 x 32 f16 operands {lo k-step 0, lo k-step 1, hi k-step 0, hi k-step 1}: lane (n, h) element j is
//  the tap of loaded row 16 s + 8 (j >> 2) + 4 h + (j & 3) for output row n + 3 of the 32 loaded rows (n < 26), times 256
//  for the high byte of the row sums.
void build_blur_mma_tables(Geometry* G) {
  int k4[4];
  gaussian_taps(k4);
  const int k[7] = {k4[0], k4[1], k4[2], k4[3], k4[2], k4[1], k4[0]};
  G->blur_bias = 128 * (k[0] + k[1] + k[2] + k[3] + k[4] + k[5] + k[6]);
  G->blur_tcol.clear();
  G->blur_mma_units.clear();
  G->blur_mma_units_small.clear();
  for (size_t l = 0; l < G->levels.size(); l++) {
    VsfLevel& L = G->levels[l];
    L.blur_tcol = (uint32_t)(G->blur_tcol.size() / 64);
    const int nbands = (L.w + 63) / 64, npairs = (nbands + 1) / 2;
    for (int b = 0; b < 2 * npairs; b++)  // (padded to whole band pairs: a padding band's weights are zero)
      for (int op = 0; op < 4; op++) {
        const int tile = 2 * b + (op >> 1);
        const int first = 32 * tile - 16 + 32 * (op & 1);
        for (int lane = 0; lane < 64; lane++) {
          const int n = lane & 31, h = lane >> 5, x = 32 * tile + n;
          int8_t wgt[16] = {0};
          if (x < L.w)
            for (int j = 0; j < 7; j++) {
              const int c = reflect101_host(x + j - 3, L.w) - (first + 16 * h);
              if (c >= 0 && c < 16) wgt[c] = (int8_t)(wgt[c] + k[j]);
            }
          uint4 v;
          memcpy(&v, wgt, 16);
          G->blur_tcol.push_back(v);
        }
      }
    // units = one workgroup each: (band pair, strip of double steps); a workgroup's four waves are 2 bands x 2 steps.
    // Two lists: long strips for batches that fill the chip anyway (a workgroup's first block is pure latency: 16 double
    // steps per unit 0.93 ms per 512 images, 4: 1.01, 2: 1.29), short ones for a frame or two (parallelism).
    const int nsteps = (L.h + VSF_BLUR_MMA_ROWS - 1) / VSF_BLUR_MMA_ROWS, ndsteps = (nsteps + 1) / 2;
    for (int pass = 0; pass < 2; pass++) {
      const int per_unit = pass == 0 ? VSF_BLUR_MMA_STEPS : VSF_BLUR_MMA_STEPS_SMALL;
      std::vector<uint32_t>& units = pass == 0 ? G->blur_mma_units : G->blur_mma_units_small;
      const int nstrips = (ndsteps + per_unit - 1) / per_unit;
      for (int st = 0; st < nstrips; st++) {
        const int s0 = (int)((long long)ndsteps * st / nstrips), s1 = (int)((long long)ndsteps * (st + 1) / nstrips);
        for (int b = 0; b < npairs; b++)
          units.push_back(((uint32_t)l << 24) | ((uint32_t)b << 16) | ((uint32_t)s0 << 8) | (uint32_t)(s1 - s0));
      }
    }
  }
  G->blur_tv.clear();
  for (int op = 0; op < 4; op++)
    for (int lane = 0; lane < 64; lane++) {
      const int n = lane & 31, h = lane >> 5, s = op & 1;
      uint16_t e[8];
      for (int j = 0; j < 8; j++) {
        const int row = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3), d = row - n;
        const int tap = (n < VSF_BLUR_MMA_ROWS && d >= 0 && d < 7) ? k[d] : 0;
        e[j] = f16_bits_of_small_int(op >= 2 ? tap * 256 : tap);
      }
      uint4 v;
      memcpy(&v, e, 16);
      G->blur_tv.push_back(v);
    }
}

// orb == true: the 50-level ORB pyramid with edge-threshold border; false: one full-resolution level with the
// 3-pixel FAST rim (FastFeatureDetector::detect).
bool build_geometry(const vsf_params& p, bool orb, bool nms, Geometry* out) {
  Geometry& G = *out;
  const int nlevels = orb ? p.nlevels : 1;
  const int border = orb ? std::max(p.edge_threshold, 3) : 3;
  G.levels.assign(nlevels, VsfLevel{});
  G.g.nlevels = nlevels;
  G.g.width = p.width;
  G.g.height = p.height;
  const double scale_factor = (double)p.scale_factor;
  // per-level budget
  std::vector<int> nfeat(nlevels, 0);
  if (orb) {
    const float factor = (float)(1.0 / scale_factor);
    float nd = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
      nfeat[l] = cvRoundF(nd);
      sum += nfeat[l];
      nd *= factor;
    }
    nfeat[nlevels - 1] = std::max(p.nfeatures - sum, 0);
  }
  uint32_t offset = 0;
  uint64_t pixels = 0;
  int max_w = 0;
  for (int l = 0; l < nlevels; l++) {
    VsfLevel& L = G.levels[l];
    L.scale = orb ? (float)std::pow(scale_factor, (double)(l - p.first_level)) : 1.f;
    L.w = cvRoundF(p.width / L.scale);
    L.h = cvRoundF(p.height / L.scale);
    if (L.w < 8 || L.h < 1 || L.w > 4095 || L.h > 4095) return false;  // (blur border window needs w >= 8)
    L.pitch = align_up(L.w, 64);
    L.offset = offset;
    offset += (uint32_t)align_up(L.pitch * align_up(L.h, 8), 256);  // (rows padded for the tiled blurred copy)
    L.nfeatures = nfeat[l];
    pixels += (uint64_t)L.w * L.h;
    max_w = std::max(max_w, L.w);
    if (L.w <= 2 * border || L.h <= 2 * border) {
      L.x_lo = L.x_hi = L.y_lo = L.y_hi = border;  // runByImageBorder clears such a level
    } else {
      L.x_lo = border;
      L.x_hi = L.w - border;
      L.y_lo = border;
      L.y_hi = L.h - border;
    }
    L.blur_vec_end = p.blur_sse2 ? (L.w - L.w % 4) : 0;
  }
  G.g.pyr_bytes = offset;
  G.g.pyramid_pixels = pixels;
  // FAST units: (240-column band) x (32-row strip) of the keypoint rectangle, one wave each (k_fast.hip)
  uint32_t cand = 0;
  int kp_off = 0, ncells = 0;
  std::vector<uint32_t> half_items;
  for (int l = 0; l < nlevels; l++) {
    VsfLevel& L = G.levels[l];
    const int vw = L.x_hi - L.x_lo, vh = L.y_hi - L.y_lo;
    L.fast_a0 = L.x_lo & ~3;
    L.nbands = vw > 0 ? (L.x_hi - L.fast_a0 + VSF_FAST_BAND_COLS - 1) / VSF_FAST_BAND_COLS : 0;
    L.nstrips = vh > 0 && vw > 0 ? (vh + VSF_FAST_STRIP_ROWS - 1) / VSF_FAST_STRIP_ROWS : 0;
    if (L.nbands > 255 || L.nstrips > 32767) return false;
    L.unit0 = ncells;
    ncells += L.nstrips * L.nbands;
    // work items: one wave per cell, except that a narrow last band is walked two strips per wave (k_fast.hip)
    const int last_w = L.nbands > 0 ? L.x_hi - (L.fast_a0 + VSF_FAST_BAND_COLS * (L.nbands - 1)) : 0;
    const bool half_last = L.nbands > 0 && L.nstrips >= 2 && last_w <= VSF_FAST_HALF_COLS;
    for (int s = 0; s < L.nstrips; s++)
      for (int b = 0; b < L.nbands; b++) {
        const uint32_t item = ((uint32_t)l << 24) | ((uint32_t)b << 16) | (uint32_t)s;
        if (half_last && b == L.nbands - 1) {
          if ((s & 1) == 0) half_items.push_back(item);
        } else {
          G.units.push_back(item);
        }
      }
    // Strict 8-neighbour NMS leaves at most one keypoint per 2x2 block, so a segment of that size cannot overflow.
    const int bw = std::min(VSF_FAST_BAND_COLS, std::max(vw, 1)), bh = std::min(VSF_FAST_STRIP_ROWS, std::max(vh, 1));
    L.seg_cap = nms ? ((bw + 1) / 2) * ((bh + 1) / 2) : bw * bh;
    L.seg_cap = std::max(L.seg_cap, 1);
    L.cand_offset = cand;
    cand += (uint32_t)L.seg_cap * (uint32_t)(L.nstrips * L.nbands);
    L.kp_offset = kp_off;
    L.kp_cap = 2 * L.nfeatures + 64;
    kp_off += L.kp_cap;
  }
  G.g.cand_entries = std::max(cand, 1u);
  G.g.nunits = ncells;
  G.g.nwork_full = (int)G.units.size();
  G.g.nwork_half = (int)half_items.size();
  G.units.insert(G.units.end(), half_items.begin(), half_items.end());
  if (G.units.empty()) G.units.push_back(0);
  G.g.lvlkp_entries = std::max(kp_off, 1);
  // resize coefficient tables (host only: the kernel evaluates the same arithmetic in place; built here to check
  // that a lane's eight x taps fit the 8-byte source window it loads)
  if (orb) {
    for (int l = 1; l < nlevels; l++) {
      VsfLevel& L = G.levels[l];
      const VsfLevel& P = G.levels[l - 1];
      L.xtab = (uint32_t)G.xt.size();
      L.ytab = (uint32_t)G.yt.size();
      build_taps(P.w, P.h, L.w, L.h, &G.xt, &G.yt);
      {
        const double sx = 1. / ((double)L.w / P.w), sy = 1. / ((double)L.h / P.h);
        memcpy(L.rscale_x, &sx, 8);
        memcpy(L.rscale_y, &sy, 8);
      }
      // resize_march_kernel reads one 8-byte source window per lane (4 output pixels): all eight taps must fit.
      if (P.w < 8) return false;
      for (int x4 = 0; x4 < L.w; x4 += 4) {
        const int base = std::min((int)G.xt[L.xtab + x4].i0, P.w - 8);
        for (int j = 0; j < 4 && x4 + j < L.w; j++) {
          const VsfTap& t = G.xt[L.xtab + x4 + j];
          if (t.i0 < base || t.i1 - base > 7) return false;
        }
      }
      // resize_strip_kernel<R> keeps the horizontal sums of R + 2 consecutive source rows (from the first output
      // row's upper tap on) and takes output row r's taps from entries r + d, r + d + 1 with d in {0, 1}: true when
      // the scale is below 1 + 1 / (R - 1) (all ORB levels at 1.04 qualify); checked here on the exact tables.
      L.resize_rows = 0;
      for (int R : {16, 8, 4}) {
        bool ok = true;
        for (int ys = 0; ys < L.h && ok; ys += R) {
          const int f = G.yt[L.ytab + ys].i0;
          for (int r = 0; r < R && ys + r < L.h && ok; r++) {
            const VsfTap& t = G.yt[L.ytab + ys + r];
            const int dlt = (int)t.i0 - f - r;
            ok = (dlt == 0 || dlt == 1) && (int)t.i1 == std::min((int)t.i0 + 1, P.h - 1);
          }
        }
        if (ok) {
          L.resize_rows = R;
          break;
        }
      }
      // the same property for 8-row strips that start at any row (pyramid_slab_kernel cuts levels where its slabs fall)
      L.resize_any8 = 1;
      for (int ys = 0; ys < L.h && L.resize_any8; ys++) {
        const int f = G.yt[L.ytab + ys].i0;
        for (int r = 0; r < 8 && ys + r < L.h && L.resize_any8; r++) {
          const VsfTap& t = G.yt[L.ytab + ys + r];
          const int dlt = (int)t.i0 - f - r;
          if (!((dlt == 0 || dlt == 1) && (int)t.i1 == std::min((int)t.i0 + 1, P.h - 1))) L.resize_any8 = 0;
        }
      }
    }
  }
  if (G.xt.empty()) G.xt.push_back(VsfTap{0, 0, 0, 0});
  if (G.yt.empty()) G.yt.push_back(VsfTap{0, 0, 0, 0});
  if (orb) {
    // matrix-core blur: taps must be int8, the row sums 16 bit, bands / steps fit the unit word
    int k4[4];
    gaussian_taps(k4);
    const int ksum = 2 * (k4[0] + k4[1] + k4[2]) + k4[3];
    if (ksum > 257 || k4[3] > 127 || G.levels[0].w > 64 * 255 || G.levels[0].h > VSF_BLUR_MMA_ROWS * 255) return false;
    build_blur_mma_tables(&G);
  }
  if (G.blur_mma_units.empty()) G.blur_mma_units.push_back(0);
  if (G.blur_mma_units_small.empty()) G.blur_mma_units_small.push_back(0);
  if (G.blur_tcol.empty()) G.blur_tcol.push_back(make_uint4(0, 0, 0, 0));
  if (G.blur_tv.empty()) G.blur_tv.push_back(make_uint4(0, 0, 0, 0));
  return true;
}

// ICAngles disc (patch 31) as byte weights for k_describe.hip: for byte phase s = (x0 - 15) & 3 the item (row r,
// dword j) covers u = 4j + b - s - 15, b = 0..3, on row v = r - 15; .x holds u + 16 and .y holds 1 for the bytes
// inside the disc (|u| <= umax[|v|]), 0 elsewhere.
std::vector<uint2> build_ic_table() {
  const std::vector<int> um = orb_umax(31);
  std::vector<uint2> t(4 * VSF_IC_ITEMS, make_uint2(0, 0));
  for (int s = 0; s < 4; s++)
    for (int item = 0; item < 31 * 9; item++) {
      const int r = item / 9, j = item % 9, v = r - 15, d = um[std::abs(v)];
      uint32_t wx = 0, wm = 0;
      for (int b = 0; b < 4; b++) {
        const int u = 4 * j + b - s - 15;
        if (std::abs(u) <= d) {
          wx |= (uint32_t)(u + 16) << (8 * b);
          wm |= 1u << (8 * b);
        }
      }
      t[(size_t)s * VSF_IC_ITEMS + item] = make_uint2(wx, wm);
    }
  return t;
}

// getGaussianKernel(7, 2, CV_32F) scaled by 256 and rounded (createSeparableLinearFilter, 8u smooth kernels).
void gaussian_taps(int k[4]) {
  const int n = 7;
  const double sigma = 2.0, scale2x = -0.5 / (sigma * sigma);
  float cf[7];
  double sum = 0;
  for (int i = 0; i < n; i++) {
    const double x = i - (n - 1) * 0.5;
    cf[i] = (float)std::exp(scale2x * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < 4; i++) k[i] = cvRoundD((double)(float)(cf[i] * sum) * 256.0);
}

}  // namespace vsfi
