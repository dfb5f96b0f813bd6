// vsf_api.hip -- host side of the C ABI (include/vsf.h): context, pyramid geometry and coefficient tables,
// HBM buffers, and the entry points that replace Frontend::ExtractFeatures (slam_frontend.cc:266-280) and
// Frontend::GetMatches (slam_frontend.cc:521-538).  All arithmetic that OpenCV 3.2 does once per image size
// on the host (layer sizes, layer scales, per-level feature budgets, resize coefficient tables, the fixed-point
// Gaussian kernel; features2d/orb.cpp, imgproc/imgwarp.cpp, imgproc/smooth.cpp) is done here at vsf_create.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "vsf_internal.h"

thread_local int vsf_tls_hip_error = 0;

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
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

struct Geometry {
  VsfGeom g{};
  std::vector<VsfLevel> levels;
  std::vector<uint32_t> units;
  std::vector<uint2> bits_items;  // k_fastbits.hip work items (empty: the geometry does not fit that kernel)
  std::vector<VsfTap> xt, yt;
  std::vector<uint32_t> blur_tiles;
  // matrix-core blur (k_blur.hip blur_mma_kernel): work units and constant MFMA operands
  std::vector<uint32_t> blur_mma_units, blur_mma_units_small;  // long strips (batches) / short strips (a frame or two)
  std::vector<uint4> blur_tcol, blur_tv;
  int blur_bias = 0;
};

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

void gaussian_taps(int k[4]);

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
  if (!(orb && nms && vsf_fast_bits_items(G.levels.data(), nlevels, 4, &G.bits_items))) G.bits_items.clear();
  G.g.lvlkp_entries = std::max(kp_off, 1);
  // resize coefficient tables (host only: the kernel evaluates the same arithmetic in place; built here to check
  // that a lane's eight x taps fit the 8-byte source window it loads) + blur tiles (ORB only)
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
    for (int l = 0; l < nlevels; l++) {
      const VsfLevel& L = G.levels[l];
      // blur work units: (level, 248-column band, 64-row strip), one wave each -- except that a narrow last band
      // (<= 120 columns) is walked two strips per wave (k_blur.hip: bit 15)
      const int nb = (L.w + VSF_BLUR_BAND_COLS - 1) / VSF_BLUR_BAND_COLS;
      const int ns = (L.h + VSF_BLUR_STRIP_ROWS - 1) / VSF_BLUR_STRIP_ROWS;
      const bool half_last = ns >= 2 && L.w - VSF_BLUR_BAND_COLS * (nb - 1) <= 120;
      for (int st = 0; st < ns; st++)
        for (int b = 0; b < nb; b++) {
          if (half_last && b == nb - 1) {
            if ((st & 1) == 0) G.blur_tiles.push_back(((uint32_t)l << 24) | ((uint32_t)b << 16) | 0x8000u | (uint32_t)st);
          } else {
            G.blur_tiles.push_back(((uint32_t)l << 24) | ((uint32_t)b << 16) | (uint32_t)st);
          }
        }
    }
  }
  if (G.xt.empty()) G.xt.push_back(VsfTap{0, 0, 0, 0});
  if (G.yt.empty()) G.yt.push_back(VsfTap{0, 0, 0, 0});
  if (G.blur_tiles.empty()) G.blur_tiles.push_back(0);
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

struct DevSet {  // device copies of one Geometry + its work buffers
  VsfDev d{};
  VsfLevel* levels = nullptr;
  uint32_t* units = nullptr;
  uint2* bits_items = nullptr;
  int n_bits_items = 0;
  uint32_t* blur_tiles = nullptr;
  uint32_t* blur_mma_units = nullptr;
  uint32_t* blur_mma_units_small = nullptr;
  uint4* blur_tcol = nullptr;
  uint4* blur_tv = nullptr;
  uint2* ic_table = nullptr;
  bool ready = false;
};

}  // namespace

struct vsf_ctx {
  vsf_params p{};
  int device = 0;
  int n_cus = 256;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // Second lane of the batched entry points: half of a batch runs on `stream`, the other half on `aux_stream`
  // (frames are independent), so latency-bound stages of one half overlap VALU-bound stages of the other.
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // the blur (matrix cores + memory) beside FAST (vector ALU) in batched calls: its own stream, forked after the pyramid
  hipStream_t blur_stream = nullptr;
  hipEvent_t ev_blur_fork = nullptr, ev_blur_done = nullptr;
  VsfSideStream side{};  // aux_stream, for the pyramid's second launch chain
  // Cross-call pipelining (vsf_set_pipeline): the pyramid of call k + 1 is built on side streams, into the other of
  // two pyramid buffers, while call k's later stages still run.
  bool pipeline = false;
  hipStream_t pipe_stream = nullptr;  // the pipelined chain's own stream when VSF_OPT_PIPE_PRIORITY asks for a priority
  int pipe_stream_priority = 0;
  uint8_t* pyr_alt = nullptr;
  int pyr_flip = 0;
  hipEvent_t ev_pyr_done = nullptr, ev_pyr_free[2] = {nullptr, nullptr}, ev_fast_done = nullptr;
  bool pyr_free_valid[2] = {false, false}, fast_done_valid = false;
  // A producer the library owns (the Bayer ingest) records this on the context's stream; a pipelined pyramid, which is
  // NOT ordered after that stream's earlier work, waits for it.
  hipEvent_t ev_ingest_done = nullptr;
  bool ingest_done_valid = false;
  // ... and ANY other producer hands over an event of its own (vsf_set_input_event): the next batched call -- its
  // pipelined pyramid included -- waits for it; one-shot.
  hipEvent_t input_event = nullptr;
  const uint8_t* last_pyr = nullptr;
  int lanes = 1;  // 1 = everything on `stream` (default), 2 = two concurrent half batches (vsf_set_lanes)
  int blur_overlap = 1;  // the blur on blur_stream beside FAST / selection (vsf_set_blur_overlap)
  int fast_resident = -1;  // vsf_set_fast_resident
  int fast_force = -1;     // vsf_tune_fast_resident only: the form of the run it is timing
  VsfTuning tuning;        // vsf_set_option
  int last_hip = 0;
  int pending_hip = 0;  // an error noted during one of THIS context's calls that returned before checking (VsfErrorScope)
  Geometry orb, fast;
  DevSet dorb, dfast;
  int gauss[4] = {0, 0, 0, 0};
  // Status words (bit 0: capacity overflow, bit 1: a JPEG stream broke off): word 0 belongs to the context's own stream
  // (batched and host-pointer calls, vsf_sync), words 1..6 to the frames that may be in flight (vsf_observe_submit) --
  // a frame's kernels run on its slot's stream beside another frame's, so each frame sets, copies and clears its own word.
  int32_t* d_status = nullptr;     // [1 + VSF_OBSERVE_MAX_SLOTS]
  uint32_t* fast_cells = nullptr;  // [2] cell counters of the resident FAST kernels (k_fast.hip)
  struct FastTune {  // resident FAST or one workgroup per four cells: what vsf_tune_fast_resident measured, per batch size
    int n = 0, choice = -1;
    hipEvent_t ev[2] = {nullptr, nullptr};
  } fast_tune;
  int32_t* h_status = nullptr;  // pinned
  // staging for the host-pointer entry points
  uint8_t* st_img = nullptr;
  size_t st_img_pitch = 0, st_img_stride = 0;
  vsf_keypoint* st_kp = nullptr;
  uint8_t* st_desc = nullptr;
  int32_t* st_counts = nullptr;
  // matcher work buffers
  int32_t* m_idx2 = nullptr;
  int32_t* m_dist2 = nullptr;
  int m_pairs = 0, m_rows = 0;
  // f1 work buffers: residuals [frames][rows], F (9 floats), matches / counts / sort keys of the temporal pairs
  float* f_residual = nullptr;
  int f_frames = 0;
  vsf_dmatch* t_matches = nullptr;
  int32_t* t_nmatches = nullptr;
  void* t_sortkeys = nullptr;
  int t_pairs = 0;
  // f2 work buffers: right->left pairs of every frame, their set indices, the pack kernel's offsets
  uint64_t* v_pairs = nullptr;
  int32_t* v_npairs = nullptr;
  int32_t* v_sets = nullptr;   // [2][v_frames]: q_set = 2f + 1, t_set = 2f
  int v_frames = 0;
  uint32_t* pk_offsets = nullptr;
  int pk_entries = 0;
  // Scratch a *_dev call has outgrown.  Such a call takes a NEW allocation (hipMalloc does not wait for the GPU) and
  // parks the old one here, because hipFree would wait for the whole device behind the caller's back; released by
  // vsf_sync / vsf_reserve / vsf_destroy, when every stream of the context is known to be idle.
  std::vector<void*> retired;
  // vsf_observe_stereo: temporal ring, per-call device scratch, pinned host staging
  struct ObserveMeta {  // pinned, device-visible: read by the kernels over PCIe (a few words per call, no copy command)
    float F[9];
    float best_percent[VSF_OBSERVE_MAX_PAIRS];
    int32_t q_set[VSF_OBSERVE_MAX_PAIRS], t_set[VSF_OBSERVE_MAX_PAIRS];
  };
  struct Observe {
    int frame_life = 0;
    uint8_t* ring = nullptr;        // [frame_life + 2][K][32]: kept frames, then the current left / right frame
    int32_t* ring_counts = nullptr; // [frame_life + 2]
    vsf_keypoint* kpf = nullptr;    // [2][K]
    vsf_dmatch* matches = nullptr;  // [slots][K] raw stereo matches
    int32_t* ints = nullptr;        // [slots] nmatches, then nfeat, npoints
    float* floats = nullptr;        // mean, thr, thr_state
    vsf_vision_feature* features = nullptr;
    uint64_t* pairs = nullptr;      // [frame_life + 1][K][2]
    int32_t* npairs = nullptr;
    // Up to three frames may be in flight (vsf_observe_submit / vsf_observe_collect; two slots with max_images >= 4,
    // three with >= 6): everything one frame's EXTRACTION writes exists once per slot -- pinned staging, per-call
    // parameters, result buffer, status word, the slot's two images of every extraction buffer, raw stereo matches.  A
    // frame runs on its slot's stream from upload to result; its TAIL (RemoveAmbigStereo ... result) first waits for the
    // previous frame's tail (an event), so the tails -- which carry the threshold and the temporal window from frame to
    // frame -- run in frame order and their buffers exist once.
    int slots = 1;
    uint8_t* h_img[VSF_OBSERVE_MAX_SLOTS] = {};       // pinned: both images at the staging pitch
    uint8_t* h_out[VSF_OBSERVE_MAX_SLOTS] = {};       // pinned, written by observe_pack_kernel
    size_t out_cap = 0;
    ObserveMeta* h_meta[VSF_OBSERVE_MAX_SLOTS] = {};
    int32_t* h_status[VSF_OBSERVE_MAX_SLOTS] = {};    // pinned copy of the status word after the frame's last kernel
    hipStream_t ex_stream[VSF_OBSERVE_MAX_SLOTS] = {};  // the stream of slot i (a one-slot context: ctx->stream)
    hipEvent_t ev_done[VSF_OBSERVE_MAX_SLOTS] = {};
    VsfSideStream side[VSF_OBSERVE_MAX_SLOTS] = {};  // (n = 0: no second pyramid chain, no shared fork / join events)
    bool done_valid[VSF_OBSERVE_MAX_SLOTS] = {};
    int64_t ticket_of[VSF_OBSERVE_MAX_SLOTS] = {-1, -1, -1, -1, -1, -1};  // submitted and not yet collected
    int64_t next_ticket = 0;
    std::vector<int> order;         // ring slots of the kept frames, oldest first
  } ob;
  // vsf_jpeg_decode_gray_batch: pinned staging + device copy of the packed headers / tables / entropy-coded segments
  // (two sets, used alternately: the host fills one while the previous call's upload / decode still use the other)
  int32_t* jp_flags = nullptr;  // [jp_flags_cap] per progressive file of a call: damaged, decode again scan after scan
  int jp_flags_cap = 0;
  uint8_t* jp_host[2] = {nullptr, nullptr};
  uint8_t* jp_dev[2] = {nullptr, nullptr};
  size_t jp_cap[2] = {0, 0};
  hipEvent_t jp_copied[2] = {nullptr, nullptr};  // the last upload out of jp_host[i] has finished
  int jp_flip = 0;
  uint8_t* png_filtered = nullptr;  // PNG: the inflated scanlines of a batch
  size_t png_filtered_cap = 0;
  int32_t* png_file_status = nullptr;
  int png_file_status_cap = 0;
  uint8_t* jp_clean = nullptr;   // parallel decode: the de-stuffed streams (layout of the upload's stream part)
  size_t jp_clean_cap = 0;
  int16_t* jp_coef = nullptr;    // ... and the luminance coefficients of the batch
  size_t jp_coef_cap = 0;
  uint8_t* mh_desc = nullptr;  // host-API descriptor staging: 2 sets
  int32_t* mh_counts = nullptr;
  vsf_dmatch* mh_matches = nullptr;
  int32_t* mh_nmatches = nullptr;
  int mh_rows = 0;
  // vsf_get_matches_multi staging: sets x rows descriptors, per-set counts / set indices / matches
  uint8_t* mm_desc = nullptr;
  int32_t* mm_counts = nullptr;  // [sets + 1] counts, then [sets] q_set, [sets] t_set
  vsf_dmatch* mm_matches = nullptr;
  int32_t* mm_nmatches = nullptr;
  int mm_sets = 0, mm_rows = 0;
  VsfImages last_images{};
  bool last_valid = false;
  bool fast_nms = true;  // NMS mode the standalone-FAST geometry was built for
  // per-stage hipEvent profiling
  bool prof_on = false;
  std::vector<hipEvent_t> ev_pool;  // pairs
  std::vector<int> ev_stage;        // stage of pair i
  std::vector<int> ev_launches;
  size_t ev_used = 0;               // pairs in flight
  double prof_ms[VSF_STAGE_COUNT] = {0};
  int64_t prof_launches[VSF_STAGE_COUNT] = {0};
};

namespace {

#define VSF_HIP(call)                     \
  do {                                    \
    hipError_t e_ = (call);               \
    if (e_ != hipSuccess) {               \
      ctx->last_hip = (int)e_;            \
      return VSF_ERR_HIP;                 \
    }                                     \
  } while (0)
// End of an entry point that launched: a failed launch (hipGetLastError) or anything a launcher / stream helper noted
// (vsf_note: event records and waits, memsets) becomes this call's VSF_ERR_HIP.
#define VSF_STICKY()                                               \
  do {                                                             \
    hipError_t e_ = hipGetLastError();                             \
    if (e_ == hipSuccess) e_ = (hipError_t)vsf_tls_hip_error;      \
    if (e_ == hipSuccess) e_ = (hipError_t)ctx->pending_hip;       \
    vsf_tls_hip_error = 0;                                         \
    ctx->pending_hip = 0;                                          \
    if (e_ != hipSuccess) {                                        \
      ctx->last_hip = (int)e_;                                     \
      return VSF_ERR_HIP;                                          \
    }                                                              \
  } while (0)

template <class T>
hipError_t upload(T** dst, const std::vector<T>& v) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), v.size() * sizeof(T));
  if (e != hipSuccess) return e;
  return hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

vsf_status alloc_devset(vsf_ctx* ctx, const Geometry& G, DevSet* ds, bool orb, int n_images) {
  VSF_HIP(upload(&ds->levels, G.levels));
  VSF_HIP(upload(&ds->units, G.units));
  if (!G.bits_items.empty()) {
    VSF_HIP(upload(&ds->bits_items, G.bits_items));
    ds->n_bits_items = (int)G.bits_items.size();
  }
  VSF_HIP(upload(&ds->blur_tiles, G.blur_tiles));
  VSF_HIP(upload(&ds->blur_mma_units, G.blur_mma_units));
  VSF_HIP(upload(&ds->blur_mma_units_small, G.blur_mma_units_small));
  VSF_HIP(upload(&ds->blur_tcol, G.blur_tcol));
  VSF_HIP(upload(&ds->blur_tv, G.blur_tv));
  VSF_HIP(upload(&ds->ic_table, build_ic_table()));
  VsfDev& d = ds->d;
  d.ic_table = ds->ic_table;
  d.levels = ds->levels;
  d.units = ds->units;
  d.bits_items = ds->bits_items;
  d.n_bits_items = ds->n_bits_items;
  const size_t n = (size_t)n_images;
  if (orb) {
    VSF_HIP(hipMalloc((void**)&d.pyr, n * G.g.pyr_bytes));
    VSF_HIP(hipMalloc((void**)&d.blur, n * G.g.pyr_bytes));
    VSF_HIP(hipMalloc((void**)&d.scratch, n * 6 * G.g.cand_entries * sizeof(uint32_t)));
    VSF_HIP(hipMalloc((void**)&d.lvlkp, n * G.g.lvlkp_entries * sizeof(VsfLevelKp)));
    VSF_HIP(hipMalloc((void**)&d.lvl_count, n * G.g.nlevels * sizeof(int32_t)));
    VSF_HIP(hipMemset(d.lvl_count, 0, n * G.g.nlevels * sizeof(int32_t)));
  }
  VSF_HIP(hipMalloc((void**)&d.cand, n * G.g.cand_entries * sizeof(uint32_t)));
  const size_t rs_bytes = n * (size_t)std::max(G.g.nunits, 1) * VSF_FAST_RS_STRIDE * sizeof(uint16_t);
  VSF_HIP(hipMalloc((void**)&d.rowstart, rs_bytes));
  VSF_HIP(hipMemset(d.rowstart, 0, rs_bytes));
  d.status = ctx->d_status;
  d.tune = &ctx->tuning;
  ds->ready = true;
  return VSF_OK;
}

void free_devset(DevSet* ds) {
  hipFree(ds->levels);
  hipFree(ds->units);
  hipFree(ds->bits_items);
  hipFree(ds->blur_tiles);
  hipFree(ds->blur_mma_units);
  hipFree(ds->blur_mma_units_small);
  hipFree(ds->blur_tcol);
  hipFree(ds->blur_tv);
  hipFree(ds->ic_table);
  hipFree(ds->d.pyr);
  hipFree(ds->d.blur);
  hipFree(ds->d.scratch);
  hipFree(ds->d.lvlkp);
  hipFree(ds->d.lvl_count);
  hipFree(ds->d.cand);
  hipFree(ds->d.rowstart);
  *ds = DevSet();
}

// ---- scratch that follows the batch size of the *_dev calls ----
// Sized at vsf_create for max_images / 2 frames and as many pairs, or by vsf_reserve.  A call that needs more never waits
// for the GPU: grow_scratch() allocates anew and retires the old buffer (kernels already queued keep using it).
template <class T>
vsf_status grow_scratch(vsf_ctx* ctx, T*& ptr, size_t bytes) {
  void* fresh = nullptr;
  VSF_HIP(hipMalloc(&fresh, std::max<size_t>(bytes, 16)));
  if (ptr) ctx->retired.push_back(static_cast<void*>(ptr));
  ptr = static_cast<T*>(fresh);
  return VSF_OK;
}

void free_retired(vsf_ctx* ctx) {  // (callers have waited for every stream of the context)
  for (void* p : ctx->retired) hipFree(p);
  ctx->retired.clear();
}

vsf_status ensure_match_buffers(vsf_ctx* ctx, int pairs, int rows) {
  if (pairs <= ctx->m_pairs && rows <= ctx->m_rows) return VSF_OK;
  pairs = std::max(pairs, ctx->m_pairs);
  rows = std::max(rows, ctx->m_rows);
  vsf_status st = grow_scratch(ctx, ctx->m_idx2, (size_t)pairs * rows * 2 * sizeof(int32_t));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->m_dist2, (size_t)pairs * rows * 2 * sizeof(int32_t));
  if (st != VSF_OK) return st;
  ctx->m_pairs = pairs;
  ctx->m_rows = rows;
  return VSF_OK;
}

vsf_status ensure_match_host_staging(vsf_ctx* ctx, int rows) {  // (host-pointer, synchronous entry points only)
  if (rows <= ctx->mh_rows) return VSF_OK;
  vsf_status st = grow_scratch(ctx, ctx->mh_desc, (size_t)2 * rows * VSF_DESC_BYTES);
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->mh_matches, (size_t)rows * sizeof(vsf_dmatch));
  if (st != VSF_OK) return st;
  if (!ctx->mh_counts) VSF_HIP(hipMalloc((void**)&ctx->mh_counts, 2 * sizeof(int32_t)));
  if (!ctx->mh_nmatches) VSF_HIP(hipMalloc((void**)&ctx->mh_nmatches, sizeof(int32_t)));
  ctx->mh_rows = rows;
  return VSF_OK;
}

vsf_status ensure_residual_buffers(vsf_ctx* ctx, int n_frames) {
  const size_t K = (size_t)ctx->p.max_keypoints;
  if (n_frames > ctx->f_frames) {
    vsf_status st = grow_scratch(ctx, ctx->f_residual, (size_t)n_frames * K * sizeof(float));
    if (st != VSF_OK) return st;
    ctx->f_frames = n_frames;
  }
  return VSF_OK;
}

vsf_status ensure_temporal_buffers(vsf_ctx* ctx, int n_pairs) {
  const size_t K = (size_t)ctx->p.max_keypoints;
  if (n_pairs <= ctx->t_pairs) return VSF_OK;
  vsf_status st = grow_scratch(ctx, ctx->t_matches, (size_t)n_pairs * K * sizeof(vsf_dmatch));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->t_nmatches, (size_t)n_pairs * sizeof(int32_t));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->t_sortkeys, (size_t)n_pairs * K * 8);
  if (st != VSF_OK) return st;
  ctx->t_pairs = n_pairs;
  return VSF_OK;
}

vsf_status ensure_vision_buffers(vsf_ctx* ctx, int n_frames) {
  const size_t K = (size_t)ctx->p.max_keypoints;
  if (n_frames <= ctx->v_frames) return VSF_OK;
  vsf_status st = grow_scratch(ctx, ctx->v_pairs, (size_t)n_frames * K * 2 * sizeof(uint64_t));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->v_npairs, (size_t)n_frames * sizeof(int32_t));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->v_sets, (size_t)2 * n_frames * sizeof(int32_t));
  if (st != VSF_OK) return st;
  // q_set[f] = 2f + 1 (right frame, "initial", cc:131), t_set[f] = 2f (left frame, "current"): written by a kernel on the
  // context's stream, in order before the launches that read it (a hipMemcpy would wait for the stream)
  vsf_launch_fill_stereo_sets(ctx->v_sets, n_frames, ctx->stream);
  ctx->v_frames = n_frames;
  return VSF_OK;
}

vsf_status ensure_pack_buffers(vsf_ctx* ctx, int n) {
  if (n <= ctx->pk_entries) return VSF_OK;
  vsf_status st = grow_scratch(ctx, ctx->pk_offsets, (size_t)n * sizeof(uint32_t));
  if (st != VSF_OK) return st;
  ctx->pk_entries = n;
  return VSF_OK;
}

vsf_status reserve_scratch(vsf_ctx* ctx, int n_frames, int n_pairs) {
  const int pairs = std::max(n_frames, n_pairs);
  vsf_status st = ensure_match_buffers(ctx, pairs, ctx->p.max_keypoints);
  if (st == VSF_OK) st = ensure_residual_buffers(ctx, n_frames);
  if (st == VSF_OK) st = ensure_temporal_buffers(ctx, pairs);  // (Calculate3DPoints matches one pair per frame)
  if (st == VSF_OK) st = ensure_vision_buffers(ctx, n_frames);
  if (st == VSF_OK) st = ensure_pack_buffers(ctx, n_frames + n_pairs);
  return st;
}

void free_observe(vsf_ctx* ctx) {
  vsf_ctx::Observe& o = ctx->ob;
  hipFree(o.ring);
  hipFree(o.ring_counts);
  hipFree(o.kpf);
  hipFree(o.matches);
  hipFree(o.ints);
  hipFree(o.floats);
  hipFree(o.features);
  hipFree(o.pairs);
  hipFree(o.npairs);
  for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++) {
    if (o.h_img[i]) hipHostFree(o.h_img[i]);
    if (o.h_out[i]) hipHostFree(o.h_out[i]);
    if (o.h_meta[i]) hipHostFree(o.h_meta[i]);
    if (o.h_status[i]) hipHostFree(o.h_status[i]);
    if (o.ex_stream[i] && o.ex_stream[i] != ctx->stream) hipStreamDestroy(o.ex_stream[i]);
    if (o.ev_done[i]) hipEventDestroy(o.ev_done[i]);
  }
  o = vsf_ctx::Observe();
}

vsf_status check_status_word(vsf_ctx* ctx) {
  VSF_HIP(hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  VSF_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t), ctx->stream));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  VSF_STICKY();
  if (*ctx->h_status & 2) return VSF_ERR_INVALID_ARG;  // a JPEG stream broke off inside its entropy-coded data
  return (*ctx->h_status & 1) ? VSF_ERR_CAPACITY : VSF_OK;
}

vsf_status validate_images(const vsf_ctx* ctx, const uint8_t* d_imgs, int n, size_t image_stride, size_t row_stride) {
  if (!d_imgs || n < 1 || n > ctx->p.max_images) return VSF_ERR_INVALID_ARG;
  if (((uintptr_t)d_imgs & 15) || (image_stride & 15) || (row_stride & 15)) return VSF_ERR_INVALID_ARG;
  if (row_stride < (size_t)ctx->p.width || image_stride < row_stride * (size_t)ctx->p.height) return VSF_ERR_INVALID_ARG;
  return VSF_OK;
}

void prof_fold(vsf_ctx* ctx) {  // stream must be idle
  for (size_t i = 0; i < ctx->ev_used; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]) == hipSuccess) {
      ctx->prof_ms[ctx->ev_stage[i]] += ms;
      ctx->prof_launches[ctx->ev_stage[i]] += ctx->ev_launches[i];
    }
  }
  ctx->ev_used = 0;
}

void sync_all_streams(vsf_ctx* ctx) {  // every stream the context launches on
  if (ctx->stream) vsf_note(hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->own_stream != ctx->stream) vsf_note(hipStreamSynchronize(ctx->own_stream));
  if (ctx->aux_stream) vsf_note(hipStreamSynchronize(ctx->aux_stream));
  if (ctx->pipe_stream) vsf_note(hipStreamSynchronize(ctx->pipe_stream));
  if (ctx->blur_stream) vsf_note(hipStreamSynchronize(ctx->blur_stream));
  for (int i = 1; i < ctx->side.n; i++)
    if (ctx->side.stream[i]) vsf_note(hipStreamSynchronize(ctx->side.stream[i]));
  for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++)
    if (ctx->ob.ex_stream[i] && ctx->ob.ex_stream[i] != ctx->stream) vsf_note(hipStreamSynchronize(ctx->ob.ex_stream[i]));
}

struct StageTimer {  // records an event pair around one stage when profiling is on
  vsf_ctx* ctx;
  size_t slot = 0;
  bool on;
  hipStream_t st;
  StageTimer(vsf_ctx* c, hipStream_t stream, int stage, int launches) : ctx(c), on(c->prof_on), st(stream) {
    if (!on) return;
    if (ctx->ev_used >= 2048) {
      sync_all_streams(ctx);
      prof_fold(ctx);
    }
    slot = ctx->ev_used++;
    while (ctx->ev_pool.size() < 2 * (slot + 1)) {
      hipEvent_t e = nullptr;
      vsf_note(hipEventCreate(&e));
      ctx->ev_pool.push_back(e);
    }
    if (ctx->ev_stage.size() <= slot) {
      ctx->ev_stage.resize(slot + 1);
      ctx->ev_launches.resize(slot + 1);
    }
    ctx->ev_stage[slot] = stage;
    ctx->ev_launches[slot] = launches;
    vsf_note(hipEventRecord(ctx->ev_pool[2 * slot], st));
  }
  ~StageTimer() {
    if (on) vsf_note(hipEventRecord(ctx->ev_pool[2 * slot + 1], st));
  }
};

// The per-image work buffers of images [i0, i0 + n) seen as a batch of their own.
VsfDev shifted(const VsfDev& d, const VsfGeom& g, int i0) {
  VsfDev o = d;
  const size_t i = (size_t)i0;
  if (o.pyr) o.pyr += i * g.pyr_bytes;
  if (o.blur) o.blur += i * g.pyr_bytes;
  o.cand += i * g.cand_entries;
  o.rowstart += i * (size_t)g.nunits * VSF_FAST_RS_STRIDE;
  if (o.scratch) o.scratch += i * 6 * g.cand_entries;
  if (o.lvlkp) o.lvlkp += i * g.lvlkp_entries;
  if (o.lvl_count) o.lvl_count += i * g.nlevels;
  return o;
}

// vsf_set_input_event: the batched call that follows waits for the caller's event on the context's stream (level 0 of the
// pyramid IS the input: FAST, Harris and the orientation read it there) -- the pipelined pyramid chain waits for it by
// itself in extract_on -- and the event is forgotten when the call returns (one-shot).
struct InputEventScope {
  vsf_ctx* ctx;
  explicit InputEventScope(vsf_ctx* c) : ctx(c) {
    if (ctx->input_event) vsf_note(hipStreamWaitEvent(ctx->stream, ctx->input_event, 0));
  }
  ~InputEventScope() { ctx->input_event = nullptr; }
};

// detectAndCompute for images [i0, i0 + n) of `im` on stream `st`.  `status`: the status word the kernels report capacity
// overflows into (the context's, or the word of the frame in flight that owns this extraction).
void extract_on(vsf_ctx* ctx, hipStream_t st, const VsfImages& im_all, int i0, int n, vsf_keypoint* d_kp,
                uint8_t* d_desc, int32_t* d_counts, bool inputs_complete = false, const VsfSideStream* own_side = nullptr,
                int32_t* status = nullptr) {
  const VsfGeom& g = ctx->orb.g;
  VsfDev d = shifted(ctx->dorb.d, g, i0);
  if (status) d.status = status;
  VsfImages im = im_all;
  im.base += (size_t)i0 * im.image_stride;
  im.n = n;
  const size_t K = (size_t)ctx->p.max_keypoints;
  const bool pipe = ctx->pipeline && inputs_complete && ctx->lanes == 1 && st == ctx->stream && i0 == 0;
  if (pipe) {
    // The pyramid depends on the input images only.  The caller promised they are complete (vsf_set_pipeline), so the
    // chain goes onto the pipe streams WITHOUT being ordered after this stream's earlier work and overlaps the previous
    // call's later stages; it writes the pyramid buffer the previous call does not use, once the call before that has
    // released it.
    const int buf = ctx->pyr_flip;
    d.pyr = buf ? ctx->pyr_alt : ctx->dorb.d.pyr;
    // (ROCm multiplexes streams onto a few hardware queues; the context's aux stream is known to run beside the
    // main one, so the chain goes there, as a single chain)
    hipStream_t ps = ctx->pipe_stream ? ctx->pipe_stream : ctx->aux_stream;
    if (ctx->pyr_free_valid[buf]) vsf_note(hipStreamWaitEvent(ps, ctx->ev_pyr_free[buf], 0));
    // ... and not before the previous call's FAST kernel has finished: FAST fills every register of the chip, the
    // stages after it (selection, descriptors, matcher) are latency-bound and leave room for the resize chain
    if (ctx->fast_done_valid && ctx->tuning.pipe_after_fast) vsf_note(hipStreamWaitEvent(ps, ctx->ev_fast_done, 0));
    // ... and not before images this library itself is still producing on the context's stream are complete
    if (ctx->ingest_done_valid) vsf_note(hipStreamWaitEvent(ps, ctx->ev_ingest_done, 0));
    // ... nor before the caller's own producer has finished them (vsf_set_input_event)
    if (ctx->input_event) vsf_note(hipStreamWaitEvent(ps, ctx->input_event, 0));
    {
      StageTimer t(ctx, ps, VSF_STAGE_PYRAMID, g.nlevels - 1);
      vsf_launch_pyramid(d, g, ctx->orb.levels.data(), im, ps, nullptr);
    }
    vsf_note(hipEventRecord(ctx->ev_pyr_done, ps));
    vsf_note(hipStreamWaitEvent(st, ctx->ev_pyr_done, 0));
  } else {
    StageTimer t(ctx, st, VSF_STAGE_PYRAMID, g.nlevels - 1);
    // one lane: the aux stream is idle, the pyramid chain of the second half of the batch runs on it
    // (own_side: a caller that runs several extractions at once on different streams brings a side-stream set of its own --
    // the context's fork / join events must not be recorded from two streams at a time)
    vsf_launch_pyramid(d, g, ctx->orb.levels.data(), im, st, own_side ? own_side : (ctx->lanes == 1 ? &ctx->side : nullptr));
  }
  ctx->last_pyr = d.pyr - (size_t)i0 * g.pyr_bytes;
  // The blur needs the pyramid only.  It lives on the matrix cores and on memory bandwidth, FAST on the vector ALU (at its
  // issue ceiling: profiles/r04/valu_ceiling.json) and the selection on latency: forked behind the pyramid onto its own
  // stream, the blur's workgroups fill in as FAST drains and run beside the selection (7.61 -> 7.25 ms per 256-frame step).
  // Measured and left out: forking behind FAST instead (7.52: the overlap with FAST's tail is lost); making room beside
  // FAST with an LDS reservation that caps FAST at four workgroups per CU (7.46: the reservation is LDS the blur needs; the
  // resident kernel below caps FAST without it); a high- or low-priority blur stream (7.39 / 7.65); slices of the blur
  // forked from INSIDE the pyramid's launch chain as soon as their levels exist (the chain's dependent launches stretch from
  // 1.28 to 1.8-2.2 ms beside the blur's memory traffic, 7.37-7.46 ms per step against 7.31); the first 3 / 6 / 10 / 16
  // levels blurred in line in front of FAST and only the rest beside it (7.25-7.36: noise).
  const bool march = ctx->tuning.blur_march != 0;
  const bool beside_ok = ctx->blur_overlap && !march && im.n >= 32 && ctx->blur_stream;
  // With the blur beside it FAST can run as ONE resident workgroup per CU (k_fast.hip): three waves per SIMD keep 92 % of
  // its own rate and leave the other 224 of a SIMD's 512 registers -- which a grid of one workgroup per four cells fills
  // for as long as cells are left -- to the blur, which then finishes INSIDE the FAST pass instead of after it.  Whether
  // that pays depends on the batch and on how well the blur hides behind the selection anyway (frames per second, grid ->
  // resident; 640x480 / 2000 features: 512 frames per step 35.4 -> 37.5 k, 256 frames 35.0 -> 36.4 k, 128 frames equal,
  // 64 frames 31.3 -> 29.8 k, 32 frames 25.7 -> 24.3 k; 1920x1080 / 8000: 32 frames 4 900 -> 4 700, 96 frames
  // 5 000 -> 5 140, 192 frames 5 030 -> 5 320; four waves per SIMD at 256 VGA frames 7.09-7.12 ms against 7.03, two 8.0).
  // No rule on batch size got all of these right, so the caller may have it MEASURED: vsf_tune_fast_resident (explicit,
  // blocking) times both forms on the caller's own batch and this call takes what it found for this batch size -- the grid
  // form for a size nobody measured.  vsf_set_fast_resident overrides.  Nothing is measured, and nothing waits, in here.
  int resident = 0;
  if (beside_ok && ctx->lanes == 1 && st == ctx->stream && i0 == 0 && !own_side) {
    if (ctx->fast_force >= 0)
      resident = ctx->fast_force;
    else if (ctx->fast_resident >= 0)
      resident = ctx->fast_resident;
    else if (ctx->fast_tune.n == im.n && ctx->fast_tune.choice > 0)
      resident = ctx->fast_tune.choice;
  }
  const bool blur_beside = beside_ok;
  auto launch_blur = [&](hipStream_t bs) {
    StageTimer t(ctx, bs, VSF_STAGE_BLUR, 1);
    // VSF_OPT_BLUR_MARCH: round 2's vector-ALU kernel (A/B measurements); default: the matrix-core kernel
    if (march)
      vsf_launch_blur(d, g, im, ctx->dorb.blur_tiles, (int)ctx->orb.blur_tiles.size(), ctx->gauss, bs);
    else if (im.n >= 32)
      vsf_launch_blur_mma(d, g, im, ctx->dorb.blur_mma_units, (int)ctx->orb.blur_mma_units.size(), ctx->dorb.blur_tcol,
                          ctx->dorb.blur_tv, ctx->orb.blur_bias, bs);
    else
      vsf_launch_blur_mma(d, g, im, ctx->dorb.blur_mma_units_small, (int)ctx->orb.blur_mma_units_small.size(),
                          ctx->dorb.blur_tcol, ctx->dorb.blur_tv, ctx->orb.blur_bias, bs);
  };
  auto fork_blur = [&]() {
    vsf_note(hipEventRecord(ctx->ev_blur_fork, st));
    vsf_note(hipStreamWaitEvent(ctx->blur_stream, ctx->ev_blur_fork, 0));
    launch_blur(ctx->blur_stream);
    vsf_note(hipEventRecord(ctx->ev_blur_done, ctx->blur_stream));
  };
  if (blur_beside) fork_blur();
  {
    StageTimer t(ctx, st, VSF_STAGE_FAST, 1);
    // VSF_OPT_FAST_BITS: the segment test on bit planes, scores only where it fires (k_fastbits.hip; the same candidates)
    const int fb = ctx->tuning.fast_bits;
    if (d.bits_items && ctx->p.fast_threshold >= 1 && (fb == 2 || (fb == 1 && im.n >= 8)))
      vsf_launch_fast_bits(d, g, im, d.bits_items, d.n_bits_items, ctx->p.fast_threshold, st);
    else
      vsf_launch_fast(d, g, im, ctx->p.fast_threshold, 1, st, blur_beside ? resident : 0, ctx->n_cus, ctx->fast_cells);
  }
  if (pipe) {
    vsf_note(hipEventRecord(ctx->ev_fast_done, st));
    ctx->fast_done_valid = true;
  }
  {
    StageTimer t(ctx, st, VSF_STAGE_SELECT, 1);
    vsf_launch_select(d, g, ctx->orb.levels.data(), im, st);
  }
  if (blur_beside)
    vsf_note(hipStreamWaitEvent(st, ctx->ev_blur_done, 0));
  else
    launch_blur(st);
  {
    StageTimer t(ctx, st, VSF_STAGE_DESCRIBE, 1);
    vsf_launch_describe(d, g, im, ctx->p.max_keypoints, d_kp + i0 * K, d_desc + i0 * K * VSF_DESC_BYTES, d_counts + i0,
                        st);
  }
  if (pipe) {  // every reader of this pyramid buffer is queued: the call after the next may overwrite it
    const int buf = ctx->pyr_flip;
    vsf_note(hipEventRecord(ctx->ev_pyr_free[buf], st));
    ctx->pyr_free_valid[buf] = true;
    ctx->pyr_flip ^= 1;
  }
}

// knnMatch(k = 2) + ratio test for pairs [p0, p0 + n) on stream `st`.
void match_on(vsf_ctx* ctx, hipStream_t st, const uint8_t* d_desc, const int32_t* d_counts, size_t set_stride,
              const int32_t* d_q_set, const int32_t* d_t_set, int p0, int n, int32_t* d_idx2, int32_t* d_dist2,
              vsf_dmatch* d_matches, int32_t* d_nmatches, int32_t* status = nullptr) {
  const int rows = ctx->p.max_keypoints;
  const size_t R = (size_t)rows;
  // implicit pairing (set 2p vs 2p + 1) is relative to the descriptor base: shift the base instead of the indices
  const uint8_t* desc = d_desc;
  const int32_t* counts = d_counts;
  if (!d_q_set) {
    desc += (size_t)(2 * p0) * set_stride;
    counts += 2 * p0;
  }
  int32_t* idx2 = d_idx2 + (size_t)p0 * R * 2;
  int32_t* dist2 = d_dist2 + (size_t)p0 * R * 2;
  {
    StageTimer t(ctx, st, VSF_STAGE_KNN2, 1);
    vsf_launch_knn2(desc, counts, set_stride, d_q_set ? d_q_set + p0 : nullptr, d_t_set ? d_t_set + p0 : nullptr, n,
                    rows, idx2, dist2, st, ctx->tuning.match_int8 != 0);
  }
  {
    StageTimer t(ctx, st, VSF_STAGE_RATIO, 1);
    vsf_launch_ratio_compact(counts, d_q_set ? d_q_set + p0 : nullptr, d_t_set ? d_t_set + p0 : nullptr, n, rows, idx2,
                             dist2, ctx->p.ratio_num, ctx->p.ratio_shift, d_matches + (size_t)p0 * R, d_nmatches + p0,
                             status ? status : ctx->d_status, st);
  }
}

// Opens / closes the second lane: work queued on aux_stream between fork and join is ordered after everything
// already on `stream` and before everything queued on it afterwards.
vsf_status fork_lane(vsf_ctx* ctx) {
  VSF_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
  VSF_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
  return VSF_OK;
}
vsf_status join_lane(vsf_ctx* ctx) {
  VSF_HIP(hipEventRecord(ctx->ev_join, ctx->aux_stream));
  VSF_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return VSF_OK;
}

// Runs body(stream, first, count) over `units` work items (images or stereo frames): all on the context's stream, or
// (vsf_set_lanes(ctx, 2)) as two halves on the two lanes.  Measured on MI355X: the stages are either VALU-bound
// (FAST, blur) or latency-bound, and a VALU-bound kernel at full occupancy leaves no registers for a second kernel's
// waves, so the second lane only fills launch gaps and tails (+5 % frames/s) while every kernel's own duration
// roughly doubles; one lane stays the default.
template <class Body>
vsf_status run_chunked(vsf_ctx* ctx, int units, Body body) {
  if (ctx->lanes < 2 || units < 2) {
    body(ctx->stream, 0, units);
    return VSF_OK;
  }
  vsf_status st = fork_lane(ctx);
  if (st != VSF_OK) return st;
  const int n0 = (units + 1) / 2;
  body(ctx->stream, 0, n0);
  body(ctx->aux_stream, n0, units - n0);
  return join_lane(ctx);
}

vsf_status extract_async(vsf_ctx* ctx, const VsfImages& im, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                         bool inputs_complete = false) {
  vsf_status st = run_chunked(ctx, im.n, [&](hipStream_t s, int i0, int n) {
    extract_on(ctx, s, im, i0, n, d_kp, d_desc, d_counts, inputs_complete);
  });
  if (st != VSF_OK) return st;
  ctx->last_images = im;
  ctx->last_valid = true;
  VSF_STICKY();
  return VSF_OK;
}

}  // namespace

hipStream_t vsf_ctx_stream(const vsf_ctx* ctx) { return ctx->stream; }
int vsf_ctx_device(const vsf_ctx* ctx) { return ctx->device; }
void vsf_ctx_set_last_error(vsf_ctx* ctx, int code) { ctx->last_hip = code; }
void vsf_ctx_absorb_noted_error(vsf_ctx* ctx) {
  if (ctx->pending_hip == 0) ctx->pending_hip = vsf_tls_hip_error;
  vsf_tls_hip_error = 0;
}

extern "C" {

vsf_status vsf_params_default(vsf_params* p, int width, int height, int max_images) {
  if (!p) return VSF_ERR_INVALID_ARG;
  std::memset(p, 0, sizeof(*p));
  p->nfeatures = 10000;
  p->scale_factor = 1.04f;
  p->nlevels = 50;
  p->edge_threshold = 31;
  p->first_level = 0;
  p->wta_k = 2;
  p->score_type = 0;
  p->patch_size = 31;
  p->fast_threshold = 20;
  p->blur_sse2 = 1;
  p->residual_order = 0;  // Eigen 3.3's a0*b0 + (a1*b1 + a2*b2)
  p->fast_detector_threshold = 10;
  p->fast_detector_nms = 1;
  p->width = width;
  p->height = height;
  p->max_images = max_images;
  p->max_keypoints = 0;
  return vsf_params_set_ratio(p, 0.6f);
}

vsf_status vsf_params_set_ratio(vsf_params* p, float nn_match_ratio) {
  if (!p || !(nn_match_ratio > 0.f) || !(nn_match_ratio < 256.f)) return VSF_ERR_INVALID_ARG;
  // nn_match_ratio widened to double (slam_frontend.cc:523) == num / 2^shift exactly.
  double r = (double)nn_match_ratio;
  uint32_t shift = 0;
  while (r != std::floor(r) && shift < 31) {
    r *= 2.0;
    ++shift;
  }
  if (r != std::floor(r) || r >= 4294967296.0) return VSF_ERR_INVALID_ARG;
  p->ratio_num = (uint32_t)r;
  p->ratio_shift = shift;
  return VSF_OK;
}

const char* vsf_status_string(vsf_status s) {
  switch (s) {
    case VSF_OK: return "ok";
    case VSF_ERR_INVALID_ARG: return "invalid argument";
    case VSF_ERR_CAPACITY: return "capacity exceeded (results truncated)";
    case VSF_ERR_HIP: return "HIP runtime error";
    case VSF_ERR_UNSUPPORTED: return "unsupported parameter combination";
    case VSF_ERR_NO_DEVICE: return "no usable GPU";
  }
  return "unknown";
}

vsf_status vsf_create(const vsf_params* p, int device, vsf_ctx** out) {
  if (!p || !out) return VSF_ERR_INVALID_ARG;
  *out = nullptr;
  if (p->first_level != 0 || p->wta_k != 2 || p->score_type != 0 || p->patch_size != 31) return VSF_ERR_UNSUPPORTED;
  if (p->nlevels < 1 || p->nlevels > VSF_MAX_LEVELS || p->width < 16 || p->height < 16 || p->max_images < 1 ||
      p->nfeatures < 0 || p->edge_threshold < 0 || !(p->scale_factor > 1.0f))
    return VSF_ERR_INVALID_ARG;
  if (p->edge_threshold < 22) return VSF_ERR_UNSUPPORTED;  // borders are not materialised: needs reach 22 <= edge
  {
    const std::vector<int> um = orb_umax(31);
    static const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int i = 0; i < 16; i++)
      if (um[i] != expect[i]) return VSF_ERR_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) return VSF_ERR_NO_DEVICE;
  vsf_ctx* ctx = new (std::nothrow) vsf_ctx();
  if (!ctx) return VSF_ERR_INVALID_ARG;
  ctx->p = *p;
  if (ctx->p.max_keypoints <= 0) ctx->p.max_keypoints = ctx->p.nfeatures + 256;
  ctx->device = device;
  auto fail = [&](vsf_status s) {
    vsf_destroy(ctx);
    return s;
  };
  if (hipSetDevice(device) != hipSuccess) return fail(VSF_ERR_NO_DEVICE);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
  }
  if (!build_geometry(ctx->p, true, true, &ctx->orb)) return fail(VSF_ERR_INVALID_ARG);
  gaussian_taps(ctx->gauss);
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(VSF_ERR_HIP);
  ctx->stream = ctx->own_stream;
  if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->side.fork, hipEventDisableTiming) != hipSuccess)
    return fail(VSF_ERR_HIP);
  if (hipStreamCreateWithFlags(&ctx->blur_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_blur_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_blur_done, hipEventDisableTiming) != hipSuccess)
    return fail(VSF_ERR_HIP);
  ctx->side.stream[0] = ctx->aux_stream;
  for (int i = 0; i < VSF_SIDE_STREAMS; i++) {
    if (i > 0 && hipStreamCreateWithFlags(&ctx->side.stream[i], hipStreamNonBlocking) != hipSuccess)
      return fail(VSF_ERR_HIP);
    if (hipEventCreateWithFlags(&ctx->side.join[i], hipEventDisableTiming) != hipSuccess) return fail(VSF_ERR_HIP);
    ctx->side.n = i + 1;
  }
  if (hipMalloc((void**)&ctx->d_status, (1 + VSF_OBSERVE_MAX_SLOTS) * sizeof(int32_t)) != hipSuccess) return fail(VSF_ERR_HIP);
  if (hipMemset(ctx->d_status, 0, (1 + VSF_OBSERVE_MAX_SLOTS) * sizeof(int32_t)) != hipSuccess) return fail(VSF_ERR_HIP);
  {
    // Three kernels ask for more dynamic LDS than the default 64 KB (the parallel sort of GetFeatureMatches, the slab
    // pyramid, the parallel JPEG decode): how much a workgroup of this device may have is asked once, their limits are
    // raised here -- checked -- and the launchers fall back to their plain forms where it is not enough.
    int per_block = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) per_block = 0;
    if (hipDeviceGetAttribute(&per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess) per_cu = 0;
    int limit = std::min(std::max(std::max(per_block, per_cu), 64 * 1024), 160 * 1024);
    if (vsf_prepare_sort_kernels(limit) != hipSuccess) {
      (void)hipGetLastError();
      limit = 64 * 1024;  // (nothing above the default was granted: the one-lane sort and the per-level pyramid launches)
    }
    ctx->tuning.lds_limit = limit;
    if (vsf_prepare_pyramid_kernels(limit) != hipSuccess) {
      (void)hipGetLastError();
      ctx->tuning.pyramid_chain = 0;
    }
    if (vsf_prepare_jpeg_kernels(limit) != hipSuccess) {
      (void)hipGetLastError();
      ctx->tuning.jpeg_serial = 1;
    }
  }
  if (hipMalloc((void**)&ctx->fast_cells, 2 * sizeof(uint32_t)) != hipSuccess) return fail(VSF_ERR_HIP);
  if (hipHostMalloc((void**)&ctx->h_status, sizeof(int32_t), hipHostMallocDefault) != hipSuccess)
    return fail(VSF_ERR_HIP);
  vsf_status st = alloc_devset(ctx, ctx->orb, &ctx->dorb, true, ctx->p.max_images);
  if (st != VSF_OK) return fail(st);
  // staging buffers of the host-pointer API (one image / one set of outputs per image slot)
  ctx->st_img_pitch = (size_t)align_up(ctx->p.width, 64);
  ctx->st_img_stride = ctx->st_img_pitch * (size_t)ctx->p.height;
  const size_t n = (size_t)ctx->p.max_images, K = (size_t)ctx->p.max_keypoints;
  if (hipMalloc((void**)&ctx->st_img, n * ctx->st_img_stride) != hipSuccess ||
      hipMalloc((void**)&ctx->st_kp, n * K * sizeof(vsf_keypoint)) != hipSuccess ||
      hipMalloc((void**)&ctx->st_desc, n * K * VSF_DESC_BYTES) != hipSuccess ||
      hipMalloc((void**)&ctx->st_counts, n * sizeof(int32_t)) != hipSuccess)
    return fail(VSF_ERR_HIP);
  // scratch of the batched *_dev calls for whole batches of this context (vsf_reserve sizes it for others)
  if (reserve_scratch(ctx, std::max(1, ctx->p.max_images / 2), std::max(1, ctx->p.max_images / 2)) != VSF_OK)
    return fail(VSF_ERR_HIP);
  if (hipDeviceSynchronize() != hipSuccess) return fail(VSF_ERR_HIP);
  *out = ctx;
  return VSF_OK;
}

void vsf_destroy(vsf_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  // every stream the context ever launched on -- the slots' streams of frames still in flight included: their kernels
  // write device buffers and pinned host memory that is freed below
  sync_all_streams(ctx);
  vsf_tls_hip_error = 0;
  if (ctx->ev_pyr_done) {
    hipEventDestroy(ctx->ev_pyr_done);
    hipEventDestroy(ctx->ev_fast_done);
    for (hipEvent_t e : ctx->ev_pyr_free) hipEventDestroy(e);
  }
  if (ctx->ev_ingest_done) hipEventDestroy(ctx->ev_ingest_done);
  hipFree(ctx->pyr_alt);
  free_devset(&ctx->dorb);
  free_devset(&ctx->dfast);
  hipFree(ctx->d_status);
  hipFree(ctx->fast_cells);
  for (hipEvent_t e : ctx->fast_tune.ev)
    if (e) hipEventDestroy(e);
  if (ctx->h_status) hipHostFree(ctx->h_status);
  hipFree(ctx->st_img);
  hipFree(ctx->st_kp);
  hipFree(ctx->st_desc);
  hipFree(ctx->st_counts);
  hipFree(ctx->m_idx2);
  hipFree(ctx->m_dist2);
  hipFree(ctx->f_residual);
  hipFree(ctx->t_matches);
  hipFree(ctx->t_nmatches);
  hipFree(ctx->t_sortkeys);
  free_observe(ctx);
  hipFree(ctx->jp_flags);
  hipFree(ctx->png_filtered);
  hipFree(ctx->png_file_status);
  hipFree(ctx->jp_clean);
  hipFree(ctx->jp_coef);
  for (int i = 0; i < 2; i++) {
    if (ctx->jp_host[i]) hipHostFree(ctx->jp_host[i]);
    hipFree(ctx->jp_dev[i]);
    if (ctx->jp_copied[i]) hipEventDestroy(ctx->jp_copied[i]);
  }
  free_retired(ctx);
  hipFree(ctx->v_pairs);
  hipFree(ctx->v_npairs);
  hipFree(ctx->v_sets);
  hipFree(ctx->pk_offsets);
  hipFree(ctx->mm_desc);
  hipFree(ctx->mm_counts);
  hipFree(ctx->mm_matches);
  hipFree(ctx->mm_nmatches);
  hipFree(ctx->mh_desc);
  hipFree(ctx->mh_counts);
  hipFree(ctx->mh_matches);
  hipFree(ctx->mh_nmatches);
  for (hipEvent_t e : ctx->ev_pool) hipEventDestroy(e);
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  if (ctx->side.fork) hipEventDestroy(ctx->side.fork);
  for (int i = 0; i < VSF_SIDE_STREAMS; i++) {
    if (ctx->side.join[i]) hipEventDestroy(ctx->side.join[i]);
    if (i > 0 && ctx->side.stream[i]) {
      hipStreamSynchronize(ctx->side.stream[i]);
      hipStreamDestroy(ctx->side.stream[i]);
    }
  }
  if (ctx->aux_stream) hipStreamDestroy(ctx->aux_stream);
  if (ctx->pipe_stream) hipStreamDestroy(ctx->pipe_stream);
  if (ctx->blur_stream) hipStreamDestroy(ctx->blur_stream);
  if (ctx->ev_blur_fork) hipEventDestroy(ctx->ev_blur_fork);
  if (ctx->ev_blur_done) hipEventDestroy(ctx->ev_blur_done);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int vsf_last_hip_error(const vsf_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

vsf_status vsf_get_params(const vsf_ctx* ctx, vsf_params* out) {
  if (!ctx || !out) return VSF_ERR_INVALID_ARG;
  *out = ctx->p;
  return VSF_OK;
}

vsf_status vsf_set_stream(vsf_ctx* ctx, void* hip_stream) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  prof_fold(ctx);
  // (The handle must be a live stream of this device: the HIP runtime of ROCm 7 dereferences a stream handle without
  // looking it up -- hipStreamQuery and the launch path alike crashed on a destroyed one in the round-4 test runs -- so a
  // stale handle cannot be refused here, as with any HIP call that takes a stream.)
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return VSF_OK;
}

// Test hook: makes the context's thread behave as if a launcher had just noted HIP error `code` (vsf_note): the next entry
// point that launches must return VSF_ERR_HIP with that code, and the one after it must work again.
vsf_status vsf_debug_inject_hip_error(vsf_ctx* ctx, int code) {
  VsfErrorScope scope_(ctx);
  if (!ctx || code <= 0) return VSF_ERR_INVALID_ARG;
  vsf_note((hipError_t)code);
  return VSF_OK;
}

vsf_status vsf_set_lanes(vsf_ctx* ctx, int lanes) {
  VsfErrorScope scope_(ctx);
  if (!ctx || lanes < 1 || lanes > 2) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->lanes = lanes;
  return VSF_OK;
}

vsf_status vsf_set_blur_overlap(vsf_ctx* ctx, int on) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->blur_overlap = on ? 1 : 0;
  return VSF_OK;
}

vsf_status vsf_set_fast_resident(vsf_ctx* ctx, int waves) {
  VsfErrorScope scope_(ctx);
  if (!ctx || waves < -1 || waves == 1 || waves > 4) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->fast_resident = waves;
  return VSF_OK;
}

vsf_status vsf_get_fast_resident(const vsf_ctx* ctx, int* waves) {
  if (!ctx || !waves) return VSF_ERR_INVALID_ARG;
  *waves = ctx->fast_resident >= 0 ? ctx->fast_resident : std::max(ctx->fast_tune.choice, 0);
  return VSF_OK;
}

vsf_status vsf_set_option(vsf_ctx* ctx, int option, int value) {
  VsfErrorScope scope_(ctx);
  if (!ctx || option < 0 || option >= VSF_OPT_COUNT) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  VsfTuning& t = ctx->tuning;
  switch (option) {
    case VSF_OPT_BLUR_MARCH: t.blur_march = value != 0; break;
    case VSF_OPT_FAST_BOTH_MAX:
      if (value < 0) return VSF_ERR_INVALID_ARG;
      t.fast_both_max = value;
      break;
    case VSF_OPT_SORT_SERIAL: t.sort_serial = value != 0; break;
    case VSF_OPT_SELECT_WIDE: t.select_wide = value != 0; break;
    case VSF_OPT_SELECT_BIG_CLASS: t.select_big_class = value != 0; break;
    case VSF_OPT_PIPE_AFTER_FAST: t.pipe_after_fast = value != 0; break;
    case VSF_OPT_MATCH_INT8: t.match_int8 = value != 0; break;
    case VSF_OPT_FAST_BITS:
      if (value < 0 || value > 2) return VSF_ERR_INVALID_ARG;
      t.fast_bits = value;
      break;
    case VSF_OPT_PIPE_PRIORITY:
      if (value < -1 || value > 1) return VSF_ERR_INVALID_ARG;
      t.pipe_priority = value;  // (takes effect with the next vsf_set_pipeline(ctx, 1))
      break;
    case VSF_OPT_JPEG_SERIAL:
      if (!value && vsf_prepare_jpeg_kernels(t.lds_limit) != hipSuccess) {  // (the parallel decoder's LDS was refused)
        (void)hipGetLastError();
        return VSF_ERR_UNSUPPORTED;
      }
      t.jpeg_serial = value != 0;
      break;
    case VSF_OPT_PYRAMID_FEW:
      if (value < 0) return VSF_ERR_INVALID_ARG;
      t.pyramid_few = value;
      break;
    case VSF_OPT_PYRAMID_CHAIN:
      if (value < 0 || value > 64) return VSF_ERR_INVALID_ARG;
      if (value > 0 && t.lds_limit < 160 * 1024 - 2048) return VSF_ERR_UNSUPPORTED;
      t.pyramid_chain = value;
      break;
    case VSF_OPT_PYRAMID_ROWS:
      if (value < 1) return VSF_ERR_INVALID_ARG;
      t.pyramid_rows = value;
      break;
    default: return VSF_ERR_INVALID_ARG;
  }
  return VSF_OK;
}

vsf_status vsf_get_option(const vsf_ctx* ctx, int option, int* value) {
  if (!ctx || !value) return VSF_ERR_INVALID_ARG;
  const VsfTuning& t = ctx->tuning;
  switch (option) {
    case VSF_OPT_BLUR_MARCH: *value = t.blur_march; break;
    case VSF_OPT_FAST_BOTH_MAX: *value = t.fast_both_max; break;
    case VSF_OPT_SORT_SERIAL: *value = t.sort_serial; break;
    case VSF_OPT_SELECT_WIDE: *value = t.select_wide; break;
    case VSF_OPT_SELECT_BIG_CLASS: *value = t.select_big_class; break;
    case VSF_OPT_PIPE_AFTER_FAST: *value = t.pipe_after_fast; break;
    case VSF_OPT_MATCH_INT8: *value = t.match_int8; break;
    case VSF_OPT_FAST_BITS: *value = t.fast_bits; break;
    case VSF_OPT_PIPE_PRIORITY: *value = t.pipe_priority; break;
    case VSF_OPT_JPEG_SERIAL: *value = t.jpeg_serial; break;
    case VSF_OPT_PYRAMID_FEW: *value = t.pyramid_few; break;
    case VSF_OPT_PYRAMID_CHAIN: *value = t.pyramid_chain; break;
    case VSF_OPT_PYRAMID_ROWS: *value = t.pyramid_rows; break;
    default: return VSF_ERR_INVALID_ARG;
  }
  return VSF_OK;
}

vsf_status vsf_set_pipeline(vsf_ctx* ctx, int on) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  VSF_HIP(hipStreamSynchronize(ctx->aux_stream));  // (the pipelined pyramid chain runs there ...
  if (ctx->pipe_stream) VSF_HIP(hipStreamSynchronize(ctx->pipe_stream));  // ... or on a stream of its own priority)
  if (on && ctx->tuning.pipe_priority != ctx->pipe_stream_priority) {
    if (ctx->pipe_stream) VSF_HIP(hipStreamDestroy(ctx->pipe_stream));
    ctx->pipe_stream = nullptr;
    ctx->pipe_stream_priority = 0;
    if (ctx->tuning.pipe_priority != 0) {
      int least = 0, greatest = 0;
      VSF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      VSF_HIP(hipStreamCreateWithPriority(&ctx->pipe_stream, hipStreamNonBlocking,
                                          ctx->tuning.pipe_priority > 0 ? least : greatest));
      ctx->pipe_stream_priority = ctx->tuning.pipe_priority;
    }
  }
  if (on && !ctx->pyr_alt) {
    VSF_HIP(hipMalloc((void**)&ctx->pyr_alt, (size_t)ctx->p.max_images * ctx->orb.g.pyr_bytes));
    VSF_HIP(hipEventCreateWithFlags(&ctx->ev_pyr_done, hipEventDisableTiming));
    VSF_HIP(hipEventCreateWithFlags(&ctx->ev_fast_done, hipEventDisableTiming));
    for (hipEvent_t& e : ctx->ev_pyr_free) VSF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ctx->pipeline = on != 0;
  ctx->pyr_free_valid[0] = ctx->pyr_free_valid[1] = false;
  ctx->fast_done_valid = false;
  return VSF_OK;
}

vsf_status vsf_sync(vsf_ctx* ctx) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  const vsf_status st = check_status_word(ctx);
  if (!ctx->retired.empty()) {  // scratch a *_dev call outgrew: nothing can be using it once every stream is idle
    sync_all_streams(ctx);
    free_retired(ctx);
  }
  return st;
}

vsf_status vsf_set_input_event(vsf_ctx* ctx, void* hip_event) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  ctx->input_event = static_cast<hipEvent_t>(hip_event);
  return VSF_OK;
}

vsf_status vsf_reserve(vsf_ctx* ctx, int n_frames, int n_pairs) {
  VsfErrorScope scope_(ctx);
  if (!ctx || n_frames < 1 || n_pairs < 0) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  sync_all_streams(ctx);
  free_retired(ctx);
  vsf_status st = reserve_scratch(ctx, n_frames, n_pairs);
  if (st != VSF_OK) return st;
  sync_all_streams(ctx);  // (the set-index fill)
  free_retired(ctx);
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_level_info(const vsf_ctx* ctx, int level, int* w, int* h, float* scale, int* nfeatures) {
  if (!ctx || level < 0 || level >= ctx->orb.g.nlevels) return VSF_ERR_INVALID_ARG;
  const VsfLevel& L = ctx->orb.levels[level];
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (scale) *scale = L.scale;
  if (nfeatures) *nfeatures = L.nfeatures;
  return VSF_OK;
}

uint64_t vsf_pyramid_pixels(const vsf_ctx* ctx) { return ctx ? ctx->orb.g.pyramid_pixels : 0; }

uint64_t vsf_algorithmic_bytes_per_image(const vsf_ctx* ctx) {
  if (!ctx) return 0;
  // resize reads + resize writes + FAST read + blur read/write + outputs.
  const auto& Ls = ctx->orb.levels;
  uint64_t P = 0, rd = 0, wr = 0;
  for (size_t l = 0; l < Ls.size(); l++) {
    const uint64_t px = (uint64_t)Ls[l].w * Ls[l].h;
    P += px;
    if (l + 1 < Ls.size()) rd += px;
    if (l >= 1) wr += px;
  }
  return rd + wr + P + 2 * P + (uint64_t)ctx->p.nfeatures * (sizeof(vsf_keypoint) + VSF_DESC_BYTES);
}

vsf_status vsf_extract_batch_dev(vsf_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride,
                                 size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_desc || !d_counts) return VSF_ERR_INVALID_ARG;
  vsf_status st = validate_images(ctx, d_imgs, n_images, image_stride, row_stride);
  if (st != VSF_OK) return st;
  VSF_HIP(hipSetDevice(ctx->device));
  VsfImages im{d_imgs, image_stride, row_stride, n_images};
  InputEventScope input(ctx);
  return extract_async(ctx, im, d_kp, d_desc, d_counts, true);
}

vsf_status vsf_tune_fast_resident(vsf_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride,
                                  size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                                  int samples, float* ms_grid, float* ms_resident) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_desc || !d_counts || samples < 1 || samples > 64 || !ms_grid || !ms_resident)
    return VSF_ERR_INVALID_ARG;
  *ms_grid = *ms_resident = 0.f;
  vsf_status st = validate_images(ctx, d_imgs, n_images, image_stride, row_stride);
  if (st != VSF_OK) return st;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_ctx::FastTune& T = ctx->fast_tune;
  T.n = n_images;
  T.choice = 0;
  // batches the blur does not run beside have one form only
  if (!(ctx->blur_overlap && !ctx->tuning.blur_march && n_images >= 32 && ctx->blur_stream && ctx->lanes == 1)) return VSF_OK;
  for (hipEvent_t& e : T.ev)
    if (!e) VSF_HIP(hipEventCreate(&e));
  const VsfImages im{d_imgs, image_stride, row_stride, n_images};
  sync_all_streams(ctx);  // nothing of an earlier call beside the timed runs
  std::vector<float> ms[2];
  vsf_status out = VSF_OK;
  for (int run = 0; run < 1 + 2 * samples && out == VSF_OK; run++) {
    const int form = run == 0 ? 0 : (run - 1) & 1;  // warm-up (grid), then grid / resident alternately on the SAME input
    ctx->fast_force = form ? 3 : 0;
    hipError_t e = hipEventRecord(T.ev[0], ctx->stream);
    // (inputs_complete = false: no cross-call pipelining inside the measurement, every run is the whole extraction)
    extract_on(ctx, ctx->stream, im, 0, n_images, d_kp, d_desc, d_counts, false);
    if (e == hipSuccess) e = hipEventRecord(T.ev[1], ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(T.ev[1]);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, T.ev[0], T.ev[1]);
    if (e != hipSuccess) {
      ctx->last_hip = (int)e;
      out = VSF_ERR_HIP;
    } else if (run > 0) {
      ms[form].push_back(t);
    }
  }
  ctx->fast_force = -1;
  ctx->last_images = im;
  ctx->last_valid = true;
  if (out != VSF_OK) return out;
  VSF_STICKY();
  for (auto& v : ms) std::sort(v.begin(), v.end());
  *ms_grid = ms[0][ms[0].size() / 2];
  *ms_resident = ms[1][ms[1].size() / 2];
  T.choice = *ms_resident < *ms_grid ? 3 : 0;
  return VSF_OK;
}

vsf_status vsf_match_batch_dev(vsf_ctx* ctx, const uint8_t* d_desc, const int32_t* d_counts, size_t set_stride,
                               const int32_t* d_q_set, const int32_t* d_t_set, int n_pairs, int32_t* d_idx2,
                               int32_t* d_dist2, vsf_dmatch* d_matches, int32_t* d_nmatches) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_desc || !d_counts || n_pairs < 1 || !d_matches || !d_nmatches || (set_stride & 15))
    return VSF_ERR_INVALID_ARG;
  if ((d_idx2 == nullptr) != (d_dist2 == nullptr) || (d_q_set == nullptr) != (d_t_set == nullptr))
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const int rows = ctx->p.max_keypoints;
  if (!d_idx2) {
    vsf_status st = ensure_match_buffers(ctx, n_pairs, rows);
    if (st != VSF_OK) return st;
    d_idx2 = ctx->m_idx2;
    d_dist2 = ctx->m_dist2;
  }
  vsf_status st = run_chunked(ctx, n_pairs, [&](hipStream_t s, int p0, int n) {
    match_on(ctx, s, d_desc, d_counts, set_stride, d_q_set, d_t_set, p0, n, d_idx2, d_dist2, d_matches, d_nmatches);
  });
  if (st != VSF_OK) return st;
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_profile_enable(vsf_ctx* ctx, int on) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  sync_all_streams(ctx);
  prof_fold(ctx);
  ctx->prof_on = on != 0;
  return VSF_OK;
}

vsf_status vsf_profile_read(vsf_ctx* ctx, double* ms_total, int64_t* launches, int reset) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !ms_total || !launches) return VSF_ERR_INVALID_ARG;
  sync_all_streams(ctx);
  prof_fold(ctx);
  for (int i = 0; i < VSF_STAGE_COUNT; i++) {
    ms_total[i] = ctx->prof_ms[i];
    launches[i] = ctx->prof_launches[i];
    if (reset) {
      ctx->prof_ms[i] = 0;
      ctx->prof_launches[i] = 0;
    }
  }
  return VSF_OK;
}

const char* vsf_stage_name(int stage) {
  static const char* names[VSF_STAGE_COUNT] = {"pyramid_resize", "fast_score_nms", "select_harris_angle", "gauss_blur7",
                                               "orb_describe",   "hamming_knn2",   "ratio_compact", "frontend_tail"};
  return (stage >= 0 && stage < VSF_STAGE_COUNT) ? names[stage] : "?";
}

vsf_status vsf_stereo_batch_dev(vsf_ctx* ctx, const uint8_t* d_imgs, int n_frames, size_t image_stride,
                                size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                                vsf_dmatch* d_matches, int32_t* d_nmatches) {
  VsfErrorScope scope_(ctx);
  if (!ctx || n_frames < 1 || !d_kp || !d_desc || !d_counts || !d_matches || !d_nmatches) return VSF_ERR_INVALID_ARG;
  vsf_status st = validate_images(ctx, d_imgs, 2 * n_frames, image_stride, row_stride);
  if (st != VSF_OK) return st;
  VSF_HIP(hipSetDevice(ctx->device));
  st = ensure_match_buffers(ctx, n_frames, ctx->p.max_keypoints);
  if (st != VSF_OK) return st;
  InputEventScope input(ctx);
  const VsfImages im{d_imgs, image_stride, row_stride, 2 * n_frames};
  const size_t set_stride = (size_t)ctx->p.max_keypoints * VSF_DESC_BYTES;
  st = run_chunked(ctx, n_frames, [&](hipStream_t s, int fa, int nf) {
    extract_on(ctx, s, im, 2 * fa, 2 * nf, d_kp, d_desc, d_counts, true);
    match_on(ctx, s, d_desc, d_counts, set_stride, nullptr, nullptr, fa, nf, ctx->m_idx2, ctx->m_dist2, d_matches,
             d_nmatches);
  });
  if (st != VSF_OK) return st;
  ctx->last_images = im;
  ctx->last_valid = true;
  VSF_STICKY();
  return VSF_OK;
}

// ---------------- reference steps between matcher and outputs ----------------

vsf_status vsf_remove_ambig_stereo_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const uint8_t* d_desc,
                                             const vsf_dmatch* d_matches, const int32_t* d_nmatches, int n_frames,
                                             const float* F, float thr_in, const float* d_thr_override,
                                             float* d_means, float* d_thr, vsf_keypoint* d_kp_out,
                                             uint8_t* d_desc_out, int32_t* d_counts_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_desc || !d_matches || !d_nmatches || n_frames < 1 || !F || !d_means || !d_kp_out ||
      !d_desc_out || !d_counts_out || (!d_thr_override && !d_thr))
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const size_t K = (size_t)ctx->p.max_keypoints;
  {
    vsf_status st = ensure_residual_buffers(ctx, n_frames);
    if (st != VSF_OK) return st;
  }
  vsf_launch_stereo_filter(d_kp, d_desc, d_matches, d_nmatches, n_frames, (int)K, nullptr, F, ctx->p.residual_order, d_thr_override, thr_in,
                           ctx->f_residual, d_means, d_thr, d_kp_out, d_desc_out, d_counts_out, ctx->stream);
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_stereo_residuals_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const vsf_dmatch* d_matches,
                                          const int32_t* d_nmatches, int n_frames, const float* F, float* d_means) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_matches || !d_nmatches || n_frames < 1 || !F || !d_means) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_status st = ensure_residual_buffers(ctx, n_frames);
  if (st != VSF_OK) return st;
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_stereo_residuals(d_kp, d_matches, d_nmatches, n_frames, ctx->p.max_keypoints, nullptr, F, ctx->p.residual_order, ctx->f_residual,
                                d_means, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_stereo_thresholds_dev(vsf_ctx* ctx, const float* d_means, int n, float* d_thr_state, float* d_thr) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_means || n < 1 || !d_thr_state || !d_thr) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_stereo_thresholds(d_means, n, d_thr_state, d_thr, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_stereo_filter_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const uint8_t* d_desc,
                                       const vsf_dmatch* d_matches, const int32_t* d_nmatches, int n_frames,
                                       const float* d_thr, vsf_keypoint* d_kp_out, uint8_t* d_desc_out,
                                       int32_t* d_counts_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_desc || !d_matches || !d_nmatches || n_frames < 1 || !d_thr || !d_kp_out || !d_desc_out ||
      !d_counts_out)
    return VSF_ERR_INVALID_ARG;
  if (n_frames > ctx->f_frames || !ctx->f_residual) return VSF_ERR_INVALID_ARG;  // no residuals of such a batch
  VSF_HIP(hipSetDevice(ctx->device));
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_stereo_filter_only(d_kp, d_desc, d_matches, d_nmatches, n_frames, ctx->p.max_keypoints, ctx->f_residual,
                                  d_thr, d_kp_out, d_desc_out, d_counts_out, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_vision_features_batch_dev(vsf_ctx* ctx, const vsf_calibration* calib, const vsf_keypoint* d_kp,
                                         const uint8_t* d_desc, const int32_t* d_counts, int n_frames,
                                         vsf_vision_feature* d_features, int32_t* d_nfeatures, int32_t* d_npoints) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !calib || !d_kp || !d_desc || !d_counts || n_frames < 1 || !d_features || !d_nfeatures)
    return VSF_ERR_INVALID_ARG;
  if (calib->triangulate_rows != 0 && calib->triangulate_rows != 4 && calib->triangulate_rows != 6)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const size_t K = (size_t)ctx->p.max_keypoints;
  {
    vsf_status st0 = ensure_vision_buffers(ctx, n_frames);
    if (st0 != VSF_OK) return st0;
  }
  // Calculate3DPoints: best_percent_ forced to 1.0 (cc:129-132)
  vsf_status st = vsf_feature_matches_batch_dev(ctx, d_desc, d_counts, K * VSF_DESC_BYTES, ctx->v_sets,
                                                ctx->v_sets + ctx->v_frames, n_frames, 1.0f, ctx->v_pairs, ctx->v_npairs);
  if (st != VSF_OK) return st;
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_vision_features(d_kp, d_counts, ctx->v_pairs, ctx->v_npairs, n_frames, (int)K, *calib, d_features,
                               d_nfeatures, d_npoints, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

size_t vsf_packed_outputs_capacity(const vsf_ctx* ctx, int n_frames, int n_pairs) {
  if (!ctx || n_frames < 0 || n_pairs < 0) return 0;
  const size_t K = (size_t)ctx->p.max_keypoints;
  return 16 + 4 * ((size_t)n_frames + n_pairs) + (size_t)n_frames * K * sizeof(vsf_vision_feature) +
         (size_t)n_pairs * K * sizeof(vsf_feature_match);
}

vsf_status vsf_pack_outputs_dev(vsf_ctx* ctx, const vsf_vision_feature* d_features, const int32_t* d_nfeatures,
                                int n_frames, const uint64_t* d_pairs, const int32_t* d_npairs, int n_pairs,
                                uint8_t* d_payload, size_t payload_cap) {
  VsfErrorScope scope_(ctx);
  if (!ctx || n_frames < 0 || n_pairs < 0 || n_frames + n_pairs < 1 || (n_frames > 0 && (!d_features || !d_nfeatures)) ||
      (n_pairs > 0 && (!d_pairs || !d_npairs)) || !d_payload || ((uintptr_t)d_payload & 3) ||
      payload_cap < 16 + 4 * ((size_t)n_frames + n_pairs))
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const int n = n_frames + n_pairs;
  {
    vsf_status st0 = ensure_pack_buffers(ctx, n);
    if (st0 != VSF_OK) return st0;
  }
  const uint32_t cap = (uint32_t)std::min<size_t>(payload_cap, 0xFFFFFFFCu);
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 2);
    vsf_launch_pack_outputs(d_features, d_nfeatures, n_frames, d_pairs, d_npairs, n_pairs, ctx->p.max_keypoints,
                            d_payload, cap, ctx->pk_offsets, ctx->d_status, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_bayer_bg_to_gray_batch_dev(vsf_ctx* ctx, const uint8_t* d_src, int n_images, int width, int height,
                                          size_t src_image_stride, size_t src_row_stride, uint8_t* d_dst,
                                          size_t dst_image_stride, size_t dst_row_stride) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_src || !d_dst || n_images < 1 || width < 1 || height < 1 || width > 16384 || height > 65535 ||
      n_images > 65535)
    return VSF_ERR_INVALID_ARG;
  if (((uintptr_t)d_src & 3) || ((uintptr_t)d_dst & 3) || (src_image_stride & 3) || (src_row_stride & 3) ||
      (dst_image_stride & 3) || (dst_row_stride & 3) || src_row_stride < (size_t)width ||
      dst_row_stride < (size_t)((width + 3) & ~3) || src_row_stride > 0x7FFFFFFF || dst_row_stride > 0x7FFFFFFF ||
      src_image_stride < src_row_stride * (size_t)height || dst_image_stride < dst_row_stride * (size_t)height)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_launch_bayer_bg_gray(d_src, n_images, width, height, src_image_stride, (int)src_row_stride, d_dst,
                           dst_image_stride, (int)dst_row_stride, ctx->stream);
  VSF_STICKY();
  // a pipelined extract that follows (vsf_set_pipeline) builds its pyramid off this stream: give it something to wait for
  if (!ctx->ev_ingest_done) VSF_HIP(hipEventCreateWithFlags(&ctx->ev_ingest_done, hipEventDisableTiming));
  VSF_HIP(hipEventRecord(ctx->ev_ingest_done, ctx->stream));
  ctx->ingest_done_valid = true;
  return VSF_OK;
}

vsf_status vsf_jpeg_decode_gray_batch(vsf_ctx* ctx, const uint8_t* const* jpeg, const size_t* nbytes, int n_images,
                                      int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                      size_t dst_row_stride) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !jpeg || !nbytes || n_images < 1 || n_images > 65535 || width < 1 || height < 1 || width > 65535 ||
      height > 65535 || !d_dst)
    return VSF_ERR_INVALID_ARG;
  if (((uintptr_t)d_dst & 3) || (dst_image_stride & 3) || (dst_row_stride & 3) || dst_row_stride < (size_t)width ||
      dst_row_stride > 0x7FFFFFFF || dst_image_stride < dst_row_stride * (size_t)height)
    return VSF_ERR_INVALID_ARG;
  for (int i = 0; i < n_images; i++)
    if (!jpeg[i] || nbytes[i] < 4 || nbytes[i] > 0x40000000u) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VsfJpegPlan plan;
  vsf_status st = vsf_jpeg_plan(jpeg, nbytes, n_images, width, height, ctx->tuning.jpeg_serial != 0, &plan);
  if (st != VSF_OK) return st;
  const int b = ctx->jp_flip;
  ctx->jp_flip ^= 1;
  if (!ctx->jp_copied[b]) VSF_HIP(hipEventCreateWithFlags(&ctx->jp_copied[b], hipEventDisableTiming));
  if (plan.total > ctx->jp_cap[b]) {
    VSF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->jp_host[b]) hipHostFree(ctx->jp_host[b]);
    hipFree(ctx->jp_dev[b]);
    ctx->jp_host[b] = ctx->jp_dev[b] = nullptr;
    ctx->jp_cap[b] = 0;
    const size_t cap = plan.total + plan.total / 4 + 4096;
    VSF_HIP(hipHostMalloc((void**)&ctx->jp_host[b], cap, hipHostMallocDefault));
    VSF_HIP(hipMalloc((void**)&ctx->jp_dev[b], cap));
    ctx->jp_cap[b] = cap;
  } else {
    // the upload of the call before the previous one has left this staging buffer (long ago: the previous call's
    // decode is what may still be running, out of the OTHER buffer)
    VSF_HIP(hipEventSynchronize(ctx->jp_copied[b]));
  }
  // files without restart intervals (what a camera driver writes): self-synchronising parallel decode; it needs the
  // de-stuffed streams and the luminance coefficients in HBM
  const size_t coef_stride = (size_t)plan.max_luma_blocks * 64 * sizeof(int16_t);
  if (plan.n_par + plan.n_prog > 0) {
    const size_t clean_need = plan.n_par > 0 ? vsf_jpeg_clean_bytes(plan.total - plan.off_stream, plan.n_par) : 0,
                 coef_need = (size_t)(plan.n_par + plan.n_prog) * coef_stride + vsf_jpeg_prog_huff_bytes(plan.n_prog_huff);
    // (the expanded Huffman tables of progressive scans live behind the coefficients)
    if (clean_need > ctx->jp_clean_cap || coef_need > ctx->jp_coef_cap) {
      VSF_HIP(hipStreamSynchronize(ctx->stream));
      if (clean_need > ctx->jp_clean_cap) {
        hipFree(ctx->jp_clean);
        ctx->jp_clean = nullptr;
        ctx->jp_clean_cap = 0;
        VSF_HIP(hipMalloc((void**)&ctx->jp_clean, clean_need + clean_need / 4));
        ctx->jp_clean_cap = clean_need + clean_need / 4;
      }
      if (coef_need > ctx->jp_coef_cap) {
        hipFree(ctx->jp_coef);
        ctx->jp_coef = nullptr;
        ctx->jp_coef_cap = 0;
        VSF_HIP(hipMalloc((void**)&ctx->jp_coef, coef_need + coef_need / 4));
        ctx->jp_coef_cap = coef_need + coef_need / 4;
      }
    }
  }
  vsf_jpeg_fill(plan, jpeg, n_images, ctx->jp_host[b]);  // the one pass over the compressed bytes on the host
  VSF_HIP(hipMemcpyAsync(ctx->jp_dev[b], ctx->jp_host[b], plan.total, hipMemcpyHostToDevice, ctx->stream));
  VSF_HIP(hipEventRecord(ctx->jp_copied[b], ctx->stream));
  if (plan.n_prog > ctx->jp_flags_cap) {  // (no wait: the outgrown buffer is retired)
    vsf_status gs = grow_scratch(ctx, ctx->jp_flags, (size_t)plan.n_prog * sizeof(int32_t));
    if (gs != VSF_OK) return gs;
    ctx->jp_flags_cap = plan.n_prog;
  }
  vsf_launch_jpeg_decode(ctx->jp_dev[b], plan.off_images, plan.off_index, plan.off_tables, plan.off_scans, plan.off_prog_huff,
                         plan.off_stream, plan.total, plan.n_par, plan.n_prog, plan.n_prog_huff,
                         reinterpret_cast<uint8_t*>(ctx->jp_coef) + (size_t)(plan.n_par + plan.n_prog) * coef_stride,
                         n_images - plan.n_par - plan.n_prog, plan.max_luma_blocks, plan.max_slots, width, height, ctx->jp_clean, ctx->jp_coef,
                         coef_stride, d_dst, dst_image_stride, (int)dst_row_stride, ctx->d_status, ctx->stream,
                         ctx->tuning.jpeg_serial != 0, ctx->jp_flags);
  VSF_STICKY();
  if (!ctx->ev_ingest_done) VSF_HIP(hipEventCreateWithFlags(&ctx->ev_ingest_done, hipEventDisableTiming));
  VSF_HIP(hipEventRecord(ctx->ev_ingest_done, ctx->stream));  // (a pipelined extract waits for its images, as after the Bayer step)
  ctx->ingest_done_valid = true;
  return VSF_OK;
}

// cv::imdecode(IMREAD_GRAYSCALE) for grayscale PNG files (slam_frontend_main.cc:99-100): chunks and CRCs on the host, inflate +
// filters on the device (k_png.hip).  Same staging and the same asynchronous contract as the JPEG entry point.
vsf_status vsf_png_decode_gray_batch(vsf_ctx* ctx, const uint8_t* const* png, const size_t* nbytes, int n_images,
                                     int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                     size_t dst_row_stride) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !png || !nbytes || n_images < 1 || n_images > 65535 || width < 1 || height < 1 || width > 65535 ||
      height > 65535 || !d_dst)
    return VSF_ERR_INVALID_ARG;
  if (((uintptr_t)d_dst & 3) || (dst_image_stride & 3) || (dst_row_stride & 3) || dst_row_stride < (size_t)width ||
      dst_row_stride > 0x7FFFFFFF || dst_image_stride < dst_row_stride * (size_t)height)
    return VSF_ERR_INVALID_ARG;
  for (int i = 0; i < n_images; i++)
    if (!png[i] || nbytes[i] < 8 || nbytes[i] > 0x40000000u) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VsfPngPlan plan;
  vsf_status st = vsf_png_plan(png, nbytes, n_images, width, height, &plan);
  if (st != VSF_OK) return st;
  const int b = ctx->jp_flip;
  ctx->jp_flip ^= 1;
  if (!ctx->jp_copied[b]) VSF_HIP(hipEventCreateWithFlags(&ctx->jp_copied[b], hipEventDisableTiming));
  if (plan.total > ctx->jp_cap[b]) {
    VSF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->jp_host[b]) hipHostFree(ctx->jp_host[b]);
    hipFree(ctx->jp_dev[b]);
    ctx->jp_host[b] = ctx->jp_dev[b] = nullptr;
    ctx->jp_cap[b] = 0;
    const size_t cap = plan.total + plan.total / 4 + 4096;
    VSF_HIP(hipHostMalloc((void**)&ctx->jp_host[b], cap, hipHostMallocDefault));
    VSF_HIP(hipMalloc((void**)&ctx->jp_dev[b], cap));
    ctx->jp_cap[b] = cap;
  } else {
    VSF_HIP(hipEventSynchronize(ctx->jp_copied[b]));  // (the upload of the call before the previous one has left this buffer)
  }
  const size_t filtered_need = plan.filtered_stride * (size_t)n_images;
  if (filtered_need > ctx->png_filtered_cap) {  // (no wait: the outgrown buffer is retired)
    vsf_status gs = grow_scratch(ctx, ctx->png_filtered, filtered_need + filtered_need / 4);
    if (gs != VSF_OK) return gs;
    ctx->png_filtered_cap = filtered_need + filtered_need / 4;
  }
  if (n_images > ctx->png_file_status_cap) {
    vsf_status gs = grow_scratch(ctx, ctx->png_file_status, (size_t)n_images * sizeof(int32_t));
    if (gs != VSF_OK) return gs;
    ctx->png_file_status_cap = n_images;
  }
  vsf_png_fill(plan, png, n_images, ctx->jp_host[b]);
  VSF_HIP(hipMemcpyAsync(ctx->jp_dev[b], ctx->jp_host[b], plan.total, hipMemcpyHostToDevice, ctx->stream));
  VSF_HIP(hipEventRecord(ctx->jp_copied[b], ctx->stream));
  vsf_launch_png_decode(ctx->jp_dev[b], plan.off_images, plan.off_pieces, plan.off_tables, plan.off_stream, n_images, width, height, ctx->png_filtered,
                        plan.filtered_stride, ctx->png_file_status, d_dst, dst_image_stride, (int)dst_row_stride,
                        ctx->d_status, plan.any_general, plan.any_rgb, ctx->stream);
  VSF_STICKY();
  if (!ctx->ev_ingest_done) VSF_HIP(hipEventCreateWithFlags(&ctx->ev_ingest_done, hipEventDisableTiming));
  VSF_HIP(hipEventRecord(ctx->ev_ingest_done, ctx->stream));
  ctx->ingest_done_valid = true;
  return VSF_OK;
}

// cv::imdecode(msg.data, IMREAD_GRAYSCALE) as the reference calls it (slam_frontend_main.cc:99-100): whatever the payload
// is.  Files are told apart by their first bytes (as cv::imdecode's findDecoder does: signature match) and handed, run by run
// of one format, to the JPEG or the PNG entry point; image i lands at d_dst + i * dst_image_stride either way.
vsf_status vsf_imdecode_gray_batch(vsf_ctx* ctx, const uint8_t* const* files, const size_t* nbytes, int n_images,
                                   int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                   size_t dst_row_stride) {
  if (!ctx || !files || !nbytes || n_images < 1 || !d_dst) return VSF_ERR_INVALID_ARG;
  static const uint8_t kPng[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  auto kind = [&](int i) -> int {  // 0 JPEG (SOI), 1 PNG, -1 neither
    if (!files[i]) return -1;
    if (nbytes[i] >= 8 && std::memcmp(files[i], kPng, 8) == 0) return 1;
    if (nbytes[i] >= 3 && files[i][0] == 0xFF && files[i][1] == 0xD8 && files[i][2] == 0xFF) return 0;
    return -1;
  };
  for (int i = 0; i < n_images; i++)
    if (kind(i) < 0) return VSF_ERR_UNSUPPORTED;  // (imdecode's other formats -- BMP, TIFF, WebP ... -- are not built)
  for (int i0 = 0; i0 < n_images;) {
    const int k = kind(i0);
    int i1 = i0 + 1;
    while (i1 < n_images && kind(i1) == k) ++i1;
    uint8_t* dst = d_dst + (size_t)i0 * dst_image_stride;
    const vsf_status st = k == 1 ? vsf_png_decode_gray_batch(ctx, files + i0, nbytes + i0, i1 - i0, width, height, dst, dst_image_stride, dst_row_stride)
                                 : vsf_jpeg_decode_gray_batch(ctx, files + i0, nbytes + i0, i1 - i0, width, height, dst, dst_image_stride, dst_row_stride);
    if (st != VSF_OK) return st;
    i0 = i1;
  }
  return VSF_OK;
}

vsf_status vsf_feature_matches_batch_dev(vsf_ctx* ctx, const uint8_t* d_desc, const int32_t* d_counts,
                                         size_t set_stride, const int32_t* d_q_set, const int32_t* d_t_set,
                                         int n_pairs, float best_percent, uint64_t* d_pairs, int32_t* d_npairs) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_pairs || !d_npairs || n_pairs < 1 || !(best_percent >= 0.f)) return VSF_ERR_INVALID_ARG;
  if (ctx->p.max_keypoints >= 65536) return VSF_ERR_UNSUPPORTED;  // (query, train) indices are packed 16 + 16 bit
  VSF_HIP(hipSetDevice(ctx->device));
  const size_t K = (size_t)ctx->p.max_keypoints;
  {
    vsf_status st0 = ensure_temporal_buffers(ctx, n_pairs);
    if (st0 != VSF_OK) return st0;
  }
  vsf_status st = vsf_match_batch_dev(ctx, d_desc, d_counts, set_stride, d_q_set, d_t_set, n_pairs, nullptr, nullptr,
                                      ctx->t_matches, ctx->t_nmatches);
  if (st != VSF_OK) return st;
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_sort_trim(ctx->t_matches, ctx->t_nmatches, n_pairs, (int)K, best_percent, nullptr, ctx->t_sortkeys,
                         d_pairs, d_npairs, ctx->stream, ctx->tuning.sort_serial != 0, ctx->tuning.lds_limit);
  }
  VSF_STICKY();
  return VSF_OK;
}

// ---------------- one submission per ObserveImage ----------------

size_t vsf_observe_capacity(const vsf_ctx* ctx, int frame_life) {
  if (!ctx || frame_life < 0 || frame_life + 1 > VSF_OBSERVE_MAX_PAIRS) return 0;
  const size_t K = (size_t)ctx->p.max_keypoints;
  return 64 + 4 * (size_t)((frame_life + 1 + 3) & ~3) + K * (28 + 28 + 32) + (size_t)(frame_life + 1) * K * 16;
}

static vsf_status ensure_observe(vsf_ctx* ctx, int frame_life) {
  vsf_ctx::Observe& o = ctx->ob;
  if (o.ring && o.frame_life == frame_life) return VSF_OK;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++)
    if (o.ex_stream[i]) VSF_HIP(hipStreamSynchronize(o.ex_stream[i]));
  float thr_state = 10000.0f;  // cc:353
  const bool had = o.floats != nullptr;
  if (had) VSF_HIP(hipMemcpy(&thr_state, o.floats + 2, sizeof(float), hipMemcpyDeviceToHost));
  free_observe(ctx);
  const size_t K = (size_t)ctx->p.max_keypoints, S = (size_t)frame_life + 2;
  o.slots = std::max(1, std::min(ctx->p.max_images / 2, VSF_OBSERVE_MAX_SLOTS));
  VSF_HIP(hipMalloc((void**)&o.ring, S * K * VSF_DESC_BYTES));
  VSF_HIP(hipMalloc((void**)&o.ring_counts, S * sizeof(int32_t)));
  VSF_HIP(hipMemset(o.ring_counts, 0, S * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.kpf, 2 * K * sizeof(vsf_keypoint)));
  VSF_HIP(hipMalloc((void**)&o.matches, VSF_OBSERVE_MAX_SLOTS * K * sizeof(vsf_dmatch)));
  VSF_HIP(hipMalloc((void**)&o.ints, 16 * sizeof(int32_t)));  // [0..5] raw stereo matches per slot, [8] features, [9] points
  VSF_HIP(hipMemset(o.ints, 0, 16 * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.floats, 4 * sizeof(float)));
  const float f4[4] = {0.f, 0.f, thr_state, 0.f};
  VSF_HIP(hipMemcpy(o.floats, f4, sizeof(f4), hipMemcpyHostToDevice));
  VSF_HIP(hipMalloc((void**)&o.features, K * sizeof(vsf_vision_feature)));
  VSF_HIP(hipMalloc((void**)&o.pairs, (size_t)(frame_life + 1) * K * 2 * sizeof(uint64_t)));
  VSF_HIP(hipMalloc((void**)&o.npairs, (size_t)(frame_life + 1) * sizeof(int32_t)));
  o.out_cap = vsf_observe_capacity(ctx, frame_life);
  for (int i = 0; i < o.slots; i++) {
    VSF_HIP(hipHostMalloc((void**)&o.h_img[i], 2 * ctx->st_img_stride, hipHostMallocMapped));
    VSF_HIP(hipHostMalloc((void**)&o.h_out[i], o.out_cap, hipHostMallocMapped));
    VSF_HIP(hipHostMalloc((void**)&o.h_meta[i], sizeof(vsf_ctx::ObserveMeta), hipHostMallocMapped));
    std::memset(o.h_meta[i], 0, sizeof(vsf_ctx::ObserveMeta));
    VSF_HIP(hipHostMalloc((void**)&o.h_status[i], sizeof(int32_t), hipHostMallocMapped));
    *o.h_status[i] = 0;
    // A stream per slot, each at a DIFFERENT stream priority (highest, default, lowest).  HIP multiplexes streams onto a few
    // hardware queues (round-robin at creation) and kernels of streams that share a queue run one after the other: with
    // streams of the default priority, one slot's stream landed on another's queue and its frames overlapped nothing
    // (kernel trace: 0.33 ms per frame, no better than one stream).  Streams of different priorities never share a queue,
    // so the chains of up to three frames -- ~25 small kernels each, bound by launch-to-launch latency -- run side by side.
    if (o.slots == 1) {
      o.ex_stream[i] = ctx->stream;
    } else {
      int prio_lo = 0, prio_hi = 0;
      VSF_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
      // (a fourth to sixth slot repeats the three priorities: it may share a hardware queue with an earlier slot -- then
      // those two take turns -- or land on a queue of its own)
      const int prio = i % 3 == 0 ? prio_hi : (i % 3 == 1 ? prio_lo : (prio_lo + prio_hi) / 2);
      VSF_HIP(hipStreamCreateWithPriority(&o.ex_stream[i], hipStreamNonBlocking, prio));
    }
    VSF_HIP(hipEventCreateWithFlags(&o.ev_done[i], hipEventDisableTiming));
  }
  o.frame_life = frame_life;
  // matcher scratch: pairs [0, frame_life] belong to the tail, pair frame_life + 1 + slot to the slot's stereo match
  vsf_status st = ensure_match_buffers(ctx, frame_life + 1 + VSF_OBSERVE_MAX_SLOTS, (int)K);
  if (st == VSF_OK) st = ensure_temporal_buffers(ctx, frame_life + 1);
  if (st == VSF_OK) st = ensure_residual_buffers(ctx, 1);
  return st;
}

vsf_status vsf_observe_reset(vsf_ctx* ctx) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++)
    if (ctx->ob.ex_stream[i]) VSF_HIP(hipStreamSynchronize(ctx->ob.ex_stream[i]));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  free_observe(ctx);
  return VSF_OK;
}

vsf_status vsf_observe_submit(vsf_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                              const vsf_calibration* calib, float best_percent, int frame_life, int64_t* ticket) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !left || !right || !calib || !ticket || !(best_percent >= 0.f) || frame_life < 0 ||
      frame_life + 1 > VSF_OBSERVE_MAX_PAIRS)
    return VSF_ERR_INVALID_ARG;
  *ticket = -1;
  if (w != ctx->p.width || h != ctx->p.height || stride < (size_t)w || ctx->p.max_images < 2) return VSF_ERR_INVALID_ARG;
  if (ctx->p.max_keypoints >= 65536) return VSF_ERR_UNSUPPORTED;
  if (calib->triangulate_rows != 0 && calib->triangulate_rows != 4 && calib->triangulate_rows != 6)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  if (ctx->ob.ring && ctx->ob.frame_life != frame_life)  // (re-sizing the window drops nothing that is still in flight)
    for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++)
      if (ctx->ob.ticket_of[i] >= 0) return VSF_ERR_INVALID_ARG;
  vsf_status st = ensure_observe(ctx, frame_life);
  if (st != VSF_OK) return st;
  vsf_ctx::Observe& o = ctx->ob;
  const int slot = (int)(o.next_ticket % o.slots);
  if (o.ticket_of[slot] >= 0) return VSF_ERR_INVALID_ARG;  // collect that frame first: its buffers are about to be reused
  hipStream_t ex = o.ex_stream[slot], s = ex;  // the frame's one stream
  const size_t K = (size_t)ctx->p.max_keypoints;
  const int Kc = (int)K;
  // ---- upload: rows into the slot's pinned staging at the device pitch, ONE copy command for both images ----
  const uint8_t* src[2] = {left, right};
  uint8_t* h_img = o.h_img[slot];
  for (int i = 0; i < 2; i++) {
    uint8_t* dst = h_img + (size_t)i * ctx->st_img_stride;
    if (stride == ctx->st_img_pitch) {  // the caller's rows already sit at the staging pitch: one copy per image
      std::memcpy(dst, src[i], (size_t)(h - 1) * stride + (size_t)w);
    } else {
      for (int y = 0; y < h; y++) std::memcpy(dst + (size_t)y * ctx->st_img_pitch, src[i] + (size_t)y * stride, (size_t)w);
    }
  }
  // (the slot's previous frame ran on this same stream: its tail has finished reading what the extraction now overwrites)
  uint8_t* d_img = ctx->st_img + (size_t)(2 * slot) * ctx->st_img_stride;
  VSF_HIP(hipMemcpyAsync(d_img, h_img, 2 * ctx->st_img_stride, hipMemcpyHostToDevice, ex));
  // ---- per-call parameters: written into pinned memory the kernels read directly ----
  const int n_past = (int)o.order.size(), n_pairs = n_past + 1, S = frame_life;
  vsf_ctx::ObserveMeta& M = *o.h_meta[slot];
  std::memcpy(M.F, calib->fundamental, sizeof(M.F));
  for (int p = 0; p < n_past; p++) {
    M.q_set[p] = o.order[p];  // oldest kept frame first: the order frame_list_ is walked in (cc:424)
    M.t_set[p] = S;
    M.best_percent[p] = best_percent;
  }
  M.q_set[n_past] = S + 1;  // Calculate3DPoints: GetFeatureMatches(right, left) with best_percent_ 1.0 (cc:129-132)
  M.t_set[n_past] = S;
  M.best_percent[n_past] = 1.0f;
  // ---- ExtractFeatures x 2 + GetMatches (cc:411-416), on the slot's stream and in the slot's buffers ----
  const VsfImages im{ctx->st_img, ctx->st_img_stride, ctx->st_img_pitch, 2 * (slot + 1)};
  vsf_keypoint* kp_raw = ctx->st_kp + (size_t)(2 * slot) * K;
  uint8_t* desc_raw = ctx->st_desc + (size_t)(2 * slot) * K * VSF_DESC_BYTES;
  int32_t* counts_raw = ctx->st_counts + 2 * slot;
  int32_t* status_word = ctx->d_status + 1 + slot;  // this frame's own (see vsf_ctx::d_status)
  extract_on(ctx, ex, im, 2 * slot, 2, ctx->st_kp, ctx->st_desc, ctx->st_counts, false, o.slots > 1 ? &o.side[slot] : nullptr,
             status_word);
  ctx->last_images = VsfImages{d_img, ctx->st_img_stride, ctx->st_img_pitch, 2};
  ctx->last_valid = true;
  int32_t* nmatches = o.ints + slot;
  vsf_dmatch* raw_matches = o.matches + (size_t)slot * K;
  {
    const size_t scratch = (size_t)(frame_life + 1 + slot) * K * 2;
    match_on(ctx, ex, desc_raw, counts_raw, K * VSF_DESC_BYTES, nullptr, nullptr, 0, 1, ctx->m_idx2 + scratch,
             ctx->m_dist2 + scratch, raw_matches, nmatches, status_word);
  }
  // the tails run in frame order: this frame's waits for the previous frame's (on another slot's stream)
  if (o.slots > 1 && o.next_ticket > 0) {
    const int prev = (int)((o.next_ticket - 1) % o.slots);
    if (o.done_valid[prev]) VSF_HIP(hipStreamWaitEvent(s, o.ev_done[prev], 0));
  }
  // ---- RemoveAmbigStereo (cc:417): the current frame lands in ring sets S (left) and S + 1 (right) ----
  float *means = o.floats, *thr = o.floats + 1, *thr_state = o.floats + 2;
  uint8_t* cur_desc = o.ring + (size_t)S * K * VSF_DESC_BYTES;
  int32_t* cur_counts = o.ring_counts + S;
  {
    StageTimer t(ctx, s, VSF_STAGE_TAIL, 3);
    vsf_launch_stereo_residuals(kp_raw, raw_matches, nmatches, 1, Kc, M.F, nullptr, ctx->p.residual_order, ctx->f_residual, means, s);
    vsf_launch_stereo_thresholds(means, 1, thr_state, thr, s);
    vsf_launch_stereo_filter_only(kp_raw, desc_raw, raw_matches, nmatches, 1, Kc, ctx->f_residual, thr, o.kpf, cur_desc,
                                  cur_counts, s);
  }
  // ---- GetFeatureMatches against every kept frame + the right->left matches of Calculate3DPoints: one matcher
  // launch, one sort launch (per-pair best_percent) ----
  {
    StageTimer t(ctx, s, VSF_STAGE_KNN2, 1);
    vsf_launch_knn2(o.ring, o.ring_counts, K * VSF_DESC_BYTES, M.q_set, M.t_set, n_pairs, Kc, ctx->m_idx2, ctx->m_dist2, s,
                    ctx->tuning.match_int8 != 0);
  }
  {
    StageTimer t(ctx, s, VSF_STAGE_RATIO, 1);
    vsf_launch_ratio_compact(o.ring_counts, M.q_set, M.t_set, n_pairs, Kc, ctx->m_idx2, ctx->m_dist2, ctx->p.ratio_num,
                             ctx->p.ratio_shift, ctx->t_matches, ctx->t_nmatches, status_word, s);
  }
  {
    StageTimer t(ctx, s, VSF_STAGE_TAIL, 3);
    vsf_launch_sort_trim(ctx->t_matches, ctx->t_nmatches, n_pairs, Kc, best_percent, M.best_percent, ctx->t_sortkeys,
                         o.pairs, o.npairs, s, ctx->tuning.sort_serial != 0, ctx->tuning.lds_limit);
    // ---- Calculate3DPoints + VisionFeature + UndistortFeaturePoints (cc:437-443) ----
    int32_t *nfeat = o.ints + 8, *npoints = o.ints + 9;
    vsf_launch_vision_features(o.kpf, cur_counts, o.pairs + (size_t)n_past * K * 2, o.npairs + n_past, 1, Kc, *calib,
                               o.features, nfeat, npoints, s);
    // ---- the compact result into pinned memory; the filtered left frame into its ring slot (cc:467-470) ----
    int ring_slot;
    if (frame_life == 0) {
      ring_slot = S + 1;  // nothing is kept: park it on the right frame's set
    } else if (n_past >= frame_life) {
      ring_slot = o.order.front();
    } else {
      ring_slot = n_past;
      for (int c = 0; c < frame_life; c++)
        if (std::find(o.order.begin(), o.order.end(), c) == o.order.end()) {
          ring_slot = c;
          break;
        }
    }
    VsfObserveArgs a;
    a.n_pairs = n_pairs;
    a.max_rows = Kc;
    a.counts_raw = counts_raw;
    a.nmatches = nmatches;
    a.counts_f = cur_counts;
    a.npoints = npoints;
    a.means = means;
    a.thr = thr;
    a.thr_state = thr_state;
    a.features = o.features;
    a.kp_f = o.kpf;
    a.desc_f = cur_desc;
    a.pairs = o.pairs;
    a.npairs = o.npairs;
    a.ring_desc = o.ring + (size_t)ring_slot * K * VSF_DESC_BYTES;
    a.ring_count = o.ring_counts + ring_slot;
    a.out = o.h_out[slot];
    a.out_cap = (uint32_t)std::min<size_t>(o.out_cap, 0xFFFFFFF0u);
    vsf_launch_observe_pack(a, s);
    if (frame_life > 0) {
      if (n_past >= frame_life) o.order.erase(o.order.begin());
      o.order.push_back(ring_slot);
    }
  }
  // the frame's own status word (everything the frame ran wrote into it, nothing else did), then "this frame is done"
  VSF_HIP(hipMemcpyAsync(o.h_status[slot], status_word, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  VSF_HIP(hipMemsetAsync(status_word, 0, sizeof(int32_t), s));
  VSF_HIP(hipEventRecord(o.ev_done[slot], s));
  o.done_valid[slot] = true;
  VSF_STICKY();
  o.ticket_of[slot] = o.next_ticket;
  *ticket = o.next_ticket++;
  return VSF_OK;
}

vsf_status vsf_observe_collect(vsf_ctx* ctx, int64_t ticket, uint8_t* out, size_t cap, size_t* out_bytes) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !out || !out_bytes || ticket < 0) return VSF_ERR_INVALID_ARG;
  *out_bytes = 0;
  vsf_ctx::Observe& o = ctx->ob;
  const int slot = (int)(ticket % std::max(o.slots, 1));
  if (!o.ring || o.ticket_of[slot] != ticket) return VSF_ERR_INVALID_ARG;
  // frames leave in the order they entered (the host's bookkeeping is sequential): an older frame must be collected first
  for (int i = 0; i < o.slots; i++)
    if (o.ticket_of[i] >= 0 && o.ticket_of[i] < ticket) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipEventSynchronize(o.ev_done[slot]));
  o.ticket_of[slot] = -1;
  vsf_status st = VSF_OK;
  if (*o.h_status[slot] & 1) st = VSF_ERR_CAPACITY;
  const uint32_t* hdr = reinterpret_cast<const uint32_t*>(o.h_out[slot]);
  if (hdr[0] != 0x4F465356u) return VSF_ERR_HIP;
  const size_t total = hdr[3];
  *out_bytes = total;
  if (hdr[11] != 0 || total > cap) return VSF_ERR_CAPACITY;
  std::memcpy(out, o.h_out[slot], total);
  return st;
}

vsf_status vsf_observe_stereo(vsf_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                              const vsf_calibration* calib, float best_percent, int frame_life, uint8_t* out,
                              size_t cap, size_t* out_bytes) {
  VsfErrorScope scope_(ctx);
  if (!out || !out_bytes) return VSF_ERR_INVALID_ARG;
  *out_bytes = 0;
  int64_t ticket = -1;
  const vsf_status st = vsf_observe_submit(ctx, left, right, w, h, stride, calib, best_percent, frame_life, &ticket);
  if (st != VSF_OK) return st;
  return vsf_observe_collect(ctx, ticket, out, cap, out_bytes);
}

// ---------------- host-pointer entry points ----------------

static vsf_status upload_image(vsf_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, int slot) {
  if (!img || w != ctx->p.width || h != ctx->p.height || stride < (size_t)w) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipMemcpy2DAsync(ctx->st_img + (size_t)slot * ctx->st_img_stride, ctx->st_img_pitch, img, stride, (size_t)w,
                           (size_t)h, hipMemcpyHostToDevice, ctx->stream));
  return VSF_OK;
}

vsf_status vsf_extract(vsf_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, vsf_keypoint* kp_out,
                       uint8_t* desc_out, int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || cap < 0 || (cap > 0 && (!kp_out || !desc_out))) return VSF_ERR_INVALID_ARG;
  *n_out = 0;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_status st = upload_image(ctx, img, w, h, stride, 0);
  if (st != VSF_OK) return st;
  VsfImages im{ctx->st_img, ctx->st_img_stride, ctx->st_img_pitch, 1};
  st = extract_async(ctx, im, ctx->st_kp, ctx->st_desc, ctx->st_counts);
  if (st != VSF_OK) return st;
  int32_t n = 0;
  VSF_HIP(hipMemcpyAsync(&n, ctx->st_counts, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  st = check_status_word(ctx);  // synchronises
  *n_out = n;
  const int m = std::min(std::min((int)n, cap), ctx->p.max_keypoints);
  if (m > 0) {
    VSF_HIP(hipMemcpy(kp_out, ctx->st_kp, (size_t)m * sizeof(vsf_keypoint), hipMemcpyDeviceToHost));
    VSF_HIP(hipMemcpy(desc_out, ctx->st_desc, (size_t)m * VSF_DESC_BYTES, hipMemcpyDeviceToHost));
  }
  if (st == VSF_OK && n > m) st = VSF_ERR_CAPACITY;
  return st;
}

vsf_status vsf_extract_pair(vsf_ctx* ctx, const uint8_t* img0, const uint8_t* img1, int w, int h, size_t stride,
                            vsf_keypoint* kp0, uint8_t* desc0, int* n0, vsf_keypoint* kp1, uint8_t* desc1, int* n1,
                            int cap) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n0 || !n1 || cap < 0 || (cap > 0 && (!kp0 || !desc0 || !kp1 || !desc1))) return VSF_ERR_INVALID_ARG;
  *n0 = *n1 = 0;
  if (ctx->p.max_images < 2) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_status st = upload_image(ctx, img0, w, h, stride, 0);
  if (st == VSF_OK) st = upload_image(ctx, img1, w, h, stride, 1);
  if (st != VSF_OK) return st;
  VsfImages im{ctx->st_img, ctx->st_img_stride, ctx->st_img_pitch, 2};
  st = extract_async(ctx, im, ctx->st_kp, ctx->st_desc, ctx->st_counts);
  if (st != VSF_OK) return st;
  int32_t n[2] = {0, 0};
  VSF_HIP(hipMemcpyAsync(n, ctx->st_counts, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  st = check_status_word(ctx);  // synchronises
  *n0 = n[0];
  *n1 = n[1];
  const size_t K = (size_t)ctx->p.max_keypoints;
  vsf_keypoint* kps[2] = {kp0, kp1};
  uint8_t* descs[2] = {desc0, desc1};
  for (int i = 0; i < 2; i++) {
    const int m = std::min(std::min((int)n[i], cap), ctx->p.max_keypoints);
    if (m > 0) {
      VSF_HIP(hipMemcpy(kps[i], ctx->st_kp + i * K, (size_t)m * sizeof(vsf_keypoint), hipMemcpyDeviceToHost));
      VSF_HIP(hipMemcpy(descs[i], ctx->st_desc + i * K * VSF_DESC_BYTES, (size_t)m * VSF_DESC_BYTES,
                        hipMemcpyDeviceToHost));
    }
    if (st == VSF_OK && n[i] > m) st = VSF_ERR_CAPACITY;
  }
  return st;
}

vsf_status vsf_fast_detect(vsf_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, int threshold, int nms,
                           vsf_keypoint* kp_out, int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || cap < 0 || (cap > 0 && !kp_out)) return VSF_ERR_INVALID_ARG;
  *n_out = 0;
  VSF_HIP(hipSetDevice(ctx->device));
  if (threshold < 0) {
    threshold = ctx->p.fast_detector_threshold;
  }
  threshold = std::min(std::max(threshold, 0), 255);
  const bool want_nms = nms != 0;
  // The candidate segments are sized for the NMS case (no two 8-adjacent keypoints); without NMS every pixel
  // may be a corner, so that geometry is (re)built with full-density segments.
  if (!ctx->dfast.ready || ctx->fast_nms != want_nms) {
    VSF_HIP(hipStreamSynchronize(ctx->stream));
    free_devset(&ctx->dfast);
    ctx->fast = Geometry();
    if (!build_geometry(ctx->p, false, want_nms, &ctx->fast)) return VSF_ERR_INVALID_ARG;
    vsf_status st0 = alloc_devset(ctx, ctx->fast, &ctx->dfast, false, 1);
    if (st0 != VSF_OK) return st0;
    ctx->fast_nms = want_nms;
  }
  vsf_status st = upload_image(ctx, img, w, h, stride, 0);
  if (st != VSF_OK) return st;
  VsfImages im{ctx->st_img, ctx->st_img_stride, ctx->st_img_pitch, 1};
  // Output capacity: grow a private buffer if the caller's cap exceeds the extract staging.
  const int kcap = ctx->p.max_keypoints;
  vsf_launch_fast(ctx->dfast.d, ctx->fast.g, im, threshold, want_nms ? 1 : 0, ctx->stream);
  vsf_keypoint* d_out = ctx->st_kp;
  vsf_keypoint* big = nullptr;
  int outcap = kcap;
  if (cap > kcap) {
    VSF_HIP(hipMalloc((void**)&big, (size_t)cap * sizeof(vsf_keypoint)));
    d_out = big;
    outcap = cap;
  }
  vsf_launch_fast_emit(ctx->dfast.d, ctx->fast.g, 1, outcap, d_out, ctx->st_counts, ctx->stream);
  int32_t n = 0;
  hipError_t e = hipMemcpyAsync(&n, ctx->st_counts, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) {
    st = check_status_word(ctx);
    if (st == VSF_ERR_HIP) e = (hipError_t)ctx->last_hip;
  }
  if (e == hipSuccess) {
    *n_out = n;
    const int m = std::min(std::min((int)n, cap), outcap);
    if (m > 0) e = hipMemcpy(kp_out, d_out, (size_t)m * sizeof(vsf_keypoint), hipMemcpyDeviceToHost);
    if (e == hipSuccess) st = n > m ? VSF_ERR_CAPACITY : VSF_OK;  // the status word only reflects `outcap`
  }
  if (big) hipFree(big);
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    return VSF_ERR_HIP;
  }
  return st;
}

static vsf_status match_host(vsf_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2,
                             int32_t* dist2, vsf_dmatch* out, int cap, int* n_out) {
  if (nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || nt >= (1 << 20)) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  if (n_out) *n_out = 0;
  if (nq == 0) return VSF_OK;
  const int rows = std::max(std::max(nq, nt), 1);
  vsf_status st = ensure_match_host_staging(ctx, rows);
  if (st != VSF_OK) return st;
  st = ensure_match_buffers(ctx, 1, ctx->mh_rows);
  if (st != VSF_OK) return st;
  const int R = ctx->mh_rows;
  const size_t set_stride = (size_t)R * VSF_DESC_BYTES;
  const int32_t counts[2] = {nq, nt};
  VSF_HIP(hipMemcpyAsync(ctx->mh_counts, counts, sizeof(counts), hipMemcpyHostToDevice, ctx->stream));
  VSF_HIP(hipMemcpyAsync(ctx->mh_desc, q, (size_t)nq * VSF_DESC_BYTES, hipMemcpyHostToDevice, ctx->stream));
  if (nt > 0)
    VSF_HIP(hipMemcpyAsync(ctx->mh_desc + set_stride, t, (size_t)nt * VSF_DESC_BYTES, hipMemcpyHostToDevice,
                           ctx->stream));
  // m_idx2/m_dist2 are laid out [pair][m_rows][2]; the kernels are given the same row capacity.
  vsf_launch_knn2(ctx->mh_desc, ctx->mh_counts, set_stride, nullptr, nullptr, 1, R, ctx->m_idx2, ctx->m_dist2,
                  ctx->stream, ctx->tuning.match_int8 != 0);
  if (out) {
    vsf_launch_ratio_compact(ctx->mh_counts, nullptr, nullptr, 1, R, ctx->m_idx2, ctx->m_dist2, ctx->p.ratio_num,
                             ctx->p.ratio_shift, ctx->mh_matches, ctx->mh_nmatches, ctx->d_status, ctx->stream);
  }
  VSF_STICKY();
  if (idx2) {
    VSF_HIP(hipMemcpyAsync(idx2, ctx->m_idx2, (size_t)nq * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    VSF_HIP(hipMemcpyAsync(dist2, ctx->m_dist2, (size_t)nq * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  int32_t nm = 0;
  if (out) VSF_HIP(hipMemcpyAsync(&nm, ctx->mh_nmatches, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  if (out) {
    if (n_out) *n_out = nm;
    const int m = std::min((int)nm, cap);
    if (m > 0) VSF_HIP(hipMemcpy(out, ctx->mh_matches, (size_t)m * sizeof(vsf_dmatch), hipMemcpyDeviceToHost));
    if (nm > cap) return VSF_ERR_CAPACITY;
  }
  return VSF_OK;
}

vsf_status vsf_get_matches_multi(vsf_ctx* ctx, const uint8_t* const* q, const int* nq, int n_sets, const uint8_t* t,
                                 int nt, vsf_dmatch* out, int cap_per_set, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || n_sets < 1 || !q || !nq || !n_out || cap_per_set < 0 || (cap_per_set > 0 && !out) || nt < 0 ||
      (nt > 0 && !t) || nt >= (1 << 20))
    return VSF_ERR_INVALID_ARG;
  int rows = std::max(nt, 1);
  for (int s = 0; s < n_sets; s++) {
    if (nq[s] < 0 || (nq[s] > 0 && !q[s])) return VSF_ERR_INVALID_ARG;
    rows = std::max(rows, nq[s]);
    n_out[s] = 0;
  }
  VSF_HIP(hipSetDevice(ctx->device));
  if (n_sets > ctx->mm_sets || rows > ctx->mm_rows) {
    const int S = std::max(n_sets, ctx->mm_sets), R = std::max(rows, ctx->mm_rows);
    VSF_HIP(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->mm_desc);
    hipFree(ctx->mm_counts);
    hipFree(ctx->mm_matches);
    hipFree(ctx->mm_nmatches);
    ctx->mm_desc = nullptr;
    ctx->mm_counts = nullptr;
    ctx->mm_matches = nullptr;
    ctx->mm_nmatches = nullptr;
    ctx->mm_sets = ctx->mm_rows = 0;
    VSF_HIP(hipMalloc((void**)&ctx->mm_desc, (size_t)(S + 1) * R * VSF_DESC_BYTES));
    VSF_HIP(hipMalloc((void**)&ctx->mm_counts, (size_t)(3 * S + 1) * sizeof(int32_t)));
    VSF_HIP(hipMalloc((void**)&ctx->mm_matches, (size_t)S * R * sizeof(vsf_dmatch)));
    VSF_HIP(hipMalloc((void**)&ctx->mm_nmatches, (size_t)S * sizeof(int32_t)));
    ctx->mm_sets = S;
    ctx->mm_rows = R;
  }
  const int S = n_sets, R = ctx->mm_rows;
  vsf_status st = ensure_match_buffers(ctx, S, R);
  if (st != VSF_OK) return st;
  const size_t set_stride = (size_t)R * VSF_DESC_BYTES;
  std::vector<int32_t> meta((size_t)3 * S + 1);
  for (int s = 0; s < S; s++) {
    meta[s] = nq[s];
    meta[S + 1 + s] = s;      // q_set
    meta[2 * S + 1 + s] = S;  // t_set: the one train set
    if (nq[s] > 0)
      VSF_HIP(hipMemcpyAsync(ctx->mm_desc + s * set_stride, q[s], (size_t)nq[s] * VSF_DESC_BYTES, hipMemcpyHostToDevice,
                             ctx->stream));
  }
  meta[S] = nt;
  if (nt > 0)
    VSF_HIP(hipMemcpyAsync(ctx->mm_desc + S * set_stride, t, (size_t)nt * VSF_DESC_BYTES, hipMemcpyHostToDevice,
                           ctx->stream));
  VSF_HIP(hipMemcpyAsync(ctx->mm_counts, meta.data(), meta.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  const int32_t* d_q = ctx->mm_counts + S + 1;
  const int32_t* d_t = ctx->mm_counts + 2 * S + 1;
  // every buffer of this call is laid out with row capacity R: m_idx2 / m_dist2 [S][R][2] (they hold at least
  // m_pairs x m_rows >= S x R entries), mm_matches [S][R]
  vsf_launch_knn2(ctx->mm_desc, ctx->mm_counts, set_stride, d_q, d_t, S, R, ctx->m_idx2, ctx->m_dist2,
                  ctx->stream, ctx->tuning.match_int8 != 0);
  vsf_launch_ratio_compact(ctx->mm_counts, d_q, d_t, S, R, ctx->m_idx2, ctx->m_dist2, ctx->p.ratio_num,
                           ctx->p.ratio_shift, ctx->mm_matches, ctx->mm_nmatches, ctx->d_status, ctx->stream);
  VSF_STICKY();
  std::vector<int32_t> nm(S);
  VSF_HIP(hipMemcpyAsync(nm.data(), ctx->mm_nmatches, (size_t)S * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  st = VSF_OK;
  for (int s = 0; s < S; s++) {
    n_out[s] = nm[s];
    const int m = std::min((int)nm[s], cap_per_set);
    if (m > 0)
      VSF_HIP(hipMemcpyAsync(out + (size_t)s * cap_per_set, ctx->mm_matches + (size_t)s * R,
                             (size_t)m * sizeof(vsf_dmatch), hipMemcpyDeviceToHost, ctx->stream));
    if (nm[s] > cap_per_set) st = VSF_ERR_CAPACITY;
  }
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  return st;
}

vsf_status vsf_knn2_hamming(vsf_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2,
                            int32_t* dist2) {
  VsfErrorScope scope_(ctx);
  if (!ctx || (nq > 0 && (!idx2 || !dist2))) return VSF_ERR_INVALID_ARG;
  return match_host(ctx, q, nq, t, nt, idx2, dist2, nullptr, 0, nullptr);
}

vsf_status vsf_get_matches(vsf_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, vsf_dmatch* out,
                           int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || cap < 0 || (cap > 0 && !out)) return VSF_ERR_INVALID_ARG;
  vsf_dmatch dummy;
  return match_host(ctx, q, nq, t, nt, nullptr, nullptr, out ? out : &dummy, cap, n_out);
}

// ---------------- introspection ----------------

vsf_status vsf_debug_retain_best(vsf_ctx* ctx, uint32_t* key_bits, uint32_t* ids, int n, int n_points, int use_lds,
                                 int mode, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || n < 0 || (n > 0 && (!key_bits || !ids))) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  std::vector<uint2> h((size_t)std::max(n, 1));
  for (int i = 0; i < n; i++) h[i] = make_uint2(key_bits[i], ids[i]);
  uint2* d = nullptr;
  uint32_t* dt = nullptr;
  int* dn = nullptr;
  VSF_HIP(hipMalloc((void**)&d, h.size() * sizeof(uint2)));
  VSF_HIP(hipMalloc((void**)&dt, 2 * h.size() * sizeof(uint32_t)));
  VSF_HIP(hipMalloc((void**)&dn, sizeof(int)));
  VSF_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(uint2), hipMemcpyHostToDevice));
  vsf_launch_retain_best_test(d, dt, n, n_points, use_lds, mode, dn, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(h.data(), d, h.size() * sizeof(uint2), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(n_out, dn, sizeof(int), hipMemcpyDeviceToHost);
  hipFree(d);
  hipFree(dt);
  hipFree(dn);
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    return VSF_ERR_HIP;
  }
  for (int i = 0; i < n; i++) {
    key_bits[i] = h[i].x;
    ids[i] = h[i].y;
  }
  return VSF_OK;
}

vsf_status vsf_debug_sort_trim(vsf_ctx* ctx, const vsf_dmatch* matches, int n_lists, int n, float best_percent,
                               int serial, uint64_t* pairs_out, int32_t* counts_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !matches || !pairs_out || !counts_out || n_lists < 1 || n < 0 || n > ctx->p.max_keypoints)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const size_t K = (size_t)ctx->p.max_keypoints;
  vsf_dmatch* dm = nullptr;
  int32_t *dn = nullptr, *dc = nullptr;
  uint64_t* dp = nullptr;
  void* dscratch = nullptr;
  hipError_t e = hipMalloc((void**)&dm, (size_t)n_lists * K * sizeof(vsf_dmatch));
  if (e == hipSuccess) e = hipMalloc((void**)&dn, (size_t)n_lists * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&dc, (size_t)n_lists * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&dp, (size_t)n_lists * K * 2 * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMalloc(&dscratch, (size_t)n_lists * K * 8);
  std::vector<int32_t> hn((size_t)n_lists, n);
  if (e == hipSuccess) e = hipMemcpy(dn, hn.data(), hn.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  for (int i = 0; i < n_lists && e == hipSuccess && n > 0; i++)
    e = hipMemcpy(dm + (size_t)i * K, matches + (size_t)i * n, (size_t)n * sizeof(vsf_dmatch), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vsf_launch_sort_trim(dm, dn, n_lists, (int)K, best_percent, nullptr, dscratch, dp, dc, ctx->stream, serial != 0,
                         ctx->tuning.lds_limit);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpy(counts_out, dc, (size_t)n_lists * sizeof(int32_t), hipMemcpyDeviceToHost);
  for (int i = 0; i < n_lists && e == hipSuccess && n > 0; i++)
    e = hipMemcpy(pairs_out + (size_t)i * n * 2, dp + (size_t)i * K * 2, (size_t)n * 2 * sizeof(uint64_t),
                  hipMemcpyDeviceToHost);
  hipFree(dm);
  hipFree(dn);
  hipFree(dc);
  hipFree(dp);
  hipFree(dscratch);
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    return VSF_ERR_HIP;
  }
  return VSF_OK;
}

vsf_status vsf_debug_level_image(vsf_ctx* ctx, int image, int level, int blurred, uint8_t* out, size_t ostride) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !out || !ctx->last_valid || image < 0 || image >= ctx->last_images.n || level < 0 ||
      level >= ctx->orb.g.nlevels)
    return VSF_ERR_INVALID_ARG;
  const VsfLevel& L = ctx->orb.levels[level];
  if (ostride < (size_t)L.w) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  const uint8_t* src;
  size_t pitch;
  if (!blurred && level == 0) {
    src = ctx->last_images.base + (size_t)image * ctx->last_images.image_stride;
    pitch = ctx->last_images.row_stride;
  } else {
    src = (blurred ? ctx->dorb.d.blur : (ctx->last_pyr ? ctx->last_pyr : ctx->dorb.d.pyr)) +
          (size_t)image * ctx->orb.g.pyr_bytes + L.offset;
    pitch = (size_t)L.pitch;
  }
  if (blurred) {  // stored in tiles (VSF_BLUR_TILE_OFFSET)
    std::vector<uint8_t> tiled((size_t)L.pitch * align_up(L.h, 8));
    VSF_HIP(hipMemcpy(tiled.data(), src, tiled.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; y++)
      for (int x = 0; x < L.w; x++) out[(size_t)y * ostride + x] = tiled[VSF_BLUR_TILE_OFFSET(L.pitch, x, y)];
    return VSF_OK;
  }
  VSF_HIP(hipMemcpy2D(out, ostride, src, pitch, (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
  return VSF_OK;
}

vsf_status vsf_debug_fast_candidates(vsf_ctx* ctx, int image, int level, vsf_keypoint* kp_out, int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || !ctx->last_valid || image < 0 || image >= ctx->last_images.n || level < 0 ||
      level >= ctx->orb.g.nlevels)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  const VsfLevel& L = ctx->orb.levels[level];
  const VsfGeom& g = ctx->orb.g;
  // Merge the unit segments (unit-local raster order + per-row starts) into the level's raster order.
  const int nu = L.nstrips * L.nbands;
  int n = 0;
  if (nu > 0) {
    std::vector<uint16_t> rs((size_t)nu * VSF_FAST_RS_STRIDE);
    VSF_HIP(hipMemcpy(rs.data(), ctx->dorb.d.rowstart + ((size_t)image * g.nunits + L.unit0) * VSF_FAST_RS_STRIDE,
                      rs.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> seg((size_t)nu * L.seg_cap);
    VSF_HIP(hipMemcpy(seg.data(), ctx->dorb.d.cand + (size_t)image * g.cand_entries + L.cand_offset,
                      seg.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int row = 0; row < L.y_hi - L.y_lo; row++) {
      const int s = row / VSF_FAST_STRIP_ROWS, r = row % VSF_FAST_STRIP_ROWS;
      for (int b = 0; b < L.nbands; b++) {
        const int u = s * L.nbands + b;
        const uint16_t* urs = rs.data() + (size_t)u * VSF_FAST_RS_STRIDE;
        for (int e = urs[r]; e < urs[r + 1]; e++, n++) {
          if (n < cap && kp_out) {
            const uint32_t cd = seg[(size_t)u * L.seg_cap + e];
            kp_out[n] =
                vsf_keypoint{(float)VSF_CAND_X(cd), (float)VSF_CAND_Y(cd), 7.f, -1.f, (float)VSF_CAND_SCORE(cd), 0, -1};
          }
        }
      }
    }
  }
  *n_out = n;
  return VSF_OK;
}

vsf_status vsf_debug_level_keypoints(vsf_ctx* ctx, int image, int level, vsf_keypoint* kp_out, int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || !ctx->last_valid || image < 0 || image >= ctx->last_images.n || level < 0 ||
      level >= ctx->orb.g.nlevels)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  const VsfLevel& L = ctx->orb.levels[level];
  const VsfGeom& g = ctx->orb.g;
  int32_t n = 0;
  VSF_HIP(hipMemcpy(&n, ctx->dorb.d.lvl_count + (size_t)image * g.nlevels + level, sizeof(int32_t),
                    hipMemcpyDeviceToHost));
  std::vector<VsfLevelKp> v(std::max(n, 1));
  if (n > 0)
    VSF_HIP(hipMemcpy(v.data(), ctx->dorb.d.lvlkp + (size_t)image * g.lvlkp_entries + L.kp_offset,
                      (size_t)n * sizeof(VsfLevelKp), hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < cap && kp_out; i++)
    kp_out[i] = vsf_keypoint{(float)(v[i].xy & 0xFFFu), (float)(v[i].xy >> 12), 31 * L.scale, v[i].angle,
                             v[i].response, level, -1};
  *n_out = n;
  return VSF_OK;
}

}  // extern "C"
