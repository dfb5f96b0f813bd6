// vsf_oracle.cc -- CPU ORACLE (test infrastructure; see vsf_oracle.h for the parity statement).
//
// Scalar restatement of the OpenCV-3.2.0 routines the reference reaches from
// src/slam_frontend.cc:274 (detectAndCompute), :271 (FAST detect) and :525 (knnMatch), and of the
// reference's own GetMatches / GetFeatureMatches / RemoveAmbigStereo.  OpenCV file names below are
// the upstream 3.2.0 module paths (the library is not vendored by the reference; SURVEY.md
// Appendix A).  Build with -ffp-contract=off and no -march flags: every float operation here
// is meant to be one separately rounded IEEE operation, as GCC emits for baseline x86-64.
#include "vsf_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---- core/fast_math.hpp: cvRound / cvFloor / cvCeil, saturate_cast ------------------------------
inline int cv_round(double v) { return (int)std::nearbyint(v); }  // cvtsd2si, round-half-even
inline int cv_round(float v) { return (int)std::nearbyintf(v); }  // cvtss2si
inline int cv_floor(double v) {
  int i = cv_round(v);
  float diff = (float)(v - i);
  return i - (diff < 0);
}
inline int cv_ceil(double v) {
  int i = cv_round(v);
  float diff = (float)(i - v);
  return i + (diff < 0);
}
inline int16_t sat_short(float v) {
  int iv = cv_round(v);
  return (int16_t)std::min(std::max(iv, (int)SHRT_MIN), (int)SHRT_MAX);
}
inline uint8_t sat_u8(int v) { return (uint8_t)std::min(std::max(v, 0), 255); }

const int8_t kPattern31[256 * 4] = {
#include "orb_pattern31.inc"
};

// ---- imgproc/imgwarp.cpp: resize, INTER_LINEAR, CV_8UC1 -----------------------------------------
constexpr int kResizeCoefBits = 11;
constexpr int kResizeCoefScale = 1 << kResizeCoefBits;

struct ResizeTables {
  std::vector<int32_t> xofs, yofs;
  std::vector<int16_t> ialpha, ibeta;
  int xmax;
};

ResizeTables BuildResizeTables(int sw, int sh, int dw, int dh) {
  ResizeTables t;
  t.xofs.resize(dw);
  t.yofs.resize(dh);
  t.ialpha.resize(2 * dw);
  t.ibeta.resize(2 * dh);
  // cv::resize: inv_scale = (double)dsize/ssize; scale = 1./inv_scale.
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;  // ksize2-1 == 0 for the 2-tap kernel
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) fx = 0, sx = sw - 1;
    }
    t.xofs[dx] = sx;
    const float c0 = 1.f - fx, c1 = fx;
    t.ialpha[2 * dx] = sat_short(c0 * kResizeCoefScale);
    t.ialpha[2 * dx + 1] = sat_short(c1 * kResizeCoefScale);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    t.yofs[dy] = sy;  // rows are clipped later, weights are kept (resizeGeneric_Invoker)
    const float c0 = 1.f - fy, c1 = fy;
    t.ibeta[2 * dy] = sat_short(c0 * kResizeCoefScale);
    t.ibeta[2 * dy + 1] = sat_short(c1 * kResizeCoefScale);
  }
  t.xmax = xmax;
  return t;
}

inline int clip_row(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

void ResizeLinearU8(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                    size_t dstride) {
  const ResizeTables t = BuildResizeTables(sw, sh, dw, dh);
  std::vector<int32_t> row0(dw), row1(dw);
  for (int dy = 0; dy < dh; dy++) {
    const int sy0 = clip_row(t.yofs[dy], 0, sh), sy1 = clip_row(t.yofs[dy] + 1, 0, sh);
    const uint8_t* S0 = src + (size_t)sy0 * sstride;
    const uint8_t* S1 = src + (size_t)sy1 * sstride;
    // HResizeLinear<uchar,int,short,2048>
    for (int dx = 0; dx < dw; dx++) {
      const int sx = t.xofs[dx];
      if (dx < t.xmax) {
        row0[dx] = S0[sx] * t.ialpha[2 * dx] + S0[sx + 1] * t.ialpha[2 * dx + 1];
        row1[dx] = S1[sx] * t.ialpha[2 * dx] + S1[sx + 1] * t.ialpha[2 * dx + 1];
      } else {
        row0[dx] = S0[sx] * kResizeCoefScale;
        row1[dx] = S1[sx] * kResizeCoefScale;
      }
    }
    // VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>
    const int b0 = t.ibeta[2 * dy], b1 = t.ibeta[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---- features2d/fast.cpp + fast_score.cpp: FAST-9/16 ---------------------------------------------
const int kFastOffsets16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

void MakeOffsets(int pixel[25], int row_stride) {
  int k = 0;
  for (; k < 16; k++) pixel[k] = kFastOffsets16[k][0] + kFastOffsets16[k][1] * row_stride;
  for (; k < 25; k++) pixel[k] = pixel[k - 16];
}

// cornerScore<16>, C path (the SSE2 path agrees on every pixel that passes the segment test).
int CornerScore16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

// FAST_t<16>: three rolling score rows, keypoints of row i-1 emitted while row i is scanned.
void Fast9_16(const uint8_t* img, int cols, int rows, size_t stride, int threshold, bool nms,
              std::vector<vsfo_keypoint>* out) {
  out->clear();
  if (cols < 7 || rows < 7) return;
  const int K = 8, N = 25;
  int pixel[25];
  MakeOffsets(pixel, (int)stride);
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t threshold_tab[512];
  for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  std::vector<uint8_t> bufmem((size_t)cols * 3, 0);
  uint8_t* buf[3] = {bufmem.data(), bufmem.data() + cols, bufmem.data() + 2 * cols};
  std::vector<int> cpmem((size_t)(cols + 1) * 3, 0);
  int* cpbuf[3] = {cpmem.data() + 1, cpmem.data() + 1 + (cols + 1), cpmem.data() + 1 + 2 * (cols + 1)};
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * stride + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    std::memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        const int v = ptr[0];
        const uint8_t* tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        if (d & 1) {
          const int vt = v - threshold;
          int count = 0;
          for (int k = 0; k < N; k++) {
            const int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nms) curr[j] = (uint8_t)CornerScore16(ptr, pixel, threshold);
                break;
              }
            } else {
              count = 0;
            }
          }
        }
        if (d & 2) {
          const int vt = v + threshold;
          int count = 0;
          for (int k = 0; k < N; k++) {
            const int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nms) curr[j] = (uint8_t)CornerScore16(ptr, pixel, threshold);
                break;
              }
            } else {
              count = 0;
            }
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      const int j = cornerpos[k];
      const int score = prev[j];
      if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                   score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        vsfo_keypoint kp = {(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1};
        out->push_back(kp);
      }
    }
  }
}

// ---- imgproc/smooth.cpp + filter.cpp: GaussianBlur 7x7 sigma 2 on CV_8U ---------------------------
void GaussianKernel7Fixed(int32_t k[7]) {
  // getGaussianKernel(7, 2, CV_32F): cf[i] = (float)exp(-0.5/(s*s) * x*x), normalised in double.
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
  for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
  // createSeparableLinearFilter: 8u smooth symmetric -> kernel.convertTo(CV_32S, 1 << 8).
  for (int i = 0; i < n; i++) k[i] = cv_round((double)cf[i] * 256.0);
}

inline int reflect101(int p, int len) {
  // borderInterpolate(BORDER_REFLECT_101)
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0)
      p = -p;
    else
      p = len - 1 - (p - len) - 1;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

void GaussianBlur7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride, bool sse2) {
  int32_t k[7];
  GaussianKernel7Fixed(k);
  std::vector<int32_t> rows((size_t)w * h);
  // RowFilter<uchar,int>: R = sum k_i * p_i (exact int32).
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + (size_t)y * sstride;
    int32_t* R = rows.data() + (size_t)y * w;
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int i = 0; i < 7; i++) s += k[i] * S[reflect101(x + i - 3, w)];
      R[x] = s;
    }
  }
  // SymmColumnFilter<FixedPtCastEx<int,uchar>(16), SymmColumnVec_32s8u>.
  const int vec_end = sse2 ? (w - w % 4) : 0;
  for (int y = 0; y < h; y++) {
    const int32_t* r[7];
    for (int j = 0; j < 7; j++) r[j] = rows.data() + (size_t)reflect101(y + j - 3, h) * w;
    uint8_t* D = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) {
      if (x < vec_end) {
        // float path: s = R0*f0; s += (R_k + R_-k)*f_k, f = k/65536 exact; cvtps2dq rounds half-even.
        float s = (float)r[3][x] * ((float)k[3] * (1.f / 65536.f));
        for (int j = 1; j <= 3; j++) s = s + (float)(r[3 + j][x] + r[3 - j][x]) * ((float)k[3 + j] * (1.f / 65536.f));
        D[x] = sat_u8(cv_round(s));
      } else {
        int s = k[3] * r[3][x];
        for (int j = 1; j <= 3; j++) s += k[3 + j] * (r[3 + j][x] + r[3 - j][x]);
        D[x] = sat_u8((s + (1 << 15)) >> 16);
      }
    }
  }
}

// ---- core/mathfuncs: fastAtan2 -------------------------------------------------------------------
const float kAtan2P1 = 0.9997878412794807f * (float)(180 / M_PI);
const float kAtan2P3 = -0.3258083974640975f * (float)(180 / M_PI);
const float kAtan2P5 = 0.1555786518463281f * (float)(180 / M_PI);
const float kAtan2P7 = -0.04432655554792128f * (float)(180 / M_PI);

float FastAtan2(float y, float x) {
  const float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((kAtan2P7 * c2 + kAtan2P5) * c2 + kAtan2P3) * c2 + kAtan2P1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((kAtan2P7 * c2 + kAtan2P5) * c2 + kAtan2P3) * c2 + kAtan2P1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---- features2d/keypoint.cpp: KeyPointsFilter ------------------------------------------------------
struct ResponseGreater {
  bool operator()(const vsfo_keypoint& a, const vsfo_keypoint& b) const { return a.response > b.response; }
};
struct ResponseGeThreshold {
  float value;
  bool operator()(const vsfo_keypoint& k) const { return k.response >= value; }
};

void RunByImageBorder(std::vector<vsfo_keypoint>* kps, int w, int h, int border) {
  if (border <= 0) return;
  if (h <= border * 2 || w <= border * 2) {
    kps->clear();
    return;
  }
  // RoiPredicate: Rect(border, border, w-2b, h-2b).contains(Point(cvRound(pt)))
  auto outside = [&](const vsfo_keypoint& k) {
    const int x = cv_round(k.x), y = cv_round(k.y);
    return !(border <= x && x < w - border && border <= y && y < h - border);
  };
  kps->erase(std::remove_if(kps->begin(), kps->end(), outside), kps->end());
}

// retainBest: the survivors' ORDER is whatever libstdc++'s nth_element + partition leave.
void RetainBest(std::vector<vsfo_keypoint>* kps, int n_points) {
  if (n_points >= 0 && kps->size() > (size_t)n_points) {
    if (n_points == 0) {
      kps->clear();
      return;
    }
    std::nth_element(kps->begin(), kps->begin() + n_points, kps->end(), ResponseGreater());
    const float ambiguous = (*kps)[n_points - 1].response;
    auto new_end = std::partition(kps->begin() + n_points, kps->end(), ResponseGeThreshold{ambiguous});
    kps->resize(new_end - kps->begin());
  }
}

// ---- features2d/orb.cpp ---------------------------------------------------------------------------
struct Level {
  int w = 0, h = 0, nfeatures = 0;
  float scale = 1.f;
  std::vector<uint8_t> img, blurred;  // tightly packed w*h
  std::vector<vsfo_keypoint> stage[5];
};

}  // namespace

struct vsfo_orb {
  vsfo_orb_params p;
  std::vector<Level> levels;
  std::vector<int> umax;
  std::vector<vsfo_keypoint> keypoints;
  std::vector<uint8_t> descriptors;
  int w = 0, h = 0;
};

namespace {

float GetScale(int level, int first_level, double scale_factor) {
  return (float)std::pow(scale_factor, (double)(level - first_level));
}

void Layout(vsfo_orb* o, int w, int h) {
  const vsfo_orb_params& p = o->p;
  const int nlevels = p.nlevels;
  const double scale_factor = (double)p.scale_factor;  // ORB_Impl stores the float argument in a double
  o->w = w;
  o->h = h;
  o->levels.assign(nlevels, Level());
  for (int l = 0; l < nlevels; l++) {
    Level& L = o->levels[l];
    L.scale = GetScale(l, p.first_level, scale_factor);
    L.w = cv_round(w / L.scale);  // int / float -> float
    L.h = cv_round(h / L.scale);
  }
  // computeKeyPoints: per-level feature budget.
  const float factor = (float)(1.0 / scale_factor);
  float ndesired = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    o->levels[l].nfeatures = cv_round(ndesired);
    sum += o->levels[l].nfeatures;
    ndesired *= factor;
  }
  o->levels[nlevels - 1].nfeatures = std::max(p.nfeatures - sum, 0);
  // umax: end of each row of the circular patch.
  const int half = p.patch_size / 2;
  o->umax.assign(half + 2, 0);
  int v, v0;
  const int vmax = cv_floor(half * std::sqrt(2.f) / 2 + 1);
  const int vmin = cv_ceil(half * std::sqrt(2.f) / 2);
  for (v = 0; v <= vmax; ++v) o->umax[v] = cv_round(std::sqrt((double)half * half - v * v));
  for (v = half, v0 = 0; v >= vmin; --v) {
    while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
    o->umax[v] = v0;
    ++v0;
  }
}

// HarrisResponses(blockSize = 7, harris_k = 0.04f)
void HarrisResponses(const Level& L, std::vector<vsfo_keypoint>* pts) {
  const int block = 7, r = block / 2, step = L.w;
  const float harris_k = 0.04f;
  const float scale = 1.f / ((1 << 2) * block * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  for (auto& kp : *pts) {
    const int x0 = cv_round(kp.x), y0 = cv_round(kp.y);
    const uint8_t* ptr0 = L.img.data() + (size_t)(y0 - r) * step + (x0 - r);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < block; i++)
      for (int j = 0; j < block; j++) {
        const uint8_t* ptr = ptr0 + i * step + j;
        const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
        const int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
        a += Ix * Ix;
        b += Iy * Iy;
        c += Ix * Iy;
      }
    kp.response = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
  }
}

// ICAngles
void ICAngles(const Level& L, const std::vector<int>& umax, int half_k, std::vector<vsfo_keypoint>* pts) {
  const int step = L.w;
  for (auto& kp : *pts) {
    const uint8_t* center = L.img.data() + (size_t)cv_round(kp.y) * step + cv_round(kp.x);
    int m_01 = 0, m_10 = 0;
    for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
    for (int v = 1; v <= half_k; ++v) {
      int v_sum = 0;
      const int d = umax[v];
      for (int u = -d; u <= d; ++u) {
        const int val_plus = center[u + v * step], val_minus = center[u - v * step];
        v_sum += (val_plus - val_minus);
        m_10 += u * (val_plus + val_minus);
      }
      m_01 += v * v_sum;
    }
    kp.angle = FastAtan2((float)m_01, (float)m_10);
  }
}

// computeOrbDescriptors, WTA_K == 2. `kp` carries level-0 coordinates.
void OrbDescriptor(const Level& L, const vsfo_keypoint& kp, uint8_t desc[32]) {
  const int step = L.w;
  const float scale = 1.f / L.scale;
  float angle = kp.angle;
  angle *= (float)(M_PI / 180.f);
  // SURVEY A.8 risk note: cos/sin taken as the correctly rounded float of the float angle.
  const float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
  const uint8_t* center = L.blurred.data() + (size_t)cv_round(kp.y * scale) * step + cv_round(kp.x * scale);
  const int8_t* pat = kPattern31;
  auto get = [&](int idx) -> int {
    const float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
    const float x = px * a - py * b;
    const float y = px * b + py * a;
    const int ix = cv_round(x), iy = cv_round(y);
    return center[iy * step + ix];
  };
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int j = 0; j < 8; j++) {
      const int t0 = get(2 * j), t1 = get(2 * j + 1);
      val |= (t0 < t1) << j;
    }
    desc[i] = (uint8_t)val;
  }
}

int OrbRun(vsfo_orb* o, const uint8_t* img, int w, int h, size_t stride) {
  const vsfo_orb_params& p = o->p;
  if (p.first_level != 0 || p.wta_k != 2 || p.score_type != 0 || p.patch_size != 31 || p.nlevels < 1) return -1;
  Layout(o, w, h);
  const int nlevels = p.nlevels;
  // Pyramid: level 0 is the image; level l is resize(level l-1 interior). Borders are never read by any
  // output (edgeThreshold 31 > Harris 4 / IC 15 / descriptor 19+3), so they are not materialised.
  for (int l = 0; l < nlevels; l++) {
    Level& L = o->levels[l];
    if (L.w < 1 || L.h < 1) return -2;
    L.img.resize((size_t)L.w * L.h);
    if (l == 0) {
      for (int y = 0; y < h; y++) std::memcpy(L.img.data() + (size_t)y * w, img + (size_t)y * stride, w);
    } else {
      const Level& P = o->levels[l - 1];
      ResizeLinearU8(P.img.data(), P.w, P.h, P.w, L.img.data(), L.w, L.h, L.w);
    }
  }
  // computeKeyPoints
  const int half = p.patch_size / 2;
  for (int l = 0; l < nlevels; l++) {
    Level& L = o->levels[l];
    std::vector<vsfo_keypoint> kps;
    Fast9_16(L.img.data(), L.w, L.h, L.w, p.fast_threshold, true, &kps);
    RunByImageBorder(&kps, L.w, L.h, p.edge_threshold);
    L.stage[0] = kps;
    RetainBest(&kps, 2 * L.nfeatures);
    for (auto& k : kps) {
      k.octave = l;
      k.size = p.patch_size * L.scale;
    }
    L.stage[1] = kps;
  }
  for (int l = 0; l < nlevels; l++) {
    Level& L = o->levels[l];
    std::vector<vsfo_keypoint> kps = L.stage[1];
    HarrisResponses(L, &kps);
    L.stage[2] = kps;
    RetainBest(&kps, L.nfeatures);
    L.stage[3] = kps;
    ICAngles(L, o->umax, half, &kps);
    L.stage[4] = kps;
  }
  o->keypoints.clear();
  for (int l = 0; l < nlevels; l++) {
    const Level& L = o->levels[l];
    for (vsfo_keypoint k : L.stage[4]) {
      k.x *= L.scale;
      k.y *= L.scale;
      o->keypoints.push_back(k);
    }
  }
  // Descriptors on the blurred pyramid.
  for (int l = 0; l < nlevels; l++) {
    Level& L = o->levels[l];
    L.blurred.resize(L.img.size());
    GaussianBlur7(L.img.data(), L.w, L.h, L.w, L.blurred.data(), L.w, p.blur_sse2 != 0);
  }
  o->descriptors.assign(o->keypoints.size() * 32, 0);
  for (size_t j = 0; j < o->keypoints.size(); j++)
    OrbDescriptor(o->levels[o->keypoints[j].octave], o->keypoints[j], o->descriptors.data() + j * 32);
  return (int)o->keypoints.size();
}

// ---- core/stat.cpp batchDistance(NORM_HAMMING, K = 2) + features2d/matchers.cpp ------------------
inline int Hamming32(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; i += 8) {
    uint64_t x, y;
    std::memcpy(&x, a + i, 8);
    std::memcpy(&y, b + i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

void Knn2Rows(const uint8_t* q, int q0, int q1, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2) {
  const int K = std::min(2, nt);
  for (int i = q0; i < q1; i++) {
    int32_t* di = dist2 + 2 * i;
    int32_t* ii = idx2 + 2 * i;
    di[0] = di[1] = INT_MAX;
    ii[0] = ii[1] = -1;
    if (K == 0) continue;
    for (int j = 0; j < nt; j++) {
      const int d = Hamming32(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < di[K - 1]) {
        int k;
        for (k = K - 2; k >= 0 && di[k] > d; k--) {
          ii[k + 1] = ii[k];
          di[k + 1] = di[k];
        }
        ii[k + 1] = j;
        di[k + 1] = d;
      }
    }
  }
}

int GetMatches(const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, vsfo_dmatch* out, int cap,
               int threads) {
  if (nq <= 0) return 0;
  if (nt < 2) return 0;  // quirk Q6: the reference would read matches[i][1] out of bounds
  std::vector<int32_t> idx((size_t)nq * 2), dist((size_t)nq * 2);
  if (threads <= 1) {
    Knn2Rows(q, 0, nq, t, nt, idx.data(), dist.data());
  } else {
    std::vector<std::thread> pool;
    const int chunk = (nq + threads - 1) / threads;
    for (int k = 0; k < threads; k++) {
      const int a = k * chunk, b = std::min(nq, a + chunk);
      if (a >= b) break;
      pool.emplace_back(Knn2Rows, q, a, b, t, nt, idx.data(), dist.data());
    }
    for (auto& th : pool) th.join();
  }
  int n = 0;
  for (int i = 0; i < nq; i++) {
    const float dist1 = (float)dist[2 * i], d2 = (float)dist[2 * i + 1];
    if (dist1 < ratio * d2) {  // float < double * float, evaluated in double (slam_frontend.cc:533)
      if (n < cap) out[n] = vsfo_dmatch{i, idx[2 * i], 0, dist1};
      n++;
    }
  }
  return n;
}

}  // namespace

extern "C" {

void vsfo_orb_params_default(vsfo_orb_params* p) {
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
}

int vsfo_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                          size_t dstride) {
  if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1) return -1;
  ResizeLinearU8(src, sw, sh, sstride, dst, dw, dh, dstride);
  return 0;
}

int vsfo_resize_tables(int sw, int sh, int dw, int dh, int32_t* xofs, int16_t* ialpha, int32_t* yofs,
                       int16_t* ibeta) {
  const ResizeTables t = BuildResizeTables(sw, sh, dw, dh);
  std::memcpy(xofs, t.xofs.data(), sizeof(int32_t) * dw);
  std::memcpy(ialpha, t.ialpha.data(), sizeof(int16_t) * 2 * dw);
  std::memcpy(yofs, t.yofs.data(), sizeof(int32_t) * dh);
  std::memcpy(ibeta, t.ibeta.data(), sizeof(int16_t) * 2 * dh);
  return t.xmax;
}

int vsfo_fast9_16(const uint8_t* img, int w, int h, size_t stride, int threshold, int nms, vsfo_keypoint* out,
                  int cap) {
  std::vector<vsfo_keypoint> kps;
  Fast9_16(img, w, h, stride, threshold, nms != 0, &kps);
  const int n = (int)kps.size();
  if (out && cap > 0) std::memcpy(out, kps.data(), sizeof(vsfo_keypoint) * std::min(n, cap));
  return n;
}

int vsfo_fast_corner_score(const uint8_t* img, size_t stride, int x, int y, int threshold) {
  int pixel[25];
  MakeOffsets(pixel, (int)stride);
  return CornerScore16(img + (size_t)y * stride + x, pixel, threshold);
}

int vsfo_gaussian_blur7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride,
                        int sse2_rounding) {
  if (!src || !dst || w < 1 || h < 1) return -1;
  GaussianBlur7(src, w, h, sstride, dst, dstride, sse2_rounding != 0);
  return 0;
}

void vsfo_gaussian_kernel7_fixed(int32_t k[7]) { GaussianKernel7Fixed(k); }

float vsfo_fast_atan2(float y, float x) { return FastAtan2(y, x); }

const int8_t* vsfo_orb_pattern31(void) { return kPattern31; }

int vsfo_retain_best(float* response, uint32_t* id, int n, int n_points) {
  std::vector<vsfo_keypoint> kps((size_t)std::max(n, 0));
  for (int i = 0; i < n; i++) {
    kps[i] = vsfo_keypoint{0.f, 0.f, 0.f, 0.f, response[i], 0, (int32_t)id[i]};
  }
  RetainBest(&kps, n_points);
  for (size_t i = 0; i < kps.size(); i++) {
    response[i] = kps[i].response;
    id[i] = (uint32_t)kps[i].class_id;
  }
  return (int)kps.size();
}

vsfo_orb* vsfo_orb_create(const vsfo_orb_params* p) {
  vsfo_orb* o = new vsfo_orb();
  o->p = *p;
  return o;
}

void vsfo_orb_destroy(vsfo_orb* o) { delete o; }

int vsfo_orb_run(vsfo_orb* o, const uint8_t* img, int w, int h, size_t stride) {
  if (!o || !img || w < 1 || h < 1 || stride < (size_t)w) return -1;
  return OrbRun(o, img, w, h, stride);
}

int vsfo_orb_nlevels(const vsfo_orb* o) { return o->p.nlevels; }

int vsfo_orb_layout(vsfo_orb* o, int w, int h) {
  Layout(o, w, h);
  return 0;
}

int vsfo_orb_level_info(const vsfo_orb* o, int level, int* w, int* h, float* scale, int* nfeatures) {
  if (level < 0 || level >= (int)o->levels.size()) return -1;
  const Level& L = o->levels[level];
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (scale) *scale = L.scale;
  if (nfeatures) *nfeatures = L.nfeatures;
  return 0;
}

int vsfo_orb_level_image(const vsfo_orb* o, int level, int blurred, uint8_t* out, size_t ostride) {
  if (level < 0 || level >= (int)o->levels.size()) return -1;
  const Level& L = o->levels[level];
  const std::vector<uint8_t>& src = blurred ? L.blurred : L.img;
  if (src.empty()) return -2;
  for (int y = 0; y < L.h; y++) std::memcpy(out + (size_t)y * ostride, src.data() + (size_t)y * L.w, L.w);
  return 0;
}

int vsfo_orb_stage_keypoints(const vsfo_orb* o, int stage, int level, vsfo_keypoint* out, int cap) {
  if (level < 0 || level >= (int)o->levels.size() || stage < 0 || stage > 4) return -1;
  const std::vector<vsfo_keypoint>& v = o->levels[level].stage[stage];
  const int n = (int)v.size();
  if (out && cap > 0) std::memcpy(out, v.data(), sizeof(vsfo_keypoint) * std::min(n, cap));
  return n;
}

int vsfo_orb_result(const vsfo_orb* o, vsfo_keypoint* kps, uint8_t* desc, int cap) {
  const int n = (int)o->keypoints.size();
  const int m = std::min(n, cap);
  if (kps && m > 0) std::memcpy(kps, o->keypoints.data(), sizeof(vsfo_keypoint) * m);
  if (desc && m > 0) std::memcpy(desc, o->descriptors.data(), (size_t)32 * m);
  return n;
}

int vsfo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2) {
  if (nq < 0 || nt < 0) return -1;
  Knn2Rows(q, 0, nq, t, nt, idx2, dist2);
  return 0;
}

int vsfo_get_matches(const uint8_t* q, int nq, const uint8_t* t, int nt, double nn_match_ratio, vsfo_dmatch* out,
                     int cap) {
  return GetMatches(q, nq, t, nt, nn_match_ratio, out, cap, 1);
}

int vsfo_get_matches_mt(const uint8_t* q, int nq, const uint8_t* t, int nt, double nn_match_ratio,
                        vsfo_dmatch* out, int cap, int threads) {
  return GetMatches(q, nq, t, nt, nn_match_ratio, out, cap, threads);
}

int vsfo_sort_and_trim(vsfo_dmatch* m, int n, float best_percent) {
  // slam_frontend.cc:289-291: std::sort with DMatch::operator< (distance only), then
  // `const int num_good_matches = matches.size() * config_.best_percent_;` (size_t * float -> float -> int).
  std::sort(m, m + n, [](const vsfo_dmatch& a, const vsfo_dmatch& b) { return a.distance < b.distance; });
  const int num_good = (int)((float)(size_t)n * best_percent);
  return std::min(std::max(num_good, 0), n);
}

// Order in which a three-term dot product of  left_ph.transpose() * F * right_ph  (slam_frontend.cc:381-383) is summed.
// 0 (default): a0 b0 + (a1 b1 + a2 b2) -- Eigen 3.3's lazy coefficient product of fixed size 3 is
//    lhs.row(i).transpose().cwiseProduct(rhs.col(j)).sum(), and its unrolled reduction (Redux.h,
//    redux_novec_unroller<Func, Derived, 0, 3>: HalfLength = 1) adds element 0 to the sum of elements 1 and 2;
// 1: (a0 b0 + a1 b1) + a2 b2, plain left to right (what rounds 1 and 2 of this repo computed).
// With the rectified F of the synthetic stream (one non-zero per row) the two agree bit for bit; with a dense F they differ
// by an ulp now and then, which moves which pairs pass `<=`, the mean and every index downstream.
static int g_residual_order = 0;
void vsfo_set_residual_order(int order) { g_residual_order = order ? 1 : 0; }
int vsfo_get_residual_order(void) { return g_residual_order; }
static inline float Dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  const float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  return g_residual_order ? (p0 + p1) + p2 : p0 + (p1 + p2);
}

int vsfo_remove_ambig_stereo(const vsfo_keypoint* left, const vsfo_keypoint* right, const vsfo_dmatch* matches,
                             int n, const float F[9], float* threshold_io, uint8_t* keep, float* residual) {
  // slam_frontend.cc:369-394.  (l^T F r).norm() of a 1x1 is sqrt(x * x) == |x| (exactly, barring under- / overflow of
  // x * x: |x| below 1e-19 or above 1.8e19, never a pixel residual).
  float avg = 0.0f;
  int kept = 0;
  const float thr = *threshold_io;
  for (int m = 0; m < n; m++) {
    const float l[3] = {left[matches[m].queryIdx].x, left[matches[m].queryIdx].y, 1.0f};
    const float r[3] = {right[matches[m].trainIdx].x, right[matches[m].trainIdx].y, 1.0f};
    float t[3];
    for (int j = 0; j < 3; j++) t[j] = Dot3(l[0], F[0 * 3 + j], l[1], F[1 * 3 + j], l[2], F[2 * 3 + j]);
    const float c = std::fabs(Dot3(t[0], r[0], t[1], r[1], t[2], r[2]));
    avg += c;
    if (residual) residual[m] = c;
    const bool k = c <= thr;
    if (keep) keep[m] = k;
    kept += k;
  }
  // cc:392-394, unconditionally: with no match this is 0.0f / 0 + 2 = NaN (quirk Q3).  The NEXT frame is then filtered
  // against NaN and loses every feature (`constraint <= NaN` is false), but its own mean runs over ALL its matches, so
  // the frame after that is filtered with a finite threshold again: the NaN lives for exactly one frame.
  *threshold_io = avg / (float)(size_t)n + 2.0f;
  return kept;
}

// ---- cv::triangulatePoints / cv::undistortPoints as Calculate3DPoints / UndistortFeaturePoints
// call them (slam_frontend.cc:117-173, 323-351) ----
namespace {

// core/src/lapack.cpp: the file-local hypot template the Jacobi routines use (not ::hypot).
inline double cvLapackHypot(double a, double b) {
  a = std::abs(a);
  b = std::abs(b);
  if (a > b) {
    b /= a;
    return a * std::sqrt(1 + b * b);
  }
  if (b > 0) {
    a /= b;
    return b * std::sqrt(1 + a * a);
  }
  return 0;
}

// core/src/lapack.cpp JacobiSVDImpl_<double> as cv::SVD::compute reaches it for an m x n matrix with m >= n
// (At = A transposed: row i of At is column i of A; eps = DBL_EPSILON * 10; one-sided Hestenes rotations, at most
// max(m, 30) sweeps, singular values sorted descending with the rows of Vt swapped along).  U is not computed: only
// W and Vt are read by cvTriangulatePoints.
void JacobiSVD64(double* At, int astep, double* W, double* Vt, int vstep, int m, int n) {
  const double eps = DBL_EPSILON * 10;
  const int max_iter = std::max(m, 30);
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) {
      const double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = sd;
    for (int k = 0; k < n; k++) Vt[i * vstep + k] = 0;
    Vt[i * vstep + i] = 1;
  }
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        double *Ai = At + i * astep, *Aj = At + j * astep;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (std::abs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = cvLapackHypot(p, beta);
        double c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = std::sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          const double t0 = c * Ai[k] + s * Aj[k];
          const double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
        double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
        for (int k = 0; k < n; k++) {
          const double t0 = c * Vi[k] + s * Vj[k];
          const double t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0;
          Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) {
      const double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = std::sqrt(sd);
  }
  for (int i = 0; i < n - 1; i++) {
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
      for (int k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
    }
  }
}

}  // namespace

int vsfo_triangulate_points(const float P1[12], const float P2[12], const float* pts1, const float* pts2, int n,
                            int rows, float* points4d) {
  // calib3d/src/triangulate.cpp cvTriangulatePoints, reached through cv::triangulatePoints with CV_32F projection
  // matrices and vector<Point2f> points (slam_frontend.cc:152): every cvmGet widens a float to double, the 4 x n
  // output has the points' type (CV_32F), cvmSet narrows the double back to float.
  // rows == 6: OpenCV <= 3.4.1 (incl. the pinned 3.2.0): per view  x*P[2]-P[0],  y*P[2]-P[1],  x*P[1]-y*P[0];
  // rows == 4: OpenCV >= 3.4.2 / 4.x dropped the third (dependent) row.  SVD of the rows x 4 system, X = Vt row 3.
  if (!P1 || !P2 || n < 0 || (n > 0 && (!pts1 || !pts2 || !points4d)) || (rows != 6 && rows != 4)) return -1;
  const float* P[2] = {P1, P2};
  const float* pts[2] = {pts1, pts2};
  const int per = rows / 2;
  for (int i = 0; i < n; i++) {
    double At[4 * 6], W[4], Vt[16];  // At[k][r] = A[r][k]
    for (int j = 0; j < 2; j++) {
      const double x = pts[j][2 * i], y = pts[j][2 * i + 1];
      for (int k = 0; k < 4; k++) {
        const double p0 = P[j][k], p1 = P[j][4 + k], p2 = P[j][8 + k];
        At[k * rows + j * per + 0] = x * p2 - p0;
        At[k * rows + j * per + 1] = y * p2 - p1;
        if (per == 3) At[k * rows + j * per + 2] = x * p1 - y * p0;
      }
    }
    JacobiSVD64(At, rows, W, Vt, 4, rows, 4);
    for (int k = 0; k < 4; k++) points4d[4 * i + k] = (float)Vt[3 * 4 + k];
  }
  return 0;
}

int vsfo_undistort_points(const float* src, int n, const float K[9], const float dist[5], float* dst) {
  // imgproc/src/undistort.cpp cvUndistortPoints(src, dst, cameraMatrix, distCoeffs, R = NULL, P = cameraMatrix)
  // (slam_frontend.cc:334-339): camera matrix and coefficients widened to double, 5 fixed-point iterations, then
  // RR = P * I applied as (RR00*x + RR01*y + RR02) * (1 / (RR20*x + RR21*y + RR22)), result narrowed to float.
  if (n < 0 || (n > 0 && (!src || !dst)) || !K || !dist) return -1;
  double A[3][3], k[14] = {0};
  for (int i = 0; i < 9; i++) A[i / 3][i % 3] = K[i];
  for (int i = 0; i < 5; i++) k[i] = dist[i];
  const double fx = A[0][0], fy = A[1][1], ifx = 1. / fx, ify = 1. / fy, cx = A[0][2], cy = A[1][2];
  const double (*RR)[3] = A;  // matP * identity
  for (int i = 0; i < n; i++) {
    double x = src[2 * i], y = src[2 * i + 1];
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    // (k[12] = k[13] = 0: the tilt matrix is the identity; invProj = 1/1)
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
    const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
    const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
    dst[2 * i] = (float)(xx * ww);
    dst[2 * i + 1] = (float)(yy * ww);
  }
  return 0;
}

int vsfo_vision_features(const vsfo_keypoint* left, const uint8_t* left_desc, const vsfo_keypoint* right,
                         const uint8_t* right_desc, int n, double nn_match_ratio, const float P_left[12],
                         const float P_right[12], const float K_left[9], const float dist_left[5], int rows,
                         vsfo_vision_feature* out, int* n_points) {
  // slam_frontend.cc:437-443 on the two frames RemoveAmbigStereo left behind (n rows each, row i <-> row i):
  //   Calculate3DPoints (:117-173): GetFeatureMatches(right, left) with best_percent_ forced to 1.0 (:129-132) ->
  //   points in SORTED-MATCH order, (x, y, z) / w in float (:159-165);
  //   features[i] = VisionFeature(i, left.keypoints_[i].pt, points[i]) (:438-442) -- indexed by KEYPOINT although
  //   `points` is in match order and may be shorter (quirk Q5: out-of-range read in the reference; a zero point here);
  //   UndistortFeaturePoints (:323-351).
  if (n < 0 || !out) return -1;
  std::vector<vsfo_dmatch> m((size_t)std::max(n, 1));
  int nm = n > 0 ? GetMatches(right_desc, n, left_desc, n, nn_match_ratio, m.data(), n, 1) : 0;
  nm = vsfo_sort_and_trim(m.data(), nm, 1.0f);
  std::vector<float> lp((size_t)2 * std::max(nm, 1)), rp((size_t)2 * std::max(nm, 1)), X((size_t)4 * std::max(nm, 1));
  for (int i = 0; i < nm; i++) {
    const vsfo_keypoint& l = left[m[i].trainIdx];   // match.feature_idx_current  (cc:137)
    const vsfo_keypoint& r = right[m[i].queryIdx];  // match.feature_idx_initial  (cc:139)
    lp[2 * i] = l.x, lp[2 * i + 1] = l.y;
    rp[2 * i] = r.x, rp[2 * i + 1] = r.y;
  }
  if (nm > 0 && vsfo_triangulate_points(P_left, P_right, lp.data(), rp.data(), nm, rows, X.data()) != 0) return -1;
  std::vector<float> px((size_t)2 * std::max(n, 1)), ux((size_t)2 * std::max(n, 1));
  for (int i = 0; i < n; i++) px[2 * i] = left[i].x, px[2 * i + 1] = left[i].y;
  if (vsfo_undistort_points(px.data(), n, K_left, dist_left, ux.data()) != 0) return -1;
  for (int i = 0; i < n; i++) {
    vsfo_vision_feature f;
    f.feature_idx_lo = (uint32_t)i;
    f.feature_idx_hi = 0;
    f.pixel[0] = ux[2 * i];
    f.pixel[1] = ux[2 * i + 1];
    if (i < nm) {
      const float w = X[4 * i + 3];
      f.point3d[0] = X[4 * i] / w;
      f.point3d[1] = X[4 * i + 1] / w;
      f.point3d[2] = X[4 * i + 2] / w;
    } else {
      f.point3d[0] = f.point3d[1] = f.point3d[2] = 0.f;
    }
    out[i] = f;
  }
  if (n_points) *n_points = nm;
  return n;
}

int vsfo_bayer_bg_to_gray(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride) {
  // demosaicing.cpp Bayer2RGB_<uchar, SIMDBayerInterpolator_8u> with code = CV_BayerBG2BGR, dcn = 3 (the SIMD
  // interpolator computes the same integers as the scalar loops restated here), into a temporary BGR image ...
  if (!src || !dst || w < 1 || h < 1) return -1;
  const int dcn = 3;
  std::vector<uint8_t> bgr((size_t)w * h * dcn, 0);
  const int dst_step = w * dcn;
  const int bayer_step = (int)sstride;
  int blue = -1, start_with_green = 0;  // BayerBG
  const int size_h = h - 2, size_w = w - 2;
  if (size_h > 0) {
    const uint8_t* bayer0 = src;
    uint8_t* dst0 = bgr.data() + dst_step + dcn + 1;
    for (int i = 0; i < size_h; bayer0 += bayer_step, dst0 += dst_step, ++i) {
      int t0, t1;
      const uint8_t* bayer = bayer0;
      uint8_t* d = dst0;
      const uint8_t* bayer_end = bayer + size_w;
      if (size_w <= 0) {
        d[-4] = d[-3] = d[-2] = d[size_w * dcn - 1] = d[size_w * dcn] = d[size_w * dcn + 1] = 0;
        blue = -blue;  // (the reference `continue`s before the flips; nothing is computed on such images anyway)
        start_with_green = !start_with_green;
        continue;
      }
      if (start_with_green) {
        t0 = (bayer[1] + bayer[bayer_step * 2 + 1] + 1) >> 1;
        t1 = (bayer[bayer_step] + bayer[bayer_step + 2] + 1) >> 1;
        d[-blue] = (uint8_t)t0;
        d[0] = bayer[bayer_step + 1];
        d[blue] = (uint8_t)t1;
        bayer++;
        d += dcn;
      }
      for (; bayer <= bayer_end - 2; bayer += 2, d += 2 * dcn) {
        t0 = (bayer[0] + bayer[2] + bayer[bayer_step * 2] + bayer[bayer_step * 2 + 2] + 2) >> 2;
        t1 = (bayer[1] + bayer[bayer_step] + bayer[bayer_step + 2] + bayer[bayer_step * 2 + 1] + 2) >> 2;
        d[-blue] = (uint8_t)t0;
        d[0] = (uint8_t)t1;
        d[blue] = bayer[bayer_step + 1];
        t0 = (bayer[2] + bayer[bayer_step * 2 + 2] + 1) >> 1;
        t1 = (bayer[bayer_step + 1] + bayer[bayer_step + 3] + 1) >> 1;
        d[3 - blue] = (uint8_t)t0;  // (blue > 0: dst[2] = t0, dst[4] = t1; blue < 0: dst[4] = t0, dst[2] = t1)
        d[3] = bayer[bayer_step + 2];
        d[3 + blue] = (uint8_t)t1;
      }
      if (bayer < bayer_end) {  // one pixel left at the end of the row
        t0 = (bayer[0] + bayer[2] + bayer[bayer_step * 2] + bayer[bayer_step * 2 + 2] + 2) >> 2;
        t1 = (bayer[1] + bayer[bayer_step] + bayer[bayer_step + 2] + bayer[bayer_step * 2 + 1] + 2) >> 2;
        d[-blue] = (uint8_t)t0;
        d[0] = (uint8_t)t1;
        d[blue] = bayer[bayer_step + 1];
        bayer++;
        d += dcn;
      }
      // the first and the last pixel of the row
      dst0[-4] = dst0[-1];
      dst0[-3] = dst0[0];
      dst0[-2] = dst0[1];
      dst0[size_w * dcn - 1] = dst0[size_w * dcn - 4];
      dst0[size_w * dcn] = dst0[size_w * dcn - 3];
      dst0[size_w * dcn + 1] = dst0[size_w * dcn - 2];
      blue = -blue;
      start_with_green = !start_with_green;
    }
  }
  // the first and the last row
  if (h > 2) {
    for (int i = 0; i < w * dcn; i++) {
      bgr[i] = bgr[i + dst_step];
      bgr[i + (size_t)(h - 1) * dst_step] = bgr[i + (size_t)(h - 2) * dst_step];
    }
  } else {
    for (int i = 0; i < w * dcn; i++) bgr[i] = bgr[i + (size_t)(h - 1) * dst_step] = 0;
  }
  // ... then color.cpp RGB2Gray<uchar> (blueIdx = 0): tab-based  (b*B2Y + g*G2Y + r*R2Y + (1 << 13)) >> 14
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const uint8_t* p = &bgr[((size_t)y * w + x) * dcn];
      dst[(size_t)y * dstride + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
    }
  return 0;
}

}  // extern "C"
