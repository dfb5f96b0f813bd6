// k_fast.hip -- K2 (+ the border half of K3): FAST-9/16 segment test, cornerScore, 3x3 non-max suppression and
// raster-order compaction, for every level of every image in ONE launch.
//
// Restates cv::FAST_t<16> and cornerScore<16> (features2d/fast.cpp, fast_score.cpp) as called by
// ORB's computeKeyPoints (threshold 20; slam_frontend.cc:274, parameters :205-213) and by
// FastFeatureDetector::detect (threshold 10, slam_frontend.cc:191,271), plus
// KeyPointsFilter::runByImageBorder(31) which is folded into the evaluated rectangle.
//
// Streaming march kernel -- no LDS, no barriers, no divergent corner/score phases:
//  * a wave owns a band of 240 keypoint columns x a strip of 32 rows of one level of one image and walks down the
//    rows; each lane holds 4 adjacent pixels per row (one coalesced 32-bit load) in a 7-row register window and
//    gets its neighbours' dwords by DPP wave shifts;
//  * corner test and score are ONE dense computation: with d_k = p_k - v on the 16-pixel circle,
//      A = max over the 16 arcs of min(d over the 9-arc),  B = -min over arcs of max(d over the arc),
//    the pixel is a FAST-9 corner iff max(A, B) > t and cornerScore is max(A, B) - 1.  Two pixels are processed per
//    VALU lane-op with packed 16-bit min/max (v_pk_min_i16 / v_pk_max_i16), circle bytes are pulled out of the
//    window with v_perm_b32; there is no data-dependent branch except a wave-wide early-out for flat rows;
//  * scores stay in registers (3-row rolling window, 4 score bytes per lane); the strict 8-neighbour NMS reads the
//    neighbours by DPP, and survivors are appended in raster order with ballot prefix ranks.
// Output per unit: a candidate segment in unit-local raster order + the start offset of every row, which
// vsf_gather.h merges into the level's global raster order.
#include "vsf_gather.h"
#include "vsf_internal.h"

namespace {

typedef short v2s __attribute__((ext_vector_type(2)));
constexpr int SR = VSF_FAST_STRIP_ROWS;

struct FastArgs {
  const VsfLevel* levels;
  const uint32_t* units;
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
  int nms;
};

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {  // lane i <- lane i-1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) {  // lane i <- lane i+1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false);
}

// One image row as seen by a lane: its own 4 pixels (d) and the neighbouring lanes' (p = left, n = right).
struct Row3 {
  uint32_t p, d, n;
};

// Bytes B and B+1 of the 12-byte run {p, d, n}, zero-extended into the two 16-bit halves.
template <int B>
__device__ __forceinline__ v2s pick2(const Row3& r) {
  static_assert(B >= 0 && B + 1 <= 11, "byte range");
  if constexpr (B + 1 <= 7) {
    constexpr uint32_t sel = (uint32_t)B | 0x0C000C00u | ((uint32_t)(B + 1) << 16);
    return __builtin_bit_cast(v2s, __builtin_amdgcn_perm(r.d, r.p, sel));
  } else {
    constexpr uint32_t sel = (uint32_t)(B - 4) | 0x0C000C00u | ((uint32_t)(B - 3) << 16);
    return __builtin_bit_cast(v2s, __builtin_amdgcn_perm(r.n, r.d, sel));
  }
}

__device__ __forceinline__ v2s vmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s vmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }

// Scores of the two pixels at bytes 4+J0, 5+J0 of the centre row (W[3]); W[0..6] are rows y-3..y+3.
// Returns the two 8-bit scores (0 = not a corner) in bits 0..7 and 8..15.
template <int J0>
__device__ __forceinline__ uint32_t score_pair(const Row3 (&W)[7], int t, int nms) {
  const v2s v = pick2<4 + J0>(W[3]);
  v2s d[16];
  // Bresenham circle, OpenCV order: (dx,dy) = (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)
  // (-3,0)(-3,1)(-2,2)(-1,3); window row index = 3 + dy.
  d[0] = pick2<4 + J0 + 0>(W[6]) - v;
  d[1] = pick2<4 + J0 + 1>(W[6]) - v;
  d[2] = pick2<4 + J0 + 2>(W[5]) - v;
  d[3] = pick2<4 + J0 + 3>(W[4]) - v;
  d[4] = pick2<4 + J0 + 3>(W[3]) - v;
  d[5] = pick2<4 + J0 + 3>(W[2]) - v;
  d[6] = pick2<4 + J0 + 2>(W[1]) - v;
  d[7] = pick2<4 + J0 + 1>(W[0]) - v;
  d[8] = pick2<4 + J0 + 0>(W[0]) - v;
  d[9] = pick2<4 + J0 - 1>(W[0]) - v;
  d[10] = pick2<4 + J0 - 2>(W[1]) - v;
  d[11] = pick2<4 + J0 - 3>(W[2]) - v;
  d[12] = pick2<4 + J0 - 3>(W[3]) - v;
  d[13] = pick2<4 + J0 - 3>(W[4]) - v;
  d[14] = pick2<4 + J0 - 2>(W[5]) - v;
  d[15] = pick2<4 + J0 - 1>(W[6]) - v;
  v2s n1[16], x1[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    n1[k] = vmin(d[k], d[(k + 1) & 15]);
    x1[k] = vmax(d[k], d[(k + 1) & 15]);
  }
  v2s n2[16], x2[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    n2[k] = vmin(n1[k], n1[(k + 2) & 15]);
    x2[k] = vmax(x1[k], x1[(k + 2) & 15]);
  }
  v2s A = {-32768, -32768}, Bm = {32767, 32767};
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const v2s n9 = vmin(vmin(n2[k], n2[(k + 4) & 15]), d[(k + 8) & 15]);
    const v2s x9 = vmax(vmax(x2[k], x2[(k + 4) & 15]), d[(k + 8) & 15]);
    A = vmax(A, n9);
    Bm = vmin(Bm, x9);
  }
  // cornerScore<16>: max(t, A, -Bm) - 1; a corner iff that maximum exceeds t.  With NMS a corner of score 0 can
  // never win (cv::FAST_t compares strictly against neighbours >= 0); without NMS only a corner marker is kept
  // (cv::FAST_t leaves the response at 0 then).
  const int a0 = A.x, a1 = A.y, b0 = -(int)Bm.x, b1 = -(int)Bm.y;
  const int s0 = max(a0, b0), s1 = max(a1, b1);
  const uint32_t r0 = s0 > t ? (nms ? (uint32_t)(s0 - 1) : 1u) : 0u, r1 = s1 > t ? (nms ? (uint32_t)(s1 - 1) : 1u) : 0u;
  return r0 | (r1 << 8);
}

// True if some pixel of the lane's four could be a corner: a 9-arc always contains circle pixel 0 or 8.
__device__ __forceinline__ bool maybe_corner(const Row3 (&W)[7], int t) {
  const uint32_t c = W[3].d, up = W[0].d, dn = W[6].d;
  bool any = false;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int v = (c >> (8 * j)) & 255, a = (up >> (8 * j)) & 255, b = (dn >> (8 * j)) & 255;
    any |= (abs(a - v) > t) | (abs(b - v) > t);
  }
  return any;
}

template <int B>
__device__ __forceinline__ int byte_of(const Row3& r) {
  if constexpr (B < 4) {
    return (int)((r.p >> (8 * B)) & 255u);
  } else if constexpr (B < 8) {
    return (int)((r.d >> (8 * (B - 4))) & 255u);
  } else {
    return (int)((r.n >> (8 * (B - 8))) & 255u);
  }
}

template <int J>
__device__ __forceinline__ bool nms_keep(const Row3& up, const Row3& mid, const Row3& dn, int nms) {
  const int s = byte_of<4 + J>(mid);
  if (s == 0) return false;
  if (!nms) return true;
  return s > byte_of<3 + J>(mid) && s > byte_of<5 + J>(mid) && s > byte_of<3 + J>(up) && s > byte_of<4 + J>(up) &&
         s > byte_of<5 + J>(up) && s > byte_of<3 + J>(dn) && s > byte_of<4 + J>(dn) && s > byte_of<5 + J>(dn);
}

__global__ __launch_bounds__(256) void fast_march_kernel(FastArgs a) {
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
  const int t = a.threshold, nms = a.nms;
  const int bx0 = L.fast_a0 + VSF_FAST_BAND_COLS * band;
  const int c0 = bx0 - 8 + 4 * lane;  // first column of this lane's 4 pixels
  const bool loadable = c0 >= 0 && c0 + 3 < pitch;
  const int ys = L.y_lo + strip * SR, ye = min(ys + SR, L.y_hi);
  // Pixels that may carry a score: FAST's 3-pixel rim and one column beyond the keypoint rectangle (for the NMS).
  const int sx_lo = max(L.x_lo - 1, 3), sx_hi = min(L.x_hi + 1, L.w - 3);
  uint32_t smask = 0, emask = 0;  // per-pixel byte masks: may be scored / may be emitted
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int x = c0 + j;
    if (lane >= 1 && lane <= 62 && x >= sx_lo && x < sx_hi) smask |= 0xFFu << (8 * j);
    if (lane >= 2 && lane <= 61 && x >= L.x_lo && x < L.x_hi && x < bx0 + VSF_FAST_BAND_COLS) emask |= 0xFFu << (8 * j);
  }
  const int unit_local = strip * L.nbands + band;
  uint32_t* seg = a.cand + (size_t)image * a.cand_entries + L.cand_offset + (size_t)unit_local * L.seg_cap;
  uint16_t* rs = a.rowstart + ((size_t)image * a.nunits + L.unit0 + unit_local) * VSF_FAST_RS_STRIDE;

  auto load_row = [&](int y) -> Row3 {
    Row3 r;
    const int yc = min(max(y, 0), L.h - 1);
    r.d = loadable ? *reinterpret_cast<const uint32_t*>(src + (size_t)yc * pitch + c0) : 0u;
    r.p = wave_shr1(r.d);
    r.n = wave_shl1(r.d);
    return r;
  };

  Row3 W[7];  // rows sy-3 .. sy+3 around the score row sy
  int sy = ys - 1;
#pragma unroll
  for (int i = 0; i < 6; i++) W[i + 1] = load_row(sy - 3 + i);
  Row3 next = load_row(sy + 3);
  Row3 S_up = {0, 0, 0}, S_mid = {0, 0, 0};  // score rows sy-2, sy-1 (4 score bytes per lane + neighbours)
  int count = 0;   // candidates emitted so far (wave-uniform)
  int my_rs = 0;   // lane l keeps rowstart[l]
  for (; sy <= ye; sy++) {
#pragma unroll
    for (int i = 0; i < 6; i++) W[i] = W[i + 1];
    W[6] = next;
    next = load_row(sy + 4);  // prefetch the next iteration's new row
    // ---- scores of row sy ----
    uint32_t S = 0;
    const bool row_ok = sy >= 3 && sy < L.h - 3;  // wave-uniform
    if (row_ok && __any(smask != 0 && maybe_corner(W, t))) {
      S = score_pair<0>(W, t, nms) | (score_pair<2>(W, t, nms) << 16);
      S &= smask;
    }
    Row3 S_dn;
    S_dn.d = S;
    S_dn.p = wave_shr1(S);
    S_dn.n = wave_shl1(S);
    // ---- NMS + emission of row sy-1 ----
    const int y = sy - 1;
    if (y >= ys && y < ye) {
      if (__any((S_mid.d & emask) != 0)) {
        const bool k0 = (emask & 0xFFu) && nms_keep<0>(S_up, S_mid, S_dn, nms);
        const bool k1 = (emask & 0xFF00u) && nms_keep<1>(S_up, S_mid, S_dn, nms);
        const bool k2 = (emask & 0xFF0000u) && nms_keep<2>(S_up, S_mid, S_dn, nms);
        const bool k3 = (emask & 0xFF000000u) && nms_keep<3>(S_up, S_mid, S_dn, nms);
        const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1), b2 = __ballot(k2), b3 = __ballot(k3);
        const unsigned long long lt = (1ull << lane) - 1ull;
        int pos = count + __popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt);
        const uint32_t sc = nms ? S_mid.d : 0u;
        if (k0) {
          if (pos < L.seg_cap) seg[pos] = VSF_CAND_PACK(c0 + 0, y, sc & 255u);
          ++pos;
        }
        if (k1) {
          if (pos < L.seg_cap) seg[pos] = VSF_CAND_PACK(c0 + 1, y, (sc >> 8) & 255u);
          ++pos;
        }
        if (k2) {
          if (pos < L.seg_cap) seg[pos] = VSF_CAND_PACK(c0 + 2, y, (sc >> 16) & 255u);
          ++pos;
        }
        if (k3) {
          if (pos < L.seg_cap) seg[pos] = VSF_CAND_PACK(c0 + 3, y, sc >> 24);
        }
        count += __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
      }
      if (lane > y - ys) my_rs = min(count, L.seg_cap);
    }
    S_up = S_mid;
    S_mid = S_dn;
  }
  if (lane <= SR) rs[lane] = (uint16_t)my_rs;
}

// Standalone FAST detect: unit segments -> contiguous cv::KeyPoint list in raster order.
__global__ __launch_bounds__(256) void fast_emit_kernel(const VsfLevel* __restrict__ levels,
                                                        const uint32_t* __restrict__ cand, uint32_t cand_entries,
                                                        const uint16_t* __restrict__ rowstart, int nunits,
                                                        int max_keypoints, vsf_keypoint* __restrict__ out,
                                                        int32_t* __restrict__ counts, int32_t* __restrict__ status) {
  __shared__ int cellpre[2048 + 8];
  __shared__ int lds4[4];
  const int image = blockIdx.x;
  const VsfLevel L = levels[0];
  const uint16_t* rs_img = rowstart + (size_t)image * nunits * VSF_FAST_RS_STRIDE;
  const uint32_t* cand_img = cand + (size_t)image * cand_entries;
  const int n = vsf_level_candidate_count<256>(L, rs_img, lds4);
  vsf_keypoint* o = out + (size_t)image * max_keypoints;
  vsf_gather_level<256>(L, cand_img, rs_img, cellpre, 2048, lds4, [&](int dst, uint32_t cd) {
    if (dst < max_keypoints) {
      vsf_keypoint kp;
      kp.x = (float)VSF_CAND_X(cd);
      kp.y = (float)VSF_CAND_Y(cd);
      kp.size = 7.f;
      kp.angle = -1.f;
      kp.response = (float)VSF_CAND_SCORE(cd);
      kp.octave = 0;
      kp.class_id = -1;
      o[dst] = kp;
    }
  });
  if (threadIdx.x == 0) {
    counts[image] = n;  // true count; the caller clamps to its capacity
    if (n > max_keypoints) atomicOr(status, 1);
  }
}

}  // namespace

void vsf_launch_fast(const VsfDev& d, const VsfGeom& g, const VsfImages& im, int threshold, int nms, hipStream_t s) {
  FastArgs a;
  a.levels = d.levels;
  a.units = d.units;
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
  a.nms = nms;
  hipLaunchKernelGGL(fast_march_kernel, dim3((g.nunits + 3) / 4, im.n), dim3(256), 0, s, a);
}

void vsf_launch_fast_emit(const VsfDev& d, const VsfGeom& g, int n_images, int max_keypoints, vsf_keypoint* d_kp,
                          int32_t* d_counts, hipStream_t s) {
  hipLaunchKernelGGL(fast_emit_kernel, dim3(n_images), dim3(256), 0, s, d.levels, d.cand, g.cand_entries, d.rowstart,
                     g.nunits, max_keypoints, d_kp, d_counts, d.status);
}
