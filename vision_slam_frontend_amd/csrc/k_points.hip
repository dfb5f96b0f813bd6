// k_points.hip -- SURVEY.md section 8(f) row f2: the tail of Frontend::ObserveImage that fills VisionFeature
// (slam_frontend.cc:437-443) for a batch of frames resident in HBM, one lane per feature:
//
//   Frontend::Calculate3DPoints (cc:117-173): the right'->left' matches sorted by distance (best_percent_ forced to 1,
//       done by vsf_feature_matches_batch_dev before this kernel) are triangulated in SORTED-MATCH order by
//       cv::triangulatePoints (calib3d/src/triangulate.cpp cvTriangulatePoints): per point the 6 x 4 system
//           x*P[2]-P[0],  y*P[2]-P[1],  x*P[1]-y*P[0]      per view      (OpenCV <= 3.4.1; 4 x 4 without the third row later)
//       in double, its right singular vector of the smallest singular value by cv::SVD::compute
//       (core/src/lapack.cpp JacobiSVDImpl_<double>: one-sided Hestenes rotations, eps = 10 * DBL_EPSILON, at most
//       max(m, 30) sweeps, descending sort), narrowed to float (the 4 x n output has the points' type), then
//       (x, y, z) / w in float (cc:159-165);
//   features[i] = VisionFeature(i, left.keypoints_[i].pt, points[i]) (cc:438-442): `points` is in match order and may
//       be shorter than the keypoint list -- the reference reads out of range there (quirk Q5); a zero point here;
//   Frontend::UndistortFeaturePoints (cc:323-351): cv::undistortPoints(pts, K_left, dist_left, noArray(), K_left)
//       (imgproc/src/undistort.cpp cvUndistortPoints): 5 fixed-point iterations in double, re-projection with K_left.
//
// Floating point, not integer work: the parity bar is a tolerance (tests/test_gpu_points.py: 1e-5 relative on point3d,
// 1e-4 px on pixel), since OpenCV's SVD may run through LAPACK in a given build.  The arithmetic still follows OpenCV's
// statement order with contraction off.
#include <cfloat>

#include "vsf_internal.h"

#pragma clang fp contract(off)

namespace {

struct PointsArgs {
  float P[2][12];   // projection_left, projection_right (row-major 3 x 4)
  float K[9];       // camera_matrix_left
  float dist[5];    // k1 k2 p1 p2 k3
  int rows;         // 6 or 4
};

__device__ __forceinline__ double lapack_hypot(double a, double b) {  // core/src/lapack.cpp hypot<_Tp>
  a = fabs(a);
  b = fabs(b);
  if (a > b) {
    b /= a;
    return a * sqrt(1 + b * b);
  }
  if (b > 0) {
    a /= b;
    return b * sqrt(1 + a * a);
  }
  return 0;
}

// Right singular vector of the smallest singular value of the M x 4 matrix whose COLUMNS are At[0..3] (M = 6 or 4;
// unused rows are zero, which changes no sum).  JacobiSVDImpl_<double>, every index a compile-time constant so the
// arrays live in registers.
__device__ __forceinline__ void smallest_right_singular_vector(double (&At)[4][6], double (&X)[4]) {
  const double eps = DBL_EPSILON * 10;
  double W[4], Vt[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) sd += At[i][k] * At[i][k];
    W[i] = sd;
#pragma unroll
    for (int k = 0; k < 4; k++) Vt[i][k] = i == k ? 1.0 : 0.0;
  }
  for (int iter = 0; iter < 30; iter++) {  // max_iter = max(m, 30) = 30
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = i + 1; j < 4; j++) {
        double a = W[i], p = 0, b = W[j];
#pragma unroll
        for (int k = 0; k < 6; k++) p += At[i][k] * At[j][k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = lapack_hypot(p, beta);
        double c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const double t0 = c * At[i][k] + s * At[j][k];
          const double t1 = -s * At[i][k] + c * At[j][k];
          At[i][k] = t0;
          At[j][k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const double t0 = c * Vt[i][k] + s * Vt[j][k];
          const double t1 = -s * Vt[i][k] + c * Vt[j][k];
          Vt[i][k] = t0;
          Vt[j][k] = t1;
        }
      }
    if (!changed) break;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) sd += At[i][k] * At[i][k];
    W[i] = sqrt(sd);
  }
  // The descending selection sort leaves the row of the smallest W in row 3 (which one among exactly equal singular
  // values is immaterial: the null space is then two-dimensional and the point undefined).
  int best = 0;
#pragma unroll
  for (int i = 1; i < 4; i++)
    if (W[i] < W[best]) best = i;
#pragma unroll
  for (int k = 0; k < 4; k++) X[k] = best == 0 ? Vt[0][k] : best == 1 ? Vt[1][k] : best == 2 ? Vt[2][k] : Vt[3][k];
}

__global__ __launch_bounds__(64) void vision_features_kernel(const vsf_keypoint* __restrict__ kp,  // [2*frames][max_rows]
                                                              const int32_t* __restrict__ counts,   // [2*frames]
                                                              const uint64_t* __restrict__ pairs,   // [frames][max_rows][2]
                                                              const int32_t* __restrict__ npairs,   // [frames]
                                                              int max_rows, PointsArgs a,
                                                              vsf_vision_feature* __restrict__ out,  // [frames][max_rows]
                                                              int32_t* __restrict__ nfeatures,       // [frames]
                                                              int32_t* __restrict__ npoints) {       // [frames] or null
  const int f = blockIdx.y;
  const int n = min(counts[2 * f], max_rows);
  const int m = min(npairs[f], n);
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i == 0) {
    nfeatures[f] = n;
    if (npoints) npoints[f] = m;
  }
  if (i >= n) return;
  const vsf_keypoint* left = kp + (size_t)(2 * f) * max_rows;
  const vsf_keypoint* right = kp + (size_t)(2 * f + 1) * max_rows;
  vsf_vision_feature o;
  o.feature_idx_lo = (uint32_t)i;
  o.feature_idx_hi = 0;
  o.point3d[0] = o.point3d[1] = o.point3d[2] = 0.f;
  if (i < m) {
    const uint64_t* pr = pairs + ((size_t)f * max_rows + i) * 2;
    const int ri = (int)pr[0], li = (int)pr[1];  // feature_idx_initial = right row, feature_idx_current = left row
    const double xy[2][2] = {{(double)left[li].x, (double)left[li].y}, {(double)right[ri].x, (double)right[ri].y}};
    double At[4][6];
    const int per = a.rows / 2;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int r = 0; r < 6; r++) At[k][r] = 0.0;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const double x = xy[j][0], y = xy[j][1];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double p0 = (double)a.P[j][k], p1 = (double)a.P[j][4 + k], p2 = (double)a.P[j][8 + k];
        const double r0 = x * p2 - p0, r1 = y * p2 - p1, r2 = x * p1 - y * p0;
        if (per == 3) {
          At[k][3 * j + 0] = r0;
          At[k][3 * j + 1] = r1;
          At[k][3 * j + 2] = r2;
        } else {
          At[k][2 * j + 0] = r0;
          At[k][2 * j + 1] = r1;
        }
      }
    }
    double X[4];
    smallest_right_singular_vector(At, X);
    const float xf = (float)X[0], yf = (float)X[1], zf = (float)X[2], wf = (float)X[3];
    o.point3d[0] = xf / wf;
    o.point3d[1] = yf / wf;
    o.point3d[2] = zf / wf;
  }
  {
    // cvUndistortPoints: camera matrix and coefficients widened to double
    const double fx = (double)a.K[0], fy = (double)a.K[4], cx = (double)a.K[2], cy = (double)a.K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = (double)a.dist[0], k1 = (double)a.dist[1], k2 = (double)a.dist[2], k3 = (double)a.dist[3],
                 k4 = (double)a.dist[4];
    double x = (double)left[i].x, y = (double)left[i].y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      // (k5..k13 are zero for a five-element coefficient vector: the numerator and the extra delta terms stay as 1 and 0)
      const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
      const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0.0 * r2 + 0.0 * r2 * r2;
      const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0.0 * r2 + 0.0 * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    // RR = P * I with P = K_left
    const double xx = (double)a.K[0] * x + (double)a.K[1] * y + (double)a.K[2];
    const double yy = (double)a.K[3] * x + (double)a.K[4] * y + (double)a.K[5];
    const double ww = 1. / ((double)a.K[6] * x + (double)a.K[7] * y + (double)a.K[8]);
    o.pixel[0] = (float)(xx * ww);
    o.pixel[1] = (float)(yy * ww);
  }
  out[(size_t)f * max_rows + i] = o;
}

// ---- compact gather payload: counts first, records sized by the counts ----
// header: u32 magic, n_frames, n_pairs, total_bytes | u32 nfeatures[n_frames] | u32 npairs[n_pairs] |
// vsf_vision_feature records frame after frame (28 B each) | vsf_feature_match records pair after pair (16 B each)
__global__ __launch_bounds__(1024) void pack_offsets_kernel(const int32_t* __restrict__ nfeatures, int n_frames,
                                                            const int32_t* __restrict__ npairs, int n_pairs, int max_rows,
                                                            uint32_t* __restrict__ payload_words, uint32_t cap_bytes,
                                                            uint32_t* __restrict__ offsets,  // [n_frames + n_pairs] (bytes)
                                                            int32_t* __restrict__ status) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const int n = n_frames + n_pairs;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 16u + 4u * (uint32_t)n;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    uint32_t c = 0, bytes = 0;
    if (i < n) {
      c = (uint32_t)min(max(i < n_frames ? nfeatures[i] : npairs[i - n_frames], 0), max_rows);
      bytes = c * (i < n_frames ? (uint32_t)sizeof(vsf_vision_feature) : (uint32_t)sizeof(vsf_feature_match));
      payload_words[4 + i] = c;
    }
    uint32_t inc = bytes;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t base = carry, tot = 0;
    for (int w = 0; w < 16; w++) {
      if (w < wid) base += wsum[w];
      tot += wsum[w];
    }
    if (i < n) offsets[i] = base + inc - bytes;
    __syncthreads();
    if (threadIdx.x == 0) carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    payload_words[0] = 0x31465356u;  // "VSF1"
    payload_words[1] = (uint32_t)n_frames;
    payload_words[2] = (uint32_t)n_pairs;
    payload_words[3] = carry;
    if (carry > cap_bytes) atomicOr(status, 1);
  }
}

__global__ __launch_bounds__(256) void pack_copy_kernel(const vsf_vision_feature* __restrict__ features, int n_frames,
                                                        const uint64_t* __restrict__ pairs, int n_pairs, int max_rows,
                                                        uint8_t* __restrict__ payload, uint32_t cap_bytes,
                                                        const uint32_t* __restrict__ offsets) {
  const int i = blockIdx.x;
  const uint32_t* pw = reinterpret_cast<const uint32_t*>(payload);
  const uint32_t count = pw[4 + i], off = offsets[i];
  const bool feat = i < n_frames;
  const uint32_t words = count * (feat ? 7u : 4u);
  if (off + 4u * words > cap_bytes) return;  // (flagged by pack_offsets_kernel)
  const uint32_t* src = feat ? reinterpret_cast<const uint32_t*>(features + (size_t)i * max_rows)
                             : reinterpret_cast<const uint32_t*>(pairs + (size_t)(i - n_frames) * max_rows * 2);
  uint32_t* dst = reinterpret_cast<uint32_t*>(payload + off);
  for (uint32_t w = threadIdx.x; w < words; w += 256) dst[w] = src[w];
}

// ---- thr[0] = *state, thr[k] = mean[k-1] + 2, *state = mean[n-1] + 2  (cc:392-394; no chain: each threshold depends on
// ONE mean, NaN included -- quirk Q3) ----
__global__ void stereo_thresholds_kernel(const float* __restrict__ mean, int n, float* __restrict__ state,
                                         float* __restrict__ thr) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  if (k == 0) {
    thr[0] = *state;
    *state = mean[n - 1] + 2.0f;
  } else {
    thr[k] = mean[k - 1] + 2.0f;
  }
}

// ---- the (query, train) set indices of Calculate3DPoints' right -> left matching: q[f] = 2f + 1, t[f] = 2f ----
__global__ void fill_stereo_sets_kernel(int32_t* __restrict__ sets, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    sets[k] = 2 * k + 1;
    sets[n + k] = 2 * k;
  }
}

}  // namespace

void vsf_launch_fill_stereo_sets(int32_t* d_sets, int n_frames, hipStream_t s) {
  hipLaunchKernelGGL(fill_stereo_sets_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, s, d_sets, n_frames);
}

void vsf_launch_vision_features(const vsf_keypoint* d_kp, const int32_t* d_counts, const uint64_t* d_pairs,
                                const int32_t* d_npairs, int n_frames, int max_rows, const vsf_calibration& c,
                                vsf_vision_feature* d_out, int32_t* d_nfeatures, int32_t* d_npoints, hipStream_t s) {
  PointsArgs a;
  for (int i = 0; i < 12; i++) {
    a.P[0][i] = c.projection_left[i];
    a.P[1][i] = c.projection_right[i];
  }
  for (int i = 0; i < 9; i++) a.K[i] = c.camera_matrix_left[i];
  for (int i = 0; i < 5; i++) a.dist[i] = c.distortion_left[i];
  a.rows = c.triangulate_rows == 4 ? 4 : 6;
  hipLaunchKernelGGL(vision_features_kernel, dim3((max_rows + 63) / 64, n_frames), dim3(64), 0, s, d_kp, d_counts, d_pairs,
                     d_npairs, max_rows, a, d_out, d_nfeatures, d_npoints);
}

void vsf_launch_pack_outputs(const vsf_vision_feature* d_features, const int32_t* d_nfeatures, int n_frames,
                             const uint64_t* d_pairs, const int32_t* d_npairs, int n_pairs, int max_rows,
                             uint8_t* d_payload, uint32_t cap_bytes, uint32_t* d_offsets, int32_t* d_status,
                             hipStream_t s) {
  hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, s, d_nfeatures, n_frames, d_npairs, n_pairs, max_rows,
                     reinterpret_cast<uint32_t*>(d_payload), cap_bytes, d_offsets, d_status);
  hipLaunchKernelGGL(pack_copy_kernel, dim3(n_frames + n_pairs), dim3(256), 0, s, d_features, n_frames, d_pairs, n_pairs,
                     max_rows, d_payload, cap_bytes, d_offsets);
}

void vsf_launch_stereo_thresholds(const float* d_means, int n, float* d_state, float* d_thr, hipStream_t s) {
  hipLaunchKernelGGL(stereo_thresholds_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_means, n, d_state, d_thr);
}
