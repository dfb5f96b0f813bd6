/*
 * vsf_oracle.h -- C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * PARITY UNPINNED: the reference (ut-amrl/vision_slam_frontend) delegates the arithmetic of
 * its hot path to OpenCV 3.2.0 (CMakeLists.txt:20), which is neither vendored in the
 * reference nor installed in this image, and the reference has no tests, fixtures or golden
 * vectors.  This oracle is a scalar restatement of the published
 * OpenCV-3.2.0 algorithms reached from src/slam_frontend.cc:266-280 (ExtractFeatures) and
 * :521-538 (GetMatches), plus the reference's own logic at :282-309 and :353-398.  It is
 * pinned only by definition-level known-answer tests (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.
 */
#ifndef VSF_ORACLE_H_
#define VSF_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint memory layout (28 bytes). */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} vsfo_keypoint;

/* cv::DMatch memory layout (16 bytes). */
typedef struct {
  int32_t queryIdx, trainIdx, imgIdx;
  float distance;
} vsfo_dmatch;

/* Arguments of cv::ORB::create (reference literals: slam_frontend.cc:205-213). */
typedef struct {
  int32_t nfeatures;      /* 10000 in the reference; 2000 / 8000 in BASELINE configs */
  float scale_factor;     /* 1.04f */
  int32_t nlevels;        /* 50 */
  int32_t edge_threshold; /* 31 */
  int32_t first_level;    /* 0 (only 0 is supported) */
  int32_t wta_k;          /* 2 (only 2 is supported) */
  int32_t score_type;     /* 0 = HARRIS_SCORE (only) */
  int32_t patch_size;     /* 31 (only) */
  int32_t fast_threshold; /* 20 */
  int32_t blur_sse2;      /* 1: SymmColumnVec_32s8u rounding (half-even) on [0,w-w%4), scalar tail;
                             0: scalar FixedPtCastEx (half-up) everywhere */
} vsfo_orb_params;

void vsfo_orb_params_default(vsfo_orb_params* p);

/* ---- building blocks (each restates one OpenCV routine) ---- */

/* cv::resize(INTER_LINEAR) for CV_8UC1 (imgproc/imgwarp.cpp). */
int vsfo_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw,
                          int dh, size_t dstride);
/* The coefficient tables resize builds: xofs[dw], ialpha[2*dw], yofs[dh], ibeta[2*dh]; returns xmax. */
int vsfo_resize_tables(int sw, int sh, int dw, int dh, int32_t* xofs, int16_t* ialpha, int32_t* yofs,
                       int16_t* ibeta);

/* cv::FAST(img, kps, threshold, nms) with TYPE_9_16 (features2d/fast.cpp). Returns count (may exceed cap;
 * only cap are written). */
int vsfo_fast9_16(const uint8_t* img, int w, int h, size_t stride, int threshold, int nms,
                  vsfo_keypoint* out, int cap);
/* cornerScore<16> for one pixel (features2d/fast_score.cpp); pixel must be >= 3 px from every edge. */
int vsfo_fast_corner_score(const uint8_t* img, size_t stride, int x, int y, int threshold);

/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on CV_8UC1, out of place. */
int vsfo_gaussian_blur7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride,
                        int sse2_rounding);
/* The 8-bit fixed-point kernel createSeparableLinearFilter derives (7 ints, sum 257). */
void vsfo_gaussian_kernel7_fixed(int32_t k[7]);

/* cv::fastAtan2(y, x) in degrees (core/mathfuncs). */
float vsfo_fast_atan2(float y, float x);

/* The rBRIEF pattern (256 x 4 int8). */
const int8_t* vsfo_orb_pattern31(void);

/* cv::KeyPointsFilter::retainBest on n (response, id) pairs with the host libstdc++ (std::nth_element +
 * std::partition), in place; returns the surviving count. */
int vsfo_retain_best(float* response, uint32_t* id, int n, int n_points);

/* ---- ORB detectAndCompute with every intermediate kept for kernel-level parity ---- */
typedef struct vsfo_orb vsfo_orb;
vsfo_orb* vsfo_orb_create(const vsfo_orb_params* p);
void vsfo_orb_destroy(vsfo_orb* o);
/* Runs detectAndCompute(image, noArray(), kps, desc). Returns the number of keypoints, <0 on error. */
int vsfo_orb_run(vsfo_orb* o, const uint8_t* img, int w, int h, size_t stride);
int vsfo_orb_nlevels(const vsfo_orb* o);
/* Level geometry (valid after a run or after vsfo_orb_layout). */
int vsfo_orb_layout(vsfo_orb* o, int w, int h);
int vsfo_orb_level_info(const vsfo_orb* o, int level, int* w, int* h, float* scale, int* nfeatures);
/* Copies level `level` (blurred=0: as used by FAST/Harris/angle; 1: as used by descriptors). */
int vsfo_orb_level_image(const vsfo_orb* o, int level, int blurred, uint8_t* out, size_t ostride);
/* Per-level keypoint lists at each stage, in level coordinates:
 *  0 FAST+NMS raster order after runByImageBorder
 *  1 after retainBest(2*n_l) on FAST score
 *  2 same list with response = Harris
 *  3 after retainBest(n_l) on Harris
 *  4 stage 3 with angle set (still level coordinates, octave = level, size = 31*scale)
 * Returns the count (writes min(count, cap)). */
int vsfo_orb_stage_keypoints(const vsfo_orb* o, int stage, int level, vsfo_keypoint* out, int cap);
/* Final outputs: keypoints (level-0 coordinates) and 32-byte descriptors. */
int vsfo_orb_result(const vsfo_orb* o, vsfo_keypoint* kps, uint8_t* desc, int cap);

/* ---- matcher: cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) + the reference's ratio test ---- */
/* idx2/dist2 are nq x 2; missing neighbours: idx -1, dist INT32_MAX. 32-byte descriptors. */
int vsfo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2,
                      int32_t* dist2);
/* Frontend::GetMatches (slam_frontend.cc:521-538); nt < 2 yields no matches (quirk Q6). */
int vsfo_get_matches(const uint8_t* q, int nq, const uint8_t* t, int nt, double nn_match_ratio,
                     vsfo_dmatch* out, int cap);
/* Same with `threads` worker threads over query rows (mirrors parallel_for_ in batchDistance). */
int vsfo_get_matches_mt(const uint8_t* q, int nq, const uint8_t* t, int nt, double nn_match_ratio,
                        vsfo_dmatch* out, int cap, int threads);
/* Frontend::GetFeatureMatches (slam_frontend.cc:282-309) minus the is_initial_ bookkeeping:
 * std::sort by distance, keep int(n * best_percent). In place; returns the kept count. */
int vsfo_sort_and_trim(vsfo_dmatch* m, int n, float best_percent);
/* Frontend::RemoveAmbigStereo (slam_frontend.cc:353-398). F is row-major 3x3. keep[i] in {0,1};
 * *threshold_io is the file-static stereo_ambig_constraint (in: current, out: updated). Returns kept. */
/* Summation order of the three-term dot products in RemoveAmbigStereo: 0 = Eigen 3.3 (a0 b0 + (a1 b1 + a2 b2), default),
 * 1 = left to right.  Process-wide switch of the checker. */
void vsfo_set_residual_order(int order);
int vsfo_get_residual_order(void);
int vsfo_remove_ambig_stereo(const vsfo_keypoint* left, const vsfo_keypoint* right,
                             const vsfo_dmatch* matches, int n, const float F[9], float* threshold_io,
                             uint8_t* keep, float* residual);

/* ---- Calculate3DPoints / UndistortFeaturePoints (slam_frontend.cc:117-173, 323-351) ---- */
/* slam_types::VisionFeature as it travels (slam_types.h:60-75; 28 bytes: u64 index, pixel, point3d). */
typedef struct {
  uint32_t feature_idx_lo, feature_idx_hi;
  float pixel[2];
  float point3d[3];
} vsfo_vision_feature;
/* cv::triangulatePoints(P1, P2, pts1, pts2) for CV_32F inputs (calib3d/src/triangulate.cpp cvTriangulatePoints +
 * core/src/lapack.cpp JacobiSVDImpl_<double>): pts are n x 2, points4d is n x 4 (column i of OpenCV's 4 x n output).
 * rows: 6 = OpenCV <= 3.4.1 (the pinned 3.2.0), 4 = later versions (third row of each view dropped). */
int vsfo_triangulate_points(const float P1[12], const float P2[12], const float* pts1, const float* pts2, int n,
                            int rows, float* points4d);
/* cv::undistortPoints(src, dst, K, dist, noArray(), K) (imgproc/src/undistort.cpp cvUndistortPoints). */
int vsfo_undistort_points(const float* src, int n, const float K[9], const float dist[5], float* dst);
/* slam_frontend.cc:437-443: Calculate3DPoints + VisionFeature(i, pt_i, points[i]) + UndistortFeaturePoints on the
 * two frames RemoveAmbigStereo rebuilt (n rows each).  Returns n; *n_points = number of triangulated points. */
int vsfo_vision_features(const vsfo_keypoint* left, const uint8_t* left_desc, const vsfo_keypoint* right,
                         const uint8_t* right_desc, int n, double nn_match_ratio, const float P_left[12],
                         const float P_right[12], const float K_left[9], const float dist_left[5], int rows,
                         vsfo_vision_feature* out, int* n_points);

/* ---- image ingest (the part after cv::imdecode): slam_frontend_main.cc:101-106 ----
 * cv::cvtColor(COLOR_BayerBG2BGR) (imgproc/demosaicing.cpp Bayer2RGB_<uchar>: bilinear, the one-pixel frame copied
 * from its neighbours) followed by cv::cvtColor(COLOR_BGR2GRAY) (color.cpp RGB2Gray<uchar>: (1868 B + 9617 G + 4899 R +
 * 8192) >> 14).  src: w x h mosaic, dst: w x h gray.  Images narrower or lower than 3 pixels come out zero, as OpenCV's
 * loops leave them. */
int vsfo_bayer_bg_to_gray(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride);

/* DecodeImage's cv::imdecode(data, IMREAD_GRAYSCALE) for baseline JPEG (slam_frontend_main.cc:99-100): T.81 Huffman
 * decoding + libjpeg's ISLOW inverse DCT of the luminance component (vsf_oracle_jpeg.cc; PINNED by libjpeg-turbo-decoded
 * fixtures, tests/golden/jpeg).  out: cap_w x cap_h bytes at `ostride`.  Returns 0; -1 malformed; -2 a JPEG process not
 * restated (progressive, arithmetic, 12 bit, multi-scan); -3 does not fit (w_out / h_out still report the size). */
int vsfo_jpeg_decode_gray(const uint8_t* data, size_t nbytes, uint8_t* out, size_t ostride, int cap_w, int cap_h,
                          int* w_out, int* h_out);

#ifdef __cplusplus
}
#endif
#endif /* VSF_ORACLE_H_ */
