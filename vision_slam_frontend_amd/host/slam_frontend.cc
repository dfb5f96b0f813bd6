// slam_frontend.cc -- host orchestration of one stereo frame, mirroring the reference's
// src/slam_frontend.cc:117-538 with the OpenCV calls replaced by the C ABI (include/vsf.h).
#include "slam_frontend.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace slam {

using slam_types::FeatureMatch;
using slam_types::OdometryFactor;
using slam_types::RobotPose;
using slam_types::SLAMNode;
using slam_types::SLAMProblem;
using slam_types::VisionFactor;
using slam_types::VisionFeature;

namespace {

Matrix3f Mul(const Matrix3f& a, const Matrix3f& b) {
  Matrix3f r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = (a(i, 0) * b(0, j) + a(i, 1) * b(1, j)) + a(i, 2) * b(2, j);
  return r;
}

Matrix3f Transpose(const Matrix3f& a) {
  Matrix3f r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = a(j, i);
  return r;
}

Matrix3f Inverse(const Matrix3f& a) {
  Matrix3f r;
  const float c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1), c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2),
              c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  const float det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02;
  const float id = 1.0f / det;
  r(0, 0) = c00 * id;
  r(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id;
  r(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
  r(1, 0) = c01 * id;
  r(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id;
  r(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
  r(2, 0) = c02 * id;
  r(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id;
  r(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
  return r;
}

Matrix3f CameraMatrix(const CameraIntrinsics& I) {  // cc:542-548
  Matrix3f M;
  const float v[9] = {I.fx, 0, I.cx, 0, I.fy, I.cy, 0, 0, 1};
  std::memcpy(M.m, v, sizeof(v));
  return M;
}

// Right singular vector of the smallest singular value of the M x 4 system whose columns are At[0..3] (M <= 6):
// one-sided (Hestenes) Jacobi in double as cv::SVD::compute runs it for cv::triangulatePoints (core/src/lapack.cpp
// JacobiSVDImpl_: rotate column pairs until every pair is orthogonal to 10 * DBL_EPSILON, at most 30 sweeps).
void SmallestRightSingularVector(double At[4][6], int M, double out[4]) {
  const double eps = 2.220446049250313e-16 * 10;
  double W[4], Vt[4][4];
  for (int i = 0; i < 4; i++) {
    W[i] = 0;
    for (int k = 0; k < M; k++) W[i] += At[i][k] * At[i][k];
    for (int k = 0; k < 4; k++) Vt[i][k] = i == k;
  }
  auto hyp = [](double a, double b) {
    a = std::fabs(a), b = std::fabs(b);
    if (a > b) return a * std::sqrt(1 + (b / a) * (b / a));
    return b > 0 ? b * std::sqrt(1 + (a / b) * (a / b)) : 0.0;
  };
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
    for (int i = 0; i < 3; i++)
      for (int j = i + 1; j < 4; j++) {
        double a = W[i], b = W[j], p = 0;
        for (int k = 0; k < M; k++) p += At[i][k] * At[j][k];
        if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = hyp(p, beta);
        double c, s;
        if (beta < 0) {
          s = std::sqrt((gamma - beta) * 0.5 / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < M; k++) {
          const double t0 = c * At[i][k] + s * At[j][k], t1 = -s * At[i][k] + c * At[j][k];
          At[i][k] = t0, At[j][k] = t1;
          a += t0 * t0, b += t1 * t1;
        }
        W[i] = a, W[j] = b;
        changed = true;
        for (int k = 0; k < 4; k++) {
          const double t0 = c * Vt[i][k] + s * Vt[j][k], t1 = -s * Vt[i][k] + c * Vt[j][k];
          Vt[i][k] = t0, Vt[j][k] = t1;
        }
      }
    if (!changed) break;
  }
  int best = 0;
  for (int i = 0; i < 4; i++) {
    W[i] = 0;
    for (int k = 0; k < M; k++) W[i] += At[i][k] * At[i][k];
    if (W[i] < W[best]) best = i;
  }
  for (int k = 0; k < 4; k++) out[k] = Vt[best][k];
}

}  // namespace

// ---- configuration: the reference's hard-coded defaults (cc:550-652) ----
FrontendConfig::FrontendConfig() {
  debug_images_ = false;  // reference: true (quirk Q10: retains every image forever); rendering is out of scope
  // reference: AKAZE (cc:553, quirk Q1); ORB is the north-star path and the only extractor built here
  descriptor_extract_type_ = DescriptorExtractorType::ORB;
  best_percent_ = 0.3f;
  nn_match_ratio_ = 0.6f;
  frame_life_ = 10;
  min_odom_rotation = (float)(10.0 / 180.0 * M_PI);
  min_odom_translation = 0.2f;
  min_vision_matches = 10;
  orb_nfeatures = 10000;  // cc:205
  residual_order = 0;     // Eigen 3.3's reduction order of a fixed-size-3 lazy product (cc:381-383)
  image_width = 0;        // 0: taken from the first observed image
  image_height = 0;

  intrinsics_left.fx = 527.873518f;
  intrinsics_left.cx = 482.823413f;
  intrinsics_left.fy = 527.276819f;
  intrinsics_left.cy = 298.033945f;
  intrinsics_left.k1 = -0.153137f;
  intrinsics_left.k2 = 0.075666f;
  intrinsics_left.p1 = -0.000227f;
  intrinsics_left.p2 = -0.000320f;
  intrinsics_left.k3 = 0;
  intrinsics_right.fx = 530.158021f;
  intrinsics_right.cx = 475.540633f;
  intrinsics_right.fy = 529.682234f;
  intrinsics_right.cy = 299.995465f;
  intrinsics_right.k1 = -0.156833f;
  intrinsics_right.k2 = 0.081841f;
  intrinsics_right.p1 = -0.000779f;
  intrinsics_right.p2 = -0.000356f;
  intrinsics_right.k3 = -0.000779f;

  const Matrix3f K_left = CameraMatrix(intrinsics_left), K_right = CameraMatrix(intrinsics_right);
  const float A_right[12] = {0.999593617649873f,  0.021411909431148f,  -0.018818333830411f, -0.131707087331978f,
                             -0.021140534893290f, 0.999671312094879f,  0.014503294761121f,  0.003232397463343f,
                             0.019122691705565f,  -0.014099571235136f, 0.999717722536176f,  -0.001146108483477f};
  const float A_left[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) {
      float l = 0, r = 0;
      for (int k = 0; k < 3; k++) {
        l += K_left(i, k) * A_left[4 * k + j];
        r += K_right(i, k) * A_right[4 * k + j];
      }
      projection_left[4 * i + j] = l;
      projection_right[4 * i + j] = r;
    }
  {  // cc:613-618: left_cam_to_robot = Eigen::Translation3f(XT) * RT
    Matrix3f RT;
    const float rt[9] = {0.009916590468f, -0.2835522866f, 0.9589055021f,  -0.9998698619f, -0.01501486552f,
                         0.005900269087f, 0.01272480238f, -0.9588392225f, -0.2836642819f};
    std::memcpy(RT.m, rt, sizeof(rt));
    left_cam_to_robot = Affine3f(RT, Vector3f(-0.01f, 0.06f, 0.5299999713897705f));
  }
  // Fundamental matrix (cc:635-644).  The reference builds the cross-product matrix from A[1], A[2], A[3] of a
  // 3-vector (out-of-range read, quirk Q2); the well-formed skew matrix of A is used here instead.
  Matrix3f rotation;
  float translation[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) rotation(i, j) = A_right[4 * i + j];
    translation[i] = A_right[4 * i + 3];
  }
  const Matrix3f KRt = Mul(K_left, Transpose(rotation));
  float A[3];
  for (int i = 0; i < 3; i++) A[i] = (KRt(i, 0) * translation[0] + KRt(i, 1) * translation[1]) + KRt(i, 2) * translation[2];
  Matrix3f C;
  const float c[9] = {0.0f, -A[2], A[1], A[2], 0.0f, -A[0], -A[1], A[0], 0.0f};
  std::memcpy(C.m, c, sizeof(c));
  fundamental = Mul(Mul(Mul(Transpose(Inverse(K_right)), rotation), Transpose(K_left)), C);
}

vsf_calibration MakeCalibration(const FrontendConfig& config) {
  vsf_calibration c;
  std::memset(&c, 0, sizeof(c));
  std::memcpy(c.projection_left, config.projection_left, sizeof(c.projection_left));
  std::memcpy(c.projection_right, config.projection_right, sizeof(c.projection_right));
  const Matrix3f K = CameraMatrix(config.intrinsics_left);  // camera_matrix_left, cc:575-578
  std::memcpy(c.camera_matrix_left, K.m, sizeof(c.camera_matrix_left));
  const CameraIntrinsics& I = config.intrinsics_left;       // distortion_coeffs_left, cc:623-628
  const float d[5] = {I.k1, I.k2, I.p1, I.p2, I.k3};
  std::memcpy(c.distortion_left, d, sizeof(d));
  std::memcpy(c.fundamental, config.fundamental.m, sizeof(c.fundamental));
  c.triangulate_rows = 6;  // OpenCV 3.2.0's cvTriangulatePoints (CMakeLists.txt:20 pins that version)
  return c;
}

// ---- Frame (cc:511-519) ----
Frame::Frame(const std::vector<vsf_keypoint>& keypoints, const std::vector<uint8_t>& descriptors, uint64_t frame_ID) {
  keypoints_ = keypoints;
  descriptors_ = descriptors;
  frame_ID_ = frame_ID;
  is_initial_ = std::vector<bool>(keypoints_.size(), true);
  initial_ids_ = std::vector<int64_t>(keypoints_.size(), -1);
}

// ---- Frontend ----
Frontend::Frontend(const std::string& config_path) : Frontend(config_path, FrontendConfig(), 0) {}

Frontend::Frontend(const std::string& /*config_path*/, const FrontendConfig& config, int device)
    : odom_initialized_(false),
      odom_timestamp_(0),
      config_(config),
      curr_frame_ID_(0),
      stereo_ambig_constraint_(10000),  // cc:353
      fused_(true),
      pipelined_(false),
      ctx_(nullptr),
      device_(device),
      last_status_(VSF_OK) {
  if (config_.descriptor_extract_type_ != FrontendConfig::DescriptorExtractorType::ORB &&
      config_.descriptor_extract_type_ != FrontendConfig::DescriptorExtractorType::FREAK)
    last_status_ = VSF_ERR_UNSUPPORTED;  // the reference would build AKAZE / BRISK / SURF / SIFT here (cc:193-232)
  if (config_.image_width > 0 && config_.image_height > 0) EnsureContext(config_.image_width, config_.image_height);
}

// Frames still in flight (pipelined mode) are dropped, not booked: nobody can read the problem any more.  vsf_destroy
// waits for every stream of the context -- the slots' included -- before it frees what their kernels write.
Frontend::~Frontend() {
  pending_count_ = 0;
  vsf_destroy(ctx_);
}

bool Frontend::EnsureContext(int width, int height) {
  if (ctx_) {
    vsf_params p;
    vsf_get_params(ctx_, &p);
    if (p.width == width && p.height == height && p.max_images >= 2 * batch_frames() && ctx_depth_ == queue_depth())
      return true;
    // the context is replaced (another image size, or pipelining switched on): the frames still in the queue belong to
    // the old one and are booked first, in order; whatever that returns, their tickets die with the context
    Flush();
    pending_count_ = 0;
    vsf_destroy(ctx_);
    ctx_ = nullptr;
  }
  vsf_params p;
  vsf_params_default(&p, width, height, 2 * batch_frames());  // extraction buffers for one batch of stereo frames
  p.nfeatures = config_.orb_nfeatures;
  p.residual_order = config_.residual_order;
  last_status_ = vsf_params_set_ratio(&p, config_.nn_match_ratio_);
  if (last_status_ != VSF_OK) return false;
  last_status_ = vsf_create(&p, device_, &ctx_);
  if (last_status_ != VSF_OK) return false;
  ctx_depth_ = queue_depth();
  last_status_ = vsf_set_option(ctx_, VSF_OPT_OBSERVE_THREAD, queue_thread_ ? 1 : 0);
  if (last_status_ == VSF_OK) last_status_ = vsf_set_option(ctx_, VSF_OPT_OBSERVE_COPY_THREAD, copy_thread_ ? 1 : 0);
  for (const auto& ov : ctx_options_)
    if (last_status_ == VSF_OK) last_status_ = vsf_set_option(ctx_, ov.first, ov.second);
  if (last_status_ == VSF_OK) last_status_ = vsf_observe_configure(ctx_, ctx_depth_, min_batch_, 0);
  pending_.assign((size_t)ctx_depth_, PendingFrame());
  pending_head_ = pending_count_ = 0;
  return last_status_ == VSF_OK;
}

// cc:250-263.  The reference copies the still-uninitialised odom_*_ into prev_odom_*_ on the first call (quirk
// Q11); here prev_* starts at the first observed pose.
void Frontend::ObserveOdometry(const Vector3f& translation, const Quaternionf& rotation, double timestamp) {
  if (!odom_initialized_) {
    init_odom_rotation_ = rotation;
    init_odom_translation_ = translation;
    prev_odom_rotation_ = rotation;
    prev_odom_translation_ = translation;
    odom_initialized_ = true;
  }
  odom_translation_ = translation;
  odom_rotation_ = rotation;
  odom_timestamp_ = timestamp;
}

// cc:175-186
bool Frontend::OdomCheck() {
  if (!odom_initialized_) return false;
  if ((prev_odom_translation_ - odom_translation_).norm() > config_.min_odom_translation) return true;
  if (prev_odom_rotation_.angularDistance(odom_rotation_) > config_.min_odom_rotation) return true;
  return false;
}

// cc:266-280: detectAndCompute (ORB) or FAST detect (+ FREAK compute, which is not built: keypoints only).
bool Frontend::ExtractFeatures(const Image& image, Frame* frame) {
  if (image.empty() || !EnsureContext(image.cols, image.rows)) {
    if (last_status_ == VSF_OK) last_status_ = VSF_ERR_INVALID_ARG;
    return false;
  }
  vsf_params p;
  vsf_get_params(ctx_, &p);
  int n = 0;
  std::vector<vsf_keypoint> kps;
  std::vector<uint8_t> desc;
  if (config_.descriptor_extract_type_ == FrontendConfig::DescriptorExtractorType::FREAK) {
    int cap = 1 << 16;
    kps.resize(cap);
    last_status_ = vsf_fast_detect(ctx_, image.data, image.cols, image.rows, image.step, -1, 1, kps.data(), cap, &n);
    if (last_status_ == VSF_ERR_CAPACITY) {
      cap = n;
      kps.resize(cap);
      last_status_ = vsf_fast_detect(ctx_, image.data, image.cols, image.rows, image.step, -1, 1, kps.data(), cap, &n);
    }
    if (last_status_ != VSF_OK) return false;
    kps.resize(n);
  } else {
    const int cap = p.max_keypoints;
    kps.resize(cap);
    desc.resize((size_t)cap * VSF_DESC_BYTES);
    last_status_ = vsf_extract(ctx_, image.data, image.cols, image.rows, image.step, kps.data(), desc.data(), cap, &n);
    if (last_status_ != VSF_OK) return false;
    kps.resize(n);
    desc.resize((size_t)n * VSF_DESC_BYTES);
  }
  *frame = Frame(kps, desc, curr_frame_ID_);
  return true;
}

bool Frontend::ExtractFeaturesPair(const Image& left, const Image& right, Frame* left_frame, Frame* right_frame) {
  const bool orb = config_.descriptor_extract_type_ != FrontendConfig::DescriptorExtractorType::FREAK;
  if (!orb || left.empty() || right.empty() || left.cols != right.cols || left.rows != right.rows ||
      left.step != right.step)
    return ExtractFeatures(left, left_frame) && ExtractFeatures(right, right_frame);
  if (!EnsureContext(left.cols, left.rows)) {
    if (last_status_ == VSF_OK) last_status_ = VSF_ERR_INVALID_ARG;
    return false;
  }
  vsf_params p;
  vsf_get_params(ctx_, &p);
  const int cap = p.max_keypoints;
  std::vector<vsf_keypoint> kl(cap), kr(cap);
  std::vector<uint8_t> dl((size_t)cap * VSF_DESC_BYTES), dr((size_t)cap * VSF_DESC_BYTES);
  int nl = 0, nr = 0;
  last_status_ = vsf_extract_pair(ctx_, left.data, right.data, left.cols, left.rows, left.step, kl.data(), dl.data(), &nl,
                                  kr.data(), dr.data(), &nr, cap);
  if (last_status_ != VSF_OK) return false;
  kl.resize(nl);
  dl.resize((size_t)nl * VSF_DESC_BYTES);
  kr.resize(nr);
  dr.resize((size_t)nr * VSF_DESC_BYTES);
  *left_frame = Frame(kl, dl, curr_frame_ID_);
  *right_frame = Frame(kr, dr, curr_frame_ID_);
  return true;
}

// cc:521-538
std::vector<vsf_dmatch> Frontend::GetMatches(const Frame& frame_query, const Frame& frame_train, double nn_match_ratio) {
  std::vector<vsf_dmatch> best_matches;
  if (nn_match_ratio != (double)config_.nn_match_ratio_ || !ctx_) {
    last_status_ = VSF_ERR_UNSUPPORTED;  // the context carries the configured ratio
    return best_matches;
  }
  const int nq = (int)frame_query.keypoints_.size(), nt = (int)frame_train.keypoints_.size();
  if (frame_query.descriptors_.size() != (size_t)nq * VSF_DESC_BYTES ||
      frame_train.descriptors_.size() != (size_t)nt * VSF_DESC_BYTES)
    return best_matches;  // no descriptors (FREAK branch)
  best_matches.resize(nq);
  int n = 0;
  last_status_ = vsf_get_matches(ctx_, frame_query.descriptors_.data(), nq, frame_train.descriptors_.data(), nt,
                                 best_matches.data(), nq, &n);
  best_matches.resize(last_status_ == VSF_OK ? n : 0);
  return best_matches;
}

// cc:282-309
VisionFactor Frontend::GetFeatureMatches(Frame* past_frame_ptr, Frame* curr_frame_ptr) {
  Frame& past_frame = *past_frame_ptr;
  Frame& curr_frame = *curr_frame_ptr;
  std::vector<FeatureMatch> pairs;
  std::vector<vsf_dmatch> matches = GetMatches(past_frame, curr_frame, config_.nn_match_ratio_);
  // cv::DMatch::operator< compares distance only; std::sort is unstable and its tie order decides who survives.
  std::sort(matches.begin(), matches.end(),
            [](const vsf_dmatch& a, const vsf_dmatch& b) { return a.distance < b.distance; });
  const int num_good_matches = (int)(matches.size() * config_.best_percent_);  // size_t * float -> float -> int
  matches.erase(matches.begin() + std::min<size_t>(std::max(num_good_matches, 0), matches.size()), matches.end());
  for (const vsf_dmatch& match : matches) {
    pairs.push_back(FeatureMatch(match.queryIdx, match.trainIdx));
    if (curr_frame.is_initial_[match.trainIdx]) {
      curr_frame.is_initial_[match.trainIdx] = false;
      curr_frame.initial_ids_[match.trainIdx] = past_frame.is_initial_[match.queryIdx]
                                                    ? (int64_t)past_frame.frame_ID_
                                                    : past_frame.initial_ids_[match.queryIdx];
    }
  }
  return VisionFactor(past_frame.frame_ID_, curr_frame.frame_ID_, pairs);
}

void Frontend::GetFeatureMatchesAll(std::vector<Frame>* past_frames, Frame* curr_frame_ptr,
                                    std::vector<VisionFactor>* out) {
  Frame& curr_frame = *curr_frame_ptr;
  const int S = (int)past_frames->size();
  if (S == 0) return;
  const int nt = (int)curr_frame.keypoints_.size();
  bool batched = ctx_ != nullptr && curr_frame.descriptors_.size() == (size_t)nt * VSF_DESC_BYTES;
  std::vector<const uint8_t*> q(S);
  std::vector<int> nq(S), nm(S, 0);
  int cap = 1;
  for (int s = 0; s < S && batched; s++) {
    const Frame& f = (*past_frames)[s];
    nq[s] = (int)f.keypoints_.size();
    if (f.descriptors_.size() != (size_t)nq[s] * VSF_DESC_BYTES) batched = false;
    q[s] = f.descriptors_.data();
    cap = std::max(cap, nq[s]);
  }
  if (!batched) {  // (FREAK branch: no descriptors) one call per past frame, as the reference
    for (Frame& past : *past_frames) out->push_back(GetFeatureMatches(&past, &curr_frame));
    return;
  }
  std::vector<vsf_dmatch> all((size_t)S * cap);
  last_status_ = vsf_get_matches_multi(ctx_, q.data(), nq.data(), S, curr_frame.descriptors_.data(), nt, all.data(), cap,
                                       nm.data());
  if (last_status_ != VSF_OK) std::fill(nm.begin(), nm.end(), 0);
  for (int s = 0; s < S; s++) {  // cc:289-308 for every past frame, in list order
    Frame& past_frame = (*past_frames)[s];
    std::vector<vsf_dmatch> matches(all.begin() + (size_t)s * cap, all.begin() + (size_t)s * cap + nm[s]);
    std::sort(matches.begin(), matches.end(),
              [](const vsf_dmatch& a, const vsf_dmatch& b) { return a.distance < b.distance; });
    const int num_good_matches = (int)(matches.size() * config_.best_percent_);
    matches.erase(matches.begin() + std::min<size_t>(std::max(num_good_matches, 0), matches.size()), matches.end());
    std::vector<FeatureMatch> pairs;
    for (const vsf_dmatch& match : matches) {
      pairs.push_back(FeatureMatch(match.queryIdx, match.trainIdx));
      if (curr_frame.is_initial_[match.trainIdx]) {
        curr_frame.is_initial_[match.trainIdx] = false;
        curr_frame.initial_ids_[match.trainIdx] = past_frame.is_initial_[match.queryIdx]
                                                      ? (int64_t)past_frame.frame_ID_
                                                      : past_frame.initial_ids_[match.queryIdx];
      }
    }
    out->push_back(VisionFactor(past_frame.frame_ID_, curr_frame.frame_ID_, pairs));
  }
}

// cc:311-321
void Frontend::AddOdometryFactor() {
  const Vector3f translation = prev_odom_rotation_.inverse() * (odom_translation_ - prev_odom_translation_);
  const Quaternionf rotation(odom_rotation_ * prev_odom_rotation_.inverse());
  odometry_factors_.push_back(OdometryFactor(curr_frame_ID_ - 1, curr_frame_ID_, translation, rotation));
}

// cc:353-398.  |l^T F r| per stereo match; survivors re-index both frames; the next frame's threshold is this frame's
// mean residual + 2.  The reference's `(left_ph.transpose() * F * right_ph).norm()` is two rounds of three-term dot
// products; Eigen 3.3 evaluates a fixed-size-3 lazy product coefficient as cwiseProduct().sum() with the unrolled
// reduction redux_novec_unroller<.., 0, 3>, which splits 1 + 2: a0 b0 + (a1 b1 + a2 b2) (config_.residual_order 0);
// .norm() of the 1 x 1 result is sqrt(x * x) == |x| barring under- / overflow of the square.
namespace {
inline float Dot3(int order, float a0, float b0, float a1, float b1, float a2, float b2) {
  const float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  return order ? (p0 + p1) + p2 : p0 + (p1 + p2);
}
}  // namespace

void Frontend::RemoveAmbigStereo(Frame* left, Frame* right, const std::vector<vsf_dmatch>& stereo_matches) {
  std::vector<vsf_keypoint> left_keypoints, right_keypoints;
  std::vector<uint8_t> left_descs, right_descs;
  const Matrix3f& F = config_.fundamental;
  float avg_constraint = 0.0f;
  for (size_t m = 0; m < stereo_matches.size(); m++) {
    const vsf_dmatch& match = stereo_matches[m];
    const vsf_keypoint& lk = left->keypoints_[match.queryIdx];
    const vsf_keypoint& rk = right->keypoints_[match.trainIdx];
    const float l[3] = {lk.x, lk.y, 1.0f}, r[3] = {rk.x, rk.y, 1.0f};
    float t[3];
    const int order = config_.residual_order;
    for (int j = 0; j < 3; j++) t[j] = Dot3(order, l[0], F(0, j), l[1], F(1, j), l[2], F(2, j));
    const float constraint = std::fabs(Dot3(order, t[0], r[0], t[1], r[1], t[2], r[2]));
    avg_constraint += constraint;
    if (constraint <= stereo_ambig_constraint_) {
      left_keypoints.push_back(lk);
      right_keypoints.push_back(rk);
      const uint8_t* ld = left->descriptors_.data() + (size_t)match.queryIdx * VSF_DESC_BYTES;
      const uint8_t* rd = right->descriptors_.data() + (size_t)match.trainIdx * VSF_DESC_BYTES;
      left_descs.insert(left_descs.end(), ld, ld + VSF_DESC_BYTES);
      right_descs.insert(right_descs.end(), rd, rd + VSF_DESC_BYTES);
    }
  }
  // cc:392-394.  Without a match this is 0/0 + 2 = NaN (quirk Q3): the next frame keeps no feature (`c <= NaN` is
  // false) but updates the threshold from its own matches, so the frame after it is filtered normally again.
  stereo_ambig_constraint_ = avg_constraint / (float)stereo_matches.size() + 2.0f;
  *left = Frame(left_keypoints, left_descs, left->frame_ID_);
  *right = Frame(right_keypoints, right_descs, right->frame_ID_);
}

// cc:117-173: right->left matches with best_percent forced to 1, then cv::triangulatePoints (per point the right
// singular vector of the smallest singular value of the 6 x 4 system OpenCV 3.2 builds: x*P2-P0, y*P2-P1, x*P1-y*P0 per
// view, in double) and the homogeneous divide in float.  Same arithmetic as the device kernel (csrc/k_points.hip);
// against real OpenCV the values agree to rounding (its SVD may run through LAPACK), not bit for bit.
void Frontend::Calculate3DPoints(Frame* left_frame, Frame* right_frame, std::vector<Vector3f>* points) {
  const float best_percent = config_.best_percent_;
  config_.best_percent_ = 1.0f;
  const VisionFactor matches = GetFeatureMatches(right_frame, left_frame);
  config_.best_percent_ = best_percent;
  if (matches.feature_matches.empty()) return;
  const float* P[2] = {config_.projection_left, config_.projection_right};
  for (const FeatureMatch& match : matches.feature_matches) {
    const vsf_keypoint& left_pt = left_frame->keypoints_[match.feature_idx_current];
    const vsf_keypoint& right_pt = right_frame->keypoints_[match.feature_idx_initial];
    const double xy[2][2] = {{left_pt.x, left_pt.y}, {right_pt.x, right_pt.y}};
    double At[4][6];
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 4; k++) {
        const double p0 = P[j][k], p1 = P[j][4 + k], p2 = P[j][8 + k];
        At[k][3 * j + 0] = xy[j][0] * p2 - p0;
        At[k][3 * j + 1] = xy[j][1] * p2 - p1;
        At[k][3 * j + 2] = xy[j][0] * p1 - xy[j][1] * p0;
      }
    double X[4];
    SmallestRightSingularVector(At, 6, X);
    const float xf = (float)X[0], yf = (float)X[1], zf = (float)X[2], wf = (float)X[3];
    points->push_back(Vector3f(xf, yf, zf) / wf);
  }
}

// cc:323-351: cv::undistortPoints(distorted, out, K_left, dist_left, noArray(), K_left): 5 fixed-point
// iterations in double per point, then re-projection with the same camera matrix (row f2 of SURVEY section 8(f)).
void Frontend::UndistortFeaturePoints(std::vector<VisionFeature>* features_ptr) {
  std::vector<VisionFeature>& features = *features_ptr;
  const CameraIntrinsics& I = config_.intrinsics_left;
  const double k[5] = {I.k1, I.k2, I.p1, I.p2, I.k3};
  const double fx = I.fx, fy = I.fy, cx = I.cx, cy = I.cy, ifx = 1. / fx, ify = 1. / fy;
  for (VisionFeature& f : features) {
    double x = (f.pixel.x() - cx) * ifx, y = (f.pixel.y() - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    f.pixel = Vector2f((float)(x * fx + cx), (float)(y * fy + cy));
  }
}

// cc:400-472 through ONE submission to the GPU (vsf_observe_stereo): upload, both ExtractFeatures, GetMatches,
// RemoveAmbigStereo, every GetFeatureMatches of the temporal loop, Calculate3DPoints, the VisionFeature records and
// UndistortFeaturePoints run on the device back to back and come home in one compact buffer; the kept frames'
// descriptors never leave HBM.  What stays here is the reference's bookkeeping (is_initial_ / initial_ids_, nodes,
// factors, the sliding window), in the reference's order.
bool Frontend::ObserveImageFused(const Image& left_image, const Image& right_image) {
  if (!EnsureContext(left_image.cols, left_image.rows)) {
    if (last_status_ == VSF_OK) last_status_ = VSF_ERR_INVALID_ARG;
    return false;
  }
  const vsf_calibration calib = MakeCalibration(config_);
  // the queue must have room: when it is full, the oldest frame is collected and booked first
  while ((int)pending_count_ >= queue_depth())
    if (!RetireOldest()) return false;
  PendingFrame& pf = pending_[(pending_head_ + pending_count_) % pending_.size()];
  last_status_ = vsf_observe_submit(ctx_, left_image.data, right_image.data, left_image.cols, left_image.rows,
                                    left_image.step, &calib, config_.best_percent_, (int)config_.frame_life_, &pf.ticket);
  if (last_status_ != VSF_OK) return false;
  pf.odom_translation = odom_translation_;
  pf.odom_rotation = odom_rotation_;
  pf.prev_odom_translation = prev_odom_translation_;
  pf.prev_odom_rotation = prev_odom_rotation_;
  pf.odom_timestamp = odom_timestamp_;
  pending_count_++;
  // cc:457-458: the pose of this frame is what the NEXT call's OdomCheck compares with
  prev_odom_rotation_ = odom_rotation_;
  prev_odom_translation_ = odom_translation_;
  if (!pipelined_) return RetireOldest();
  // results that are already there are booked now (no waiting, nothing sent early): at a camera's rate the problem stays a
  // frame or two behind instead of a queue's depth; when frames stream in faster than the GPU serves them this finds nothing
  // and the queue fills as before
  while (pending_count_ > 1) {
    int ready = 0;
    if (vsf_observe_poll(ctx_, pending_[pending_head_].ticket, &ready) != VSF_OK || !ready) break;
    if (!RetireOldest()) return false;
  }
  return true;
}

bool Frontend::Flush() {
  while (pending_count_ > 0)
    if (!RetireOldest()) return false;
  return true;
}

// The second half of ObserveImageFused for the oldest frame in flight: wait for its result, decode it, and do the
// reference's bookkeeping (cc:424-470) with the odometry that frame's call saw.
bool Frontend::RetireOldest() {
  const PendingFrame pf = pending_[pending_head_];
  pending_head_ = (pending_head_ + 1) % pending_.size();
  pending_count_--;
  size_t bytes = 0;
  const uint8_t* b = nullptr;  // the result, read where the GPU wrote it (the context's pinned result ring)
  last_status_ = vsf_observe_collect_view(ctx_, pf.ticket, &b, &bytes);
  if (last_status_ != VSF_OK) return false;
  uint32_t hdr[16];
  std::memcpy(hdr, b, sizeof(hdr));
  const int n_pairs = (int)hdr[1], nfeat = (int)hdr[2];
  std::memcpy(&stereo_ambig_constraint_, &hdr[10], sizeof(float));  // the static's value after this frame (cc:392-394)
  if (n_pairs != (int)frame_list_.size() + 1) {  // the context's window and frame_list_ went out of step
    last_status_ = VSF_ERR_INVALID_ARG;
    return false;
  }
  std::vector<uint32_t> npairs((size_t)n_pairs);
  std::memcpy(npairs.data(), b + 64, (size_t)n_pairs * 4);
  size_t off = 64 + 4 * (size_t)((n_pairs + 3) & ~3);
  const uint8_t* feat_bytes = b + off;
  off += (size_t)nfeat * sizeof(vsf_vision_feature);
  std::vector<const uint8_t*> pair_bytes((size_t)n_pairs);
  for (int p = 0; p < n_pairs; p++) {
    pair_bytes[p] = b + off;
    off += (size_t)npairs[p] * sizeof(vsf_feature_match);
  }
  Frame curr_frame;  // Frame(kps, desc, curr_frame_ID_), built in place (cc:511-519)
  curr_frame.frame_ID_ = curr_frame_ID_;
  curr_frame.keypoints_.resize((size_t)nfeat);
  curr_frame.descriptors_.resize((size_t)nfeat * VSF_DESC_BYTES);
  if (nfeat > 0) {
    std::memcpy(curr_frame.keypoints_.data(), b + off, (size_t)nfeat * sizeof(vsf_keypoint));
    std::memcpy(curr_frame.descriptors_.data(), b + off + (size_t)nfeat * sizeof(vsf_keypoint), curr_frame.descriptors_.size());
  }
  curr_frame.is_initial_.assign((size_t)nfeat, true);
  curr_frame.initial_ids_.assign((size_t)nfeat, -1);
  auto book = [](const Frame& past_frame, Frame* curr, const uint8_t* bytes, uint32_t n, std::vector<FeatureMatch>* pairs) {
    if (pairs) pairs->reserve(n);
    for (uint32_t k = 0; k < n; k++) {  // cc:294-306
      vsf_feature_match m;
      std::memcpy(&m, bytes + (size_t)k * sizeof(m), sizeof(m));
      if (pairs) pairs->push_back(FeatureMatch(m.feature_idx_initial, m.feature_idx_current));
      if (m.feature_idx_current < curr->is_initial_.size() && curr->is_initial_[m.feature_idx_current]) {
        curr->is_initial_[m.feature_idx_current] = false;
        curr->initial_ids_[m.feature_idx_current] =
            m.feature_idx_initial < past_frame.is_initial_.size() && !past_frame.is_initial_[m.feature_idx_initial]
                ? past_frame.initial_ids_[m.feature_idx_initial]
                : (int64_t)past_frame.frame_ID_;
      }
    }
  };
  for (int p = 0; p + 1 < n_pairs; p++) {  // the temporal loop, cc:424-434
    std::vector<FeatureMatch> pairs;
    book(frame_list_[p], &curr_frame, pair_bytes[p], npairs[p], &pairs);
    vision_factors_.emplace_back(frame_list_[p].frame_ID_, curr_frame.frame_ID_, std::move(pairs));
  }
  {  // Calculate3DPoints' GetFeatureMatches(right, left) (cc:131): every row of the right frame is still `initial`
    Frame right_temp_frame;
    right_temp_frame.frame_ID_ = curr_frame_ID_;
    book(right_temp_frame, &curr_frame, pair_bytes[n_pairs - 1], npairs[n_pairs - 1], nullptr);
  }
  std::vector<VisionFeature> features;
  features.reserve((size_t)nfeat);
  for (int i = 0; i < nfeat; i++) {  // cc:438-443, computed on the device
    vsf_vision_feature f;
    std::memcpy(&f, feat_bytes + (size_t)i * sizeof(f), sizeof(f));
    features.push_back(VisionFeature((uint64_t)f.feature_idx_lo | ((uint64_t)f.feature_idx_hi << 32),
                                     Vector2f(f.pixel[0], f.pixel[1]), Vector3f(f.point3d[0], f.point3d[1], f.point3d[2])));
  }
  // cc:444-470 with the odometry of this frame's call
  const Vector3f loc = init_odom_rotation_.inverse() * (pf.odom_translation - init_odom_translation_);
  const Quaternionf angle = pf.odom_rotation * init_odom_rotation_.inverse();
  nodes_.emplace_back(curr_frame_ID_, pf.odom_timestamp, RobotPose(loc, angle), std::move(features));
  if (curr_frame_ID_ > 0) {  // AddOdometryFactor, cc:311-321
    const Vector3f translation = pf.prev_odom_rotation.inverse() * (pf.odom_translation - pf.prev_odom_translation);
    const Quaternionf rotation(pf.odom_rotation * pf.prev_odom_rotation.inverse());
    odometry_factors_.push_back(OdometryFactor(curr_frame_ID_ - 1, curr_frame_ID_, translation, rotation));
  }
  curr_frame_ID_++;
  if (frame_list_.size() >= config_.frame_life_ && !frame_list_.empty()) frame_list_.erase(frame_list_.begin());
  frame_list_.push_back(std::move(curr_frame));
  return true;
}

// cc:444-470: node, odometry factor, sliding window.
void Frontend::FinishNode(const Frame& curr_frame, const std::vector<VisionFeature>& features) {
  const Vector3f loc = init_odom_rotation_.inverse() * (odom_translation_ - init_odom_translation_);
  const Quaternionf angle = odom_rotation_ * init_odom_rotation_.inverse();
  nodes_.push_back(SLAMNode(curr_frame_ID_, odom_timestamp_, RobotPose(loc, angle), features));
  if (curr_frame_ID_ > 0) AddOdometryFactor();
  prev_odom_rotation_ = odom_rotation_;
  prev_odom_translation_ = odom_translation_;
  curr_frame_ID_++;
  if (frame_list_.size() >= config_.frame_life_ && !frame_list_.empty()) frame_list_.erase(frame_list_.begin());
  frame_list_.push_back(curr_frame);
}

// cc:400-472
bool Frontend::ObserveImage(const Image& left_image, const Image& right_image, double /*time*/) {
  if (!OdomCheck()) return false;
  if (fused_ && config_.descriptor_extract_type_ == FrontendConfig::DescriptorExtractorType::ORB && !left_image.empty() &&
      !right_image.empty() && left_image.cols == right_image.cols && left_image.rows == right_image.rows &&
      left_image.step == right_image.step && config_.orb_nfeatures + 256 < 65536 && config_.frame_life_ >= 1 &&
      config_.frame_life_ + 1 <= 64)
    return ObserveImageFused(left_image, right_image);
  Frame curr_frame, right_temp_frame;
  if (!ExtractFeaturesPair(left_image, right_image, &curr_frame, &right_temp_frame)) return false;
  const std::vector<vsf_dmatch> stereo_matches = GetMatches(curr_frame, right_temp_frame, config_.nn_match_ratio_);
  RemoveAmbigStereo(&curr_frame, &right_temp_frame, stereo_matches);
  GetFeatureMatchesAll(&frame_list_, &curr_frame, &vision_factors_);
  std::vector<Vector3f> points;
  Calculate3DPoints(&curr_frame, &right_temp_frame, &points);
  std::vector<VisionFeature> features;
  for (uint64_t i = 0; i < curr_frame.keypoints_.size(); i++) {
    // The reference indexes points[i] by keypoint although `points` is in sorted-match order and may be shorter
    // (quirk Q5, out-of-range read); the same index is used where it exists, a zero point otherwise.
    const Vector3f p3 = i < points.size() ? points[i] : Vector3f();
    features.push_back(VisionFeature(i, Vector2f(curr_frame.keypoints_[i].x, curr_frame.keypoints_[i].y), p3));
  }
  UndistortFeaturePoints(&features);
  FinishNode(curr_frame, features);
  return true;
}

void Frontend::GetSLAMProblem(SLAMProblem* problem) const {
  Sync();  // (pipelined mode: frames still on the GPU are booked first)
  *problem = SLAMProblem(nodes_, vision_factors_, odometry_factors_);
}

int Frontend::GetNumPoses() {
  Flush();
  return (int)nodes_.size();
}

}  // namespace slam
