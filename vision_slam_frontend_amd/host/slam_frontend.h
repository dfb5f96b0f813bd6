// slam_frontend.h -- host-side mirror of the reference's slam::Frontend (src/slam_frontend.h:117-142) on top of
// the C ABI of include/vsf.h.  Same public method names and semantics:
//   ObserveImage (cc:400-472), ObserveOdometry (cc:250-263), GetSLAMProblem (cc:498-503), GetNumPoses (cc:505),
//   GetConfig (h:142), debug-image getters (cc:474-495, return empty: debug rendering is out of scope).
// cv::Mat is replaced by slam::Image (a non-owning view) and Eigen types by the PODs of slam_types.h; both swaps
// are mechanical for a maintainer who has OpenCV / Eigen.  The two private methods that call
// OpenCV in the reference -- ExtractFeatures (cc:266) and GetMatches (cc:521) -- call vsf_extract /
// vsf_get_matches here; everything else is the reference's own host logic restated.
#ifndef VSF_HOST_SLAM_FRONTEND_H_
#define VSF_HOST_SLAM_FRONTEND_H_

#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vsf.h"
#include "slam_types.h"

namespace slam {

using slam_types::Quaternionf;
using slam_types::Vector2f;
using slam_types::Vector3f;

// Non-owning 8-bit single-channel image view (stands in for `const cv::Mat&`).
struct Image {
  const uint8_t* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;
  Image() {}
  Image(const uint8_t* d, int r, int c, size_t s) : data(d), rows(r), cols(c), step(s) {}
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
};

// src/slam_frontend.h:42-56
struct CameraIntrinsics {
  float k1, k2, k3;
  float p1, p2;
  float fx, fy, cx, cy;
};

struct Matrix3f {
  float m[9];  // row-major
  float operator()(int r, int c) const { return m[3 * r + c]; }
  float& operator()(int r, int c) { return m[3 * r + c]; }
};

// Eigen::Affine3f as the reference uses it (h:96, cc:613-618, slam_frontend_main.cc:341-344): a rotation and a
// translation, `Translation3f(XT) * RT`.
struct Affine3f {
  Matrix3f linear_;  // row-major
  Vector3f translation_;
  Affine3f() : linear_{{1, 0, 0, 0, 1, 0, 0, 0, 1}} {}
  Affine3f(const Matrix3f& rotation, const Vector3f& translation) : linear_(rotation), translation_(translation) {}
  Vector3f translation() const { return translation_; }
  // (Eigen's Transform::rotation() of an affine transform takes the closest rotation of the linear part by SVD; the linear
  // part IS a rotation here -- cc:614-617 -- so it is returned as it is: equal to float rounding)
  Matrix3f rotation() const { return linear_; }
  Matrix3f linear() const { return linear_; }
  Vector3f operator*(const Vector3f& p) const {
    return Vector3f((linear_(0, 0) * p.x() + linear_(0, 1) * p.y()) + linear_(0, 2) * p.z() + translation_.x(),
                    (linear_(1, 0) * p.x() + linear_(1, 1) * p.y()) + linear_(1, 2) * p.z() + translation_.y(),
                    (linear_(2, 0) * p.x() + linear_(2, 1) * p.y()) + linear_(2, 2) * p.z() + translation_.z());
  }
};

// src/slam_frontend.h:58-97; defaults are the reference's (cc:550-652) except where noted in the .cc.
struct FrontendConfig {
  enum class DescriptorExtractorType { AKAZE, ORB, BRISK, SURF, SIFT, FREAK };
  FrontendConfig();
  bool debug_images_;
  DescriptorExtractorType descriptor_extract_type_;
  float best_percent_;
  float nn_match_ratio_;
  float min_odom_translation;
  float min_odom_rotation;
  uint32_t min_vision_matches;
  uint32_t frame_life_;
  CameraIntrinsics intrinsics_left, intrinsics_right;
  float projection_left[12], projection_right[12];  // 3x4 row-major (cv::Mat CV_32F in the reference)
  Matrix3f fundamental;
  // Affine transform from the frame of the left camera to the robot (h:96; literals cc:613-618; the reference's caller
  // reads it through GetConfig() for the CameraExtrinsics message and the point cloud, slam_frontend_main.cc:158, 342).
  Affine3f left_cam_to_robot;
  // ORB parameters the reference hard-codes in cv::ORB::create (cc:205-213); exposed so BASELINE configs can set
  // nfeatures = 2000 / 8000.
  int orb_nfeatures;
  // How RemoveAmbigStereo's three-term dot products are summed (vsf_params::residual_order): 0 = as Eigen 3.3 does,
  // a0 b0 + (a1 b1 + a2 b2) (default); 1 = left to right.
  int residual_order;
  // Image geometry the GPU context is created for (the reference takes it from the first cv::Mat).
  int image_width, image_height;
};

// FrontendConfig's stereo calibration in the layout of the C ABI (include/vsf.h vsf_calibration).
vsf_calibration MakeCalibration(const FrontendConfig& config);

// src/slam_frontend.h:100-114
class Frame {
 public:
  Frame(const std::vector<vsf_keypoint>& keypoints, const std::vector<uint8_t>& descriptors, uint64_t frame_ID);
  Frame() : frame_ID_(0) {}
  uint64_t frame_ID_;
  std::vector<vsf_keypoint> keypoints_;
  std::vector<bool> is_initial_;
  std::vector<int64_t> initial_ids_;
  std::vector<uint8_t> descriptors_;  // keypoints_.size() x 32, row-major (cv::Mat CV_8U in the reference)
};

class Frontend {
 public:
  // config_path is ignored exactly as in the reference (quirk Q1: FrontendConfig::Load is never defined).
  explicit Frontend(const std::string& config_path);
  Frontend(const std::string& config_path, const FrontendConfig& config, int device = 0);
  ~Frontend();
  Frontend(const Frontend&) = delete;
  Frontend& operator=(const Frontend&) = delete;

  // True iff a new SLAM node was added.  Never throws; a failing GPU call is reported by last_status().
  bool ObserveImage(const Image& left_image, const Image& right_image, double time);
  void ObserveOdometry(const Vector3f& translation, const Quaternionf& rotation, double timestamp);
  void GetSLAMProblem(slam_types::SLAMProblem* problem) const;
  int GetNumPoses();
  FrontendConfig GetConfig() { return config_; }
  std::vector<Image> getDebugImages() { return {}; }
  Image GetLastDebugImage() { return Image(); }
  Image GetLastDebugStereoImage() { return Image(); }
  std::vector<Image> getDebugStereoImages() { return {}; }

  // Additions (not in the reference): error reporting instead of abort, and read access for tests.
  vsf_status last_status() const { return last_status_; }
  // true (default): ObserveImage is one GPU submission (vsf_observe_stereo); false: one C-ABI call per reference call
  // (vsf_extract_pair, vsf_get_matches, ...) with the reference's host steps in between.  Same results; choose before
  // the first ObserveImage.
  void set_fused(bool on) { fused_ = on; }
  // ObserveImage's return value is OdomCheck's decision (cc:404-409): nothing in the reference's control flow needs a
  // frame's features before the next frame arrives.  With pipelining on (fused mode; choose before the first
  // ObserveImage) a call copies its frame into the GPU context's queue (vsf_observe_submit) and returns; frames that wait
  // there leave for the GPU as ONE batched extraction + tail, and a frame's result is collected and booked -- in frame
  // order, with the odometry of ITS call -- as soon as a later call finds it finished, at the latest when the queue is full
  // (queue_depth() frames later) or when anything reads the problem (GetSLAMProblem, GetNumPoses, the accessors below,
  // Flush).  Same nodes, factors and bytes as the
  // synchronous mode; a GPU failure then surfaces in last_status() some calls late.
  void set_pipelined(bool on) { pipelined_ = on; }
  // Frames ObserveImage may leave in the queue when pipelined (1..1024, default 256) and the most frames one batch carries
  // (default 128; the context's extraction buffers are sized for it: ~25 MB of HBM per 640x480 frame; the queue's staging
  // and result rings are pinned host memory: depth x (two images + vsf_observe_capacity)).  Measured on an MI355X at
  // 640x480 / 2000 features: depth 32 19 k frames/s, 64 25 k, 128 (64 per batch) 28 k, 256 (128 per batch) 32 k.
  void set_queue_depth(int n) { depth_ = n < 1 ? 1 : (n > 1024 ? 1024 : n); }
  void set_frames_in_flight(int n) { set_queue_depth(n); }  // (the name of rounds 3-5)
  void set_batch_frames(int n) { batch_frames_ = n < 1 ? 1 : (n > 256 ? 256 : n); }
  // While the GPU is busy, fewer waiting frames than this stay in the queue (0: a whole batch, or half the queue's depth
  // when that is less; 1: whatever waits leaves as soon as fewer than two batches are on the GPU).
  void set_min_batch(int n) { min_batch_ = n < 0 ? 0 : n; }
  // The queue's host threads: the staging-copy helper (VSF_OPT_OBSERVE_COPY_THREAD, on by default) and the launcher
  // (VSF_OPT_OBSERVE_THREAD, off by default: with frames gathering into batches it only pays on a host whose launches
  // are what bounds the caller, and costs where depth = batch).
  void set_queue_thread(bool on) { queue_thread_ = on; }
  void set_copy_thread(bool on) { copy_thread_ = on; }
  // Any vsf_option of the context (applied when it is created): launch choices only, results never depend on them.
  void set_context_option(int option, int value) { ctx_options_.push_back({option, value}); }
  // The GPU context behind the object (nullptr before the first image / without image_width): for tools that read its
  // per-stage timers or queue statistics; whoever calls an entry point on it shares the object's single-caller rule.
  vsf_ctx* context() const { return ctx_; }
  // vsf_observe_stats of the context (frames, batches, largest batch, ...): how the queue coalesced.
  void queue_stats(int64_t out[11]) const { for (int i = 0; i < 11; i++) out[i] = 0; if (ctx_) vsf_observe_stats(ctx_, out, 11); }
  int queue_depth() const { return pipelined_ ? depth_ : 1; }
  int frames_in_flight() const { return queue_depth(); }
  int batch_frames() const { return pipelined_ ? (depth_ < batch_frames_ ? depth_ : batch_frames_) : 1; }
  bool Flush();  // collects and books every frame still in flight; false (and last_status()) if one of them failed
  float stereo_ambig_constraint() const { Sync(); return stereo_ambig_constraint_; }
  const std::vector<Frame>& frame_list() const { Sync(); return frame_list_; }
  const std::vector<slam_types::SLAMNode>& nodes() const { Sync(); return nodes_; }
  const std::vector<slam_types::VisionFactor>& vision_factors() const { Sync(); return vision_factors_; }
  const std::vector<slam_types::OdometryFactor>& odometry_factors() const { Sync(); return odometry_factors_; }

 private:
  bool OdomCheck();
  bool ExtractFeatures(const Image& image, Frame* curr_frame);
  // The two ExtractFeatures calls of ObserveImage (cc:411-412) as one batch of two images (vsf_extract_pair).
  bool ExtractFeaturesPair(const Image& left, const Image& right, Frame* left_frame, Frame* right_frame);
  // The temporal loop of ObserveImage (cc:424-434): GetFeatureMatches of every past frame against the new one, with
  // the matcher run once for all of them (vsf_get_matches_multi); same factors, same order, same bookkeeping.
  void GetFeatureMatchesAll(std::vector<Frame>* past_frames, Frame* curr_frame,
                            std::vector<slam_types::VisionFactor>* out);
  slam_types::VisionFactor GetFeatureMatches(Frame* past_frame_ptr, Frame* curr_frame_ptr);
  std::vector<vsf_dmatch> GetMatches(const Frame& frame_query, const Frame& frame_train, double nn_match_ratio);
  void RemoveAmbigStereo(Frame* left, Frame* right, const std::vector<vsf_dmatch>& stereo_matches);
  void AddOdometryFactor();
  void UndistortFeaturePoints(std::vector<slam_types::VisionFeature>* features);
  void Calculate3DPoints(Frame* left_frame, Frame* right_frame, std::vector<Vector3f>* points);
  bool EnsureContext(int width, int height);
  bool ObserveImageFused(const Image& left_image, const Image& right_image);
  void FinishNode(const Frame& curr_frame, const std::vector<slam_types::VisionFeature>& features);
  // A frame the GPU is still working on, with the odometry its ObserveImage call saw (cc:444-458 reads it at the END of
  // the call; between submit and collect the driver may already have delivered the next pose).
  struct PendingFrame {
    int64_t ticket;
    Vector3f odom_translation, prev_odom_translation;
    Quaternionf odom_rotation, prev_odom_rotation;
    double odom_timestamp;
  };
  bool RetireOldest();
  void Sync() const { const_cast<Frontend*>(this)->Flush(); }

  bool odom_initialized_;
  Vector3f init_odom_translation_;
  Quaternionf init_odom_rotation_;
  Vector3f prev_odom_translation_;
  Quaternionf prev_odom_rotation_;
  Vector3f odom_translation_;
  Quaternionf odom_rotation_;
  double odom_timestamp_;
  FrontendConfig config_;
  uint64_t curr_frame_ID_;
  std::vector<Frame> frame_list_;
  std::vector<slam_types::VisionFactor> vision_factors_;
  std::vector<slam_types::SLAMNode> nodes_;
  std::vector<slam_types::OdometryFactor> odometry_factors_;
  // The reference keeps this in a file-static shared by all instances (cc:353, quirk Q3); here it is per object.
  float stereo_ambig_constraint_;
  bool fused_;
  bool pipelined_;
  int depth_ = 256, batch_frames_ = 128, min_batch_ = 0;
  bool queue_thread_ = false, copy_thread_ = true;
  std::vector<std::pair<int, int>> ctx_options_;
  std::vector<PendingFrame> pending_;  // a ring: pending_head_ is the oldest, pending_count_ frames wait
  size_t pending_head_ = 0, pending_count_ = 0;
  int ctx_depth_ = 0;
  vsf_ctx* ctx_;
  int device_;
  vsf_status last_status_;
};

}  // namespace slam

#endif  // VSF_HOST_SLAM_FRONTEND_H_
