// frontend_capi.cc -- flat C view of slam::Frontend so tests (ctypes) and foreign callers can drive the host
// class: same call sequence as the reference's driver (slam_frontend_main.cc:132,147,321).
#include <chrono>
#include <cstring>

#include "slam_frontend.h"
#include "slam_to_ros.h"

using slam::Frontend;
using slam::FrontendConfig;

extern "C" {

void* vsfh_frontend_create(int nfeatures, int width, int height, int device, const float* fundamental9,
                           float best_percent, int frame_life) {
  FrontendConfig cfg;
  cfg.orb_nfeatures = nfeatures;
  cfg.image_width = width;
  cfg.image_height = height;
  if (fundamental9) std::memcpy(cfg.fundamental.m, fundamental9, 9 * sizeof(float));
  if (best_percent > 0) cfg.best_percent_ = best_percent;
  if (frame_life > 0) cfg.frame_life_ = (uint32_t)frame_life;
  return new Frontend("", cfg, device);
}

// The stereo calibration of a default-constructed FrontendConfig (the reference's hard-coded constants,
// slam_frontend.cc:565-644) in the layout the C ABI takes -- the one place tests, bench.py and the multi-GPU path get it.
void vsfh_default_calibration(vsf_calibration* out) { *out = slam::MakeCalibration(FrontendConfig()); }

void vsfh_set_fused(void* f, int on) { static_cast<Frontend*>(f)->set_fused(on != 0); }
void vsfh_set_pipelined(void* f, int on) { static_cast<Frontend*>(f)->set_pipelined(on != 0); }
void vsfh_set_frames_in_flight(void* f, int n) { static_cast<Frontend*>(f)->set_queue_depth(n); }
void vsfh_set_queue(void* f, int depth, int batch_frames, int min_batch) {
  Frontend* fe = static_cast<Frontend*>(f);
  if (depth > 0) fe->set_queue_depth(depth);
  if (batch_frames > 0) fe->set_batch_frames(batch_frames);
  fe->set_min_batch(min_batch);
}

void vsfh_set_queue_threads(void* f, int launcher, int copy) {
  static_cast<Frontend*>(f)->set_queue_thread(launcher != 0);
  static_cast<Frontend*>(f)->set_copy_thread(copy != 0);
}

// The reference's driver loop (slam_frontend_main.cc:271-328) for n_frames stereo frames taken in turn from `frames`
// (n_src x 2 x h x w bytes): ObserveOdometry (a pose 0.3 m further on: OdomCheck accepts every frame) + ObserveImage per
// frame, no Python between the calls.  The clock starts at frame `warm` (after a Flush) and stops behind the final Flush.
// read_every > 0: GetSLAMProblem after every read_every-th new node, as the reference's driver does after every one
// (main.cc:320-321) -- that read waits for every frame still in the queue.
// Returns the steady frames per second; *mean_call_ms / *max_call_ms: time inside ObserveImage; < 0 on failure.
double vsfh_time_sequence(void* f, const uint8_t* frames, int n_src, int w, int h, int n_frames, int warm, int read_every,
                          double* mean_call_ms, double* max_call_ms) {
  using Clock = std::chrono::steady_clock;
  Frontend* fe = static_cast<Frontend*>(f);
  const slam::Quaternionf q(1, 0, 0, 0);
  const int first = fe->GetNumPoses();
  if (first == 0) fe->ObserveOdometry(slam::Vector3f(0, 0, 0), q, 0.0);
  Clock::time_point t0 = Clock::now();
  double sum = 0, worst = 0;
  for (int k = 0; k < n_frames; k++) {
    if (k == warm) {
      fe->Flush();
      t0 = Clock::now();
    }
    const uint8_t* l = frames + (size_t)(k % n_src) * 2 * w * h;
    fe->ObserveOdometry(slam::Vector3f(0.3f * (first + k + 1), 0, 0), q, 1.0 + first + k);
    const Clock::time_point a = Clock::now();
    const bool added = fe->ObserveImage(slam::Image(l, h, w, (size_t)w), slam::Image(l + (size_t)w * h, h, w, (size_t)w),
                                        1.0 + first + k);
    const double dt = std::chrono::duration<double>(Clock::now() - a).count();
    if (!added || fe->last_status() != VSF_OK) return -1.0;
    if (read_every > 0 && (k + 1) % read_every == 0) {
      slam_types::SLAMProblem problem;
      fe->GetSLAMProblem(&problem);
      if (problem.nodes.empty()) return -1.0;
    }
    if (k >= warm) {
      sum += dt;
      if (dt > worst) worst = dt;
    }
  }
  if (!fe->Flush()) return -1.0;
  const double wall = std::chrono::duration<double>(Clock::now() - t0).count();
  const int n = n_frames - warm;
  if (mean_call_ms) *mean_call_ms = n > 0 ? 1e3 * sum / n : 0;
  if (max_call_ms) *max_call_ms = 1e3 * worst;
  return n > 0 && wall > 0 ? n / wall : 0.0;
}
int vsfh_flush(void* f) { return static_cast<Frontend*>(f)->Flush() ? 1 : 0; }

void vsfh_frontend_destroy(void* f) { delete static_cast<Frontend*>(f); }

void vsfh_observe_odometry(void* f, const float t[3], const float q_wxyz[4], double ts) {
  static_cast<Frontend*>(f)->ObserveOdometry(slam::Vector3f(t[0], t[1], t[2]),
                                             slam::Quaternionf(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]), ts);
}

int vsfh_observe_image(void* f, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride, double time) {
  return static_cast<Frontend*>(f)->ObserveImage(slam::Image(left, h, w, stride), slam::Image(right, h, w, stride), time)
             ? 1
             : 0;
}

int vsfh_last_status(void* f) { return (int)static_cast<Frontend*>(f)->last_status(); }
int vsfh_num_poses(void* f) { return static_cast<Frontend*>(f)->GetNumPoses(); }
float vsfh_stereo_ambig_constraint(void* f) { return static_cast<Frontend*>(f)->stereo_ambig_constraint(); }

void vsfh_get_fundamental(void* f, float out9[9]) {
  const FrontendConfig c = static_cast<Frontend*>(f)->GetConfig();
  std::memcpy(out9, c.fundamental.m, 9 * sizeof(float));
}

int vsfh_num_vision_factors(void* f) {
  slam_types::SLAMProblem p;
  static_cast<Frontend*>(f)->GetSLAMProblem(&p);
  return (int)p.vision_factors.size();
}

int vsfh_vision_factor(void* f, int i, uint64_t* pose_initial, uint64_t* pose_current, uint64_t* pairs, int cap) {
  const auto& vf = static_cast<Frontend*>(f)->vision_factors();
  if (i < 0 || i >= (int)vf.size()) return -1;
  *pose_initial = vf[i].pose_idx_initial;
  *pose_current = vf[i].pose_idx_current;
  const int n = (int)vf[i].feature_matches.size();
  for (int k = 0; k < n && k < cap; k++) {
    pairs[2 * k] = vf[i].feature_matches[k].feature_idx_initial;
    pairs[2 * k + 1] = vf[i].feature_matches[k].feature_idx_current;
  }
  return n;
}

// pose7 = loc xyz + angle wxyz; feat = cap x 6: feature_idx, pixel x, pixel y, point3d xyz
int vsfh_node(void* f, int i, uint64_t* node_idx, double* timestamp, float pose7[7], float* feat, int cap) {
  const auto& nodes = static_cast<Frontend*>(f)->nodes();
  if (i < 0 || i >= (int)nodes.size()) return -1;
  const slam_types::SLAMNode& n = nodes[i];
  *node_idx = n.node_idx;
  *timestamp = n.timestamp;
  const float p[7] = {n.pose.loc.x(),   n.pose.loc.y(),   n.pose.loc.z(),  n.pose.angle.w(),
                      n.pose.angle.x(), n.pose.angle.y(), n.pose.angle.z()};
  std::memcpy(pose7, p, sizeof(p));
  const int m = (int)n.features.size();
  for (int k = 0; k < m && k < cap; k++) {
    const slam_types::VisionFeature& v = n.features[k];
    const float r[6] = {(float)v.feature_idx, v.pixel.x(), v.pixel.y(), v.point3d.x(), v.point3d.y(), v.point3d.z()};
    std::memcpy(feat + 6 * k, r, sizeof(r));
  }
  return m;
}

int vsfh_num_odometry_factors(void* f) { return (int)static_cast<Frontend*>(f)->odometry_factors().size(); }

int vsfh_odometry_factor(void* f, int i, uint64_t ij[2], float tq[7]) {
  const auto& of = static_cast<Frontend*>(f)->odometry_factors();
  if (i < 0 || i >= (int)of.size()) return -1;
  ij[0] = of[i].pose_i;
  ij[1] = of[i].pose_j;
  const float r[7] = {of[i].translation.x(), of[i].translation.y(), of[i].translation.z(), of[i].rotation.w(),
                      of[i].rotation.x(),    of[i].rotation.y(),    of[i].rotation.z()};
  std::memcpy(tq, r, sizeof(r));
  return 0;
}

// Keypoints / descriptors of the i-th retained frame (after RemoveAmbigStereo re-indexing).
int vsfh_frame(void* f, int i, uint64_t* frame_id, vsf_keypoint* kp, uint8_t* desc, int cap) {
  const auto& fl = static_cast<Frontend*>(f)->frame_list();
  if (i < 0 || i >= (int)fl.size()) return -1;
  *frame_id = fl[i].frame_ID_;
  const int n = (int)fl[i].keypoints_.size();
  const int m = n < cap ? n : cap;
  if (m > 0 && kp) std::memcpy(kp, fl[i].keypoints_.data(), (size_t)m * sizeof(vsf_keypoint));
  if (m > 0 && desc && fl[i].descriptors_.size() >= (size_t)m * VSF_DESC_BYTES)
    std::memcpy(desc, fl[i].descriptors_.data(), (size_t)m * VSF_DESC_BYTES);
  return n;
}


// FrontendConfig::left_cam_to_robot (h:96, cc:613-618) as the reference's caller reads it through GetConfig(): rotation
// (row-major 3 x 3) and translation; and the two calibration messages that caller writes into its bag from it
// (slam_frontend_main.cc:341-365): CameraExtrinsics (48 B) and CameraIntrinsics (32 B) payloads.
void vsfh_left_cam_to_robot(void* f, float rotation9[9], float translation3[3]) {
  const FrontendConfig c = static_cast<Frontend*>(f)->GetConfig();
  std::memcpy(rotation9, c.left_cam_to_robot.rotation().m, 9 * sizeof(float));
  const slam::Vector3f t = c.left_cam_to_robot.translation();
  translation3[0] = t.x(), translation3[1] = t.y(), translation3[2] = t.z();
}

void vsfh_serialize_calibration(void* f, uint8_t extrinsics48[48], uint8_t intrinsics32[32]) {
  const FrontendConfig c = static_cast<Frontend*>(f)->GetConfig();
  const slam::Vector3f rT = c.left_cam_to_robot.translation();
  const float t[3] = {rT.x(), rT.y(), rT.z()};
  const slam_types::CameraExtrinsics a = slam_to_ros::ExtrinsicsFromAffine(c.left_cam_to_robot.rotation().m, t);
  slam_types::CameraIntrinsics k;
  k.fx = c.intrinsics_left.fx;
  k.cx = c.intrinsics_left.cx;
  k.fy = c.intrinsics_left.fy;
  k.cy = c.intrinsics_left.cy;
  std::vector<uint8_t> b;
  slam_to_ros::SerializeExtrinsics(a, &b);
  std::memcpy(extrinsics48, b.data(), 48);
  slam_to_ros::SerializeIntrinsics(k, &b);
  std::memcpy(intrinsics32, b.data(), 32);
}

// ROS-1 wire bytes of the current SLAMProblem (slam_to_ros.h; what the reference writes into its output bag,
// slam_frontend_main.cc:341-374).  Returns the payload size; copies min(size, cap) bytes.
size_t vsfh_serialize_problem(void* f, uint8_t* out, size_t cap) {
  slam_types::SLAMProblem p;
  static_cast<Frontend*>(f)->GetSLAMProblem(&p);
  std::vector<uint8_t> bytes;
  slam_to_ros::SerializeSLAMProblem(p, &bytes);
  if (out && cap) std::memcpy(out, bytes.data(), bytes.size() < cap ? bytes.size() : cap);
  return bytes.size();
}

}  // extern "C"
