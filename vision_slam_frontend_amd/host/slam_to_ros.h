// slam_to_ros.h -- SURVEY.md section 8(f) row f3: the ROS-1 wire format of the frontend's outputs, as a standalone
// encoder (no ROS here or on the GPU box).  The reference converts slam_types to the generated message structs in
// src/slam_to_ros.h:36-124 and lets roscpp serialise them into the output bag (slam_frontend_main.cc:341-374); the
// bytes below are what that produces for msg/*.msg:
//
//   ROS-1 serialisation: little endian, fields in declaration order, fixed-size primitives raw, a variable-length
//   array T[] as uint32 count + elements, nested messages inline.  geometry_msgs/Point and /Vector3 are three
//   float64 (x, y, z); geometry_msgs/Quaternion is four float64 in the order x, y, z, w.
//
//   FeatureMatch    uint64 id_initial, id_current                                              16 B
//   VisionFeature   uint64 id, Point pixel (z = 0, slam_to_ros.h:49), Point point3d            56 B
//   VisionFactor    uint64 pose_initial, pose_current, FeatureMatch[]                    20 + 16 n B
//   RobotPose       Vector3 loc, Quaternion angle                                              56 B
//   SLAMNode        uint64 id, float64 timestamp, RobotPose, VisionFeature[]             76 + 56 n B
//   OdometryFactor  uint64 pose_i, pose_j, Vector3 translation, Quaternion rotation            72 B
//   SLAMProblem     SLAMNode[], VisionFactor[], OdometryFactor[]
//   CameraIntrinsics  float64 fx, fy, cx, cy (the order of the .msg, not of the struct's assignment)   32 B
//   CameraExtrinsics  float64[3] translation, float64[3] rotation (fixed-size arrays: no count)        48 B
//
// float members widen to float64 exactly as the reference's converters assign them.
//
// The md5sums below are what a bag writer puts into its connection headers beside type=vision_slam_frontend/<Name>:
// computed from the field lists in tools/ros_md5.py by genmsg's rule (the same tool reproduces the known
// md5sums of geometry_msgs/Point, Vector3 and Quaternion); tests/test_ros_md5.py keeps the two in step.
#ifndef VSF_HOST_SLAM_TO_ROS_H_
#define VSF_HOST_SLAM_TO_ROS_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "slam_types.h"

namespace slam_to_ros {

constexpr const char* kMd5CameraExtrinsics = "c717804541b0303dd7ecf159fee7cb1d";
constexpr const char* kMd5CameraIntrinsics = "5df0bc21162586fd1b49ebbef1d3f4d9";
constexpr const char* kMd5FeatureMatch = "0d28d3a5322605f478774fcde33f4524";
constexpr const char* kMd5OdometryFactor = "970d60bad18953b634b3a66b9842f10f";
constexpr const char* kMd5RobotPose = "8f338562bbd18f0890b5c1878b66b8b5";
constexpr const char* kMd5SLAMNode = "0e51780741a149812c832d6ae619d556";
constexpr const char* kMd5SLAMProblem = "a5ec5d26ada8532dd747269f672dd5f8";
constexpr const char* kMd5VisionFactor = "da3820cf3e135bcd0b601cacc8cf0809";
constexpr const char* kMd5VisionFeature = "9cccef6835ecad8ee3bfb40a17ba67fb";

class Writer {
 public:
  explicit Writer(std::vector<uint8_t>* out) : out_(out) {}
  void u32(uint32_t v) { raw(&v, 4); }
  void u64(uint64_t v) { raw(&v, 8); }
  void f64(double v) { raw(&v, 8); }

 private:
  void raw(const void* p, size_t n) {  // host is little endian (x86-64), as is the wire
    const uint8_t* b = static_cast<const uint8_t*>(p);
    out_->insert(out_->end(), b, b + n);
  }
  std::vector<uint8_t>* out_;
};

inline void Write(Writer* w, const slam_types::FeatureMatch& m) {  // FeatureMatchToRos, slam_to_ros.h:36-42
  w->u64(m.feature_idx_initial);
  w->u64(m.feature_idx_current);
}

inline void Write(Writer* w, const slam_types::VisionFeature& f) {  // VisionFeatureToRos, :44-58
  w->u64(f.feature_idx);
  w->f64(f.pixel.x());
  w->f64(f.pixel.y());
  w->f64(0.0);
  w->f64(f.point3d.x());
  w->f64(f.point3d.y());
  w->f64(f.point3d.z());
}

inline void Write(Writer* w, const slam_types::RobotPose& p) {  // RobotPoseToRos, :60-71
  w->f64(p.loc.x());
  w->f64(p.loc.y());
  w->f64(p.loc.z());
  w->f64(p.angle.x());
  w->f64(p.angle.y());
  w->f64(p.angle.z());
  w->f64(p.angle.w());
}

inline void Write(Writer* w, const slam_types::SLAMNode& n) {  // SLAMNodeToRos, :73-83
  w->u64(n.node_idx);
  w->f64(n.timestamp);
  Write(w, n.pose);
  w->u32((uint32_t)n.features.size());
  for (const auto& f : n.features) Write(w, f);
}

inline void Write(Writer* w, const slam_types::VisionFactor& c) {  // VisionFactorToRos, :85-94
  w->u64(c.pose_idx_initial);
  w->u64(c.pose_idx_current);
  w->u32((uint32_t)c.feature_matches.size());
  for (const auto& m : c.feature_matches) Write(w, m);
}

inline void Write(Writer* w, const slam_types::OdometryFactor& o) {  // OdometryFactorToRos, :96-109
  w->u64(o.pose_i);
  w->u64(o.pose_j);
  w->f64(o.translation.x());
  w->f64(o.translation.y());
  w->f64(o.translation.z());
  w->f64(o.rotation.x());
  w->f64(o.rotation.y());
  w->f64(o.rotation.z());
  w->f64(o.rotation.w());
}

inline void Write(Writer* w, const slam_types::CameraIntrinsics& k) {  // IntrinsicsToRos, :126-135; CameraIntrinsics.msg
  w->f64(k.fx);
  w->f64(k.fy);
  w->f64(k.cx);
  w->f64(k.cy);
}

inline void Write(Writer* w, const slam_types::CameraExtrinsics& a) {  // ExtrinsicsToRos, :137-146; CameraExtrinsics.msg
  for (int i = 0; i < 3; i++) w->f64(a.translation[i]);
  for (int i = 0; i < 3; i++) w->f64(a.rotation[i]);
}

// slam_frontend_main.cc:341-352: translation + rotation of an affine camera-to-robot transform as scaled angle-axis,
// `AngleAxisf rR(extrinsics.rotation())`: Eigen builds the quaternion of the matrix (Shoemake's branches) and from it
// angle = 2 atan2(|vec|, |w|), axis = vec / (+-|vec|) with the sign of w -- (1, 0, 0) and angle 0 for a zero vector part;
// the message carries axis.normalized() * angle, or zeros when angle <= 1e-8.
// R: row-major 3 x 3 rotation, t: translation.
inline slam_types::CameraExtrinsics ExtrinsicsFromAffine(const float R[9], const float t[3]) {
  auto m = [&](int r, int c) { return R[3 * r + c]; };
  float q[4];  // x y z w
  float tr = m(0, 0) + m(1, 1) + m(2, 2);
  if (tr > 0.f) {
    tr = std::sqrt(tr + 1.0f);
    q[3] = 0.5f * tr;
    tr = 0.5f / tr;
    q[0] = (m(2, 1) - m(1, 2)) * tr;
    q[1] = (m(0, 2) - m(2, 0)) * tr;
    q[2] = (m(1, 0) - m(0, 1)) * tr;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    tr = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0f);
    q[i] = 0.5f * tr;
    tr = 0.5f / tr;
    q[3] = (m(k, j) - m(j, k)) * tr;
    q[j] = (m(j, i) + m(i, j)) * tr;
    q[k] = (m(k, i) + m(i, k)) * tr;
  }
  float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  float angle = 0.f, axis[3] = {1.f, 0.f, 0.f};
  if (n != 0.f) {
    angle = 2.0f * std::atan2(n, std::fabs(q[3]));
    if (q[3] < 0.f) n = -n;
    for (int i = 0; i < 3; i++) axis[i] = q[i] / n;
  }
  const float an = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  slam_types::CameraExtrinsics a;
  for (int i = 0; i < 3; i++) {
    a.translation[i] = t[i];
    a.rotation[i] = angle > 1e-8 ? (axis[i] / an) * angle : 0.f;
  }
  return a;
}

// The two calibration messages the reference's driver writes beside the problem (topics "extrinsics" and "intrinsics",
// slam_frontend_main.cc:353-365), as payload bytes.
inline void SerializeExtrinsics(const slam_types::CameraExtrinsics& a, std::vector<uint8_t>* out) {
  out->clear();
  Writer w(out);
  Write(&w, a);
}
inline void SerializeIntrinsics(const slam_types::CameraIntrinsics& k, std::vector<uint8_t>* out) {
  out->clear();
  Writer w(out);
  Write(&w, k);
}

// SLAMProblemToRos (:111-124) + roscpp serialisation: the payload of one vision_slam_frontend/SLAMProblem message.
inline void SerializeSLAMProblem(const slam_types::SLAMProblem& p, std::vector<uint8_t>* out) {
  out->clear();
  Writer w(out);
  w.u32((uint32_t)p.nodes.size());
  for (const auto& n : p.nodes) Write(&w, n);
  w.u32((uint32_t)p.vision_factors.size());
  for (const auto& c : p.vision_factors) Write(&w, c);
  w.u32((uint32_t)p.odometry_factors.size());
  for (const auto& o : p.odometry_factors) Write(&w, o);
}

inline size_t SerializedSize(const slam_types::SLAMProblem& p) {
  size_t n = 12;
  for (const auto& node : p.nodes) n += 76 + 56 * node.features.size();
  for (const auto& c : p.vision_factors) n += 20 + 16 * c.feature_matches.size();
  n += 72 * p.odometry_factors.size();
  return n;
}

}  // namespace slam_to_ros
#endif  // VSF_HOST_SLAM_TO_ROS_H_
