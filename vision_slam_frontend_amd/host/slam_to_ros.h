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
//
// float members widen to float64 exactly as the reference's converters assign them.
#ifndef VSF_HOST_SLAM_TO_ROS_H_
#define VSF_HOST_SLAM_TO_ROS_H_

#include <cstdint>
#include <cstring>
#include <vector>

#include "slam_types.h"

namespace slam_to_ros {

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
