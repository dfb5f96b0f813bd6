// slam_types.h -- output records of the frontend, field for field the reference's src/slam_types.h:39-187
// (and msg/*.msg), with Eigen::Vector2f/Vector3f/Quaternionf replaced by the small PODs below (Eigen is not
// available in this toolchain).  Member names are unchanged so code written against the reference compiles
// after a typedef swap.
#ifndef VSF_HOST_SLAM_TYPES_H_
#define VSF_HOST_SLAM_TYPES_H_

#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace slam_types {

struct Vector2f {
  float x_, y_;
  Vector2f() : x_(0), y_(0) {}
  Vector2f(float x, float y) : x_(x), y_(y) {}
  float x() const { return x_; }
  float y() const { return y_; }
};

struct Vector3f {
  float x_, y_, z_;
  Vector3f() : x_(0), y_(0), z_(0) {}
  Vector3f(float x, float y, float z) : x_(x), y_(y), z_(z) {}
  float x() const { return x_; }
  float y() const { return y_; }
  float z() const { return z_; }
  Vector3f operator+(const Vector3f& o) const { return Vector3f(x_ + o.x_, y_ + o.y_, z_ + o.z_); }
  Vector3f operator-(const Vector3f& o) const { return Vector3f(x_ - o.x_, y_ - o.y_, z_ - o.z_); }
  Vector3f operator*(float s) const { return Vector3f(x_ * s, y_ * s, z_ * s); }
  Vector3f operator/(float s) const { return Vector3f(x_ / s, y_ / s, z_ / s); }
  Vector3f cross(const Vector3f& o) const {
    return Vector3f(y_ * o.z_ - z_ * o.y_, z_ * o.x_ - x_ * o.z_, x_ * o.y_ - y_ * o.x_);
  }
  float squaredNorm() const { return x_ * x_ + y_ * y_ + z_ * z_; }
  float norm() const { return std::sqrt(squaredNorm()); }
};

// Unit quaternion with Eigen's (w, x, y, z) constructor order and the operations the frontend uses.
struct Quaternionf {
  float w_, x_, y_, z_;
  Quaternionf() : w_(1), x_(0), y_(0), z_(0) {}
  Quaternionf(float w, float x, float y, float z) : w_(w), x_(x), y_(y), z_(z) {}
  float w() const { return w_; }
  float x() const { return x_; }
  float y() const { return y_; }
  float z() const { return z_; }
  Vector3f vec() const { return Vector3f(x_, y_, z_); }
  Quaternionf conjugate() const { return Quaternionf(w_, -x_, -y_, -z_); }
  float squaredNorm() const { return w_ * w_ + x_ * x_ + y_ * y_ + z_ * z_; }
  Quaternionf inverse() const {
    const float n2 = squaredNorm();
    if (n2 > 0) return Quaternionf(w_ / n2, -x_ / n2, -y_ / n2, -z_ / n2);
    return Quaternionf(0, 0, 0, 0);
  }
  Quaternionf operator*(const Quaternionf& b) const {
    return Quaternionf(w_ * b.w_ - x_ * b.x_ - y_ * b.y_ - z_ * b.z_, w_ * b.x_ + x_ * b.w_ + y_ * b.z_ - z_ * b.y_,
                       w_ * b.y_ + y_ * b.w_ + z_ * b.x_ - x_ * b.z_, w_ * b.z_ + z_ * b.w_ + x_ * b.y_ - y_ * b.x_);
  }
  // Rotation of a vector (Eigen's QuaternionBase::_transformVector).
  Vector3f operator*(const Vector3f& v) const {
    const Vector3f uv = vec().cross(v) * 2.0f;
    return v + uv * w_ + vec().cross(uv);
  }
  float angularDistance(const Quaternionf& other) const {
    const Quaternionf d = (*this) * other.conjugate();
    return 2.0f * std::atan2(d.vec().norm(), std::fabs(d.w_));
  }
};

// slam_types.h:39-58
struct CameraIntrinsics {
  float fx, fy, cx, cy;
};
struct CameraExtrinsics {
  float translation[3];
  float rotation[3];
};

// slam_types.h:60-75 (VisionFeature.msg: uint64 id, Point pixel, Point point3d)
struct VisionFeature {
  uint64_t feature_idx;
  Vector2f pixel;
  Vector3f point3d;
  VisionFeature() {}
  VisionFeature(uint64_t idx, const Vector2f& p, const Vector3f& point3d)
      : feature_idx(idx), pixel(p), point3d(point3d) {}
};

// slam_types.h:77-89 (FeatureMatch.msg: uint64 id_initial, id_current)
struct FeatureMatch {
  uint64_t feature_idx_initial;
  uint64_t feature_idx_current;
  FeatureMatch() {}
  FeatureMatch(uint64_t fid_initial, uint64_t fid_current)
      : feature_idx_initial(fid_initial), feature_idx_current(fid_current) {}
};

// slam_types.h:91-108
struct VisionFactor {
  uint64_t pose_idx_initial;
  uint64_t pose_idx_current;
  std::vector<FeatureMatch> feature_matches;
  VisionFactor() {}
  VisionFactor(uint64_t pose_initial, uint64_t pose_current, const std::vector<FeatureMatch>& feature_matches)
      : pose_idx_initial(pose_initial), pose_idx_current(pose_current), feature_matches(feature_matches) {}
  VisionFactor(uint64_t pose_initial, uint64_t pose_current, std::vector<FeatureMatch>&& feature_matches)  // (addition)
      : pose_idx_initial(pose_initial), pose_idx_current(pose_current), feature_matches(std::move(feature_matches)) {}
};

// slam_types.h:110-130
struct RobotPose {
  Vector3f loc;
  Quaternionf angle;
  RobotPose() {}
  RobotPose(const Vector3f& loc, const Quaternionf& angle) : loc(loc), angle(angle) {}
};

// slam_types.h:132-150
struct OdometryFactor {
  uint64_t pose_i;
  uint64_t pose_j;
  Vector3f translation;
  Quaternionf rotation;
  OdometryFactor() {}
  OdometryFactor(uint64_t pose_i, uint64_t pose_j, Vector3f translation, Quaternionf rotation)
      : pose_i(pose_i), pose_j(pose_j), translation(translation), rotation(rotation) {}
};

// slam_types.h:152-169
struct SLAMNode {
  uint64_t node_idx;
  double timestamp;
  RobotPose pose;
  std::vector<VisionFeature> features;
  SLAMNode() {}
  SLAMNode(uint64_t idx, double timestamp, const RobotPose& pose, const std::vector<VisionFeature>& features)
      : node_idx(idx), timestamp(timestamp), pose(pose), features(features) {}
  SLAMNode(uint64_t idx, double timestamp, const RobotPose& pose, std::vector<VisionFeature>&& features)  // (addition)
      : node_idx(idx), timestamp(timestamp), pose(pose), features(std::move(features)) {}
};

// slam_types.h:171-187
struct SLAMProblem {
  std::vector<SLAMNode> nodes;
  std::vector<VisionFactor> vision_factors;
  std::vector<OdometryFactor> odometry_factors;
  SLAMProblem() {}
  SLAMProblem(const std::vector<SLAMNode>& nodes, const std::vector<VisionFactor>& vision_factors,
              const std::vector<OdometryFactor>& odometry_factors)
      : nodes(nodes), vision_factors(vision_factors), odometry_factors(odometry_factors) {}
};

}  // namespace slam_types

#endif  // VSF_HOST_SLAM_TYPES_H_
