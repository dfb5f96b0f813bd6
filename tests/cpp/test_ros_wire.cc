// Known-answer test of host/slam_to_ros.h: a hand-built SLAMProblem against bytes assembled
// field by field from the ROS-1 serialisation rules and msg/*.msg.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../vision_slam_frontend_amd/host/slam_to_ros.h"

using namespace slam_types;

static std::vector<uint8_t> expect;
static void U32(uint32_t v) { for (int i = 0; i < 4; i++) expect.push_back((uint8_t)(v >> (8 * i))); }
static void U64(uint64_t v) { for (int i = 0; i < 8; i++) expect.push_back((uint8_t)(v >> (8 * i))); }
static void F64(double d) { uint64_t v; std::memcpy(&v, &d, 8); U64(v); }

int main() {
  SLAMProblem p;
  std::vector<VisionFeature> feats = {VisionFeature(0, Vector2f(10.5f, 20.25f), Vector3f(1.f, -2.f, 3.5f)),
                                      VisionFeature(1, Vector2f(0.1f, 479.f), Vector3f(0.f, 0.f, 0.f))};
  p.nodes.push_back(SLAMNode(0, 123.456, RobotPose(Vector3f(1.f, 2.f, 3.f), Quaternionf(0.5f, 0.1f, 0.2f, 0.3f)), feats));
  p.nodes.push_back(SLAMNode(1, 124.0, RobotPose(Vector3f(4.f, 5.f, 6.f), Quaternionf(1.f, 0.f, 0.f, 0.f)), {}));
  p.vision_factors.push_back(VisionFactor(0, 1, {FeatureMatch(7, 9), FeatureMatch(1, 0)}));
  p.odometry_factors.push_back(OdometryFactor(0, 1, Vector3f(3.f, 3.f, 3.f), Quaternionf(0.9f, 0.f, 0.1f, 0.f)));

  // nodes
  U32(2);
  U64(0); F64(123.456);
  F64(1.f); F64(2.f); F64(3.f);                      // loc
  F64(0.1f); F64(0.2f); F64(0.3f); F64(0.5f);        // Quaternion x y z w
  U32(2);
  U64(0); F64(10.5f); F64(20.25f); F64(0.0); F64(1.f); F64(-2.f); F64(3.5f);
  U64(1); F64(0.1f); F64(479.f); F64(0.0); F64(0.f); F64(0.f); F64(0.f);
  U64(1); F64(124.0);
  F64(4.f); F64(5.f); F64(6.f);
  F64(0.f); F64(0.f); F64(0.f); F64(1.f);
  U32(0);
  // vision factors
  U32(1);
  U64(0); U64(1); U32(2); U64(7); U64(9); U64(1); U64(0);
  // odometry factors
  U32(1);
  U64(0); U64(1); F64(3.f); F64(3.f); F64(3.f); F64(0.f); F64(0.1f); F64(0.f); F64(0.9f);

  std::vector<uint8_t> got;
  slam_to_ros::SerializeSLAMProblem(p, &got);
  int bad = 0;
  if (got.size() != expect.size() || got.size() != slam_to_ros::SerializedSize(p)) {
    std::printf("size %zu vs %zu vs %zu\n", got.size(), expect.size(), slam_to_ros::SerializedSize(p));
    bad = 1;
  } else if (std::memcmp(got.data(), expect.data(), got.size()) != 0) {
    std::printf("bytes differ\n");
    bad = 1;
  }
  // record sizes quoted in SURVEY.md 8(a) row a8
  if (expect.size() != 12 + (76 + 2 * 56) + 76 + (20 + 2 * 16) + 72) bad = 1;
  // ---- CameraIntrinsics.msg: float64 fx, fy, cx, cy (declaration order) ----
  {
    slam_types::CameraIntrinsics k;
    k.fx = 527.873518f, k.cx = 482.823413f, k.fy = 527.276819f, k.cy = 298.033945f;  // main.cc:358-361 assigns fx cx fy cy
    expect.clear();
    F64(527.873518f); F64(527.276819f); F64(482.823413f); F64(298.033945f);
    std::vector<uint8_t> b;
    slam_to_ros::SerializeIntrinsics(k, &b);
    if (b.size() != 32 || b != expect) std::printf("intrinsics differ\n"), bad = 1;
  }
  // ---- CameraExtrinsics.msg: float64[3] translation, float64[3] rotation, no counts ----
  {
    // a quarter turn about z: quaternion (w, x, y, z) = (cos 45, 0, 0, sin 45): angle = 2 atan2(sin 45, cos 45) = pi / 2
    const float Rz[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, t[3] = {-0.01f, 0.06f, 0.53f};
    const slam_types::CameraExtrinsics a = slam_to_ros::ExtrinsicsFromAffine(Rz, t);
    const float half = std::sqrt(0.0f + 1.0f + 0.0f + 0.0f + 1.0f) * 0.5f;  // w = 0.5 sqrt(trace + 1), trace = 1
    const float zq = (1.0f - (-1.0f)) * (0.5f / std::sqrt(2.0f));           // z = (m10 - m01) * 0.5 / sqrt(trace + 1)
    const float angle = 2.0f * std::atan2(zq, half);
    expect.clear();
    F64(-0.01f); F64(0.06f); F64(0.53f);
    F64(0.0f); F64(0.0f); F64((zq / zq) / 1.0f * angle);
    std::vector<uint8_t> b;
    slam_to_ros::SerializeExtrinsics(a, &b);
    if (b.size() != 48 || b != expect) std::printf("extrinsics (quarter turn) differ\n"), bad = 1;
    if (std::fabs(a.rotation[2] - 1.5707964f) > 1e-6f) std::printf("quarter turn is %.8f\n", a.rotation[2]), bad = 1;
    // the identity: angle 0 <= 1e-8 -> zeros (main.cc:347-351)
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const slam_types::CameraExtrinsics z = slam_to_ros::ExtrinsicsFromAffine(I, t);
    if (z.rotation[0] != 0.f || z.rotation[1] != 0.f || z.rotation[2] != 0.f) std::printf("identity\n"), bad = 1;
    // a half turn about x (trace = -1: the other branch of the quaternion construction): (pi, 0, 0)
    const float Rx[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1};
    const slam_types::CameraExtrinsics h = slam_to_ros::ExtrinsicsFromAffine(Rx, t);
    if (std::fabs(h.rotation[0] - 3.14159265f) > 1e-6f || h.rotation[1] != 0.f || h.rotation[2] != 0.f)
      std::printf("half turn %.8f %.8f %.8f\n", h.rotation[0], h.rotation[1], h.rotation[2]), bad = 1;
    // the reference's own left_cam_to_robot (cc:613-618) against the angle-axis of that matrix evaluated in double
    // (log map: angle = acos((trace - 1) / 2), axis ~ (m21 - m12, m02 - m20, m10 - m01))
    const float RT[9] = {0.009916590468f, -0.2835522866f, 0.9589055021f,  -0.9998698619f, -0.01501486552f,
                         0.005900269087f, 0.01272480238f, -0.9588392225f, -0.2836642819f};
    const slam_types::CameraExtrinsics e = slam_to_ros::ExtrinsicsFromAffine(RT, t);
    const double trd = (double)RT[0] + RT[4] + RT[8], ang = std::acos((trd - 1.0) / 2.0);
    double ax[3] = {(double)RT[7] - RT[5], (double)RT[2] - RT[6], (double)RT[3] - RT[1]};
    const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    for (int i = 0; i < 3; i++)
      if (std::fabs(e.rotation[i] - ax[i] / an * ang) > 2e-5) std::printf("left_cam_to_robot axis %d: %.7f vs %.7f\n", i, e.rotation[i], ax[i] / an * ang), bad = 1;
  }
  std::printf(bad ? "FAIL\n" : "ok %zu bytes + intrinsics + extrinsics\n", got.size());
  return bad;
}
