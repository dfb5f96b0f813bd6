// Known-answer test of host/slam_to_ros.h: a hand-built SLAMProblem against bytes assembled
// field by field from the ROS-1 serialisation rules and msg/*.msg.
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
  std::printf(bad ? "FAIL\n" : "ok %zu bytes\n", got.size());
  return bad;
}
