// test_pin_npy.cc -- the .npy reader / writer of the OpenCV pin kit (tools/pin_npy.h) without OpenCV:
//   test_pin_npy write <dir>   writes one file of every kind tools/pin_with_opencv.cc writes (keypoint records, DMatch
//                              records, 2-D uint8, flat uint8, (n, 2) int32, (n, 4) float32, an EMPTY keypoint array)
//   test_pin_npy read <dir>    reads what tests/test_pinned_by_opencv.py wrote with numpy.save (a 2-D uint8 image, float32
//                              matrices) and prints checksums the test compares
// Built and run by tests/test_pinned_by_opencv.py::test_cpp_npy_files_round_trip_through_numpy (also under ASan / UBSan).
#include <cstdio>
#include <string>

#include "../../tools/pin_npy.h"

struct KeyPointRec {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
struct DMatchRec {
  int32_t queryIdx, trainIdx, imgIdx;
  float distance;
};

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  const std::string mode = argv[1], dir = std::string(argv[2]) + "/";
  try {
    if (mode == "write") {
      std::vector<KeyPointRec> kp;
      for (int i = 0; i < 5; i++) kp.push_back(KeyPointRec{1.5f * i, 2.25f * i, 31.0f * (1 + i), 45.5f * i, 1e-3f * i, i, -1});
      pin_npy::write(dir + "kp.npy", pin_npy::kKeyPointDescr, {kp.size()}, kp.data(), kp.size() * sizeof(KeyPointRec));
      pin_npy::write(dir + "kp_empty.npy", pin_npy::kKeyPointDescr, {0}, nullptr, 0);
      std::vector<DMatchRec> dm = {{0, 7, 0, 12.f}, {3, 1, 0, 40.f}};
      pin_npy::write(dir + "matches.npy", pin_npy::kDMatchDescr, {dm.size()}, dm.data(), dm.size() * sizeof(DMatchRec));
      std::vector<uint8_t> img(7 * 13);
      for (size_t i = 0; i < img.size(); i++) img[i] = (uint8_t)(i * 3);
      pin_npy::write(dir + "desc.npy", "|u1", {7, 13}, img.data(), img.size());
      pin_npy::write(dir + "flat.npy", "|u1", {img.size()}, img.data(), img.size());
      std::vector<int32_t> idx = {1, -1, 5, 2, 0x7FFFFFFF, 0};
      pin_npy::write(dir + "idx.npy", "<i4", {3, 2}, idx.data(), idx.size() * 4);
      std::vector<float> p4 = {1.f, 2.f, 3.f, 4.f, -0.5f, 1e-20f, 3e8f, 1.f};
      pin_npy::write(dir + "p4.npy", "<f4", {2, 4}, p4.data(), p4.size() * 4);
      return 0;
    }
    if (mode == "digest") {  // the level checksum of tools/pin_with_opencv.cc, restated (that file needs OpenCV to compile)
      const pin_npy::Array im = pin_npy::read(dir + "image.npy");
      auto digest = [](const uint8_t* p, size_t n) {
        uint64_t d = 0;
        for (size_t i = 0; i < n; i++) d += ((uint64_t)p[i] + 1u) * (((uint64_t)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull) | 1ull);
        return d;
      };
      std::printf("%llu %llu\n", (unsigned long long)digest(im.data.data(), im.data.size()), (unsigned long long)digest(nullptr, 0));
      return 0;
    }
    if (mode == "read") {
      const pin_npy::Array im = pin_npy::read(dir + "image.npy");
      unsigned long sum = 0;
      for (uint8_t b : im.data) sum = sum * 31 + b;
      std::printf("image %s %zu %zu %zu %lu\n", im.descr.c_str(), im.shape.size(), im.shape[0], im.shape[1], sum % 1000000007ul);
      const pin_npy::Array P = pin_npy::read(dir + "P.npy");
      const float* f = reinterpret_cast<const float*>(P.data.data());
      std::printf("P %s %zu %zu %.9g %.9g\n", P.descr.c_str(), P.shape[0], P.shape[1], (double)f[0], (double)f[11]);
      const pin_npy::Array d = pin_npy::read(dir + "dist.npy");
      std::printf("dist %s %zu %zu\n", d.descr.c_str(), d.count(), d.itemsize);
      const pin_npy::Array k = pin_npy::read(dir + "kp_np.npy");  // a structured array numpy wrote
      std::printf("kp %zu %zu\n", k.count(), k.itemsize);
      return 0;
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 2;
}
