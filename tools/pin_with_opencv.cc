// pin_with_opencv.cc -- THE PIN KIT: runs the real OpenCV routines of the reference's hot path on committed inputs and
// writes their outputs in the layout tests/test_pinned_by_opencv.py compares the in-repo oracle with, stage by stage.
//
// This file CANNOT be compiled in the build container of this repository (no OpenCV there: SURVEY.md section 8(c)), and it
// has never been compiled: it is plain OpenCV 3.x C++ kept deliberately small (every call below is one of the reference's
// own call sites or the OpenCV routine behind it).  Everything that does NOT depend on OpenCV -- the .npy reader / writer
// (tools/pin_npy.h, tests/cpp/test_pin_npy.cc), the case list, the file names, the comparison and its diagnosis -- is
// exercised in the CPU test-suite with the oracle standing in for OpenCV (tests/test_pinned_by_opencv.py).
//
// On a machine WITH OpenCV 3.2.0 (the version the reference pins, CMakeLists.txt:20; build it with -DWITH_IPP=OFF or the
// 8-bit cv::resize goes through ippicv and rounds differently -- INTEGRATION.md section 8):
//
//   python3 tools/pin_inputs.py /tmp/pin_in                      # inputs + cases.txt (numpy only)
//   g++ -O2 -std=c++11 tools/pin_with_opencv.cc -o /tmp/pin_with_opencv $(pkg-config --cflags --libs opencv)
//   mkdir -p tests/golden/opencv && /tmp/pin_with_opencv /tmp/pin_in tests/golden/opencv   # <case>__*.npy + VERSION.txt
//   python3 -m pytest tests/test_pinned_by_opencv.py -q          # oracle == OpenCV, stage by stage; names the first divergence
//
// Reference call sites reproduced (file:line in ut-amrl/vision_slam_frontend, src/):
//   slam_frontend.cc:205-213  cv::ORB::create(nfeatures, 1.04f, 50, 31, 0, 2, cv::ORB::HARRIS_SCORE, 31, 20)
//   slam_frontend.cc:274-277  ->detectAndCompute(image, cv::noArray(), keypoints, descriptors)
//   slam_frontend.cc:191,271  cv::FastFeatureDetector::create(10, true)->detect(image, keypoints)
//   slam_frontend.cc:247,525  cv::BFMatcher(cv::NORM_HAMMING).knnMatch(query, train, matches, 2)
//   slam_frontend.cc:529-536  ratio test  best.distance < nn_match_ratio * second.distance  (nn_match_ratio: a double holding 0.6f)
//   slam_frontend.cc:153-157  cv::triangulatePoints(projection_left, projection_right, left_points, right_points, out)
//   slam_frontend.cc:335-340  cv::undistortPoints(pts, out, camera_matrix_left, distortion_coeffs_left, cv::noArray(), camera_matrix_left)
// and, inside cv::ORB (OpenCV 3.2.0 modules/features2d/src/orb.cpp), the two image operations whose rounding the oracle had
// to restate from memory:
//   cv::resize(prevLevel, level, Size(cvRound(cols / scale), cvRound(rows / scale)), 0, 0, cv::INTER_LINEAR)
//   cv::GaussianBlur(level, blurred, Size(7, 7), 2, 2, cv::BORDER_REFLECT_101)
#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/core/utility.hpp>
#include <opencv2/core/version.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>

#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "pin_npy.h"

namespace {

struct KeyPointRec {  // cv::KeyPoint's seven fields, 28 bytes (include/vsf.h: vsf_keypoint)
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
struct DMatchRec {  // cv::DMatch, 16 bytes (include/vsf.h: vsf_dmatch)
  int32_t queryIdx, trainIdx, imgIdx;
  float distance;
};

cv::Mat load_u8(const std::string& path) {
  const pin_npy::Array a = pin_npy::read(path);
  if (a.descr != "|u1" || a.shape.size() != 2) throw std::runtime_error("expected a 2-D uint8 array: " + path);
  cv::Mat m((int)a.shape[0], (int)a.shape[1], CV_8UC1);
  std::memcpy(m.data, a.data.data(), a.data.size());  // (a fresh Mat is continuous)
  return m;
}

cv::Mat load_f32(const std::string& path, int rows, int cols) {
  const pin_npy::Array a = pin_npy::read(path);
  if (a.descr != "<f4" || a.count() != (size_t)rows * cols) throw std::runtime_error("expected float32 x " + std::to_string(rows * cols) + ": " + path);
  cv::Mat m(rows, cols, CV_32F);
  std::memcpy(m.data, a.data.data(), a.data.size());
  return m;
}

void save_keypoints(const std::string& path, const std::vector<cv::KeyPoint>& kps) {
  std::vector<KeyPointRec> r(kps.size());
  for (size_t i = 0; i < kps.size(); i++)
    r[i] = KeyPointRec{kps[i].pt.x, kps[i].pt.y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, kps[i].class_id};
  pin_npy::write(path, pin_npy::kKeyPointDescr, {r.size()}, r.data(), r.size() * sizeof(KeyPointRec));
}

void save_u8_rows(const std::string& path, const cv::Mat& m) {  // 2-D uint8, rows copied one by one (m may be a view)
  std::vector<uint8_t> buf((size_t)m.rows * m.cols);
  for (int y = 0; y < m.rows; y++) std::memcpy(buf.data() + (size_t)y * m.cols, m.ptr<uint8_t>(y), (size_t)m.cols);
  pin_npy::write(path, "|u1", {(size_t)m.rows, (size_t)m.cols}, buf.data(), buf.size());
}

// Position-dependent checksum of a level's bytes (tests/pin_compare.py: level_digest computes the same with numpy): every
// case stores one per level and image kind, the full bytes only for the cases cases.txt marks `full` (a whole set of
// pyramids is 60 MB; the digests say WHICH level differs, and whoever has OpenCV can dump that level again).
uint64_t level_digest(const uint8_t* p, size_t n) {
  uint64_t d = 0;
  for (size_t i = 0; i < n; i++) d += ((uint64_t)p[i] + 1u) * (((uint64_t)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull) | 1ull);
  return d;
}

// The scale pyramid as cv::ORB::detectAndCompute builds it (orb.cpp: level l is resized from level l - 1, its size comes
// from the ORIGINAL image's size and the level's float scale), and the blurred copy the descriptors are sampled from.
void pyramid_and_blur(const cv::Mat& image, int nlevels, double scale_factor, std::vector<uint8_t>* pyr, std::vector<uint8_t>* blur,
                      std::vector<int32_t>* shapes, std::vector<uint64_t>* pyr_digest, std::vector<uint64_t>* blur_digest) {
  cv::Mat prev = image;
  for (int level = 0; level < nlevels; level++) {
    cv::Mat cur;
    if (level == 0) {
      cur = image;
    } else {
      const float scale = (float)std::pow(scale_factor, (double)level);  // ORB_Impl::getScale (firstLevel = 0)
      const cv::Size sz(cvRound(image.cols / scale), cvRound(image.rows / scale));
      cv::resize(prev, cur, sz, 0, 0, cv::INTER_LINEAR);
    }
    cv::Mat blurred;
    cv::GaussianBlur(cur, blurred, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    shapes->push_back(cur.rows);
    shapes->push_back(cur.cols);
    const size_t at = pyr->size();
    for (int y = 0; y < cur.rows; y++) {
      pyr->insert(pyr->end(), cur.ptr<uint8_t>(y), cur.ptr<uint8_t>(y) + cur.cols);
      blur->insert(blur->end(), blurred.ptr<uint8_t>(y), blurred.ptr<uint8_t>(y) + blurred.cols);
    }
    pyr_digest->push_back(level_digest(pyr->data() + at, pyr->size() - at));
    blur_digest->push_back(level_digest(blur->data() + at, blur->size() - at));
    prev = cur;
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cerr << "usage: " << argv[0] << " <input dir written by tools/pin_inputs.py> <output dir, e.g. tests/golden/opencv>\n";
    return 2;
  }
  const std::string in = std::string(argv[1]) + "/", out = std::string(argv[2]) + "/";
  // OpenCV's OWN code paths are what the oracle restates: an IPP build routes the 8-bit cv::resize (and more) through
  // ippicv, whose rounding differs.  --keep-ipp leaves the build's default in force.
  const bool keep_ipp = argc > 3 && std::string(argv[3]) == "--keep-ipp";
  if (!keep_ipp) cv::ipp::setUseIPP(false);
  try {
    const cv::Mat P_left = load_f32(in + "projection_left.npy", 3, 4), P_right = load_f32(in + "projection_right.npy", 3, 4);
    const cv::Mat K_left = load_f32(in + "camera_matrix_left.npy", 3, 3), dist_left = load_f32(in + "distortion_left.npy", 5, 1);
    std::ifstream cases(in + "cases.txt");
    if (!cases) throw std::runtime_error("no cases.txt in " + in);
    std::string line;
    int n_cases = 0;
    while (std::getline(cases, line)) {
      if (line.empty() || line[0] == '#') continue;
      std::istringstream ls(line);
      std::string name, left_file, right_file, detail;
      int nfeatures = 0;
      ls >> name >> nfeatures >> left_file >> right_file >> detail;  // detail: `full` = keep the pyramid bytes, else digests only
      if (name.empty() || nfeatures < 1 || right_file.empty()) throw std::runtime_error("bad line in cases.txt: " + line);
      const cv::Mat image[2] = {load_u8(in + left_file), load_u8(in + right_file)};
      const std::string pre = out + name + "__";
      // ---- ExtractFeatures (cc:266-280): ORB with the reference's literals; nfeatures is the case's (BASELINE configs) ----
      std::vector<cv::KeyPoint> kps[2];
      cv::Mat desc[2];
      for (int e = 0; e < 2; e++) {
        cv::Ptr<cv::ORB> orb = cv::ORB::create(nfeatures, 1.04f, 50, 31, 0, 2, cv::ORB::HARRIS_SCORE, 31, 20);
        orb->detectAndCompute(image[e], cv::noArray(), kps[e], desc[e]);
        const std::string side = e == 0 ? "L_" : "R_";
        save_keypoints(pre + side + "kp.npy", kps[e]);
        if (desc[e].empty()) desc[e] = cv::Mat(0, 32, CV_8UC1);
        if (desc[e].type() != CV_8UC1 || desc[e].cols != 32) throw std::runtime_error("unexpected descriptor matrix");
        save_u8_rows(pre + side + "desc.npy", desc[e]);
      }
      // ---- the FREAK branch's detector (cc:191, 271) on the left image ----
      {
        std::vector<cv::KeyPoint> fk;
        cv::FastFeatureDetector::create(10, true)->detect(image[0], fk);
        save_keypoints(pre + "L_fast10.npy", fk);
      }
      // ---- the two image operations inside ORB, level by level, on the left image ----
      {
        std::vector<uint8_t> pyr, blur;
        std::vector<int32_t> shapes;
        std::vector<uint64_t> pd, bd;
        pyramid_and_blur(image[0], 50, (double)1.04f, &pyr, &blur, &shapes, &pd, &bd);
        if (detail == "full") {
          pin_npy::write(pre + "L_pyramid.npy", "|u1", {pyr.size()}, pyr.data(), pyr.size());
          pin_npy::write(pre + "L_blur.npy", "|u1", {blur.size()}, blur.data(), blur.size());
        }
        pin_npy::write(pre + "L_pyramid_digest.npy", "<u8", {pd.size()}, pd.data(), pd.size() * 8);
        pin_npy::write(pre + "L_blur_digest.npy", "<u8", {bd.size()}, bd.data(), bd.size() * 8);
        pin_npy::write(pre + "L_level_shapes.npy", "<i4", {shapes.size() / 2, 2}, shapes.data(), shapes.size() * 4);
      }
      // ---- GetMatches (cc:521-538): knnMatch(k = 2) + the ratio test, in the reference's types ----
      std::vector<std::vector<cv::DMatch> > knn;
      cv::BFMatcher matcher(cv::NORM_HAMMING);
      if (desc[0].rows > 0 && desc[1].rows > 0) matcher.knnMatch(desc[0], desc[1], knn, 2);
      std::vector<int32_t> idx(2 * knn.size(), -1), dist(2 * knn.size(), 0x7FFFFFFF);
      std::vector<DMatchRec> good;
      const double nn_match_ratio = 0.6f;  // cc:523 (double parameter) <- cc:555 (float member 0.6f)
      for (size_t i = 0; i < knn.size(); i++) {
        for (size_t k = 0; k < knn[i].size() && k < 2; k++) {
          idx[2 * i + k] = knn[i][k].trainIdx;
          dist[2 * i + k] = (int32_t)knn[i][k].distance;
        }
        if (knn[i].size() >= 2 && knn[i][0].distance < nn_match_ratio * knn[i][1].distance) {  // cc:533
          const cv::DMatch& m = knn[i][0];
          good.push_back(DMatchRec{m.queryIdx, m.trainIdx, m.imgIdx, m.distance});
        }
      }
      pin_npy::write(pre + "knn_idx.npy", "<i4", {knn.size(), 2}, idx.data(), idx.size() * 4);
      pin_npy::write(pre + "knn_dist.npy", "<i4", {knn.size(), 2}, dist.data(), dist.size() * 4);
      pin_npy::write(pre + "matches.npy", pin_npy::kDMatchDescr, {good.size()}, good.data(), good.size() * sizeof(DMatchRec));
      // ---- Calculate3DPoints' and UndistortFeaturePoints' OpenCV calls on the matched points ----
      std::vector<cv::Point2f> lp, rp;
      for (size_t i = 0; i < good.size(); i++) {
        lp.push_back(kps[0][good[i].queryIdx].pt);
        rp.push_back(kps[1][good[i].trainIdx].pt);
      }
      std::vector<float> p4(4 * good.size(), 0.f), und(2 * good.size(), 0.f);
      if (!good.empty()) {
        cv::Mat tri;
        cv::triangulatePoints(P_left, P_right, lp, rp, tri);  // cc:153-157: 4 x N, CV_32F for float inputs
        if (tri.rows != 4 || tri.cols != (int)good.size() || tri.type() != CV_32F) throw std::runtime_error("unexpected triangulatePoints output");
        for (int r = 0; r < 4; r++)
          for (int c = 0; c < tri.cols; c++) p4[(size_t)c * 4 + r] = tri.at<float>(r, c);  // stored point-major: (N, 4)
        std::vector<cv::Point2f> up;
        cv::undistortPoints(lp, up, K_left, dist_left, cv::noArray(), K_left);  // cc:335-340
        for (size_t i = 0; i < up.size(); i++) und[2 * i] = up[i].x, und[2 * i + 1] = up[i].y;
      }
      pin_npy::write(pre + "points4d.npy", "<f4", {good.size(), 4}, p4.data(), p4.size() * 4);
      pin_npy::write(pre + "undistorted.npy", "<f4", {good.size(), 2}, und.data(), und.size() * 4);
      std::cout << name << ": " << kps[0].size() << " / " << kps[1].size() << " keypoints, " << good.size() << " matches\n";
      n_cases++;
    }
    std::ofstream ver(out + "VERSION.txt");
    ver << "OpenCV " << CV_VERSION << "\n" << "cases " << n_cases << "\n" << "useOptimized " << cv::useOptimized() << "\n";
    ver << "useIPP " << cv::ipp::useIPP() << (keep_ipp ? " (--keep-ipp: a pyramid mismatch may be ippicv's rounding)" : " (switched off for this run)") << "\n";
    ver << cv::getBuildInformation();
  } catch (const std::exception& e) {
    std::cerr << "pin_with_opencv: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
