// stl_permutations.cc -- part of the pin kit: the permutations libstdc++ leaves behind in the two places where the
// reference's results depend on them, dumped as text for committed (deterministically generated) key sets, so that the
// question "does GCC 7's libstdc++ (the reference's toolchain: Ubuntu 18.04 / ROS melodic, jenkins-ci-build.sh:2) permute like
// the GCC 11 this repository's oracle was built with?" is settled by diffing two files:
//
//   cv::KeyPointsFilter::retainBest (features2d/keypoint.cpp; called twice per pyramid level by cv::ORB, slam_frontend.cc:274):
//       std::nth_element(begin, begin + n, end, response-greater); ambiguous = kps[n - 1].response;
//       new_end = std::partition(begin + n, end, response >= ambiguous); resize
//   Frontend::GetFeatureMatches (slam_frontend.cc:289-291): std::sort(matches.begin(), matches.end()) by DMatch::operator<
//       (distance only, unstable), then the first int(size * best_percent)
//
// Output (stdout): one line per case, "<kind> <case id> <n> <keep> : <ids of the survivors in the order libstdc++ leaves>".
// Key sets: a fixed 64-bit LCG (below) drives sizes, key ranges (few distinct values = many ties, which is where
// implementations differ) and cut points; nothing else is random.  tests/test_stl_permutations.py builds this file with the
// local compiler and compares with tests/golden/opencv/stl_permutations_gcc7.txt when the pin kit has produced it.
//   g++ -O2 -std=c++11 tools/pin/stl_permutations.cc -o stl_permutations && ./stl_permutations > out.txt
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Kp {
  float response;
  int id;
};
struct KpGreater {
  bool operator()(const Kp& a, const Kp& b) const { return a.response > b.response; }
};
struct Dm {
  float distance;
  int id;
  bool operator<(const Dm& o) const { return distance < o.distance; }
};

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(g_state >> 33);
}

int main() {
  std::printf("# libstdc++ %d (__GLIBCXX__), compiler %s\n", (int)__GLIBCXX__, __VERSION__);
  // ---- retainBest: sizes around the per-level budgets of cv::ORB (8 .. 900 kept of up to 9000), FAST scores (0..255: ties
  // everywhere) and Harris responses (floats with a few exact repeats) ----
  for (int c = 0; c < 400; c++) {
    const int n = 2 + (int)(rnd() % (c < 300 ? 1200 : 9000));
    const int keep = 1 + (int)(rnd() % (unsigned)n);
    const int mode = c % 4;  // 0: byte scores, 1: 16 distinct values, 2: floats with repeats, 3: sorted descending byte scores
    std::vector<Kp> v((size_t)n);
    for (int i = 0; i < n; i++) {
      float r;
      if (mode == 0 || mode == 3) r = (float)(rnd() % 200 + 20);
      else if (mode == 1) r = (float)(rnd() % 16);
      else r = (rnd() % 8 == 0) ? 0.001f * (float)(rnd() % 50) : 1e-6f * (float)(rnd() % 1000000);
      v[(size_t)i] = Kp{r, i};
    }
    if (mode == 3) std::stable_sort(v.begin(), v.end(), KpGreater());
    if (keep < n) {
      std::nth_element(v.begin(), v.begin() + keep, v.end(), KpGreater());
      const float ambiguous = v[(size_t)keep - 1].response;
      std::vector<Kp>::iterator new_end =
          std::partition(v.begin() + keep, v.end(), [ambiguous](const Kp& k) { return k.response >= ambiguous; });
      v.resize((size_t)(new_end - v.begin()));
    }
    std::printf("retainBest %d %d %d :", c, n, keep);
    for (const Kp& k : v) std::printf(" %d", k.id);
    std::printf("\n");
  }
  // ---- std::sort by distance: Hamming distances 0..256 (ties by the hundred), then the cut ----
  for (int c = 0; c < 400; c++) {
    const int n = (int)(rnd() % (c < 300 ? 700 : 6000));
    const int spread = 1 + (int)(rnd() % 120);
    std::vector<Dm> v((size_t)n);
    for (int i = 0; i < n; i++) v[(size_t)i] = Dm{(float)(rnd() % (unsigned)spread + (c % 3 == 0 ? 0 : 10)), i};
    if (c % 5 == 4) std::reverse(v.begin(), v.end());
    std::sort(v.begin(), v.end());
    const float best_percent = (c % 2) ? 0.3f : 1.0f;
    const int good = (int)((float)v.size() * best_percent);
    std::printf("sort %d %d %d :", c, n, good);
    for (int i = 0; i < good; i++) std::printf(" %d", v[(size_t)i].id);
    std::printf("\n");
  }
  return 0;
}
