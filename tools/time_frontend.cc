// time_frontend.cc -- frames per second of the drop-in API itself: slam::Frontend::ObserveOdometry + ObserveImage called
// from C++ exactly where the reference's driver calls them (slam_frontend_main.cc:132,147), no Python in the loop.
//   build:  make -C tools time_frontend       run:  tools/time_frontend frames.raw W H NFRAMES [nfeatures ...]
// frames.raw = NFRAMES x 2 x H x W bytes (tools/time_frontend.py --dump writes it).  Prints one JSON object.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../vision_slam_frontend_amd/host/slam_frontend.h"

using Clock = std::chrono::steady_clock;

static double seconds(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s frames.raw W H NFRAMES [nfeatures ...]\n", argv[0]);
    return 2;
  }
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), NB = std::atoi(argv[4]);
  std::vector<uint8_t> raw((size_t)NB * 2 * W * H);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(raw.data(), 1, raw.size(), f) != raw.size()) {
    std::fprintf(stderr, "cannot read %zu bytes from %s\n", raw.size(), argv[1]);
    return 2;
  }
  std::fclose(f);
  std::vector<int> nfs;
  for (int i = 5; i < argc; i++) nfs.push_back(std::atoi(argv[i]));
  if (nfs.empty()) nfs = {2000, 10000};
  std::printf("{\"what\": \"slam::Frontend::ObserveImage from C++, %dx%d, frame_life 10 (tools/time_frontend.cc)\", \"results\": {", W, H);
  bool first = true;
  for (int nf : nfs) {
    struct Mode { const char* name; bool fused, pipelined; int frames; };
    const Mode modes[] = {{"fused", true, false, 160}, {"call_by_call", false, false, 96}, {"pipelined", true, true, 432}};
    for (const Mode& m : modes) {
      slam::FrontendConfig cfg;
      cfg.orb_nfeatures = nf;
      cfg.image_width = W;
      cfg.image_height = H;
      const float F[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};  // rectified synthetic pair
      for (int i = 0; i < 9; i++) cfg.fundamental.m[i] = F[i];
      slam::Frontend fe("", cfg, 0);
      fe.set_fused(m.fused);
      fe.set_pipelined(m.pipelined);
      const slam::Quaternionf q(1, 0, 0, 0);
      fe.ObserveOdometry(slam::Vector3f(0, 0, 0), q, 0.0);
      const int warm = 32;
      Clock::time_point t0;
      double worst = 0, sum = 0;
      for (int k = 0; k < m.frames; k++) {
        if (k == warm) {
          fe.Flush();
          t0 = Clock::now();
        }
        const uint8_t* l = raw.data() + (size_t)(k % NB) * 2 * W * H;
        fe.ObserveOdometry(slam::Vector3f(0.3f * (k + 1), 0, 0), q, 1.0 + k);
        const Clock::time_point a = Clock::now();
        const bool added = fe.ObserveImage(slam::Image(l, H, W, (size_t)W), slam::Image(l + (size_t)W * H, H, W, (size_t)W), 1.0 + k);
        const double dt = seconds(a, Clock::now());
        if (!added || fe.last_status() != VSF_OK) {
          std::fprintf(stderr, "ObserveImage failed at frame %d (status %d)\n", k, (int)fe.last_status());
          return 1;
        }
        if (k >= warm) {
          sum += dt;
          if (dt > worst) worst = dt;
        }
      }
      fe.Flush();
      const double wall = seconds(t0, Clock::now());
      const int n = m.frames - warm;
      std::printf("%s\"%s_%d\": {\"frames_per_s\": %.1f, \"observe_image_ms_mean\": %.4f, \"observe_image_ms_max\": %.4f, "
                  "\"steady_frames\": %d, \"nodes\": %d}",
                  first ? "" : ", ", m.name, nf, n / wall, 1e3 * sum / n, 1e3 * worst, n, fe.GetNumPoses());
      first = false;
    }
  }
  std::printf("}}\n");
  return 0;
}
