// time_frontend.cc -- frames per second of the drop-in API itself: slam::Frontend::ObserveOdometry + ObserveImage called
// from C++ exactly where the reference's driver calls them (slam_frontend_main.cc:132,147), no Python in the loop.
//   build:  make -C tools time_frontend       run:  tools/time_frontend frames.raw W H NFRAMES [nfeatures ...]
// frames.raw = NFRAMES x 2 x H x W bytes (tools/time_frontend.py --dump writes it).  Prints one JSON object.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
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
  std::vector<std::string> only;  // "+name": run only the modes whose name contains it
  std::vector<std::pair<int, int>> opts;  // "@option=value": a vsf_option for every context (experiments)
  for (int i = 5; i < argc; i++) {
    if (argv[i][0] == '@') {
      int o = 0, v = 0;
      if (std::sscanf(argv[i] + 1, "%d=%d", &o, &v) == 2) opts.push_back({o, v});
    } else if (argv[i][0] == '+') only.push_back(argv[i] + 1);
    else nfs.push_back(std::atoi(argv[i]));
  }
  if (nfs.empty()) nfs = {2000, 10000};
  std::printf("{\"what\": \"slam::Frontend::ObserveImage from C++, %dx%d, frame_life 10 (tools/time_frontend.cc)\", \"results\": {", W, H);
  bool first = true;
  for (int nf : nfs) {
    struct Mode { const char* name; bool fused, pipelined; int frames, depth, batch, min_batch, read_every; int thread = 0; int tail_min = 0; int copy_thread = 1; };
    // read_every 1: the reference's driver unchanged -- GetSLAMProblem after every new node (slam_frontend_main.cc:320-321),
    // which waits for the queue; queued_*: the queue of host/slam_frontend.h at depth / frames per batch / min_batch
    // queued_dD_bB: the queue of host/slam_frontend.h at depth D, B frames per batch at most ("queued": the class's defaults)
    const Mode modes[] = {{"fused", true, false, 160, 1, 1, 0, 0},
                          {"unchanged_caller", true, false, 160, 1, 1, 0, 1},
                          {"call_by_call", false, false, 96, 1, 1, 0, 0},
                          {"queued_d8_b8", true, true, 832, 8, 8, 0, 0},
                          {"queued_d32_b32", true, true, 1632, 32, 32, 0, 0},
                          {"queued_d128_b64", true, true, 3232, 128, 64, 0, 0},
                          {"queued_d64_b64", true, true, 1632, 64, 64, 0, 0},
                          {"queued_launcher_thread", true, true, 3232, 0, 0, 0, 0, 1},
                          {"queued_no_copy_thread", true, true, 3232, 0, 0, 0, 0, 0, 0, 0},
                          {"queued", true, true, 3232, 0, 0, 0, 0}};
    for (const Mode& m : modes) {
      if (!only.empty()) {
        bool hit = false;
        for (const std::string& o : only) hit = hit || std::string(m.name).find(o) != std::string::npos;
        if (!hit) continue;
      }
      slam::FrontendConfig cfg;
      cfg.orb_nfeatures = nf;
      cfg.image_width = W;
      cfg.image_height = H;
      const float F[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};  // rectified synthetic pair
      for (int i = 0; i < 9; i++) cfg.fundamental.m[i] = F[i];
      slam::Frontend fe("", cfg, 0);
      fe.set_fused(m.fused);
      fe.set_pipelined(m.pipelined);
      if (m.depth > 0) fe.set_queue_depth(m.depth);
      if (m.batch > 0) fe.set_batch_frames(m.batch);
      fe.set_min_batch(m.min_batch);
      fe.set_queue_thread(m.thread != 0);
      fe.set_copy_thread(m.copy_thread != 0);
      fe.set_context_option(VSF_OPT_PYRAMID_TAIL_MIN, m.tail_min);
      for (const auto& ov : opts) fe.set_context_option(ov.first, ov.second);
      const slam::Quaternionf q(1, 0, 0, 0);
      fe.ObserveOdometry(slam::Vector3f(0, 0, 0), q, 0.0);
      const int warm = 32;
      const bool stages = !only.empty() && argc > 5 && std::string(argv[argc - 1]) == "stages";  // "... +mode stages": per-stage GPU time
      Clock::time_point t0;
      double worst = 0, sum = 0;
      for (int k = 0; k < m.frames; k++) {
        if (k == warm) {
          fe.Flush();
          if (stages && fe.context()) vsf_profile_enable(fe.context(), 1);
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
        if (m.read_every > 0 && (k + 1) % m.read_every == 0) {
          slam_types::SLAMProblem problem;
          if (k < 64) fe.GetSLAMProblem(&problem);  // (the copy grows with every node: the reference's O(n^2); bounded here)
          else fe.Flush();
        }
        if (k >= warm) {
          sum += dt;
          if (dt > worst) worst = dt;
        }
      }
      fe.Flush();
      const double wall = seconds(t0, Clock::now());
      const int n = m.frames - warm;
      if (stages && fe.context()) {
        double ms[VSF_STAGE_COUNT];
        int64_t launches[VSF_STAGE_COUNT];
        vsf_profile_read(fe.context(), ms, launches, 1);
        vsf_profile_enable(fe.context(), 0);
        for (int i = 0; i < VSF_STAGE_COUNT; i++)
          std::fprintf(stderr, "%-22s %s %8.2f us per frame (%lld launches)\n", m.name, vsf_stage_name(i), 1e3 * ms[i] / n, (long long)launches[i]);
      }
      int64_t qs[11];
      fe.queue_stats(qs);
      std::printf("%s\"%s_%d\": {\"frames_per_s\": %.1f, \"observe_image_ms_mean\": %.4f, \"observe_image_ms_max\": %.4f, "
                  "\"steady_frames\": %d, \"nodes\": %d, \"batches\": %lld, \"largest_batch\": %lld, \"forced\": %lld, "
                  "\"slot_waits\": %lld, \"copy_us_per_frame\": %.1f, \"launch_us_per_frame\": %.1f, \"wait_us_per_frame\": %.1f}",
                  first ? "" : ", ", m.name, nf, n / wall, 1e3 * sum / n, 1e3 * worst, n, fe.GetNumPoses(), (long long)qs[1],
                  (long long)qs[2], (long long)qs[4], (long long)qs[5], 1e-3 * qs[8] / m.frames, 1e-3 * qs[9] / m.frames,
                  1e-3 * qs[10] / m.frames);
      first = false;
    }
  }
  std::printf("}}\n");
  return 0;
}
