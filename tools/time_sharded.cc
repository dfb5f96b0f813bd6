// time_sharded.cc -- the sharded hot path of BASELINE configs[3] WITHOUT Python: one thread per GPU, one vsf context and
// one vsf_comm (RCCL) each, the exchange of DESIGN.md section 7 (its ten steps: NOTES.md section 7) through the C ABI of include/vsf.h only.  What the drop-in for
// the reference's driver (slam_frontend_main.cc:251, 132) would run on an 8-GPU node.
//
//   build:  make -C tools time_sharded
//   run:    tools/time_sharded frames.raw W H NFRAMES nfeatures frames_per_rank window steps [payloads.bin] [gpus]
//
// frames.raw = NFRAMES x 2 x H x W bytes in global frame order (step-major, rank-major: frame_block() of
// vision_slam_frontend_amd/distributed.py); every rank uploads its own blocks once (frames resident in HBM, as in bench.py)
// and takes them in turn.  With `payloads.bin` rank 0 writes, per
// step and rank, a u32 byte count and the gathered payload -- the bytes tests/test_gpu_comm.py compares with the Python
// path's.  Prints one JSON object (frames/s over the timed steps, per-rank times, the RCCL version, ranks seen).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/vsf.h"

extern "C" void vsfh_default_calibration(vsf_calibration* out);

using Clock = std::chrono::steady_clock;

// A failed call ends the PROCESS: the other ranks' threads are waiting in a barrier or inside an RCCL collective for this one
// and would wait for ever.
#define CK(call)                                                                                   \
  do {                                                                                             \
    const int st_ = (int)(call);                                                                   \
    if (st_ != 0) {                                                                                \
      std::fprintf(stderr, "rank %d: %s failed with %d (%s:%d)\n", rank, #call, st_, __FILE__, __LINE__); \
      std::fflush(stderr);                                                                         \
      (void)failed;                                                                                \
      std::_Exit(1);                                                                               \
    }                                                                                              \
  } while (0)

// Watchdog: every rank notes the stage it has reached; when NO rank reaches a new one for VSF_SHARDED_STALL_S seconds
// (default 300) the process prints each rank's stage and leaves with 124 -- a rank stuck inside ncclCommInitRank or a
// collective must not eat the caller's timeout silently (the same rule as bench.py's launcher).
static std::atomic<const char*> g_stage[64];
static std::atomic<long> g_stage_gen(0);
static std::atomic<bool> g_finished(false);
static void at_stage(int rank, const char* name) {
  if (rank >= 0 && rank < 64) g_stage[rank].store(name);
  g_stage_gen.fetch_add(1);
}
static void watchdog_main(int world, double stall_s) {
  long seen = -1;
  Clock::time_point last = Clock::now();
  while (!g_finished.load()) {
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    const long g = g_stage_gen.load();
    if (g != seen) seen = g, last = Clock::now();
    if (std::chrono::duration<double>(Clock::now() - last).count() > stall_s) {
      std::fprintf(stderr, "time_sharded: no rank reached a new stage for %.0f s -- giving up.  Last stage per rank:\n", stall_s);
      for (int r = 0; r < world && r < 64; r++) {
        const char* st = g_stage[r].load();
        std::fprintf(stderr, "  rank %d: %s\n", r, st ? st : "(not started)");
      }
      std::fflush(stderr);
      std::_Exit(124);
    }
  }
}

struct Barrier {  // (C++17: no std::barrier)
  std::mutex m;
  std::condition_variable cv;
  int n, count = 0, gen = 0;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const int g = gen;
    if (++count == n) {
      count = 0;
      gen++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};

struct Shared {
  int W, H, nf, B, window, steps, world, nframes;
  const uint8_t* frames;
  uint8_t id[VSF_COMM_ID_BYTES];
  Barrier* bar;
  std::vector<double> rank_seconds;
  std::vector<int> ranks_seen;
  int rccl_version = 0;
  std::vector<std::vector<uint8_t>> payloads;  // rank 0: [step * world + r]
  bool keep = false;
  int fast_form = 0;
  float tune_ms[2] = {0.f, 0.f};
};

template <class T>
static T* dmalloc(size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemset(p, 0, n * sizeof(T));
  return static_cast<T*>(p);
}

static void rank_main(int rank, Shared* S, std::atomic<bool>* failed) {
  const int B = S->B, Wn = S->window, world = S->world, W = S->W, H = S->H;
  at_stage(rank, "hipSetDevice / vsf_create");
  CK(hipSetDevice(rank));
  // Two contexts, two streams (as ShardedStereoFrontend): the extraction of step s + 1 on the main stream beside the tail
  // of step s -- RemoveAmbigStereo, temporal matches, 3-D points, payload, every exchange -- on a high-priority stream.
  int prio_lo = 0, prio_hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  hipStream_t s_main = nullptr, s_tail = nullptr;
  CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&s_tail, hipStreamNonBlocking, prio_hi));
  vsf_params p;
  CK(vsf_params_default(&p, W, H, 2 * B));
  p.nfeatures = S->nf;
  vsf_ctx *ctx = nullptr, *tctx = nullptr;
  CK(vsf_create(&p, rank, &ctx));
  CK(vsf_get_params(ctx, &p));
  vsf_params pt = p;
  pt.max_images = 2;
  CK(vsf_create(&pt, rank, &tctx));
  CK(vsf_set_stream(ctx, s_main));
  CK(vsf_set_stream(tctx, s_tail));
  CK(vsf_reserve(tctx, B, std::max(B * Wn, 1)));  // the tail context serves B frames and B x window pairs (set-up, blocking)
  const size_t K = (size_t)p.max_keypoints;
  vsf_comm* comm = nullptr;
  at_stage(rank, "vsf_comm_create (ncclCommInitRank)");
  CK(vsf_comm_create(tctx, S->id, rank, world, &comm));
  vsf_calibration calib;
  vsfh_default_calibration(&calib);
  const float F[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};  // rectified synthetic pairs: l^T F r = y_r - y_l
  std::memcpy(calib.fundamental, F, sizeof(F));

  at_stage(rank, "device buffers and uploads");
  // ---- device buffers (the layout of ShardedStereoFrontend) ----
  const size_t img_bytes = (size_t)W * H;
  const int blocks = S->nframes / (world * B);  // steps' worth of input; resident in HBM, taken in turn
  uint8_t* d_img = dmalloc<uint8_t>((size_t)blocks * 2 * B * img_bytes);
  vsf_keypoint* d_kp[2];
  uint8_t* d_desc[2];
  int32_t *d_counts[2], *d_nmatches[2];
  vsf_dmatch* d_matches[2];
  for (int b = 0; b < 2; b++) {  // the raw outputs of the extraction, double-buffered
    d_kp[b] = dmalloc<vsf_keypoint>(2 * B * K);
    d_desc[b] = dmalloc<uint8_t>(2 * B * K * 32);
    d_counts[b] = dmalloc<int32_t>(2 * B);
    d_matches[b] = dmalloc<vsf_dmatch>(B * K);
    d_nmatches[b] = dmalloc<int32_t>(B);
  }
  float* d_means = dmalloc<float>(B);
  float* d_means_all = dmalloc<float>((size_t)world * B);
  float* d_thr_all = dmalloc<float>((size_t)world * B);
  float* d_thr_state = dmalloc<float>(1);
  const int nsets = 2 * B + 2 * world * Wn + 1, empty_set = nsets - 1;
  vsf_keypoint* d_kpf = dmalloc<vsf_keypoint>(2 * B * K);
  uint8_t* d_descf = dmalloc<uint8_t>((size_t)nsets * K * 32);
  int32_t* d_countsf = dmalloc<int32_t>(nsets);
  uint8_t* d_tail_desc = dmalloc<uint8_t>((size_t)std::max(Wn, 1) * K * 32);
  int32_t* d_tail_counts = dmalloc<int32_t>(std::max(Wn, 1));
  vsf_vision_feature* d_feat = dmalloc<vsf_vision_feature>(B * K);
  int32_t* d_nfeat = dmalloc<int32_t>(B);
  const int NP = B * Wn;
  uint64_t* d_pairs = dmalloc<uint64_t>((size_t)std::max(NP, 1) * K * 2);
  int32_t* d_npairs = dmalloc<int32_t>(std::max(NP, 1));
  const size_t cap = vsf_packed_outputs_capacity(ctx, B, NP);
  constexpr int SLOTS = 3;  // payload slots: a slot's gather has finished (stream order) before the slot is packed again
  uint8_t* d_payload[SLOTS];
  int32_t* d_sizes[SLOTS];
  int32_t* h_sizes[SLOTS];
  hipEvent_t ev_sizes[SLOTS];
  uint8_t* d_recv[SLOTS];
  for (int k = 0; k < SLOTS; k++) {
    d_payload[k] = dmalloc<uint8_t>(cap);
    d_sizes[k] = dmalloc<int32_t>(world);
    CK(hipHostMalloc((void**)&h_sizes[k], sizeof(int32_t) * world, hipHostMallocDefault));
    CK(hipEventCreateWithFlags(&ev_sizes[k], hipEventDisableTiming));
    d_recv[k] = rank == 0 ? dmalloc<uint8_t>((size_t)world * cap) : nullptr;
  }
  hipEvent_t raw_ready[2], raw_free[2];
  for (int b = 0; b < 2; b++) {
    CK(hipEventCreateWithFlags(&raw_ready[b], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&raw_free[b], hipEventDisableTiming));
  }
  int32_t* d_rank_ids = dmalloc<int32_t>(world + 1);
  float* d_tune = dmalloc<float>(2 * (world + 1));
  if (!d_img || !d_kp[1] || !d_desc[1] || !d_descf || !d_payload[SLOTS - 1] || !d_tune || (rank == 0 && !d_recv[SLOTS - 1])) CK(1);
  const float thr0 = 10000.0f;  // cc:353
  CK(hipMemcpy(d_thr_state, &thr0, 4, hipMemcpyHostToDevice));
  for (int k = 0; k < blocks; k++)  // this rank's block of every step
    CK(hipMemcpy(d_img + (size_t)k * 2 * B * img_bytes, S->frames + ((size_t)k * world + rank) * B * 2 * img_bytes,
                 (size_t)2 * B * img_bytes, hipMemcpyHostToDevice));
  // static schedule of the temporal pairs (distributed.temporal_pair_sets) for (step parity, first step), on the device
  int32_t *d_qset[2][2], *d_tset[2][2];
  for (int parity = 0; parity < 2; parity++)
    for (int first = 0; first < 2; first++) {
      std::vector<int32_t> q, t;
      auto region = [&](int par) { return 2 * B + par * world * Wn; };
      for (int i = 0; i < B; i++)
        for (int w = Wn; w > 0; w--) {
          const int past = i - w;
          int qs;
          if (past >= 0) qs = 2 * past;
          else if (rank > 0) qs = region(parity) + (rank - 1) * Wn + (Wn + past);
          else if (first) qs = empty_set;
          else qs = region(1 - parity) + (world - 1) * Wn + (Wn + past);
          q.push_back(qs);
          t.push_back(2 * i);
        }
      d_qset[parity][first] = dmalloc<int32_t>(std::max(NP, 1));
      d_tset[parity][first] = dmalloc<int32_t>(std::max(NP, 1));
      if (NP > 0) {
        CK(hipMemcpy(d_qset[parity][first], q.data(), q.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tset[parity][first], t.data(), t.size() * 4, hipMemcpyHostToDevice));
      }
    }
  CK(vsf_set_pipeline(ctx, 1));  // every step's input is complete in HBM: the next step's pyramid beside this step's later stages

  // ---- handshake: the ranks that really take part, through the backend that carries the step's exchanges ----
  at_stage(rank, "handshake (first vsf_allgather_dev)");
  {
    const int32_t me = rank;
    CK(hipMemcpy(d_rank_ids + world, &me, 4, hipMemcpyHostToDevice));
    CK(vsf_allgather_dev(tctx, comm, d_rank_ids + world, d_rank_ids, 4));
    CK(vsf_sync(tctx));
    if (rank == 0) {
      S->ranks_seen.resize(world);
      CK(hipMemcpy(S->ranks_seen.data(), d_rank_ids, 4 * world, hipMemcpyDeviceToHost));
      CK(vsf_comm_info(comm, nullptr, nullptr, &S->rccl_version));
    }
  }
  // (which batches have two FAST forms at all is the library's knowledge: its tune call reports 0 / 0 for one that has not)
  at_stage(rank, "tune");
  bool fast_eligible = false;
  int fast_form = 0;
  if (!S->keep) {
    float ms[2] = {0.f, 0.f};
    CK(vsf_tune_fast_resident(ctx, d_img, 2 * B, img_bytes, W, d_kp[0], d_desc[0], d_counts[0], 1, &ms[0], &ms[1]));
    fast_eligible = ms[0] > 0.f && ms[1] > 0.f;
  }

  int next_gather = 0;  // first step whose payload has not been handed to the gather yet
  auto issue_gather = [&](int step) {  // called with the tail stream's work of `step` queued; its sizes reach the host first
    const int k = step % SLOTS;
    CK(hipEventSynchronize(ev_sizes[k]));
    size_t nbytes = 0;
    for (int r = 0; r < world; r++) nbytes = std::max(nbytes, (size_t)h_sizes[k][r]);
    nbytes = std::min((nbytes + 15) & ~(size_t)15, cap);
    CK(vsf_gather_payload_dev(tctx, comm, d_payload[k], nbytes, d_recv[k], cap, 0));
    if (rank == 0 && S->keep) {
      CK(vsf_sync(tctx));
      for (int r = 0; r < world; r++) {
        std::vector<uint8_t> pl((size_t)h_sizes[k][r]);
        CK(hipMemcpy(pl.data(), d_recv[k] + (size_t)r * cap, pl.size(), hipMemcpyDeviceToHost));
        S->payloads.push_back(std::move(pl));
      }
    }
    next_gather = step + 1;
  };

  auto run_step = [&](int s) {
    const int parity = s & 1, b = s & 1, slot = s % SLOTS;
    const uint8_t* img = d_img + (size_t)(s % blocks) * 2 * B * img_bytes;
    // 1: extract(L), extract(R), GetMatches (cc:411-416) on the main stream, into raw buffer b
    if (s >= 2) CK(hipStreamWaitEvent(s_main, raw_free[b], 0));
    CK(vsf_stereo_batch_dev(ctx, img, B, img_bytes, W, d_kp[b], d_desc[b], d_counts[b], d_matches[b], d_nmatches[b]));
    CK(hipEventRecord(raw_ready[b], s_main));
    // 2-5: RemoveAmbigStereo with the threshold chain over ALL ranks' frames (cc:353-398), on the tail stream
    CK(hipStreamWaitEvent(s_tail, raw_ready[b], 0));
    CK(vsf_stereo_residuals_batch_dev(tctx, d_kp[b], d_matches[b], d_nmatches[b], B, F, d_means));
    CK(vsf_allgather_dev(tctx, comm, d_means, d_means_all, (size_t)B * 4));
    CK(vsf_stereo_thresholds_dev(tctx, d_means_all, world * B, d_thr_state, d_thr_all));
    CK(vsf_stereo_filter_batch_dev(tctx, d_kp[b], d_desc[b], d_matches[b], d_nmatches[b], B, d_thr_all + (size_t)rank * B, d_kpf,
                                   d_descf, d_countsf));
    CK(hipEventRecord(raw_free[b], s_tail));  // nothing below reads the raw outputs
    if (Wn > 0) {
      // 6: every rank's last `window` filtered LEFT frames: the temporal predecessors of the next rank's first frames
      for (int j = 0; j < Wn; j++) {
        const int set = 2 * (B - Wn + j);
        CK(hipMemcpyAsync(d_tail_desc + (size_t)j * K * 32, d_descf + (size_t)set * K * 32, K * 32, hipMemcpyDeviceToDevice, s_tail));
        CK(hipMemcpyAsync(d_tail_counts + j, d_countsf + set, 4, hipMemcpyDeviceToDevice, s_tail));
      }
      const int r0 = 2 * B + parity * world * Wn;
      CK(vsf_allgather_dev(tctx, comm, d_tail_desc, d_descf + (size_t)r0 * K * 32, (size_t)Wn * K * 32));
      CK(vsf_allgather_dev(tctx, comm, d_tail_counts, d_countsf + r0, (size_t)Wn * 4));
      // 7: GetFeatureMatches(past, current) (cc:424-434)
      CK(vsf_feature_matches_batch_dev(tctx, d_descf, d_countsf, K * 32, d_qset[parity][s == 0], d_tset[parity][s == 0], NP, 0.3f,
                                       d_pairs, d_npairs));
    }
    // 8-9: Calculate3DPoints + UndistortFeaturePoints (cc:437-443), the compact payload
    CK(vsf_vision_features_batch_dev(tctx, &calib, d_kpf, d_descf, d_countsf, B, d_feat, d_nfeat, nullptr));
    CK(vsf_pack_outputs_dev(tctx, d_feat, d_nfeat, B, d_pairs, d_npairs, NP, d_payload[slot], cap));
    // 10: payload sizes to every rank and on to pinned host memory; the sized gather follows ONE STEP LATER, when the sizes
    // have arrived without stalling this thread
    CK(vsf_allgather_dev(tctx, comm, d_payload[slot] + 12, d_sizes[slot], 4));
    CK(hipMemcpyAsync(h_sizes[slot], d_sizes[slot], 4 * world, hipMemcpyDeviceToHost, s_tail));
    CK(hipEventRecord(ev_sizes[slot], s_tail));
    while (next_gather < s) issue_gather(next_gather);
  };
  int step = 0;  // global step counter (payload slots, raw buffers and the temporal window follow it)
  auto drain = [&]() {
    while (next_gather < step) issue_gather(next_gather);
    CK(vsf_sync(ctx));
    CK(vsf_sync(tctx));
  };
  // ---- the one measured launch choice, on whole steps (bench.py's ShardedStereoFrontend.tune): per FAST form one untimed
  // step and four timed ones on the rotating batches; the two times go through ONE all-gather, the maximum over the ranks
  // decides, every rank sets the same form ----
  if (!S->keep && fast_eligible) {
    float ms[2] = {0.f, 0.f};
    for (int form = 0; form < 2; form++) {
      CK(vsf_set_fast_resident(ctx, form ? 3 : 0));
      run_step(step++);
      drain();
      const Clock::time_point a = Clock::now();
      for (int k = 0; k < 4; k++) run_step(step++);
      drain();
      ms[form] = (float)(1e3 * std::chrono::duration<double>(Clock::now() - a).count() / 4);
    }
    CK(hipMemcpy(d_tune + 2 * world, ms, 8, hipMemcpyHostToDevice));
    CK(vsf_allgather_dev(tctx, comm, d_tune + 2 * world, d_tune, 8));
    CK(vsf_sync(tctx));
    std::vector<float> every(2 * world);
    CK(hipMemcpy(every.data(), d_tune, 8 * world, hipMemcpyDeviceToHost));
    float g = 0.f, r = 0.f;
    for (int k = 0; k < world; k++) g = std::max(g, every[2 * k]), r = std::max(r, every[2 * k + 1]);
    fast_form = (r > 0.f && r < g) ? 3 : 0;
    CK(vsf_set_fast_resident(ctx, fast_form));
    if (rank == 0) S->fast_form = fast_form, S->tune_ms[0] = g, S->tune_ms[1] = r;
  }
  at_stage(rank, "warm-up steps");
  const int warm = S->keep ? 0 : 3;
  for (int k = 0; k < warm; k++) run_step(step++);
  drain();
  at_stage(rank, "barrier before the timed steps");
  S->bar->wait();
  at_stage(rank, "timed steps");
  const Clock::time_point t0 = Clock::now();
  for (int k = 0; k < S->steps; k++) run_step(step++);
  drain();
  S->rank_seconds[rank] = std::chrono::duration<double>(Clock::now() - t0).count();
  at_stage(rank, "barrier after the timed steps");
  S->bar->wait();
  at_stage(rank, "teardown");
  vsf_comm_destroy(comm);
  vsf_destroy(tctx);
  vsf_destroy(ctx);
}

int main(int argc, char** argv) {
  if (argc < 9) {
    std::fprintf(stderr, "usage: %s frames.raw W H NFRAMES nfeatures frames_per_rank window steps [payloads.bin] [gpus]\n", argv[0]);
    return 2;
  }
  Shared S;
  S.W = std::atoi(argv[2]);
  S.H = std::atoi(argv[3]);
  S.nframes = std::atoi(argv[4]);
  S.nf = std::atoi(argv[5]);
  S.B = std::atoi(argv[6]);
  S.window = std::atoi(argv[7]);
  S.steps = std::atoi(argv[8]);
  const std::string out = argc > 9 ? argv[9] : "";
  int ngpu = 0;
  if (hipGetDeviceCount(&ngpu) != hipSuccess || ngpu < 1) {
    std::fprintf(stderr, "no GPU\n");
    return 2;
  }
  S.world = argc > 10 ? std::min(std::atoi(argv[10]), ngpu) : ngpu;
  S.keep = !out.empty();
  if (S.window > S.B || S.nframes < S.world * S.B || S.steps < 1) {
    std::fprintf(stderr, "need window <= frames_per_rank and at least world * frames_per_rank frames\n");
    return 2;
  }
  std::vector<uint8_t> raw((size_t)S.nframes * 2 * S.W * S.H);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(raw.data(), 1, raw.size(), f) != raw.size()) {
    std::fprintf(stderr, "cannot read %zu bytes from %s\n", raw.size(), argv[1]);
    return 2;
  }
  std::fclose(f);
  S.frames = raw.data();
  if (vsf_comm_unique_id(S.id) != VSF_OK) {
    std::fprintf(stderr, "no RCCL (vsf_comm_unique_id)\n");
    return 3;
  }
  Barrier bar(S.world);
  S.bar = &bar;
  S.rank_seconds.assign(S.world, 0.0);
  std::atomic<bool> failed(false);
  const char* stall_env = std::getenv("VSF_SHARDED_STALL_S");
  const double stall_s = stall_env && std::atof(stall_env) > 0 ? std::atof(stall_env) : 300.0;
  std::thread dog(watchdog_main, S.world, stall_s);
  std::vector<std::thread> th;
  for (int r = 0; r < S.world; r++) th.emplace_back(rank_main, r, &S, &failed);
  for (auto& t : th) t.join();
  g_finished.store(true);
  dog.join();
  if (failed.load()) return 1;
  double slowest = 0;
  for (double v : S.rank_seconds) slowest = std::max(slowest, v);
  if (S.keep) {
    FILE* o = std::fopen(out.c_str(), "wb");
    if (!o) return 2;
    for (const auto& pl : S.payloads) {
      const uint32_t n = (uint32_t)pl.size();
      std::fwrite(&n, 4, 1, o);
      std::fwrite(pl.data(), 1, pl.size(), o);
    }
    std::fclose(o);
  }
  std::printf("{\"what\": \"sharded hot path through the C ABI, one thread per GPU (tools/time_sharded.cc)\", \"n_gpus\": %d, "
              "\"frames_per_rank\": %d, \"window\": %d, \"steps\": %d, \"stereo_frames_per_s\": %.1f, \"ms_per_step\": %.3f, "
              "\"fast_resident\": %d, \"tune_ms_grid\": %.3f, \"tune_ms_resident\": %.3f, \"rccl_version\": %d, \"ranks_seen\": [",
              S.world, S.B, S.window, S.steps, (double)S.world * S.B * S.steps / slowest, 1e3 * slowest / S.steps, S.fast_form,
              S.tune_ms[0], S.tune_ms[1], S.rccl_version);
  for (int r = 0; r < (int)S.ranks_seen.size(); r++) std::printf("%s%d", r ? ", " : "", S.ranks_seen[r]);
  std::printf("], \"note\": \"frames resident in HBM, the step's tail and every exchange on a second stream beside the next "
              "step's extraction, the FAST form measured and agreed over the ranks before the timed steps: the composition of "
              "bench.py, through include/vsf.h alone\"}\n");
  return 0;
}
