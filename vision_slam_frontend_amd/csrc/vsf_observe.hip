// vsf_observe.hip -- Frontend::ObserveImage (slam_frontend.cc:400-472) as a QUEUE of stereo frames.
//
// vsf_observe_submit copies a frame's two images into pinned staging and returns a ticket; frames that wait are coalesced
// into ONE batched extraction + ONE batched tail:
//   upload      one copy command for the batch's images (a copy command costs ~180 us whatever it carries up to 10 MB)
//   extraction  ExtractFeatures x 2 and the stereo GetMatches of every frame of the batch (cc:411-416): the batched kernels
//   tail        RemoveAmbigStereo with the threshold chain in frame order (cc:417, 353, 392-394), every GetFeatureMatches of
//               the temporal loop (cc:424-434) and the right -> left match of Calculate3DPoints (cc:129-132) as ONE matcher
//               launch + one sort launch over the batch's pair list, the VisionFeature records (cc:437-443), one compact
//               result per frame written straight into pinned host memory.
// A frame's launch-bound chain of ~30 small kernels costs the host 5.4 us per launch and the GPU a launch-to-launch latency
// per kernel whatever the batch holds, so a batch of n frames costs little more than a batch of one until the chip is full.
// When a batch leaves (batch_to_launch): when a whole batch waits; when `min_batch` frames wait (a whole batch by default while the queue holds two, else half the queue) and
// fewer than `in_flight` batches are on the GPU; when the GPU is idle and no frame has arrived for 100 us; or when somebody
// collects a frame that still waits (a lone frame: the synchronous call is a batch of one).  While the GPU is busy, frames
// accumulate -- the batch size follows the caller's rate by itself.
// The kept frames' filtered descriptors live in a ring of descriptor sets in HBM (frame g in set g % ring); the pair list
// of a batch addresses them by set index, so a frame matches against frames of earlier batches and of its own alike.
// Results are those of one frame at a time, bit for bit (tests/test_gpu_observe.py).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "vsf_ctx.h"

using namespace vsfi;

// One row-wise copy of an image into the staging ring (rows at the device pitch): 5-12 us per 640x480 image on one core,
// depending on the host (its memory, its neighbours).
static void stage_image(uint8_t* dst, size_t dst_pitch, const uint8_t* src, size_t src_pitch, size_t width, int rows) {
  if (dst_pitch == src_pitch) {
    std::memcpy(dst, src, (size_t)(rows - 1) * src_pitch + width);
  } else {
    for (int y = 0; y < rows; y++) std::memcpy(dst + (size_t)y * dst_pitch, src + (size_t)y * src_pitch, width);
  }
}

// The staging copy is what a queued frame costs its caller once the launches have a thread of their own: 13 us per frame on
// one box, 23 us on another (the same run: 31.5 k and 28.9 k frames/s -- on the second the caller never waits for the GPU).
// While frames stream in (the previous one is still in the queue) a helper thread takes the right image: it spins for a job
// while it is hot and goes to sleep 300 us after the last one, so a caller that submits and collects frame by frame never
// meets it (a wake-up costs more than the copy saves).  It touches host memory only: no HIP call, no context state.
struct vsf_ctx::ObserveCopyHelper {
  struct Job {
    uint8_t* dst;
    const uint8_t* src;
    size_t dst_pitch, src_pitch, width;
    int rows;
  };
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<int> state{0};  // 0 no job, 1 job posted, 2 job done
  std::atomic<bool> hot{false}, stop{false};
  bool wake = false;
  Job job{};
  ObserveCopyHelper() { th = std::thread([this] { run(); }); }
  ~ObserveCopyHelper() {
    {
      std::lock_guard<std::mutex> g(m);
      stop.store(true);
    }
    cv.notify_all();
    th.join();
  }
  void run() {
    using Clock = std::chrono::steady_clock;
    while (!stop.load(std::memory_order_acquire)) {
      hot.store(true, std::memory_order_release);
      Clock::time_point last = Clock::now();
      while (!stop.load(std::memory_order_relaxed)) {
        if (state.load(std::memory_order_acquire) == 1) {
          stage_image(job.dst, job.dst_pitch, job.src, job.src_pitch, job.width, job.rows);
          state.store(2, std::memory_order_release);
          last = Clock::now();
        } else {
          __builtin_ia32_pause();
          if (Clock::now() - last > std::chrono::microseconds(300)) break;
        }
      }
      hot.store(false, std::memory_order_release);
      std::unique_lock<std::mutex> g(m);
      // (a job posted between the last look and `hot = false` is still served: the wait's predicate sees it)
      cv.wait(g, [this] { return stop.load() || wake || state.load(std::memory_order_acquire) == 1; });
      wake = false;
    }
  }
  // true: the helper took `j` (wait() must follow); false: it sleeps -- woken for the frames behind this one -- and the
  // caller copies `j` itself.
  bool post(const Job& j) {
    if (!hot.load(std::memory_order_acquire)) {
      {
        std::lock_guard<std::mutex> g(m);
        wake = true;
      }
      cv.notify_one();
      return false;
    }
    job = j;
    state.store(1, std::memory_order_release);
    if (!hot.load(std::memory_order_acquire)) cv.notify_one();  // (it was on its way to sleep: the predicate serves the job)
    return true;
  }
  void wait() {
    while (state.load(std::memory_order_acquire) != 2) __builtin_ia32_pause();
    state.store(0, std::memory_order_relaxed);
  }
};

// Who launches.  A batch costs the host 0.1 ms (a lone frame) to 0.4 ms (the batched pyramid alone is 50-100 launches).
// By default the caller launches, between two submits (4-5 us per frame at 64-128 frames per batch).  With
// VSF_OPT_OBSERVE_THREAD a queue of depth >= 4 has a LAUNCHER thread instead: the caller stages frames and the thread sends
// whatever the policy releases, polling the GPU's state while frames wait (measured slower wherever depth == batch size, the
// same elsewhere: off by default).  The caller still launches by itself where waiting for the thread would cost more than it
// saves: when it collects a frame that still waits (the synchronous call: submit, collect), and for every other entry
// point of the context, which first sends everything that waits (VsfErrorScope -> vsf_ctx_enter), so that nothing else ever
// runs beside the thread.  `launching` is the baton: whoever holds it is alone inside launch_batch.
struct vsf_ctx::ObserveLauncher {
  std::mutex mu;  // guards next_ticket / next_launch / next_collect, launching, stop, status
  std::condition_variable cv_thread, cv_caller;
  bool launching = false, stop = false, has_thread = false;
  vsf_status status = VSF_OK;  // first failure of a launch: sticky until the queue is rebuilt
  std::thread th;
};

struct vsf_ctx::ObserveBatchMeta {
  int32_t n_frames, n_pairs;
  // followed by (offsets in int32 words from the start of the block, fixed by the queue's sizes):
  //   q_set[max_pairs] | t_set[max_pairs] | best_percent[max_pairs] (float) | out_sets[2 bmax] | frames[bmax]
};

namespace {

struct MetaView {
  int32_t* q_set;
  int32_t* t_set;
  float* best_percent;
  int32_t* out_sets;
  VsfObserveFrame* frames;
};

size_t meta_bytes(int max_pairs, int bmax) {
  return 16 + (size_t)max_pairs * 12 + (size_t)bmax * 8 + (size_t)bmax * sizeof(VsfObserveFrame);
}

MetaView meta_view(vsf_ctx::ObserveBatchMeta* m, int max_pairs, int bmax) {
  uint8_t* b = reinterpret_cast<uint8_t*>(m) + 16;
  MetaView v;
  v.q_set = reinterpret_cast<int32_t*>(b);
  v.t_set = v.q_set + max_pairs;
  v.best_percent = reinterpret_cast<float*>(v.t_set + max_pairs);
  v.out_sets = reinterpret_cast<int32_t*>(v.best_percent + max_pairs);
  v.frames = reinterpret_cast<VsfObserveFrame*>(v.out_sets + 2 * bmax);
  return v;
}

inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool same_calibration(const vsf_calibration& a, const vsf_calibration& b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

}  // namespace

namespace vsfi {

void stop_observe_threads(vsf_ctx* ctx) {
  vsf_ctx::Observe& o = ctx->ob;
  if (o.launcher && o.launcher->has_thread) {
    {
      std::lock_guard<std::mutex> g(o.launcher->mu);
      o.launcher->stop = true;
    }
    o.launcher->cv_thread.notify_all();
    o.launcher->th.join();
    o.launcher->has_thread = false;
  }
  delete o.copy_helper;
  o.copy_helper = nullptr;
}

void free_observe(vsf_ctx* ctx) {
  vsf_ctx::Observe& o = ctx->ob;
  stop_observe_threads(ctx);
  delete o.launcher;
  hipFree(o.sets);
  hipFree(o.set_counts);
  hipFree(o.residual);
  hipFree(o.floats);
  hipFree(o.kpf);
  hipFree(o.ints);
  hipFree(o.ex_idx2);
  hipFree(o.ex_dist2);
  hipFree(o.t_idx2);
  hipFree(o.t_dist2);
  hipFree(o.t_matches);
  hipFree(o.t_nmatches);
  hipFree(o.t_sortkeys);
  hipFree(o.pairs);
  hipFree(o.npairs);
  hipFree(o.features);
  if (o.h_img) hipHostFree(o.h_img);
  if (o.h_out) hipHostFree(o.h_out);
  for (vsf_ctx::ObserveBatch& b : o.batch) {
    hipFree(b.d_img);
    hipFree(b.kp_raw);
    hipFree(b.desc_raw);
    hipFree(b.counts_raw);
    hipFree(b.matches);
    hipFree(b.nmatches);
    hipFree(b.status);
    if (b.h_meta) hipHostFree(b.h_meta);
    if (b.ev_uploaded) hipEventDestroy(b.ev_uploaded);
    if (b.ev_extracted) hipEventDestroy(b.ev_extracted);
    if (b.ev_done) hipEventDestroy(b.ev_done);
  }
  if (o.copy_stream) hipStreamDestroy(o.copy_stream);
  if (o.tail_stream) hipStreamDestroy(o.tail_stream);
  o = vsf_ctx::Observe();
}

}  // namespace vsfi

namespace {

void launcher_thread(vsf_ctx* ctx);

vsf_status ensure_observe(vsf_ctx* ctx, int frame_life) {
  vsf_ctx::Observe& o = ctx->ob;
  if (o.ready && o.frame_life == frame_life) return VSF_OK;
  sync_all_streams(ctx);
  float thr_state = 10000.0f;  // cc:353
  if (o.floats) VSF_HIP(hipMemcpy(&thr_state, o.floats + 2 * o.bmax + 1, sizeof(float), hipMemcpyDeviceToHost));
  free_observe(ctx);
  const size_t K = (size_t)ctx->p.max_keypoints;
  const int frames_cap = std::max(1, ctx->p.max_images / 2);  // the extraction's own buffers hold max_images images
  o.depth = ctx->ob_depth > 0 ? ctx->ob_depth : frames_cap;
  o.bmax = std::min(o.depth, frames_cap);
  o.frame_life = frame_life;
  o.ring = frame_life + o.bmax;
  o.max_pairs = o.bmax * (frame_life + 1);
  const size_t B = (size_t)o.bmax, P = (size_t)o.max_pairs, S = (size_t)o.ring + B;
  VSF_HIP(hipMalloc((void**)&o.sets, S * K * VSF_DESC_BYTES));
  VSF_HIP(hipMalloc((void**)&o.set_counts, S * sizeof(int32_t)));
  VSF_HIP(hipMemset(o.set_counts, 0, S * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.residual, B * K * sizeof(float)));
  VSF_HIP(hipMalloc((void**)&o.floats, (2 * B + 2) * sizeof(float)));
  VSF_HIP(hipMemset(o.floats, 0, (2 * B + 2) * sizeof(float)));
  VSF_HIP(hipMemcpy(o.floats + 2 * B + 1, &thr_state, sizeof(float), hipMemcpyHostToDevice));
  VSF_HIP(hipMalloc((void**)&o.kpf, 2 * B * K * sizeof(vsf_keypoint)));
  VSF_HIP(hipMalloc((void**)&o.ints, 4 * B * sizeof(int32_t)));
  VSF_HIP(hipMemset(o.ints, 0, 4 * B * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.ex_idx2, B * K * 2 * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.ex_dist2, B * K * 2 * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.t_idx2, P * K * 2 * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.t_dist2, P * K * 2 * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.t_matches, P * K * sizeof(vsf_dmatch)));
  VSF_HIP(hipMalloc((void**)&o.t_nmatches, P * sizeof(int32_t)));
  VSF_HIP(hipMalloc(&o.t_sortkeys, P * K * 8));
  VSF_HIP(hipMalloc((void**)&o.pairs, P * K * 2 * sizeof(uint64_t)));
  VSF_HIP(hipMalloc((void**)&o.npairs, P * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.features, B * K * sizeof(vsf_vision_feature)));
  o.out_cap = vsf_observe_capacity(ctx, frame_life);
  o.out_stride = (o.out_cap + 255) & ~(size_t)255;
  VSF_HIP(hipHostMalloc((void**)&o.h_img, (size_t)o.depth * 2 * ctx->st_img_stride, hipHostMallocMapped));
  VSF_HIP(hipHostMalloc((void**)&o.h_out, (size_t)o.depth * o.out_stride, hipHostMallocMapped));
  for (vsf_ctx::ObserveBatch& b : o.batch) {
    VSF_HIP(hipMalloc((void**)&b.d_img, 2 * B * ctx->st_img_stride));
    VSF_HIP(hipMalloc((void**)&b.kp_raw, 2 * B * K * sizeof(vsf_keypoint)));
    VSF_HIP(hipMalloc((void**)&b.desc_raw, 2 * B * K * VSF_DESC_BYTES));
    VSF_HIP(hipMalloc((void**)&b.counts_raw, 2 * B * sizeof(int32_t)));
    VSF_HIP(hipMalloc((void**)&b.matches, B * K * sizeof(vsf_dmatch)));
    VSF_HIP(hipMalloc((void**)&b.nmatches, B * sizeof(int32_t)));
    VSF_HIP(hipMalloc((void**)&b.status, 2 * B * sizeof(int32_t)));
    VSF_HIP(hipMemset(b.status, 0, 2 * B * sizeof(int32_t)));
    VSF_HIP(hipHostMalloc((void**)&b.h_meta, meta_bytes(o.max_pairs, o.bmax), hipHostMallocMapped));
    std::memset(b.h_meta, 0, meta_bytes(o.max_pairs, o.bmax));
    VSF_HIP(hipEventCreateWithFlags(&b.ev_uploaded, hipEventDisableTiming));
    VSF_HIP(hipEventCreateWithFlags(&b.ev_extracted, hipEventDisableTiming));
    VSF_HIP(hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming));
  }
  o.frames.assign((size_t)o.depth, vsf_ctx::ObserveFrame());
  // The tail rides a high-priority stream: it is short, latency-bound and what the host waits for; streams of different
  // priorities never share a hardware queue, so it runs beside the next batch's extraction instead of taking turns with it.
  int prio_lo = 0, prio_hi = 0;
  VSF_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  VSF_HIP(hipStreamCreateWithPriority(&o.tail_stream, hipStreamNonBlocking, prio_hi));
  // ... and the uploads the LOWEST: with the default priority the copy stream may land on the hardware queue of the
  // context's stream (HIP hands its few queues out round-robin) and the next batch's upload then waits for this batch's
  // extraction instead of running beside it -- which it did or did not from one context to the next (17 k or 27 k frames/s)
  VSF_HIP(hipStreamCreateWithPriority(&o.copy_stream, hipStreamNonBlocking, prio_lo));
  VSF_HIP(hipDeviceSynchronize());
  o.launcher = new (std::nothrow) vsf_ctx::ObserveLauncher();
  if (!o.launcher) return VSF_ERR_INVALID_ARG;
  o.ready = true;
  // the two host threads of a deep queue: VSF_OPT_OBSERVE_THREAD (without it the caller launches everything) and
  // VSF_OPT_OBSERVE_COPY_THREAD (without it the caller stages both images)
  if (o.depth >= 4 && ctx->tuning.observe_copy_thread) o.copy_helper = new (std::nothrow) vsf_ctx::ObserveCopyHelper();
  if (o.depth >= 4 && ctx->tuning.observe_thread) {
    o.launcher->has_thread = true;
    o.launcher->th = std::thread(launcher_thread, ctx);
  }
  return VSF_OK;
}

int batches_on_gpu(vsf_ctx* ctx) {  // launched and not finished (a query costs 0.1 us)
  int n = 0;
  for (vsf_ctx::ObserveBatch& b : ctx->ob.batch)
    if (b.used && hipEventQuery(b.ev_done) != hipSuccess) n++;
  (void)hipGetLastError();  // (hipErrorNotReady is not an error)
  return n;
}

// Queues frames [t0, t0 + n) -- they wait in consecutive staging slots -- as one batch.
vsf_status launch_batch(vsf_ctx* ctx, int64_t t0, int n, bool solo) {
  vsf_ctx::Observe& o = ctx->ob;
  const int64_t t_begin = now_ns();
  const int bi = (int)(o.batches % vsf_ctx::kObserveBatchSlots);
  vsf_ctx::ObserveBatch& b = o.batch[bi];
  // The slot's previous batch must have left the GPU: its kernels read the pinned parameter block that is rewritten below
  // (with `in_flight` batches on the GPU and four slots it has, long ago).
  if (b.used) {
    if (hipEventQuery(b.ev_done) != hipSuccess) o.stat_slot_waits++;
    (void)hipGetLastError();
    VSF_HIP(hipEventSynchronize(b.ev_done));
  }
  const size_t K = (size_t)ctx->p.max_keypoints;
  const int Kc = (int)K, life = o.frame_life;
  const vsf_ctx::ObserveFrame& f0 = o.frames[(size_t)(t0 % o.depth)];
  // solo: a lone frame with nothing else on the GPU runs on ONE stream from upload to result (no event hops in its chain);
  // otherwise copy, extraction and tail have a stream each, so that the next batch's upload and extraction run beside
  // this one's tail.
  hipStream_t s_copy = solo ? ctx->stream : o.copy_stream, s_ex = ctx->stream, s_tail = solo ? ctx->stream : o.tail_stream;
  // ---- upload: ONE copy command (two when the frames wrap around the staging ring) ----
  {
    const int slot0 = (int)(t0 % o.depth), first = std::min(n, o.depth - slot0);
    const size_t frame_bytes = 2 * ctx->st_img_stride;
    VSF_HIP(hipMemcpyAsync(b.d_img, o.h_img + (size_t)slot0 * frame_bytes, (size_t)first * frame_bytes, hipMemcpyHostToDevice,
                           s_copy));
    if (first < n)
      VSF_HIP(hipMemcpyAsync(b.d_img + (size_t)first * frame_bytes, o.h_img, (size_t)(n - first) * frame_bytes,
                             hipMemcpyHostToDevice, s_copy));
    if (s_copy != s_ex) {
      VSF_HIP(hipEventRecord(b.ev_uploaded, s_copy));
      VSF_HIP(hipStreamWaitEvent(s_ex, b.ev_uploaded, 0));
    }
  }
  // ---- the batch's parameters, in pinned memory the kernels read directly ----
  const MetaView M = meta_view(b.h_meta, o.max_pairs, o.bmax);
  int n_pairs = n, max_pairs_per_frame = 1;
  for (int f = 0; f < n; f++) {
    const int64_t g = t0 + f;  // frames since the queue was built: frame g lives in set g % ring
    const int n_past = (int)std::min<int64_t>(g, life), left_set = (int)(g % o.ring), right_set = o.ring + f;
    M.out_sets[2 * f] = left_set;
    M.out_sets[2 * f + 1] = right_set;
    M.q_set[f] = right_set;  // Calculate3DPoints: GetFeatureMatches(right, left) with best_percent_ 1.0 (cc:129-132)
    M.t_set[f] = left_set;
    M.best_percent[f] = 1.0f;
    VsfObserveFrame& fm = M.frames[f];
    fm.left_set = left_set;
    fm.n_past = n_past;
    fm.tp0 = n_pairs;
    fm.out_slot = (int)(g % o.depth);
    for (int p = 0; p < n_past; p++) {  // oldest kept frame first: the order frame_list_ is walked in (cc:424)
      M.q_set[n_pairs] = (int)((g - n_past + p) % o.ring);
      M.t_set[n_pairs] = left_set;
      M.best_percent[n_pairs] = f0.best_percent;
      n_pairs++;
    }
    max_pairs_per_frame = std::max(max_pairs_per_frame, n_past + 1);
  }
  b.h_meta->n_frames = n;
  b.h_meta->n_pairs = n_pairs;
  // ---- ExtractFeatures x 2 + GetMatches of every frame (cc:411-416) ----
  const VsfImages im{b.d_img, ctx->st_img_stride, ctx->st_img_pitch, 2 * n};
  // (Measured and left out: the pyramid of a batch on a side stream beside the previous batch's later stages, the
  // cross-call pipelining of vsf_set_pipeline -- 27.5 -> 23.4 k frames/s at 64 frames per batch, 31.9 -> 27.5 k at 128: the
  // single chain of 49 dependent launches is slower than the two chains + image-major kernel it replaces and takes the
  // vector ALU from the stages it runs beside.)
  extract_on(ctx, s_ex, im, 0, 2 * n, b.kp_raw, b.desc_raw, b.counts_raw, false, nullptr, b.status, 1);
  ctx->last_images = im;
  ctx->last_valid = true;
  match_on(ctx, s_ex, b.desc_raw, b.counts_raw, K * VSF_DESC_BYTES, nullptr, nullptr, 0, n, o.ex_idx2, o.ex_dist2, b.matches,
           b.nmatches, b.status);
  if (s_tail != s_ex) {
    VSF_HIP(hipEventRecord(b.ev_extracted, s_ex));
    VSF_HIP(hipStreamWaitEvent(s_tail, b.ev_extracted, 0));
  }
  // the tails run in frame order: behind the previous batch's, whatever stream that ran on
  if (o.last_batch >= 0 && o.batch[o.last_batch].done_stream != s_tail)
    VSF_HIP(hipStreamWaitEvent(s_tail, o.batch[o.last_batch].ev_done, 0));
  // ---- RemoveAmbigStereo (cc:417): residuals, the threshold chain in frame order, the rebuilt frames ----
  float *means = o.floats, *thr = o.floats + o.bmax, *thr_state = o.floats + 2 * o.bmax + 1;
  int32_t *counts_f = o.ints, *nfeat = o.ints + 2 * o.bmax, *npoints = o.ints + 3 * o.bmax;
  {
    StageTimer t(ctx, s_tail, VSF_STAGE_TAIL, 3);
    // (a lone frame: the three steps in one launch)
    if (n != 1 || !vsf_launch_stereo_one_frame(b.kp_raw, b.desc_raw, b.matches, b.nmatches, Kc, f0.calib.fundamental,
                                               ctx->p.residual_order, means, thr, thr_state, o.kpf, o.sets, counts_f,
                                               M.out_sets, o.set_counts, s_tail)) {
      vsf_launch_stereo_residuals(b.kp_raw, b.matches, b.nmatches, n, Kc, nullptr, f0.calib.fundamental, ctx->p.residual_order,
                                  o.residual, means, s_tail);
      vsf_launch_stereo_thresholds(means, n, thr_state, thr, s_tail);
      vsf_launch_stereo_filter_only(b.kp_raw, b.desc_raw, b.matches, b.nmatches, n, Kc, o.residual, thr, o.kpf, o.sets,
                                    counts_f, s_tail, M.out_sets, o.set_counts);
    }
  }
  // ---- every GetFeatureMatches of the batch: one matcher launch, one sort launch (per-pair best_percent) ----
  {
    StageTimer t(ctx, s_tail, VSF_STAGE_KNN2, 1);
    // (rows_hint: what the caller's last collected frames held -- a filtered frame is a few hundred rows of the capacity)
    vsf_launch_knn2(o.sets, o.set_counts, K * VSF_DESC_BYTES, M.q_set, M.t_set, n_pairs, Kc, o.t_idx2, o.t_dist2, s_tail,
                    o.rows_hint);
  }
  {
    StageTimer t(ctx, s_tail, VSF_STAGE_RATIO, 1);
    vsf_launch_ratio_compact(o.set_counts, M.q_set, M.t_set, n_pairs, Kc, o.t_idx2, o.t_dist2, ctx->p.ratio_num,
                             ctx->p.ratio_shift, o.t_matches, o.t_nmatches, b.status, s_tail);
  }
  {
    StageTimer t(ctx, s_tail, VSF_STAGE_TAIL, 3);
    vsf_launch_sort_trim(o.t_matches, o.t_nmatches, n_pairs, Kc, f0.best_percent, M.best_percent, o.t_sortkeys, o.pairs,
                         o.npairs, s_tail, false, ctx->tuning.lds_limit);
    // ---- Calculate3DPoints + VisionFeature + UndistortFeaturePoints (cc:437-443): pairs [0, n) are the right -> left ones ----
    vsf_launch_vision_features(o.kpf, counts_f, o.pairs, o.npairs, n, Kc, f0.calib, o.features, nfeat, npoints, s_tail);
    // ---- one compact result per frame, into its slot of the pinned result ring ----
    VsfObserveArgs a;
    a.n_frames = n;
    a.max_rows = Kc;
    a.counts_raw = b.counts_raw;
    a.nmatches = b.nmatches;
    a.counts_f = counts_f;
    a.npoints = npoints;
    a.means = means;
    a.thr = thr;
    a.features = o.features;
    a.kp_f = o.kpf;
    a.desc_sets = o.sets;
    a.pairs = o.pairs;
    a.npairs = o.npairs;
    a.status = b.status;
    a.frames = M.frames;
    a.out = o.h_out;
    a.out_stride = o.out_stride;
    a.out_cap = (uint32_t)std::min<size_t>(o.out_cap, 0xFFFFFFF0u);
    vsf_launch_observe_pack(a, max_pairs_per_frame, s_tail);
  }
  VSF_HIP(hipEventRecord(b.ev_done, s_tail));
  b.used = true;
  b.done_stream = s_tail;
  for (int f = 0; f < n; f++) o.frames[(size_t)((t0 + f) % o.depth)].batch = bi;
  o.last_batch = bi;
  o.batches++;
  o.stat_frames += n;
  o.stat_max_batch = std::max<int64_t>(o.stat_max_batch, n);
  if (solo) o.stat_solo++;
  o.stat_launch_ns += now_ns() - t_begin;
  VSF_STICKY();
  return VSF_OK;
}

// How many of the waiting frames leave now (0: none).  mu held, nobody launching.
int batch_to_launch(vsf_ctx* ctx, bool force) {
  vsf_ctx::Observe& o = ctx->ob;
  const int pending = (int)(o.next_ticket - o.next_launch);
  if (pending <= 0) return 0;
  if (force || pending >= o.bmax) return std::min(pending, o.bmax);
  // An idle GPU takes whatever waits.  A busy one is in no hurry: frames wait for company because a batch costs ~50-100
  // launches whatever it carries (measured on the caller's thread: batches of 1-8 frames 14 k frames/s, of 32-64 frames
  // 27 k).  How much company: in steady state a batch leaves the moment `min_batch` frames wait, so min_batch IS the batch
  // size -- by default a whole batch when the queue is deep enough for the caller to fill the next one meanwhile (depth >= 2
  // batches), else half the queue, so that staging and the GPU still overlap (tools/exp/min_batch.sh: depth 64 / 32 per
  // batch 19.8 -> 25.0 k frames/s against half a batch, 128 / 64 27.9 -> 28.8 k, 256 / 128 32.0 -> 32.4 k; at depth =
  // batch size half the queue is what it was).
  // ... and "idle" must not be mistaken for "nobody is coming": while frames stream in (the last one arrived less than
  // 100 us ago) even an idle GPU waits for min_batch of them.  Without that a GPU that once ran dry keeps being fed batches of
  // a few frames, each gone before the next has gathered (measured: the same queue at 15 k or 32 k frames/s).
  const int busy = batches_on_gpu(ctx);
  const int min_batch = ctx->ob_min_batch > 0 ? std::min(ctx->ob_min_batch, o.bmax) : std::max(1, std::min(o.bmax, o.depth / 2));
  if (busy < ctx->ob_in_flight && pending >= min_batch) return pending;
  return (busy == 0 && now_ns() - o.last_submit_ns > 100000) ? pending : 0;
}

// One batch, by whoever holds the lock: takes the baton, launches outside the lock, publishes next_launch.
vsf_status launch_one(vsf_ctx* ctx, std::unique_lock<std::mutex>& lk, int n) {
  vsf_ctx::Observe& o = ctx->ob;
  vsf_ctx::ObserveLauncher& L = *o.launcher;
  const int64_t t0 = o.next_launch;
  const bool solo = n == 1 && batches_on_gpu(ctx) == 0;
  L.launching = true;
  lk.unlock();
  const vsf_status st = launch_batch(ctx, t0, n, solo);
  lk.lock();
  L.launching = false;
  if (st == VSF_OK)
    o.next_launch = t0 + n;
  else if (L.status == VSF_OK)
    L.status = st;
  L.cv_caller.notify_all();
  if (L.has_thread) L.cv_thread.notify_one();
  return st;
}

// The caller's side.  force: everything that waits leaves now (somebody collects one of them, the parameters change, or
// another entry point of the context is about to run); otherwise whatever the policy releases.  mu held on entry and exit.
vsf_status caller_pump(vsf_ctx* ctx, std::unique_lock<std::mutex>& lk, bool force) {
  vsf_ctx::Observe& o = ctx->ob;
  vsf_ctx::ObserveLauncher& L = *o.launcher;
  while (true) {
    if (L.launching) {  // the thread is at it
      if (!force) return VSF_OK;
      L.cv_caller.wait(lk);
      continue;
    }
    if (L.status != VSF_OK) return L.status;
    const int n = batch_to_launch(ctx, force);
    if (n == 0) return VSF_OK;
    if (force) o.stat_forced++;
    const vsf_status st = launch_one(ctx, lk, n);
    if (st != VSF_OK) return st;
  }
}

void launcher_thread(vsf_ctx* ctx) {
  vsf_ctx::Observe& o = ctx->ob;
  vsf_ctx::ObserveLauncher& L = *o.launcher;
  if (hipSetDevice(ctx->device) != hipSuccess) {
    std::lock_guard<std::mutex> g(L.mu);
    L.status = VSF_ERR_HIP;
    return;
  }
  std::unique_lock<std::mutex> lk(L.mu);
  while (!L.stop) {
    if (L.launching || L.status != VSF_OK || o.next_launch >= o.next_ticket) {
      L.cv_thread.wait(lk);  // (a submit into an empty queue, the end of a launch and stop notify)
      continue;
    }
    const int n = batch_to_launch(ctx, false);
    if (n == 0) {  // frames wait for company or for the GPU: its state changes without a notification
      L.cv_thread.wait_for(lk, std::chrono::microseconds(40));
      continue;
    }
    (void)launch_one(ctx, lk, n);
  }
}

}  // namespace

// Every entry point of the context except the queue's own comes through here (VsfErrorScope): what waits in the queue
// leaves first and the launcher thread is idle afterwards -- it only wakes for frames that wait.
void vsf_ctx_enter(vsf_ctx* ctx) {
  vsf_ctx::Observe& o = ctx->ob;
  if (!o.ready || !o.launcher || !o.launcher->has_thread) return;
  if (hipSetDevice(ctx->device) != hipSuccess) return;
  std::unique_lock<std::mutex> lk(o.launcher->mu);
  (void)caller_pump(ctx, lk, true);
  while (o.launcher->launching) o.launcher->cv_caller.wait(lk);
}

extern "C" {

size_t vsf_observe_capacity(const vsf_ctx* ctx, int frame_life) {
  if (!ctx || frame_life < 0 || frame_life + 1 > VSF_OBSERVE_MAX_PAIRS) return 0;
  const size_t K = (size_t)ctx->p.max_keypoints;
  return 64 + 4 * (size_t)((frame_life + 1 + 3) & ~3) + K * (28 + 28 + 32) + (size_t)(frame_life + 1) * K * 16;
}

vsf_status vsf_observe_configure(vsf_ctx* ctx, int depth, int min_batch, int in_flight) {
  VsfErrorScope scope_(ctx);  // (sends what waits; the launcher thread is idle afterwards)
  if (!ctx || depth < 0 || depth > 1024 || min_batch < 0 || in_flight < 0 || in_flight > vsf_ctx::kObserveBatchSlots - 1)
    return VSF_ERR_INVALID_ARG;
  if (ctx->ob.ready && ctx->ob.next_collect != ctx->ob.next_ticket) return VSF_ERR_INVALID_ARG;  // frames in the queue
  if (ctx->ob.ready && depth != ctx->ob_depth) {  // the queue is rebuilt by the next submit; the threshold and the window go
    VSF_HIP(hipSetDevice(ctx->device));
    stop_observe_threads(ctx);
    sync_all_streams(ctx);
    free_observe(ctx);
  }
  ctx->ob_depth = depth;
  ctx->ob_min_batch = min_batch;
  ctx->ob_in_flight = in_flight > 0 ? in_flight : 2;
  return VSF_OK;
}

vsf_status vsf_observe_stats(const vsf_ctx* ctx, int64_t* out, int n) {
  if (!ctx || !out || n < 1) return VSF_ERR_INVALID_ARG;
  const vsf_ctx::Observe& o = ctx->ob;
  const int64_t v[11] = {o.stat_frames, o.batches, o.stat_max_batch, o.stat_solo, o.stat_forced, o.stat_slot_waits,
                         (int64_t)o.depth, (int64_t)o.bmax, o.stat_copy_ns, o.stat_launch_ns, o.stat_wait_ns};
  for (int i = 0; i < n && i < 11; i++) out[i] = v[i];
  return VSF_OK;
}

vsf_status vsf_observe_reset(vsf_ctx* ctx) {
  VsfErrorScope scope_(ctx, false);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  stop_observe_threads(ctx);  // (frames that still wait are dropped)
  sync_all_streams(ctx);
  free_observe(ctx);
  return VSF_OK;
}

vsf_status vsf_observe_submit(vsf_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                              const vsf_calibration* calib, float best_percent, int frame_life, int64_t* ticket) {
  VsfErrorScope scope_(ctx, false);
  if (!ctx || !left || !right || !calib || !ticket || !(best_percent >= 0.f) || frame_life < 0 ||
      frame_life + 1 > VSF_OBSERVE_MAX_PAIRS)
    return VSF_ERR_INVALID_ARG;
  *ticket = -1;
  if (w != ctx->p.width || h != ctx->p.height || stride < (size_t)w || ctx->p.max_images < 2) return VSF_ERR_INVALID_ARG;
  if (ctx->p.max_keypoints >= 65536) return VSF_ERR_UNSUPPORTED;
  if (calib->triangulate_rows != 0 && calib->triangulate_rows != 4 && calib->triangulate_rows != 6)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_ctx::Observe& o = ctx->ob;
  // (re-sizing the window drops nothing that is still in the queue)
  if (o.ready && o.frame_life != frame_life) {
    if (o.next_collect != o.next_ticket) return VSF_ERR_INVALID_ARG;
    stop_observe_threads(ctx);
  }
  vsf_status st = ensure_observe(ctx, frame_life);
  if (st != VSF_OK) return st;
  if (o.next_ticket - o.next_collect >= o.depth) return VSF_ERR_INVALID_ARG;  // collect the oldest frame first
  vsf_ctx::ObserveLauncher& L = *o.launcher;
  const int slot = (int)(o.next_ticket % o.depth);
  {
    // a batch shares one calibration and one best_percent: a frame that brings others sends what waits first
    std::unique_lock<std::mutex> lk(L.mu);
    if (L.status != VSF_OK) return L.status;
    if (o.next_launch < o.next_ticket) {
      const vsf_ctx::ObserveFrame& w0 = o.frames[(size_t)((o.next_ticket - 1) % o.depth)];
      if (w0.best_percent != best_percent || !same_calibration(w0.calib, *calib)) {
        st = caller_pump(ctx, lk, true);
        if (st != VSF_OK) return st;
      }
    }
  }
  // ---- the two images into the frame's slot of the pinned staging ring, rows at the device pitch (the slot's previous
  // frame has been collected: its upload is long done) ----
  uint8_t* h_img = o.h_img + (size_t)slot * 2 * ctx->st_img_stride;
  const int64_t t_copy = now_ns();
  const vsf_ctx::ObserveCopyHelper::Job jr{h_img + ctx->st_img_stride, right, ctx->st_img_pitch, stride, (size_t)w, h};
  // frames are streaming in (the previous one is still in the queue): the helper thread takes the right image
  const bool helped = o.copy_helper && o.next_ticket > o.next_collect && o.copy_helper->post(jr);
  stage_image(h_img, ctx->st_img_pitch, left, stride, (size_t)w, h);
  if (helped)
    o.copy_helper->wait();
  else
    stage_image(jr.dst, jr.dst_pitch, jr.src, jr.src_pitch, jr.width, jr.rows);
  o.stat_copy_ns += now_ns() - t_copy;
  vsf_ctx::ObserveFrame& fr = o.frames[(size_t)slot];
  fr.calib = *calib;
  fr.best_percent = best_percent;
  fr.batch = -1;
  std::unique_lock<std::mutex> lk(L.mu);
  const bool was_empty = o.next_launch == o.next_ticket;
  *ticket = o.next_ticket++;
  o.last_submit_ns = now_ns();
  if (L.has_thread) {
    // the thread launches: it sleeps while nothing waits and polls while something does.  (A caller that collects right
    // away launches the frame itself there -- waking the thread would cost more than the launch.)
    if (was_empty) L.cv_thread.notify_one();
    return VSF_OK;
  }
  return caller_pump(ctx, lk, false);
}

// Waits for the frame of `ticket` and points at its result inside the pinned result ring (valid until `depth` further
// frames have been submitted).
static vsf_status observe_wait(vsf_ctx* ctx, int64_t ticket, const uint8_t** view, size_t* bytes) {
  vsf_ctx::Observe& o = ctx->ob;
  *bytes = 0;
  // frames leave in the order they entered (the host's bookkeeping is sequential)
  if (!o.ready || ticket < 0 || ticket != o.next_collect || ticket >= o.next_ticket) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_ctx::ObserveLauncher& L = *o.launcher;
  const int slot = (int)(ticket % o.depth);
  const vsf_ctx::ObserveBatch* b = nullptr;
  {
    std::unique_lock<std::mutex> lk(L.mu);
    if (ticket >= o.next_launch) {  // it still waits in staging: everything that waits leaves now
      const vsf_status st = caller_pump(ctx, lk, true);
      if (st != VSF_OK) return st;
    }
    if (ticket >= o.next_launch) return L.status != VSF_OK ? L.status : VSF_ERR_HIP;
    b = &o.batch[o.frames[(size_t)slot].batch];
  }
  {
    const int64_t t_wait = now_ns();
    VSF_HIP(hipEventSynchronize(b->ev_done));
    o.stat_wait_ns += now_ns() - t_wait;
  }
  const uint8_t* res = o.h_out + (size_t)slot * o.out_stride;
  const uint32_t* hdr = reinterpret_cast<const uint32_t*>(res);
  vsf_status st = VSF_OK;
  {
    std::unique_lock<std::mutex> lk(L.mu);
    o.next_collect = ticket + 1;
    if (!L.has_thread) st = caller_pump(ctx, lk, false);  // (the GPU may have room again)
  }
  if (hdr[0] != 0x4F465356u) return VSF_ERR_HIP;
  const_cast<uint32_t*>(hdr)[0] = 0;  // (the slot's next frame must write its own)
  // the filtered frames' size, for the matcher's launch choice: the largest of the last few frames with room to grow
  o.rows_hint = std::max((int)hdr[2] * 2 + 64, o.rows_hint - o.rows_hint / 8);
  *view = res;
  *bytes = hdr[3];
  if (st != VSF_OK) return st;
  if (hdr[11] != 0) return VSF_ERR_CAPACITY;  // the result does not fit its slot
  return hdr[12] != 0 ? VSF_ERR_CAPACITY : VSF_OK;
}

vsf_status vsf_observe_poll(vsf_ctx* ctx, int64_t ticket, int* ready) {
  VsfErrorScope scope_(ctx, false);
  if (!ctx || !ready) return VSF_ERR_INVALID_ARG;
  *ready = 0;
  vsf_ctx::Observe& o = ctx->ob;
  if (!o.ready || ticket < o.next_collect || ticket >= o.next_ticket) return VSF_ERR_INVALID_ARG;
  const vsf_ctx::ObserveBatch* b = nullptr;
  {
    std::lock_guard<std::mutex> g(o.launcher->mu);
    if (ticket >= o.next_launch) return o.launcher->status;  // it still waits in staging (nothing is forced)
    b = &o.batch[o.frames[(size_t)(ticket % o.depth)].batch];
  }
  VSF_HIP(hipSetDevice(ctx->device));
  const hipError_t e = hipEventQuery(b->ev_done);
  if (e == hipSuccess)
    *ready = 1;
  else if (e != hipErrorNotReady) {
    ctx->last_hip = (int)e;
    return VSF_ERR_HIP;
  }
  (void)hipGetLastError();
  return VSF_OK;
}

vsf_status vsf_observe_collect(vsf_ctx* ctx, int64_t ticket, uint8_t* out, size_t cap, size_t* out_bytes) {
  VsfErrorScope scope_(ctx, false);
  if (!ctx || !out || !out_bytes) return VSF_ERR_INVALID_ARG;
  const uint8_t* view = nullptr;
  const vsf_status st = observe_wait(ctx, ticket, &view, out_bytes);
  if (!view) return st;
  const uint32_t* hdr = reinterpret_cast<const uint32_t*>(view);
  if (hdr[11] != 0 || *out_bytes > cap) return VSF_ERR_CAPACITY;
  std::memcpy(out, view, *out_bytes);
  reinterpret_cast<uint32_t*>(out)[0] = 0x4F465356u;
  return st;
}

vsf_status vsf_observe_collect_view(vsf_ctx* ctx, int64_t ticket, const uint8_t** out, size_t* out_bytes) {
  VsfErrorScope scope_(ctx, false);
  if (!ctx || !out || !out_bytes) return VSF_ERR_INVALID_ARG;
  *out = nullptr;
  return observe_wait(ctx, ticket, out, out_bytes);
}

vsf_status vsf_observe_stereo(vsf_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                              const vsf_calibration* calib, float best_percent, int frame_life, uint8_t* out,
                              size_t cap, size_t* out_bytes) {
  VsfErrorScope scope_(ctx, false);
  if (!out || !out_bytes) return VSF_ERR_INVALID_ARG;
  *out_bytes = 0;
  int64_t ticket = -1;
  const vsf_status st = vsf_observe_submit(ctx, left, right, w, h, stride, calib, best_percent, frame_life, &ticket);
  if (st != VSF_OK) return st;
  return vsf_observe_collect(ctx, ticket, out, cap, out_bytes);
}

}  // extern "C"
