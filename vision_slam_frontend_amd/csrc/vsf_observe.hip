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
// When a batch leaves: whenever fewer than `in_flight` batches are on the GPU (a lone frame leaves at once: the synchronous
// call is a batch of one), when a whole batch waits, or when somebody collects a frame that still waits.  While the GPU is
// busy, frames accumulate -- the batch size follows the caller's rate by itself.
// The kept frames' filtered descriptors live in a ring of descriptor sets in HBM (frame g in set g % ring); the pair list
// of a batch addresses them by set index, so a frame matches against frames of earlier batches and of its own alike.
// Results are those of one frame at a time, bit for bit (tests/test_gpu_observe.py).
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "vsf_ctx.h"

using namespace vsfi;

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

bool same_calibration(const vsf_calibration& a, const vsf_calibration& b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

}  // namespace

namespace vsfi {

void free_observe(vsf_ctx* ctx) {
  vsf_ctx::Observe& o = ctx->ob;
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
  VSF_HIP(hipStreamCreateWithFlags(&o.copy_stream, hipStreamNonBlocking));
  VSF_HIP(hipDeviceSynchronize());
  o.ready = true;
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
vsf_status launch_batch(vsf_ctx* ctx, int64_t t0, int n) {
  vsf_ctx::Observe& o = ctx->ob;
  const int bi = (int)(o.batches % vsf_ctx::kObserveBatchSlots);
  vsf_ctx::ObserveBatch& b = o.batch[bi];
  // The slot's previous batch must have left the GPU: its kernels read the pinned parameter block that is rewritten below
  // (with `in_flight` batches on the GPU and four slots it has, long ago).
  if (b.used) VSF_HIP(hipEventSynchronize(b.ev_done));
  const size_t K = (size_t)ctx->p.max_keypoints;
  const int Kc = (int)K, life = o.frame_life;
  const vsf_ctx::ObserveFrame& f0 = o.frames[(size_t)(t0 % o.depth)];
  // A lone frame with nothing else on the GPU runs on ONE stream from upload to result (no event hops in its chain);
  // otherwise copy, extraction and tail have a stream each, so that the next batch's upload and extraction run beside
  // this one's tail.
  const bool solo = n == 1 && batches_on_gpu(ctx) == 0;
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
    vsf_launch_stereo_residuals(b.kp_raw, b.matches, b.nmatches, n, Kc, nullptr, f0.calib.fundamental, ctx->p.residual_order,
                                o.residual, means, s_tail);
    vsf_launch_stereo_thresholds(means, n, thr_state, thr, s_tail);
    vsf_launch_stereo_filter_only(b.kp_raw, b.desc_raw, b.matches, b.nmatches, n, Kc, o.residual, thr, o.kpf, o.sets, counts_f,
                                  s_tail, M.out_sets, o.set_counts);
  }
  // ---- every GetFeatureMatches of the batch: one matcher launch, one sort launch (per-pair best_percent) ----
  {
    StageTimer t(ctx, s_tail, VSF_STAGE_KNN2, 1);
    vsf_launch_knn2(o.sets, o.set_counts, K * VSF_DESC_BYTES, M.q_set, M.t_set, n_pairs, Kc, o.t_idx2, o.t_dist2, s_tail,
                    ctx->tuning.match_int8 != 0);
  }
  {
    StageTimer t(ctx, s_tail, VSF_STAGE_RATIO, 1);
    vsf_launch_ratio_compact(o.set_counts, M.q_set, M.t_set, n_pairs, Kc, o.t_idx2, o.t_dist2, ctx->p.ratio_num,
                             ctx->p.ratio_shift, o.t_matches, o.t_nmatches, b.status, s_tail);
  }
  {
    StageTimer t(ctx, s_tail, VSF_STAGE_TAIL, 3);
    vsf_launch_sort_trim(o.t_matches, o.t_nmatches, n_pairs, Kc, f0.best_percent, M.best_percent, o.t_sortkeys, o.pairs,
                         o.npairs, s_tail, ctx->tuning.sort_serial != 0, ctx->tuning.lds_limit);
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
  o.next_launch = t0 + n;
  VSF_STICKY();
  return VSF_OK;
}

// Sends waiting frames to the GPU.  force: everything that waits, now (somebody collects one of them).
vsf_status pump(vsf_ctx* ctx, bool force) {
  vsf_ctx::Observe& o = ctx->ob;
  while (o.next_launch < o.next_ticket) {
    const int pending = (int)(o.next_ticket - o.next_launch);
    int n = 0;
    if (force || pending >= o.bmax) {
      n = std::min(pending, o.bmax);
    } else {
      const int busy = batches_on_gpu(ctx);
      if (busy < ctx->ob_in_flight && (busy == 0 || pending >= ctx->ob_min_batch)) n = pending;
    }
    if (n == 0) break;
    const vsf_status st = launch_batch(ctx, o.next_launch, n);
    if (st != VSF_OK) return st;
  }
  return VSF_OK;
}

}  // namespace

extern "C" {

size_t vsf_observe_capacity(const vsf_ctx* ctx, int frame_life) {
  if (!ctx || frame_life < 0 || frame_life + 1 > VSF_OBSERVE_MAX_PAIRS) return 0;
  const size_t K = (size_t)ctx->p.max_keypoints;
  return 64 + 4 * (size_t)((frame_life + 1 + 3) & ~3) + K * (28 + 28 + 32) + (size_t)(frame_life + 1) * K * 16;
}

vsf_status vsf_observe_configure(vsf_ctx* ctx, int depth, int min_batch, int in_flight) {
  VsfErrorScope scope_(ctx);
  if (!ctx || depth < 0 || depth > 1024 || min_batch < 0 || in_flight < 0 || in_flight > vsf_ctx::kObserveBatchSlots - 1)
    return VSF_ERR_INVALID_ARG;
  if (ctx->ob.ready && ctx->ob.next_collect != ctx->ob.next_ticket) return VSF_ERR_INVALID_ARG;  // frames in the queue
  if (ctx->ob.ready && depth != ctx->ob_depth) {  // the queue is rebuilt by the next submit; the threshold and the window go
    VSF_HIP(hipSetDevice(ctx->device));
    sync_all_streams(ctx);
    free_observe(ctx);
  }
  ctx->ob_depth = depth;
  ctx->ob_min_batch = std::max(1, min_batch);
  ctx->ob_in_flight = in_flight > 0 ? in_flight : 2;
  return VSF_OK;
}

vsf_status vsf_observe_reset(vsf_ctx* ctx) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  sync_all_streams(ctx);
  free_observe(ctx);
  return VSF_OK;
}

vsf_status vsf_observe_submit(vsf_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                              const vsf_calibration* calib, float best_percent, int frame_life, int64_t* ticket) {
  VsfErrorScope scope_(ctx);
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
  if (o.ready && o.frame_life != frame_life && o.next_collect != o.next_ticket) return VSF_ERR_INVALID_ARG;
  vsf_status st = ensure_observe(ctx, frame_life);
  if (st != VSF_OK) return st;
  if (o.next_ticket - o.next_collect >= o.depth) return VSF_ERR_INVALID_ARG;  // collect the oldest frame first
  // a batch shares one calibration and one best_percent: a frame that brings others starts a new batch
  if (o.next_launch < o.next_ticket) {
    const vsf_ctx::ObserveFrame& w0 = o.frames[(size_t)(o.next_launch % o.depth)];
    if (w0.best_percent != best_percent || !same_calibration(w0.calib, *calib)) {
      st = pump(ctx, true);
      if (st != VSF_OK) return st;
    }
  }
  // ---- the two images into the frame's slot of the pinned staging ring, rows at the device pitch ----
  const int slot = (int)(o.next_ticket % o.depth);
  const uint8_t* src[2] = {left, right};
  uint8_t* h_img = o.h_img + (size_t)slot * 2 * ctx->st_img_stride;
  for (int i = 0; i < 2; i++) {
    uint8_t* dst = h_img + (size_t)i * ctx->st_img_stride;
    if (stride == ctx->st_img_pitch) {  // the caller's rows already sit at the staging pitch: one copy per image
      std::memcpy(dst, src[i], (size_t)(h - 1) * stride + (size_t)w);
    } else {
      for (int y = 0; y < h; y++) std::memcpy(dst + (size_t)y * ctx->st_img_pitch, src[i] + (size_t)y * stride, (size_t)w);
    }
  }
  vsf_ctx::ObserveFrame& fr = o.frames[(size_t)slot];
  fr.calib = *calib;
  fr.best_percent = best_percent;
  fr.batch = -1;
  *ticket = o.next_ticket++;
  return pump(ctx, false);
}

// Waits for the frame of `ticket` and points at its result inside the pinned result ring (valid until `depth` further
// frames have been submitted).
static vsf_status observe_wait(vsf_ctx* ctx, int64_t ticket, const uint8_t** view, size_t* bytes) {
  vsf_ctx::Observe& o = ctx->ob;
  *bytes = 0;
  // frames leave in the order they entered (the host's bookkeeping is sequential)
  if (!o.ready || ticket < 0 || ticket != o.next_collect || ticket >= o.next_ticket) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  if (ticket >= o.next_launch) {  // it still waits in staging: everything that waits leaves now
    const vsf_status st = pump(ctx, true);
    if (st != VSF_OK) return st;
  }
  const int slot = (int)(ticket % o.depth);
  const vsf_ctx::ObserveBatch& b = o.batch[o.frames[(size_t)slot].batch];
  VSF_HIP(hipEventSynchronize(b.ev_done));
  o.next_collect = ticket + 1;
  const uint8_t* res = o.h_out + (size_t)slot * o.out_stride;
  const uint32_t* hdr = reinterpret_cast<const uint32_t*>(res);
  if (hdr[0] != 0x4F465356u) return VSF_ERR_HIP;
  const_cast<uint32_t*>(hdr)[0] = 0;  // (the slot's next frame must write its own)
  *view = res;
  *bytes = hdr[3];
  vsf_status st = pump(ctx, false);  // (the GPU may have room again)
  if (st != VSF_OK) return st;
  if (hdr[11] != 0) return VSF_ERR_CAPACITY;  // the result does not fit its slot
  return hdr[12] != 0 ? VSF_ERR_CAPACITY : VSF_OK;
}

vsf_status vsf_observe_collect(vsf_ctx* ctx, int64_t ticket, uint8_t* out, size_t cap, size_t* out_bytes) {
  VsfErrorScope scope_(ctx);
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
  VsfErrorScope scope_(ctx);
  if (!ctx || !out || !out_bytes) return VSF_ERR_INVALID_ARG;
  *out = nullptr;
  return observe_wait(ctx, ticket, out, out_bytes);
}

vsf_status vsf_observe_stereo(vsf_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                              const vsf_calibration* calib, float best_percent, int frame_life, uint8_t* out,
                              size_t cap, size_t* out_bytes) {
  VsfErrorScope scope_(ctx);
  if (!out || !out_bytes) return VSF_ERR_INVALID_ARG;
  *out_bytes = 0;
  int64_t ticket = -1;
  const vsf_status st = vsf_observe_submit(ctx, left, right, w, h, stride, calib, best_percent, frame_life, &ticket);
  if (st != VSF_OK) return st;
  return vsf_observe_collect(ctx, ticket, out, cap, out_bytes);
}

}  // extern "C"
