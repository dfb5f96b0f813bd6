// vsf_observe.hip -- one submission per Frontend::ObserveImage (slam_frontend.cc:400-472).
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

namespace vsfi {

void free_observe(vsf_ctx* ctx) {
  vsf_ctx::Observe& o = ctx->ob;
  hipFree(o.ring);
  hipFree(o.ring_counts);
  hipFree(o.kpf);
  hipFree(o.matches);
  hipFree(o.ints);
  hipFree(o.floats);
  hipFree(o.features);
  hipFree(o.pairs);
  hipFree(o.npairs);
  for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++) {
    if (o.h_img[i]) hipHostFree(o.h_img[i]);
    if (o.h_out[i]) hipHostFree(o.h_out[i]);
    if (o.h_meta[i]) hipHostFree(o.h_meta[i]);
    if (o.h_status[i]) hipHostFree(o.h_status[i]);
    if (o.ex_stream[i] && o.ex_stream[i] != ctx->stream) hipStreamDestroy(o.ex_stream[i]);
    if (o.ev_done[i]) hipEventDestroy(o.ev_done[i]);
  }
  o = vsf_ctx::Observe();
}

}  // namespace vsfi

extern "C" {

// ---------------- one submission per ObserveImage ----------------

size_t vsf_observe_capacity(const vsf_ctx* ctx, int frame_life) {
  if (!ctx || frame_life < 0 || frame_life + 1 > VSF_OBSERVE_MAX_PAIRS) return 0;
  const size_t K = (size_t)ctx->p.max_keypoints;
  return 64 + 4 * (size_t)((frame_life + 1 + 3) & ~3) + K * (28 + 28 + 32) + (size_t)(frame_life + 1) * K * 16;
}

static vsf_status ensure_observe(vsf_ctx* ctx, int frame_life) {
  vsf_ctx::Observe& o = ctx->ob;
  if (o.ring && o.frame_life == frame_life) return VSF_OK;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++)
    if (o.ex_stream[i]) VSF_HIP(hipStreamSynchronize(o.ex_stream[i]));
  float thr_state = 10000.0f;  // cc:353
  const bool had = o.floats != nullptr;
  if (had) VSF_HIP(hipMemcpy(&thr_state, o.floats + 2, sizeof(float), hipMemcpyDeviceToHost));
  free_observe(ctx);
  const size_t K = (size_t)ctx->p.max_keypoints, S = (size_t)frame_life + 2;
  o.slots = std::max(1, std::min(ctx->p.max_images / 2, VSF_OBSERVE_MAX_SLOTS));
  VSF_HIP(hipMalloc((void**)&o.ring, S * K * VSF_DESC_BYTES));
  VSF_HIP(hipMalloc((void**)&o.ring_counts, S * sizeof(int32_t)));
  VSF_HIP(hipMemset(o.ring_counts, 0, S * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.kpf, 2 * K * sizeof(vsf_keypoint)));
  VSF_HIP(hipMalloc((void**)&o.matches, VSF_OBSERVE_MAX_SLOTS * K * sizeof(vsf_dmatch)));
  VSF_HIP(hipMalloc((void**)&o.ints, 16 * sizeof(int32_t)));  // [0..5] raw stereo matches per slot, [8] features, [9] points
  VSF_HIP(hipMemset(o.ints, 0, 16 * sizeof(int32_t)));
  VSF_HIP(hipMalloc((void**)&o.floats, 4 * sizeof(float)));
  const float f4[4] = {0.f, 0.f, thr_state, 0.f};
  VSF_HIP(hipMemcpy(o.floats, f4, sizeof(f4), hipMemcpyHostToDevice));
  VSF_HIP(hipMalloc((void**)&o.features, K * sizeof(vsf_vision_feature)));
  VSF_HIP(hipMalloc((void**)&o.pairs, (size_t)(frame_life + 1) * K * 2 * sizeof(uint64_t)));
  VSF_HIP(hipMalloc((void**)&o.npairs, (size_t)(frame_life + 1) * sizeof(int32_t)));
  o.out_cap = vsf_observe_capacity(ctx, frame_life);
  for (int i = 0; i < o.slots; i++) {
    VSF_HIP(hipHostMalloc((void**)&o.h_img[i], 2 * ctx->st_img_stride, hipHostMallocMapped));
    VSF_HIP(hipHostMalloc((void**)&o.h_out[i], o.out_cap, hipHostMallocMapped));
    VSF_HIP(hipHostMalloc((void**)&o.h_meta[i], sizeof(vsf_ctx::ObserveMeta), hipHostMallocMapped));
    std::memset(o.h_meta[i], 0, sizeof(vsf_ctx::ObserveMeta));
    VSF_HIP(hipHostMalloc((void**)&o.h_status[i], sizeof(int32_t), hipHostMallocMapped));
    *o.h_status[i] = 0;
    // A stream per slot, each at a DIFFERENT stream priority (highest, default, lowest).  HIP multiplexes streams onto a few
    // hardware queues (round-robin at creation) and kernels of streams that share a queue run one after the other: with
    // streams of the default priority, one slot's stream landed on another's queue and its frames overlapped nothing
    // (kernel trace: 0.33 ms per frame, no better than one stream).  Streams of different priorities never share a queue,
    // so the chains of up to three frames -- ~25 small kernels each, bound by launch-to-launch latency -- run side by side.
    if (o.slots == 1) {
      o.ex_stream[i] = ctx->stream;
    } else {
      int prio_lo = 0, prio_hi = 0;
      VSF_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
      // (a fourth to sixth slot repeats the three priorities: it may share a hardware queue with an earlier slot -- then
      // those two take turns -- or land on a queue of its own)
      const int prio = i % 3 == 0 ? prio_hi : (i % 3 == 1 ? prio_lo : (prio_lo + prio_hi) / 2);
      VSF_HIP(hipStreamCreateWithPriority(&o.ex_stream[i], hipStreamNonBlocking, prio));
    }
    VSF_HIP(hipEventCreateWithFlags(&o.ev_done[i], hipEventDisableTiming));
  }
  o.frame_life = frame_life;
  // matcher scratch: pairs [0, frame_life] belong to the tail, pair frame_life + 1 + slot to the slot's stereo match
  vsf_status st = ensure_match_buffers(ctx, frame_life + 1 + VSF_OBSERVE_MAX_SLOTS, (int)K);
  if (st == VSF_OK) st = ensure_temporal_buffers(ctx, frame_life + 1);
  if (st == VSF_OK) st = ensure_residual_buffers(ctx, 1);
  return st;
}

vsf_status vsf_observe_reset(vsf_ctx* ctx) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++)
    if (ctx->ob.ex_stream[i]) VSF_HIP(hipStreamSynchronize(ctx->ob.ex_stream[i]));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
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
  if (ctx->ob.ring && ctx->ob.frame_life != frame_life)  // (re-sizing the window drops nothing that is still in flight)
    for (int i = 0; i < VSF_OBSERVE_MAX_SLOTS; i++)
      if (ctx->ob.ticket_of[i] >= 0) return VSF_ERR_INVALID_ARG;
  vsf_status st = ensure_observe(ctx, frame_life);
  if (st != VSF_OK) return st;
  vsf_ctx::Observe& o = ctx->ob;
  const int slot = (int)(o.next_ticket % o.slots);
  if (o.ticket_of[slot] >= 0) return VSF_ERR_INVALID_ARG;  // collect that frame first: its buffers are about to be reused
  hipStream_t ex = o.ex_stream[slot], s = ex;  // the frame's one stream
  const size_t K = (size_t)ctx->p.max_keypoints;
  const int Kc = (int)K;
  // ---- upload: rows into the slot's pinned staging at the device pitch, ONE copy command for both images ----
  const uint8_t* src[2] = {left, right};
  uint8_t* h_img = o.h_img[slot];
  for (int i = 0; i < 2; i++) {
    uint8_t* dst = h_img + (size_t)i * ctx->st_img_stride;
    if (stride == ctx->st_img_pitch) {  // the caller's rows already sit at the staging pitch: one copy per image
      std::memcpy(dst, src[i], (size_t)(h - 1) * stride + (size_t)w);
    } else {
      for (int y = 0; y < h; y++) std::memcpy(dst + (size_t)y * ctx->st_img_pitch, src[i] + (size_t)y * stride, (size_t)w);
    }
  }
  // (the slot's previous frame ran on this same stream: its tail has finished reading what the extraction now overwrites)
  uint8_t* d_img = ctx->st_img + (size_t)(2 * slot) * ctx->st_img_stride;
  VSF_HIP(hipMemcpyAsync(d_img, h_img, 2 * ctx->st_img_stride, hipMemcpyHostToDevice, ex));
  // ---- per-call parameters: written into pinned memory the kernels read directly ----
  const int n_past = (int)o.order.size(), n_pairs = n_past + 1, S = frame_life;
  vsf_ctx::ObserveMeta& M = *o.h_meta[slot];
  std::memcpy(M.F, calib->fundamental, sizeof(M.F));
  for (int p = 0; p < n_past; p++) {
    M.q_set[p] = o.order[p];  // oldest kept frame first: the order frame_list_ is walked in (cc:424)
    M.t_set[p] = S;
    M.best_percent[p] = best_percent;
  }
  M.q_set[n_past] = S + 1;  // Calculate3DPoints: GetFeatureMatches(right, left) with best_percent_ 1.0 (cc:129-132)
  M.t_set[n_past] = S;
  M.best_percent[n_past] = 1.0f;
  // ---- ExtractFeatures x 2 + GetMatches (cc:411-416), on the slot's stream and in the slot's buffers ----
  const VsfImages im{ctx->st_img, ctx->st_img_stride, ctx->st_img_pitch, 2 * (slot + 1)};
  vsf_keypoint* kp_raw = ctx->st_kp + (size_t)(2 * slot) * K;
  uint8_t* desc_raw = ctx->st_desc + (size_t)(2 * slot) * K * VSF_DESC_BYTES;
  int32_t* counts_raw = ctx->st_counts + 2 * slot;
  int32_t* status_word = ctx->d_status + 1 + slot;  // this frame's own (see vsf_ctx::d_status)
  extract_on(ctx, ex, im, 2 * slot, 2, ctx->st_kp, ctx->st_desc, ctx->st_counts, false, o.slots > 1 ? &o.side[slot] : nullptr,
             status_word);
  ctx->last_images = VsfImages{d_img, ctx->st_img_stride, ctx->st_img_pitch, 2};
  ctx->last_valid = true;
  int32_t* nmatches = o.ints + slot;
  vsf_dmatch* raw_matches = o.matches + (size_t)slot * K;
  {
    const size_t scratch = (size_t)(frame_life + 1 + slot) * K * 2;
    match_on(ctx, ex, desc_raw, counts_raw, K * VSF_DESC_BYTES, nullptr, nullptr, 0, 1, ctx->m_idx2 + scratch,
             ctx->m_dist2 + scratch, raw_matches, nmatches, status_word);
  }
  // the tails run in frame order: this frame's waits for the previous frame's (on another slot's stream)
  if (o.slots > 1 && o.next_ticket > 0) {
    const int prev = (int)((o.next_ticket - 1) % o.slots);
    if (o.done_valid[prev]) VSF_HIP(hipStreamWaitEvent(s, o.ev_done[prev], 0));
  }
  // ---- RemoveAmbigStereo (cc:417): the current frame lands in ring sets S (left) and S + 1 (right) ----
  float *means = o.floats, *thr = o.floats + 1, *thr_state = o.floats + 2;
  uint8_t* cur_desc = o.ring + (size_t)S * K * VSF_DESC_BYTES;
  int32_t* cur_counts = o.ring_counts + S;
  {
    StageTimer t(ctx, s, VSF_STAGE_TAIL, 3);
    vsf_launch_stereo_residuals(kp_raw, raw_matches, nmatches, 1, Kc, M.F, nullptr, ctx->p.residual_order, ctx->f_residual, means, s);
    vsf_launch_stereo_thresholds(means, 1, thr_state, thr, s);
    vsf_launch_stereo_filter_only(kp_raw, desc_raw, raw_matches, nmatches, 1, Kc, ctx->f_residual, thr, o.kpf, cur_desc,
                                  cur_counts, s);
  }
  // ---- GetFeatureMatches against every kept frame + the right->left matches of Calculate3DPoints: one matcher
  // launch, one sort launch (per-pair best_percent) ----
  {
    StageTimer t(ctx, s, VSF_STAGE_KNN2, 1);
    vsf_launch_knn2(o.ring, o.ring_counts, K * VSF_DESC_BYTES, M.q_set, M.t_set, n_pairs, Kc, ctx->m_idx2, ctx->m_dist2, s,
                    ctx->tuning.match_int8 != 0);
  }
  {
    StageTimer t(ctx, s, VSF_STAGE_RATIO, 1);
    vsf_launch_ratio_compact(o.ring_counts, M.q_set, M.t_set, n_pairs, Kc, ctx->m_idx2, ctx->m_dist2, ctx->p.ratio_num,
                             ctx->p.ratio_shift, ctx->t_matches, ctx->t_nmatches, status_word, s);
  }
  {
    StageTimer t(ctx, s, VSF_STAGE_TAIL, 3);
    vsf_launch_sort_trim(ctx->t_matches, ctx->t_nmatches, n_pairs, Kc, best_percent, M.best_percent, ctx->t_sortkeys,
                         o.pairs, o.npairs, s, ctx->tuning.sort_serial != 0, ctx->tuning.lds_limit);
    // ---- Calculate3DPoints + VisionFeature + UndistortFeaturePoints (cc:437-443) ----
    int32_t *nfeat = o.ints + 8, *npoints = o.ints + 9;
    vsf_launch_vision_features(o.kpf, cur_counts, o.pairs + (size_t)n_past * K * 2, o.npairs + n_past, 1, Kc, *calib,
                               o.features, nfeat, npoints, s);
    // ---- the compact result into pinned memory; the filtered left frame into its ring slot (cc:467-470) ----
    int ring_slot;
    if (frame_life == 0) {
      ring_slot = S + 1;  // nothing is kept: park it on the right frame's set
    } else if (n_past >= frame_life) {
      ring_slot = o.order.front();
    } else {
      ring_slot = n_past;
      for (int c = 0; c < frame_life; c++)
        if (std::find(o.order.begin(), o.order.end(), c) == o.order.end()) {
          ring_slot = c;
          break;
        }
    }
    VsfObserveArgs a;
    a.n_pairs = n_pairs;
    a.max_rows = Kc;
    a.counts_raw = counts_raw;
    a.nmatches = nmatches;
    a.counts_f = cur_counts;
    a.npoints = npoints;
    a.means = means;
    a.thr = thr;
    a.thr_state = thr_state;
    a.features = o.features;
    a.kp_f = o.kpf;
    a.desc_f = cur_desc;
    a.pairs = o.pairs;
    a.npairs = o.npairs;
    a.ring_desc = o.ring + (size_t)ring_slot * K * VSF_DESC_BYTES;
    a.ring_count = o.ring_counts + ring_slot;
    a.out = o.h_out[slot];
    a.out_cap = (uint32_t)std::min<size_t>(o.out_cap, 0xFFFFFFF0u);
    vsf_launch_observe_pack(a, s);
    if (frame_life > 0) {
      if (n_past >= frame_life) o.order.erase(o.order.begin());
      o.order.push_back(ring_slot);
    }
  }
  // the frame's own status word (everything the frame ran wrote into it, nothing else did), then "this frame is done"
  VSF_HIP(hipMemcpyAsync(o.h_status[slot], status_word, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  VSF_HIP(hipMemsetAsync(status_word, 0, sizeof(int32_t), s));
  VSF_HIP(hipEventRecord(o.ev_done[slot], s));
  o.done_valid[slot] = true;
  VSF_STICKY();
  o.ticket_of[slot] = o.next_ticket;
  *ticket = o.next_ticket++;
  return VSF_OK;
}

vsf_status vsf_observe_collect(vsf_ctx* ctx, int64_t ticket, uint8_t* out, size_t cap, size_t* out_bytes) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !out || !out_bytes || ticket < 0) return VSF_ERR_INVALID_ARG;
  *out_bytes = 0;
  vsf_ctx::Observe& o = ctx->ob;
  const int slot = (int)(ticket % std::max(o.slots, 1));
  if (!o.ring || o.ticket_of[slot] != ticket) return VSF_ERR_INVALID_ARG;
  // frames leave in the order they entered (the host's bookkeeping is sequential): an older frame must be collected first
  for (int i = 0; i < o.slots; i++)
    if (o.ticket_of[i] >= 0 && o.ticket_of[i] < ticket) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipEventSynchronize(o.ev_done[slot]));
  o.ticket_of[slot] = -1;
  vsf_status st = VSF_OK;
  if (*o.h_status[slot] & 1) st = VSF_ERR_CAPACITY;
  const uint32_t* hdr = reinterpret_cast<const uint32_t*>(o.h_out[slot]);
  if (hdr[0] != 0x4F465356u) return VSF_ERR_HIP;
  const size_t total = hdr[3];
  *out_bytes = total;
  if (hdr[11] != 0 || total > cap) return VSF_ERR_CAPACITY;
  std::memcpy(out, o.h_out[slot], total);
  return st;
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
