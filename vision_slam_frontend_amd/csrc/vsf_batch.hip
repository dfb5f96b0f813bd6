// vsf_batch.hip -- the device-pointer, asynchronous, batched entry points: extraction, matching, and the reference's own
// steps between matcher and outputs (RemoveAmbigStereo, GetFeatureMatches, Calculate3DPoints, the packed payload).
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

extern "C" {

vsf_status vsf_extract_batch_dev(vsf_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride,
                                 size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  InputEventScope input(ctx);
  if (!d_kp || !d_desc || !d_counts) return VSF_ERR_INVALID_ARG;
  vsf_status st = validate_images(ctx, d_imgs, n_images, image_stride, row_stride);
  if (st != VSF_OK) return st;
  VSF_HIP(hipSetDevice(ctx->device));
  VsfImages im{d_imgs, image_stride, row_stride, n_images};
  input.wait();
  return extract_async(ctx, im, d_kp, d_desc, d_counts, true);
}

vsf_status vsf_tune_fast_resident(vsf_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride,
                                  size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                                  int samples, float* ms_grid, float* ms_resident) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_desc || !d_counts || samples < 1 || samples > 64 || !ms_grid || !ms_resident)
    return VSF_ERR_INVALID_ARG;
  *ms_grid = *ms_resident = 0.f;
  vsf_status st = validate_images(ctx, d_imgs, n_images, image_stride, row_stride);
  if (st != VSF_OK) return st;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_ctx::FastTune& T = ctx->fast_tune;
  T.n = n_images;
  T.choice = 0;
  // batches the blur does not run beside have one form only
  if (!(ctx->blur_overlap && n_images >= 32 && ctx->blur_stream && ctx->lanes == 1)) return VSF_OK;
  for (hipEvent_t& e : T.ev)
    if (!e) VSF_HIP(hipEventCreate(&e));
  const VsfImages im{d_imgs, image_stride, row_stride, n_images};
  sync_all_streams(ctx);  // nothing of an earlier call beside the timed runs
  std::vector<float> ms[2];
  vsf_status out = VSF_OK;
  for (int run = 0; run < 1 + 2 * samples && out == VSF_OK; run++) {
    const int form = run == 0 ? 0 : (run - 1) & 1;  // warm-up (grid), then grid / resident alternately on the SAME input
    ctx->fast_force = form ? 3 : 0;
    hipError_t e = hipEventRecord(T.ev[0], ctx->stream);
    // (inputs_complete = false: no cross-call pipelining inside the measurement, every run is the whole extraction)
    extract_on(ctx, ctx->stream, im, 0, n_images, d_kp, d_desc, d_counts, false);
    if (e == hipSuccess) e = hipEventRecord(T.ev[1], ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(T.ev[1]);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, T.ev[0], T.ev[1]);
    if (e != hipSuccess) {
      ctx->last_hip = (int)e;
      out = VSF_ERR_HIP;
    } else if (run > 0) {
      ms[form].push_back(t);
    }
  }
  ctx->fast_force = -1;
  ctx->last_images = im;
  ctx->last_valid = true;
  if (out != VSF_OK) return out;
  VSF_STICKY();
  for (auto& v : ms) std::sort(v.begin(), v.end());
  *ms_grid = ms[0][ms[0].size() / 2];
  *ms_resident = ms[1][ms[1].size() / 2];
  T.choice = *ms_resident < *ms_grid ? 3 : 0;
  return VSF_OK;
}

vsf_status vsf_match_batch_dev(vsf_ctx* ctx, const uint8_t* d_desc, const int32_t* d_counts, size_t set_stride,
                               const int32_t* d_q_set, const int32_t* d_t_set, int n_pairs, int32_t* d_idx2,
                               int32_t* d_dist2, vsf_dmatch* d_matches, int32_t* d_nmatches) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_desc || !d_counts || n_pairs < 1 || !d_matches || !d_nmatches || (set_stride & 15))
    return VSF_ERR_INVALID_ARG;
  if ((d_idx2 == nullptr) != (d_dist2 == nullptr) || (d_q_set == nullptr) != (d_t_set == nullptr))
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const int rows = ctx->p.max_keypoints;
  if (!d_idx2) {
    vsf_status st = ensure_match_buffers(ctx, n_pairs, rows);
    if (st != VSF_OK) return st;
    d_idx2 = ctx->m_idx2;
    d_dist2 = ctx->m_dist2;
  }
  vsf_status st = run_chunked(ctx, n_pairs, [&](hipStream_t s, int p0, int n) {
    match_on(ctx, s, d_desc, d_counts, set_stride, d_q_set, d_t_set, p0, n, d_idx2, d_dist2, d_matches, d_nmatches);
  });
  if (st != VSF_OK) return st;
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_stereo_batch_dev(vsf_ctx* ctx, const uint8_t* d_imgs, int n_frames, size_t image_stride,
                                size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                                vsf_dmatch* d_matches, int32_t* d_nmatches) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  InputEventScope input(ctx);
  if (n_frames < 1 || !d_kp || !d_desc || !d_counts || !d_matches || !d_nmatches) return VSF_ERR_INVALID_ARG;
  vsf_status st = validate_images(ctx, d_imgs, 2 * n_frames, image_stride, row_stride);
  if (st != VSF_OK) return st;
  VSF_HIP(hipSetDevice(ctx->device));
  st = ensure_match_buffers(ctx, n_frames, ctx->p.max_keypoints);
  if (st != VSF_OK) return st;
  input.wait();
  const VsfImages im{d_imgs, image_stride, row_stride, 2 * n_frames};
  const size_t set_stride = (size_t)ctx->p.max_keypoints * VSF_DESC_BYTES;
  st = run_chunked(ctx, n_frames, [&](hipStream_t s, int fa, int nf) {
    extract_on(ctx, s, im, 2 * fa, 2 * nf, d_kp, d_desc, d_counts, true);
    match_on(ctx, s, d_desc, d_counts, set_stride, nullptr, nullptr, fa, nf, ctx->m_idx2, ctx->m_dist2, d_matches,
             d_nmatches);
  });
  if (st != VSF_OK) return st;
  ctx->last_images = im;
  ctx->last_valid = true;
  VSF_STICKY();
  return VSF_OK;
}

// ---------------- reference steps between matcher and outputs ----------------

vsf_status vsf_remove_ambig_stereo_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const uint8_t* d_desc,
                                             const vsf_dmatch* d_matches, const int32_t* d_nmatches, int n_frames,
                                             const float* F, float thr_in, const float* d_thr_override,
                                             float* d_means, float* d_thr, vsf_keypoint* d_kp_out,
                                             uint8_t* d_desc_out, int32_t* d_counts_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_desc || !d_matches || !d_nmatches || n_frames < 1 || !F || !d_means || !d_kp_out ||
      !d_desc_out || !d_counts_out || (!d_thr_override && !d_thr))
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const size_t K = (size_t)ctx->p.max_keypoints;
  {
    vsf_status st = ensure_residual_buffers(ctx, n_frames);
    if (st != VSF_OK) return st;
  }
  vsf_launch_stereo_filter(d_kp, d_desc, d_matches, d_nmatches, n_frames, (int)K, nullptr, F, ctx->p.residual_order, d_thr_override, thr_in,
                           ctx->f_residual, d_means, d_thr, d_kp_out, d_desc_out, d_counts_out, ctx->stream);
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_stereo_residuals_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const vsf_dmatch* d_matches,
                                          const int32_t* d_nmatches, int n_frames, const float* F, float* d_means) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_matches || !d_nmatches || n_frames < 1 || !F || !d_means) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_status st = ensure_residual_buffers(ctx, n_frames);
  if (st != VSF_OK) return st;
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_stereo_residuals(d_kp, d_matches, d_nmatches, n_frames, ctx->p.max_keypoints, nullptr, F, ctx->p.residual_order, ctx->f_residual,
                                d_means, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_stereo_thresholds_dev(vsf_ctx* ctx, const float* d_means, int n, float* d_thr_state, float* d_thr) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_means || n < 1 || !d_thr_state || !d_thr) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_stereo_thresholds(d_means, n, d_thr_state, d_thr, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_stereo_filter_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const uint8_t* d_desc,
                                       const vsf_dmatch* d_matches, const int32_t* d_nmatches, int n_frames,
                                       const float* d_thr, vsf_keypoint* d_kp_out, uint8_t* d_desc_out,
                                       int32_t* d_counts_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_kp || !d_desc || !d_matches || !d_nmatches || n_frames < 1 || !d_thr || !d_kp_out || !d_desc_out ||
      !d_counts_out)
    return VSF_ERR_INVALID_ARG;
  if (n_frames > ctx->f_frames || !ctx->f_residual) return VSF_ERR_INVALID_ARG;  // no residuals of such a batch
  VSF_HIP(hipSetDevice(ctx->device));
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_stereo_filter_only(d_kp, d_desc, d_matches, d_nmatches, n_frames, ctx->p.max_keypoints, ctx->f_residual,
                                  d_thr, d_kp_out, d_desc_out, d_counts_out, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_vision_features_batch_dev(vsf_ctx* ctx, const vsf_calibration* calib, const vsf_keypoint* d_kp,
                                         const uint8_t* d_desc, const int32_t* d_counts, int n_frames,
                                         vsf_vision_feature* d_features, int32_t* d_nfeatures, int32_t* d_npoints) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !calib || !d_kp || !d_desc || !d_counts || n_frames < 1 || !d_features || !d_nfeatures)
    return VSF_ERR_INVALID_ARG;
  if (calib->triangulate_rows != 0 && calib->triangulate_rows != 4 && calib->triangulate_rows != 6)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const size_t K = (size_t)ctx->p.max_keypoints;
  {
    vsf_status st0 = ensure_vision_buffers(ctx, n_frames);
    if (st0 != VSF_OK) return st0;
  }
  // Calculate3DPoints: best_percent_ forced to 1.0 (cc:129-132)
  vsf_status st = vsf_feature_matches_batch_dev(ctx, d_desc, d_counts, K * VSF_DESC_BYTES, ctx->v_sets,
                                                ctx->v_sets + ctx->v_frames, n_frames, 1.0f, ctx->v_pairs, ctx->v_npairs);
  if (st != VSF_OK) return st;
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_vision_features(d_kp, d_counts, ctx->v_pairs, ctx->v_npairs, n_frames, (int)K, *calib, d_features,
                               d_nfeatures, d_npoints, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

size_t vsf_packed_outputs_capacity(const vsf_ctx* ctx, int n_frames, int n_pairs) {
  if (!ctx || n_frames < 0 || n_pairs < 0) return 0;
  const size_t K = (size_t)ctx->p.max_keypoints;
  return 16 + 4 * ((size_t)n_frames + n_pairs) + (size_t)n_frames * K * sizeof(vsf_vision_feature) +
         (size_t)n_pairs * K * sizeof(vsf_feature_match);
}

vsf_status vsf_pack_outputs_dev(vsf_ctx* ctx, const vsf_vision_feature* d_features, const int32_t* d_nfeatures,
                                int n_frames, const uint64_t* d_pairs, const int32_t* d_npairs, int n_pairs,
                                uint8_t* d_payload, size_t payload_cap) {
  VsfErrorScope scope_(ctx);
  if (!ctx || n_frames < 0 || n_pairs < 0 || n_frames + n_pairs < 1 || (n_frames > 0 && (!d_features || !d_nfeatures)) ||
      (n_pairs > 0 && (!d_pairs || !d_npairs)) || !d_payload || ((uintptr_t)d_payload & 3) ||
      payload_cap < 16 + 4 * ((size_t)n_frames + n_pairs))
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const int n = n_frames + n_pairs;
  {
    vsf_status st0 = ensure_pack_buffers(ctx, n);
    if (st0 != VSF_OK) return st0;
  }
  const uint32_t cap = (uint32_t)std::min<size_t>(payload_cap, 0xFFFFFFFCu);
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 2);
    vsf_launch_pack_outputs(d_features, d_nfeatures, n_frames, d_pairs, d_npairs, n_pairs, ctx->p.max_keypoints,
                            d_payload, cap, ctx->pk_offsets, ctx->d_status, ctx->stream);
  }
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_feature_matches_batch_dev(vsf_ctx* ctx, const uint8_t* d_desc, const int32_t* d_counts,
                                         size_t set_stride, const int32_t* d_q_set, const int32_t* d_t_set,
                                         int n_pairs, float best_percent, uint64_t* d_pairs, int32_t* d_npairs) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_pairs || !d_npairs || n_pairs < 1 || !(best_percent >= 0.f)) return VSF_ERR_INVALID_ARG;
  if (ctx->p.max_keypoints >= 65536) return VSF_ERR_UNSUPPORTED;  // (query, train) indices are packed 16 + 16 bit
  VSF_HIP(hipSetDevice(ctx->device));
  const size_t K = (size_t)ctx->p.max_keypoints;
  {
    vsf_status st0 = ensure_temporal_buffers(ctx, n_pairs);
    if (st0 != VSF_OK) return st0;
  }
  vsf_status st = vsf_match_batch_dev(ctx, d_desc, d_counts, set_stride, d_q_set, d_t_set, n_pairs, nullptr, nullptr,
                                      ctx->t_matches, ctx->t_nmatches);
  if (st != VSF_OK) return st;
  {
    StageTimer t(ctx, ctx->stream, VSF_STAGE_TAIL, 1);
    vsf_launch_sort_trim(ctx->t_matches, ctx->t_nmatches, n_pairs, (int)K, best_percent, nullptr, ctx->t_sortkeys,
                         d_pairs, d_npairs, ctx->stream, false, ctx->tuning.lds_limit);
  }
  VSF_STICKY();
  return VSF_OK;
}

}  // extern "C"
