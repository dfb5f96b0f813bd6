// vsf_api.hip -- the context of the C ABI (include/vsf.h): creation and destruction, streams, options, the scratch that
// follows the batch size, and the two composed stages every entry point is made of: extract_on (detectAndCompute,
// slam_frontend.cc:266-280) and match_on (knnMatch + ratio test, slam_frontend.cc:521-538).
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

thread_local int vsf_tls_hip_error = 0;

namespace vsfi {

vsf_status alloc_devset(vsf_ctx* ctx, const Geometry& G, DevSet* ds, bool orb, int n_images) {
  VSF_HIP(upload(&ds->levels, G.levels));
  VSF_HIP(upload(&ds->units, G.units));
  VSF_HIP(upload(&ds->blur_mma_units, G.blur_mma_units));
  VSF_HIP(upload(&ds->blur_mma_units_small, G.blur_mma_units_small));
  VSF_HIP(upload(&ds->blur_tcol, G.blur_tcol));
  VSF_HIP(upload(&ds->blur_tv, G.blur_tv));
  VSF_HIP(upload(&ds->ic_table, build_ic_table()));
  VsfDev& d = ds->d;
  d.ic_table = ds->ic_table;
  d.levels = ds->levels;
  d.units = ds->units;
  const size_t n = (size_t)n_images;
  if (orb) {
    VSF_HIP(hipMalloc((void**)&d.pyr, n * G.g.pyr_bytes));
    VSF_HIP(hipMalloc((void**)&d.blur, n * G.g.pyr_bytes));
    VSF_HIP(hipMalloc((void**)&d.scratch, n * 6 * G.g.cand_entries * sizeof(uint32_t)));
    VSF_HIP(hipMalloc((void**)&d.lvlkp, n * G.g.lvlkp_entries * sizeof(VsfLevelKp)));
    VSF_HIP(hipMalloc((void**)&d.lvl_count, n * G.g.nlevels * sizeof(int32_t)));
    VSF_HIP(hipMemset(d.lvl_count, 0, n * G.g.nlevels * sizeof(int32_t)));
  }
  VSF_HIP(hipMalloc((void**)&d.cand, n * G.g.cand_entries * sizeof(uint32_t)));
  const size_t rs_bytes = n * (size_t)std::max(G.g.nunits, 1) * VSF_FAST_RS_STRIDE * sizeof(uint16_t);
  VSF_HIP(hipMalloc((void**)&d.rowstart, rs_bytes));
  VSF_HIP(hipMemset(d.rowstart, 0, rs_bytes));
  d.status = ctx->d_status;
  d.tune = &ctx->tuning;
  ds->ready = true;
  return VSF_OK;
}

void free_devset(DevSet* ds) {
  hipFree(ds->levels);
  hipFree(ds->units);
  hipFree(ds->blur_mma_units);
  hipFree(ds->blur_mma_units_small);
  hipFree(ds->blur_tcol);
  hipFree(ds->blur_tv);
  hipFree(ds->ic_table);
  hipFree(ds->d.pyr);
  hipFree(ds->d.blur);
  hipFree(ds->d.scratch);
  hipFree(ds->d.lvlkp);
  hipFree(ds->d.lvl_count);
  hipFree(ds->d.cand);
  hipFree(ds->d.rowstart);
  *ds = DevSet();
}

void free_retired(vsf_ctx* ctx) {  // (callers have waited for every stream of the context)
  for (void* p : ctx->retired) hipFree(p);
  ctx->retired.clear();
  for (void* p : ctx->retired_host) hipHostFree(p);
  ctx->retired_host.clear();
}

vsf_status ensure_match_buffers(vsf_ctx* ctx, int pairs, int rows) {
  if (pairs <= ctx->m_pairs && rows <= ctx->m_rows) return VSF_OK;
  pairs = std::max(pairs, ctx->m_pairs);
  rows = std::max(rows, ctx->m_rows);
  vsf_status st = grow_scratch(ctx, ctx->m_idx2, (size_t)pairs * rows * 2 * sizeof(int32_t));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->m_dist2, (size_t)pairs * rows * 2 * sizeof(int32_t));
  if (st != VSF_OK) return st;
  ctx->m_pairs = pairs;
  ctx->m_rows = rows;
  return VSF_OK;
}

vsf_status ensure_match_host_staging(vsf_ctx* ctx, int rows) {  // (host-pointer, synchronous entry points only)
  if (rows <= ctx->mh_rows) return VSF_OK;
  vsf_status st = grow_scratch(ctx, ctx->mh_desc, (size_t)2 * rows * VSF_DESC_BYTES);
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->mh_matches, (size_t)rows * sizeof(vsf_dmatch));
  if (st != VSF_OK) return st;
  if (!ctx->mh_counts) VSF_HIP(hipMalloc((void**)&ctx->mh_counts, 2 * sizeof(int32_t)));
  if (!ctx->mh_nmatches) VSF_HIP(hipMalloc((void**)&ctx->mh_nmatches, sizeof(int32_t)));
  ctx->mh_rows = rows;
  return VSF_OK;
}

vsf_status ensure_residual_buffers(vsf_ctx* ctx, int n_frames) {
  const size_t K = (size_t)ctx->p.max_keypoints;
  if (n_frames > ctx->f_frames) {
    vsf_status st = grow_scratch(ctx, ctx->f_residual, (size_t)n_frames * K * sizeof(float));
    if (st != VSF_OK) return st;
    ctx->f_frames = n_frames;
  }
  return VSF_OK;
}

vsf_status ensure_temporal_buffers(vsf_ctx* ctx, int n_pairs) {
  const size_t K = (size_t)ctx->p.max_keypoints;
  if (n_pairs <= ctx->t_pairs) return VSF_OK;
  vsf_status st = grow_scratch(ctx, ctx->t_matches, (size_t)n_pairs * K * sizeof(vsf_dmatch));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->t_nmatches, (size_t)n_pairs * sizeof(int32_t));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->t_sortkeys, (size_t)n_pairs * K * 8);
  if (st != VSF_OK) return st;
  ctx->t_pairs = n_pairs;
  return VSF_OK;
}

vsf_status ensure_vision_buffers(vsf_ctx* ctx, int n_frames) {
  const size_t K = (size_t)ctx->p.max_keypoints;
  if (n_frames <= ctx->v_frames) return VSF_OK;
  vsf_status st = grow_scratch(ctx, ctx->v_pairs, (size_t)n_frames * K * 2 * sizeof(uint64_t));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->v_npairs, (size_t)n_frames * sizeof(int32_t));
  if (st == VSF_OK) st = grow_scratch(ctx, ctx->v_sets, (size_t)2 * n_frames * sizeof(int32_t));
  if (st != VSF_OK) return st;
  // q_set[f] = 2f + 1 (right frame, "initial", cc:131), t_set[f] = 2f (left frame, "current"): written by a kernel on the
  // context's stream, in order before the launches that read it (a hipMemcpy would wait for the stream)
  vsf_launch_fill_stereo_sets(ctx->v_sets, n_frames, ctx->stream);
  ctx->v_frames = n_frames;
  return VSF_OK;
}

vsf_status ensure_pack_buffers(vsf_ctx* ctx, int n) {
  if (n <= ctx->pk_entries) return VSF_OK;
  vsf_status st = grow_scratch(ctx, ctx->pk_offsets, (size_t)n * sizeof(uint32_t));
  if (st != VSF_OK) return st;
  ctx->pk_entries = n;
  return VSF_OK;
}

vsf_status reserve_scratch(vsf_ctx* ctx, int n_frames, int n_pairs) {
  const int pairs = std::max(n_frames, n_pairs);
  vsf_status st = ensure_match_buffers(ctx, pairs, ctx->p.max_keypoints);
  if (st == VSF_OK) st = ensure_residual_buffers(ctx, n_frames);
  if (st == VSF_OK) st = ensure_temporal_buffers(ctx, pairs);  // (Calculate3DPoints matches one pair per frame)
  if (st == VSF_OK) st = ensure_vision_buffers(ctx, n_frames);
  if (st == VSF_OK) st = ensure_pack_buffers(ctx, n_frames + n_pairs);
  return st;
}

vsf_status check_status_word(vsf_ctx* ctx) {
  VSF_HIP(hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  VSF_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t), ctx->stream));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  VSF_STICKY();
  if (*ctx->h_status & 2) return VSF_ERR_INVALID_ARG;  // a JPEG stream broke off inside its entropy-coded data
  return (*ctx->h_status & 1) ? VSF_ERR_CAPACITY : VSF_OK;
}

vsf_status validate_images(const vsf_ctx* ctx, const uint8_t* d_imgs, int n, size_t image_stride, size_t row_stride) {
  if (!d_imgs || n < 1 || n > ctx->p.max_images) return VSF_ERR_INVALID_ARG;
  if (((uintptr_t)d_imgs & 15) || (image_stride & 15) || (row_stride & 15)) return VSF_ERR_INVALID_ARG;
  if (row_stride < (size_t)ctx->p.width || image_stride < row_stride * (size_t)ctx->p.height) return VSF_ERR_INVALID_ARG;
  return VSF_OK;
}

void prof_fold(vsf_ctx* ctx) {  // stream must be idle
  for (size_t i = 0; i < ctx->ev_used; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]) == hipSuccess) {
      ctx->prof_ms[ctx->ev_stage[i]] += ms;
      ctx->prof_launches[ctx->ev_stage[i]] += ctx->ev_launches[i];
    }
  }
  ctx->ev_used = 0;
}

void sync_all_streams(vsf_ctx* ctx) {  // every stream the context launches on
  if (ctx->stream) vsf_note(hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->own_stream != ctx->stream) vsf_note(hipStreamSynchronize(ctx->own_stream));
  if (ctx->aux_stream) vsf_note(hipStreamSynchronize(ctx->aux_stream));
  if (ctx->pipe_stream) vsf_note(hipStreamSynchronize(ctx->pipe_stream));
  if (ctx->blur_stream) vsf_note(hipStreamSynchronize(ctx->blur_stream));
  for (int i = 1; i < ctx->side.n; i++)
    if (ctx->side.stream[i]) vsf_note(hipStreamSynchronize(ctx->side.stream[i]));
  if (ctx->ob.copy_stream) vsf_note(hipStreamSynchronize(ctx->ob.copy_stream));
  if (ctx->ob.tail_stream) vsf_note(hipStreamSynchronize(ctx->ob.tail_stream));
}

// The per-image work buffers of images [i0, i0 + n) seen as a batch of their own.
VsfDev shifted(const VsfDev& d, const VsfGeom& g, int i0) {
  VsfDev o = d;
  const size_t i = (size_t)i0;
  if (o.pyr) o.pyr += i * g.pyr_bytes;
  if (o.blur) o.blur += i * g.pyr_bytes;
  o.cand += i * g.cand_entries;
  o.rowstart += i * (size_t)g.nunits * VSF_FAST_RS_STRIDE;
  if (o.scratch) o.scratch += i * 6 * g.cand_entries;
  if (o.lvlkp) o.lvlkp += i * g.lvlkp_entries;
  if (o.lvl_count) o.lvl_count += i * g.nlevels;
  return o;
}

// detectAndCompute for images [i0, i0 + n) of `im` on stream `st`.  `status`: the status word the kernels report capacity
// overflows into (the context's, or the word of the frame in flight that owns this extraction).
void extract_on(vsf_ctx* ctx, hipStream_t st, const VsfImages& im_all, int i0, int n, vsf_keypoint* d_kp,
                uint8_t* d_desc, int32_t* d_counts, bool inputs_complete, const VsfSideStream* own_side, int32_t* status,
                int status_stride) {
  const VsfGeom& g = ctx->orb.g;
  VsfDev d = shifted(ctx->dorb.d, g, i0);
  if (status) {
    d.status = status;
    d.status_stride = status_stride;
  }
  VsfImages im = im_all;
  im.base += (size_t)i0 * im.image_stride;
  im.n = n;
  const size_t K = (size_t)ctx->p.max_keypoints;
  const bool pipe = ctx->pipeline && inputs_complete && ctx->lanes == 1 && st == ctx->stream && i0 == 0;
  if (pipe) {
    // The pyramid depends on the input images only.  The caller promised they are complete (vsf_set_pipeline), so the
    // chain goes onto the pipe streams WITHOUT being ordered after this stream's earlier work and overlaps the previous
    // call's later stages; it writes the pyramid buffer the previous call does not use, once the call before that has
    // released it.
    const int buf = ctx->pyr_flip;
    d.pyr = buf ? ctx->pyr_alt : ctx->dorb.d.pyr;
    // (ROCm multiplexes streams onto a few hardware queues; the context's aux stream is known to run beside the
    // main one, so the chain goes there, as a single chain)
    hipStream_t ps = ctx->pipe_stream ? ctx->pipe_stream : ctx->aux_stream;
    if (ctx->pyr_free_valid[buf]) vsf_note(hipStreamWaitEvent(ps, ctx->ev_pyr_free[buf], 0));
    // ... and not before the previous call's FAST kernel has finished: FAST fills every register of the chip, the
    // stages after it (selection, descriptors, matcher) are latency-bound and leave room for the resize chain
    if (ctx->fast_done_valid && ctx->tuning.pipe_after_fast) vsf_note(hipStreamWaitEvent(ps, ctx->ev_fast_done, 0));
    // ... and not before images this library itself is still producing on the context's stream are complete
    if (ctx->ingest_done_valid) vsf_note(hipStreamWaitEvent(ps, ctx->ev_ingest_done, 0));
    // ... nor before the caller's own producer has finished them (vsf_set_input_event)
    if (ctx->input_event) vsf_note(hipStreamWaitEvent(ps, ctx->input_event, 0));
    {
      StageTimer t(ctx, ps, VSF_STAGE_PYRAMID, g.nlevels - 1);
      vsf_launch_pyramid(d, g, ctx->orb.levels.data(), im, ps, nullptr);
    }
    vsf_note(hipEventRecord(ctx->ev_pyr_done, ps));
    vsf_note(hipStreamWaitEvent(st, ctx->ev_pyr_done, 0));
  } else {
    StageTimer t(ctx, st, VSF_STAGE_PYRAMID, g.nlevels - 1);
    // one lane: the aux stream is idle, the pyramid chain of the second half of the batch runs on it
    // (own_side: a caller that runs several extractions at once on different streams brings a side-stream set of its own --
    // the context's fork / join events must not be recorded from two streams at a time)
    vsf_launch_pyramid(d, g, ctx->orb.levels.data(), im, st, own_side ? own_side : (ctx->lanes == 1 ? &ctx->side : nullptr));
  }
  ctx->last_pyr = d.pyr - (size_t)i0 * g.pyr_bytes;
  // The blur needs the pyramid only.  It lives on the matrix cores and on memory bandwidth, FAST on the vector ALU (at its
  // issue ceiling: profiles/r04/valu_ceiling.json) and the selection on latency: forked behind the pyramid onto its own
  // stream, the blur's workgroups fill in as FAST drains and run beside the selection (7.61 -> 7.25 ms per 256-frame step).
  // Measured and left out: forking behind FAST instead (7.52: the overlap with FAST's tail is lost); making room beside
  // FAST with an LDS reservation that caps FAST at four workgroups per CU (7.46: the reservation is LDS the blur needs; the
  // resident kernel below caps FAST without it); a high- or low-priority blur stream (7.39 / 7.65); slices of the blur
  // forked from INSIDE the pyramid's launch chain as soon as their levels exist (the chain's dependent launches stretch from
  // 1.28 to 1.8-2.2 ms beside the blur's memory traffic, 7.37-7.46 ms per step against 7.31); the first 3 / 6 / 10 / 16
  // levels blurred in line in front of FAST and only the rest beside it (7.25-7.36: noise).
  const bool beside_ok = ctx->blur_overlap && im.n >= 32 && ctx->blur_stream;
  // With the blur beside it FAST can run as ONE resident workgroup per CU (k_fast.hip): three waves per SIMD keep 92 % of
  // its own rate and leave the other 224 of a SIMD's 512 registers -- which a grid of one workgroup per four cells fills
  // for as long as cells are left -- to the blur, which then finishes INSIDE the FAST pass instead of after it.  Whether
  // that pays depends on the batch and on how well the blur hides behind the selection anyway (frames per second, grid ->
  // resident; 640x480 / 2000 features: 512 frames per step 35.4 -> 37.5 k, 256 frames 35.0 -> 36.4 k, 128 frames equal,
  // 64 frames 31.3 -> 29.8 k, 32 frames 25.7 -> 24.3 k; 1920x1080 / 8000: 32 frames 4 900 -> 4 700, 96 frames
  // 5 000 -> 5 140, 192 frames 5 030 -> 5 320; four waves per SIMD at 256 VGA frames 7.09-7.12 ms against 7.03, two 8.0).
  // No rule on batch size got all of these right, so the caller may have it MEASURED: vsf_tune_fast_resident (explicit,
  // blocking) times both forms on the caller's own batch and this call takes what it found for this batch size -- the grid
  // form for a size nobody measured.  vsf_set_fast_resident overrides.  Nothing is measured, and nothing waits, in here.
  int resident = 0;
  if (beside_ok && ctx->lanes == 1 && st == ctx->stream && i0 == 0 && !own_side) {
    if (ctx->fast_force >= 0)
      resident = ctx->fast_force;
    else if (ctx->fast_resident >= 0)
      resident = ctx->fast_resident;
    else if (ctx->fast_tune.n == im.n && ctx->fast_tune.choice > 0)
      resident = ctx->fast_tune.choice;
  }
  const bool blur_beside = beside_ok;
  auto launch_blur = [&](hipStream_t bs) {
    StageTimer t(ctx, bs, VSF_STAGE_BLUR, 1);
    if (im.n >= 32)
      vsf_launch_blur_mma(d, g, im, ctx->dorb.blur_mma_units, (int)ctx->orb.blur_mma_units.size(), ctx->dorb.blur_tcol,
                          ctx->dorb.blur_tv, ctx->orb.blur_bias, bs);
    else
      vsf_launch_blur_mma(d, g, im, ctx->dorb.blur_mma_units_small, (int)ctx->orb.blur_mma_units_small.size(),
                          ctx->dorb.blur_tcol, ctx->dorb.blur_tv, ctx->orb.blur_bias, bs);
  };
  auto fork_blur = [&]() {
    vsf_note(hipEventRecord(ctx->ev_blur_fork, st));
    vsf_note(hipStreamWaitEvent(ctx->blur_stream, ctx->ev_blur_fork, 0));
    launch_blur(ctx->blur_stream);
    vsf_note(hipEventRecord(ctx->ev_blur_done, ctx->blur_stream));
  };
  if (blur_beside) fork_blur();
  {
    StageTimer t(ctx, st, VSF_STAGE_FAST, 1);
    vsf_launch_fast(d, g, im, ctx->p.fast_threshold, 1, st, blur_beside ? resident : 0, ctx->n_cus, ctx->fast_cells);
  }
  if (pipe) {
    vsf_note(hipEventRecord(ctx->ev_fast_done, st));
    ctx->fast_done_valid = true;
  }
  {
    StageTimer t(ctx, st, VSF_STAGE_SELECT, 1);
    vsf_launch_select(d, g, ctx->orb.levels.data(), im, st);
  }
  if (blur_beside)
    vsf_note(hipStreamWaitEvent(st, ctx->ev_blur_done, 0));
  else
    launch_blur(st);
  {
    StageTimer t(ctx, st, VSF_STAGE_DESCRIBE, 1);
    vsf_launch_describe(d, g, im, ctx->p.max_keypoints, d_kp + i0 * K, d_desc + i0 * K * VSF_DESC_BYTES, d_counts + i0,
                        st);
  }
  if (pipe) {  // every reader of this pyramid buffer is queued: the call after the next may overwrite it
    const int buf = ctx->pyr_flip;
    vsf_note(hipEventRecord(ctx->ev_pyr_free[buf], st));
    ctx->pyr_free_valid[buf] = true;
    ctx->pyr_flip ^= 1;
  }
}

// knnMatch(k = 2) + ratio test for pairs [p0, p0 + n) on stream `st`.
void match_on(vsf_ctx* ctx, hipStream_t st, const uint8_t* d_desc, const int32_t* d_counts, size_t set_stride,
              const int32_t* d_q_set, const int32_t* d_t_set, int p0, int n, int32_t* d_idx2, int32_t* d_dist2,
              vsf_dmatch* d_matches, int32_t* d_nmatches, int32_t* status) {
  const int rows = ctx->p.max_keypoints;
  const size_t R = (size_t)rows;
  // implicit pairing (set 2p vs 2p + 1) is relative to the descriptor base: shift the base instead of the indices
  const uint8_t* desc = d_desc;
  const int32_t* counts = d_counts;
  if (!d_q_set) {
    desc += (size_t)(2 * p0) * set_stride;
    counts += 2 * p0;
  }
  int32_t* idx2 = d_idx2 + (size_t)p0 * R * 2;
  int32_t* dist2 = d_dist2 + (size_t)p0 * R * 2;
  {
    StageTimer t(ctx, st, VSF_STAGE_KNN2, 1);
    vsf_launch_knn2(desc, counts, set_stride, d_q_set ? d_q_set + p0 : nullptr, d_t_set ? d_t_set + p0 : nullptr, n,
                    rows, idx2, dist2, st);
  }
  {
    StageTimer t(ctx, st, VSF_STAGE_RATIO, 1);
    vsf_launch_ratio_compact(counts, d_q_set ? d_q_set + p0 : nullptr, d_t_set ? d_t_set + p0 : nullptr, n, rows, idx2,
                             dist2, ctx->p.ratio_num, ctx->p.ratio_shift, d_matches + (size_t)p0 * R, d_nmatches + p0,
                             status ? status : ctx->d_status, st);
  }
}

// Opens / closes the second lane: work queued on aux_stream between fork and join is ordered after everything
// already on `stream` and before everything queued on it afterwards.
vsf_status fork_lane(vsf_ctx* ctx) {
  VSF_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
  VSF_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
  return VSF_OK;
}
vsf_status join_lane(vsf_ctx* ctx) {
  VSF_HIP(hipEventRecord(ctx->ev_join, ctx->aux_stream));
  VSF_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return VSF_OK;
}

vsf_status extract_async(vsf_ctx* ctx, const VsfImages& im, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                         bool inputs_complete) {
  vsf_status st = run_chunked(ctx, im.n, [&](hipStream_t s, int i0, int n) {
    extract_on(ctx, s, im, i0, n, d_kp, d_desc, d_counts, inputs_complete);
  });
  if (st != VSF_OK) return st;
  ctx->last_images = im;
  ctx->last_valid = true;
  VSF_STICKY();
  return VSF_OK;
}

// The second pyramid buffer and the events of cross-call pipelining (vsf_set_pipeline, the ObserveImage queue).
vsf_status ensure_pipeline_buffers(vsf_ctx* ctx) {
  if (ctx->pyr_alt) return VSF_OK;
  VSF_HIP(hipMalloc((void**)&ctx->pyr_alt, (size_t)ctx->p.max_images * ctx->orb.g.pyr_bytes));
  VSF_HIP(hipEventCreateWithFlags(&ctx->ev_pyr_done, hipEventDisableTiming));
  VSF_HIP(hipEventCreateWithFlags(&ctx->ev_fast_done, hipEventDisableTiming));
  for (hipEvent_t& e : ctx->ev_pyr_free) VSF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return VSF_OK;
}

}  // namespace vsfi

hipStream_t vsf_ctx_stream(const vsf_ctx* ctx) { return ctx->stream; }
int vsf_ctx_device(const vsf_ctx* ctx) { return ctx->device; }
void vsf_ctx_set_last_error(vsf_ctx* ctx, int code) { ctx->last_hip = code; }
void vsf_ctx_absorb_noted_error(vsf_ctx* ctx) {
  if (ctx->pending_hip == 0) ctx->pending_hip = vsf_tls_hip_error;
  vsf_tls_hip_error = 0;
}

extern "C" {

vsf_status vsf_params_default(vsf_params* p, int width, int height, int max_images) {
  if (!p) return VSF_ERR_INVALID_ARG;
  std::memset(p, 0, sizeof(*p));
  p->nfeatures = 10000;
  p->scale_factor = 1.04f;
  p->nlevels = 50;
  p->edge_threshold = 31;
  p->first_level = 0;
  p->wta_k = 2;
  p->score_type = 0;
  p->patch_size = 31;
  p->fast_threshold = 20;
  p->blur_sse2 = 1;
  p->residual_order = 0;  // Eigen 3.3's a0*b0 + (a1*b1 + a2*b2)
  p->fast_detector_threshold = 10;
  p->fast_detector_nms = 1;
  p->width = width;
  p->height = height;
  p->max_images = max_images;
  p->max_keypoints = 0;
  return vsf_params_set_ratio(p, 0.6f);
}

vsf_status vsf_params_set_ratio(vsf_params* p, float nn_match_ratio) {
  if (!p || !(nn_match_ratio > 0.f) || !(nn_match_ratio < 256.f)) return VSF_ERR_INVALID_ARG;
  // nn_match_ratio widened to double (slam_frontend.cc:523) == num / 2^shift exactly.
  double r = (double)nn_match_ratio;
  uint32_t shift = 0;
  while (r != std::floor(r) && shift < 31) {
    r *= 2.0;
    ++shift;
  }
  if (r != std::floor(r) || r >= 4294967296.0) return VSF_ERR_INVALID_ARG;
  p->ratio_num = (uint32_t)r;
  p->ratio_shift = shift;
  return VSF_OK;
}

const char* vsf_status_string(vsf_status s) {
  switch (s) {
    case VSF_OK: return "ok";
    case VSF_ERR_INVALID_ARG: return "invalid argument";
    case VSF_ERR_CAPACITY: return "capacity exceeded (results truncated)";
    case VSF_ERR_HIP: return "HIP runtime error";
    case VSF_ERR_UNSUPPORTED: return "unsupported parameter combination";
    case VSF_ERR_NO_DEVICE: return "no usable GPU";
  }
  return "unknown";
}

vsf_status vsf_create(const vsf_params* p, int device, vsf_ctx** out) {
  if (!p || !out) return VSF_ERR_INVALID_ARG;
  *out = nullptr;
  if (p->first_level != 0 || p->wta_k != 2 || p->score_type != 0 || p->patch_size != 31) return VSF_ERR_UNSUPPORTED;
  if (p->nlevels < 1 || p->nlevels > VSF_MAX_LEVELS || p->width < 16 || p->height < 16 || p->max_images < 1 ||
      p->nfeatures < 0 || p->edge_threshold < 0 || !(p->scale_factor > 1.0f))
    return VSF_ERR_INVALID_ARG;
  if (p->edge_threshold < 22) return VSF_ERR_UNSUPPORTED;  // borders are not materialised: needs reach 22 <= edge
  {
    const std::vector<int> um = orb_umax(31);
    static const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int i = 0; i < 16; i++)
      if (um[i] != expect[i]) return VSF_ERR_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) return VSF_ERR_NO_DEVICE;
  vsf_ctx* ctx = new (std::nothrow) vsf_ctx();
  if (!ctx) return VSF_ERR_INVALID_ARG;
  ctx->p = *p;
  if (ctx->p.max_keypoints <= 0) ctx->p.max_keypoints = ctx->p.nfeatures + 256;
  ctx->device = device;
  auto fail = [&](vsf_status s) {
    vsf_destroy(ctx);
    return s;
  };
  if (hipSetDevice(device) != hipSuccess) return fail(VSF_ERR_NO_DEVICE);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
  }
  if (!build_geometry(ctx->p, true, true, &ctx->orb)) return fail(VSF_ERR_INVALID_ARG);
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(VSF_ERR_HIP);
  ctx->stream = ctx->own_stream;
  if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->side.fork, hipEventDisableTiming) != hipSuccess)
    return fail(VSF_ERR_HIP);
  if (hipStreamCreateWithFlags(&ctx->blur_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_blur_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_blur_done, hipEventDisableTiming) != hipSuccess)
    return fail(VSF_ERR_HIP);
  ctx->side.stream[0] = ctx->aux_stream;
  for (int i = 0; i < VSF_SIDE_STREAMS; i++) {
    if (i > 0 && hipStreamCreateWithFlags(&ctx->side.stream[i], hipStreamNonBlocking) != hipSuccess)
      return fail(VSF_ERR_HIP);
    if (hipEventCreateWithFlags(&ctx->side.join[i], hipEventDisableTiming) != hipSuccess) return fail(VSF_ERR_HIP);
    ctx->side.n = i + 1;
  }
  if (hipMalloc((void**)&ctx->d_status, 4 * sizeof(int32_t)) != hipSuccess) return fail(VSF_ERR_HIP);
  if (hipMemset(ctx->d_status, 0, 4 * sizeof(int32_t)) != hipSuccess) return fail(VSF_ERR_HIP);
  {
    // Three kernels ask for more dynamic LDS than the default 64 KB (the parallel sort of GetFeatureMatches, the slab
    // pyramid, the parallel JPEG decode): how much a workgroup of this device may have is asked once, their limits are
    // raised here -- checked -- and the launchers fall back to their plain forms where it is not enough.
    int per_block = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) per_block = 0;
    if (hipDeviceGetAttribute(&per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess) per_cu = 0;
    int limit = std::min(std::max(std::max(per_block, per_cu), 64 * 1024), 160 * 1024);
    if (vsf_prepare_sort_kernels(limit) != hipSuccess) {
      (void)hipGetLastError();
      limit = 64 * 1024;  // (nothing above the default was granted: the one-lane sort and the per-level pyramid launches)
    }
    ctx->tuning.lds_limit = limit;
    if (vsf_prepare_pyramid_kernels(limit) != hipSuccess) {
      (void)hipGetLastError();
      ctx->tuning.pyramid_chain = 0;
    }
    if (vsf_prepare_jpeg_kernels(limit) != hipSuccess) {
      (void)hipGetLastError();
      ctx->tuning.jpeg_serial = 1;
    }
  }
  if (hipMalloc((void**)&ctx->fast_cells, 2 * sizeof(uint32_t)) != hipSuccess) return fail(VSF_ERR_HIP);
  if (hipHostMalloc((void**)&ctx->h_status, sizeof(int32_t), hipHostMallocDefault) != hipSuccess)
    return fail(VSF_ERR_HIP);
  vsf_status st = alloc_devset(ctx, ctx->orb, &ctx->dorb, true, ctx->p.max_images);
  if (st != VSF_OK) return fail(st);
  // staging buffers of the host-pointer API (one image / one set of outputs per image slot)
  ctx->st_img_pitch = (size_t)align_up(ctx->p.width, 64);
  ctx->st_img_stride = ctx->st_img_pitch * (size_t)ctx->p.height;
  const size_t n = (size_t)ctx->p.max_images, K = (size_t)ctx->p.max_keypoints;
  if (hipMalloc((void**)&ctx->st_img, n * ctx->st_img_stride) != hipSuccess ||
      hipMalloc((void**)&ctx->st_kp, n * K * sizeof(vsf_keypoint)) != hipSuccess ||
      hipMalloc((void**)&ctx->st_desc, n * K * VSF_DESC_BYTES) != hipSuccess ||
      hipMalloc((void**)&ctx->st_counts, n * sizeof(int32_t)) != hipSuccess)
    return fail(VSF_ERR_HIP);
  // scratch of the batched *_dev calls for whole batches of this context (vsf_reserve sizes it for others)
  if (reserve_scratch(ctx, std::max(1, ctx->p.max_images / 2), std::max(1, ctx->p.max_images / 2)) != VSF_OK)
    return fail(VSF_ERR_HIP);
  if (hipDeviceSynchronize() != hipSuccess) return fail(VSF_ERR_HIP);
  *out = ctx;
  return VSF_OK;
}

void vsf_destroy(vsf_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  stop_observe_threads(ctx);  // (the queue's launcher may be in the middle of a batch)
  // every stream the context ever launched on -- the slots' streams of frames still in flight included: their kernels
  // write device buffers and pinned host memory that is freed below
  sync_all_streams(ctx);
  vsf_tls_hip_error = 0;
  if (ctx->ev_pyr_done) {
    hipEventDestroy(ctx->ev_pyr_done);
    hipEventDestroy(ctx->ev_fast_done);
    for (hipEvent_t e : ctx->ev_pyr_free) hipEventDestroy(e);
  }
  if (ctx->ev_ingest_done) hipEventDestroy(ctx->ev_ingest_done);
  hipFree(ctx->pyr_alt);
  free_devset(&ctx->dorb);
  free_devset(&ctx->dfast);
  hipFree(ctx->d_status);
  hipFree(ctx->fast_cells);
  for (hipEvent_t e : ctx->fast_tune.ev)
    if (e) hipEventDestroy(e);
  if (ctx->h_status) hipHostFree(ctx->h_status);
  hipFree(ctx->st_img);
  hipFree(ctx->st_kp);
  hipFree(ctx->st_desc);
  hipFree(ctx->st_counts);
  hipFree(ctx->m_idx2);
  hipFree(ctx->m_dist2);
  hipFree(ctx->f_residual);
  hipFree(ctx->t_matches);
  hipFree(ctx->t_nmatches);
  hipFree(ctx->t_sortkeys);
  free_observe(ctx);
  hipFree(ctx->jp_flags);
  hipFree(ctx->png_filtered);
  hipFree(ctx->png_file_status);
  hipFree(ctx->jp_clean);
  hipFree(ctx->jp_coef);
  for (int i = 0; i < 2; i++) {
    if (ctx->jp_host[i]) hipHostFree(ctx->jp_host[i]);
    hipFree(ctx->jp_dev[i]);
    if (ctx->jp_copied[i]) hipEventDestroy(ctx->jp_copied[i]);
  }
  free_retired(ctx);
  hipFree(ctx->v_pairs);
  hipFree(ctx->v_npairs);
  hipFree(ctx->v_sets);
  hipFree(ctx->pk_offsets);
  hipFree(ctx->mm_desc);
  hipFree(ctx->mm_counts);
  hipFree(ctx->mm_matches);
  hipFree(ctx->mm_nmatches);
  hipFree(ctx->mh_desc);
  hipFree(ctx->mh_counts);
  hipFree(ctx->mh_matches);
  hipFree(ctx->mh_nmatches);
  for (hipEvent_t e : ctx->ev_pool) hipEventDestroy(e);
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  if (ctx->side.fork) hipEventDestroy(ctx->side.fork);
  for (int i = 0; i < VSF_SIDE_STREAMS; i++) {
    if (ctx->side.join[i]) hipEventDestroy(ctx->side.join[i]);
    if (i > 0 && ctx->side.stream[i]) {
      hipStreamSynchronize(ctx->side.stream[i]);
      hipStreamDestroy(ctx->side.stream[i]);
    }
  }
  if (ctx->aux_stream) hipStreamDestroy(ctx->aux_stream);
  if (ctx->pipe_stream) hipStreamDestroy(ctx->pipe_stream);
  if (ctx->blur_stream) hipStreamDestroy(ctx->blur_stream);
  if (ctx->ev_blur_fork) hipEventDestroy(ctx->ev_blur_fork);
  if (ctx->ev_blur_done) hipEventDestroy(ctx->ev_blur_done);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int vsf_last_hip_error(const vsf_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

vsf_status vsf_get_params(const vsf_ctx* ctx, vsf_params* out) {
  if (!ctx || !out) return VSF_ERR_INVALID_ARG;
  *out = ctx->p;
  return VSF_OK;
}

vsf_status vsf_set_stream(vsf_ctx* ctx, void* hip_stream) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  prof_fold(ctx);
  // (The handle must be a live stream of this device: the HIP runtime of ROCm 7 dereferences a stream handle without
  // looking it up -- hipStreamQuery and the launch path alike crashed on a destroyed one in the round-4 test runs -- so a
  // stale handle cannot be refused here, as with any HIP call that takes a stream.)
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return VSF_OK;
}

vsf_status vsf_set_lanes(vsf_ctx* ctx, int lanes) {
  VsfErrorScope scope_(ctx);
  if (!ctx || lanes < 1 || lanes > 2) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->lanes = lanes;
  return VSF_OK;
}

vsf_status vsf_set_blur_overlap(vsf_ctx* ctx, int on) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->blur_overlap = on ? 1 : 0;
  return VSF_OK;
}

vsf_status vsf_set_fast_resident(vsf_ctx* ctx, int waves) {
  VsfErrorScope scope_(ctx);
  if (!ctx || waves < -1 || waves == 1 || waves > 4) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->fast_resident = waves;
  return VSF_OK;
}

vsf_status vsf_get_fast_resident(const vsf_ctx* ctx, int* waves) {
  if (!ctx || !waves) return VSF_ERR_INVALID_ARG;
  *waves = ctx->fast_resident >= 0 ? ctx->fast_resident : std::max(ctx->fast_tune.choice, 0);
  return VSF_OK;
}

vsf_status vsf_set_option(vsf_ctx* ctx, int option, int value) {
  VsfErrorScope scope_(ctx);
  if (!ctx || option < 0 || option >= VSF_OPT_COUNT) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  VsfTuning& t = ctx->tuning;
  switch (option) {
    case VSF_OPT_FAST_BOTH_MAX:
      if (value < 0) return VSF_ERR_INVALID_ARG;
      t.fast_both_max = value;
      break;
    case VSF_OPT_SELECT_WIDE: t.select_wide = value != 0; break;
    case VSF_OPT_SELECT_BIG_CLASS: t.select_big_class = value != 0; break;
    case VSF_OPT_PIPE_AFTER_FAST: t.pipe_after_fast = value != 0; break;
    case VSF_OPT_OBSERVE_THREAD: t.observe_thread = value != 0; break;  // (read when the queue is built)
    case VSF_OPT_OBSERVE_COPY_THREAD: t.observe_copy_thread = value != 0; break;
    case VSF_OPT_PIPE_PRIORITY:
      if (value < -1 || value > 1) return VSF_ERR_INVALID_ARG;
      t.pipe_priority = value;  // (takes effect with the next vsf_set_pipeline(ctx, 1))
      break;
    case VSF_OPT_PYRAMID_FEW:
      if (value < 0) return VSF_ERR_INVALID_ARG;
      t.pyramid_few = value;
      break;
    case VSF_OPT_PYRAMID_CHAIN:
      if (value < 0 || value > 64) return VSF_ERR_INVALID_ARG;
      if (value > 0 && t.lds_limit < 160 * 1024 - 2048) return VSF_ERR_UNSUPPORTED;
      t.pyramid_chain = value;
      break;
    case VSF_OPT_PYRAMID_ROWS:
      if (value < 1) return VSF_ERR_INVALID_ARG;
      t.pyramid_rows = value;
      break;
    case VSF_OPT_PYRAMID_TAIL_MIN:
      if (value < 0) return VSF_ERR_INVALID_ARG;
      t.pyramid_tail_min = value;
      break;
    default: return VSF_ERR_INVALID_ARG;
  }
  return VSF_OK;
}

vsf_status vsf_get_option(const vsf_ctx* ctx, int option, int* value) {
  if (!ctx || !value) return VSF_ERR_INVALID_ARG;
  const VsfTuning& t = ctx->tuning;
  switch (option) {
    case VSF_OPT_FAST_BOTH_MAX: *value = t.fast_both_max; break;
    case VSF_OPT_SELECT_WIDE: *value = t.select_wide; break;
    case VSF_OPT_SELECT_BIG_CLASS: *value = t.select_big_class; break;
    case VSF_OPT_PIPE_AFTER_FAST: *value = t.pipe_after_fast; break;
    case VSF_OPT_OBSERVE_THREAD: *value = t.observe_thread; break;
    case VSF_OPT_OBSERVE_COPY_THREAD: *value = t.observe_copy_thread; break;
    case VSF_OPT_PIPE_PRIORITY: *value = t.pipe_priority; break;
    case VSF_OPT_PYRAMID_FEW: *value = t.pyramid_few; break;
    case VSF_OPT_PYRAMID_CHAIN: *value = t.pyramid_chain; break;
    case VSF_OPT_PYRAMID_ROWS: *value = t.pyramid_rows; break;
    case VSF_OPT_PYRAMID_TAIL_MIN: *value = t.pyramid_tail_min; break;
    default: return VSF_ERR_INVALID_ARG;
  }
  return VSF_OK;
}

vsf_status vsf_set_pipeline(vsf_ctx* ctx, int on) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  VSF_HIP(hipStreamSynchronize(ctx->aux_stream));  // (the pipelined pyramid chain runs there ...
  if (ctx->pipe_stream) VSF_HIP(hipStreamSynchronize(ctx->pipe_stream));  // ... or on a stream of its own priority)
  if (on && ctx->tuning.pipe_priority != ctx->pipe_stream_priority) {
    if (ctx->pipe_stream) VSF_HIP(hipStreamDestroy(ctx->pipe_stream));
    ctx->pipe_stream = nullptr;
    ctx->pipe_stream_priority = 0;
    if (ctx->tuning.pipe_priority != 0) {
      int least = 0, greatest = 0;
      VSF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      VSF_HIP(hipStreamCreateWithPriority(&ctx->pipe_stream, hipStreamNonBlocking,
                                          ctx->tuning.pipe_priority > 0 ? least : greatest));
      ctx->pipe_stream_priority = ctx->tuning.pipe_priority;
    }
  }
  if (on) {
    const vsf_status st = ensure_pipeline_buffers(ctx);
    if (st != VSF_OK) return st;
  }
  ctx->pipeline = on != 0;
  ctx->pyr_free_valid[0] = ctx->pyr_free_valid[1] = false;
  ctx->fast_done_valid = false;
  return VSF_OK;
}

vsf_status vsf_sync(vsf_ctx* ctx) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  const vsf_status st = check_status_word(ctx);
  if (!ctx->retired.empty() || !ctx->retired_host.empty()) {  // scratch a *_dev call outgrew: nothing can be using it once every stream is idle
    sync_all_streams(ctx);
    free_retired(ctx);
  }
  return st;
}

vsf_status vsf_set_input_event(vsf_ctx* ctx, void* hip_event) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  ctx->input_event = static_cast<hipEvent_t>(hip_event);
  return VSF_OK;
}

vsf_status vsf_reserve(vsf_ctx* ctx, int n_frames, int n_pairs) {
  VsfErrorScope scope_(ctx);
  if (!ctx || n_frames < 1 || n_pairs < 0) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  sync_all_streams(ctx);
  free_retired(ctx);
  vsf_status st = reserve_scratch(ctx, n_frames, n_pairs);
  if (st != VSF_OK) return st;
  sync_all_streams(ctx);  // (the set-index fill)
  free_retired(ctx);
  VSF_STICKY();
  return VSF_OK;
}

vsf_status vsf_level_info(const vsf_ctx* ctx, int level, int* w, int* h, float* scale, int* nfeatures) {
  if (!ctx || level < 0 || level >= ctx->orb.g.nlevels) return VSF_ERR_INVALID_ARG;
  const VsfLevel& L = ctx->orb.levels[level];
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (scale) *scale = L.scale;
  if (nfeatures) *nfeatures = L.nfeatures;
  return VSF_OK;
}

uint64_t vsf_pyramid_pixels(const vsf_ctx* ctx) { return ctx ? ctx->orb.g.pyramid_pixels : 0; }

uint64_t vsf_algorithmic_bytes_per_image(const vsf_ctx* ctx) {
  if (!ctx) return 0;
  // resize reads + resize writes + FAST read + blur read/write + outputs.
  const auto& Ls = ctx->orb.levels;
  uint64_t P = 0, rd = 0, wr = 0;
  for (size_t l = 0; l < Ls.size(); l++) {
    const uint64_t px = (uint64_t)Ls[l].w * Ls[l].h;
    P += px;
    if (l + 1 < Ls.size()) rd += px;
    if (l >= 1) wr += px;
  }
  return rd + wr + P + 2 * P + (uint64_t)ctx->p.nfeatures * (sizeof(vsf_keypoint) + VSF_DESC_BYTES);
}

}  // extern "C"
