// vsf_debug.hip -- introspection for kernel-level parity tests and the roofline model: per-stage timers, level images,
// candidate / keypoint dumps, the selection and sort test hooks, the error-plumbing hook.
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

// Test hook: makes the context's thread behave as if a launcher had just noted HIP error `code` (vsf_note): the next entry
// point that launches must return VSF_ERR_HIP with that code, and the one after it must work again.
vsf_status vsf_debug_inject_hip_error(vsf_ctx* ctx, int code) {
  VsfErrorScope scope_(ctx);
  if (!ctx || code <= 0) return VSF_ERR_INVALID_ARG;
  vsf_note((hipError_t)code);
  return VSF_OK;
}

vsf_status vsf_debug_jpeg_serial(vsf_ctx* ctx, int on) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  if (!on && vsf_prepare_jpeg_kernels(ctx->tuning.lds_limit) != hipSuccess) {  // (the parallel decoder's LDS was refused)
    (void)hipGetLastError();
    return VSF_ERR_UNSUPPORTED;
  }
  ctx->tuning.jpeg_serial = on != 0;
  return VSF_OK;
}

vsf_status vsf_profile_enable(vsf_ctx* ctx, int on) {
  VsfErrorScope scope_(ctx);
  if (!ctx) return VSF_ERR_INVALID_ARG;
  sync_all_streams(ctx);
  prof_fold(ctx);
  ctx->prof_on = on != 0;
  return VSF_OK;
}

vsf_status vsf_profile_read(vsf_ctx* ctx, double* ms_total, int64_t* launches, int reset) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !ms_total || !launches) return VSF_ERR_INVALID_ARG;
  sync_all_streams(ctx);
  prof_fold(ctx);
  for (int i = 0; i < VSF_STAGE_COUNT; i++) {
    ms_total[i] = ctx->prof_ms[i];
    launches[i] = ctx->prof_launches[i];
    if (reset) {
      ctx->prof_ms[i] = 0;
      ctx->prof_launches[i] = 0;
    }
  }
  return VSF_OK;
}

const char* vsf_stage_name(int stage) {
  static const char* names[VSF_STAGE_COUNT] = {"pyramid_resize", "fast_score_nms", "select_harris_angle", "gauss_blur7",
                                               "orb_describe",   "hamming_knn2",   "ratio_compact", "frontend_tail"};
  return (stage >= 0 && stage < VSF_STAGE_COUNT) ? names[stage] : "?";
}

// ---------------- introspection ----------------

vsf_status vsf_debug_retain_best(vsf_ctx* ctx, uint32_t* key_bits, uint32_t* ids, int n, int n_points, int use_lds,
                                 int mode, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || n < 0 || (n > 0 && (!key_bits || !ids))) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  std::vector<uint2> h((size_t)std::max(n, 1));
  for (int i = 0; i < n; i++) h[i] = make_uint2(key_bits[i], ids[i]);
  uint2* d = nullptr;
  uint32_t* dt = nullptr;
  int* dn = nullptr;
  VSF_HIP(hipMalloc((void**)&d, h.size() * sizeof(uint2)));
  VSF_HIP(hipMalloc((void**)&dt, 2 * h.size() * sizeof(uint32_t)));
  VSF_HIP(hipMalloc((void**)&dn, sizeof(int)));
  VSF_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(uint2), hipMemcpyHostToDevice));
  vsf_launch_retain_best_test(d, dt, n, n_points, use_lds, mode, dn, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(h.data(), d, h.size() * sizeof(uint2), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(n_out, dn, sizeof(int), hipMemcpyDeviceToHost);
  hipFree(d);
  hipFree(dt);
  hipFree(dn);
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    return VSF_ERR_HIP;
  }
  for (int i = 0; i < n; i++) {
    key_bits[i] = h[i].x;
    ids[i] = h[i].y;
  }
  return VSF_OK;
}

vsf_status vsf_debug_sort_trim(vsf_ctx* ctx, const vsf_dmatch* matches, int n_lists, int n, float best_percent,
                               int serial, uint64_t* pairs_out, int32_t* counts_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !matches || !pairs_out || !counts_out || n_lists < 1 || n < 0 || n > ctx->p.max_keypoints)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  const size_t K = (size_t)ctx->p.max_keypoints;
  vsf_dmatch* dm = nullptr;
  int32_t *dn = nullptr, *dc = nullptr;
  uint64_t* dp = nullptr;
  void* dscratch = nullptr;
  hipError_t e = hipMalloc((void**)&dm, (size_t)n_lists * K * sizeof(vsf_dmatch));
  if (e == hipSuccess) e = hipMalloc((void**)&dn, (size_t)n_lists * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&dc, (size_t)n_lists * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&dp, (size_t)n_lists * K * 2 * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMalloc(&dscratch, (size_t)n_lists * K * 8);
  std::vector<int32_t> hn((size_t)n_lists, n);
  if (e == hipSuccess) e = hipMemcpy(dn, hn.data(), hn.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  for (int i = 0; i < n_lists && e == hipSuccess && n > 0; i++)
    e = hipMemcpy(dm + (size_t)i * K, matches + (size_t)i * n, (size_t)n * sizeof(vsf_dmatch), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vsf_launch_sort_trim(dm, dn, n_lists, (int)K, best_percent, nullptr, dscratch, dp, dc, ctx->stream, serial != 0,
                         ctx->tuning.lds_limit);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpy(counts_out, dc, (size_t)n_lists * sizeof(int32_t), hipMemcpyDeviceToHost);
  for (int i = 0; i < n_lists && e == hipSuccess && n > 0; i++)
    e = hipMemcpy(pairs_out + (size_t)i * n * 2, dp + (size_t)i * K * 2, (size_t)n * 2 * sizeof(uint64_t),
                  hipMemcpyDeviceToHost);
  hipFree(dm);
  hipFree(dn);
  hipFree(dc);
  hipFree(dp);
  hipFree(dscratch);
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    return VSF_ERR_HIP;
  }
  return VSF_OK;
}

vsf_status vsf_debug_level_image(vsf_ctx* ctx, int image, int level, int blurred, uint8_t* out, size_t ostride) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !out || !ctx->last_valid || image < 0 || image >= ctx->last_images.n || level < 0 ||
      level >= ctx->orb.g.nlevels)
    return VSF_ERR_INVALID_ARG;
  const VsfLevel& L = ctx->orb.levels[level];
  if (ostride < (size_t)L.w) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  const uint8_t* src;
  size_t pitch;
  if (!blurred && level == 0) {
    src = ctx->last_images.base + (size_t)image * ctx->last_images.image_stride;
    pitch = ctx->last_images.row_stride;
  } else {
    src = (blurred ? ctx->dorb.d.blur : (ctx->last_pyr ? ctx->last_pyr : ctx->dorb.d.pyr)) +
          (size_t)image * ctx->orb.g.pyr_bytes + L.offset;
    pitch = (size_t)L.pitch;
  }
  if (blurred) {  // stored in tiles (VSF_BLUR_TILE_OFFSET)
    std::vector<uint8_t> tiled((size_t)L.pitch * align_up(L.h, 8));
    VSF_HIP(hipMemcpy(tiled.data(), src, tiled.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; y++)
      for (int x = 0; x < L.w; x++) out[(size_t)y * ostride + x] = tiled[VSF_BLUR_TILE_OFFSET(L.pitch, x, y)];
    return VSF_OK;
  }
  VSF_HIP(hipMemcpy2D(out, ostride, src, pitch, (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
  return VSF_OK;
}

vsf_status vsf_debug_fast_candidates(vsf_ctx* ctx, int image, int level, vsf_keypoint* kp_out, int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || !ctx->last_valid || image < 0 || image >= ctx->last_images.n || level < 0 ||
      level >= ctx->orb.g.nlevels)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  const VsfLevel& L = ctx->orb.levels[level];
  const VsfGeom& g = ctx->orb.g;
  // Merge the unit segments (unit-local raster order + per-row starts) into the level's raster order.
  const int nu = L.nstrips * L.nbands;
  int n = 0;
  if (nu > 0) {
    std::vector<uint16_t> rs((size_t)nu * VSF_FAST_RS_STRIDE);
    VSF_HIP(hipMemcpy(rs.data(), ctx->dorb.d.rowstart + ((size_t)image * g.nunits + L.unit0) * VSF_FAST_RS_STRIDE,
                      rs.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> seg((size_t)nu * L.seg_cap);
    VSF_HIP(hipMemcpy(seg.data(), ctx->dorb.d.cand + (size_t)image * g.cand_entries + L.cand_offset,
                      seg.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int row = 0; row < L.y_hi - L.y_lo; row++) {
      const int s = row / VSF_FAST_STRIP_ROWS, r = row % VSF_FAST_STRIP_ROWS;
      for (int b = 0; b < L.nbands; b++) {
        const int u = s * L.nbands + b;
        const uint16_t* urs = rs.data() + (size_t)u * VSF_FAST_RS_STRIDE;
        for (int e = urs[r]; e < urs[r + 1]; e++, n++) {
          if (n < cap && kp_out) {
            const uint32_t cd = seg[(size_t)u * L.seg_cap + e];
            kp_out[n] =
                vsf_keypoint{(float)VSF_CAND_X(cd), (float)VSF_CAND_Y(cd), 7.f, -1.f, (float)VSF_CAND_SCORE(cd), 0, -1};
          }
        }
      }
    }
  }
  *n_out = n;
  return VSF_OK;
}

vsf_status vsf_debug_level_keypoints(vsf_ctx* ctx, int image, int level, vsf_keypoint* kp_out, int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || !ctx->last_valid || image < 0 || image >= ctx->last_images.n || level < 0 ||
      level >= ctx->orb.g.nlevels)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  const VsfLevel& L = ctx->orb.levels[level];
  const VsfGeom& g = ctx->orb.g;
  int32_t n = 0;
  VSF_HIP(hipMemcpy(&n, ctx->dorb.d.lvl_count + (size_t)image * g.nlevels + level, sizeof(int32_t),
                    hipMemcpyDeviceToHost));
  std::vector<VsfLevelKp> v(std::max(n, 1));
  if (n > 0)
    VSF_HIP(hipMemcpy(v.data(), ctx->dorb.d.lvlkp + (size_t)image * g.lvlkp_entries + L.kp_offset,
                      (size_t)n * sizeof(VsfLevelKp), hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < cap && kp_out; i++)
    kp_out[i] = vsf_keypoint{(float)(v[i].xy & 0xFFFu), (float)(v[i].xy >> 12), 31 * L.scale, v[i].angle,
                             v[i].response, level, -1};
  *n_out = n;
  return VSF_OK;
}

}  // extern "C"
