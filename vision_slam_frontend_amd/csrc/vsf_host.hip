// vsf_host.hip -- the host-pointer, synchronous entry points: one call == one reference call (ExtractFeatures,
// slam_frontend.cc:266-280; GetMatches, slam_frontend.cc:521-538).
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

// ---------------- host-pointer entry points ----------------

static vsf_status upload_image(vsf_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, int slot) {
  if (!img || w != ctx->p.width || h != ctx->p.height || stride < (size_t)w) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipMemcpy2DAsync(ctx->st_img + (size_t)slot * ctx->st_img_stride, ctx->st_img_pitch, img, stride, (size_t)w,
                           (size_t)h, hipMemcpyHostToDevice, ctx->stream));
  return VSF_OK;
}

vsf_status vsf_extract(vsf_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, vsf_keypoint* kp_out,
                       uint8_t* desc_out, int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || cap < 0 || (cap > 0 && (!kp_out || !desc_out))) return VSF_ERR_INVALID_ARG;
  *n_out = 0;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_status st = upload_image(ctx, img, w, h, stride, 0);
  if (st != VSF_OK) return st;
  VsfImages im{ctx->st_img, ctx->st_img_stride, ctx->st_img_pitch, 1};
  st = extract_async(ctx, im, ctx->st_kp, ctx->st_desc, ctx->st_counts);
  if (st != VSF_OK) return st;
  int32_t n = 0;
  VSF_HIP(hipMemcpyAsync(&n, ctx->st_counts, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  st = check_status_word(ctx);  // synchronises
  *n_out = n;
  const int m = std::min(std::min((int)n, cap), ctx->p.max_keypoints);
  if (m > 0) {
    VSF_HIP(hipMemcpy(kp_out, ctx->st_kp, (size_t)m * sizeof(vsf_keypoint), hipMemcpyDeviceToHost));
    VSF_HIP(hipMemcpy(desc_out, ctx->st_desc, (size_t)m * VSF_DESC_BYTES, hipMemcpyDeviceToHost));
  }
  if (st == VSF_OK && n > m) st = VSF_ERR_CAPACITY;
  return st;
}

vsf_status vsf_extract_pair(vsf_ctx* ctx, const uint8_t* img0, const uint8_t* img1, int w, int h, size_t stride,
                            vsf_keypoint* kp0, uint8_t* desc0, int* n0, vsf_keypoint* kp1, uint8_t* desc1, int* n1,
                            int cap) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n0 || !n1 || cap < 0 || (cap > 0 && (!kp0 || !desc0 || !kp1 || !desc1))) return VSF_ERR_INVALID_ARG;
  *n0 = *n1 = 0;
  if (ctx->p.max_images < 2) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_status st = upload_image(ctx, img0, w, h, stride, 0);
  if (st == VSF_OK) st = upload_image(ctx, img1, w, h, stride, 1);
  if (st != VSF_OK) return st;
  VsfImages im{ctx->st_img, ctx->st_img_stride, ctx->st_img_pitch, 2};
  st = extract_async(ctx, im, ctx->st_kp, ctx->st_desc, ctx->st_counts);
  if (st != VSF_OK) return st;
  int32_t n[2] = {0, 0};
  VSF_HIP(hipMemcpyAsync(n, ctx->st_counts, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  st = check_status_word(ctx);  // synchronises
  *n0 = n[0];
  *n1 = n[1];
  const size_t K = (size_t)ctx->p.max_keypoints;
  vsf_keypoint* kps[2] = {kp0, kp1};
  uint8_t* descs[2] = {desc0, desc1};
  for (int i = 0; i < 2; i++) {
    const int m = std::min(std::min((int)n[i], cap), ctx->p.max_keypoints);
    if (m > 0) {
      VSF_HIP(hipMemcpy(kps[i], ctx->st_kp + i * K, (size_t)m * sizeof(vsf_keypoint), hipMemcpyDeviceToHost));
      VSF_HIP(hipMemcpy(descs[i], ctx->st_desc + i * K * VSF_DESC_BYTES, (size_t)m * VSF_DESC_BYTES,
                        hipMemcpyDeviceToHost));
    }
    if (st == VSF_OK && n[i] > m) st = VSF_ERR_CAPACITY;
  }
  return st;
}

vsf_status vsf_fast_detect(vsf_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, int threshold, int nms,
                           vsf_keypoint* kp_out, int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || cap < 0 || (cap > 0 && !kp_out)) return VSF_ERR_INVALID_ARG;
  *n_out = 0;
  VSF_HIP(hipSetDevice(ctx->device));
  if (threshold < 0) {
    threshold = ctx->p.fast_detector_threshold;
  }
  threshold = std::min(std::max(threshold, 0), 255);
  const bool want_nms = nms != 0;
  // The candidate segments are sized for the NMS case (no two 8-adjacent keypoints); without NMS every pixel
  // may be a corner, so that geometry is (re)built with full-density segments.
  if (!ctx->dfast.ready || ctx->fast_nms != want_nms) {
    VSF_HIP(hipStreamSynchronize(ctx->stream));
    free_devset(&ctx->dfast);
    ctx->fast = Geometry();
    if (!build_geometry(ctx->p, false, want_nms, &ctx->fast)) return VSF_ERR_INVALID_ARG;
    vsf_status st0 = alloc_devset(ctx, ctx->fast, &ctx->dfast, false, 1);
    if (st0 != VSF_OK) return st0;
    ctx->fast_nms = want_nms;
  }
  vsf_status st = upload_image(ctx, img, w, h, stride, 0);
  if (st != VSF_OK) return st;
  VsfImages im{ctx->st_img, ctx->st_img_stride, ctx->st_img_pitch, 1};
  // Output capacity: grow a private buffer if the caller's cap exceeds the extract staging.
  const int kcap = ctx->p.max_keypoints;
  vsf_launch_fast(ctx->dfast.d, ctx->fast.g, im, threshold, want_nms ? 1 : 0, ctx->stream);
  vsf_keypoint* d_out = ctx->st_kp;
  vsf_keypoint* big = nullptr;
  int outcap = kcap;
  if (cap > kcap) {
    VSF_HIP(hipMalloc((void**)&big, (size_t)cap * sizeof(vsf_keypoint)));
    d_out = big;
    outcap = cap;
  }
  vsf_launch_fast_emit(ctx->dfast.d, ctx->fast.g, 1, outcap, d_out, ctx->st_counts, ctx->stream);
  int32_t n = 0;
  hipError_t e = hipMemcpyAsync(&n, ctx->st_counts, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) {
    st = check_status_word(ctx);
    if (st == VSF_ERR_HIP) e = (hipError_t)ctx->last_hip;
  }
  if (e == hipSuccess) {
    *n_out = n;
    const int m = std::min(std::min((int)n, cap), outcap);
    if (m > 0) e = hipMemcpy(kp_out, d_out, (size_t)m * sizeof(vsf_keypoint), hipMemcpyDeviceToHost);
    if (e == hipSuccess) st = n > m ? VSF_ERR_CAPACITY : VSF_OK;  // the status word only reflects `outcap`
  }
  if (big) hipFree(big);
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    return VSF_ERR_HIP;
  }
  return st;
}

static vsf_status match_host(vsf_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2,
                             int32_t* dist2, vsf_dmatch* out, int cap, int* n_out) {
  if (nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || nt >= (1 << 20)) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  if (n_out) *n_out = 0;
  if (nq == 0) return VSF_OK;
  const int rows = std::max(std::max(nq, nt), 1);
  vsf_status st = ensure_match_host_staging(ctx, rows);
  if (st != VSF_OK) return st;
  st = ensure_match_buffers(ctx, 1, ctx->mh_rows);
  if (st != VSF_OK) return st;
  const int R = ctx->mh_rows;
  const size_t set_stride = (size_t)R * VSF_DESC_BYTES;
  const int32_t counts[2] = {nq, nt};
  VSF_HIP(hipMemcpyAsync(ctx->mh_counts, counts, sizeof(counts), hipMemcpyHostToDevice, ctx->stream));
  VSF_HIP(hipMemcpyAsync(ctx->mh_desc, q, (size_t)nq * VSF_DESC_BYTES, hipMemcpyHostToDevice, ctx->stream));
  if (nt > 0)
    VSF_HIP(hipMemcpyAsync(ctx->mh_desc + set_stride, t, (size_t)nt * VSF_DESC_BYTES, hipMemcpyHostToDevice,
                           ctx->stream));
  // m_idx2/m_dist2 are laid out [pair][m_rows][2]; the kernels are given the same row capacity.
  vsf_launch_knn2(ctx->mh_desc, ctx->mh_counts, set_stride, nullptr, nullptr, 1, R, ctx->m_idx2, ctx->m_dist2,
                  ctx->stream);
  if (out) {
    vsf_launch_ratio_compact(ctx->mh_counts, nullptr, nullptr, 1, R, ctx->m_idx2, ctx->m_dist2, ctx->p.ratio_num,
                             ctx->p.ratio_shift, ctx->mh_matches, ctx->mh_nmatches, ctx->d_status, ctx->stream);
  }
  VSF_STICKY();
  if (idx2) {
    VSF_HIP(hipMemcpyAsync(idx2, ctx->m_idx2, (size_t)nq * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    VSF_HIP(hipMemcpyAsync(dist2, ctx->m_dist2, (size_t)nq * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  int32_t nm = 0;
  if (out) VSF_HIP(hipMemcpyAsync(&nm, ctx->mh_nmatches, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  if (out) {
    if (n_out) *n_out = nm;
    const int m = std::min((int)nm, cap);
    if (m > 0) VSF_HIP(hipMemcpy(out, ctx->mh_matches, (size_t)m * sizeof(vsf_dmatch), hipMemcpyDeviceToHost));
    if (nm > cap) return VSF_ERR_CAPACITY;
  }
  return VSF_OK;
}

vsf_status vsf_get_matches_multi(vsf_ctx* ctx, const uint8_t* const* q, const int* nq, int n_sets, const uint8_t* t,
                                 int nt, vsf_dmatch* out, int cap_per_set, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || n_sets < 1 || !q || !nq || !n_out || cap_per_set < 0 || (cap_per_set > 0 && !out) || nt < 0 ||
      (nt > 0 && !t) || nt >= (1 << 20))
    return VSF_ERR_INVALID_ARG;
  int rows = std::max(nt, 1);
  for (int s = 0; s < n_sets; s++) {
    if (nq[s] < 0 || (nq[s] > 0 && !q[s])) return VSF_ERR_INVALID_ARG;
    rows = std::max(rows, nq[s]);
    n_out[s] = 0;
  }
  VSF_HIP(hipSetDevice(ctx->device));
  if (n_sets > ctx->mm_sets || rows > ctx->mm_rows) {
    const int S = std::max(n_sets, ctx->mm_sets), R = std::max(rows, ctx->mm_rows);
    VSF_HIP(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->mm_desc);
    hipFree(ctx->mm_counts);
    hipFree(ctx->mm_matches);
    hipFree(ctx->mm_nmatches);
    ctx->mm_desc = nullptr;
    ctx->mm_counts = nullptr;
    ctx->mm_matches = nullptr;
    ctx->mm_nmatches = nullptr;
    ctx->mm_sets = ctx->mm_rows = 0;
    VSF_HIP(hipMalloc((void**)&ctx->mm_desc, (size_t)(S + 1) * R * VSF_DESC_BYTES));
    VSF_HIP(hipMalloc((void**)&ctx->mm_counts, (size_t)(3 * S + 1) * sizeof(int32_t)));
    VSF_HIP(hipMalloc((void**)&ctx->mm_matches, (size_t)S * R * sizeof(vsf_dmatch)));
    VSF_HIP(hipMalloc((void**)&ctx->mm_nmatches, (size_t)S * sizeof(int32_t)));
    ctx->mm_sets = S;
    ctx->mm_rows = R;
  }
  const int S = n_sets, R = ctx->mm_rows;
  vsf_status st = ensure_match_buffers(ctx, S, R);
  if (st != VSF_OK) return st;
  const size_t set_stride = (size_t)R * VSF_DESC_BYTES;
  std::vector<int32_t> meta((size_t)3 * S + 1);
  for (int s = 0; s < S; s++) {
    meta[s] = nq[s];
    meta[S + 1 + s] = s;      // q_set
    meta[2 * S + 1 + s] = S;  // t_set: the one train set
    if (nq[s] > 0)
      VSF_HIP(hipMemcpyAsync(ctx->mm_desc + s * set_stride, q[s], (size_t)nq[s] * VSF_DESC_BYTES, hipMemcpyHostToDevice,
                             ctx->stream));
  }
  meta[S] = nt;
  if (nt > 0)
    VSF_HIP(hipMemcpyAsync(ctx->mm_desc + S * set_stride, t, (size_t)nt * VSF_DESC_BYTES, hipMemcpyHostToDevice,
                           ctx->stream));
  VSF_HIP(hipMemcpyAsync(ctx->mm_counts, meta.data(), meta.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  const int32_t* d_q = ctx->mm_counts + S + 1;
  const int32_t* d_t = ctx->mm_counts + 2 * S + 1;
  // every buffer of this call is laid out with row capacity R: m_idx2 / m_dist2 [S][R][2] (they hold at least
  // m_pairs x m_rows >= S x R entries), mm_matches [S][R]
  vsf_launch_knn2(ctx->mm_desc, ctx->mm_counts, set_stride, d_q, d_t, S, R, ctx->m_idx2, ctx->m_dist2,
                  ctx->stream);
  vsf_launch_ratio_compact(ctx->mm_counts, d_q, d_t, S, R, ctx->m_idx2, ctx->m_dist2, ctx->p.ratio_num,
                           ctx->p.ratio_shift, ctx->mm_matches, ctx->mm_nmatches, ctx->d_status, ctx->stream);
  VSF_STICKY();
  std::vector<int32_t> nm(S);
  VSF_HIP(hipMemcpyAsync(nm.data(), ctx->mm_nmatches, (size_t)S * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  st = VSF_OK;
  for (int s = 0; s < S; s++) {
    n_out[s] = nm[s];
    const int m = std::min((int)nm[s], cap_per_set);
    if (m > 0)
      VSF_HIP(hipMemcpyAsync(out + (size_t)s * cap_per_set, ctx->mm_matches + (size_t)s * R,
                             (size_t)m * sizeof(vsf_dmatch), hipMemcpyDeviceToHost, ctx->stream));
    if (nm[s] > cap_per_set) st = VSF_ERR_CAPACITY;
  }
  VSF_HIP(hipStreamSynchronize(ctx->stream));
  return st;
}

vsf_status vsf_knn2_hamming(vsf_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2,
                            int32_t* dist2) {
  VsfErrorScope scope_(ctx);
  if (!ctx || (nq > 0 && (!idx2 || !dist2))) return VSF_ERR_INVALID_ARG;
  return match_host(ctx, q, nq, t, nt, idx2, dist2, nullptr, 0, nullptr);
}

vsf_status vsf_get_matches(vsf_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, vsf_dmatch* out,
                           int cap, int* n_out) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !n_out || cap < 0 || (cap > 0 && !out)) return VSF_ERR_INVALID_ARG;
  vsf_dmatch dummy;
  return match_host(ctx, q, nq, t, nt, nullptr, nullptr, out ? out : &dummy, cap, n_out);
}

}  // extern "C"
