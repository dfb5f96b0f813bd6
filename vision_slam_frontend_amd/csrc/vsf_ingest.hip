// vsf_ingest.hip -- SURVEY section 8(f) row f4: the entry points of the image ingest (slam_frontend_main.cc:98-109):
// cv::imdecode(IMREAD_GRAYSCALE) for JPEG and PNG payloads, COLOR_BayerBG2BGR + COLOR_BGR2GRAY.
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

namespace {

// The pinned staging buffer + its device copy of slot b, grown without waiting for the GPU: the old pair goes to the
// context's retired lists (freed by vsf_sync / vsf_destroy once every stream is idle).
vsf_status grow_ingest_staging(vsf_ctx* ctx, int b, size_t cap) {
  void* host = nullptr;
  VSF_HIP(hipHostMalloc(&host, cap, hipHostMallocDefault));
  void* dev = nullptr;
  if (hipMalloc(&dev, cap) != hipSuccess) {
    hipHostFree(host);
    (void)hipGetLastError();
    return VSF_ERR_HIP;
  }
  if (ctx->jp_host[b]) ctx->retired_host.push_back(ctx->jp_host[b]);
  if (ctx->jp_dev[b]) ctx->retired.push_back(ctx->jp_dev[b]);
  ctx->jp_host[b] = static_cast<uint8_t*>(host);
  ctx->jp_dev[b] = static_cast<uint8_t*>(dev);
  ctx->jp_cap[b] = cap;
  return VSF_OK;
}

}  // namespace

extern "C" {

vsf_status vsf_bayer_bg_to_gray_batch_dev(vsf_ctx* ctx, const uint8_t* d_src, int n_images, int width, int height,
                                          size_t src_image_stride, size_t src_row_stride, uint8_t* d_dst,
                                          size_t dst_image_stride, size_t dst_row_stride) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !d_src || !d_dst || n_images < 1 || width < 1 || height < 1 || width > 16384 || height > 65535 ||
      n_images > 65535)
    return VSF_ERR_INVALID_ARG;
  if (((uintptr_t)d_src & 3) || ((uintptr_t)d_dst & 3) || (src_image_stride & 3) || (src_row_stride & 3) ||
      (dst_image_stride & 3) || (dst_row_stride & 3) || src_row_stride < (size_t)width ||
      dst_row_stride < (size_t)((width + 3) & ~3) || src_row_stride > 0x7FFFFFFF || dst_row_stride > 0x7FFFFFFF ||
      src_image_stride < src_row_stride * (size_t)height || dst_image_stride < dst_row_stride * (size_t)height)
    return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  vsf_launch_bayer_bg_gray(d_src, n_images, width, height, src_image_stride, (int)src_row_stride, d_dst,
                           dst_image_stride, (int)dst_row_stride, ctx->stream);
  VSF_STICKY();
  // a pipelined extract that follows (vsf_set_pipeline) builds its pyramid off this stream: give it something to wait for
  if (!ctx->ev_ingest_done) VSF_HIP(hipEventCreateWithFlags(&ctx->ev_ingest_done, hipEventDisableTiming));
  VSF_HIP(hipEventRecord(ctx->ev_ingest_done, ctx->stream));
  ctx->ingest_done_valid = true;
  return VSF_OK;
}

vsf_status vsf_jpeg_decode_gray_batch(vsf_ctx* ctx, const uint8_t* const* jpeg, const size_t* nbytes, int n_images,
                                      int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                      size_t dst_row_stride) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !jpeg || !nbytes || n_images < 1 || n_images > 65535 || width < 1 || height < 1 || width > 65535 ||
      height > 65535 || !d_dst)
    return VSF_ERR_INVALID_ARG;
  if (((uintptr_t)d_dst & 3) || (dst_image_stride & 3) || (dst_row_stride & 3) || dst_row_stride < (size_t)width ||
      dst_row_stride > 0x7FFFFFFF || dst_image_stride < dst_row_stride * (size_t)height)
    return VSF_ERR_INVALID_ARG;
  for (int i = 0; i < n_images; i++)
    if (!jpeg[i] || nbytes[i] < 4 || nbytes[i] > 0x40000000u) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VsfJpegPlan plan;
  vsf_status st = vsf_jpeg_plan(jpeg, nbytes, n_images, width, height, ctx->tuning.jpeg_serial != 0, &plan);
  if (st != VSF_OK) return st;
  const int b = ctx->jp_flip;
  ctx->jp_flip ^= 1;
  if (!ctx->jp_copied[b]) VSF_HIP(hipEventCreateWithFlags(&ctx->jp_copied[b], hipEventDisableTiming));
  if (plan.total > ctx->jp_cap[b]) {
    // (no wait for the GPU: the outgrown pair is retired -- an upload or a decode already queued may still be using it --
    // and released by the next vsf_sync, like every other scratch a *_dev call outgrows)
    const vsf_status gs = grow_ingest_staging(ctx, b, plan.total + plan.total / 4 + 4096);
    if (gs != VSF_OK) return gs;
  } else {
    // the upload of the call before the previous one has left this staging buffer (long ago: the previous call's
    // decode is what may still be running, out of the OTHER buffer)
    VSF_HIP(hipEventSynchronize(ctx->jp_copied[b]));
  }
  // files without restart intervals (what a camera driver writes): self-synchronising parallel decode; it needs the
  // de-stuffed streams and the luminance coefficients in HBM
  const size_t coef_stride = (size_t)plan.max_luma_blocks * 64 * sizeof(int16_t);
  if (plan.n_par + plan.n_prog > 0) {
    const size_t clean_need = plan.n_par > 0 ? vsf_jpeg_clean_bytes(plan.total - plan.off_stream, plan.n_par) : 0,
                 coef_need = (size_t)(plan.n_par + plan.n_prog) * coef_stride + vsf_jpeg_prog_huff_bytes(plan.n_prog_huff);
    // (the expanded Huffman tables of progressive scans live behind the coefficients)
    if (clean_need > ctx->jp_clean_cap) {  // (no wait: outgrown buffers are retired)
      const vsf_status gs = grow_scratch(ctx, ctx->jp_clean, clean_need + clean_need / 4);
      if (gs != VSF_OK) return gs;
      ctx->jp_clean_cap = clean_need + clean_need / 4;
    }
    if (coef_need > ctx->jp_coef_cap) {
      const vsf_status gs = grow_scratch(ctx, ctx->jp_coef, coef_need + coef_need / 4);
      if (gs != VSF_OK) return gs;
      ctx->jp_coef_cap = coef_need + coef_need / 4;
    }
  }
  vsf_jpeg_fill(plan, jpeg, n_images, ctx->jp_host[b]);  // the one pass over the compressed bytes on the host
  VSF_HIP(hipMemcpyAsync(ctx->jp_dev[b], ctx->jp_host[b], plan.total, hipMemcpyHostToDevice, ctx->stream));
  VSF_HIP(hipEventRecord(ctx->jp_copied[b], ctx->stream));
  if (plan.n_prog > ctx->jp_flags_cap) {  // (no wait: the outgrown buffer is retired)
    vsf_status gs = grow_scratch(ctx, ctx->jp_flags, (size_t)plan.n_prog * sizeof(int32_t));
    if (gs != VSF_OK) return gs;
    ctx->jp_flags_cap = plan.n_prog;
  }
  vsf_launch_jpeg_decode(ctx->jp_dev[b], plan.off_images, plan.off_index, plan.off_tables, plan.off_scans, plan.off_prog_huff,
                         plan.off_stream, plan.total, plan.n_par, plan.n_prog, plan.n_prog_huff,
                         reinterpret_cast<uint8_t*>(ctx->jp_coef) + (size_t)(plan.n_par + plan.n_prog) * coef_stride,
                         n_images - plan.n_par - plan.n_prog, plan.max_luma_blocks, plan.max_slots, width, height, ctx->jp_clean, ctx->jp_coef,
                         coef_stride, d_dst, dst_image_stride, (int)dst_row_stride, ctx->d_status, ctx->stream,
                         ctx->tuning.jpeg_serial != 0, ctx->jp_flags);
  VSF_STICKY();
  if (!ctx->ev_ingest_done) VSF_HIP(hipEventCreateWithFlags(&ctx->ev_ingest_done, hipEventDisableTiming));
  VSF_HIP(hipEventRecord(ctx->ev_ingest_done, ctx->stream));  // (a pipelined extract waits for its images, as after the Bayer step)
  ctx->ingest_done_valid = true;
  return VSF_OK;
}

// cv::imdecode(IMREAD_GRAYSCALE) for grayscale PNG files (slam_frontend_main.cc:99-100): chunks and CRCs on the host, inflate +
// filters on the device (k_png.hip).  Same staging and the same asynchronous contract as the JPEG entry point.
vsf_status vsf_png_decode_gray_batch(vsf_ctx* ctx, const uint8_t* const* png, const size_t* nbytes, int n_images,
                                     int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                     size_t dst_row_stride) {
  VsfErrorScope scope_(ctx);
  if (!ctx || !png || !nbytes || n_images < 1 || n_images > 65535 || width < 1 || height < 1 || width > 65535 ||
      height > 65535 || !d_dst)
    return VSF_ERR_INVALID_ARG;
  if (((uintptr_t)d_dst & 3) || (dst_image_stride & 3) || (dst_row_stride & 3) || dst_row_stride < (size_t)width ||
      dst_row_stride > 0x7FFFFFFF || dst_image_stride < dst_row_stride * (size_t)height)
    return VSF_ERR_INVALID_ARG;
  for (int i = 0; i < n_images; i++)
    if (!png[i] || nbytes[i] < 8 || nbytes[i] > 0x40000000u) return VSF_ERR_INVALID_ARG;
  VSF_HIP(hipSetDevice(ctx->device));
  VsfPngPlan plan;
  vsf_status st = vsf_png_plan(png, nbytes, n_images, width, height, &plan);
  if (st != VSF_OK) return st;
  const int b = ctx->jp_flip;
  ctx->jp_flip ^= 1;
  if (!ctx->jp_copied[b]) VSF_HIP(hipEventCreateWithFlags(&ctx->jp_copied[b], hipEventDisableTiming));
  if (plan.total > ctx->jp_cap[b]) {
    // (no wait for the GPU: the outgrown pair is retired -- an upload or a decode already queued may still be using it --
    // and released by the next vsf_sync, like every other scratch a *_dev call outgrows)
    const vsf_status gs = grow_ingest_staging(ctx, b, plan.total + plan.total / 4 + 4096);
    if (gs != VSF_OK) return gs;
  } else {
    VSF_HIP(hipEventSynchronize(ctx->jp_copied[b]));  // (the upload of the call before the previous one has left this buffer)
  }
  const size_t filtered_need = plan.filtered_stride * (size_t)n_images;
  if (filtered_need > ctx->png_filtered_cap) {  // (no wait: the outgrown buffer is retired)
    vsf_status gs = grow_scratch(ctx, ctx->png_filtered, filtered_need + filtered_need / 4);
    if (gs != VSF_OK) return gs;
    ctx->png_filtered_cap = filtered_need + filtered_need / 4;
  }
  if (n_images > ctx->png_file_status_cap) {
    vsf_status gs = grow_scratch(ctx, ctx->png_file_status, (size_t)n_images * sizeof(int32_t));
    if (gs != VSF_OK) return gs;
    ctx->png_file_status_cap = n_images;
  }
  vsf_png_fill(plan, png, n_images, ctx->jp_host[b]);
  VSF_HIP(hipMemcpyAsync(ctx->jp_dev[b], ctx->jp_host[b], plan.total, hipMemcpyHostToDevice, ctx->stream));
  VSF_HIP(hipEventRecord(ctx->jp_copied[b], ctx->stream));
  vsf_launch_png_decode(ctx->jp_dev[b], plan.off_images, plan.off_pieces, plan.off_tables, plan.off_stream, n_images, width, height, ctx->png_filtered,
                        plan.filtered_stride, ctx->png_file_status, d_dst, dst_image_stride, (int)dst_row_stride,
                        ctx->d_status, plan.any_general, plan.any_rgb, ctx->stream);
  VSF_STICKY();
  if (!ctx->ev_ingest_done) VSF_HIP(hipEventCreateWithFlags(&ctx->ev_ingest_done, hipEventDisableTiming));
  VSF_HIP(hipEventRecord(ctx->ev_ingest_done, ctx->stream));
  ctx->ingest_done_valid = true;
  return VSF_OK;
}

// cv::imdecode(msg.data, IMREAD_GRAYSCALE) as the reference calls it (slam_frontend_main.cc:99-100): whatever the payload
// is.  Files are told apart by their first bytes (as cv::imdecode's findDecoder does: signature match) and handed, run by run
// of one format, to the JPEG or the PNG entry point; image i lands at d_dst + i * dst_image_stride either way.
vsf_status vsf_imdecode_gray_batch(vsf_ctx* ctx, const uint8_t* const* files, const size_t* nbytes, int n_images,
                                   int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                   size_t dst_row_stride) {
  if (!ctx || !files || !nbytes || n_images < 1 || !d_dst) return VSF_ERR_INVALID_ARG;
  static const uint8_t kPng[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  auto kind = [&](int i) -> int {  // 0 JPEG (SOI), 1 PNG, -1 neither
    if (!files[i]) return -1;
    if (nbytes[i] >= 8 && std::memcmp(files[i], kPng, 8) == 0) return 1;
    if (nbytes[i] >= 3 && files[i][0] == 0xFF && files[i][1] == 0xD8 && files[i][2] == 0xFF) return 0;
    return -1;
  };
  for (int i = 0; i < n_images; i++)
    if (kind(i) < 0) return VSF_ERR_UNSUPPORTED;  // (imdecode's other formats -- BMP, TIFF, WebP ... -- are not built)
  for (int i0 = 0; i0 < n_images;) {
    const int k = kind(i0);
    int i1 = i0 + 1;
    while (i1 < n_images && kind(i1) == k) ++i1;
    uint8_t* dst = d_dst + (size_t)i0 * dst_image_stride;
    const vsf_status st = k == 1 ? vsf_png_decode_gray_batch(ctx, files + i0, nbytes + i0, i1 - i0, width, height, dst, dst_image_stride, dst_row_stride)
                                 : vsf_jpeg_decode_gray_batch(ctx, files + i0, nbytes + i0, i1 - i0, width, height, dst, dst_image_stride, dst_row_stride);
    if (st != VSF_OK) return st;
    i0 = i1;
  }
  return VSF_OK;
}

}  // extern "C"
