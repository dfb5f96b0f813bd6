// vsf_jpeg_host_check.cc -- entry point of the sanitizer build of the JPEG host half (make asan): runs a batch of files
// through vsf_jpeg_plan + vsf_jpeg_fill exactly as vsf_jpeg_decode_gray_batch does before its upload, into a heap buffer
// of exactly plan.total bytes (so that AddressSanitizer sees any write past the planned layout).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vsf_internal.h"

extern "C" int vsf_jpeg_host_check(const uint8_t* const* jpeg, const size_t* nbytes, int n, int width, int height,
                                   int force_serial, uint64_t* total_out, uint32_t* checksum_out) {
  VsfJpegPlan plan;
  const vsf_status st = vsf_jpeg_plan(jpeg, nbytes, n, width, height, force_serial != 0, &plan);
  if (total_out) *total_out = 0;
  if (checksum_out) *checksum_out = 0;
  if (st != VSF_OK) return (int)st;
  std::vector<uint8_t> blob(plan.total);
  vsf_jpeg_fill(plan, jpeg, n, blob.data());
  // (self-test of the harness: with this variable set the function writes one byte past its buffer, which the sanitizer
  // must catch -- tests/test_jpeg_host_asan.py checks that it does, i.e. that the instrumentation is live)
  if (std::getenv("VSF_ASAN_SELFTEST")) {
    volatile uint8_t* past = blob.data() + blob.size();
    *past = 1;
  }
  uint32_t sum = 0;
  for (uint8_t b : blob) sum = sum * 16777619u ^ b;  // (every byte of the upload is read once)
  if (total_out) *total_out = plan.total;
  if (checksum_out) *checksum_out = sum;
  return (int)VSF_OK;
}

// The same for the PNG host half (vsf_png_host.cc): vsf_png_plan + vsf_png_fill into a buffer of exactly plan.total bytes.
extern "C" int vsf_png_host_check(const uint8_t* const* png, const size_t* nbytes, int n, int width, int height,
                                  uint64_t* total_out, uint32_t* checksum_out) {
  VsfPngPlan plan;
  const vsf_status st = vsf_png_plan(png, nbytes, n, width, height, &plan);
  if (total_out) *total_out = 0;
  if (checksum_out) *checksum_out = 0;
  if (st != VSF_OK) return (int)st;
  std::vector<uint8_t> blob(plan.total);
  vsf_png_fill(plan, png, n, blob.data());
  uint32_t sum = 0;
  for (uint8_t b : blob) sum = sum * 16777619u ^ b;
  if (total_out) *total_out = plan.total;
  if (checksum_out) *checksum_out = sum;
  return (int)VSF_OK;
}
