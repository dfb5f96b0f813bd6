// vsf_png_host.h -- what the host half (vsf_png_host.cc: chunks, CRCs, the upload plan; plain C++, parses UNTRUSTED bytes,
// built with AddressSanitizer by `make asan`) and the device half (k_png.hip) of the PNG decoder share.
#ifndef VSF_PNG_HOST_H_
#define VSF_PNG_HOST_H_

#include <stdint.h>

namespace vsf_png {

struct DevImage {
  uint32_t stream_off;   // the zlib stream (the IDAT payloads in file order) inside the packed stream buffer, 4-byte aligned
  uint32_t stream_len;
  uint32_t row_bytes;    // bytes of one filtered row without its filter-type byte
  uint8_t bpp;           // the filters' pixel distance in bytes: 1 (gray <= 8 bit), 2 (gray 16, gray + alpha 8), 4 (gray + alpha 16)
  uint8_t depth;         // bits per sample: 1, 2, 4, 8 or 16
  uint8_t pad_[2];
  uint32_t piece_first;  // the file's IDAT payloads: entries [piece_first, piece_first + piece_count) of the upload's list of
  uint32_t piece_count;  // their END offsets inside the zlib stream (libpng hands zlib at most 8192 bytes of ONE chunk at a time)
  uint32_t pad2_[2];
};
static_assert(sizeof(DevImage) == 32, "DevImage layout");
constexpr int kDevImageWords = 8;
constexpr uint32_t kIdatReadSize = 8192;  // PNG_IDAT_READ_SIZE (= PNG_ZBUF_SIZE): bytes of a chunk libpng feeds zlib per refill

constexpr int kWindow = 32768;        // deflate's history
constexpr int kFlushChunk = 16384;    // bytes the inflate kernel moves from its LDS window to HBM at a time

}  // namespace vsf_png
#endif  // VSF_PNG_HOST_H_
