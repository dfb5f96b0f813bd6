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
  uint8_t bpp;           // the filters' pixel distance in bytes: 1 (gray / palette <= 8 bit), 2 (gray 16, gray + alpha 8), 3, 4, 6, 8
  uint8_t depth;         // bits per sample: 1, 2, 4, 8 or 16
  uint8_t kind;          // kGray: the first byte of a pixel is the answer; kPalette: it indexes a table of 256 gray values;
                         // kRgb8 / kRgb16: red, green, blue (and alpha, unused) samples weighted as libpng's rgb_to_gray does
  uint8_t flags;         // kGammaTables (kRgb8: the file's gamma matters -- `table` holds 256 bytes "to linear" and 256 "from
                         // linear"), kAdam7 (the scanlines come in seven passes)
  uint32_t piece_first;  // the file's IDAT payloads: entries [piece_first, piece_first + piece_count) of the upload's list of
  uint32_t piece_count;  // their END offsets inside the zlib stream (libpng hands zlib at most 8192 bytes of ONE chunk at a time)
  uint32_t table;        // kPalette / gamma_tables: offset of the file's table(s) in the upload's table region
  uint32_t expected;     // bytes the zlib stream must deliver: every scanline of every pass with its filter byte
};
constexpr uint8_t kGray = 0, kPalette = 1, kRgb8 = 2, kRgb16 = 3;
constexpr uint8_t kGammaTables = 1, kAdam7 = 2;
// Adam7 (PNG specification 8.2): pass p holds the pixels (x0 + i dx, y0 + j dy)
struct Adam7Pass {
  int x0, y0, dx, dy;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline Adam7Pass adam7_pass(int p) {
  const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
  return Adam7Pass{x0[p], y0[p], dx[p], dy[p]};
}
static_assert(sizeof(DevImage) == 32, "DevImage layout");
constexpr int kDevImageWords = 8;
constexpr uint32_t kIdatReadSize = 8192;  // PNG_IDAT_READ_SIZE (= PNG_ZBUF_SIZE): bytes of a chunk libpng feeds zlib per refill

constexpr int kWindow = 32768;        // deflate's history
constexpr int kFlushChunk = 16384;    // bytes the inflate kernel moves from its LDS window to HBM at a time

}  // namespace vsf_png
#endif  // VSF_PNG_HOST_H_
