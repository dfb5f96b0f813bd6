// vsf_png_host.cc -- host half of vsf_png_decode_gray_batch (cv::imdecode(IMREAD_GRAYSCALE) for PNG files,
// slam_frontend_main.cc:98-100): walks the chunks of every file (PNG specification, 2nd edition, section 5), checks their
// CRCs as libpng does (a damaged critical chunk makes png_read_* fail and cv::imdecode return an empty image; a damaged
// ancillary chunk is skipped), and lays out ONE upload: the image descriptors and the zlib streams (IDAT payloads in file
// order).  Inflating and unfiltering happen on the device (k_png.hip).  The bytes come from a ROS bag or a network topic,
// i.e. they are UNTRUSTED: every length is checked against the file's end before it is used.  Plain C++ (no HIP code), so
// that the same translation unit builds with -fsanitize=address,undefined (make asan).
//
// Built: grayscale files (colour type 0 at 1, 2, 4, 8, 16 bits, colour type 4 at 8 and 16 bits), non-interlaced -- what a
// camera driver's image_transport writes for mono8 / mono16 / bayer topics.  A gray read of those needs no colour
// arithmetic: 16-bit samples keep their high byte (png_set_strip_16), alpha is dropped (png_set_strip_alpha), 1 / 2 / 4-bit
// samples are replicated to 8 bits (png_set_expand_gray_1_2_4_to_8) -- grfmt_png.cpp's settings for IMREAD_GRAYSCALE.
// Colour and palette files (libpng's rgb_to_gray, which depends on the file's gamma chunks) and Adam7 files return
// VSF_ERR_UNSUPPORTED.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "vsf_internal.h"
#include "vsf_png_host.h"

using namespace vsf_png;

namespace {

struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
const CrcTables kCrc;

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {  // slicing by 8; crc is the running (inverted) register
  while (n >= 8) {
    uint32_t a, b;
    std::memcpy(&a, p, 4);
    std::memcpy(&b, p + 4, 4);
    a ^= crc;
    crc = kCrc.t[7][a & 0xFF] ^ kCrc.t[6][(a >> 8) & 0xFF] ^ kCrc.t[5][(a >> 16) & 0xFF] ^ kCrc.t[4][a >> 24] ^
          kCrc.t[3][b & 0xFF] ^ kCrc.t[2][(b >> 8) & 0xFF] ^ kCrc.t[1][(b >> 16) & 0xFF] ^ kCrc.t[0][b >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) crc = kCrc.t[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return crc;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const uint8_t kSignature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

struct Piece {
  uint32_t off, len;
};

vsf_status parse_png(const uint8_t* f, size_t n, int width, int height, DevImage* im, std::vector<Piece>* pieces) {
  pieces->clear();
  if (n < 8 + 25 + 12 || std::memcmp(f, kSignature, 8) != 0) return VSF_ERR_INVALID_ARG;
  size_t pos = 8;
  bool have_ihdr = false, have_idat = false, idat_run_over = false, have_iend = false;
  int channels = 1;
  uint64_t stream_len = 0;
  while (!have_iend) {
    if (n - pos < 12) return VSF_ERR_INVALID_ARG;
    const uint32_t len = be32(f + pos);
    if (len > 0x7FFFFFFFu || (size_t)len > n - pos - 12) return VSF_ERR_INVALID_ARG;
    const uint8_t* type = f + pos + 4;
    const uint8_t* data = f + pos + 8;
    const bool critical = (type[0] & 0x20) == 0;
    const bool crc_ok = (crc32_update(0xFFFFFFFFu, type, (size_t)len + 4) ^ 0xFFFFFFFFu) == be32(data + len);
    pos += 12 + (size_t)len;
    if (!crc_ok) {
      if (critical) return VSF_ERR_INVALID_ARG;  // png_crc_error: png_chunk_error for critical chunks
      continue;                                  // ancillary: a warning, the chunk is skipped
    }
    if (!have_ihdr) {
      if (std::memcmp(type, "IHDR", 4) != 0 || len != 13) return VSF_ERR_INVALID_ARG;
      have_ihdr = true;
      const uint32_t w = be32(data), h = be32(data + 4);
      const int depth = data[8], ctype = data[9], comp = data[10], filt = data[11], lace = data[12];
      if (w == 0 || h == 0 || w > 0x7FFFFFFFu || h > 0x7FFFFFFFu || comp != 0 || filt != 0 || lace > 1) return VSF_ERR_INVALID_ARG;
      const bool depth_ok = ctype == 0   ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                            : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                            : (ctype == 2 || ctype == 4 || ctype == 6) ? (depth == 8 || depth == 16)
                                                                       : false;
      if (!depth_ok) return VSF_ERR_INVALID_ARG;
      if (w != (uint32_t)width || h != (uint32_t)height) return VSF_ERR_INVALID_ARG;
      if (ctype == 2 || ctype == 3 || ctype == 6 || lace == 1) return VSF_ERR_UNSUPPORTED;
      channels = ctype == 4 ? 2 : 1;
      const uint64_t row_bits = (uint64_t)w * (uint64_t)(depth * channels);
      const uint64_t row_bytes = (row_bits + 7) / 8;
      if (row_bytes > 0x0FFFFFFFu || (row_bytes + 1) * (uint64_t)h > 0xF0000000u) return VSF_ERR_INVALID_ARG;
      im->row_bytes = (uint32_t)row_bytes;
      im->bpp = (uint8_t)std::max(1, depth * channels / 8);
      im->depth = (uint8_t)depth;
      im->pad_[0] = im->pad_[1] = 0;
      im->pad2_[0] = im->pad2_[1] = 0;
      continue;
    }
    if (std::memcmp(type, "IHDR", 4) == 0) return VSF_ERR_INVALID_ARG;
    if (std::memcmp(type, "IDAT", 4) == 0) {
      if (idat_run_over) return VSF_ERR_INVALID_ARG;  // (IDAT chunks must follow one another)
      have_idat = true;
      if (len > 0) {
        pieces->push_back(Piece{(uint32_t)(data - f), len});
        stream_len += len;
        if (stream_len > 0x40000000u) return VSF_ERR_INVALID_ARG;
      }
      continue;
    }
    if (have_idat) idat_run_over = true;
    if (std::memcmp(type, "IEND", 4) == 0) {
      have_iend = true;
      continue;
    }
    if (std::memcmp(type, "PLTE", 4) == 0) continue;  // ("ignored in grayscale PNG": a benign error, a warning on read)
    if (critical) return VSF_ERR_INVALID_ARG;          // png_handle_unknown: unhandled critical chunk
  }
  if (!have_idat || stream_len < 6) return VSF_ERR_INVALID_ARG;
  // the zlib header (RFC 1950): deflate with a window of at most 32 KiB, no preset dictionary, check bits
  uint8_t hdr[2];
  {
    size_t k = 0;
    for (const Piece& p : *pieces)
      for (uint32_t i = 0; i < p.len && k < 2; i++) hdr[k++] = f[p.off + i];
  }
  if ((hdr[0] & 0x0F) != 8 || (hdr[0] >> 4) > 7 || (hdr[1] & 0x20) != 0 || (((uint32_t)hdr[0] << 8) | hdr[1]) % 31 != 0)
    return VSF_ERR_INVALID_ARG;
  im->stream_len = (uint32_t)stream_len;
  return VSF_OK;
}

}  // namespace

// Step 1: every file's chunks (CRC pass included, a few threads when there is enough of it).
vsf_status vsf_png_plan(const uint8_t* const* png, const size_t* nbytes, int n, int width, int height, VsfPngPlan* plan) {
  std::vector<DevImage> images((size_t)n);
  std::vector<std::vector<Piece>> pieces((size_t)n);
  std::vector<vsf_status> status((size_t)n, VSF_OK);
  auto parse_range = [&](int i0, int i1) {
    for (int i = i0; i < i1; i++) {
      std::memset(&images[i], 0, sizeof(DevImage));
      status[i] = parse_png(png[i], nbytes[i], width, height, &images[i], &pieces[i]);
    }
  };
  size_t all = 0;
  for (int i = 0; i < n; i++) all += nbytes[i];
  const int workers = (int)std::min<size_t>({(size_t)4, all >> 21, (size_t)n, (size_t)std::max(1u, std::thread::hardware_concurrency())});
  if (workers <= 1) {
    parse_range(0, n);
  } else {
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; w++) pool.emplace_back(parse_range, (int)((int64_t)n * w / workers), (int)((int64_t)n * (w + 1) / workers));
    parse_range(0, n / workers);
    for (auto& th : pool) th.join();
  }
  for (int i = 0; i < n; i++)
    if (status[i] != VSF_OK) return status[i];  // (the first file in error decides, as in a loop over the files)
  size_t stream_bytes = 0;
  uint32_t max_filtered = 0;
  plan->piece_first.assign((size_t)n + 1, 0);
  plan->piece_off.clear();
  plan->piece_len.clear();
  plan->stream_off.resize((size_t)n);
  plan->stream_len.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    images[i].stream_off = (uint32_t)stream_bytes;
    plan->stream_off[i] = images[i].stream_off;
    plan->stream_len[i] = images[i].stream_len;
    stream_bytes += (images[i].stream_len + 3u + 32u) & ~(size_t)3;
    if (stream_bytes > 0xF0000000u) return VSF_ERR_INVALID_ARG;
    for (const Piece& p : pieces[i]) {
      plan->piece_off.push_back(p.off);
      plan->piece_len.push_back(p.len);
    }
    plan->piece_first[i + 1] = (uint32_t)plan->piece_off.size();
    max_filtered = std::max(max_filtered, (images[i].row_bytes + 1) * (uint32_t)height);
  }
  // the end of every IDAT payload inside its file's zlib stream (the device finds libpng's refill boundaries from them)
  std::vector<uint32_t> piece_end(plan->piece_len.size());
  for (int i = 0; i < n; i++) {
    uint32_t acc = 0;
    for (uint32_t p = plan->piece_first[i]; p < plan->piece_first[i + 1]; p++) {
      acc += plan->piece_len[p];
      piece_end[p] = acc;
    }
    images[i].piece_first = plan->piece_first[i];
    images[i].piece_count = plan->piece_first[i + 1] - plan->piece_first[i];
  }
  plan->filtered_stride = ((size_t)max_filtered + 15 + 16) & ~(size_t)15;
  plan->off_images = 0;
  plan->off_pieces = (images.size() * sizeof(DevImage) + 15) & ~(size_t)15;
  plan->off_stream = (plan->off_pieces + piece_end.size() * sizeof(uint32_t) + 15) & ~(size_t)15;
  plan->total = plan->off_stream + stream_bytes + 16;
  plan->head.assign(plan->off_stream, 0);
  std::memcpy(plan->head.data(), images.data(), images.size() * sizeof(DevImage));
  if (!piece_end.empty()) std::memcpy(plan->head.data() + plan->off_pieces, piece_end.data(), piece_end.size() * sizeof(uint32_t));
  return VSF_OK;
}

// Step 2: writes the upload into `dst` (pinned staging, plan->total bytes).
void vsf_png_fill(const VsfPngPlan& plan, const uint8_t* const* png, int n, uint8_t* dst) {
  std::memcpy(dst, plan.head.data(), plan.head.size());
  auto copy_range = [&](int i0, int i1) {
    for (int i = i0; i < i1; i++) {
      uint8_t* d = dst + plan.off_stream + plan.stream_off[i];
      size_t k = 0;
      for (uint32_t p = plan.piece_first[i]; p < plan.piece_first[i + 1]; p++) {
        std::memcpy(d + k, png[i] + plan.piece_off[p], plan.piece_len[p]);
        k += plan.piece_len[p];
      }
      const size_t padded = (plan.stream_len[i] + 3u + 32u) & ~(size_t)3;
      std::memset(d + k, 0, padded - k);
    }
  };
  const size_t stream_bytes = plan.total - plan.off_stream;
  const int workers = (int)std::min<size_t>({(size_t)4, stream_bytes >> 22, (size_t)n, (size_t)std::max(1u, std::thread::hardware_concurrency())});
  if (workers <= 1) {
    copy_range(0, n);
  } else {
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; w++) pool.emplace_back(copy_range, (int)((int64_t)n * w / workers), (int)((int64_t)n * (w + 1) / workers));
    copy_range(0, n / workers);
    for (auto& th : pool) th.join();
  }
  std::memset(dst + plan.total - 16, 0, 16);
}
