// vsf_png_host.cc -- host half of vsf_png_decode_gray_batch (cv::imdecode(IMREAD_GRAYSCALE) for PNG files,
// slam_frontend_main.cc:98-100): walks the chunks of every file (PNG specification, 2nd edition, section 5), checks their
// CRCs as libpng does (a damaged critical chunk makes png_read_* fail and cv::imdecode return an empty image; a damaged
// ancillary chunk is skipped), and lays out ONE upload: the image descriptors and the zlib streams (IDAT payloads in file
// order).  Inflating and unfiltering happen on the device (k_png.hip).  The bytes come from a ROS bag or a network topic,
// i.e. they are UNTRUSTED: every length is checked against the file's end before it is used.  Plain C++ (no HIP code), so
// that the same translation unit builds with -fsanitize=address,undefined (make asan).
//
// Built: every colour type, interlaced (Adam7: seven passes, each a small image with filter bytes of its own) or not.  Grayscale files (colour type 0 at 1, 2, 4, 8, 16 bits, colour type 4 at 8 and 16
// bits) are what a camera driver's image_transport writes for mono8 / mono16 / bayer topics; a gray read of those needs no
// colour arithmetic: 16-bit samples keep their high byte (png_set_strip_16), alpha is dropped (png_set_strip_alpha),
// 1 / 2 / 4-bit samples are replicated to 8 bits (png_set_expand_gray_1_2_4_to_8) -- grfmt_png.cpp's settings for
// IMREAD_GRAYSCALE.  Colour files (types 2, 6) and palette files (type 3) go through libpng's rgb_to_gray with the
// coefficients grfmt_png.cpp passes (0.299, 0.587 -> 9797, 19234, 3737 of 32768), restated here and in k_png.hip from
// pngrtran.c (png_do_rgb_to_gray, png_build_gamma_table): the integer weighted sum, truncated for 8-bit samples and rounded
// for 16-bit ones -- or, when the file says its samples are not linear (a gAMA chunk outside 0.95 .. 1.05, or sRGB), the sum of
// the LINEARISED samples mapped back, through two 256-entry tables built as libpng builds them.  What that restatement does
// not cover returns VSF_ERR_UNSUPPORTED: 16-bit colour with such a gamma, iCCP, more than one gAMA / sRGB, one out of
// range, primaries (cHRM) other than sRGB's beside a gamma chunk.  (sRGB wins over a gAMA beside it, as in libpng.)
#include <algorithm>
#include <cmath>
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

// ---- libpng's gamma arithmetic (png.c: png_reciprocal, png_gamma_significant, png_gamma_8bit_correct, png_build_8bit_table)
int32_t png_reciprocal(int32_t a) {
  const double r = std::floor(1E10 / a + .5);
  return (r <= 2147483647. && r >= -2147483648.) ? (int32_t)r : 0;
}
bool png_gamma_significant(int32_t g) { return g < 95000 || g > 105000; }  // PNG_FP_1 -+ PNG_GAMMA_THRESHOLD_FIXED
void png_build_8bit_table(uint8_t* table, int32_t gamma_val) {
  for (int i = 0; i < 256; i++) table[i] = (uint8_t)i;
  if (png_gamma_significant(gamma_val))
    for (int i = 1; i < 255; i++) table[i] = (uint8_t)std::floor(255 * std::pow(i / 255., gamma_val * .00001) + .5);
}
// png_do_rgb_to_gray for one 8-bit pixel: the coefficients of png_set_rgb_to_gray(1, 0.299, 0.587)
constexpr uint32_t kRc = 9797, kGc = 19234, kBc = 3737;
uint8_t rgb_to_gray8(uint32_t r, uint32_t g, uint32_t b, const uint8_t* to_1, const uint8_t* from_1) {
  if (to_1 == nullptr) return (uint8_t)((kRc * r + kGc * g + kBc * b) >> 15);
  if (r == g && r == b) return (uint8_t)r;  // (gamma_table is the identity: file gamma x its reciprocal)
  return from_1[(kRc * to_1[r] + kGc * to_1[g] + kBc * to_1[b] + 16384) >> 15];
}

vsf_status parse_png(const uint8_t* f, size_t n, int width, int height, DevImage* im, std::vector<Piece>* pieces,
                     std::vector<uint8_t>* table) {
  pieces->clear();
  table->clear();
  if (n < 8 + 25 + 12 || std::memcmp(f, kSignature, 8) != 0) return VSF_ERR_INVALID_ARG;
  size_t pos = 8;
  bool have_ihdr = false, have_idat = false, idat_run_over = false, have_iend = false, have_plte = false;
  int channels = 1, ctype = 0, depth = 0;
  bool colour = false;             // rgb_to_gray has work to do
  int n_gama = 0, n_srgb = 0, n_chrm = 0, n_iccp = 0;
  uint32_t gama = 0;
  bool unsupported_colourspace = false, odd_chrm = false;
  const uint8_t* plte = nullptr;
  uint32_t plte_entries = 0;
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
    // png_check_chunk_name: four letters, or png_chunk_error("invalid chunk type"); png_check_chunk_length: anything but IDAT is
    // at most PNG_USER_CHUNK_MALLOC_MAX = 8 000 000 bytes ("chunk data is too large") -- both before the CRC is looked at
    for (int k = 0; k < 4; k++)
      if (!((type[k] >= 'A' && type[k] <= 'Z') || (type[k] >= 'a' && type[k] <= 'z'))) return VSF_ERR_INVALID_ARG;
    const bool is_idat = std::memcmp(type, "IDAT", 4) == 0;
    if (!is_idat && len > 8000000u) return VSF_ERR_INVALID_ARG;
    // the chunk's PLACE counts whatever its CRC says: libpng wants IHDR first ("missing IHDR": the position check of every
    // handler comes before its CRC check), and any chunk between two IDATs ends the run of IDATs ("Not enough image data" /
    // "Too many IDATs found" later on), a damaged ancillary one included
    if (!have_ihdr && std::memcmp(type, "IHDR", 4) != 0) return VSF_ERR_INVALID_ARG;
    if (have_idat && !is_idat) idat_run_over = true;
    if (!crc_ok) {
      if (critical) return VSF_ERR_INVALID_ARG;  // png_crc_error: png_chunk_error for critical chunks
      continue;                                  // ancillary: a warning, the chunk's contents are skipped
    }
    if (!have_ihdr) {
      if (std::memcmp(type, "IHDR", 4) != 0 || len != 13) return VSF_ERR_INVALID_ARG;
      have_ihdr = true;
      const uint32_t w = be32(data), h = be32(data + 4);
      depth = data[8];
      ctype = data[9];
      const int comp = data[10], filt = data[11], lace = data[12];
      if (w == 0 || h == 0 || w > 0x7FFFFFFFu || h > 0x7FFFFFFFu || comp != 0 || filt != 0 || lace > 1) return VSF_ERR_INVALID_ARG;
      const bool depth_ok = ctype == 0   ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                            : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                            : (ctype == 2 || ctype == 4 || ctype == 6) ? (depth == 8 || depth == 16)
                                                                       : false;
      if (!depth_ok) return VSF_ERR_INVALID_ARG;
      if (w != (uint32_t)width || h != (uint32_t)height) return VSF_ERR_INVALID_ARG;
      colour = ctype == 2 || ctype == 3 || ctype == 6;
      channels = ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 1;
      const uint64_t row_bits = (uint64_t)w * (uint64_t)(depth * channels);
      const uint64_t row_bytes = (row_bits + 7) / 8;
      if (row_bytes > 0x0FFFFFFFu || (row_bytes + 1) * (uint64_t)h > 0xF0000000u) return VSF_ERR_INVALID_ARG;
      im->row_bytes = (uint32_t)row_bytes;
      im->bpp = (uint8_t)std::max(1, depth * channels / 8);
      im->depth = (uint8_t)depth;
      im->kind = ctype == 3 ? kPalette : (ctype == 2 || ctype == 6) ? (depth == 16 ? kRgb16 : kRgb8) : kGray;
      im->flags = lace == 1 ? kAdam7 : 0;
      im->table = 0;
      {  // what the stream must deliver
        uint64_t total = 0;
        for (int p = 0; p < (lace == 1 ? 7 : 1); p++) {
          const Adam7Pass g = lace == 1 ? adam7_pass(p) : Adam7Pass{0, 0, 1, 1};
          const uint64_t wp = w > (uint32_t)g.x0 ? ((uint64_t)w - g.x0 + g.dx - 1) / g.dx : 0, hp = h > (uint32_t)g.y0 ? ((uint64_t)h - g.y0 + g.dy - 1) / g.dy : 0;
          if (wp == 0 || hp == 0) continue;
          total += ((wp * (uint64_t)(depth * channels) + 7) / 8 + 1) * hp;
        }
        if (total > 0xF0000000u) return VSF_ERR_INVALID_ARG;
        im->expected = (uint32_t)total;
      }
      continue;
    }
    if (std::memcmp(type, "IHDR", 4) == 0) return VSF_ERR_INVALID_ARG;
    if (is_idat) {
      if (idat_run_over) return VSF_ERR_INVALID_ARG;  // (IDAT chunks must follow one another)
      if (ctype == 3 && !have_plte) return VSF_ERR_INVALID_ARG;  // "Missing PLTE before IDAT"
      have_idat = true;
      if (len > 0) {
        pieces->push_back(Piece{(uint32_t)(data - f), len});
        stream_len += len;
        if (stream_len > 0x40000000u) return VSF_ERR_INVALID_ARG;
      }
      continue;
    }
    if (std::memcmp(type, "IEND", 4) == 0) {
      have_iend = true;
      continue;
    }
    if (std::memcmp(type, "PLTE", 4) == 0) {  // png_handle_PLTE
      if (have_plte) return VSF_ERR_INVALID_ARG;  // "duplicate"
      if (have_idat) continue;                     // "out of place": a benign error, the chunk is skipped
      have_plte = true;
      if (!colour) continue;                       // "ignored in grayscale PNG"
      if (len > 3 * 256 || len % 3 != 0) {
        if (ctype == 3) return VSF_ERR_INVALID_ARG;  // "invalid"
        continue;
      }
      if (ctype == 3) {
        plte = data;
        plte_entries = std::min<uint32_t>(len / 3, 1u << depth);  // (more entries than the depth can name are dropped)
      }
      continue;
    }
    if (colour) {  // what decides whether rgb_to_gray works on the samples as they are (pngrutil.c png_handle_gAMA / sRGB / iCCP)
      const bool in_place = !have_plte && !have_idat;
      if (std::memcmp(type, "gAMA", 4) == 0) {
        if (!in_place || len != 4) continue;  // "out of place" / "invalid": skipped
        n_gama++;
        gama = be32(data);
        if (gama < 16 || gama > 625000000u) unsupported_colourspace = true;  // "gamma value out of range"
        continue;
      }
      if (std::memcmp(type, "sRGB", 4) == 0) {
        if (!in_place || len != 1) continue;
        n_srgb++;
        if (data[0] >= 4) unsupported_colourspace = true;
        continue;
      }
      if (std::memcmp(type, "iCCP", 4) == 0) {
        n_iccp++;
        continue;
      }
      if (std::memcmp(type, "cHRM", 4) == 0) {
        if (!in_place || len != 32) continue;
        // (the primaries do not touch the coefficients asked for by name -- but a cHRM chunk libpng finds fault with makes
        // it ignore the gamma chunks behind it; the sRGB primaries, which it knows, are let through, others only alone)
        static const uint32_t srgb_xy[8] = {31270, 32900, 64000, 33000, 30000, 60000, 15000, 6000};
        bool standard = true;
        for (int k = 0; k < 8; k++) standard = standard && be32(data + 4 * k) == srgb_xy[k];
        if (!standard || n_chrm > 0) odd_chrm = true;
        n_chrm++;
        continue;
      }
    }
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
  // libpng inflates with the window the header declares (1 << (CINFO + 8)) and refuses a match that reaches further back
  // ("invalid distance too far back"); the device kernel's window is the full 32 KiB.  Every encoder in use writes CINFO 7
  // for an image of this size; a smaller declared window is refused here rather than decoded more leniently than libpng.
  if ((hdr[0] >> 4) < 7) return VSF_ERR_UNSUPPORTED;
  im->stream_len = (uint32_t)stream_len;
  if (colour) {
    if (unsupported_colourspace || n_iccp > 0 || n_gama > 1 || n_srgb > 1 || (odd_chrm && n_gama + n_srgb > 0)) return VSF_ERR_UNSUPPORTED;
    // png_init_read_transformations: the screen's gamma defaults to the reciprocal of the file's; tables are built when either
    // is "significant"
    const int32_t file_gamma = n_srgb ? 45455 : n_gama ? (int32_t)gama : 100000;
    const int32_t screen_gamma = png_reciprocal(file_gamma);
    const bool tables = png_gamma_significant(file_gamma) || png_gamma_significant(screen_gamma);
    uint8_t to_1[256], from_1[256];
    if (tables) {
      if (im->kind == kRgb16) return VSF_ERR_UNSUPPORTED;  // (libpng's 16-bit tables are not restated)
      png_build_8bit_table(to_1, png_reciprocal(file_gamma));
      png_build_8bit_table(from_1, screen_gamma > 0 ? png_reciprocal(screen_gamma) : file_gamma);
    }
    if (im->kind == kPalette) {  // the palette's entries as gray values (entries the file does not define are black)
      table->assign(256, 0);
      for (uint32_t i = 0; i < 256; i++) {
        const uint32_t r = i < plte_entries ? plte[3 * i] : 0, g = i < plte_entries ? plte[3 * i + 1] : 0, b = i < plte_entries ? plte[3 * i + 2] : 0;
        (*table)[i] = rgb_to_gray8(r, g, b, tables ? to_1 : nullptr, tables ? from_1 : nullptr);
      }
    } else if (tables) {
      im->flags |= kGammaTables;
      table->assign(to_1, to_1 + 256);
      table->insert(table->end(), from_1, from_1 + 256);
    }
  }
  return VSF_OK;
}

}  // namespace

// Step 1: every file's chunks (CRC pass included, a few threads when there is enough of it).
vsf_status vsf_png_plan(const uint8_t* const* png, const size_t* nbytes, int n, int width, int height, VsfPngPlan* plan) {
  std::vector<DevImage> images((size_t)n);
  std::vector<std::vector<Piece>> pieces((size_t)n);
  std::vector<std::vector<uint8_t>> tables((size_t)n);
  std::vector<vsf_status> status((size_t)n, VSF_OK);
  auto parse_range = [&](int i0, int i1) {
    for (int i = i0; i < i1; i++) {
      std::memset(&images[i], 0, sizeof(DevImage));
      status[i] = parse_png(png[i], nbytes[i], width, height, &images[i], &pieces[i], &tables[i]);
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
    max_filtered = std::max(max_filtered, images[i].expected);
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
  size_t table_bytes = 0;
  plan->any_rgb = plan->any_general = false;
  for (int i = 0; i < n; i++) {
    images[i].table = (uint32_t)table_bytes;
    table_bytes += tables[i].size();
    plan->any_rgb = plan->any_rgb || images[i].kind == kRgb8 || images[i].kind == kRgb16;
    plan->any_general = plan->any_general || images[i].kind == kPalette || (images[i].kind == kGray && (images[i].flags & kAdam7) != 0);
  }
  plan->filtered_stride = ((size_t)max_filtered + 15 + 16) & ~(size_t)15;
  plan->off_images = 0;
  plan->off_pieces = (images.size() * sizeof(DevImage) + 15) & ~(size_t)15;
  plan->off_tables = (plan->off_pieces + piece_end.size() * sizeof(uint32_t) + 15) & ~(size_t)15;
  plan->off_stream = (plan->off_tables + table_bytes + 15) & ~(size_t)15;
  plan->total = plan->off_stream + stream_bytes + 16;
  plan->head.assign(plan->off_stream, 0);
  std::memcpy(plan->head.data(), images.data(), images.size() * sizeof(DevImage));
  if (!piece_end.empty()) std::memcpy(plan->head.data() + plan->off_pieces, piece_end.data(), piece_end.size() * sizeof(uint32_t));
  for (int i = 0; i < n; i++)
    if (!tables[i].empty()) std::memcpy(plan->head.data() + plan->off_tables + images[i].table, tables[i].data(), tables[i].size());
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
