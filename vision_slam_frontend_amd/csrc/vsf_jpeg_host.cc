// vsf_jpeg_host.cc -- host half of vsf_jpeg_decode_gray_batch (cv::imdecode(IMREAD_GRAYSCALE) for baseline and
// progressive JPEG, slam_frontend_main.cc:98-100): parses the markers and tables of every file (ITU-T T.81 Annex B), builds the decoders'
// lookup tables once per distinct table set and lays out ONE upload.  The bytes come from a ROS bag or a network topic,
// i.e. they are UNTRUSTED: every length is checked against the file's end before it is used.  Plain C++ (no HIP code), so
// that the same translation unit builds with -fsanitize=address,undefined (make asan) and runs the fixtures and a few
// thousand mutated files on the CPU (tests/test_jpeg_host_asan.py).
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "vsf_internal.h"
#include "vsf_jpeg_host.h"

using namespace vsf_jpeg;

namespace {

const uint8_t kZigzagHost[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- host: markers and tables (ITU-T T.81 Annex B) ----
struct HostHuff {
  bool present = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
};

struct HostTableSet {  // what one file's scan uses, as read from its DQT / DHT segments (compared before anything is built:
  int nslots = 0;      // consecutive frames of a camera carry the same tables)
  uint8_t bits[kMaxSlots][17];
  uint8_t vals[kMaxSlots][256];
  uint16_t qt_luma[64];
  bool same(const HostTableSet& o) const {
    return nslots == o.nslots && std::memcmp(bits, o.bits, sizeof(bits[0]) * nslots) == 0 &&
           std::memcmp(vals, o.vals, sizeof(vals[0]) * nslots) == 0 && std::memcmp(qt_luma, o.qt_luma, sizeof(qt_luma)) == 0;
  }
};

// BITS[1..16] must describe a prefix code (T.81 Annex C): after the codes of length l the next free code must still fit in
// l bits -- libjpeg's jdhuff.c test (`code >= 1 << si` => JERR_BAD_HUFF_TABLE; the all-ones code stays unused), so a file
// cv::imdecode refuses is refused here as well.  Without it an oversubscribed table indexes far past the lookup tables.
bool huff_counts_ok(const uint8_t bits[17]) {
  int32_t code = 0;
  for (int l = 1; l <= 16; l++) {
    code += bits[l];
    if (bits[l] && code >= (1 << l)) return false;
    code <<= 1;
  }
  return true;
}

// A DC table's symbols are magnitude categories 0..15: jpeg_make_d_derived_tbl (jdhuff.c) answers anything larger with
// JERR_BAD_HUFF_TABLE when the table is set up for a scan's DC decoding, so cv::imdecode refuses such a file.  (Here a larger
// symbol would also mean shifts by more than the reader's window in receive_extend.)
bool dc_symbols_ok(const HostHuff& h) {
  int total = 0;
  for (int l = 1; l <= 16; l++) total += h.bits[l];
  for (int k = 0; k < total && k < 256; k++)
    if (h.vals[k] > 15) return false;
  return true;
}

bool build_dev_huff(const HostHuff& h, DevHuff* d) {
  std::memset(d, 0, sizeof(*d));
  if (!huff_counts_ok(h.bits)) return false;  // (parse_jpeg has refused such a table already: second line of defence)
  for (auto& e : d->look) e = kNoCode;
  for (auto& t : d->sub)
    for (auto& e : t) e = kNoCode;
  int32_t code = 0;
  int k = 0;
  for (int l = 1; l <= 16; l++) {
    const int32_t mincode = code;
    d->valoff[l] = k - mincode;
    for (int i = 0; i < h.bits[l]; i++, k++, code++) {
      if (l <= kLookBits) {  // every 9-bit prefix that starts with this code
        const int shift = kLookBits - l;
        for (int f = 0; f < (1 << shift); f++)
          d->look[((uint32_t)code << shift) | (uint32_t)f] = (uint16_t)((l << 8) | h.vals[k]);
      } else {  // second level: the table of this code's 9-bit prefix, every 7-bit continuation that starts with its rest
        const uint32_t p9 = (uint32_t)code >> (l - kLookBits);
        if (!(d->look[p9] & kLongCode)) d->look[p9] = (uint16_t)(kLongCode | std::min<uint32_t>(d->nsub++, 255u));
        const uint32_t ti = d->look[p9] & 255u;
        if (ti < (uint32_t)kMaxSub) {
          const int shift = 16 - l;
          const uint32_t rest = (uint32_t)code & ((1u << (l - kLookBits)) - 1u);
          for (int f = 0; f < (1 << shift); f++) d->sub[ti][(rest << shift) | (uint32_t)f] = (uint16_t)((l << 8) | h.vals[k]);
        }
      }
    }
    d->maxcode[l] = h.bits[l] ? code - 1 : -1;
    code <<= 1;
  }
  d->maxcode[17] = 0x7FFFFFFF;
  d->valoff[0] = 0;
  std::memcpy(d->vals, h.vals, 256);
  return true;
}

// End of an entropy-coded segment (T.81 B.1.1.2, B.1.1.5): the first 0xFF that is followed by anything but 0x00 (a stuffed
// FF), RSTn or another 0xFF (a fill byte); `n` when the data runs out first.
size_t ecs_end(const uint8_t* d, size_t pos, size_t n) {
  while (pos + 1 < n) {
    const uint8_t* q = static_cast<const uint8_t*>(std::memchr(d + pos, 0xFF, n - 1 - pos));
    if (!q) return n;
    pos = (size_t)(q - d);
    const uint8_t b = d[pos + 1];
    if (b == 0x00 || (b >= 0xD0 && b <= 0xD7)) {
      pos += 2;
    } else if (b == 0xFF) {
      pos += 1;
    } else {
      return pos;
    }
  }
  return n;
}

struct ProgFile {                 // what a progressive file adds to its image descriptor
  std::vector<DevScan> scans;     // (off = offset in the FILE until vsf_jpeg_plan rebases it; huff = index into `huffs`)
  std::vector<HostHuff> huffs;    // the tables those scans use, each as it stood when its scan began
};

}  // namespace

// What jpeg_finish_decompress reads behind the data of a file's one scan (jdmarker.c read_markers, jdinput.c
// consume_markers): every marker from the one the entropy decoder stopped at to EOI.  Most are skipped or parsed quietly; some
// end the read with a fatal error -- cv::imdecode then returns NOTHING although every row had been decoded: a marker code
// libjpeg does not know (0x02..0xBF, 0xDE, 0xDF, 0xF0..0xFD: JERR_UNKNOWN_MARKER), another frame or a frame type it does
// not read, another SOI, another SOS (JERR_EOI_EXPECTED after its header has parsed), a table segment that does not parse.
// Bytes behind the end of the file read as the memory source supplies them: 0xFF 0xD9 over and over.
// p[0..n): everything behind the scan header; restart_interval > 0: RSTn markers in the data belong to the decoder.
// Returns false for a file libjpeg gives up on.
// *rst_in_step (restart_interval > 0): the data hold exactly `intervals` - 1 restart markers, numbered 0, 1, ... 7, 0 ... in
// turn, and EOI behind them -- what the parallel decoder's interval table assumes; any other file of the kind goes to the
// one-wave decoder, which looks for its restart markers as libjpeg does.
static bool baseline_tail_ok(const uint8_t* p, size_t n, int restart_interval, size_t intervals, bool* rst_in_step) {
  auto at = [&](size_t i) -> int { return i < n ? p[i] : (((i - n) & 1) ? 0xD9 : 0xFF); };
  size_t i = 0;
  const size_t hard_end = n + 4096;  // (the virtual tail is EOI after EOI: a walk that gets this far has met one)
  // where the entropy decoder stops: the first marker in the data
  int m = -1;
  size_t n_rst = 0;
  bool in_step = true;
  // (the data of an undamaged file are scanned to their end here: memchr from 0xFF to 0xFF, not byte by byte)
  auto next_ff = [&](size_t from) -> size_t {
    if (from < n) {
      const void* f = std::memchr(p + from, 0xFF, n - from);
      return f ? (size_t)((const uint8_t*)f - p) : n;  // (at n the virtual tail begins: 0xFF 0xD9 ...)
    }
    return ((from - n) & 1) ? from + 1 : from;
  };
  while (i < hard_end) {
    i = next_ff(i);
    size_t j = i + 1;
    while (j < hard_end && at(j) == 0xFF) j++;
    const int c = at(j);
    i = j + 1;
    if (c == 0) continue;                                                // a stuffed 0xFF
    if (restart_interval > 0 && c >= 0xD0 && c <= 0xD7) {                // the decoder's own
      if (c - 0xD0 != (int)(n_rst & 7)) in_step = false;
      n_rst++;
      continue;
    }
    m = c;
    break;
  }
  if (rst_in_step) *rst_in_step = in_step && n_rst + 1 == intervals && m == 0xD9;
  while (m >= 0 && i < hard_end) {
    if (m == 0xD9) return true;                                          // EOI
    if (m == 0xD8) return false;                                         // JERR_SOI_DUPLICATE
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) {
      // (parameterless: traced and passed over)
    } else if ((m >= 0xC0 && m <= 0xCF) && m != 0xC4 && m != 0xCC) {
      return false;                                                      // JERR_SOF_DUPLICATE / JERR_SOF_UNSUPPORTED
    } else if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE || m == 0xDC) {     // APPn, COM, DNL: skip_variable
      const long length = ((at(i) << 8) | at(i + 1)) - 2;
      i += 2;
      if (length > 0) i += (size_t)length;
    } else if (m == 0xDD) {                                              // DRI
      if (((at(i) << 8) | at(i + 1)) != 4) return false;                 // JERR_BAD_LENGTH
      i += 4;
    } else if (m == 0xC4) {                                              // DHT: get_dht
      long length = ((at(i) << 8) | at(i + 1)) - 2;
      i += 2;
      while (length > 16) {
        int index = at(i++);
        long count = 0;
        for (int k = 0; k < 16; k++) count += at(i++);
        length -= 1 + 16;
        if (count > 256 || count > length) return false;                 // JERR_BAD_HUFF_TABLE
        i += (size_t)count;
        length -= count;
        if (index & 0x10) index -= 0x10;
        if (index < 0 || index >= 4) return false;                       // JERR_DHT_INDEX
      }
      if (length != 0) return false;                                     // JERR_BAD_LENGTH
    } else if (m == 0xDB) {                                              // DQT: get_dqt
      long length = ((at(i) << 8) | at(i + 1)) - 2;
      i += 2;
      while (length > 0) {
        const int b = at(i++);
        length--;
        const int prec = b >> 4;
        if ((b & 15) >= 4) return false;                                 // JERR_DQT_INDEX
        long count;
        if (prec)
          count = length < 128 ? length >> 1 : 64;
        else
          count = length < 64 ? length : 64;
        i += (size_t)(prec ? 2 * count : count);
        length -= prec ? 2 * count : count;
      }
      if (length != 0) return false;                                     // JERR_BAD_LENGTH
    } else if (m == 0xCC) {                                              // DAC: get_dac
      long length = ((at(i) << 8) | at(i + 1)) - 2;
      i += 2;
      while (length > 0) {
        const int index = at(i++), val = at(i++);
        length -= 2;
        if (index >= 32) return false;                                   // JERR_DAC_INDEX
        if (index < 16 && (val & 15) > (val >> 4)) return false;         // JERR_DAC_VALUE
      }
      if (length != 0) return false;                                     // JERR_BAD_LENGTH
    } else if (m == 0xDA) {                                              // SOS: get_sos, then JERR_EOI_EXPECTED
      return false;  // (whatever its header says: a second scan in a file that announced one is fatal either way)
    } else if (restart_interval > 0 && m < 0xC0) {
      // (in a file with restart intervals such a code is mostly met by process_restart, whose jpeg_resync_to_restart
      // discards "invalid" markers while it looks for the next RSTn; whether one survives to the end cannot be told without
      // decoding: let through)
    } else {
      return false;                                                      // JERR_UNKNOWN_MARKER
    }
    // next_marker: on to the next 0xFF that is followed by something
    m = -1;
    while (i < hard_end) {
      i = next_ff(i);
      size_t j = i + 1;
      while (j < hard_end && at(j) == 0xFF) j++;
      const int c = at(j);
      i = j + 1;
      if (c == 0) continue;
      m = c;
      break;
    }
  }
  return true;
}

// Parses one JPEG file; fills the image descriptor (without stream_off / tables) and the table set it needs.
// Returns VSF_OK, VSF_ERR_INVALID_ARG (malformed, or not width x height) or VSF_ERR_UNSUPPORTED.
// Progressive files (SOF2, T.81 Annex G; libjpeg jdphuff.c): every scan that carries the luminance component is listed in
// `prog` with the Huffman tables in force when it starts; scans of chroma alone are stepped over.  The progression must be
// the orderly one (each scan continues where the last one over the same coefficients stopped) and must end with every
// luminance coefficient at full precision: anything else libjpeg answers with an approximation, and is refused here.
static vsf_status parse_jpeg(const uint8_t* data, size_t nbytes, int width, int height, DevImage* im, HostTableSet* tab,
                             size_t* scan_begin, ProgFile* prog) {
  if (!data || nbytes < 4 || data[0] != 0xFF || data[1] != 0xD8) return VSF_ERR_INVALID_ARG;
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  HostHuff dc[4], ac[4];
  int ncomp = 0, cid[3], ch[3], cv[3], ctq[3], W = 0, H = 0, restart_interval = 0;
  bool have_sof = false, progressive = false, yq_latched = false;
  bool multiscan = false;  // a sequential frame whose components come in several scans: its scans are listed like progressive ones
  uint16_t yq[64];
  int cbits[64];  // progressive: precision still missing per luminance coefficient (-1: nothing received yet)
  for (int& b : cbits) b = -1;
  size_t pos = 2;
  std::memset(im, 0, sizeof(*im));
  prog->scans.clear();
  prog->huffs.clear();
  while (pos + 4 <= nbytes) {
    if (data[pos] != 0xFF) return VSF_ERR_INVALID_ARG;
    while (pos < nbytes && data[pos] == 0xFF) pos++;
    if (pos >= nbytes) return VSF_ERR_INVALID_ARG;
    const int m = data[pos++];
    if (m == 0xD9) {
      if (progressive || multiscan) break;
      return VSF_ERR_INVALID_ARG;
    }
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (pos + 2 > nbytes) return VSF_ERR_INVALID_ARG;
    const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
    if (len < 2 || pos + len > nbytes) return VSF_ERR_INVALID_ARG;
    const uint8_t* s = data + pos + 2;
    const size_t n = len - 2;
    if (m == 0xDA && have_sof && !progressive && n >= 1 && s[0] != ncomp) multiscan = true;
    if (m == 0xDB) {
      for (size_t i = 0; i < n;) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        i++;
        if (tq > 3 || pq > 1 || i + 64 * (size_t)(pq + 1) > n) return VSF_ERR_INVALID_ARG;
        for (int k = 0; k < 64; k++, i += pq + 1)
          qt[tq][kZigzagHost[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
        qt_present[tq] = true;
      }
    } else if (m == 0xC4) {
      for (size_t i = 0; i < n;) {
        if (i + 17 > n) return VSF_ERR_INVALID_ARG;
        const int tc = s[i] >> 4, th = s[i] & 15;
        if (tc > 1 || th > 3) return VSF_ERR_INVALID_ARG;
        HostHuff& h = tc ? ac[th] : dc[th];
        int total = 0;
        for (int l = 1; l <= 16; l++) total += (h.bits[l] = s[i + l]);
        i += 17;
        if (total > 256 || i + total > n || !huff_counts_ok(h.bits)) return VSF_ERR_INVALID_ARG;
        std::memset(h.vals, 0, sizeof(h.vals));
        std::memcpy(h.vals, s + i, (size_t)total);
        i += total;
        h.present = true;
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      if (have_sof) return VSF_ERR_INVALID_ARG;
      progressive = m == 0xC2;
      if (n < 6 || s[0] != 8) return VSF_ERR_UNSUPPORTED;
      H = (s[1] << 8) | s[2];
      W = (s[3] << 8) | s[4];
      ncomp = s[5];
      if (ncomp == 4) return VSF_ERR_UNSUPPORTED;
      if ((ncomp != 1 && ncomp != 3) || n < (size_t)(6 + 3 * ncomp)) return VSF_ERR_INVALID_ARG;
      for (int c = 0; c < ncomp; c++) {
        cid[c] = s[6 + 3 * c];
        ch[c] = s[7 + 3 * c] >> 4;
        cv[c] = s[7 + 3 * c] & 15;
        ctq[c] = s[8 + 3 * c];
        if (ch[c] < 1 || ch[c] > 4 || cv[c] < 1 || cv[c] > 4 || ctq[c] > 3) return VSF_ERR_INVALID_ARG;
      }
      have_sof = true;
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return VSF_ERR_UNSUPPORTED;  // lossless, arithmetic, hierarchical
    } else if (m == 0xDD) {
      if (n < 2) return VSF_ERR_INVALID_ARG;
      restart_interval = (s[0] << 8) | s[1];
    } else if (m == 0xDA && (progressive || multiscan)) {
      if (!have_sof || n < 1) return VSF_ERR_INVALID_ARG;
      const int ns = s[0];
      if (ns < 1 || ns > ncomp || n < (size_t)(4 + 2 * ns)) return VSF_ERR_INVALID_ARG;
      DevScan sc;
      std::memset(&sc, 0, sizeof(sc));
      int td[3], ta[3];
      bool has_luma = false;
      for (int c = 0; c < ns; c++) {
        int ci = -1;
        for (int f = 0; f < ncomp; f++)
          if (cid[f] == s[1 + 2 * c]) ci = f;
        if (ci < 0 || (c > 0 && sc.comp[c - 1] >= ci)) return VSF_ERR_INVALID_ARG;  // frame order (T.81 B.2.3)
        sc.comp[c] = (uint8_t)ci;
        td[c] = s[2 + 2 * c] >> 4;
        ta[c] = s[2 + 2 * c] & 15;
        if (td[c] > 3 || ta[c] > 3) return VSF_ERR_INVALID_ARG;
        has_luma |= ci == 0;
      }
      const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
      if (ns > 1) {  // jdinput.c per_scan_setup: an interleaved scan holds at most D_MAX_BLOCKS_IN_MCU = 10 blocks per MCU
        int blocks = 0;
        for (int c = 0; c < ns; c++) blocks += ch[sc.comp[c]] * cv[sc.comp[c]];
        if (blocks > 10) return VSF_ERR_INVALID_ARG;  // JERR_BAD_MCU_SIZE
      }
      if (Ss == 0 && Ah == 0)  // a scan that decodes DC symbols, luminance or not: libjpeg derives every table the scan names
        for (int c = 0; c < ns; c++)
          if (dc[td[c]].present && !dc_symbols_ok(dc[td[c]])) return VSF_ERR_INVALID_ARG;  // JERR_BAD_HUFF_TABLE
      if (multiscan) {  // a sequential scan: the whole band at full precision
        if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return VSF_ERR_INVALID_ARG;
      } else {  // jdphuff.c start_pass_phuff_decoder: the legal shapes of a progressive scan
        if (Ss == 0 ? Se != 0 : (ns != 1 || Se < Ss || Se > 63)) return VSF_ERR_INVALID_ARG;
        if ((Ah != 0 && Al != Ah - 1) || Al > 13) return VSF_ERR_INVALID_ARG;
      }
      const size_t begin = pos + len, end = ecs_end(data, begin, nbytes);
      if (has_luma) {
        for (int k = Ss; k <= Se; k++) {
          if (Ah != (cbits[k] < 0 ? 0 : cbits[k]) || (cbits[k] >= 0 && Ah == 0)) return VSF_ERR_UNSUPPORTED;
          cbits[k] = Al;
        }
        if (Ss > 0 && cbits[0] < 0) return VSF_ERR_UNSUPPORTED;  // AC before any DC scan
        if (!yq_latched) {  // jdinput.c latch_quant_tables: the table in force at the component's first scan
          if (!qt_present[ctq[0]]) return VSF_ERR_INVALID_ARG;
          std::memcpy(yq, qt[ctq[0]], sizeof(yq));
          yq_latched = true;
        }
        sc.off = (uint32_t)begin;
        sc.len = (uint32_t)(end - begin);
        sc.restart_interval = restart_interval;
        sc.ncomp = (uint8_t)ns;
        sc.Ss = (uint8_t)Ss;
        sc.Se = (uint8_t)Se;
        sc.Ah = (uint8_t)Ah;
        sc.Al = (uint8_t)Al;
        auto table_index = [&](const HostHuff& h) -> int {  // in prog->huffs (a table is listed once per file)
          if (!h.present) return -1;
          for (size_t i = 0; i < prog->huffs.size(); i++)
            if (std::memcmp(prog->huffs[i].bits, h.bits, 17) == 0 && std::memcmp(prog->huffs[i].vals, h.vals, 256) == 0) return (int)i;
          prog->huffs.push_back(h);
          return (int)prog->huffs.size() - 1;
        };
        for (int c = 0; c < ns; c++) {
          sc.huff[c] = sc.huff_ac[c] = 0;
          if (Ss == 0 && Ah != 0) continue;  // DC refinement: raw bits
          const int first = table_index(Ss == 0 ? dc[td[c]] : ac[ta[c]]);
          const int second = multiscan ? table_index(ac[ta[c]]) : 0;
          if (first < 0 || second < 0) return VSF_ERR_INVALID_ARG;
          sc.huff[c] = (uint32_t)first;
          sc.huff_ac[c] = (uint32_t)second;
        }
        if (prog->scans.size() >= 1024) return VSF_ERR_INVALID_ARG;  // (64 coefficients x 14 bits bound an orderly file far below)
        prog->scans.push_back(sc);
      }
      pos = end;
      continue;
    } else if (m == 0xDA) {
      if (!have_sof || n < 1) return VSF_ERR_INVALID_ARG;
      const int ns = s[0];
      if (ns != ncomp) return VSF_ERR_UNSUPPORTED;  // one interleaved scan only
      if (n < (size_t)(4 + 2 * ns)) return VSF_ERR_INVALID_ARG;
      std::vector<std::pair<int, int>> slots;  // (class, id) in use
      auto slot_of = [&](int cls, int id) {
        for (size_t i = 0; i < slots.size(); i++)
          if (slots[i].first == cls && slots[i].second == id) return (int)i;
        slots.emplace_back(cls, id);
        return (int)slots.size() - 1;
      };
      for (int c = 0; c < ns; c++) {
        if (s[1 + 2 * c] != cid[c]) return VSF_ERR_UNSUPPORTED;
        const int td = s[2 + 2 * c] >> 4, ta = s[2 + 2 * c] & 15;
        if (td > 3 || ta > 3 || !dc[td].present || !ac[ta].present || !qt_present[ctq[c]]) return VSF_ERR_INVALID_ARG;
        if (!dc_symbols_ok(dc[td])) return VSF_ERR_INVALID_ARG;  // JERR_BAD_HUFF_TABLE in libjpeg
        im->dc_slot[c] = slot_of(0, td);
        im->ac_slot[c] = slot_of(1, ta);
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return VSF_ERR_UNSUPPORTED;
      int hmax = 1, vmax = 1;
      for (int c = 0; c < ncomp; c++) {
        hmax = std::max(hmax, ch[c]);
        vmax = std::max(vmax, cv[c]);
      }
      if (ncomp > 1 && (ch[0] != hmax || cv[0] != vmax)) return VSF_ERR_UNSUPPORTED;  // luminance would need upsampling
      if (W != width || H != height) return VSF_ERR_INVALID_ARG;
      const bool single = ncomp == 1;  // T.81 A.2.2: a one-component scan has one block per MCU
      const int mw = single ? 8 : 8 * hmax, mh = single ? 8 : 8 * vmax;
      if (!single && hmax * vmax > kGroupBlocks) return VSF_ERR_UNSUPPORTED;
      if (!single) {  // jdinput.c per_scan_setup: at most D_MAX_BLOCKS_IN_MCU = 10 blocks per MCU (JERR_BAD_MCU_SIZE)
        int blocks = 0;
        for (int c = 0; c < ncomp; c++) blocks += ch[c] * cv[c];
        if (blocks > 10) return VSF_ERR_INVALID_ARG;
      }
      im->ncomp = ncomp;
      im->restart_interval = restart_interval;
      im->mcus_x = (W + mw - 1) / mw;
      im->mcus_y = (H + mh - 1) / mh;
      for (int c = 0; c < ncomp; c++) {
        im->h[c] = single ? 1 : ch[c];
        im->v[c] = single ? 1 : cv[c];
      }
      tab->nslots = (int)slots.size();
      for (size_t i = 0; i < slots.size(); i++) {
        const HostHuff& h = slots[i].first ? ac[slots[i].second] : dc[slots[i].second];
        std::memcpy(tab->bits[i], h.bits, 17);
        std::memcpy(tab->vals[i], h.vals, 256);
      }
      std::memcpy(tab->qt_luma, qt[ctq[0]], sizeof(tab->qt_luma));
      *scan_begin = pos + len;
      return *scan_begin < nbytes ? VSF_OK : VSF_ERR_INVALID_ARG;  // (what lies behind the scan's data: vsf_jpeg_plan, baseline_tail_ok)
    }
    pos += len;
  }
  if (!(progressive || multiscan) || !have_sof || prog->scans.empty() || !yq_latched) return VSF_ERR_INVALID_ARG;
  for (int k = 0; k < 64; k++)
    if (cbits[k] != 0) return VSF_ERR_UNSUPPORTED;  // the scans stop short of full precision
  {
    int hmax = 1, vmax = 1;
    for (int c = 0; c < ncomp; c++) {
      hmax = std::max(hmax, ch[c]);
      vmax = std::max(vmax, cv[c]);
    }
    if (ch[0] != hmax || cv[0] != vmax) return VSF_ERR_UNSUPPORTED;  // luminance would need upsampling
    if (W != width || H != height) return VSF_ERR_INVALID_ARG;
    // the coefficient buffer is laid out as the frame's interleaved MCUs (what the IDCT kernel walks): for a gray frame that
    // is one block per MCU whatever its sampling factors say
    const bool single = ncomp == 1;
    const int mw = single ? 8 : 8 * hmax, mh = single ? 8 : 8 * vmax;
    im->ncomp = ncomp;
    im->restart_interval = 0;
    im->mcus_x = (W + mw - 1) / mw;
    im->mcus_y = (H + mh - 1) / mh;
    for (int c = 0; c < ncomp; c++) {
      im->h[c] = single ? 1 : ch[c];
      im->v[c] = single ? 1 : cv[c];
    }
    im->n_scans = (int32_t)prog->scans.size();
    tab->nslots = 0;
    std::memcpy(tab->qt_luma, yq, sizeof(tab->qt_luma));
    *scan_begin = prog->scans[0].off;
    for (DevScan& sc : prog->scans) sc.off -= (uint32_t)*scan_begin;
  }
  return VSF_OK;
}

// Host half of vsf_jpeg_decode_gray_batch, step 1: parses every file and lays out ONE upload -- image descriptors,
// distinct table sets (consecutive frames of a camera share theirs: compared with the previous file's first), packed
// entropy-coded segments -- without touching the segments themselves.
vsf_status vsf_jpeg_plan(const uint8_t* const* jpeg, const size_t* nbytes, int n, int width, int height, bool force_serial,
                         VsfJpegPlan* plan) {
  std::vector<DevImage> images((size_t)n);
  std::vector<DevTables> tables;
  std::vector<HostTableSet> sets;
  std::vector<int> set_par_ok;
  std::vector<DevScan> scans;      // progressive files: their luminance scans, file after file
  std::vector<DevHuffSrc> prog_huff;  // ... and the tables of those scans, as the files define them
  ProgFile prog;
  plan->scan_begin.assign((size_t)n, 0);
  plan->max_luma_blocks = 0;
  plan->max_slots = 1;
  size_t stream_bytes = 0;
  for (int i = 0; i < n; i++) {
    HostTableSet t;
    const vsf_status st = parse_jpeg(jpeg[i], nbytes[i], width, height, &images[i], &t, &plan->scan_begin[i], &prog);
    if (st != VSF_OK) return st;
    if (images[i].n_scans > 0) {
      images[i].first_scan = (uint32_t)scans.size();
      const uint32_t huff0 = (uint32_t)prog_huff.size();
      for (const HostHuff& h : prog.huffs) {  // (parse_jpeg has checked the counts: huff_counts_ok)
        prog_huff.emplace_back();
        std::memset(prog_huff.back().bits, 0, sizeof(prog_huff.back().bits));
        std::memcpy(prog_huff.back().bits, h.bits, 17);
        std::memcpy(prog_huff.back().vals, h.vals, 256);
      }
      for (DevScan sc : prog.scans) {
        for (int c = 0; c < sc.ncomp; c++) {
          sc.huff[c] += huff0;
          sc.huff_ac[c] += huff0;
        }
        scans.push_back(sc);
      }
      if (prog_huff.size() * sizeof(DevHuffLite) > 0x40000000u) return VSF_ERR_INVALID_ARG;
    }
    int found = -1;
    for (int k = (int)sets.size() - 1; k >= 0 && found < 0; k--)
      if (sets[k].same(t)) found = k;
    if (found < 0) {  // a new table set: the kernels' lookup tables are built once per set
      found = (int)sets.size();
      sets.push_back(t);
      plan->max_slots = std::max(plan->max_slots, t.nslots);
      tables.emplace_back();
      DevTables& d = tables.back();
      std::memset(&d, 0, sizeof(d));
      int ok = 1;
      for (int k = 0; k < t.nslots; k++) {
        HostHuff h;
        std::memcpy(h.bits, t.bits[k], 17);
        std::memcpy(h.vals, t.vals[k], 256);
        if (!build_dev_huff(h, &d.huff[k])) return VSF_ERR_INVALID_ARG;
        if (d.huff[k].nsub > (uint32_t)kMaxSub) ok = 0;
      }
      std::memcpy(d.qt_luma, t.qt_luma, sizeof(d.qt_luma));
      set_par_ok.push_back(ok);
    }
    images[i].tables = (uint32_t)found;
    images[i].par_ok = set_par_ok[(size_t)found];
    images[i].stream_off = (uint32_t)stream_bytes;
    images[i].stream_len = (uint32_t)(nbytes[i] - plan->scan_begin[i]);
    stream_bytes += (images[i].stream_len + 3u + 32u) & ~(size_t)3;
    plan->max_luma_blocks = std::max(plan->max_luma_blocks, images[i].mcus_x * images[i].mcus_y * images[i].h[0] * images[i].v[0]);
    if (stream_bytes > 0xF0000000u) return VSF_ERR_INVALID_ARG;
  }
  {  // What jpeg_finish_decompress makes of the bytes behind every one-scan file's data (baseline_tail_ok): that is a scan of
     // all of the entropy-coded data for markers -- a few threads when there is enough of it.
    std::vector<uint8_t> ok((size_t)n, 1);
    auto walk = [&](int i0, int i1) {
      for (int i = i0; i < i1; i++)
        if (images[i].n_scans == 0) {
          const int ri = images[i].restart_interval;
          const size_t intervals = ri > 0 ? ((size_t)images[i].mcus_x * images[i].mcus_y + ri - 1) / ri : 0;
          bool in_step = true;
          ok[i] = baseline_tail_ok(jpeg[i] + plan->scan_begin[i], nbytes[i] - plan->scan_begin[i], ri, intervals, &in_step) ? 1 : 0;
          if (ri > 0 && !in_step) images[i].par_ok = 0;  // (restart markers out of step: the one-wave decoder's)
        }
    };
    size_t all = 0;
    for (int i = 0; i < n; i++) all += nbytes[i];
    const int workers = (int)std::min<size_t>({(size_t)8, all >> 20, (size_t)n, (size_t)std::max(1u, std::thread::hardware_concurrency())});
    if (workers <= 1) {
      walk(0, n);
    } else {
      std::vector<std::thread> pool;
      for (int w = 1; w < workers; w++) pool.emplace_back(walk, (int)((int64_t)n * w / workers), (int)((int64_t)n * (w + 1) / workers));
      walk(0, n / workers);
      for (auto& th : pool) th.join();
    }
    for (int i = 0; i < n; i++)
      if (!ok[i]) return VSF_ERR_INVALID_ARG;
  }
  // which decoder takes which file: those without restart intervals first, progressive files next
  std::vector<uint32_t> index;
  auto parallel = [&](int i) {
    const DevImage& im = images[i];
    if (!im.par_ok || force_serial || im.n_scans > 0) return false;
    if (im.restart_interval == 0) return true;
    // restart intervals: the table of their start offsets must fit the scratch behind the clean stream
    const size_t intervals = ((size_t)im.mcus_x * im.mcus_y + im.restart_interval - 1) / im.restart_interval;
    return 4 * (intervals + 2) <= (size_t)im.stream_len + kTransSlack;
  };
  auto decoder = [&](int i) { return parallel(i) ? 0 : (images[i].n_scans > 0 ? 1 : 2); };
  plan->n_par = plan->n_prog = 0;
  for (int pass = 0; pass < 3; pass++)
    for (int i = 0; i < n; i++)
      if (decoder(i) == pass) {
        index.push_back((uint32_t)i);
        plan->n_par += pass == 0;
        plan->n_prog += pass == 1;
      }
  plan->off_images = 0;
  plan->off_index = (images.size() * sizeof(DevImage) + 15) & ~(size_t)15;
  plan->off_tables = (plan->off_index + index.size() * sizeof(uint32_t) + 15) & ~(size_t)15;
  plan->off_scans = (plan->off_tables + tables.size() * sizeof(DevTables) + 15) & ~(size_t)15;
  plan->off_prog_huff = (plan->off_scans + scans.size() * sizeof(DevScan) + 15) & ~(size_t)15;
  plan->off_stream = (plan->off_prog_huff + prog_huff.size() * sizeof(DevHuffSrc) + 15) & ~(size_t)15;
  plan->n_prog_huff = (int)prog_huff.size();
  plan->total = plan->off_stream + stream_bytes + 16;
  plan->head.assign(plan->off_stream, 0);
  std::memcpy(plan->head.data() + plan->off_images, images.data(), images.size() * sizeof(DevImage));
  std::memcpy(plan->head.data() + plan->off_index, index.data(), index.size() * sizeof(uint32_t));
  std::memcpy(plan->head.data() + plan->off_tables, tables.data(), tables.size() * sizeof(DevTables));
  if (!scans.empty()) std::memcpy(plan->head.data() + plan->off_scans, scans.data(), scans.size() * sizeof(DevScan));
  if (!prog_huff.empty()) std::memcpy(plan->head.data() + plan->off_prog_huff, prog_huff.data(), prog_huff.size() * sizeof(DevHuffSrc));
  plan->stream_off.resize((size_t)n);
  plan->stream_len.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    plan->stream_off[i] = images[i].stream_off;
    plan->stream_len[i] = images[i].stream_len;
  }
  return VSF_OK;
}

// Step 2: writes the upload into `dst` (pinned staging, plan->total bytes): one pass over the compressed bytes, shared
// by a few threads when there is enough of it (one core copies ~19 GB/s into pinned memory).
void vsf_jpeg_fill(const VsfJpegPlan& plan, const uint8_t* const* jpeg, int n, uint8_t* dst) {
  std::memcpy(dst, plan.head.data(), plan.head.size());
  auto copy_range = [&](int i0, int i1) {
    for (int i = i0; i < i1; i++) {
      uint8_t* d = dst + plan.off_stream + plan.stream_off[i];
      std::memcpy(d, jpeg[i] + plan.scan_begin[i], plan.stream_len[i]);
      const size_t padded = (plan.stream_len[i] + 3u + 32u) & ~(size_t)3;
      std::memset(d + plan.stream_len[i], 0, padded - plan.stream_len[i]);
    }
  };
  const size_t stream_bytes = plan.total - plan.off_stream;
  const int workers = (int)std::min<size_t>({(size_t)4, stream_bytes >> 22, (size_t)n,
                                             (size_t)std::max(1u, std::thread::hardware_concurrency())});
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

