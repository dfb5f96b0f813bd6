// vsf_oracle_jpeg.cc -- CPU ORACLE (test infrastructure, NOT product code) for SURVEY.md section 8(f) row f4, the part
// in front of the Bayer conversion: DecodeImage's  cv::imdecode(msg.data, cv::IMREAD_GRAYSCALE)  for the JPEG payloads
// of sensor_msgs::CompressedImage (slam_frontend_main.cc:98-100).
//
// OpenCV 3.2's JPEG reader (imgcodecs/src/grfmt_jpeg.cpp) hands the stream to libjpeg with
// cinfo.out_color_space = JCS_GRAYSCALE for a gray read and leaves dct_method at its default JDCT_ISLOW, so the result
// is: ITU-T T.81 baseline entropy decoding (Annex F: Huffman categories, EXTEND, DC prediction, restart intervals), and
// for the luminance component only libjpeg's jidctint.c (jpeg_idct_islow: 13-bit constants, PASS1_BITS = 2, exact 32-bit
// integer arithmetic) followed by the range limit; chroma components are parsed and dropped.  Both are restated here
// from the standard and from the published libjpeg algorithm.
//
// PROGRESSIVE files (SOF2; T.81 Annex G, libjpeg jdphuff.c) take decode_progressive() below: the scans that carry the
// luminance component are decoded into one coefficient array (DC first / DC refinement / AC first with end-of-band runs /
// AC refinement with correction bits), scans of chroma alone are stepped over, and when the last scan has brought every
// luminance coefficient to full precision the same dequantisation + inverse DCT follows.  A file whose scans stop short
// of that is answered with -2: libjpeg would show an approximation (with its block smoothing) that is not restated.
//
// PINNED (unlike the rest of the oracle): tests/golden/jpeg/ holds JPEG files and their JCS_GRAYSCALE decode by
// libjpeg-turbo (through Pillow, which happens to be in the image; tools/make_jpeg_golden.py); this file reproduces
// every one of them bit for bit (tests/test_jpeg_oracle.py).
#include <cstdint>
#include <cstring>
#include <vector>

#include "vsf_oracle.h"

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {  // T.81 Annex C / F.2.2.3: code lengths -> MINCODE / MAXCODE / VALPTR
  bool present = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int32_t mincode[17], maxcode[18], valptr[17];
  bool derive() {  // false: the counts oversubscribe the code space (libjpeg jdhuff.c: JERR_BAD_HUFF_TABLE)
    int32_t code = 0;
    int k = 0;
    bool ok = true;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      if (bits[l] && code >= (1 << l)) ok = false;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    return ok;
  }
  // jdhuff.c jpeg_make_d_derived_tbl(isDC = TRUE): a DC table's symbols are magnitude categories, 0..15; anything larger
  // is JERR_BAD_HUFF_TABLE when a scan sets the table up for DC decoding
  bool dc_symbols_ok() const {
    int total = 0;
    for (int l = 1; l <= 16; l++) total += bits[l];
    for (int k = 0; k < total && k < 256; k++)
      if (vals[k] > 15) return false;
    return true;
  }
};

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
};

struct BitReader {  // entropy-coded segment: MSB first, FF00 -> FF, stops feeding at any other marker
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int n = 0;
  bool hit_marker = false;
  bool starved = false;  // a bit was asked for that the data do not hold: jdhuff.c's insufficient_data
  int get_bit() {
    if (n == 0) {
      uint8_t b = 0;
      if (!hit_marker && p < end) {
        b = *p++;
        if (b == 0xFF) {
          if (p < end && *p == 0x00) {
            p++;
          } else {  // a marker: leave it in place, feed zero bits (libjpeg does the same with a warning)
            p--;
            hit_marker = true;
            starved = true;
            b = 0;
          }
        }
      } else {
        starved = true;
      }
      acc = b;
      n = 8;
    }
    n--;
    return (acc >> n) & 1;
  }
  int receive(int s) {
    int v = 0;
    for (int i = 0; i < s; i++) v = (v << 1) | get_bit();
    return v;
  }
  int decode(const Huff& h) {  // F.2.2.3 DECODE
    int32_t code = get_bit();
    int l = 1;
    while (l <= 16 && code > h.maxcode[l]) {
      code = (code << 1) | get_bit();
      l++;
    }
    if (l > 16) return 0;  // corrupt stream
    return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  // RSTn between restart intervals: drop the remaining bits, step over the marker
  bool restart() {
    n = 0;
    hit_marker = false;
    while (p + 1 < end) {
      if (p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) {
        p += 2;
        starved = false;  // (process_restart: the marker was there, the next interval has its data)
        return true;
      }
      if (p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF) return false;  // another marker: stream is broken
      p++;
    }
    return false;
  }
  // The same as libjpeg does it for a sequential scan (jdhuff.c process_restart, jdmarker.c read_restart_marker and
  // jpeg_resync_to_restart): the bits left are dropped; the next marker is looked for (bytes in front of it are passed over);
  // if it is the restart marker that is due (`next`: 0..7) it is taken and the interval has its data.  If not: a code below
  // 0xC0 is passed over and the search goes on; any other marker that is no RSTn stays where it is -- and so does an RSTn one
  // or two numbers AHEAD of the one due: the interval is decoded from no data (its first MCU from zero bits, the rest gray) and
  // the marker is met again at the next boundary; an RSTn one or two numbers BEHIND is passed over; any other RSTn is taken as
  // if it were the one due.  Never fails.
  void restart_as_libjpeg(int& next) {
    n = 0;
    hit_marker = false;
    for (;;) {
      while (p < end && *p != 0xFF) p++;                // next_marker: up to an 0xFF ...
      const uint8_t* q = p;
      while (q < end && *q == 0xFF) q++;                // ... and the byte behind the last of them
      const int m = q < end ? *q : 0xD9;                // (the memory source ends every file with an EOI of its own)
      if (q < end && m == 0x00) {                       // a stuffed 0xFF is no marker
        p = q + 1;
        continue;
      }
      bool take = false, leave = false;
      if (m == 0xD0 + next) {
        take = true;
      } else if (m < 0xC0) {
        // (invalid: passed over)
      } else if (m < 0xD0 || m > 0xD7) {
        leave = true;
      } else {
        const int ahead = (m - 0xD0 - next) & 7;
        if (ahead == 1 || ahead == 2)
          leave = true;
        else if (ahead == 7 || ahead == 6) {
          // (a prior restart: passed over)
        } else {
          take = true;
        }
      }
      if (take) {
        p = q < end ? q + 1 : end;
        starved = false;  // (insufficient_data is cleared only when no marker is left unread)
        break;
      }
      if (leave) {
        hit_marker = true;  // (p stays at the marker: no data until it has been dealt with)
        break;
      }
      p = q < end ? q + 1 : end;
      if (p >= end) {       // (only EOIs from here on: one of them is "left")
        hit_marker = true;
        break;
      }
    }
    next = (next + 1) & 7;
  }
};

inline int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }  // F.2.2.1

// libjpeg jidctint.c jpeg_idct_islow on one dequantised 8 x 8 block (natural order), result incl. the range limit.  JLONG is
// `long`: 64 bits wherever the reference runs (LP64), so the sums never overflow and are cut to int at the two DESCALEs --
// which shows on coefficients no encoder writes (damaged files).
inline uint8_t range_limit(int32_t x) {
  // sample_range_limit + CENTERJSAMPLE indexed with (x & RANGE_MASK), RANGE_MASK = 1023 (jdmaster.c prepare_range_limit_table)
  const int t = x & 1023;
  if (t < 128) return (uint8_t)(t + 128);
  if (t < 512) return 255;
  if (t < 896) return 0;
  return (uint8_t)(t - 896);
}
void idct_islow(const int32_t* in, uint8_t* out, size_t ostride) {
  typedef long long L;
  const L F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
          F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  const int CONST_BITS = 13, PASS1_BITS = 2;
  int ws[64];
  for (int pass = 0; pass < 2; pass++) {
    for (int i = 0; i < 8; i++) {
      L d[8];
      for (int k = 0; k < 8; k++) d[k] = pass == 0 ? (L)in[8 * k + i] : (L)ws[8 * i + k];
      L z2 = d[2], z3 = d[6];
      L z1 = (z2 + z3) * F0541;
      L tmp2 = z1 + z3 * (-F1847);
      L tmp3 = z1 + z2 * F0765;
      z2 = d[0];
      z3 = d[4];
      L tmp0 = (z2 + z3) * 8192;
      L tmp1 = (z2 - z3) * 8192;
      const L tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = d[7];
      tmp1 = d[5];
      tmp2 = d[3];
      tmp3 = d[1];
      z1 = tmp0 + tmp3;
      z2 = tmp1 + tmp2;
      z3 = tmp0 + tmp2;
      L z4 = tmp1 + tmp3;
      const L z5 = (z3 + z4) * F1175;
      tmp0 *= F0298;
      tmp1 *= F2053;
      tmp2 *= F3072;
      tmp3 *= F1501;
      z1 *= -F0899;
      z2 *= -F2562;
      z3 *= -F1961;
      z4 *= -F0390;
      z3 += z5;
      z4 += z5;
      tmp0 += z1 + z3;
      tmp1 += z2 + z4;
      tmp2 += z2 + z3;
      tmp3 += z1 + z4;
      const L r[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0,
                      tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
      const int shift = pass == 0 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS + 3;
      for (int k = 0; k < 8; k++) {
        const int v = (int)((r[k] + (1LL << (shift - 1))) >> shift);
        if (pass == 0)
          ws[8 * k + i] = v;
        else
          out[i * ostride + k] = range_limit(v);
      }
    }
  }
}

// ---- progressive (SOF2) -------------------------------------------------------------------------------------------
// End of an entropy-coded segment: the first 0xFF followed by anything but 0x00 (a stuffed FF), RSTn, or another 0xFF
// (fill byte).  T.81 B.1.1.2 / B.1.1.5.
size_t ecs_end(const uint8_t* d, size_t pos, size_t n) {
  while (pos + 1 < n) {
    const uint8_t* q = (const uint8_t*)std::memchr(d + pos, 0xFF, n - 1 - pos);
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

int parse_dqt(const uint8_t* s, size_t n, uint16_t qt[4][64], bool present[4]) {
  size_t i = 0;
  while (i < n) {
    const int pq = s[i] >> 4, tq = s[i] & 15;
    i++;
    if (tq > 3 || pq > 1 || i + 64 * (size_t)(pq + 1) > n) return -1;
    for (int k = 0; k < 64; k++) {
      qt[tq][kZigzag[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
      i += pq + 1;
    }
    present[tq] = true;
  }
  return 0;
}

int parse_dht(const uint8_t* s, size_t n, Huff dc[4], Huff ac[4]) {
  size_t i = 0;
  while (i < n) {
    if (i + 17 > n) return -1;
    const int tc = s[i] >> 4, th = s[i] & 15;
    if (tc > 1 || th > 3) return -1;
    Huff& h = tc ? ac[th] : dc[th];
    int total = 0;
    for (int l = 1; l <= 16; l++) total += (h.bits[l] = s[i + l]);
    i += 17;
    if (total > 256 || i + total > n) return -1;
    std::memset(h.vals, 0, sizeof(h.vals));
    std::memcpy(h.vals, s + i, total);
    i += total;
    h.present = true;
    if (!h.derive()) return -1;
  }
  return 0;
}

struct ScanComp {
  int ci, td, ta;
};

// One scan of a progressive file that carries the luminance component (frame component 0).  jdphuff.c:
// decode_mcu_DC_first / decode_mcu_DC_refine / decode_mcu_AC_first / decode_mcu_AC_refine, process_restart.
// `coef`: the luminance coefficients, blocks in raster order over a grid `bw` blocks wide, natural order inside a block.
int decode_prog_scan(const uint8_t* ecs, const uint8_t* ecs_stop, const std::vector<Comp>& comps,
                     const std::vector<ScanComp>& sc, const Huff dc[4], const Huff ac[4], int Ss, int Se, int Ah, int Al,
                     int restart_interval, int W, int H, int hmax, int vmax, int mcus_x, int mcus_y, int bw,
                     std::vector<int16_t>& coef) {
  BitReader br{ecs, ecs_stop};
  int pred[4] = {0, 0, 0, 0};
  int eobrun = 0;
  int until_restart = restart_interval;
  const bool interleaved = sc.size() > 1;
  int units_x, units_y;  // MCUs of this scan
  if (interleaved) {
    units_x = mcus_x;
    units_y = mcus_y;
  } else {  // T.81 A.2.2: one block per MCU, the component's own block grid (not padded to whole frame MCUs)
    const Comp& c = comps[sc[0].ci];
    units_x = ((W * c.h + hmax - 1) / hmax + 7) / 8;
    units_y = ((H * c.v + vmax - 1) / vmax + 7) / 8;
  }
  const int p1 = 1 << Al, m1 = -(1 << Al);
  auto one_block = [&](int16_t* blk, const ScanComp& s, int slot) {
    if (Ss == 0 && Se == 63) {  // a scan of a SEQUENTIAL file (several scans, each component in one of them): F.2.2.1, F.2.2.2
      const int t = br.decode(dc[s.td]);
      pred[slot] += extend(br.receive(t), t);
      if (blk) blk[0] = (int16_t)pred[slot];
      for (int k = 1; k < 64;) {
        const int rs = br.decode(ac[s.ta]);
        const int r = rs >> 4, sz = rs & 15;
        if (sz == 0) {
          if (r != 15) break;
          k += 16;
          continue;
        }
        k += r;
        const int v = extend(br.receive(sz), sz);
        // (damaged data can run past the block's end: jpeg_natural_order has sixteen spare entries that all say 63 -- the
        // value lands on the last coefficient, jdhuff.c decode_mcu_slow)
        if (blk) blk[kZigzag[k > 63 ? 63 : k]] = (int16_t)v;
        k++;
      }
      return;
    }
    if (Ss == 0) {
      if (Ah == 0) {  // DC first: the difference coded as in the sequential process, stored shifted left by Al
        const int t = br.decode(dc[s.td]);
        pred[slot] += extend(br.receive(t), t);
        if (blk) blk[0] = (int16_t)(pred[slot] * (1 << Al));
      } else if (br.get_bit()) {  // DC refinement: one more bit of every DC value
        if (blk) blk[0] = (int16_t)(blk[0] | p1);
      }
      return;
    }
    const Huff& h = ac[s.ta];
    if (Ah == 0) {  // AC first
      if (eobrun > 0) {
        eobrun--;
        return;
      }
      for (int k = Ss; k <= Se; k++) {
        const int rs = br.decode(h);
        int r = rs >> 4;
        const int sz = rs & 15;
        if (sz) {
          k += r;
          const int v = extend(br.receive(sz), sz);
          blk[kZigzag[k > 63 ? 63 : k]] = (int16_t)(v * (1 << Al));  // (jdphuff.c decode_mcu_AC_first: the padded order table)
        } else if (r == 15) {
          k += 15;
        } else {  // EOBr: this band is finished in this and the next 2^r + extra - 1 blocks
          eobrun = 1 << r;
          if (r) eobrun += br.receive(r);
          eobrun--;
          break;
        }
      }
      return;
    }
    // AC refinement: new coefficients enter with magnitude 1 << Al, every coefficient that is already non-zero gets a
    // correction bit as the decoder passes it
    auto correct = [&](int16_t& c) {
      if (br.get_bit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
    };
    int k = Ss;
    if (eobrun == 0) {
      for (; k <= Se; k++) {
        const int rs = br.decode(h);
        int r = rs >> 4;
        int sz = rs & 15;
        int val = 0;
        if (sz) {
          val = br.get_bit() ? p1 : m1;  // (size must be 1; libjpeg warns and carries on the same way)
        } else if (r != 15) {
          eobrun = 1 << r;
          if (r) eobrun += br.receive(r);
          break;  // the rest of the block is handled by the end-of-band logic
        }
        while (k <= Se) {
          int16_t& c = blk[kZigzag[k]];
          if (c != 0) {
            correct(c);
          } else if (--r < 0) {
            break;  // the target zero-valued coefficient
          }
          k++;
        }
        if (val) blk[kZigzag[k > 63 ? 63 : k]] = (int16_t)val;  // (behind a band that ends at 63: natural_order[64] = 63)
      }
    }
    if (eobrun > 0) {
      for (; k <= Se; k++) {
        int16_t& c = blk[kZigzag[k]];
        if (c != 0) correct(c);
      }
      eobrun--;
    }
  };
  for (int uy = 0; uy < units_y; uy++)
    for (int ux = 0; ux < units_x; ux++) {
      if (restart_interval && until_restart == 0) {
        if (!br.restart()) return -1;
        pred[0] = pred[1] = pred[2] = pred[3] = 0;
        eobrun = 0;
        until_restart = restart_interval;
      }
      if (interleaved) {
        for (size_t si = 0; si < sc.size(); si++) {
          const Comp& c = comps[sc[si].ci];
          for (int by = 0; by < c.v; by++)
            for (int bx = 0; bx < c.h; bx++) {
              int16_t* blk = sc[si].ci == 0 ? &coef[((size_t)(uy * c.v + by) * bw + (size_t)(ux * c.h + bx)) * 64] : nullptr;
              one_block(blk, sc[si], (int)si);
            }
        }
      } else {
        one_block(&coef[((size_t)uy * bw + ux) * 64], sc[0], 0);
      }
      if (restart_interval) until_restart--;
    }
  return 0;
}

// `sequential`: the frame is SOF0 / SOF1 and its components come in several scans (full-band scans, Ss = 0, Se = 63, no
// successive approximation) -- libjpeg reads those through the same coefficient buffer (jdcoefct.c, has_multiple_scans).
int decode_progressive(const uint8_t* data, size_t nbytes, uint8_t* out, size_t ostride, int cap_w, int cap_h, int* w_out,
                       int* h_out, bool sequential = false) {
  uint16_t qt[4][64], yq[64];
  bool qt_present[4] = {false, false, false, false};
  bool yq_latched = false;
  Huff dc[4], ac[4];
  std::vector<Comp> comps;
  int W = 0, H = 0, restart_interval = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0, bw = 0, bh = 0;
  bool have_sof = false;
  std::vector<int16_t> coef;
  int cbits[64];  // precision still missing per luminance coefficient (jdphuff.c coef_bits): -1 = nothing received yet
  for (int& b : cbits) b = -1;
  size_t pos = 2;
  while (pos + 2 <= nbytes) {
    if (data[pos] != 0xFF) return -1;
    while (pos < nbytes && data[pos] == 0xFF) pos++;
    if (pos >= nbytes) break;
    const int m = data[pos++];
    if (m == 0xD9) break;
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (pos + 2 > nbytes) return -1;
    const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
    if (len < 2 || pos + len > nbytes) return -1;
    const uint8_t* s = data + pos + 2;
    const size_t n = len - 2;
    if (m == 0xDB) {
      if (parse_dqt(s, n, qt, qt_present)) return -1;
    } else if (m == 0xC4) {
      if (parse_dht(s, n, dc, ac)) return -1;
    } else if (m == (sequential ? 0xC0 : 0xC2) || (sequential && m == 0xC1)) {
      if (have_sof) return -1;
      if (n < 6 || s[0] != 8) return -2;
      H = (s[1] << 8) | s[2];
      W = (s[3] << 8) | s[4];
      const int nf = s[5];
      if (W < 1 || H < 1 || (nf != 1 && nf != 3) || n < (size_t)(6 + 3 * nf)) return nf == 4 ? -2 : -1;
      comps.resize(nf);
      for (int c = 0; c < nf; c++) {
        comps[c].id = s[6 + 3 * c];
        comps[c].h = s[7 + 3 * c] >> 4;
        comps[c].v = s[7 + 3 * c] & 15;
        comps[c].tq = s[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) return -1;
        hmax = comps[c].h > hmax ? comps[c].h : hmax;
        vmax = comps[c].v > vmax ? comps[c].v : vmax;
      }
      if (comps[0].h != hmax || comps[0].v != vmax) return -2;  // luminance would need upsampling
      if (w_out) *w_out = W;
      if (h_out) *h_out = H;
      if (W > cap_w || H > cap_h || !out) return -3;
      mcus_x = (W + 8 * hmax - 1) / (8 * hmax);
      mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
      bw = mcus_x * hmax;
      bh = mcus_y * vmax;
      coef.assign((size_t)bw * bh * 64, 0);
      have_sof = true;
    } else if (m >= 0xC0 && m <= 0xCF && m != 0xC8 && m != 0xCC) {
      return have_sof ? -1 : -2;
    } else if (m == 0xDD) {
      if (n < 2) return -1;
      restart_interval = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (!have_sof || n < 1) return -1;
      const int ns = s[0];
      if (ns < 1 || ns > (int)comps.size() || n < (size_t)(4 + 2 * ns)) return -1;
      std::vector<ScanComp> sc((size_t)ns);
      bool has_luma = false;
      for (int c = 0; c < ns; c++) {
        int ci = -1;
        for (size_t f = 0; f < comps.size(); f++)
          if (comps[f].id == s[1 + 2 * c]) ci = (int)f;
        if (ci < 0) return -1;
        for (int e = 0; e < c; e++)
          if (sc[e].ci >= ci) return -1;  // components of a scan follow the frame's order (B.2.3)
        sc[c] = {ci, s[2 + 2 * c] >> 4, s[2 + 2 * c] & 15};
        if (sc[c].td > 3 || sc[c].ta > 3) return -1;
        has_luma |= ci == 0;
      }
      const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
      if (ns > 1) {  // jdinput.c per_scan_setup: D_MAX_BLOCKS_IN_MCU = 10 (JERR_BAD_MCU_SIZE)
        int blocks = 0;
        for (const ScanComp& c : sc) blocks += comps[(size_t)c.ci].h * comps[(size_t)c.ci].v;
        if (blocks > 10) return -1;
      }
      if (Ss == 0 && Ah == 0)
        for (const ScanComp& c : sc)
          if (dc[c.td].present && !dc[c.td].dc_symbols_ok()) return -1;  // JERR_BAD_HUFF_TABLE
      if (sequential) {
        if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return -1;
      } else {  // jdphuff.c start_pass_phuff_decoder: the legal shapes of a progressive scan
        if (Ss == 0 ? Se != 0 : (ns != 1 || Se < Ss || Se > 63)) return -1;
        if ((Ah != 0 && Al != Ah - 1) || Al > 13) return -1;
      }
      const size_t begin = pos + len, end = ecs_end(data, begin, nbytes);
      if (has_luma) {
        for (int k = Ss; k <= Se; k++) {  // every scan must continue where the previous one over this coefficient stopped
          if (Ah != (cbits[k] < 0 ? 0 : cbits[k]) || (cbits[k] >= 0 && Ah == 0)) return -2;
          cbits[k] = Al;
        }
        if (Ss > 0 && cbits[0] < 0) return -2;  // AC before any DC scan
        for (const ScanComp& c : sc)
          if (sequential ? (!dc[c.td].present || !ac[c.ta].present)
                         : (Ss == 0 ? (Ah == 0 && !dc[c.td].present) : !ac[c.ta].present))
            return -1;
        if (!yq_latched) {  // jdinput.c latch_quant_tables: the table in force at the component's first scan
          if (!qt_present[comps[0].tq]) return -1;
          std::memcpy(yq, qt[comps[0].tq], sizeof(yq));
          yq_latched = true;
        }
        const int rc = decode_prog_scan(data + begin, data + end, comps, sc, dc, ac, Ss, Se, Ah, Al, restart_interval, W, H,
                                        hmax, vmax, mcus_x, mcus_y, bw, coef);
        if (rc) return rc;
      }
      pos = end;
      continue;
    }
    pos += len;
  }
  if (!have_sof || !yq_latched) return -1;
  for (int k = 0; k < 64; k++)
    if (cbits[k] != 0) return -2;  // the scans stop short of full precision
  const int pw = bw * 8, ph = bh * 8;
  std::vector<uint8_t> plane((size_t)pw * ph);
  for (int by = 0; by < bh; by++)
    for (int bx = 0; bx < bw; bx++) {
      int32_t deq[64];
      const int16_t* blk = &coef[((size_t)by * bw + bx) * 64];
      for (int k = 0; k < 64; k++) deq[k] = (int32_t)blk[k] * (int32_t)yq[k];
      idct_islow(deq, &plane[(size_t)by * 8 * pw + (size_t)bx * 8], (size_t)pw);
    }
  for (int y = 0; y < H; y++) std::memcpy(out + (size_t)y * ostride, &plane[(size_t)y * pw], (size_t)W);
  return 0;
}

}  // namespace

namespace {
// What jpeg_finish_decompress reads behind the data of a file's one scan (jdmarker.c read_markers, jdinput.c
// consume_markers): every marker from the one the entropy decoder stopped at to EOI.  Most are skipped or parsed quietly; some
// end the read with a fatal error -- cv::imdecode then returns NOTHING although every row had been decoded: a marker code
// libjpeg does not know (0x02..0xBF, 0xDE, 0xDF, 0xF0..0xFD: JERR_UNKNOWN_MARKER), another frame or a frame type it does
// not read, another SOI, another SOS (JERR_EOI_EXPECTED after its header has parsed), a table segment that does not parse.
// Bytes behind the end of the file read as the memory source supplies them: 0xFF 0xD9 over and over.
// p[0..n): everything behind the scan header; restart_interval > 0: RSTn markers in the data belong to the decoder.
// Returns false for a file libjpeg gives up on.
static bool baseline_tail_ok(const uint8_t* p, size_t n, int restart_interval) {
  auto at = [&](size_t i) -> int { return i < n ? p[i] : (((i - n) & 1) ? 0xD9 : 0xFF); };
  size_t i = 0;
  const size_t hard_end = n + 4096;  // (the virtual tail is EOI after EOI: a walk that gets this far has met one)
  // where the entropy decoder stops: the first marker in the data
  int m = -1;
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
    if (restart_interval > 0 && c >= 0xD0 && c <= 0xD7) continue;        // the decoder's own
    m = c;
    break;
  }
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
}  // namespace

extern "C" int vsfo_jpeg_decode_gray(const uint8_t* data, size_t nbytes, uint8_t* out, size_t ostride, int cap_w,
                                     int cap_h, int* w_out, int* h_out) {
  // returns 0 ok; -1 malformed; -2 a JPEG process this decoder does not restate (arithmetic, 12 bit, lossless,
  // progressive files whose scans stop short of full precision); -3 the image does not fit cap_w x cap_h
  if (!data || nbytes < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  std::vector<Comp> comps;
  int W = 0, H = 0, restart_interval = 0;
  size_t pos = 2;
  bool have_sof = false;
  while (pos + 4 <= nbytes) {
    if (data[pos] != 0xFF) return -1;
    while (pos < nbytes && data[pos] == 0xFF) pos++;  // fill bytes
    if (pos >= nbytes) return -1;
    const int m = data[pos++];
    if (m == 0xD9) return -1;  // EOI before any scan
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (pos + 2 > nbytes) return -1;
    const size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
    if (len < 2 || pos + len > nbytes) return -1;
    const uint8_t* s = data + pos + 2;
    const size_t n = len - 2;
    if (m == 0xDB) {  // DQT
      size_t i = 0;
      while (i < n) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        i++;
        if (tq > 3 || pq > 1 || i + 64 * (pq + 1) > n) return -1;
        for (int k = 0; k < 64; k++) {
          qt[tq][kZigzag[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
          i += pq + 1;
        }
        qt_present[tq] = true;
      }
    } else if (m == 0xC4) {  // DHT
      size_t i = 0;
      while (i < n) {
        if (i + 17 > n) return -1;
        const int tc = s[i] >> 4, th = s[i] & 15;
        if (tc > 1 || th > 3) return -1;
        Huff& h = tc ? ac[th] : dc[th];
        int total = 0;
        for (int l = 1; l <= 16; l++) total += (h.bits[l] = s[i + l]);
        i += 17;
        if (total > 256 || i + total > n) return -1;
        std::memset(h.vals, 0, sizeof(h.vals));
        std::memcpy(h.vals, s + i, total);
        i += total;
        h.present = true;
        if (!h.derive()) return -1;
      }
    } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1: sequential DCT, Huffman
      if (n < 6 || s[0] != 8) return -2;
      H = (s[1] << 8) | s[2];
      W = (s[3] << 8) | s[4];
      const int nf = s[5];
      if (W < 1 || H < 1 || (nf != 1 && nf != 3) || n < (size_t)(6 + 3 * nf)) return nf == 4 ? -2 : -1;
      comps.resize(nf);
      for (int c = 0; c < nf; c++) {
        comps[c].id = s[6 + 3 * c];
        comps[c].h = s[7 + 3 * c] >> 4;
        comps[c].v = s[7 + 3 * c] & 15;
        comps[c].tq = s[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) return -1;
      }
      have_sof = true;
    } else if (m == 0xC2) {
      return have_sof ? -1 : decode_progressive(data, nbytes, out, ostride, cap_w, cap_h, w_out, h_out);
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return -2;  // lossless, arithmetic, hierarchical
    } else if (m == 0xDD) {  // DRI
      if (n < 2) return -1;
      restart_interval = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {  // SOS: the one scan
      if (!have_sof || n < 1) return -1;
      const int ns = s[0];
      if (ns != (int)comps.size())  // the components come in several scans
        return decode_progressive(data, nbytes, out, ostride, cap_w, cap_h, w_out, h_out, true);
      if (n < (size_t)(4 + 2 * ns)) return -1;
      for (int c = 0; c < ns; c++) {
        if (s[1 + 2 * c] != comps[c].id) return -2;
        comps[c].td = s[2 + 2 * c] >> 4;
        comps[c].ta = s[2 + 2 * c] & 15;
        if (comps[c].td > 3 || comps[c].ta > 3 || !dc[comps[c].td].present || !ac[comps[c].ta].present ||
            !qt_present[comps[c].tq])
          return -1;
        if (!dc[comps[c].td].dc_symbols_ok()) return -1;  // JERR_BAD_HUFF_TABLE
      }
      if (ns > 1) {  // jdinput.c per_scan_setup: D_MAX_BLOCKS_IN_MCU = 10 (JERR_BAD_MCU_SIZE)
        int blocks = 0;
        for (const Comp& c : comps) blocks += c.h * c.v;
        if (blocks > 10) return -1;
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return -2;
      pos += len;
      break;
    }
    pos += len;
  }
  if (!have_sof || comps.empty() || pos >= nbytes) return -1;
  if (!baseline_tail_ok(data + pos, nbytes - pos, restart_interval)) return -1;
  if (w_out) *w_out = W;
  if (h_out) *h_out = H;
  if (W > cap_w || H > cap_h || !out) return -3;
  int hmax = 1, vmax = 1;
  for (const Comp& c : comps) {
    hmax = c.h > hmax ? c.h : hmax;
    vmax = c.v > vmax ? c.v : vmax;
  }
  // libjpeg would upsample a luminance component that is not sampled at full rate; such files do not occur (and a
  // single-component scan is always one block per MCU, whatever its sampling factors say: T.81 A.2.2)
  if (comps.size() > 1 && (comps[0].h != hmax || comps[0].v != vmax)) return -2;
  const bool single = comps.size() == 1;
  const int mcu_w = single ? 8 : 8 * hmax, mcu_h = single ? 8 : 8 * vmax;
  const int mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
  const int yh = single ? 1 : comps[0].h, yv = single ? 1 : comps[0].v;
  const int pw = mcus_x * mcu_w, ph = mcus_y * mcu_h;  // padded luminance plane
  std::vector<uint8_t> plane((size_t)pw * ph);
  BitReader br{data + pos, data + nbytes};
  int pred[4] = {0, 0, 0, 0};
  int until_restart = restart_interval, next_restart = 0;
  for (int my = 0; my < mcus_y; my++)
    for (int mx = 0; mx < mcus_x; mx++) {
      if (restart_interval && until_restart == 0) {
        br.restart_as_libjpeg(next_restart);
        pred[0] = pred[1] = pred[2] = pred[3] = 0;
        until_restart = restart_interval;
      }
      // jdhuff.c decode_mcu: "If we've run out of data, just leave the MCU set to zeroes.  This way, we return uniform gray
      // for the remainder of the segment."  The MCU in which the data ran out was finished on zero bits.
      const bool skipped = br.starved;
      for (size_t ci = 0; ci < comps.size(); ci++) {
        const Comp& c = comps[ci];
        const int bh = single ? 1 : c.h, bv = single ? 1 : c.v;
        for (int by = 0; by < bv; by++)
          for (int bx = 0; bx < bh; bx++) {
            int32_t coef[64];
            std::memset(coef, 0, sizeof(coef));
            if (skipped) {
              if (ci == 0) idct_islow(coef, &plane[(size_t)(my * yv * 8 + by * 8) * pw + (size_t)(mx * yh * 8 + bx * 8)], (size_t)pw);
              continue;
            }
            // F.2.2.1 DC, F.2.2.2 AC
            const int t = br.decode(dc[c.td]);
            pred[ci] += extend(br.receive(t), t);
            coef[0] = pred[ci] * (int32_t)qt[c.tq][0];
            for (int k = 1; k < 64;) {
              const int rs = br.decode(ac[c.ta]);
              const int r = rs >> 4, sz = rs & 15;
              if (sz == 0) {
                if (r != 15) break;  // EOB
                k += 16;
                continue;
              }
              k += r;
              const int nat = kZigzag[k > 63 ? 63 : k];  // (past the block's end: the last coefficient, as libjpeg's padded order table has it)
              coef[nat] = extend(br.receive(sz), sz) * (int32_t)qt[c.tq][nat];
              k++;
            }
            if (ci == 0)  // only the luminance is reconstructed (component_needed, jdmaster.c / jdapimin.c)
              idct_islow(coef, &plane[(size_t)(my * yv * 8 + by * 8) * pw + (size_t)(mx * yh * 8 + bx * 8)], (size_t)pw);
          }
      }
      if (restart_interval) until_restart--;
    }
  for (int y = 0; y < H; y++) std::memcpy(out + (size_t)y * ostride, &plane[(size_t)y * pw], (size_t)W);
  return 0;
}
