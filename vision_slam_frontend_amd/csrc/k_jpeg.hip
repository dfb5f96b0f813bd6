// k_jpeg.hip -- SURVEY.md section 8(f) row f4, the decode in front of the Bayer conversion: DecodeImage's
//   cv::imdecode(msg.data, cv::IMREAD_GRAYSCALE)                                   (slam_frontend_main.cc:99-100)
// for the baseline-JPEG payloads of sensor_msgs::CompressedImage, a batch of images per call, result in HBM.
//
// What OpenCV 3.2 does there (imgcodecs/src/grfmt_jpeg.cpp -> libjpeg, out_color_space = JCS_GRAYSCALE, default
// JDCT_ISLOW): ITU-T T.81 baseline entropy decoding and, for the luminance component only, libjpeg's jidctint.c inverse
// DCT (13-bit constants, PASS1_BITS = 2, exact 32-bit integers) + range limit; chroma is parsed and dropped.
//
// Two decoders, chosen per file.  The PARALLEL one further down (jpeg_par_decode_kernel) takes nearly everything: 256
// threads per image decode 256 segments of the stream speculatively and iterate until their states agree, or -- files
// with restart intervals -- decode the intervals, whose start states are known, side by side.  Files whose Huffman
// tables need more second-level lookup tables than DevHuff holds, files with more restart intervals than the scratch
// has room to list -- and everything under VSF_OPT_JPEG_SERIAL -- take the one described here: the entropy-coded segment is walked serially, ONE WAVE PER IMAGE -- its SCALAR unit walks the Huffman
// codes (wave-uniform code: state in SGPRs, stream words and 9-bit lookahead tables through scalar loads, T.81 F.2.2.3
// for longer codes; FF00 unstuffing; RSTn / DC-prediction resets), drops chroma blocks and parks up to 16 luminance
// blocks of coefficients in LDS; then the 64 lanes dequantise and run the two IDCT passes (lane = block x column, then
// block x row) and store the pixels -- and a batch runs as many waves as it has images: a latency-bound kernel that
// occupies a few percent of the chip's issue slots, scales with the images in flight up to ~2500, and is meant to run on
// a stream of its own BESIDE the extraction of earlier batches.  Markers and tables are parsed on the host (the
// compressed bytes come from host memory anyway), which also builds the lookahead tables once per distinct table set.
//
// PROGRESSIVE files (SOF2: what a web service or an image library writes, not a camera driver) take a third decoder:
// prog_scan below decodes one luminance scan into the coefficient buffer (T.81 Annex G as libjpeg's jdphuff.c does);
// jpeg_prog_pipe_kernel runs the scans of a file in the waves of one workgroup, each refinement scan a block row behind
// the scans it refines, jpeg_prog_kernel one wave per file scan after scan (large batches, and damaged files again); the
// IDCT kernel of the parallel decoder finishes the job.
//
// Checked bit for bit against JPEG files decoded by libjpeg-turbo (tests/golden/jpeg, tests/test_gpu_jpeg.py).
// Sequential files whose components come in several scans take that kernel too (a scan then decodes a block whole).
// Arithmetic / 12-bit / lossless files and progressive files whose scans stop short of full precision are refused
// (VSF_ERR_UNSUPPORTED).
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

__constant__ uint8_t c_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Everything the entropy decoders READ -- stream words, Huffman tables, image and scan descriptors -- was written before the
// kernel started and never changes under it.  Read through the CONSTANT address space, a wave-uniform address is fetched
// by a scalar load whatever else the kernel does; through ordinary global pointers a kernel that also fences or uses
// atomics (the progressive one) gets vector loads + v_readfirstlane for the same reads.  (What DOES cost a factor is
// reader state that ends up in scratch memory -- a closure indexed at run time, a reference picked by a run-time select:
// private loads count as different per lane, and the whole bit walk then runs on the vector unit under exec masks.  The
// progressive kernel below is written to avoid that: 23 VGPRs, no scratch.)
template <class T>
using cptr = const __attribute__((address_space(4))) T*;
template <class T>
__device__ __forceinline__ cptr<T> in_constant(const T* p) {
  return (cptr<T>)(uintptr_t)p;
}

// ---- lane 0's view of the entropy-coded segment ----
// ---- the entropy decoder's view of the segment.  Everything in here is WAVE-UNIFORM: all 64 lanes run the same decode
// with the same values, so the compiler keeps the state in SGPRs, the bit arithmetic on the scalar ALU (one cycle per
// instruction instead of four, short dependent latency) and fetches stream words and table entries with scalar loads
// through the constant cache (dword loads only on gfx950: bytes and 16-bit entries are cut out of their dword).  A first
// version that let lane 0 alone walk the codes through LDS tables took ~330 cycles per symbol; this one takes ~1/3.
struct BitReader {
  const uint32_t* words;  // the entropy-coded segment in HBM, as dwords (its offset in the packed buffer is 4-aligned)
  uint32_t pos, len;      // next raw byte, segment length
  uint64_t acc;           // the next `n` bits of the de-stuffed stream sit in the low n bits, oldest on top
  int n;
  bool marker;            // a marker has been met: zero bits are fed from here on (until restart())
  int pad = 0;            // zero bits fed so far: the walk has run out of data once it has taken one of them (starved())
  bool dry = false;       // ... and stays so over a restart whose marker was not the one due (restart_as_libjpeg)
  __device__ __forceinline__ uint32_t at(uint32_t p) const { return (in_constant(words)[p >> 2] >> (8u * (p & 3u))) & 255u; }
  __device__ __forceinline__ void fill() {  // >= 33 bits available afterwards (a code + its extra bits need <= 16 + 15)
    if (n <= 32 && !marker && pos + 4u <= len) {
      // four raw bytes at once unless one of them is 0xFF (stuffing or a marker: the byte-wise path sorts it out)
      const uint64_t two = (uint64_t)in_constant(words)[pos >> 2] | ((uint64_t)in_constant(words)[(pos >> 2) + 1] << 32);
      const uint32_t v = (uint32_t)(two >> (8u * (pos & 3u)));
      const uint32_t nv = ~v;
      if ((((nv - 0x01010101u) & ~nv) & 0x80808080u) == 0u) {  // no byte of v is 0xFF
        const uint32_t be = (v << 24) | ((v & 0xFF00u) << 8) | ((v >> 8) & 0xFF00u) | (v >> 24);
        acc = (acc << 32) | be;
        n += 32;
        pos += 4;
        return;
      }
    }
    while (n <= 32) {
      uint32_t b = 0;
      if (!marker) {
        if (pos < len) {
          b = at(pos);
          if (b == 0xFFu) {
            const uint32_t nx = pos + 1 < len ? at(pos + 1) : 0xD9u;
            if (nx == 0) {
              pos += 2;
            } else {
              marker = true;
              b = 0;
            }
          } else {
            pos++;
          }
        } else {
          marker = true;
        }
      }
      if (marker) pad += 8;
      acc = (acc << 8) | b;
      n += 8;
    }
  }
  // jdhuff.c's insufficient_data: a bit has been taken that the data did not hold
  __device__ __forceinline__ bool starved() const { return dry || n < pad; }
  __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)(acc >> (n - k)) & ((1u << k) - 1u); }
  __device__ __forceinline__ void drop(int k) { n -= k; }
  __device__ __forceinline__ int get_bit() {  // one raw bit
    fill();
    const int b = (int)peek(1);
    drop(1);
    return b;
  }
  __device__ __forceinline__ int receive(int k) {  // k <= 16 raw bits
    if (k == 0) return 0;
    fill();
    const int v = (int)peek(k);
    drop(k);
    return v;
  }
  __device__ __forceinline__ int receive_extend(int s) {  // T.81 F.2.2.1 RECEIVE + EXTEND
    if (s == 0) return 0;
    const int v = (int)peek(s);
    drop(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
  }
  template <class Table>
  __device__ __forceinline__ int decode(const Table* h) {
    const uint32_t p = peek(kLookBits);
    return finish(h, (in_constant(reinterpret_cast<const uint32_t*>(h->look))[p >> 1] >> (16u * (p & 1u))) & 0xFFFFu);
  }
  // the rest of a decode once the lookahead entry `e` of the next 9 bits is at hand
  template <class Table>
  __device__ __forceinline__ int finish(const Table* h, uint32_t e) {
    if (!(e & kLongCode)) {
      drop((int)(e >> 8));
      return (int)(e & 255u);
    }
    int l = kLookBits + 1;
    int32_t code = (int32_t)peek(l);
    while (l <= 16 && code > in_constant(h->maxcode)[l]) {
      l++;
      code = (int32_t)peek(l <= 16 ? l : 16);
    }
    if (l > 16) {  // no code: libjpeg's walk stops at the 17-bit sentinel, symbol 0 (the fast path without a warning)
      drop(17);
      return 0;
    }
    drop(l);
    const uint32_t vi = (uint32_t)(in_constant(h->valoff)[l] + code) & 255u;
    return (int)((in_constant(reinterpret_cast<const uint32_t*>(h->vals))[vi >> 2] >> (8u * (vi & 3u))) & 255u);
  }
  // A restart of a sequential scan as libjpeg makes it (jdhuff.c process_restart, jdmarker.c read_restart_marker and
  // jpeg_resync_to_restart): the bits left are dropped and the next marker is looked for (bytes in front of it are passed
  // over).  The restart marker that is due (`next`: 0..7) is taken and the interval has its data.  Otherwise: a code below
  // 0xC0 is passed over and the search goes on; any other marker that is no RSTn stays where it is, and so does an RSTn one
  // or two numbers AHEAD of the one due -- the interval is decoded from no data and the marker is met again at the next
  // boundary; an RSTn one or two numbers BEHIND is passed over; any other RSTn is taken as if it were the one due.
  // "Out of data" (insufficient_data) is cleared only when no marker is left unread.
  __device__ __forceinline__ void restart_as_libjpeg(int& next) {
    const bool was = starved();
    acc = 0;
    n = 0;
    pad = 0;
    marker = false;
    bool take = false;
    for (;;) {
      while (pos < len && at(pos) != 0xFFu) pos++;
      uint32_t q = pos;
      while (q < len && at(q) == 0xFFu) q++;
      const int m = q < len ? (int)at(q) : 0xD9;  // (the memory source ends every file with an EOI of its own)
      if (q < len && m == 0) {                     // a stuffed 0xFF is no marker
        pos = q + 1;
        continue;
      }
      bool leave = false;
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
        else if (ahead != 7 && ahead != 6)
          take = true;
      }
      if (take) {
        pos = q < len ? q + 1 : len;
        break;
      }
      if (leave) {
        marker = true;  // (pos stays at the marker: no data until it has been dealt with)
        break;
      }
      pos = q < len ? q + 1 : len;
      if (pos >= len) {  // (only EOIs from here on: one of them is "left")
        marker = true;
        break;
      }
    }
    dry = take ? false : was;
    next = (next + 1) & 7;
  }
  __device__ __forceinline__ bool restart() {  // drop the remaining bits, step over RSTn (the progressive decoder's)
    // (fill() never pulls bytes from beyond a marker, so `pos` is at the marker when the interval's data is used up)
    acc = 0;
    n = 0;
    marker = false;
    pad = 0;
    while (pos + 1 < len) {
      const uint32_t a = at(pos), b = at(pos + 1);
      if (a == 0xFFu && b >= 0xD0u && b <= 0xD7u) {
        pos += 2;
        return true;
      }
      if (a == 0xFFu && b != 0u && b != 0xFFu) return false;
      pos++;
    }
    return false;
  }
};

__device__ __forceinline__ uint8_t range_limit(int32_t x) {  // sample_range_limit + CENTERJSAMPLE, index x & 1023
  const int t = x & 1023;
  return (uint8_t)(t < 128 ? t + 128 : (t < 512 ? 255 : (t < 896 ? 0 : t - 896)));
}

// jidctint.c: one 8-point pass on d[0..7]; r[k] are the values before DESCALE
template <class T>
__device__ __forceinline__ void idct8_t(const T d[8], T r[8]) {
  const T F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
          F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  T z2 = d[2], z3 = d[6];
  T z1 = (z2 + z3) * F0541;
  T tmp2 = z1 + z3 * (-F1847);
  T tmp3 = z1 + z2 * F0765;
  z2 = d[0];
  z3 = d[4];
  T tmp0 = (z2 + z3) * (T)8192;
  T tmp1 = (z2 - z3) * (T)8192;
  const T tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = d[7];
  tmp1 = d[5];
  tmp2 = d[3];
  tmp3 = d[1];
  z1 = tmp0 + tmp3;
  z2 = tmp1 + tmp2;
  z3 = tmp0 + tmp2;
  T z4 = tmp1 + tmp3;
  const T z5 = (z3 + z4) * F1175;
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
  r[0] = tmp10 + tmp3;
  r[1] = tmp11 + tmp2;
  r[2] = tmp12 + tmp1;
  r[3] = tmp13 + tmp0;
  r[4] = tmp13 - tmp0;
  r[5] = tmp12 - tmp1;
  r[6] = tmp11 - tmp2;
  r[7] = tmp10 - tmp3;
}
// One pass with its DESCALE(., shift), as jidctint.c computes it where JLONG is 64 bits wide (long on every LP64 build of
// libjpeg): the sums cannot overflow there, and the result is cut to int afterwards.  An output is at most 11363 * sum |d|
// (the largest row of the 13-bit basis), so with every |d| below 2^14 -- any block an encoder writes -- 32-bit arithmetic
// gives the same bits (unsigned: wrap-around on the way is harmless, the sums are ring operations); blocks beyond that
// (damaged files) take the 64-bit path.
struct Idct8 {
  int32_t v[8];
};
__device__ __attribute__((noinline)) Idct8 idct8_wide(Idct8 in, int shift) {  // (out of line: damaged files only)
  long long w[8], r[8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = in.v[k];
  idct8_t<long long>(w, r);
  Idct8 out;
#pragma unroll
  for (int k = 0; k < 8; k++) out.v[k] = (int32_t)((r[k] + (1ll << (shift - 1))) >> shift);
  return out;
}
__device__ __forceinline__ void idct8_descaled(const int32_t d[8], int32_t out[8], int shift) {
  uint32_t most = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) most |= (uint32_t)(d[k] ^ (d[k] >> 31));  // (ones' complement magnitudes: a bound on |d| is all it takes)
  if (most < (1u << 14)) {
    uint32_t u[8], r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) u[k] = (uint32_t)d[k];
    idct8_t<uint32_t>(u, r);
#pragma unroll
    for (int k = 0; k < 8; k++) out[k] = (int32_t)(r[k] + (1u << (shift - 1))) >> shift;
  } else {
    Idct8 in;
#pragma unroll
    for (int k = 0; k < 8; k++) in.v[k] = d[k];
    const Idct8 o = idct8_wide(in, shift);
#pragma unroll
    for (int k = 0; k < 8; k++) out[k] = o.v[k];
  }
}

// `index`: the images (positions in `images` and in the destination) this launch decodes, one per workgroup
__global__ __launch_bounds__(64) void jpeg_gray_kernel(const DevImage* __restrict__ images,
                                                        const uint32_t* __restrict__ index,
                                                        const DevTables* __restrict__ tables,
                                                        const uint8_t* __restrict__ stream, uint32_t stream_total,
                                                        int width, int height,
                                                        uint8_t* __restrict__ dst, size_t dst_image_stride,
                                                        int dst_pitch, int32_t* __restrict__ status) {
  __shared__ uint16_t s_qt[64];
  __shared__ __attribute__((aligned(16))) int16_t s_coef[kGroupBlocks][64];
  __shared__ __attribute__((aligned(16))) int32_t s_ws[kGroupBlocks][64];
  __shared__ int32_t s_dest[kGroupBlocks];  // y0 << 16 | x0
  const int lane = threadIdx.x;
  const uint32_t image = index[blockIdx.x];
  const DevImage& im = images[image];
  const DevTables* tab = tables + im.tables;
  s_qt[lane] = tab->qt_luma[lane];
  uint8_t* out = dst + (size_t)image * dst_image_stride;
  // geometry and decoder state: wave-uniform (see BitReader)
  const int ncomp = im.ncomp, restart_interval = im.restart_interval, mcus_x = im.mcus_x;
  const int nmcu = mcus_x * im.mcus_y;
  const int h0 = im.h[0], v0 = im.v[0], luma_per_mcu = h0 * v0;
  const int nblk1 = ncomp > 1 ? im.h[1] * im.v[1] : 0, nblk2 = ncomp > 2 ? im.h[2] * im.v[2] : 0;
  const DevHuff *dc0 = &tab->huff[im.dc_slot[0]], *ac0 = &tab->huff[im.ac_slot[0]];
  const DevHuff *dc1 = &tab->huff[im.dc_slot[ncomp > 1 ? 1 : 0]], *ac1 = &tab->huff[im.ac_slot[ncomp > 1 ? 1 : 0]];
  const DevHuff *dc2 = &tab->huff[im.dc_slot[ncomp > 2 ? 2 : 0]], *ac2 = &tab->huff[im.ac_slot[ncomp > 2 ? 2 : 0]];
  const uint32_t* zz32 = reinterpret_cast<const uint32_t*>(c_zigzag);
  BitReader br{reinterpret_cast<const uint32_t*>(stream + im.stream_off), 0u, im.stream_len, 0ull, 0, false};
  (void)stream_total;
  int pred0 = 0, pred1 = 0, pred2 = 0;
  int mcu = 0, until_restart = restart_interval, next_restart = 0;
  bool broken = false;
  // one block: DC difference + AC run / size pairs (T.81 F.2.2.1, F.2.2.2); coefficients go to `coef` in natural order
  // (LUMA) or nowhere (chroma is parsed only)
  auto block = [&](const DevHuff* hd, const DevHuff* ha, int& pred, int16_t* coef) {
    br.fill();
    const int t = br.decode(hd) & 15;  // (the host refuses DC tables with symbols above 15, as libjpeg does: second line of defence)
    br.fill();
    pred += br.receive_extend(t);
    if (coef) coef[0] = (int16_t)pred;
    for (int k = 1; k < 64;) {
      br.fill();
      const int rs = br.decode(ha);
      const int r = rs >> 4, sz = rs & 15;
      if (sz == 0) {
        if (r != 15) break;  // EOB
        k += 16;
        continue;
      }
      k += r;
      const int val = br.receive_extend(sz);
      // (damaged data can run past the block's end: libjpeg's jpeg_natural_order has sixteen spare entries that all say 63,
      // so the value lands on the last coefficient and the block ends)
      if (coef) coef[k > 63 ? 63u : (zz32[k >> 2] >> (8 * (k & 3))) & 63u] = (int16_t)val;
      k++;
    }
  };
  __syncthreads();
  for (;;) {
    {  // clear the coefficient buffer
      uint32_t* c = reinterpret_cast<uint32_t*>(&s_coef[0][0]);
      for (int i = lane; i < kGroupBlocks * 32; i += 64) c[i] = 0u;
    }
    __syncthreads();
    // ---- entropy decoding of as many whole MCUs as fit the coefficient buffer: uniform, every lane the same ----
    int count = 0;
    while (mcu < nmcu && count + luma_per_mcu <= kGroupBlocks && !broken) {
      if (restart_interval && until_restart == 0) {
        br.restart_as_libjpeg(next_restart);
        pred0 = pred1 = pred2 = 0;
        until_restart = restart_interval;
      }
      const int my = mcu / mcus_x, mx = mcu - my * mcus_x;
      // jdhuff.c decode_mcu: "If we've run out of data, just leave the MCU set to zeroes.  This way, we return uniform gray
      // for the remainder of the segment."  (The MCU in which the data ran out is finished on zero bits.)
      const bool skipped = br.starved();
      for (int by = 0; by < v0; by++)
        for (int bx = 0; bx < h0; bx++) {
          if (!skipped) block(dc0, ac0, pred0, s_coef[count]);
          s_dest[count] = ((my * v0 * 8 + by * 8) << 16) | (mx * h0 * 8 + bx * 8);
          count++;
        }
      if (!skipped) {
        for (int b = 0; b < nblk1; b++) block(dc1, ac1, pred1, nullptr);
        for (int b = 0; b < nblk2; b++) block(dc2, ac2, pred2, nullptr);
      }
      if (restart_interval) until_restart--;
      mcu++;
    }
    const bool done = mcu >= nmcu || broken;
    __syncthreads();
    // ---- dequantise + IDCT of the parked blocks: 8 blocks x 8 columns, then 8 blocks x 8 rows, per round ----
    for (int b0 = 0; b0 < count; b0 += 8) {
      const int b = b0 + (lane >> 3), i = lane & 7;
      if (b < count) {
        int32_t d[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = (int32_t)s_coef[b][8 * k + i] * (int32_t)s_qt[8 * k + i];
        idct8_descaled(d, r, 11);  // DESCALE(., CONST_BITS - PASS1_BITS)
#pragma unroll
        for (int k = 0; k < 8; k++) s_ws[b][8 * k + i] = r[k];
      }
      __syncthreads();
      if (b < count) {
        int32_t d[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = s_ws[b][8 * i + k];
        idct8_descaled(d, r, 18);  // DESCALE(., 13 + 2 + 3)
        const int x0 = s_dest[b] & 0xFFFF, y = (s_dest[b] >> 16) + i;
        if (y < height && x0 < width) {
          uint8_t px[8];
#pragma unroll
          for (int k = 0; k < 8; k++) px[k] = range_limit(r[k]);
          uint8_t* row = out + (size_t)y * dst_pitch + x0;
          if (x0 + 8 <= width) {
            uint32_t lo, hi;
            memcpy(&lo, px, 4);
            memcpy(&hi, px + 4, 4);
            reinterpret_cast<uint32_t*>(row)[0] = lo;  // (x0 % 8 == 0, base and pitch are multiples of 4)
            reinterpret_cast<uint32_t*>(row)[1] = hi;
          } else {
            for (int k = 0; k < 8 && x0 + k < width; k++) row[k] = px[k];
          }
        }
      }
      __syncthreads();
    }
    if (done) break;
  }
  if (lane == 0 && broken) atomicOr(status, 2);
}

// ---------------------------------------------------------------------------------------------------------------------
// Progressive files (T.81 Annex G; libjpeg jdphuff.c decode_mcu_DC_first / _DC_refine / _AC_first / _AC_refine).
// One wave per file.  The host lists the scans that carry the luminance component (chroma-only scans cannot change a gray
// read); the wave decodes them in file order into the file's slot of the coefficient buffer -- blocks in the order of the
// frame's interleaved MCUs, natural order inside a block: what jpeg_idct_kernel reads -- and that kernel turns them into
// pixels afterwards.  The bit parsing is WAVE-UNIFORM as in the one-wave sequential decoder above (state in SGPRs, scalar
// loads); the 64 lanes are the 64 coefficients of the block at hand, lane k owning zigzag position k:
//   sequential      (a SOF0 / SOF1 file whose components come in several scans: Ss = 0, Se = 63) a block whole, as below.
//   DC first        the difference is decoded as in the sequential process; lane 0 stores prediction << Al.
//   DC refinement   one raw bit per block; a set bit goes into the coefficient with a fire-and-forget atomic OR.
//   AC first        run / size symbols place values << Al at zigzag positions: `lane == k` keeps each in its lane's
//                   register, non-zero lanes store once per block; end-of-band runs skip whole blocks.
//   AC refinement   the lanes load the block (the NEXT block's load is issued before this block is parsed, so that its
//                   latency hides behind the parse); a ballot gives the map of non-zero coefficients the scalar parse needs
//                   (every non-zero coefficient it passes costs one correction bit, zero ones count down the run); the
//                   parse collects three 64-bit maps -- corrections, new +1 << Al, new -1 << Al -- and the lanes apply them.
// Scans are separated by a device-scope fence: a scan reads what earlier scans wrote through other lanes.
// ---------------------------------------------------------------------------------------------------------------------
// Expands the Huffman tables of progressive scans (BITS / HUFFVAL as in the file -> the 9-bit lookahead table and the
// MAXCODE / VALPTR arrays of T.81 F.2.2.3), one wave per table.
__global__ __launch_bounds__(64) void jpeg_expand_huff_kernel(const DevHuffSrc* __restrict__ src, DevHuffLite* __restrict__ dst) {
  const DevHuffSrc& in = src[blockIdx.x];
  DevHuffLite& out = dst[blockIdx.x];
  const int lane = threadIdx.x;
  int32_t maxcode[17], valoff[17];
  int32_t code = 0;
  int k = 0;
#pragma unroll
  for (int l = 1; l <= 16; l++) {
    const int n = in.bits[l];
    valoff[l] = k - code;
    code += n;
    k += n;
    maxcode[l] = n ? code - 1 : -1;
    code <<= 1;
  }
  if (lane == 0) {
    out.maxcode[0] = -1;
    out.valoff[0] = 0;
#pragma unroll
    for (int l = 1; l <= 16; l++) {
      out.maxcode[l] = maxcode[l];
      out.valoff[l] = valoff[l];
    }
    out.maxcode[17] = 0x7FFFFFFF;
  }
  for (int p = lane; p < (1 << kLookBits); p += 64) {
    uint32_t e = kLongCode;  // a longer code (or none at all: the MAXCODE walk sorts that out)
    bool found = false;
#pragma unroll
    for (int l = 1; l <= kLookBits; l++) {  // F.2.2.3: the shortest length whose MAXCODE holds the prefix
      const int32_t c = p >> (kLookBits - l);
      if (!found && c <= maxcode[l]) {
        e = ((uint32_t)l << 8) | in.vals[(valoff[l] + c) & 255];
        found = true;
      }
    }
    out.look[p] = (uint16_t)e;
  }
  reinterpret_cast<uint32_t*>(out.vals)[lane] = reinterpret_cast<const uint32_t*>(in.vals)[lane];
}

// What a scan of a PIPELINED decode (jpeg_prog_pipe_kernel) must stay behind: block rows completed per scan of the file, in
// LDS (0x7FFFFFFF: finished), and the up to four earlier scans whose coefficients this one refines (-1: none).
struct ProgWait {
  volatile __attribute__((address_space(3))) int* progress;  // (LDS-typed: ds_read / ds_write, not flat instructions)
  int p0, p1, p2, p3;
};
constexpr int kProgSpinCap = 1 << 19;

// One luminance-carrying scan of a progressive (or multi-scan sequential) file into the file's coefficient slot: T.81 Annex G
// as libjpeg's jdphuff.c decodes it.  Wave-uniform (the reader state lives in SGPRs); the 64 lanes are the 64 coefficients
// of the block at hand.  PIPE: the scans of a file run in different waves of one workgroup; a refinement scan waits, block
// row by block row, for the scans it refines (ProgWait) and every scan publishes the rows it has completed.
template <bool PIPE, class Image>
__device__ __forceinline__ void prog_scan(const Image& im, int si, const DevScan* __restrict__ scans,
                                          const DevHuffLite* __restrict__ huffs, const uint8_t* __restrict__ stream,
                                          int16_t* __restrict__ coef, int width, int height, int lane, int nat, const ProgWait W,
                                          bool& broken, bool& suspect) {
  const int h0 = im.h[0], v0 = im.v[0], lum = h0 * v0, mcus_x = im.mcus_x;
  // Every wait is bounded (kProgSpinCap polls of ~130 cycles, ~60 ms: a hundred times a row's decode): a wave that gives up
  // flags the file as suspect -- the one-wave kernel then decodes it again -- and goes on, so the grid always drains.
  auto wait_one = [&](int p, int need) __attribute__((always_inline)) {
    int spins = 0;
    while (W.progress[p] < need) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kProgSpinCap) {
        suspect = true;
        break;
      }
    }
  };
  auto wait_rows = [&](int need) __attribute__((always_inline)) {
    if (PIPE) {
      if (W.p0 >= 0) wait_one(W.p0, need);
      if (W.p1 >= 0) wait_one(W.p1, need);
      if (W.p2 >= 0) wait_one(W.p2, need);
      if (W.p3 >= 0) wait_one(W.p3, need);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  };
  auto publish_rows = [&](int rows) __attribute__((always_inline)) {
    if (PIPE) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (this wave's coefficient stores have landed)
      if (lane == 0) W.progress[si] = rows;
    }
  };
  {
    if (!PIPE) __threadfence();  // (pipelined: the row-wise hand-over below orders the scans of a file)
    const auto& sc = *in_constant(scans + (im.first_scan + si));
    const int Ss = sc.Ss, Se = sc.Se, Ah = sc.Ah, Al = sc.Al, ns = sc.ncomp, restart_interval = sc.restart_interval;
    const uint32_t o = im.stream_off + sc.off;
    BitReader br{reinterpret_cast<const uint32_t*>(stream + (o & ~3u)), o & 3u, (o & 3u) + sc.len, 0ull, 0, false};
    const int p1 = 1 << Al, m1 = -(1 << Al);
    auto below = [](int n) __attribute__((always_inline)) -> uint64_t { return n >= 64 ? ~0ull : (1ull << n) - 1ull; };  // positions 0 .. n - 1
    const uint64_t band = below(Se + 1) & ~below(Ss);
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int eobrun = 0, until_restart = restart_interval;
    // the MCUs of this scan: the frame's interleaved MCUs, or -- a scan of the luminance alone -- its own blocks in raster
    // order over ceil(W / 8) x ceil(H / 8) (T.81 A.2.2; the luminance is sampled at the full rate)
    const bool interleaved = ns > 1;
    const int units_x = interleaved ? mcus_x : (width + 7) >> 3, units_y = interleaved ? im.mcus_y : (height + 7) >> 3;
    auto block_of = [&](int by, int bx) __attribute__((always_inline)) -> int {  // a luminance block's place in the coefficient buffer
      const int my = by / v0, mx = bx / h0;
      return (my * mcus_x + mx) * lum + (by - my * v0) * h0 + (bx - mx * h0);
    };
    if (Ss == 0) {
      // ---- DC scans ----
      const DevHuffLite *t0 = huffs + sc.huff[0], *t1 = huffs + sc.huff[ns > 1 ? 1 : 0], *t2 = huffs + sc.huff[ns > 2 ? 2 : 0];
      const DevHuffLite *a0 = huffs + sc.huff_ac[0], *a1 = huffs + sc.huff_ac[ns > 1 ? 1 : 0], *a2 = huffs + sc.huff_ac[ns > 2 ? 2 : 0];
      // (the tables are ARGUMENTS: picked from captured references by a run-time index they would pin the closure, and with
      // it the whole reader state, in scratch memory -- the bit walk then runs on the vector unit under exec masks)
      auto dc_block = [&](const DevHuffLite* t, const DevHuffLite* ta, int& pred, int16_t* blk) __attribute__((always_inline)) {  // blk: a luminance block, or null
        if (Se == 63) {
          // a scan of a SEQUENTIAL file whose components come in several scans: the whole block, F.2.2.1 + F.2.2.2
          br.fill();
          const int s = br.decode(t) & 15;  // (DC symbols above 15 are refused on the host, as by libjpeg)
          br.fill();
          pred += br.receive_extend(s);
          int mine = lane == 0 ? pred : 0;
          for (int k = 1; k < 64;) {
            br.fill();
            const int rs = br.decode(ta);
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
              if (r != 15) break;
              k += 16;
              continue;
            }
            k += r;
            br.fill();
            const int v = br.receive_extend(sz);
            if (lane == min(k, 63)) mine = v;  // (past the block's end: on the last coefficient, as libjpeg's padded order table has it)
            k++;
          }
          if (blk && mine != 0) blk[nat] = (int16_t)mine;
        } else if (Ah == 0) {
          br.fill();
          const int s = br.decode(t) & 15;  // (DC symbols above 15 are refused on the host, as by libjpeg)
          br.fill();
          pred += br.receive_extend(s);
          if (blk && lane == 0) blk[0] = (int16_t)(pred * p1);
        } else if (br.get_bit()) {
          if (blk && lane == 0) atomicOr(reinterpret_cast<unsigned int*>(blk), (unsigned int)p1);  // coefficient 0: low half
        }
      };
      if (Ah != 0 && !interleaved && restart_interval == 0) {
        // one raw bit per block and nothing else in the stream: 16 blocks per read, a lane each
        const int nblocks = units_x * units_y;
        for (int b0 = 0; b0 < nblocks; b0 += 16) {
          const int cnt = nblocks - b0 < 16 ? nblocks - b0 : 16;
          wait_rows((b0 + cnt - 1) / units_x + 1);
          const uint32_t got = (uint32_t)br.receive(cnt);
          if (lane < cnt && ((got >> (cnt - 1 - lane)) & 1u)) {
            const int b = b0 + lane, by = b / units_x;
            atomicOr(reinterpret_cast<unsigned int*>(coef + (size_t)block_of(by, b - by * units_x) * 64), (unsigned int)p1);
          }
          publish_rows((b0 + cnt) / units_x);
        }
      } else {
        for (int uy = 0; uy < units_y && !broken; uy++) {
          wait_rows(interleaved ? (uy + 1) * v0 : uy + 1);
          for (int ux = 0; ux < units_x; ux++) {
            if (restart_interval && until_restart == 0) {
              if (!br.restart()) {
                broken = true;
                break;
              }
              pred0 = pred1 = pred2 = 0;
              until_restart = restart_interval;
            }
            if (!interleaved) {
              dc_block(t0, a0, pred0, coef + (size_t)block_of(uy, ux) * 64);
            } else {
              for (int c = 0; c < ns; c++) {
                const int ci = sc.comp[c], hh = im.h[ci], vv = im.v[ci];
                const DevHuffLite* t = c == 0 ? t0 : (c == 1 ? t1 : t2);
                const DevHuffLite* ta = c == 0 ? a0 : (c == 1 ? a1 : a2);
                int pred = c == 0 ? pred0 : (c == 1 ? pred1 : pred2);  // (by value: a reference picked at run time would
                for (int b = 0; b < hh * vv; b++)                      //  put the three predictions into scratch memory)
                  dc_block(t, ta, pred, ci == 0 ? coef + ((size_t)(uy * mcus_x + ux) * lum + b) * 64 : nullptr);
                pred0 = c == 0 ? pred : pred0;
                pred1 = c == 1 ? pred : pred1;
                pred2 = c == 2 ? pred : pred2;
              }
            }
            if (restart_interval) until_restart--;
          }
          publish_rows(interleaved ? (uy + 1) * v0 : uy + 1);
        }
      }
    } else {
      // ---- AC scans: the luminance alone, block after block in raster order ----
      const DevHuffLite* t = huffs + sc.huff[0];
      const int nblocks = units_x * units_y;
      // (a file's time is its longest chain -- its refinement scans, each a step behind the scans it refines: they go first
      // on their SIMD.  512 files: 45.1 -> 40.9 ms; priority for every AC scan: 44.1, for the wide refinement scans alone: 41.7)
      if (PIPE && Ah != 0) __builtin_amdgcn_s_setprio(3);
      int bx = 0, by = 0;
      int16_t* blk = coef + (size_t)block_of(0, 0) * 64;
      wait_rows(1);
      int16_t next = Ah ? blk[nat] : (int16_t)0;
      for (int b = 0; b < nblocks; b++) {
        if (restart_interval && until_restart == 0) {
          if (!br.restart()) {
            broken = true;
            break;
          }
          eobrun = 0;
          until_restart = restart_interval;
        }
        int16_t c = next;
        int16_t* const here = blk;
        if (++bx == units_x) {
          bx = 0;
          by++;
          if (b + 1 < nblocks) wait_rows(by + 1);  // (the next row's first block is fetched below)
        }
        if (b + 1 < nblocks) {
          blk = coef + (size_t)block_of(by, bx) * 64;
          if (Ah) next = blk[nat];
        }
        if (Ah == 0) {
          // AC first
          if (eobrun > 0) {
            eobrun--;
          } else {
            int mine = 0;
            for (int k = Ss; k <= Se; k++) {
              br.fill();
              const int rs = br.decode(t);
              const int r = rs >> 4, sz = rs & 15;
              if (sz) {
                k += r;
                br.fill();
                const int v = br.receive_extend(sz);
                if (k > Se) suspect = true;  // (damaged data: a value behind the band -- libjpeg writes it too, past the block's end on coefficient 63)
                if (lane == min(k, 63)) mine = v * p1;
              } else if (r == 15) {
                k += 15;
              } else {
                eobrun = 1 << r;
                if (r) eobrun += br.receive(r);
                eobrun--;
                break;
              }
            }
            if (mine != 0) here[nat] = (int16_t)mine;
          }
        } else {
          // AC refinement.  `nz`: the coefficients of the band that are non-zero so far.  A symbol (run r, new value of
          // magnitude 1 << Al) moves the decoder to the (r + 1)-th ZERO coefficient from k on; every non-zero one it
          // passes on the way owes a correction bit.  Both are answered from the map: the target by clearing r low
          // bits of the zero map, the passed ones 16 positions at a time -- their bits are read together and each lane
          // picks its own by its rank among them.
          const uint64_t nz = __ballot(c != 0) & band;
          int my_corr = 0, my_new = 0;
          auto corrections = [&](uint64_t P) __attribute__((always_inline)) {
            while (P) {
              const int lo = __builtin_ctzll(P);
              const uint64_t Wn = P & (0xFFFFull << lo);
              const int cnt = __builtin_popcountll(Wn);
              const uint32_t got = (uint32_t)br.receive(cnt);
              if ((Wn >> lane) & 1ull) my_corr = (int)(got >> (cnt - 1 - __builtin_popcountll(Wn & below(lane)))) & 1;
              P &= ~Wn;
            }
          };
          int k = Ss;
          if (eobrun == 0) {
            while (k <= Se) {
              br.fill();
              const int rs = br.decode(t);
              const int r = rs >> 4, sz = rs & 15;
              int sign = 0;
              if (sz) {
                sign = br.get_bit() ? 1 : -1;  // (the size must be 1; libjpeg warns and carries on the same way)
              } else if (r != 15) {
                eobrun = 1 << r;
                if (r) eobrun += br.receive(r);
                break;
              }
              const uint64_t ahead = band & ~below(k);
              uint64_t zeros = ~nz & ahead;
              for (int i = 0; i < r && zeros; i++) zeros &= zeros - 1ull;
              const int pos = zeros ? __builtin_ctzll(zeros) : Se + 1;  // (no such zero: the run ends behind the band)
              corrections(nz & ahead & below(pos));
              if (sign && pos > Se) suspect = true;  // (damaged data: the run ends behind the band)
              if (sign && lane == min(pos, 63)) my_new = sign;  // (libjpeg writes it there; behind a band that ends at 63: on 63)
              k = pos + 1;
            }
          }
          if (eobrun > 0) {
            if (k <= Se) corrections(nz & ~below(k));
            eobrun--;
          }
          int cv = c;
          bool changed = false;
          if (my_corr && (cv & p1) == 0) {
            cv = cv >= 0 ? cv + p1 : cv + m1;
            changed = true;
          }
          if (my_new) {
            cv = my_new > 0 ? p1 : m1;
            changed = true;
          }
          if (changed) here[nat] = (int16_t)cv;
        }
        if (restart_interval) until_restart--;
        if (bx == 0) publish_rows(by);  // (the block just finished was its row's last)
      }
    }
    publish_rows(0x7FFFFFFF);
    if (PIPE) __builtin_amdgcn_s_setprio(0);
  }
}

__global__ __launch_bounds__(64) void jpeg_prog_kernel(const DevImage* __restrict__ images,
                                                        const uint32_t* __restrict__ index,
                                                        const DevScan* __restrict__ scans,
                                                        const DevHuffLite* __restrict__ huffs,
                                                        const uint8_t* __restrict__ stream, int16_t* __restrict__ coef_all,
                                                        size_t coef_stride, int slot0, int width, int height,
                                                        int32_t* __restrict__ status,
                                                        const int32_t* __restrict__ only_flagged) {
  if (only_flagged && in_constant(only_flagged)[blockIdx.x] == 0) return;  // (the second pass over a batch: damaged files only)
  const int lane = threadIdx.x;
  const uint32_t image = in_constant(index)[blockIdx.x];
  const auto& im = *in_constant(images + image);
  int16_t* coef = coef_all + (size_t)(slot0 + blockIdx.x) * (coef_stride / sizeof(int16_t));
  const int h0 = im.h[0], v0 = im.v[0], lum = h0 * v0, mcus_x = im.mcus_x;
  const int nlb = mcus_x * im.mcus_y * lum;
  {  // nothing is known yet
    uint32_t* c32 = reinterpret_cast<uint32_t*>(coef);
    for (int i = lane; i < nlb * 32; i += 64) c32[i] = 0u;
  }
  const int nat = c_zigzag[lane];  // where this lane's coefficient sits inside a block
  bool broken = false;
  bool suspect = false;
  const ProgWait none{nullptr, -1, -1, -1, -1};
  for (int si = 0; si < im.n_scans && !broken; si++)
    prog_scan<false>(im, si, scans, huffs, stream, coef, width, height, lane, nat, none, broken, suspect);
  if (lane == 0 && broken) atomicOr(status, 2);
}

// The scans of a progressive file, PIPELINED (round 5).  A scan is a serial bit stream, but the scans of a file depend on
// each other only through the coefficients: a first scan (Ah = 0) on nothing, a refinement scan on the scans that brought
// its band to the previous precision -- and on those only for the blocks it is about to touch.  So one WORKGROUP per file
// runs the file's scans in kProgWaves waves at once (wave w takes scans w, w + kProgWaves, ...): every scan publishes, in
// LDS, the block rows it has completed, and a refinement scan stays one row behind the (at most four) latest earlier
// scans that cover its band.  The waves of a workgroup are resident together, a scan's prerequisites have lower indices and
// a wave takes its scans in ascending order, so the scan with the lowest index never waits and the pipeline always
// drains -- no assumption about dispatch order, no spinning on another workgroup.  Coefficient hand-over stays inside
// the CU (one L1: a workgroup-scope release / acquire around the LDS counter).  A file's time is its longest chain instead
// of the sum of its scans.
// Damaged data keeps the one-wave decoder's results bit for bit: a scan that meets a missing restart marker, or places a
// value behind its band (libjpeg does the same), flags the file, and the one-wave kernel -- launched behind this one over
// the flagged files only -- decodes it again scan after scan.
#ifndef VSF_PROG_WAVES
#define VSF_PROG_WAVES 6
#endif
constexpr int kProgWaves = VSF_PROG_WAVES;
constexpr int kProgPipeMaxFiles = 896;  // progressive files per call up to which the pipelined form is used (see the launcher)
constexpr int kProgMaxScans = 1024;  // (the host's parser refuses more: vsf_jpeg_host.cc)

__global__ __launch_bounds__(64 * kProgWaves) void jpeg_prog_pipe_kernel(const DevImage* __restrict__ images,
                                                                         const uint32_t* __restrict__ index,
                                                                         const DevScan* __restrict__ scans,
                                                                         const DevHuffLite* __restrict__ huffs,
                                                                         const uint8_t* __restrict__ stream,
                                                                         int16_t* __restrict__ coef_all, size_t coef_stride,
                                                                         int slot0, int width, int height,
                                                                         int32_t* __restrict__ flags) {
  __shared__ int progress[kProgMaxScans];
  __shared__ int redo;  // bit 0: a scan broke off at a missing restart marker, bit 1: a value was placed behind a band
  const int tid = threadIdx.x, lane = tid & 63;
  // (wave-uniform by construction, but only readfirstlane tells the compiler: without it the scan index, and with it the
  // whole reader state, lives in vector registers and the bit walk leaves the scalar unit)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t image = in_constant(index)[blockIdx.x];
  const auto& im = *in_constant(images + image);
  int16_t* coef = coef_all + (size_t)(slot0 + blockIdx.x) * (coef_stride / sizeof(int16_t));
  const int nlb = im.mcus_x * im.mcus_y * im.h[0] * im.v[0];
  const int n_scans = min(im.n_scans, kProgMaxScans);
  {  // nothing is known yet
    uint32_t* c32 = reinterpret_cast<uint32_t*>(coef);
    for (int i = tid; i < nlb * 32; i += 64 * kProgWaves) c32[i] = 0u;
    for (int i = tid; i < n_scans; i += 64 * kProgWaves) progress[i] = 0;
    if (tid == 0) redo = 0;
  }
  __syncthreads();
  const int nat = c_zigzag[lane];
  auto below = [](int n) __attribute__((always_inline)) -> uint64_t { return n >= 64 ? ~0ull : (1ull << n) - 1ull; };
  for (int si = wave; si < n_scans; si += kProgWaves) {
    // the scans this one refines: for every coefficient of its band the LATEST earlier scan that covers it (that scan has
    // itself waited for the ones before it)
    const auto& sc = *in_constant(scans + (im.first_scan + si));
    ProgWait W{(volatile __attribute__((address_space(3))) int*)progress, -1, -1, -1, -1};
    if (sc.Ah != 0) {
      uint64_t open = below(sc.Se + 1) & ~below(sc.Ss);
      int found = 0;
      for (int p = si - 1; p >= 0 && open; p--) {
        const auto& e = *in_constant(scans + (im.first_scan + p));
        const uint64_t b = below(e.Se + 1) & ~below(e.Ss);
        if (b & open) {
          open &= ~b;
          if (found == 0) W.p0 = p;
          if (found == 1) W.p1 = p;
          if (found == 2) W.p2 = p;
          if (found == 3) W.p3 = p;
          if (found >= 4) {  // (a crafted scan script: more than four pieces; wait for the extra one to finish altogether)
            int spins = 0;
            while (W.progress[p] != 0x7FFFFFFF && ++spins <= kProgSpinCap) __builtin_amdgcn_s_sleep(8);
            if (spins > kProgSpinCap && lane == 0) atomicOr(&redo, 2);
          }
          found++;
        }
      }
    }
    bool broken = false, suspect = false;
    prog_scan<true>(im, si, scans, huffs, stream, coef, width, height, lane, nat, W, broken, suspect);
    if (lane == 0) {
      progress[si] = 0x7FFFFFFF;  // (also after a break: nobody may wait for this scan any longer)
      if (broken || suspect) atomicOr(&redo, (broken ? 1 : 0) | (suspect ? 2 : 0));
    }
  }
  __syncthreads();
  if (tid == 0) flags[blockIdx.x] = redo;
}

// =====================================================================================================================
// Parallel entropy decoding.  Files without restart intervals (what a camera driver writes) first; files WITH them need
// no guessing and use steps 1 and 3 only (the end of this comment).
//
// A Huffman-coded stream has no markers to split it at, but it is SELF-SYNCHRONISING: a decoder started at a wrong bit
// soon falls into step with the true one (Klein & Wiseman; for JPEG on GPUs: Weissenberger & Schmidt 2018/2021).  So a
// 256-thread workgroup per image
//   1. removes the byte stuffing in parallel (FF00 -> FF, stop at the first marker) -- 16 coalesced bytes per thread and
//      round, the 0xFF / 0x00 tests on whole dwords -- into a "clean" stream in which a position is ONE integer (a bit
//      index), cuts it into 256 segments and stores it once more SEGMENT-MAJOR (dword j of segment s at [j][s], as
//      big-endian dwords), so that the threads of a wave, each walking its own segment, fetch neighbouring addresses;
//   2. thread t decodes segment t from an ASSUMED state (bit t * B, first block of an MCU, DC next) up to the first
//      symbol boundary past its end and records that end state (bit, block in MCU, zigzag index).  Thread 0's assumption
//      is true.  Then, round after round, every thread restarts from its predecessor's end state whenever that state
//      changed, until no end state changes: then start[t] == end[t-1] for all t, and by induction from thread 0 every
//      start is the TRUE decoder's state.  A thread whose decode has merged with the true one never moves again, so the
//      work is one segment per thread and round, and gray streams settle in two or three rounds;
//      A re-run that arrives at a state its thread's earlier run went through (remembered at a few block ends) stops
//      there: the rest is known;
//   3. prefix-sums the blocks each segment completes and the luminance DC differences it read, and decodes once more,
//      now gathering each luminance block's coefficients (the DC as a value) in LDS; finished blocks leave for the coefficient buffer as whole 128-byte
//      lines, copied by the wave together (a block belongs to the thread it starts with).
// A second kernel does dequantisation + IDCT for all blocks of all images at once.
// Files with restart intervals: step 1 also drops the RSTn markers and lists the clean offset behind each (where the
// next interval begins: byte-aligned, first block of an MCU, DC prediction zero); thread t then runs step 3 over
// intervals t, t + 256, ... of the clean stream as it lies.  11 k -> 138 k images/s against the one-wave decoder.  Latency per image: a few segment decodes instead of the whole stream.
// Round-2 history of this kernel for 512 files of 114 KB (profiles/r02/README.md): 4.03 ms with byte-wise stuffing
// removal, table references the compiler parked in scratch memory, and one 2-byte store per coefficient; 1.2 ms as
// described here (stuffing removal 1.27 -> 0.07 ms, each decode pass 0.55 -> 0.27 ms, the writing pass 1.33 -> 0.53 ms).
// =====================================================================================================================
constexpr int kTileBytes = kParThreads * 16;  // bytes of the raw stream one round of the stuffing removal covers
// (kParThreads, kOverlap, kTransSlack: vsf_jpeg_host.h -- the host's plan sizes its scratch with them)

// The decode loop reads its tables out of LDS through explicitly LDS-typed pointers and offsets: references to one of
// several tables picked per lane made the compiler keep a pointer array in scratch memory and fetch the entries with
// generic loads -- two trips to HBM-backed memory per symbol, 1.8 k cycles per loop round.
using lds_u16 = const __attribute__((address_space(3))) uint16_t*;
using lds_u8 = const __attribute__((address_space(3))) uint8_t*;
constexpr uint32_t kHuff16 = sizeof(DevHuff) / 2;            // one table, in 16-bit entries
constexpr uint32_t kSub16 = offsetof(DevHuff, sub) / 2;      // its second-level tables

struct ParGeom {  // wave-uniform: blocks per MCU and the table slots of the components (8 bits each, component 0 lowest;
  int lum, n1, m;  // packed so that the pick is arithmetic: a choice between six variables became a load from scratch)
  uint32_t dc_slots, ac_slots;
  const uint32_t* trans;  // the clean stream, SEGMENT-MAJOR: dword j of segment s (big-endian: stream bit q is bit
  uint32_t segdw, magic;  // 31 - (q & 31) of dword q >> 5) sits at trans[j * kParThreads + s]; rows segdw .. segdw +
                          // kOverlap - 1 of a column repeat the next segment's first rows; magic = floor(2^32 / segdw) + 1
  __device__ __forceinline__ uint32_t fetch(uint32_t b) const {  // dword b of the stream, wherever the caller stands
    const uint32_t s = __umulhi(b, magic);  // b / segdw (exact while b * segdw < 2^32)
    return s < (uint32_t)kParThreads ? trans[(b - s * segdw) * (uint32_t)kParThreads + s] : 0u;  // zero bits past the end
  }
};

// A lane's place in its segment's column: row jb and the two dwords (w0, w1) the current symbol is cut from.  Stepping to
// the next row is an add (the column's kOverlap extra rows cover every overshoot of the counting passes); only the end
// of a long block in the writing pass leaves the column and asks ParGeom::fetch.
struct ParWin {
  const uint32_t* col;  // trans + the lane's column
  uint32_t rows;        // rows a column holds (segdw + kOverlap)
  uint32_t gbase;       // dword index of the lane's row 0 in the whole stream
  uint32_t jb, w0, w1;
  const uint32_t* lin;  // non-null (files with restart intervals: no segments): the clean stream itself, `rows` dwords
  __device__ __forceinline__ uint32_t row(const ParGeom& G, uint32_t r) const {
    if (lin) return r < rows ? __builtin_bswap32(lin[r]) : 0u;  // (workgroup-uniform)
    return r < rows ? col[(size_t)r * kParThreads] : G.fetch(gbase + r);
  }
  __device__ __forceinline__ void open(const ParGeom& G, uint32_t q) {
    jb = (q >> 5) - gbase;  // (0 for a start state; "negative" for the last thread of a stream that ends before its segment)
    w0 = row(G, jb);
    w1 = row(G, jb + 1u);
  }
};

// One symbol (a DC size + its bits, or an AC run/size + its bits) of the block at (c, k) -- block inside the MCU, zigzag
// index (0: a DC size comes next) -- read at bit position q.  Straight-line code but for the second table lookup: the
// lanes of a wave sit at different places of different blocks, so every branch in here would be taken by somebody in
// every round.  Returns true when the symbol ends its block; kk / val: the coefficient it carries (kk < 0: none).
// kClip: bits from `clip` on read as zero (an interval of a file with restart intervals: behind its end libjpeg feeds zero
// bits, while the clean stream goes on with the next interval's data).
template <bool kClip = false>
__device__ __forceinline__ bool par_symbol(const ParGeom& G, ParWin& W, lds_u16 tab, uint32_t& q, int c, int& k, int& kk,
                                           int& val, uint32_t clip = 0xFFFFFFFFu) {
  // the 32 bits from q on (a code + its extra bits need <= 27, so q moves on by at most one dword per symbol)
  if ((q >> 5) - W.gbase != W.jb) {
    W.jb++;
    W.w0 = W.w1;
    W.w1 = W.row(G, W.jb + 1u);
  }
  uint32_t x = (uint32_t)(((((uint64_t)W.w0) << 32) | W.w1) >> (32u - (q & 31u)));
  if (kClip && q + 32u > clip) x = q >= clip ? 0u : x & (0xFFFFFFFFu << (32u - (clip - q)));
  const bool is0 = c < G.lum, is1 = c < G.lum + G.n1;
  const bool isdc = k == 0;
  const uint32_t slot = (((isdc ? G.dc_slots : G.ac_slots) >> (is0 ? 0 : (is1 ? 8 : 16))) & 255u) * kHuff16;
  uint32_t e = tab[slot + (x >> (32 - kLookBits))];
  if (e & kLongCode) {
    const uint32_t ti = e & 255u;
    e = ti < (uint32_t)kMaxSub ? tab[slot + kSub16 + (ti << kSubBits) + ((x >> (32 - 16)) & ((1u << kSubBits) - 1u))]
                               : (uint32_t)kNoCode;
  }
  const int len = (int)(e >> 8), sym = (int)(e & 255u);
  const int s = sym & 15, r = isdc ? 0 : sym >> 4;
  // RECEIVE + EXTEND (T.81 F.2.2.1): s bits after the code; v < 2^(s-1) stands for v - 2^s + 1
  const int v = (int)__builtin_amdgcn_ubfe(x, (uint32_t)(32 - len - s), (uint32_t)s);
  const int ones = (1 << s) - 1;
  val = 2 * v > ones ? v : v - ones;
  q += (uint32_t)(len + s);
  const bool stop = !isdc && s == 0;  // EOB (r != 15) or ZRL (r == 15): no coefficient
  const bool zrl = stop && r == 15;
  const int at = k + r;
  kk = stop ? -1 : min(at, 63);  // (past the block's end -- damaged data -- the value lands on coefficient 63, as libjpeg's padded order table has it)
  k = zrl ? k + 16 : at + 1;
  return (stop && !zrl) || k > 63;
}

// What a counting pass remembers of its way: the decoder's state at the ends of its 2nd, 4th, 8th and 16th block, and how
// many blocks / how much luminance DC were still to come from there.  A later pass over the same segment from another
// start state that arrives at one of these states has MERGED with the remembered pass -- the rest of its way is known.
constexpr int kMarks = 4;
static_assert(kMarks == 4, "par_count tests the four marks by name");
struct ParMarks {
  uint32_t q[kMarks];    // bit position (0xFFFFFFFF: none)
  uint32_t cr[kMarks];   // block inside the MCU | blocks from here to the end of the segment << 8
  int dc[kMarks];        // luminance DC differences from here to the end of the segment, summed
};

// Decodes from (q, c, k) to the first symbol boundary at or past `limit`, or until it meets one of `marks`' states (then
// `merged`: the end state is the remembered pass's, which the caller still holds).  blocks / dcsum: blocks completed and sum of the luminance DC differences read in [start, limit).  A pass
// that runs to the end leaves its own marks.
__device__ __forceinline__ void par_count(const ParGeom& G, ParWin& W, lds_u16 tab, uint32_t limit, uint32_t& q_io, int& c_io,
                                          int& k_io, ParMarks& marks, uint32_t& blocks, int& dcsum, bool& merged) {
  uint32_t q = q_io, done = 0;
  int c = c_io, k = k_io, dc = 0;
  merged = false;
  W.open(G, q);
  uint32_t tq[kMarks], tcn[kMarks];
  int td[kMarks];
#pragma unroll
  for (int i = 0; i < kMarks; i++) {
    tq[i] = 0xFFFFFFFFu;
    tcn[i] = 0u;
    td[i] = 0;
  }
  while (q < limit && !merged) {
    {
      int kk, val;
      const bool lum = c < G.lum;
      const bool end = par_symbol(G, W, tab, q, c, k, kk, val);
      if (kk == 0 && lum) dc += val;
      if (end) {
        k = 0;
        done++;
        c = c + 1 == G.m ? 0 : c + 1;
        if (q == marks.q[0] || q == marks.q[1] || q == marks.q[2] || q == marks.q[3]) {  // (rare: kept out of the way)
#pragma unroll
          for (int i = 0; i < kMarks; i++)
            if (q == marks.q[i] && (uint32_t)c == (marks.cr[i] & 255u)) {
              merged = true;
              done += marks.cr[i] >> 8;
              dc += marks.dc[i];
            }
        } else if (done <= (2u << (kMarks - 1)) && (done & (done - 1u)) == 0u) {
#pragma unroll
          for (int i = 0; i < kMarks; i++)
            if (done == (2u << i)) {
              tq[i] = q;
              tcn[i] = (uint32_t)c | (done << 8);
              td[i] = dc;
            }
        }
      }
    }
  }
  if (!merged) {
#pragma unroll
    for (int i = 0; i < kMarks; i++) {
      marks.q[i] = tq[i];
      marks.cr[i] = (tcn[i] & 255u) | ((done - (tcn[i] >> 8)) << 8);
      marks.dc[i] = dc - td[i];
    }
    q_io = q;
    c_io = c;
    k_io = k;
  }
  blocks = done;
  dcsum = dc;
}

// The writing pass over a segment whose true start state is (q, c, k), g = blocks completed before it in the whole image.
// A block belongs to the thread it STARTS with: the rest of a block under way at the segment's start is decoded and
// skipped, the last block is decoded to its end beyond `limit`.  Luminance coefficients (the DC as a value: `pred` is the
// sum of the DC differences before the segment) are gathered in the thread's 64-entry block in LDS (`blk`: kBlkStride dwords apart, zero at entry); whenever lanes finish
// luminance blocks the whole wave copies them out, one 128-byte line per block, to coef[(g / m) * lum + c] and clears
// them.  ALL lanes of a wave call this (the loop is wave-uniform; lanes that are done idle along).
// `finish`: the lane also takes what lies behind the end of the data (`data_end`, in bits) as libjpeg takes it (jdhuff.c
// decode_mcu, insufficient_data): the MCU in which the data run out is decoded to its end on zero bits, every MCU behind it
// is left zero -- uniform gray -- and never loops for ever: every symbol consumes at least one bit of at most 2^32.
constexpr int kBlkStride = 33;  // dwords between the LDS blocks of neighbouring threads (32 + 1: the banks spread)
template <bool kClip = false>
__device__ __forceinline__ uint32_t par_write(const ParGeom& G, ParWin& W, lds_u16 tab, lds_u8 zz, uint32_t* blk_wave,
                                              int lane, uint32_t limit, uint32_t q, int c, int k, uint32_t g, int pred,
                                              uint32_t* __restrict__ coef32, uint32_t total_blocks, bool finish,
                                              uint32_t data_end) {
  uint32_t done = 0;
  uint32_t mb = ((g - (uint32_t)c) / (uint32_t)G.m) * (uint32_t)G.lum;  // luminance blocks of the MCUs before
  bool skip = k != 0;                                                    // inside a block somebody else started
  int16_t* mine = reinterpret_cast<int16_t*>(blk_wave + lane * kBlkStride);
  auto has_work = [&]() {
    const bool more = g + done < total_blocks;
    return q < limit || (k != 0 && more) || (finish && more && (c != 0 || q <= data_end) && q < 0xFFFF0000u);
  };
  bool busy = has_work();
  if (__ballot(busy) != 0ull) W.open(G, q);
  while (__ballot(busy) != 0ull) {
    bool flush = false;
    uint32_t dst = 0;
    if (busy) {
      int kk, val;
      const bool more = g + done < total_blocks;
      const bool lum = c < G.lum;
      const bool end = par_symbol<kClip>(G, W, tab, q, c, k, kk, val, data_end);
      if (lum && !skip && more && kk >= 0) {
        if (kk == 0) val = pred += val;  // (a block's DC symbol is its first: never inside a skipped rest)
        mine[zz[kk]] = (int16_t)val;
      }
      if (end) {
        flush = lum && !skip && more;
        dst = mb + (uint32_t)c;
        skip = false;
        k = 0;
        done++;
        if (++c == G.m) {
          c = 0;
          mb += (uint32_t)G.lum;
        }
      }
      busy = has_work();
    }
    uint64_t todo = __ballot(flush);
    while (todo) {  // wave-uniform: one finished block per round, 32 lanes x 4 bytes
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)dst, src);
      if (lane < 32) {
        uint32_t* p = blk_wave + src * kBlkStride + lane;
        coef32[(size_t)d * 32 + lane] = *p;
        *p = 0u;
      }
    }
  }
  {  // the MCUs a finishing lane has left out: zero coefficients (the buffer holds the last call's)
    const uint32_t first = (g + done) / (uint32_t)G.m * (uint32_t)G.lum, last = total_blocks / (uint32_t)G.m * (uint32_t)G.lum;
    uint64_t todo = __ballot(finish && first < last);
    while (todo) {
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)first, src) * 32u, hi = (uint32_t)__builtin_amdgcn_readlane((int)last, src) * 32u;
      for (uint32_t i = lo + (uint32_t)lane; i < hi; i += 64u) coef32[i] = 0u;
    }
  }
  return q;  // (where the lane stopped)
}

__global__ __launch_bounds__(kParThreads) void jpeg_par_decode_kernel(const DevImage* __restrict__ images,
                                                                       const uint32_t* __restrict__ index,
                                                                       const DevTables* __restrict__ tables,
                                                                       const uint8_t* __restrict__ stream,
                                                                       uint32_t* __restrict__ clean_all,
                                                                       uint32_t* __restrict__ trans_all,
                                                                       int16_t* __restrict__ coef_all, size_t coef_stride,
                                                                       int max_slots, int32_t* __restrict__ status) {
  constexpr int kWaves = kParThreads / 64;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_tab[];  // max_slots Huffman tables (DevTables::huff): what
                                                                     // the batch's files use -- two for gray streams
  __shared__ uint32_t s_q[kParThreads + 1], s_ck[kParThreads + 1];  // end states; [t] = start of segment t (entry 0: truth)
  __shared__ uint32_t s_cnt[kParThreads];
  __shared__ int s_dc[kParThreads];
  __shared__ uint32_t s_scan[2][kWaves], s_rscan[2][kWaves];
  __shared__ uint32_t s_end, s_changed;
  __shared__ uint8_t s_zz[64];
  __shared__ uint32_t s_blk[kParThreads * kBlkStride];  // one coefficient block per thread (the writing pass)
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  for (int i = t; i < kParThreads * kBlkStride; i += kParThreads) s_blk[i] = 0u;
  const DevImage& im = images[index[blockIdx.x]];  // (the coefficient buffer is indexed by the launch's own numbering)
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tables + im.tables);
    for (int i = t; i < max_slots * (int)(sizeof(DevHuff) / 4); i += kParThreads) s_tab[i] = src[i];
  }
  if (t < 64) s_zz[t] = c_zigzag[t];
  if (t == 0) s_end = 0xFFFFFFFFu;
  const uint8_t* raw = stream + im.stream_off;
  const uint32_t* raw32 = reinterpret_cast<const uint32_t*>(raw);  // (stream_off is a multiple of 4)
  uint8_t* clean8 = reinterpret_cast<uint8_t*>(clean_all) + im.stream_off;
  const uint32_t* clean = clean_all + (im.stream_off >> 2);
  const uint32_t len = im.stream_len;
  __syncthreads();
  // ---- 1. remove the byte stuffing: raw -> clean, a tile of 16 bytes per thread at a time (coalesced) ----
  // Per dword: flags (bit 7 of each byte) for "is 0xFF" and "is 0x00"; a zero after an 0xFF is dropped, an 0xFF before a
  // non-zero byte is a marker and ends the entropy-coded data (EOI, normally).  The clean offset of the first marker --
  // or of the end of the segment -- is the length of the clean stream.
  // Files with restart intervals (ri > 0): an RSTn marker (0xFF, 0xD0..0xD7) is dropped as well and the clean offset
  // behind it -- where the next interval begins, byte-aligned -- goes into the interval table (ivl, in the space the
  // segment-major copy would take).
  const int ri = im.restart_interval;
  uint32_t* trans = trans_all + ((im.stream_off + (uint32_t)blockIdx.x * (uint32_t)kTransSlack) >> 2);
  uint32_t* ivl = trans;
  const uint32_t ivl_cap = (len + (uint32_t)kTransSlack) >> 2;  // entries of the table (the host sends only files that fit)
  uint32_t kept_before = 0, rst_before = 0;  // clean bytes / RSTn markers of earlier tiles (uniform)
  for (uint32_t tile = 0, round = 0; tile < len; tile += kTileBytes, round++) {
    const uint32_t i = tile + (uint32_t)t * 16u;
    uint32_t d[4] = {0u, 0u, 0u, 0u}, ff[4], zz[4], rs[4] = {0u, 0u, 0u, 0u};
    uint32_t prev_ff = 0u, next_zero = 0x80u, next_rst = 0u;
    const int valid = i < len ? (int)min(16u, len - i) : 0;
    if (valid) {  // (the host pads every segment with 32 zero bytes: these loads stay inside)
#pragma unroll
      for (int j = 0; j < 4; j++) d[j] = raw32[(i >> 2) + j];
      prev_ff = i && raw[i - 1] == 0xFFu ? 0x80000000u : 0u;
      const uint32_t nx = raw[i + 16];
      next_zero = nx == 0u ? 0x80u : 0u;
      next_rst = ri && (nx & 0xF8u) == 0xD0u && i + 16u < len ? 0x80u : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t n = ~d[j];
      ff[j] = ~((((n & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | n) | 0x7F7F7F7Fu);
      zz[j] = ~((((d[j] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d[j]) | 0x7F7F7F7Fu);
    }
    if (ri) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t y = (d[j] ^ 0xD0D0D0D0u) & 0xF8F8F8F8u;  // zero bytes: 0xD0..0xD7
        rs[j] = ~((((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) | 0x7F7F7F7Fu);
      }
    }
    uint32_t drop[4], mark[4], rstb[4];
    int kc = 0, rc = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t live = valid >= 4 * j + 4 ? 0x80808080u : (valid > 4 * j ? (0x80808080u >> (8 * (4 * j + 4 - valid))) : 0u);
      const uint32_t after_ff = __builtin_amdgcn_alignbit(ff[j], j ? ff[j - 1] : prev_ff, 24);  // the byte before is 0xFF
      const uint32_t rst_first = ff[j] & __builtin_amdgcn_alignbit(j < 3 ? rs[j + 1] : next_rst, rs[j], 8);  // 0xFF of an RSTn
      rstb[j] = rs[j] & after_ff & live;                                                                    // its second byte
      drop[j] = ((zz[j] & after_ff) | rst_first | rstb[j]) & live;
      mark[j] = ff[j] & ~__builtin_amdgcn_alignbit(j < 3 ? zz[j + 1] : next_zero, zz[j], 8) & ~rst_first & live;
      kc += __popc(live & ~drop[j]);
      rc += __popc(rstb[j]);
    }
    uint32_t inc = (uint32_t)kc, incr = (uint32_t)rc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(inc, o, 64);
      const uint32_t vr = __shfl_up(incr, o, 64);
      if (lane >= o) {
        inc += v;
        incr += vr;
      }
    }
    if (lane == 63) {
      s_scan[round & 1u][wid] = inc;
      s_rscan[round & 1u][wid] = incr;
    }
    __syncthreads();
    uint32_t off = kept_before + inc - (uint32_t)kc, tile_total = 0;
    uint32_t ridx = rst_before + incr - (uint32_t)rc, tile_rst = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) {
      const uint32_t v = s_scan[round & 1u][w], vr = s_rscan[round & 1u][w];
      if (w < wid) {
        off += v;
        ridx += vr;
      }
      tile_total += v;
      tile_rst += vr;
    }
    kept_before += tile_total;
    rst_before += tile_rst;
    if (rc) {  // (rare) interval ridx + 1 begins at the clean offset behind this marker
      uint32_t o = off;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t bit = 1u << (8 * (j & 3) + 7);
        if (rstb[j >> 2] & bit) {
          ++ridx;
          if (ridx < ivl_cap) ivl[ridx] = o;
        }
        if (j < valid && !(drop[j >> 2] & bit)) o++;
      }
    }
    if (valid) {
      if ((mark[0] | mark[1] | mark[2] | mark[3]) != 0u) {  // the clean offset of the first marker in here
        uint32_t o = off;
        bool found = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (!found && mark[j]) {
            const uint32_t below = (1u << (__ffs(mark[j]) - 8)) - 1u;  // the bytes before the marker in this dword
            o += __popc(~drop[j] & 0x80808080u & below);
            found = true;
          }
          if (!found) o += __popc(~drop[j] & 0x80808080u);
        }
        atomicMin(&s_end, o);
      }
      if (valid == 16 && (drop[0] | drop[1] | drop[2] | drop[3]) == 0u) {
        __builtin_memcpy(clean8 + off, d, 16);
      } else {
        uint32_t o = off;
#pragma unroll
        for (int j = 0; j < 16; j++)
          if (j < valid && !((drop[j >> 2] >> (8 * (j & 3) + 7)) & 1u)) clean8[o++] = (uint8_t)(d[j >> 2] >> (8 * (j & 3)));
      }
    }
  }
  __syncthreads();
  const uint32_t L = min(s_end, kept_before);  // clean bytes
  if (t < 8) clean8[L + t] = 0;  // (zero bits are what libjpeg feeds past the end of the data; the buffer is padded)
  __threadfence_block();
  __syncthreads();
  const int ncomp = im.ncomp;
  const uint32_t nbits = L * 8u;
  const uint32_t ndw = (L + 3u) >> 2;
  ParGeom G;
  G.lum = im.h[0] * im.v[0];
  G.n1 = ncomp > 1 ? im.h[1] * im.v[1] : 0;
  G.m = G.lum + G.n1 + (ncomp > 2 ? im.h[2] * im.v[2] : 0);  // blocks per MCU
  G.dc_slots = (uint32_t)im.dc_slot[0] | ((uint32_t)im.dc_slot[ncomp > 1 ? 1 : 0] << 8) |
               ((uint32_t)im.dc_slot[ncomp > 2 ? 2 : 0] << 16);
  G.ac_slots = (uint32_t)im.ac_slot[0] | ((uint32_t)im.ac_slot[ncomp > 1 ? 1 : 0] << 8) |
               ((uint32_t)im.ac_slot[ncomp > 2 ? 2 : 0] << 16);
  const lds_u16 tab = (lds_u16)(&s_tab[0]);
  const lds_u8 zz = (lds_u8)(&s_zz[0]);
  uint32_t* coef32 = reinterpret_cast<uint32_t*>(coef_all + (size_t)blockIdx.x * (coef_stride / sizeof(int16_t)));
  if (ri) {
    // ---- files with restart intervals: every interval starts in a known state (byte-aligned, first block of an MCU,
    // DC prediction zero), so there is nothing to guess: thread t decodes intervals t, t + 256, ... straight into the
    // coefficient buffer (the writing pass of the other files, over the clean stream as it lies) ----
    const uint32_t nmcu = (uint32_t)(im.mcus_x * im.mcus_y);
    const uint32_t want = (nmcu + (uint32_t)ri - 1u) / (uint32_t)ri, have = min(rst_before + 1u, ivl_cap);
    const uint32_t n_use = min(want, have);
    bool broken = have < want;  // a restart marker is missing (what the one-wave decoder reports, too)
    if (t == 0) ivl[0] = 0u;
    __threadfence_block();
    __syncthreads();
    G.trans = nullptr;
    G.segdw = 1u;
    G.magic = 0u;
    ParWin W;
    W.col = nullptr;
    W.lin = clean;
    W.rows = ndw;
    W.gbase = 0u;
    W.jb = W.w0 = W.w1 = 0u;
    for (uint32_t j0 = 0; j0 < n_use; j0 += kParThreads) {
      const uint32_t j = j0 + (uint32_t)t;
      const bool live = j < n_use;
      const uint32_t begin = live ? ivl[j] * 8u : 0u, end = live && j + 1u < have ? ivl[j + 1u] * 8u : nbits;
      const uint32_t g = j * (uint32_t)ri * (uint32_t)G.m;
      const uint32_t blocks = live ? min((uint32_t)ri, nmcu - j * (uint32_t)ri) * (uint32_t)G.m : 0u;
      // (an interval whose data run out before its blocks do: gray MCUs, a warning in libjpeg)
      (void)par_write<true>(G, W, tab, zz, s_blk + wid * 64 * kBlkStride, lane, 0u, begin, 0, 0, g, 0, coef32, g + blocks, true, end);
    }
    if (broken) atomicOr(status, 2);
    return;
  }
  // ---- 2. segment end states until they stop changing ----
  const uint32_t seg = max(64u, ((nbits + kParThreads - 1) / kParThreads + 31u) & ~31u);
  const uint32_t limit = min((uint32_t)(t + 1) * seg, nbits);
  const uint32_t segdw = seg >> 5;
  for (uint32_t j = 0; j < segdw + kOverlap; j++) {  // the segment-major copy (zero bits past the end, as libjpeg feeds them)
    const uint32_t b = (uint32_t)t * segdw + j;
    trans[j * (uint32_t)kParThreads + (uint32_t)t] = b < ndw ? __builtin_bswap32(clean[b]) : 0u;
  }
  __threadfence_block();
  __syncthreads();
  G.trans = trans;
  G.segdw = segdw;
  G.magic = 0xFFFFFFFFu / segdw + 1u;
  ParWin W;
  W.col = trans + t;
  W.lin = nullptr;
  W.rows = segdw + kOverlap;
  W.gbase = (uint32_t)t * segdw;
  W.jb = W.w0 = W.w1 = 0u;
  uint32_t sq = (uint32_t)t * seg, sck = 0;  // assumed start: first block of an MCU, DC next (true for t == 0)
  ParMarks marks;
#pragma unroll
  for (int i = 0; i < kMarks; i++) {
    marks.q[i] = 0xFFFFFFFFu;
    marks.cr[i] = 0u;
    marks.dc[i] = 0;
  }
  if (t == 0) {
    s_q[0] = 0;
    s_ck[0] = 0;
    s_changed = 0;
  }
  {
    uint32_t q = min(sq, nbits), cnt;
    int c = 0, k = 0, dc;
    bool merged;
    par_count(G, W, tab, limit, q, c, k, marks, cnt, dc, merged);
    s_cnt[t] = cnt;
    s_dc[t] = dc;
    s_q[t + 1] = q;
    s_ck[t + 1] = ((uint32_t)c << 8) | (uint32_t)k;
  }
  __syncthreads();
  for (int round = 0; round < kParThreads; round++) {
    const uint32_t nq = s_q[t], nck = s_ck[t];  // the predecessor's end state (the truth for t == 0)
    bool redo = t > 0 && (nq != sq || nck != sck), merged = false;
    uint32_t q = nq, cnt = 0;
    int c = (int)(nck >> 8), k = (int)(nck & 255u), dc = 0;
    if (redo) par_count(G, W, tab, limit, q, c, k, marks, cnt, dc, merged);
    __syncthreads();  // (everybody has read its predecessor's state)
    if (redo) {
      sq = nq;
      sck = nck;
      s_cnt[t] = cnt;
      s_dc[t] = dc;
      if (!merged) {  // (merged: the rest of the way, end state included, is the one already on record)
        const uint32_t eck = ((uint32_t)c << 8) | (uint32_t)k;
        if (s_q[t + 1] != q || s_ck[t + 1] != eck) atomicOr(&s_changed, 1u);
        s_q[t + 1] = q;
        s_ck[t + 1] = eck;
      }
    }
    __syncthreads();
    const uint32_t any = s_changed;
    __syncthreads();
    if (t == 0) s_changed = 0;
    if (!any) break;  // uniform: nothing moved, so nothing will
  }
  __syncthreads();
  // (thread 0 never re-ran: its start was true; sq / sck of the others now equal their predecessors' end states)
  // ---- 3. block and DC offsets, then the writing pass ----
  const uint32_t mine = s_cnt[t];
  const int mine_dc = s_dc[t];
  uint32_t incb = mine;
  int incd = mine_dc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incb, o, 64);
    const int d = __shfl_up(incd, o, 64);
    if (lane >= o) {
      incb += v;
      incd += d;
    }
  }
  if (lane == 63) {
    s_scan[0][wid] = incb;
    s_scan[1][wid] = (uint32_t)incd;
  }
  __syncthreads();
  uint32_t g = incb - mine;
  int pred = incd - mine_dc;
  for (int w = 0; w < wid; w++) {
    g += s_scan[0][w];
    pred += (int)s_scan[1][w];
  }
  {
    const uint32_t total_blocks = (uint32_t)(im.mcus_x * im.mcus_y * G.m);
    uint32_t q = t == 0 ? 0u : s_q[t];
    const uint32_t ck = t == 0 ? 0u : s_ck[t];
    int c = (int)(ck >> 8), k = (int)(ck & 255u);
    (void)par_write(G, W, tab, zz, s_blk + wid * 64 * kBlkStride, lane, limit, q, c, k, g, pred, coef32, total_blocks,
              t == kParThreads - 1, nbits);
  }
}

// dequantisation + IDCT + range limit of 8 luminance blocks per wave, from the coefficient buffer
__global__ __launch_bounds__(64) void jpeg_idct_kernel(const DevImage* __restrict__ images,
                                                        const uint32_t* __restrict__ index,
                                                        const DevTables* __restrict__ tables,
                                                        const int16_t* __restrict__ coef_all, size_t coef_stride, int width,
                                                        int height, uint8_t* __restrict__ dst, size_t dst_image_stride,
                                                        int dst_pitch) {
  __shared__ int32_t s_ws[8][64];
  const uint32_t image = index[blockIdx.y];
  const DevImage& im = images[image];
  const int lum_w = im.h[0], lum = im.h[0] * im.v[0];
  const int nlb = im.mcus_x * im.mcus_y * lum;
  const int lane = threadIdx.x, b = blockIdx.x * 8 + (lane >> 3), i = lane & 7;
  const int16_t* coef = coef_all + (size_t)blockIdx.y * (coef_stride / sizeof(int16_t)) + (size_t)b * 64;
  const uint16_t* qt = tables[im.tables].qt_luma;
  if (b < nlb) {
    int32_t d[8], r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = (int32_t)coef[8 * k + i] * (int32_t)qt[8 * k + i];
    idct8_descaled(d, r, 11);
#pragma unroll
    for (int k = 0; k < 8; k++) s_ws[lane >> 3][8 * k + i] = r[k];
  }
  __syncthreads();
  if (b < nlb) {
    int32_t d[8], r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = s_ws[lane >> 3][8 * i + k];
    idct8_descaled(d, r, 18);
    const int mcu = b / lum, c = b - mcu * lum;
    const int my = mcu / im.mcus_x, mx = mcu - my * im.mcus_x;
    const int x0 = (mx * lum_w + c % lum_w) * 8, y = (my * im.v[0] + c / lum_w) * 8 + i;
    if (y < height && x0 < width) {
      uint8_t px[8];
#pragma unroll
      for (int k = 0; k < 8; k++) px[k] = range_limit(r[k]);
      uint8_t* row = dst + (size_t)image * dst_image_stride + (size_t)y * dst_pitch + x0;
      if (x0 + 8 <= width) {
        uint32_t lo, hi;
        memcpy(&lo, px, 4);
        memcpy(&hi, px + 4, 4);
        reinterpret_cast<uint32_t*>(row)[0] = lo;
        reinterpret_cast<uint32_t*>(row)[1] = hi;
      } else {
        for (int k = 0; k < 8 && x0 + k < width; k++) row[k] = px[k];
      }
    }
  }
}

}  // namespace

// Bytes of the parallel decoder's stream scratch: the clean streams in the layout of the upload's stream part, then
// their segment-major copies.
size_t vsf_jpeg_clean_bytes(size_t stream_bytes, int n_par) {
  const size_t linear = (stream_bytes + 64 + 255) & ~(size_t)255;
  return n_par > 0 ? 2 * linear + (size_t)n_par * kTransSlack : linear;
}

size_t vsf_jpeg_prog_huff_bytes(int n_tables) { return (size_t)n_tables * sizeof(DevHuffLite); }

// The parallel decoder's static + dynamic LDS exceed the default 64 KB for colour files: raised (checked) at vsf_create.
hipError_t vsf_prepare_jpeg_kernels(int lds_limit) {
  const int need = (int)(kMaxSlots * sizeof(DevHuff));
  if (need > lds_limit) return hipErrorInvalidValue;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(jpeg_par_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, need);
}

// Both decoders over one upload: the files without restart intervals (n_par of them, listed first in the index array
// at off_index) take the self-synchronising parallel decode, the others the one-wave-per-image decode.  d_clean has
// vsf_jpeg_clean_bytes(total - off_stream, n_par) bytes, d_coef holds n_par * coef_stride bytes (coef_stride = 128 * luminance blocks of the
// largest padded image): every luminance block is written whole.
void vsf_launch_jpeg_decode(const uint8_t* d_blob, size_t off_images, size_t off_index, size_t off_tables, size_t off_scans,
                            size_t off_prog_huff, size_t off_stream, size_t total, int n_par, int n_prog, int n_prog_huff,
                            void* d_prog_huff, int n_ser, int max_luma_blocks, int max_slots, int width, int height,
                            uint8_t* d_clean,
                            int16_t* d_coef, size_t coef_stride, uint8_t* d_dst, size_t dst_image_stride, int dst_pitch,
                            int32_t* d_status, hipStream_t s, bool prog_serial, int32_t* d_prog_flags) {
  const DevImage* images = reinterpret_cast<const DevImage*>(d_blob + off_images);
  const DevTables* tables = reinterpret_cast<const DevTables*>(d_blob + off_tables);
  const uint32_t* index = reinterpret_cast<const uint32_t*>(d_blob + off_index);
  if (n_par > 0) {
    // (static + dynamic LDS exceed 64 KB for colour files: vsf_prepare_jpeg_kernels raised the limit at vsf_create)
    hipLaunchKernelGGL(jpeg_par_decode_kernel, dim3(n_par), dim3(kParThreads), (size_t)max_slots * sizeof(DevHuff), s, images, index, tables, d_blob + off_stream,
                       reinterpret_cast<uint32_t*>(d_clean),
                       reinterpret_cast<uint32_t*>(d_clean + vsf_jpeg_clean_bytes(total - off_stream, 0)), d_coef, coef_stride,
                       max_slots, d_status);
  }
  if (n_prog > 0) {  // progressive files: their slots of the coefficient buffer follow the parallel decoder's
    if (n_prog_huff > 0)
      hipLaunchKernelGGL(jpeg_expand_huff_kernel, dim3(n_prog_huff), dim3(64), 0, s,
                         reinterpret_cast<const DevHuffSrc*>(d_blob + off_prog_huff), static_cast<DevHuffLite*>(d_prog_huff));
    const int32_t* only = nullptr;
    // The pipelined form pays while the CUs' scalar units have slots left: 256 files per call 7.1 k images/s against 3.7 k
    // scan after scan, 512 files 11.5 against 7.1 k, 768 files 14.9 against 10.3 k; from ~1000 files on the one-wave form
    // fills the scalar units by itself (1024: 13.1 / 13.3 k, 1536: 12.1 / 15.8 k, 2048: 13.5 / 16.7 k).
    if (!prog_serial && d_prog_flags && n_prog <= kProgPipeMaxFiles) {
      hipLaunchKernelGGL(jpeg_prog_pipe_kernel, dim3(n_prog), dim3(64 * kProgWaves), 0, s, images, index + n_par,
                         reinterpret_cast<const DevScan*>(d_blob + off_scans), static_cast<const DevHuffLite*>(d_prog_huff),
                         d_blob + off_stream, d_coef, coef_stride, n_par, width, height, d_prog_flags);
      only = d_prog_flags;  // (the one-wave kernel below then repeats the files that kernel flagged, and no others)
    }
    hipLaunchKernelGGL(jpeg_prog_kernel, dim3(n_prog), dim3(64), 0, s, images, index + n_par,
                       reinterpret_cast<const DevScan*>(d_blob + off_scans), static_cast<const DevHuffLite*>(d_prog_huff),
                       d_blob + off_stream, d_coef, coef_stride, n_par, width, height, d_status, only);
  }
  if (n_par + n_prog > 0)
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_luma_blocks + 7) / 8, n_par + n_prog), dim3(64), 0, s, images, index, tables,
                       d_coef, coef_stride, width, height, d_dst, dst_image_stride, dst_pitch);
  if (n_ser > 0)
    hipLaunchKernelGGL(jpeg_gray_kernel, dim3(n_ser), dim3(64), 0, s, images, index + n_par + n_prog, tables, d_blob + off_stream,
                       (uint32_t)((total - off_stream) & ~(size_t)3), width, height, d_dst, dst_image_stride, dst_pitch,
                       d_status);
}
