// k_png.hip -- SURVEY 8(f) row f4, the other format DecodeImage's cv::imdecode(msg.data, IMREAD_GRAYSCALE) reads
// (slam_frontend_main.cc:99-100): PNG.  Restates zlib's inflate (RFC 1951; the three block types, the canonical Huffman
// codes, the limits inflate_table enforces) and libpng's row filters (PNG specification section 9: None, Sub, Up, Average,
// Paeth) for grayscale files; the chunk walk and the CRCs are the host's (vsf_png_host.cc).
//
// png_inflate_kernel: one wave per file.  A deflate stream is one serial chain of codes -- where a symbol begins is known
// only when the one before it has been read -- but what WOULD be read from a bit position does not depend on the chain.
// So the lanes guess: in a span of 256 stream bits every lane looks up, for four bit positions, the symbol that would
// begin there (a literal, or a match with its extra bits, distance code and extra bits: two table reads from LDS, 10 and 9
// bits wide, built in the kernel for every block), and the serial part shrinks to following the symbols' lengths from the
// span's first bit: one v_readlane, an add, a min and a bit set per symbol, all scalar.  The guesses on that chain are the
// stream's symbols; their literals enter the ring at once, their matches in order.  What the tables do not hold is done as
// before: a canonical code is decoded by its definition with the lanes as the 15 possible lengths -- lane L holds the first
// code of length L, how many there are and where their symbols start in the sorted symbol list; it takes the first L bits
// and tests `bits - first < count`; the one lane that answers yes (a prefix code: at most one can) names the length -- and
// the walk outside the spans (block headers, code-length codes, the image's last bytes, everything behind them) is
// wave-uniform scalar code over a bit buffer fed from a register of stream dwords.  Output bytes go into a 32 KiB ring in
// LDS -- deflate's window -- matches as lane-parallel ring-to-ring copies (a match that overlaps itself reads its period
// from in front of its start); every completed 16 KiB leave for HBM as whole lines, their Adler-32 taken on the way.  What
// follows the image's last byte is read as far as zlib reads it in the call that delivers the last row (see the kernel).
//
// png_unfilter_kernel: one wave per file, 64 rows at a time as a wavefront (lane k is one byte behind lane k - 1, whose
// reconstructed byte of one step ago is its "above", of two steps ago its "above left"): every filter type per lane,
// branch-free.  Only the byte plane a gray read keeps is reconstructed (the filters work per byte plane: a 16-bit
// sample's low byte and an alpha sample never feed the high byte of the gray sample).  png_unfilter_general_kernel is the
// same for palette files (indices through 256 gray values) and interlaced ones (seven passes); png_unfilter_rgb_kernel
// reconstructs the colour planes in place and weights them as libpng's rgb_to_gray does.
#include "vsf_internal.h"
#include "vsf_png_host.h"

namespace {

using namespace vsf_png;

template <class T>
using cptr = const __attribute__((address_space(4))) T*;
template <class T>
__device__ __forceinline__ cptr<T> in_constant(const T* p) {
  return (cptr<T>)(uintptr_t)p;
}

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {  // lane i <- lane i-1 (lane 0: 0)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}

__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
#ifndef VSF_PNG_SPAN_REGS
#define VSF_PNG_SPAN_REGS 4
#endif
constexpr int kSpanRegs = VSF_PNG_SPAN_REGS, kSpanBits = 64 * kSpanRegs;  // symbols a lane tries per span; stream bits a span covers
constexpr int kLitBits = 10, kDistBits = 9;  // direct lookup tables in LDS: codes of up to this many bits

constexpr int kMaxLit = 288, kMaxDist = 32, kMaxCodes = kMaxLit + kMaxDist;
__constant__ uint8_t c_cl_order[20] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 0};

// ---- the deflate stream, least significant bit first; wave-uniform ----
// The stream's dwords come 64 at a time, one per lane (a coalesced vector load issued 64 dwords before its first use), and
// are handed to the scalar bit buffer one v_readlane at a time: no load latency on the path of a symbol.  `left` counts the
// bits the walk may still take (to the end of the stream, or -- behind the image's last byte -- to the end of the piece of
// input libpng would have handed zlib).
struct BitsLsb {
  const uint32_t* words;
  uint32_t nwords;    // dwords that exist (the upload pads every stream with zeros)
  uint32_t base;      // `cur` holds dwords base .. base + 63, `nxt` base + 32 .. base + 95 (the load of a later `cur`)
  uint32_t wpos;      // next dword to enter the bit buffer
  uint32_t cur, nxt;  // (per lane)
  uint64_t acc;
  int n;
  int64_t left;
  __device__ __forceinline__ uint32_t load64(uint32_t at) const {
    const uint32_t i = at + threadIdx.x;
    return i < nwords ? words[i] : 0u;
  }
  __device__ __forceinline__ void start(uint32_t at) {
    base = wpos = at;
    acc = 0;
    n = 0;
    cur = load64(at);
    nxt = load64(at + 32u);
  }
  // `cur` moves on by 32 dwords.  Every bit still in the buffer stays inside it (34: two dwords of them at most), which is
  // what lets the span decoder below pick the walk up from `cur` at any time.
  __device__ __forceinline__ void advance() {
    cur = nxt;
    base += 32u;
    nxt = load64(base + 32u);
  }
  __device__ __forceinline__ void fill() {  // >= 33 bits afterwards
    if (n <= 32) {
      if (wpos - base >= 34u) advance();
      const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(wpos - base));
      acc |= (uint64_t)w << n;
      ++wpos;
      n += 32;
    }
  }
  // the walk's position in bits from the start of `cur`, and the buffer set up again at such a position (< 63 * 32)
  __device__ __forceinline__ int rel_pos() const { return (int)(wpos - base) * 32 - n; }
  __device__ __forceinline__ void restart_at(int rel) {
    const int r = rel >> 5, b = rel & 31;
    acc = (uint64_t)((uint32_t)__builtin_amdgcn_readlane((int)cur, r) >> b);
    n = 32 - b;
    wpos = base + (uint32_t)r + 1u;
    fill();
  }
  __device__ __forceinline__ uint32_t peek() const { return (uint32_t)acc; }
  __device__ __forceinline__ void drop(int k) {
    acc >>= k;
    n -= k;
    left -= k;
  }
  __device__ __forceinline__ uint32_t take(int k) {  // k <= 16, after fill()
    const uint32_t v = (uint32_t)acc & ((1u << k) - 1u);
    drop(k);
    return v;
  }
  // bits / bytes of the stream consumed so far (whole bytes still in the buffer given back)
  __device__ __forceinline__ uint64_t bit_pos() const { return (uint64_t)wpos * 32u - (uint64_t)n; }
  __device__ __forceinline__ uint32_t byte_pos() const { return wpos * 4u - (uint32_t)(n >> 3); }
};

// One canonical code as the lanes hold it: lane L (1..15) knows the codes of length L.
struct LaneCode {
  uint32_t first, cnt, off;
};

// Decodes one symbol index (position in the sorted symbol list) from the next bits; *len = its length, or 0: no code.
__device__ __forceinline__ uint32_t decode_index(uint32_t peek, const LaneCode& c, int lane, int* len) {
  const uint32_t rev = __builtin_bitreverse32(peek);  // the stream's first bit on top
  const uint32_t top = rev >> ((32 - lane) & 31);  // (lanes other than 1 .. 15 hold cnt = 0: whatever they see, they say no)
  const uint32_t d = top - c.first;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(d < c.cnt);
  if (m == 0ull) {
    *len = 0;
    return 0u;
  }
  const int L = __builtin_ctzll(m);
  *len = L;
  return (uint32_t)__builtin_amdgcn_readlane((int)(c.off + d), L);
}

// Builds a canonical code from nsym lengths at lens[0..nsym) (LDS): per-lane description + the symbols sorted by (length,
// symbol) into sorted[0..) (LDS).  Returns false for a set of lengths zlib's inflate_table refuses: over-subscribed, or
// incomplete with anything but a single one-bit code (allow_lone; never for the code-length code).
struct BuiltCode {
  LaneCode c;
  bool ok;
};
// With `table`: also the direct lookup of the codes of up to `tbits` bits -- entry [the next tbits bits of the stream] =
// what the symbol says (table_entry), 0 where a longer code begins (the lanes decode those).
// An entry: bits 0-3 the code's length, 4-7 the number of extra bits behind it, 8-23 the value (a literal, or what the extra
// bits are added to: a match's length from 3, its distance from 1), bit 24: a match's length, bit 31: no literal, length or
// distance -- the end of the block, a symbol that must not occur, or (kLongCode, the length 0) a code the table does not hold.
constexpr uint32_t kMatchBit = 1u << 24, kHaltBit = 1u << 31, kLongCode = kHaltBit;
__device__ __forceinline__ uint32_t table_entry(uint32_t sym, int L, bool distance) {
  uint32_t eb = 0, value = sym, flags = 0;
  if (distance) {
    if (sym >= 30u) {
      flags = kHaltBit;  // "invalid distance code"
    } else if (sym < 4u) {
      value = 1u + sym;
    } else {
      eb = (sym >> 1) - 1u;
      value = 1u + ((2u + (sym & 1u)) << eb);
    }
  } else if (sym == 256u || sym > 285u) {
    flags = kHaltBit;  // the end of the block; 286, 287: "invalid literal/length code"
    value = 0;
  } else if (sym > 256u) {
    const uint32_t l = sym - 257u;
    flags = kMatchBit;
    if (l < 8u) {
      value = 3u + l;
    } else if (l == 28u) {
      value = 258u;
    } else {
      eb = (l >> 2) - 1u;
      value = 3u + ((4u + (l & 3u)) << eb);
    }
  }
  return (uint32_t)L | (eb << 4) | (value << 8) | flags;
}
template <int REGS>
__device__ __attribute__((noinline)) BuiltCode build_code(const lds_u8* lens, int nsym, lds_u16* sorted, bool allow_lone,
                                                          lds_u32* table, int tbits, bool distance) {
  const int lane = threadIdx.x;
  if (table)
    for (int i = lane; i < (1 << tbits); i += 64) table[i] = kLongCode;
  uint32_t len_r[REGS];
  int pos_r[REGS];
#pragma unroll
  for (int r = 0; r < REGS; r++) {
    const int s = r * 64 + lane;
    len_r[r] = s < nsym ? lens[s] : 0u;
    pos_r[r] = 0;
  }
  LaneCode c{0u, 0u, 0u};
  int code = 0, offs = 0, left = 1, maxlen = 0;
  bool ok = true;
  for (int L = 1; L <= 15; L++) {
    left <<= 1;
    int n = 0;
#pragma unroll
    for (int r = 0; r < REGS; r++) {
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(len_r[r] == (uint32_t)L);
      const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
      if (len_r[r] == (uint32_t)L) {
        pos_r[r] = offs + n + below;
        if (table && L <= tbits) {  // (every lane in here has a code of this length: the same number of entries each)
          const uint32_t rev = __builtin_bitreverse32((uint32_t)(code + n + below)) >> (32 - L);
          const uint32_t e = table_entry((uint32_t)(r * 64 + lane), L, distance);
          for (int k = 0; k < (1 << (tbits - L)); k++) table[rev | ((uint32_t)k << L)] = e;
        }
      }
      n += __popcll(bal);
    }
    left -= n;
    if (left < 0) ok = false;
    if (n > 0) maxlen = L;
    if (lane == L) c = LaneCode{(uint32_t)code, (uint32_t)n, (uint32_t)offs};
    code = (code + n) << 1;
    offs += n;
  }
  // incomplete: inflate_table allows it for a lone one-bit code of the literal / length and distance codes (and for an empty set)
  if (left > 0 && (!allow_lone || (offs > 0 && maxlen != 1))) ok = false;
#pragma unroll
  for (int r = 0; r < REGS; r++)
    if (len_r[r] != 0u) sorted[pos_r[r]] = (uint16_t)(r * 64 + lane);
  return BuiltCode{c, ok};
}

// Adler-32 (RFC 1950) of ring[from .. from + n) continued from (a, b), returned as a | b << 32 -- n <= kFlushChunk, `from` a multiple of it -- and,
// with `store`, the same bytes written to out + from in 16-byte pieces (a partial last piece is written whole: the
// image's slot has the slack).  One copy of this code in the kernel (it is needed at five places of the walk).
__device__ __attribute__((noinline)) uint64_t ring_chunk(lds_u8* ring, uint8_t* out, uint32_t from, uint32_t n, uint32_t a,
                                                         uint32_t b, bool store) {
  const int lane = threadIdx.x;
  constexpr uint32_t M = kWindow - 1;
  const uint32_t seg = kFlushChunk / 64;  // bytes per lane
  const uint32_t lo = lane * seg, hi = min(lo + seg, n);
  uint32_t s1 = 0, s2 = 0;
  for (uint32_t i = lo; i < hi; i += 16) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 q = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(ring + ((from + i) & M));
    if (store) *reinterpret_cast<u32x4*>(out + from + i) = q;
    const uint32_t wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t idx = i + 4u * k;
      // bytes behind the end do not count
      const uint32_t keep = idx + 4u <= hi ? 0xFFFFFFFFu : idx >= hi ? 0u : (1u << (8u * (hi - idx))) - 1u;
      const uint32_t w = wv[k] & keep;
      const uint32_t sum4 = __builtin_amdgcn_sad_u8(w, 0u, 0u);
      s1 += sum4;
      // a byte counts once for every byte from it to the end: (n - idx) * (d0 + d1 + d2 + d3) - (0 d0 + 1 d1 + 2 d2 + 3 d3)
      s2 += (n - idx) * sum4 - __builtin_amdgcn_udot4(w, 0x03020100u, 0u, false);
    }
  }
  uint64_t t1 = s1, t2 = s2;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    t1 += __shfl_xor(t1, o, 64);
    t2 += __shfl_xor(t2, o, 64);
  }
  const uint64_t na = ((uint64_t)a + t1) % 65521u;
  const uint64_t nb = ((uint64_t)b + (uint64_t)n * a + t2) % 65521u;
  return na | (nb << 32);
}

// (a -DVSF_PNG_STATS build counts what the span decoder meets: tools/exp/png_stats.py)
#ifdef VSF_PNG_STATS
__device__ unsigned long long g_png_stats[16];
#define PNG_STAT(i, v) do { if (lane == 0) atomicAdd(&g_png_stats[i], (unsigned long long)(v)); } while (0)
#else
#define PNG_STAT(i, v) do { } while (0)
#endif

struct PngArgs {
  const uint8_t* blob;
  size_t off_images, off_pieces, off_tables, off_stream;
  uint8_t* filtered;       // [n][filtered_stride]: the inflated scanlines (filter byte + row bytes)
  size_t filtered_stride;
  int width, height;
  uint8_t* dst;
  size_t dst_image_stride;
  int dst_pitch;
  int32_t* status;         // bit 1: a file's data is broken (what libpng answers with png_error)
  int32_t* file_status;    // [n]: 0 ok, 1 broken: the unfilter kernel leaves such an image alone
};

__device__ __forceinline__ DevImage load_image(const PngArgs& a, int image) {
  DevImage im;
  const uint32_t* iw = reinterpret_cast<const uint32_t*>(a.blob + a.off_images) + kDevImageWords * image;
  im.stream_off = in_constant(iw)[0];
  im.stream_len = in_constant(iw)[1];
  im.row_bytes = in_constant(iw)[2];
  const uint32_t t = in_constant(iw)[3];
  im.bpp = (uint8_t)(t & 0xFFu);
  im.depth = (uint8_t)((t >> 8) & 0xFFu);
  im.kind = (uint8_t)((t >> 16) & 0xFFu);
  im.flags = (uint8_t)(t >> 24);
  im.piece_first = in_constant(iw)[4];
  im.piece_count = in_constant(iw)[5];
  im.table = in_constant(iw)[6];
  im.expected = in_constant(iw)[7];
  return im;
}

// The end (inside the zlib stream) of the piece of input that holds byte b: libpng hands zlib at most 8192 bytes of ONE chunk.
__device__ __attribute__((noinline)) uint32_t piece_end_of(const uint8_t* blob, size_t off_pieces, uint32_t piece_first, uint32_t piece_count,
                                                          uint32_t stream_len, uint32_t b) {
  const uint32_t* pe = reinterpret_cast<const uint32_t*>(blob + off_pieces) + piece_first;
  uint32_t begin = 0, end = stream_len;
  for (uint32_t k = 0; k < piece_count; k++) {
    const uint32_t e = in_constant(pe)[k];
    if (b < e) {
      end = e;
      break;
    }
    begin = e;
  }
  return min(min(end, begin + ((b - begin) / kIdatReadSize + 1u) * kIdatReadSize), stream_len);
}
// Where the drain (see below) may read to when it begins at bit bp: the end of the piece at hand, or -- zlib having taken all
// of that (`exhausted`, or the position says so) -- of the piece libpng fetches next (bit 63 of the answer: such a piece);
// 0: there is nothing left to fetch.
__device__ __attribute__((noinline)) uint64_t drain_limit(const uint8_t* blob, size_t off_pieces, uint32_t piece_first, uint32_t piece_count,
                                                          uint32_t stream_len, uint64_t bp, bool exhausted) {
  const uint32_t pulled = (uint32_t)((bp + 7u) / 8u);  // bytes zlib has taken
  const uint32_t here = piece_end_of(blob, off_pieces, piece_first, piece_count, stream_len, pulled ? pulled - 1u : 0u);
  if (!exhausted && pulled < here) return (uint64_t)here * 8u;
  if (here >= stream_len) return 0ull;
  return (uint64_t)piece_end_of(blob, off_pieces, piece_first, piece_count, stream_len, here) * 8u | (1ull << 63);
}

// What happens behind the image's last byte is zlib's business too.  libpng asks zlib for the last row with whatever is
// left of the CURRENT piece of input (at most 8192 bytes of one IDAT chunk: PNG_IDAT_READ_SIZE); having written the last
// byte, inflate() goes on reading symbols that need no room in the output -- an end-of-block code, block headers, code
// tables, empty stored blocks, the Adler-32 check -- and an error it meets there fails the read (png_error), exactly as
// in front of the last byte.  It stops without an error at the first symbol that needs room (after decoding a match's
// length and distance codes), at the end of that piece of input, or at the end of the stream.  If the stream has not
// ended, cv::imdecode's png_read_end then DRAINS it (png_read_finish_IDAT: the rest of the compressed data is inflated into
// a scratch buffer piece after piece): what zlib finds wrong there is a warning, and so is a stream that goes on behind
// the image -- but IDAT data that runs out before the stream has ended is png_error("Not enough image data"): a file cut
// inside its last bytes, the check value included, is refused although every pixel was there.  (One way out: a piece
// fetched for the drain that yields no output and no end lets libpng's loop finish quietly.)  So the walk below has three
// modes: in front of the last byte running out of input is an error; behind it the input ends at the piece boundary and
// running out starts the drain; in the drain nothing is written, errors end the walk, and running out is an error again.
__global__ __launch_bounds__(64) void png_inflate_kernel(PngArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t ring[kWindow];
  __shared__ uint8_t lens[kMaxCodes + 8];
  __shared__ uint8_t cl_lens[32];
  __shared__ uint16_t sorted[kMaxCodes];
  __shared__ uint32_t lit_table[1 << kLitBits];
  __shared__ uint32_t dist_table[1 << kDistBits];
  const int lane = threadIdx.x;
  const int image = blockIdx.x;
  const DevImage im = load_image(a, image);
  const uint32_t expected = im.expected;  // every scanline (of every pass) with its filter byte
  uint8_t* out = a.filtered + (size_t)image * a.filtered_stride;
  constexpr uint32_t M = kWindow - 1;

  BitsLsb br;
  br.words = reinterpret_cast<const uint32_t*>(a.blob + a.off_stream + im.stream_off);
  br.nwords = (im.stream_len + 3u) / 4u + 8u;
  br.left = (int64_t)im.stream_len * 8;
  br.start(0);
  br.fill();
  br.drop(16);  // the zlib header (checked on the host)

  uint32_t pos = 0, flushed = 0;  // bytes produced; bytes already in HBM (a multiple of kFlushChunk)
  uint32_t pend = 0;              // lane k: the k-th literal not yet in the ring
  int npend = 0;
  bool bad = false, stop = false, tail = false;
  // Behind the last row (see the comment above the kernel): `drain` = the rest of the stream is read as png_read_end reads
  // it, for nothing but its end; `extra`: bytes it would have produced; `refilled`: the piece at hand is one libpng fetched
  // for that; `whole`: the walk may go on to the end of the stream.
  bool drain = false, refilled = false, whole = false;
  uint32_t extra = 0, carry = 0;
  uint32_t adler_a = 1, adler_b = 0;  // Adler-32 of out[0 .. flushed)

  lds_u8* const ring_lds = (lds_u8*)ring;
  auto flush_chunks = [&]() {  // completed 16 KiB pieces of the ring leave for HBM, their Adler-32 taken on the way
    while (pos - flushed >= (uint32_t)kFlushChunk) {
      const uint64_t ab = ring_chunk(ring_lds, out, flushed, kFlushChunk, adler_a, adler_b, true);
      adler_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ab);
      adler_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ab >> 32));
      flushed += kFlushChunk;
    }
  };
  auto flush_literals = [&]() {
    if (npend > 0) {
      if (lane < npend) ring[(pos + lane) & M] = (uint8_t)pend;
      pos += (uint32_t)npend;
      npend = 0;
      flush_chunks();
    }
  };
  // What zlib finds wrong in front of the image's last byte, or behind it in the call that delivers the last row, fails the
  // read; what it finds wrong while png_read_end drains the rest is a warning and ends the walk.
  auto fail = [&]() {
    if (!drain) bad = true;
    stop = true;
  };
  // The last row is out and zlib has stopped -- at the end of its piece of input (`exhausted`), or at a symbol that needs
  // room.  png_read_end (png_read_finish_IDAT) now inflates what is left into a scratch buffer, piece after piece, until the
  // stream ends or breaks (fine either way) -- or the IDAT data runs out first: png_error("Not enough image data").  One
  // way out without either: a piece fetched for this that gives no output at all ends the loop quietly.
  auto begin_drain = [&](bool exhausted) {
    drain = true;
    const uint64_t bp = br.bit_pos();
    const uint64_t lim = drain_limit(a.blob, a.off_pieces, im.piece_first, im.piece_count, im.stream_len, bp, exhausted);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lim), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(lim >> 32));
    if ((lo | hi) == 0u) {  // nothing left to fetch
      bad = true;
      stop = true;
      return;
    }
    refilled = (hi >> 31) != 0u;
    br.left = (int64_t)(((uint64_t)(hi & 0x7FFFFFFFu) << 32) | lo) - (int64_t)bp;
  };
  // k more bits of input?  If not: an error in front of the image's last byte; behind it the end of zlib's call, and the
  // beginning of the drain; in the drain the end of the input -- an error unless the piece gave nothing.
  auto lacks = [&](int k) -> bool {
    if ((int64_t)k <= br.left) return false;
    if (tail && !drain) {
      begin_drain(true);
      if (stop) return true;
      if ((int64_t)k <= br.left) return false;
    }
    if (drain && !whole) {
      if (refilled && extra == 0u) {  // (libpng's loop ends: no output, no error)
        stop = true;
        return true;
      }
      whole = true;
      br.left = (int64_t)im.stream_len * 8 - (int64_t)br.bit_pos();
      if ((int64_t)k <= br.left) return false;
    }
    bad = true;  // "Not enough image data"
    stop = true;
    return true;
  };
  // The image's last byte has been produced: from here on the input ends where libpng's current piece of it ends.
  auto enter_tail = [&]() {
    flush_literals();
    tail = true;
    const uint64_t bp = br.bit_pos();
    const uint32_t last_byte = (uint32_t)((bp + 7u) / 8u) - 1u;  // the last byte zlib has pulled
    const uint32_t pe = (uint32_t)__builtin_amdgcn_readfirstlane((int)piece_end_of(a.blob, a.off_pieces, im.piece_first, im.piece_count, im.stream_len, last_byte));
    br.left = (int64_t)((uint64_t)pe * 8u) - (int64_t)bp;
  };

  bool last = false;
  while (!stop && !bad) {
    if (!tail && pos + (uint32_t)npend >= expected) enter_tail();
    if (last) {  // the end of the stream: to the byte boundary, then the Adler-32 of everything produced, big-endian
      flush_literals();
      br.drop(br.n & 7);
      br.fill();
      if (lacks(32)) break;
      if (drain) {  // (a wrong check value is a warning here, a right one the end)
        stop = true;
        break;
      }
      const uint32_t v = br.take(16);
      br.fill();
      const uint32_t v2 = br.take(16);
      const uint32_t le = v | (v2 << 16);
      const uint32_t stored = (le >> 24) | ((le >> 8) & 0xFF00u) | ((le << 8) & 0xFF0000u) | (le << 24);
      const uint64_t ab = ring_chunk(ring_lds, out, flushed, pos - flushed, adler_a, adler_b, false);
      const uint32_t fa = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ab);
      const uint32_t fb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ab >> 32));
      if (stored != ((fb << 16) | fa)) bad = true;  // inflate: "incorrect data check"
      stop = true;
      break;
    }
    br.fill();
    if (lacks(3)) break;
    last = br.take(1) != 0u;
    const uint32_t btype = br.take(2);
    if (btype == 3u) {  // inflate: "invalid block type"
      fail();
      break;
    }
    if (btype == 0u) {  // stored: to the byte boundary, LEN, ~LEN, LEN bytes
      flush_literals();
      br.drop(br.n & 7);
      br.fill();
      if (lacks(32)) break;
      const uint32_t len = br.take(16);
      br.fill();
      const uint32_t nlen = br.take(16);
      if ((len ^ 0xFFFFu) != nlen) {  // inflate: "invalid stored block lengths"
        fail();
        break;
      }
      if (len == 0u) continue;
      const uint32_t src0 = br.byte_pos();
      if (tail) {  // bytes to copy and no room for them: inflate leaves, the drain takes them
        if (!drain) begin_drain(false);
        if (stop) break;
        extra += len;
        if (src0 + len > im.stream_len) {  // the input ends inside them
          bad = true;
          break;
        }
        const uint32_t next = src0 + len;
        br.start(next >> 2);
        br.fill();
        br.drop((int)(next & 3u) * 8);
        whole = true;
        br.left = (int64_t)im.stream_len * 8 - (int64_t)next * 8;
        continue;
      }
      const uint32_t want = min(len, expected - pos);
      if (src0 + want > im.stream_len) {  // the stream ends inside the bytes the image still needs
        bad = true;
        break;
      }
      const uint8_t* sbytes = reinterpret_cast<const uint8_t*>(br.words);
      for (uint32_t i = 0; i < want; i += 64) {
        if (i + lane < want) ring[(pos + lane) & M] = sbytes[src0 + i + lane];
        pos += min(64u, want - i);
        flush_chunks();
      }
      if (want < len) {  // the image is complete in the middle of the block: inflate leaves, the drain takes the rest
        enter_tail();
        begin_drain(false);
        if (stop) break;
        extra += len - want;
        if (src0 + len > im.stream_len) {
          bad = true;
          break;
        }
        const uint32_t next = src0 + len;
        br.start(next >> 2);
        br.fill();
        br.drop((int)(next & 3u) * 8);
        whole = true;
        br.left = (int64_t)im.stream_len * 8 - (int64_t)next * 8;
        continue;
      }
      // the bit buffer continues behind the block's bytes
      const uint32_t next = src0 + len;
      {  // (the bits up to `next` are spent: `left` follows the position)
        const int64_t spent = (int64_t)next * 8 - (int64_t)br.bit_pos();
        br.start(next >> 2);
        br.fill();
        br.drop((int)(next & 3u) * 8);
        br.left -= spent - (int64_t)(next & 3u) * 8;
      }
      continue;
    }
    // ---- the block's two codes
    int hlit = kMaxLit, hdist = kMaxDist;
    if (btype == 1u) {  // fixed (RFC 1951 3.2.6)
      for (int s = lane; s < kMaxLit; s += 64) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
      if (lane < kMaxDist) lens[kMaxLit + lane] = 5;
    } else {
      br.fill();
      if (lacks(14)) break;
      hlit = (int)br.take(5) + 257;
      hdist = (int)br.take(5) + 1;
      const int hclen = (int)br.take(4) + 4;
      if (hlit > 286 || hdist > 30) {  // inflate: "too many length or distance symbols"
        fail();
        break;
      }
      // the code-length code: 19 symbols, 3-bit lengths in a fixed order
      if (lane < 32) cl_lens[lane] = 0;
      __syncthreads();
      for (int i = 0; i < hclen && !stop; i++) {
        br.fill();
        if (lacks(3)) break;
        const uint32_t v = br.take(3);
        const int sym = in_constant(c_cl_order)[i];
        if (lane == 0) cl_lens[sym] = (uint8_t)v;
      }
      if (stop) break;
      __syncthreads();
      const BuiltCode clb = build_code<1>((const lds_u8*)cl_lens, 19, (lds_u16*)sorted, false, nullptr, 0, false);
      const LaneCode clc = clb.c;
      if (__builtin_amdgcn_readfirstlane((int)clb.ok) == 0) {  // inflate: "invalid code lengths set"
        fail();
        break;
      }
      __syncthreads();
      const uint32_t clsym = lane < 19 ? sorted[lane] : 0u;
      __syncthreads();
      const int total = hlit + hdist;
      int i = 0;
      uint32_t prev = 0;
      while (i < total && !stop && !bad) {
        br.fill();
        int L;
        const uint32_t idx = decode_index(br.peek(), clc, lane, &L);
        if (L == 0) {  // (a code-length code without a single code: nothing it could decode)
          if (!lacks(1)) fail();
          break;
        }
        const uint32_t sym = (uint32_t)__builtin_amdgcn_readlane((int)clsym, (int)(idx & 63u));
        const int extra = sym == 16u ? 2 : sym == 17u ? 3 : sym == 18u ? 7 : 0;
        if (lacks(L + extra)) break;
        br.drop(L);
        if (sym < 16u) {
          if (lane == 0) lens[i] = (uint8_t)sym;
          prev = sym;
          ++i;
        } else {
          int rep;
          uint32_t val = 0;
          if (sym == 16u) {
            if (i == 0) {  // inflate: "invalid bit length repeat"
              fail();
              break;
            }
            rep = 3 + (int)br.take(2);
            val = prev;
          } else if (sym == 17u) {
            rep = 3 + (int)br.take(3);
            prev = 0;
          } else {
            rep = 11 + (int)br.take(7);
            prev = 0;
          }
          if (i + rep > total) {
            fail();
            break;
          }
          for (int k = lane; k < rep; k += 64) lens[i + k] = (uint8_t)val;
          i += rep;
        }
      }
      if (bad || stop) break;
      __syncthreads();
      if (lens[256] == 0) {  // inflate: "invalid code -- missing end-of-block"
        fail();
        break;
      }
    }
    __syncthreads();
    const BuiltCode lb = build_code<5>((const lds_u8*)lens, hlit, (lds_u16*)sorted, true, (lds_u32*)lit_table, kLitBits, false);
    const BuiltCode db = build_code<1>((const lds_u8*)lens + hlit, hdist, (lds_u16*)sorted + kMaxLit, true, (lds_u32*)dist_table, kDistBits, true);
    const LaneCode lc = lb.c, dc = db.c;
    if (__builtin_amdgcn_readfirstlane((int)lb.ok) == 0 || __builtin_amdgcn_readfirstlane((int)db.ok) == 0) {  // inflate: "invalid literal/lengths set", "invalid distances set"
      fail();
      break;
    }
    __syncthreads();
    // the sorted symbol lists in registers, two symbols per lane: literal / length indices 0..127, 128..255, 256..287
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(sorted);
    const uint32_t P0 = sp[lane], P1 = sp[64 + lane], P2 = lane < (kMaxLit - 256) / 2 ? sp[128 + lane] : 0u;
    const uint32_t dsy = lane < kMaxDist ? sorted[kMaxLit + lane] : 0u;
    __syncthreads();
    // (codes are sorted by length, so the frequent symbols sit in the first register: one v_readlane)
    auto lit_symbol = [&](uint32_t idx) -> uint32_t {
      const int li = (int)((idx >> 1) & 63u);
      uint32_t v;
      if (idx < 128u) {
        v = (uint32_t)__builtin_amdgcn_readlane((int)P0, li);
      } else {
        const uint32_t v1 = (uint32_t)__builtin_amdgcn_readlane((int)P1, li), v2 = (uint32_t)__builtin_amdgcn_readlane((int)P2, li);
        v = idx < 256u ? v1 : v2;
      }
      return (v >> ((idx & 1u) * 16u)) & 0xFFFFu;
    };

    // ---- the block's symbols
    while (!stop && !bad) {
      if (!tail && pos + (uint32_t)npend >= expected) enter_tail();
      if (!tail) {
        // Literals and matches in a loop of their own with nothing but scalar state.  It ends, with the symbol still in the
        // stream, at anything out of the ordinary -- the end of the block, a code that is none, the end of the input near --
        // and when the image is complete or a piece of the ring is due in HBM; the general walk below takes over there.
        int budget = (int)min(br.left, (int64_t)(1 << 30));
        const int budget0 = budget;
        int room = (int)min(expected - pos - (uint32_t)npend, 1u << 30);
        bool spans = true;
        while (true) {
          // ---- Spans: 256 symbols at a time, four per lane.  Lane i reads the stream from bits i, 64 + i, 128 + i and 192 + i of
          // the walk's position as if a symbol began there -- a literal, or a match with its extra bits, distance code and extra
          // bits, all by way of the lookup tables -- and says how many bits that is; the walk then only has to follow those
          // lengths from bit 0 (one v_readlane per symbol) to know which guesses began where a symbol begins.  Their literals go
          // into the ring at once, their matches one after the other, each copied by the whole wave.  Anything else -- a code the
          // tables do not hold, the end of the block, a match reaching in front of the data, the end of the image or of the
          // input near -- ends the span in front of that symbol and is the business of the code further down.
          if (spans && budget >= kSpanBits + 128 && room > kSpanBits) {
            if (npend > 0) {
              if (lane < npend) ring[(pos + lane) & M] = (uint8_t)pend;
              pos += (uint32_t)npend;
              npend = 0;
            }
            int rel = br.rel_pos();
            while (true) {
              if (rel >= 1024) {
                br.advance();
                rel -= 1024;
              }
              if (budget < kSpanBits + 128 || room <= kSpanBits) break;
              // every lane's view of the stream: 64 bits from each of its positions
              const int r = rel >> 5;
              const int t = (rel & 31) + lane;
              uint32_t sel[2 * kSpanRegs + 1];
              {
                uint32_t w[2 * kSpanRegs + 3];
#pragma unroll
                for (int m = 0; m < 2 * kSpanRegs + 3; m++) w[m] = (uint32_t)__builtin_amdgcn_readlane((int)br.cur, r + m);
#pragma unroll
                for (int m = 0; m < 2 * kSpanRegs + 1; m++) sel[m] = t < 32 ? w[m] : t < 64 ? w[m + 1] : w[m + 2];
              }
              // (three passes, so that the eight table reads are two batches in flight and not eight round trips)
              uint32_t view[kSpanRegs], high[kSpanRegs], ent[kSpanRegs], ent2[kSpanRegs], behind[kSpanRegs];
              uint32_t len[kSpanRegs], dist[kSpanRegs], bits[kSpanRegs], hop[kSpanRegs], shift[kSpanRegs];
              int match_mask[kSpanRegs];  // 0 / -1
              uint64_t halts[kSpanRegs];
#pragma unroll
              for (int k = 0; k < kSpanRegs; k++) {
                view[k] = __builtin_amdgcn_alignbit(sel[2 * k + 1], sel[2 * k], (uint32_t)t & 31u);
                high[k] = __builtin_amdgcn_alignbit(sel[2 * k + 2], sel[2 * k + 1], (uint32_t)t & 31u);
                ent[k] = lit_table[view[k] & ((1u << kLitBits) - 1u)];
              }
#pragma unroll
              for (int k = 0; k < kSpanRegs; k++) {
                const uint32_t L = ent[k] & 15u, eb = (ent[k] >> 4) & 15u;
                len[k] = ((ent[k] >> 8) & 0xFFFFu) + __builtin_amdgcn_ubfe(view[k], L, eb);  // (a literal: its value)
                shift[k] = L + eb;
                behind[k] = __builtin_amdgcn_alignbit(high[k], view[k], shift[k]);
                ent2[k] = dist_table[behind[k] & ((1u << kDistBits) - 1u)];
              }
#pragma unroll
              for (int k = 0; k < kSpanRegs; k++) {
                match_mask[k] = __builtin_amdgcn_sbfe((int)ent[k], 24, 1);
                const uint32_t L2 = ent2[k] & 15u, eb2 = (ent2[k] >> 4) & 15u;
                dist[k] = ((ent2[k] >> 8) & 0xFFFFu) + __builtin_amdgcn_ubfe(behind[k], L2, eb2);
                bits[k] = shift[k] + ((L2 + eb2) & (uint32_t)match_mask[k]);
                const bool halt = (int)(ent[k] | (ent2[k] & (uint32_t)match_mask[k])) < 0;
                hop[k] = halt ? 64u : bits[k];
                halts[k] = __builtin_amdgcn_ballot_w64(halt);
              }
              // the positions a symbol really begins at: register after register, from where the one before ended
              uint64_t active[kSpanRegs];
              int end = 0;
              {
                int entry = 0;
                bool open = true;
#pragma unroll
                for (int k = 0; k < kSpanRegs; k++) {
                  active[k] = 0ull;
                  if (open) {
                    uint64_t chain = 0ull;
                    uint32_t q = (uint32_t)entry;
                    asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(q));
                    while (true) {
                      do {  // (behind the register's end q stays 64: lane select 64 reads lane 0, bit 64 is bit 0 -- see below)
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                          q = min(q + (uint32_t)__builtin_amdgcn_readlane((int)hop[k], (int)q), 64u);
                          asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(q));
                        }
                      } while (q < 64u);
                      if (entry != 0) chain &= ~1ull;
                      const int last = 63 - __builtin_clzll(chain);
                      if (!((halts[k] >> last) & 1ull)) {
                        const int e = last + __builtin_amdgcn_readlane((int)bits[k], last);  // (>= 64)
                        end = 64 * k + e;
                        entry = e - 64;
                        break;
                      }
                      // The walk has met a symbol the tables do not hold.  A literal with a long code -- the usual case -- is
                      // decoded here, by the lanes, and entered where the tables' answer would be: the walk goes on.
                      bool mended = false;
                      if (((uint32_t)__builtin_amdgcn_readlane((int)ent[k], last) & 15u) == 0u) {
                        int L;
                        const uint32_t idx = decode_index((uint32_t)__builtin_amdgcn_readlane((int)view[k], last), lc, lane, &L);
                        if (L != 0) {
                          const uint32_t sym = lit_symbol(idx);
                          if (sym < 256u) {
                            asm("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %5, m0"
                                : "+v"(len[k]), "+v"(bits[k]), "+v"(hop[k])
                                : "s"(sym), "s"(last), "s"(L));
                            halts[k] &= ~(1ull << last);
                            q = (uint32_t)last;
                            mended = true;
                          }
                        }
                      }
                      if (!mended) {
                        end = 64 * k + last;
                        open = false;
                        break;
                      }
                    }
                    active[k] = chain & ~halts[k];
                  }
                }
              }
              if (end == 0) break;  // (the symbol at the walk's position is one for the code below)
              uint64_t matches[kSpanRegs], any_match = 0ull;
#pragma unroll
              for (int k = 0; k < kSpanRegs; k++) {
                matches[k] = active[k] & __builtin_amdgcn_ballot_w64(match_mask[k] != 0);
                any_match |= matches[k];
              }
              uint32_t offs[kSpanRegs], total = 0;
              if (any_match == 0ull) {
#pragma unroll
                for (int k = 0; k < kSpanRegs; k++) {
                  offs[k] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(active[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active[k], 0u));
                  total += (uint32_t)__popcll(active[k]);
                }
              } else {
                int mine[kSpanRegs];
#pragma unroll
                for (int k = 0; k < kSpanRegs; k++) {
                  mine[k] = (active[k] >> lane) & 1ull ? (match_mask[k] ? (int)len[k] : 1) : 0;
                  const int incl = wave_incl_scan(mine[k]);
                  offs[k] = total + (uint32_t)(incl - mine[k]);
                  total += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
                }
                // The span ends in front of a match that reaches in front of the data ("invalid distance too far back"), in front
                // of one from so far back that its bytes share their places in the ring with the span's later ones, which are
                // written first (at the full distance of 32 KiB a byte's place is its source's: no harm), and where it would
                // leave more than 8 KiB in the ring for the next flush.
                bool cut = false, none = false;
#pragma unroll
                for (int k = 0; k < kSpanRegs; k++) {
                  if (cut) {
                    active[k] = 0ull;
                    matches[k] = 0ull;
                  } else {
                    const uint64_t far = matches[k] & __builtin_amdgcn_ballot_w64(dist[k] > pos + offs[k] || (dist[k] != (uint32_t)kWindow && dist[k] + (total - offs[k]) > (uint32_t)kWindow) ||
                                                                                  offs[k] + (uint32_t)mine[k] > 8192u);
                    if (far != 0ull) {
                      const int at = __builtin_ctzll(far);
                      const uint64_t keep = (1ull << at) - 1ull;
                      active[k] &= keep;
                      matches[k] &= keep;
                      end = 64 * k + at;
                      total = (uint32_t)__builtin_amdgcn_readlane((int)offs[k], at);
                      cut = true;
                      none = end == 0;
                    }
                  }
                }
                if (none) break;
                if ((int)total >= room) {  // the image ends inside this span: symbol by symbol from here on
                  spans = false;
                  break;
                }
              }
#pragma unroll
              for (int k = 0; k < kSpanRegs; k++)
                if (((active[k] >> lane) & 1ull) && !match_mask[k]) ring[(pos + offs[k]) & M] = (uint8_t)len[k];
#pragma unroll
              for (int k = 0; k < kSpanRegs; k++) {
                uint64_t todo = matches[k];
                while (todo != 0ull) {
                  const int at_lane = __builtin_ctzll(todo);
                  todo &= todo - 1ull;
                  const uint32_t mlen = (uint32_t)__builtin_amdgcn_readlane((int)len[k], at_lane), mdist = (uint32_t)__builtin_amdgcn_readlane((int)dist[k], at_lane);
                  const uint32_t at = pos + (uint32_t)__builtin_amdgcn_readlane((int)offs[k], at_lane);
                  if (mdist >= mlen) {
                    for (uint32_t i = lane; i < mlen; i += 64) {
                      const uint8_t bt = ring[(at - mdist + i) & M];
                      ring[(at + i) & M] = bt;
                    }
                  } else {  // the match overlaps itself: byte i repeats byte i mod dist (all of them in front of its start)
                    const float rcp = 1.0f / (float)mdist;
                    for (uint32_t i = lane; i < mlen; i += 64) {
                      const uint32_t qq = (uint32_t)((float)i * rcp);
                      int rr = (int)i - (int)(qq * mdist);
                      if (rr < 0) rr += (int)mdist;
                      if (rr >= (int)mdist) rr -= (int)mdist;
                      const uint8_t bt = ring[(at - mdist + (uint32_t)rr) & M];
                      ring[(at + i) & M] = bt;
                    }
                  }
                }
              }
#ifdef VSF_PNG_STATS
              PNG_STAT(0, 1);
              PNG_STAT(1, total);
              PNG_STAT(2, end);
              {
                int nt = 0, nm = 0;
                for (int k = 0; k < kSpanRegs; k++) {
                  nt += __popcll(active[k]);
                  nm += __popcll(active[k] & __builtin_amdgcn_ballot_w64(match_mask[k] != 0));
                }
                PNG_STAT(3, nt);
                PNG_STAT(4, nm);
                PNG_STAT(5, end < kSpanBits ? 1 : 0);
              }
#endif
              pos += total;
              room -= (int)total;
              budget -= end;
              rel += end;
              flush_chunks();
            }
            PNG_STAT(6, 1);
            br.restart_at(rel);
          }
          PNG_STAT(8, 1);
          br.fill();
          if (budget < 64) break;  // (64 bits: more than a length code, a distance code and their extra bits)
          // one symbol: the next ten bits look it up directly; a longer code (entry 0) is decoded by the lanes
          uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)lit_table[br.peek() & ((1u << kLitBits) - 1u)]);
          int L = (int)(e & 15u);
          if (L == 0) {
            const uint32_t idx = decode_index(br.peek(), lc, lane, &L);
            if (L == 0) break;
            e = table_entry(lit_symbol(idx), L, false);
          }
          if ((int)e < 0) break;  // the end of the block, or a symbol that must not occur: the walk below
          if ((e & kMatchBit) == 0u) {
            br.acc >>= L;
            br.n -= L;
            budget -= L;
            const uint32_t sym = (e >> 8) & 0xFFu;
            asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(pend) : "s"(sym), "s"(npend));
            ++npend;
            if (--room == 0) break;
            if (npend == 64) {  // the 64 waiting literals enter the ring
              ring[(pos + lane) & M] = (uint8_t)pend;
              pos += 64u;
              npend = 0;
              if (pos - flushed >= (uint32_t)kFlushChunk) break;
            }
            continue;
          }
          // a match: length code (+ extra bits), distance code (+ extra bits)
          br.acc >>= L;
          br.n -= L;
          budget -= L;
          const int eb = (int)((e >> 4) & 15u);
          const uint32_t len = ((e >> 8) & 0xFFFFu) + ((uint32_t)br.acc & ((1u << eb) - 1u));
          br.acc >>= eb;
          br.n -= eb;
          budget -= eb;
          br.fill();
          uint32_t e2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dist_table[br.peek() & ((1u << kDistBits) - 1u)]);
          int L2 = (int)(e2 & 15u);
          if (L2 == 0) {
            const uint32_t idx2 = decode_index(br.peek(), dc, lane, &L2);
            e2 = table_entry((uint32_t)__builtin_amdgcn_readlane((int)dsy, (int)(idx2 & 31u)), L2, true);
          }
          if (L2 == 0 || (int)e2 < 0) {  // "invalid distance code" (the bits were there: budget)
            bad = true;
            break;
          }
          br.acc >>= L2;
          br.n -= L2;
          budget -= L2;
          const int eb2 = (int)((e2 >> 4) & 15u);
          const uint32_t dist = ((e2 >> 8) & 0xFFFFu) + ((uint32_t)br.acc & ((1u << eb2) - 1u));
          br.acc >>= eb2;
          br.n -= eb2;
          budget -= eb2;
          if (npend > 0) {  // the waiting literals come first
            if (lane < npend) ring[(pos + lane) & M] = (uint8_t)pend;
            pos += (uint32_t)npend;
            npend = 0;
          }
          if (dist > pos) {  // "invalid distance too far back"
            bad = true;
            break;
          }
          const uint32_t want = min(len, (uint32_t)room);
          if (dist >= want) {
            for (uint32_t i = lane; i < want; i += 64) {
              const uint8_t b = ring[(pos - dist + i) & M];
              ring[(pos + i) & M] = b;
            }
          } else {  // the match overlaps itself: byte i repeats byte i mod dist (all of them in front of its start)
            const float rcp = 1.0f / (float)dist;
            for (uint32_t i = lane; i < want; i += 64) {
              const uint32_t q = (uint32_t)((float)i * rcp);
              int rr = (int)i - (int)(q * dist);
              if (rr < 0) rr += (int)dist;
              if (rr >= (int)dist) rr -= (int)dist;
              const uint8_t b = ring[(pos - dist + (uint32_t)rr) & M];
              ring[(pos + i) & M] = b;
            }
          }
          pos += want;
          room -= (int)want;
          if (want < len) {  // the image is complete in the middle of the match: inflate leaves, the drain takes the rest
            carry = len - want;
            break;
          }
          if (room == 0 || pos - flushed >= (uint32_t)kFlushChunk) break;
        }
        br.left -= (int64_t)(budget0 - budget);
        flush_chunks();
        if (stop || bad) break;
        if (carry != 0u) {
          enter_tail();
          begin_drain(false);
          if (stop) break;
          extra += carry;
          carry = 0u;
          continue;
        }
        if (pos + (uint32_t)npend >= expected) continue;  // (the image is complete: the walk goes on as zlib's does)
      }
      br.fill();
      int L;
      const uint32_t idx = decode_index(br.peek(), lc, lane, &L);
      if (L == 0) {  // no code starts like this (possible only in a code of one symbol)
        if (!lacks(1)) fail();
        break;
      }
      if (lacks(L)) break;
      br.drop(L);
      const uint32_t sym = lit_symbol(idx);
      if (sym < 256u) {
        if (tail) {  // a literal and no room for it: inflate leaves, the drain counts it
          if (!drain) begin_drain(false);
          if (stop) break;
          extra += 1u;
          continue;
        }
        // (lane select through M0: the value already takes the instruction's one constant-bus slot)
        asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(pend) : "s"(sym), "s"(npend));
        if (++npend == 64) flush_literals();
        continue;
      }
      if (sym == 256u) break;  // end of block
      const uint32_t l = sym - 257u;
      if (l > 28u) {  // symbols 286, 287: "invalid literal/length code"
        fail();
        break;
      }
      uint32_t len;
      if (l < 8u) {
        len = 3u + l;
      } else if (l == 28u) {
        len = 258u;
      } else {
        const int eb = (int)(l >> 2) - 1;
        if (lacks(eb)) break;
        len = 3u + ((4u + (l & 3u)) << eb) + br.take(eb);
      }
      br.fill();
      int L2;
      const uint32_t idx2 = decode_index(br.peek(), dc, lane, &L2);
      if (L2 == 0) {  // "invalid distance code" (a block without distance codes, or a code of one symbol)
        if (!lacks(1)) fail();
        break;
      }
      if (lacks(L2)) break;
      br.drop(L2);
      const uint32_t dsym = (uint32_t)__builtin_amdgcn_readlane((int)dsy, (int)(idx2 & 31u));
      if (dsym > 29u) {  // "invalid distance code"
        fail();
        break;
      }
      uint32_t dist;
      if (dsym < 4u) {
        dist = 1u + dsym;
      } else {
        const int eb = (int)(dsym >> 1) - 1;
        if (lacks(eb)) break;
        dist = 1u + ((2u + (dsym & 1u)) << eb) + br.take(eb);
      }
      if (tail) {  // a match and no room for it: inflate leaves (it checks the distance only when it copies): the drain's
        if (!drain) begin_drain(false);
        if (stop) break;
        if (dist > pos + extra) {  // "invalid distance too far back", a warning there
          stop = true;
          break;
        }
        extra += len;
        continue;
      }
      flush_literals();
      if (dist > pos) {  // "invalid distance too far back"
        bad = true;
        break;
      }
      const uint32_t want = min(len, expected - pos);
      const float rcp = 1.0f / (float)dist;
      for (uint32_t i0 = 0; i0 < want; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint32_t off = i;
        if (dist < want) {  // the match overlaps itself: byte i repeats byte i mod dist (all of them in front of its start)
          const uint32_t q = (uint32_t)((float)i * rcp);
          int rr = (int)i - (int)(q * dist);
          if (rr < 0) rr += (int)dist;
          if (rr >= (int)dist) rr -= (int)dist;
          off = (uint32_t)rr;
        }
        if (i < want) {
          const uint8_t b = ring[(pos - dist + off) & M];
          ring[(pos + i) & M] = b;
        }
      }
      pos += want;
      flush_chunks();
      if (want < len) {  // the image is complete in the middle of the match: inflate leaves, the drain takes the rest
        enter_tail();
        begin_drain(false);
        if (stop) break;
        extra += len - want;
      }
    }
  }
  flush_literals();
  if (!bad && pos < expected) bad = true;  // "Not enough image data"
  // what is left in the ring (less than a flush chunk)
  if (pos > flushed) ring_chunk(ring_lds, out, flushed, pos - flushed, adler_a, adler_b, true);
  if (lane == 0) {
    a.file_status[image] = bad ? 1 : 0;
    if (bad) atomicOr(a.status, 2);
  }
}

// Filters (PNG specification 9.2), one wave per image, rows r0 .. r0 + 63 as a wavefront.
// This one: gray and gray + alpha files, not interlaced -- what a camera driver writes.
__global__ __launch_bounds__(64) void png_unfilter_kernel(PngArgs a) {
  const int lane = threadIdx.x;
  const int image = blockIdx.x;
  if (in_constant(a.file_status)[image] != 0) return;  // (written by the kernel in front on the same stream)
  const DevImage im = load_image(a, image);
  if (im.kind != kGray || (im.flags & kAdam7) != 0) return;  // (the kernels below)
  uint8_t* filt = a.filtered + (size_t)image * a.filtered_stride;
  uint8_t* dst = a.dst + (size_t)image * a.dst_image_stride;
  const int bpp = im.bpp, depth = im.depth;
  const int rb1 = (int)im.row_bytes + 1;
  const int nb = (int)im.row_bytes / bpp;  // bytes of the plane kept: one per pixel (depth >= 8), or the packed row
  const int w = a.width, h = a.height;
  bool bad = false;
  for (int r0 = 0; r0 < h; r0 += 64) {
    const int row = r0 + lane;
    const bool rowok = row < h;
    uint8_t* frow = filt + (size_t)(rowok ? row : 0) * rb1;
    const uint8_t* above_row = filt + (size_t)(r0 > 0 ? r0 - 1 : 0) * rb1;  // lane 0's "above": reconstructed in place by lane 63
    const int ftype = rowok ? frow[0] : 0;
    if (ftype > 4) bad = true;
    uint32_t ra = 0, rc = 0, last = 0;
    for (int s0 = 0; s0 < nb + 63; s0 += 8) {
      uint32_t F[8], U[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = s0 + u - lane;
        const bool act = rowok && j >= 0 && j < nb;
        F[u] = act ? frow[1 + j * bpp] : 0u;
        U[u] = (act && lane == 0 && r0 > 0) ? above_row[1 + j * bpp] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = s0 + u - lane;
        const bool act = rowok && j >= 0 && j < nb;
        uint32_t rbv = wave_shr1(last);
        if (lane == 0) rbv = U[u];
        const int ia = (int)ra, ib = (int)rbv, ic = (int)rc;
        const int pa = abs(ib - ic), pb = abs(ia - ic), pc = abs(ia + ib - 2 * ic);
        const int paeth = (pa <= pb && pa <= pc) ? ia : (pb <= pc ? ib : ic);
        const int pred = ftype == 1 ? ia : ftype == 2 ? ib : ftype == 3 ? ((ia + ib) >> 1) : ftype == 4 ? paeth : 0;
        const uint32_t R = (F[u] + (uint32_t)pred) & 255u;
        rc = rbv;
        ra = act ? R : 0u;
        last = act ? R : 0u;
        if (act) {
          if (lane == 63) frow[1 + j * bpp] = (uint8_t)R;  // the next 64 rows' "above"
          if (depth >= 8) {
            dst[(size_t)row * a.dst_pitch + j] = (uint8_t)R;
          } else {  // 1, 2, 4 bits: the samples of a byte, most significant first, replicated to 8 bits
            const int ppb = 8 / depth;
            const uint32_t mask = (1u << depth) - 1u, mul = 255u / mask;
            for (int p = 0; p < ppb; p++) {
              const int x = j * ppb + p;
              if (x < w) dst[(size_t)row * a.dst_pitch + x] = (uint8_t)(((R >> (8 - depth * (p + 1))) & mask) * mul);
            }
          }
        }
      }
    }
    __threadfence();  // lane 63's row is the next group's lane 0's "above"
  }
  if (bad) atomicOr(a.status, 2);  // "bad adaptive filter value": libpng stops with png_error; the image is not to be used
}

// The same for palette files (the index goes through the file's 256 gray values) and interlaced files of either kind: seven
// passes, each a small image with filter bytes of its own, written to their places (Adam7).
__global__ __launch_bounds__(64) void png_unfilter_general_kernel(PngArgs a) {
  const int lane = threadIdx.x;
  const int image = blockIdx.x;
  if (in_constant(a.file_status)[image] != 0) return;  // (written by the kernel in front on the same stream)
  const DevImage im = load_image(a, image);
  if (im.kind == kRgb8 || im.kind == kRgb16) return;       // (png_unfilter_rgb_kernel's)
  if (im.kind == kGray && (im.flags & kAdam7) == 0) return;  // (png_unfilter_kernel's)
  const uint8_t* lut = im.kind == kPalette ? a.blob + a.off_tables + im.table : nullptr;  // palette index -> gray value
  uint8_t* filt = a.filtered + (size_t)image * a.filtered_stride;
  uint8_t* dst = a.dst + (size_t)image * a.dst_image_stride;
  const int bpp = im.bpp, depth = im.depth;
  const int bits = depth * (im.kind == kPalette ? 1 : bpp >= 2 && depth >= 8 ? bpp * 8 / depth : 1);  // bits per pixel in the file
  bool bad = false;
  // An interlaced file is seven small images one after the other (a pass without pixels has no bytes); the others one.
  const int passes = (im.flags & kAdam7) ? 7 : 1;
  for (int pass = 0; pass < passes; pass++) {
    const Adam7Pass g = (im.flags & kAdam7) ? adam7_pass(pass) : Adam7Pass{0, 0, 1, 1};
    const int w = a.width > g.x0 ? (a.width - g.x0 + g.dx - 1) / g.dx : 0, h = a.height > g.y0 ? (a.height - g.y0 + g.dy - 1) / g.dy : 0;
    if (w == 0 || h == 0) continue;
    const int row_bytes = (w * bits + 7) / 8;
    const int rb1 = row_bytes + 1;
    const int nb = row_bytes / bpp;  // bytes of the plane kept: one per pixel (depth >= 8), or the packed row
    for (int r0 = 0; r0 < h; r0 += 64) {
      const int row = r0 + lane;
      const bool rowok = row < h;
      uint8_t* frow = filt + (size_t)(rowok ? row : 0) * rb1;
      const uint8_t* above_row = filt + (size_t)(r0 > 0 ? r0 - 1 : 0) * rb1;  // lane 0's "above": reconstructed in place by lane 63
      const int ftype = rowok ? frow[0] : 0;
      if (ftype > 4) bad = true;
      uint8_t* drow = dst + (size_t)(g.y0 + row * g.dy) * a.dst_pitch + g.x0;
      uint32_t ra = 0, rc = 0, last = 0;
      for (int s0 = 0; s0 < nb + 63; s0 += 8) {
        uint32_t F[8], U[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = s0 + u - lane;
          const bool act = rowok && j >= 0 && j < nb;
          F[u] = act ? frow[1 + j * bpp] : 0u;
          U[u] = (act && lane == 0 && r0 > 0) ? above_row[1 + j * bpp] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = s0 + u - lane;
          const bool act = rowok && j >= 0 && j < nb;
          uint32_t rbv = wave_shr1(last);
          if (lane == 0) rbv = U[u];
          const int ia = (int)ra, ib = (int)rbv, ic = (int)rc;
          const int pa = abs(ib - ic), pb = abs(ia - ic), pc = abs(ia + ib - 2 * ic);
          const int paeth = (pa <= pb && pa <= pc) ? ia : (pb <= pc ? ib : ic);
          const int pred = ftype == 1 ? ia : ftype == 2 ? ib : ftype == 3 ? ((ia + ib) >> 1) : ftype == 4 ? paeth : 0;
          const uint32_t R = (F[u] + (uint32_t)pred) & 255u;
          rc = rbv;
          ra = act ? R : 0u;
          last = act ? R : 0u;
          if (act) {
            if (lane == 63) frow[1 + j * bpp] = (uint8_t)R;  // the next 64 rows' "above"
            if (depth >= 8) {
              drow[j * g.dx] = lut ? lut[R] : (uint8_t)R;
            } else {  // 1, 2, 4 bits: the samples of a byte, most significant first -- replicated to 8 bits, or looked up
              const int ppb = 8 / depth;
              const uint32_t mask = (1u << depth) - 1u, mul = 255u / mask;
              for (int p = 0; p < ppb; p++) {
                const int x = j * ppb + p;
                const uint32_t v = (R >> (8 - depth * (p + 1))) & mask;
                if (x < w) drow[x * g.dx] = lut ? lut[v] : (uint8_t)(v * mul);
              }
            }
          }
        }
      }
      __threadfence();  // lane 63's row is the next group's lane 0's "above"
    }
    filt += (size_t)rb1 * h;
  }
  if (bad) atomicOr(a.status, 2);  // "bad adaptive filter value": libpng stops with png_error; the image is not to be used
}

// The same for colour files (types 2 and 6): every byte plane the gray value needs is reconstructed IN PLACE -- red, green,
// blue, both bytes of each for 16-bit samples; alpha is left as it is -- 64 rows at a time as above, one plane after the
// other; then the 64 rows are weighted pixel by pixel as libpng's png_do_rgb_to_gray weights them with the coefficients
// grfmt_png.cpp asks for (9797, 19234, 3737 of 32768): 8-bit samples truncated; 16-bit samples rounded, of which the read
// keeps the high byte (png_set_strip_16 comes behind); with the file's gamma tables (8-bit only: the host refuses the rest)
// gray = from_linear[(weighted sum of to_linear[r, g, b] + 16384) >> 15] unless r = g = b.
__global__ __launch_bounds__(64) void png_unfilter_rgb_kernel(PngArgs a) {
  const int lane = threadIdx.x;
  const int image = blockIdx.x;
  if (in_constant(a.file_status)[image] != 0) return;
  const DevImage im = load_image(a, image);
  if (im.kind != kRgb8 && im.kind != kRgb16) return;
  uint8_t* filt = a.filtered + (size_t)image * a.filtered_stride;
  uint8_t* dst = a.dst + (size_t)image * a.dst_image_stride;
  const int bpp = im.bpp;
  const int planes = im.kind == kRgb16 ? 6 : 3;
  const uint8_t* to_1 = (im.flags & kGammaTables) ? a.blob + a.off_tables + im.table : nullptr;
  const uint8_t* from_1 = to_1 ? to_1 + 256 : nullptr;
  bool bad = false;
  const int passes = (im.flags & kAdam7) ? 7 : 1;
  for (int pass = 0; pass < passes; pass++) {
    const Adam7Pass g = (im.flags & kAdam7) ? adam7_pass(pass) : Adam7Pass{0, 0, 1, 1};
    const int w = a.width > g.x0 ? (a.width - g.x0 + g.dx - 1) / g.dx : 0, h = a.height > g.y0 ? (a.height - g.y0 + g.dy - 1) / g.dy : 0;
    if (w == 0 || h == 0) continue;
    const int rb1 = w * bpp + 1;
    for (int r0 = 0; r0 < h; r0 += 64) {
      const int row = r0 + lane;
      const bool rowok = row < h;
      uint8_t* frow = filt + (size_t)(rowok ? row : 0) * rb1;
      const uint8_t* above_row = filt + (size_t)(r0 > 0 ? r0 - 1 : 0) * rb1;  // lane 0's "above": reconstructed in place by lane 63
      const int ftype = rowok ? frow[0] : 0;
      if (ftype > 4) bad = true;
      for (int plane = 0; plane < planes; plane++) {
        uint32_t ra = 0, rc = 0, last = 0;
        for (int s0 = 0; s0 < w + 63; s0 += 8) {
          uint32_t F[8], U[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int j = s0 + u - lane;
            const bool act = rowok && j >= 0 && j < w;
            F[u] = act ? frow[1 + j * bpp + plane] : 0u;
            U[u] = (act && lane == 0 && r0 > 0) ? above_row[1 + j * bpp + plane] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int j = s0 + u - lane;
            const bool act = rowok && j >= 0 && j < w;
            uint32_t rbv = wave_shr1(last);
            if (lane == 0) rbv = U[u];
            const int ia = (int)ra, ib = (int)rbv, ic = (int)rc;
            const int pa = abs(ib - ic), pb = abs(ia - ic), pc = abs(ia + ib - 2 * ic);
            const int paeth = (pa <= pb && pa <= pc) ? ia : (pb <= pc ? ib : ic);
            const int pred = ftype == 1 ? ia : ftype == 2 ? ib : ftype == 3 ? ((ia + ib) >> 1) : ftype == 4 ? paeth : 0;
            const uint32_t R = (F[u] + (uint32_t)pred) & 255u;
            rc = rbv;
            ra = act ? R : 0u;
            last = act ? R : 0u;
            if (act) frow[1 + j * bpp + plane] = (uint8_t)R;
          }
        }
      }
      __threadfence();  // the rows are read back below by other lanes, lane 63's by the next group's lane 0
      const int rows = min(64, h - r0);
      for (int rr = 0; rr < rows; rr++) {
        const uint8_t* src = filt + (size_t)(r0 + rr) * rb1 + 1;
        uint8_t* out = dst + (size_t)(g.y0 + (r0 + rr) * g.dy) * a.dst_pitch + g.x0;
        for (int x = lane; x < w; x += 64) {
          const uint8_t* px = src + (size_t)x * bpp;
          uint32_t gray;
          if (planes == 3) {
            const uint32_t r = px[0], gr = px[1], b = px[2];
            if (to_1 == nullptr)
              gray = (9797u * r + 19234u * gr + 3737u * b) >> 15;
            else if (r == gr && r == b)
              gray = r;
            else
              gray = from_1[(9797u * to_1[r] + 19234u * to_1[gr] + 3737u * to_1[b] + 16384u) >> 15];
          } else {
            const uint32_t r = ((uint32_t)px[0] << 8) | px[1], gr = ((uint32_t)px[2] << 8) | px[3], b = ((uint32_t)px[4] << 8) | px[5];
            gray = ((9797u * r + 19234u * gr + 3737u * b + 16384u) >> 15) >> 8;
          }
          out[x * g.dx] = (uint8_t)gray;
        }
      }
    }
    filt += (size_t)rb1 * h;
  }
  if (bad) atomicOr(a.status, 2);  // "bad adaptive filter value"
}

}  // namespace

void vsf_launch_png_decode(const uint8_t* d_blob, size_t off_images, size_t off_pieces, size_t off_tables, size_t off_stream, int n, int width, int height,
                           uint8_t* d_filtered, size_t filtered_stride, int32_t* d_file_status, uint8_t* d_dst,
                           size_t dst_image_stride, int dst_pitch, int32_t* d_status, bool any_general, bool any_rgb, hipStream_t s) {
  PngArgs a;
  a.blob = d_blob;
  a.off_images = off_images;
  a.off_pieces = off_pieces;
  a.off_tables = off_tables;
  a.off_stream = off_stream;
  a.filtered = d_filtered;
  a.filtered_stride = filtered_stride;
  a.width = width;
  a.height = height;
  a.dst = d_dst;
  a.dst_image_stride = dst_image_stride;
  a.dst_pitch = dst_pitch;
  a.status = d_status;
  a.file_status = d_file_status;
  hipLaunchKernelGGL(png_inflate_kernel, dim3(n), dim3(64), 0, s, a);
  hipLaunchKernelGGL(png_unfilter_kernel, dim3(n), dim3(64), 0, s, a);
  if (any_general) hipLaunchKernelGGL(png_unfilter_general_kernel, dim3(n), dim3(64), 0, s, a);
  if (any_rgb) hipLaunchKernelGGL(png_unfilter_rgb_kernel, dim3(n), dim3(64), 0, s, a);
}

#ifdef VSF_PNG_STATS
extern "C" int vsf_debug_png_stats(unsigned long long* out16, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_png_stats), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_png_stats), 16 * sizeof(unsigned long long));
}
#endif
