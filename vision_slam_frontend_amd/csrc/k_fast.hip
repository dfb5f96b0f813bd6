// k_fast.hip -- K2 (+ the border half of K3): FAST-9/16 segment test, cornerScore, 3x3 non-max suppression and
// raster-order compaction, for every level of every image in ONE launch.
//
// Restates cv::FAST_t<16> and cornerScore<16> (features2d/fast.cpp, fast_score.cpp) as called by
// ORB's computeKeyPoints (threshold 20; slam_frontend.cc:274, parameters :205-213) and by
// FastFeatureDetector::detect (threshold 10, slam_frontend.cc:191,271), plus
// KeyPointsFilter::runByImageBorder(31) which is folded into the evaluated rectangle.
//
// Streaming march kernel -- no LDS, no barriers, no divergent corner/score phases:
//  * a wave owns a band of 248 keypoint columns x a strip of 32 rows of one level of one image and walks down the
//    rows; each lane holds 4 adjacent pixels per row (one coalesced 32-bit load) in a 7-row register window (nine
//    rotating register sets incl. two rows of read-ahead, no copies) and gets its neighbours' dwords by DPP wave shifts;
//  * corner test and score are ONE dense computation: with d_k = p_k - v on the 16-pixel circle,
//      A = max over the 16 arcs of min(d over the 9-arc),  B = -min over arcs of max(d over the arc)
//    (computed on the p_k, v subtracted once at the end),
//    the pixel is a FAST-9 corner iff max(A, B) > t and cornerScore is max(A, B) - 1.  Two pixels are processed per
//    VALU lane-op on packed 16-bit halves; circle bytes are pulled out of the window with v_perm_b32.  Arcs 2j and 2j + 1
//    share eight pixels, so max(arc 2j, arc 2j + 1) = min(octet, max(d[2j], d[2j + 9])) (min / max distribute): an octet
//    is two quads of two pixel pairs -> 8 + 8 + 8 two-input ops, 8 THREE-input ops and a 4-op reduction = 36 packed ops
//    per polarity (59 with prefix / suffix minima of the circle halves, 80 naively).  The three-input ops are gfx950's
//    v_pk_minimum3_f16 / v_pk_maximum3_f16 applied to the pixel values as they sit in the halves: 0..255 are f16
//    denormals, ordered like the integers and not flushed in the default mode (tools/exp/m3.hip: all 2^24 triples, full
//    issue rate).  With 1.86 G wave-instructions per 512-image launch the kernel issues at 93 % of the vector ALU's rate:
//    the floor of a dense score;
//  * the only branch is wave-wide: a row is skipped when v_sad_u8 of every lane's 4 pixels against the rows 3 above
//    and 3 below stays <= t (circle pixels 0 and 8: every 9-arc contains one of them);
//  * scores stay in registers; the strict 8-neighbour NMS is packed too (row-wise 3-maxima shared between the rows
//    above and below), and survivors are appended in raster order with ballot prefix ranks.
// Output per unit: a candidate segment in unit-local raster order + the start offset of every row, which
// vsf_gather.h merges into the level's global raster order.
#include "vsf_gather.h"
#include <cstdlib>

#include "vsf_internal.h"

namespace {

typedef short v2s __attribute__((ext_vector_type(2)));
constexpr int SR = VSF_FAST_STRIP_ROWS;

struct FastArgs {
  const VsfLevel* levels;
  const uint32_t* units;
  int nunits;  // cells (strip x band) per image: stride of the row-start tables
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;
  uint32_t pyr_bytes;
  uint32_t* cand;
  uint32_t cand_entries;
  uint16_t* rowstart;
  int threshold;
  int nms;
};

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {  // lane i <- lane i-1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) {  // lane i <- lane i+1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
}

// *p, read by every lane and declared the same in all of them (v_readfirstlane per dword)
template <class T>
__device__ __forceinline__ T uniform_copy(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dwords");
  T out;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(p);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (size_t i = 0; i < sizeof(T) / 4; i++) dst[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)src[i]);
  return out;
}

// One image row as a lane holds it while the row is inside the 7-row window.  The lane's 4 pixels are bytes 4..7 of the
// 12-byte run {previous lane's dword, own dword, next lane's dword}; b<B> is the pair (byte B, byte B + 2) zero-extended
// into the two 16-bit halves -- the layout the score works on: pixel pairs (x, x + 2) and (x + 1, x + 3), so that the
// second pair's circle pixel at column offset dx IS the first pair's at dx + 1.  Over its seven steps in the window a row
// is asked for exactly the eight pairs b1 .. b8 (as row y +- 3: b3..b6, y +- 2: b2, b3, b6, b7, y +- 1 and y: b1, b2, b7, b8
// and, as the centre row, b4 / b5 themselves), so they are made ONCE when the row enters -- 3 mask / shift operations, 4
// DPP moves and 4 v_alignbit -- instead of one v_perm_b32 per use (34 per step in round 3's kernel: profiles/r04/).
struct RowP {
  uint32_t d;                            // the raw dword (row pre-check)
  v2s b1, b2, b3, b4, b5, b6, b7, b8;
};

__device__ __forceinline__ v2s as_v2s(uint32_t v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ void prep_row(RowP& r) {
  const uint32_t e = r.d & 0x00FF00FFu, o = (r.d >> 8) & 0x00FF00FFu;  // (byte 4, byte 6), (byte 5, byte 7)
  const uint32_t ep = wave_shr1(e), op = wave_shr1(o);                 // (0, 2), (1, 3)
  const uint32_t en = wave_shl1(e), on = wave_shl1(o);                 // (8, 10), (9, 11)
  r.b1 = as_v2s(op);
  r.b2 = as_v2s(__builtin_amdgcn_alignbit(e, ep, 16));                 // (2, 4)
  r.b3 = as_v2s(__builtin_amdgcn_alignbit(o, op, 16));                 // (3, 5)
  r.b4 = as_v2s(e);
  r.b5 = as_v2s(o);
  r.b6 = as_v2s(__builtin_amdgcn_alignbit(en, e, 16));                 // (6, 8)
  r.b7 = as_v2s(__builtin_amdgcn_alignbit(on, o, 16));                 // (7, 9)
  r.b8 = as_v2s(en);
}

__device__ __forceinline__ v2s vmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s vmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
// Three-input packed minimum / maximum of values 0..255 held in 16-bit halves (see score_pair).
__device__ __forceinline__ v2s vmin3(v2s a, v2s b, v2s c) {
  v2s r;
  asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ v2s vmax3(v2s a, v2s b, v2s c) {
  v2s r;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max(a - b, 0) per half (v_pk_sub_u16 with clamp): a, b in 0 .. 255
__device__ __forceinline__ v2s vsubsat(v2s a, v2s b) {
  typedef unsigned short v2us __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(v2s, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b)));
}

// Scores of two pixels of the centre row R3 from the 16 circle pixel pairs d[k] and the centre pair v.
// Returns, per 16-bit half, max(cornerScore + 1 - offs, 0) with offs = max(t, 1): zero for a pixel that is no corner (or,
// at t = 0, a corner of score 0, which can never win the strict suppression: cv::FAST_t compares against neighbours >= 0),
// cornerScore - (offs - 1) >= 1 otherwise -- an order-preserving offset, so the suppression compares these values as it
// would the scores and the emitter adds offs - 1 back.  (NMS == false: a marker 1 for every corner, as cv::FAST_t leaves
// the response at 0 then; `toff` = t in that instance.)
template <bool NMS>
__device__ __forceinline__ v2s score_from_circle(const v2s (&d)[16], v2s v, v2s toff) {
  // Arc k = d[k .. k+8] (indices mod 16).  Arcs 2j and 2j+1 share the eight pixels C = d[2j+1 .. 2j+8]:
  //   max(min(d[2j], C), min(C, d[2j+9])) = min(C, max(d[2j], d[2j+9]))                       (distributive lattice)
  // and C is four pixel pairs E_i = min(d[2i+1], d[2i+2]), i = j .. j+3, i.e. two quads F_i = min(E_i, E_{i+1}):
  //   A = max_j min3(F_j, F_{j+2}, max(d[2j], d[2j+9])),     Bm likewise with min and max exchanged.
  // 8 + 8 + 8 two-input ops, 8 three-input ops and a 4-op reduction = 36 per polarity (59 with prefix / suffix minima of
  // the two circle halves).  The three-input ops are gfx950's v_pk_minimum3_f16 / v_pk_maximum3_f16 on the pixel values
  // as they sit in the 16-bit halves: 0..255 are f16 denormals, whose order is the integer order (f16 denormals are
  // never flushed in the default mode; exhaustively checked on MI355X for all 2^24 triples, tools/exp/m3.hip).
  v2s E[8], G[8], F[8], H[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    E[j] = vmin(d[2 * j + 1], d[(2 * j + 2) & 15]);
    G[j] = vmax(d[2 * j + 1], d[(2 * j + 2) & 15]);
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    F[j] = vmin(E[j], E[(j + 1) & 7]);
    H[j] = vmax(G[j], G[(j + 1) & 7]);
  }
  v2s ta[8], tb[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    ta[j] = vmin3(F[j], F[(j + 2) & 7], vmax(d[2 * j], d[(2 * j + 9) & 15]));
    tb[j] = vmax3(H[j], H[(j + 2) & 7], vmin(d[2 * j], d[(2 * j + 9) & 15]));
  }
  const v2s A = vmax(vmax3(vmax3(ta[0], ta[1], ta[2]), vmax3(ta[3], ta[4], ta[5]), ta[6]), ta[7]);
  const v2s Bm = vmin(vmin3(vmin3(tb[0], tb[1], tb[2]), vmin3(tb[3], tb[4], tb[5]), tb[6]), tb[7]);
  // cornerScore<16> = max(t, A - v, v - Bm) - 1; a corner iff max(A - v, v - Bm) > t.  Saturating subtractions: 4 ops.
  const v2s sc = vsubsat(vmax(vsubsat(A, v), vsubsat(v, Bm)), toff);
  const v2s one = {1, 1};
  return NMS ? sc : vmin(sc, one);
}

// Pixels 0 and 2 (J0 = 0) or 1 and 3 (J0 = 1) of the lane; R0..R6 are rows y-3 .. y+3.
// Bresenham circle, OpenCV order: (dx,dy) = (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)
// (-3,0)(-3,1)(-2,2)(-1,3); window row = R[3 + dy]; pair index = 4 + J0 + dx.
template <bool NMS>
__device__ __forceinline__ v2s score_pair0(const RowP& R0, const RowP& R1, const RowP& R2, const RowP& R3, const RowP& R4,
                                           const RowP& R5, const RowP& R6, v2s toff) {
  const v2s d[16] = {R6.b4, R6.b5, R5.b6, R4.b7, R3.b7, R2.b7, R1.b6, R0.b5, R0.b4, R0.b3, R1.b2, R2.b1, R3.b1, R4.b1, R5.b2, R6.b3};
  return score_from_circle<NMS>(d, R3.b4, toff);
}
template <bool NMS>
__device__ __forceinline__ v2s score_pair1(const RowP& R0, const RowP& R1, const RowP& R2, const RowP& R3, const RowP& R4,
                                           const RowP& R5, const RowP& R6, v2s toff) {
  const v2s d[16] = {R6.b5, R6.b6, R5.b7, R4.b8, R3.b8, R2.b8, R1.b7, R0.b6, R0.b5, R0.b4, R1.b3, R2.b2, R3.b2, R4.b2, R5.b3, R6.b4};
  return score_from_circle<NMS>(d, R3.b5, toff);
}

// One score row as the NMS needs it, in the same pair layout: s02 = (s0, s2), s13 = (s1, s3) of the lane's pixels and
// the horizontal maxima of their neighbours.
struct ScoreRow {
  v2s s02, s13;
  v2s h3a, h3b;     // max(s[x-1], s[x], s[x+1]) for pixels (0, 2) and (1, 3)
  v2s h2a, h2b;     // max(s[x-1], s[x+1])
};

__device__ __forceinline__ ScoreRow make_score_row(v2s s02, v2s s13) {
  ScoreRow o;
  o.s02 = s02;
  o.s13 = s13;
  // left / right neighbours of pixels (0, 2) are (prev lane's s3, s1) and (s1, s3); of pixels (1, 3): (s0, s2) and (s2, next s0)
  const v2s x = as_v2s(__builtin_amdgcn_alignbit(as_u32(s13), wave_shr1(as_u32(s13)), 16));
  const v2s y = as_v2s(__builtin_amdgcn_alignbit(wave_shl1(as_u32(s02)), as_u32(s02), 16));
  o.h2a = vmax(x, s13);
  o.h2b = vmax(s02, y);
  o.h3a = vmax3(x, s13, s02);
  o.h3b = vmax3(s02, y, s13);
  return o;
}

// HALF = false: the wave is one unit (cell) of 248 keypoint columns x 32 rows.
// HALF = true : the two 32-lane halves of the wave are two cells of the level's LAST, narrow band (<= 120 keypoint
//               columns = 30 lanes + 2 halo lanes) in two consecutive strips; each half keeps its own candidate
//               segment, row-start table and counter, so downstream nothing changes.  (The pyramid's 50 levels leave a
//               narrow remainder band almost everywhere: 510 -> 446 waves per 640x480 image.)
// NMS = false (standalone FAST without suppression only) keeps every corner and a zero response.
// (`work`: index into a.units, wave-uniform -- the work item and everything derived from it: level, band, strip, row
// addresses, is scalar)
template <bool HALF, bool NMS>
__device__ __forceinline__ void fast_march_body(const FastArgs& a, int work, int image) {
  const int lane = threadIdx.x & 63;
  // (explicitly wave-uniform: in the resident kernels' cell loop these loads follow the previous cell's stores, so the
  // compiler will not prove them invariant and use scalar loads; vector loads would make every row address, the buffer
  // descriptor included, a per-lane value)
  const uint32_t ud = uniform_copy(a.units + work);
  const int level = (int)(ud >> 24), band = (int)((ud >> 16) & 0xFF), strip0 = (int)(ud & 0x7FFF);
  const VsfLevel L = uniform_copy(a.levels + level);
  const uint8_t* src;
  int pitch;
  if (level == 0) {
    src = a.img0 + (size_t)image * a.img0_stride;
    pitch = a.img0_pitch;
  } else {
    src = a.pyr + (size_t)image * a.pyr_bytes + L.offset;
    pitch = L.pitch;
  }
  const int t = a.threshold;
  // score offset (see score_from_circle): the emitter adds offs - 1 back
  const int offs = NMS ? max(t, 1) : t;
  const v2s toff = {(short)offs, (short)offs};
  constexpr int HL = HALF ? 32 : 64;                        // lanes per cell
  const int half = HALF ? (lane >> 5) : 0, hl = lane & (HL - 1);
  const int strip = strip0 + half;
  const bool valid = strip < L.nstrips;                     // (the second half of the last odd strip has no cell)
  const int bx0 = L.fast_a0 + VSF_FAST_BAND_COLS * band;
  const int c0 = bx0 - 4 + 4 * hl;  // first column of this lane's 4 pixels (lane 1 starts the band)
  const bool loadable = c0 >= 0 && c0 + 3 < pitch;
  const int ys = L.y_lo + strip * SR;                       // per lane when HALF
  const int nrows = valid ? min(SR, L.y_hi - ys) : 0;       // rows of this lane's cell
  const int nrows0 = min(SR, L.y_hi - (L.y_lo + strip0 * SR));  // rows of the first cell (>= the second's): uniform
  // Pixels that may carry a score: FAST's 3-pixel rim and one column beyond the keypoint rectangle (for the NMS).
  const int sx_lo = max(L.x_lo - 1, 3), sx_hi = min(L.x_hi + 1, L.w - 3);
  uint32_t sm02 = 0, sm13 = 0, em02 = 0, em13 = 0;  // per-pixel 16-bit masks in the pair layout: may be scored / emitted
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int x = c0 + j;
    const uint32_t bit = 0xFFFFu << (16 * (j >> 1));
    // lanes 1 .. HL-2 score and emit; lane 0 scores its last pixel only, lane HL-1 its first (the columns next to the band)
    const bool sc_ok = ((hl >= 1 && hl <= HL - 2) || (hl == 0 && j == 3) || (hl == HL - 1 && j == 0)) && x >= sx_lo && x < sx_hi;
    const bool em_ok = hl >= 1 && hl <= HL - 2 && x >= L.x_lo && x < L.x_hi && x < bx0 + VSF_FAST_BAND_COLS;
    if (j & 1) {
      if (sc_ok) sm13 |= bit;
      if (em_ok) em13 |= bit;
    } else {
      if (sc_ok) sm02 |= bit;
      if (em_ok) em02 |= bit;
    }
  }
  if (!valid) sm02 = sm13 = em02 = em13 = 0;
  const int unit_local = (valid ? strip : strip0) * L.nbands + band;
  uint16_t* rs = a.rowstart + ((size_t)image * a.nunits + L.unit0 + unit_local) * VSF_FAST_RS_STRIDE;
  const int seg_cap = L.seg_cap, hrow = L.h;
  // Candidate stores go through a buffer descriptor over the cell's segment: an entry beyond the segment's capacity is
  // dropped by the range check of the store itself (no compare, no 64-bit address arithmetic per store).  HALF: one
  // descriptor from the first cell's segment to the end of the second's (L.nbands segments further), the second cell's
  // lanes add that distance -- and check their capacity themselves, since a first-cell overflow would still be in range.
  uint32_t* seg0 = a.cand + (size_t)image * a.cand_entries + L.cand_offset + (size_t)(strip0 * L.nbands + band) * L.seg_cap;
  const __amdgpu_buffer_rsrc_t seg_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(seg0, 0, (HALF ? L.nbands + 1 : 1) * seg_cap * 4, 0x00020000);
  const uint32_t seg_off = (HALF && half && valid) ? (uint32_t)(L.nbands * seg_cap * 4) : 0u;

  // buffer loads: a lane's column offset sits in a VGPR, the row offset in an SGPR (per lane when HALF); reads outside
  // the level return 0
  const __amdgpu_buffer_rsrc_t src_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, pitch * hrow, 0x00020000);
  // (a lane outside the row gets an offset beyond the buffer instead of a branch around the load: no divergent control
  // flow, so the row offset stays in an SGPR)
  const uint32_t col_off = loadable ? (uint32_t)c0 : 0xFFFFFFF0u;
  auto load_row = [&](int q) -> uint32_t {  // row ys - 1 + q of this lane's cell
    const int yc = min(max(ys - 1 + q, 0), hrow - 1);
    if (HALF)
      return __builtin_amdgcn_raw_buffer_load_b32(
          src_rsrc, loadable ? __umul24((uint32_t)yc, (uint32_t)pitch) + (uint32_t)c0 : 0xFFFFFFF0u, 0u, 0);  // (full-rate multiply)
    return __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, col_off, (uint32_t)(yc * pitch), 0);
  };

  const unsigned long long scorable = __builtin_amdgcn_ballot_w64((sm02 | sm13) != 0u);
  const v2s zero2 = {0, 0};
  ScoreRow S0 = make_score_row(zero2, zero2), S1 = S0, S2 = S0;  // score rows rotate through three sets: q-2, q-1, q
  int count_lo = 0, count_hi = 0;  // candidates emitted so far by the cell(s) (wave-uniform)
  uint32_t my_rs = 0;              // lane hl keeps rowstart[hl] of its cell
  // (cell row, first column, score offset) of an emitted candidate: score << 24 | y << 12 | x
  const uint32_t yx0 = ((uint32_t)ys << 12) + (uint32_t)c0 + ((uint32_t)(NMS ? offs - 1 : 0) << 24);

  // One step: scores of cell row q - 1 (image row ys - 1 + q) from the window R0..R6 = image rows ys - 4 + q .. ys + 2 + q,
  // then NMS + emission of cell row q - 2.
  auto step = [&](int q, const RowP& R0, const RowP& R1, const RowP& R2, const RowP& R3, const RowP& R4,
                  const RowP& R5, RowP& R6, const ScoreRow& S_up, const ScoreRow& S_mid, ScoreRow& S_dn) {
    prep_row(R6);  // the row that has just entered the window
    v2s s02 = zero2, s13 = zero2;
    const int sy = ys - 1 + q;
    const bool row_ok = sy >= 3 && sy < hrow - 3;  // (wave-uniform unless HALF)
    // a 9-arc contains circle pixel 0 (row sy+3) or 8 (row sy-3), both in the centre pixel's column
    const uint32_t far = max(__builtin_amdgcn_sad_u8(R0.d, R3.d, 0u), __builtin_amdgcn_sad_u8(R6.d, R3.d, 0u));
    // (one v_cmp; the lanes that may score at all are a ballot taken once, the row test is scalar unless HALF)
    const unsigned long long want = __builtin_amdgcn_ballot_w64(far > (uint32_t)t) & scorable &
                                    (HALF ? __builtin_amdgcn_ballot_w64(row_ok) : (row_ok ? ~0ull : 0ull));
    if (want != 0ull) {
      s02 = score_pair0<NMS>(R0, R1, R2, R3, R4, R5, R6, toff);  // pixels 0 and 2
      s13 = score_pair1<NMS>(R0, R1, R2, R3, R4, R5, R6, toff);  // pixels 1 and 3
      s02 = as_v2s(as_u32(s02) & ((!HALF || row_ok) ? sm02 : 0u));
      s13 = as_v2s(as_u32(s13) & ((!HALF || row_ok) ? sm13 : 0u));
    }
    S_dn = make_score_row(s02, s13);
    const int r = q - 2;  // cell row to emit
    if (r >= 0 && r < nrows0) {
      if (__any((as_u32(S_mid.s02) | as_u32(S_mid.s13)) != 0)) {
        // keep iff score > every 8-neighbour (strict); without NMS every marked corner is kept.  k02 / k13: the score (or
        // marker) of a survivor, 0 elsewhere.
        uint32_t k02, k13;
        if (NMS) {
          const v2s nb02 = vmax3(S_up.h3a, S_dn.h3a, S_mid.h2a), nb13 = vmax3(S_up.h3b, S_dn.h3b, S_mid.h2b);
          k02 = as_u32((nb02 - S_mid.s02) >> 15) & as_u32(S_mid.s02) & em02;  // (all ones where the score exceeds nb)
          k13 = as_u32((nb13 - S_mid.s13) >> 15) & as_u32(S_mid.s13) & em13;
        } else {
          k02 = as_u32(S_mid.s02) & em02;
          k13 = as_u32(S_mid.s13) & em13;
        }
        if (HALF && r >= nrows) k02 = k13 = 0;  // (the second cell may have fewer rows)
        // rank among the cell's lanes: v_mbcnt counts the set bits below this lane (two instructions per ballot, chained
        // through the accumulator); the upper cell of a half-wave pair subtracts the lower cell's bits (scalar)
        auto below = [](unsigned long long b, int acc) -> int {
          return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, (uint32_t)acc));
        };
        const uint32_t yx = yx0 + ((uint32_t)r << 12);
        if (NMS) {
          // Strict NMS: two neighbouring pixels cannot both survive, so at most one of pixels 0 / 1 and at most one of
          // pixels 2 / 3 -- slot A (the low halves) and slot B (the high halves), in raster order: two ballots / ranks /
          // stores instead of four.  kk = (score of slot A, score of slot B).
          const uint32_t kk = k02 | k13;
          const bool hasA = (kk & 0xFFFFu) != 0u, hasB = (kk >> 16) != 0u;
          const unsigned long long bA = __ballot(hasA), bB = __ballot(hasB);
          if ((bA | bB) != 0ull) {  // wave-uniform
            int pos = below(bB, below(bA, half ? count_hi : count_lo));
            if (HALF) pos -= half ? __popc((uint32_t)bA) + __popc((uint32_t)bB) : 0;
            // entry = (score + offs - 1) << 24 | y << 12 | x;  x = c0 + (0 or 1) for slot A, c0 + 2 + (0 or 1) for slot B
            const uint32_t eA = (kk << 24) + yx + ((k13 & 0xFFFFu) != 0u ? 1u : 0u);
            const uint32_t eB = ((kk & 0xFFFF0000u) << 8) + yx + 2u + ((k13 >> 16) != 0u ? 1u : 0u);
            const int posB = pos + (hasA ? 1 : 0);
            if (hasA && (!HALF || pos < seg_cap)) __builtin_amdgcn_raw_buffer_store_b32(eA, seg_rsrc, seg_off + 4u * (uint32_t)pos, 0, 0);
            if (hasB && (!HALF || posB < seg_cap)) __builtin_amdgcn_raw_buffer_store_b32(eB, seg_rsrc, seg_off + 4u * (uint32_t)posB, 0, 0);
            if (HALF) {
              count_lo += __popc((uint32_t)bA) + __popc((uint32_t)bB);
              count_hi += __popc((uint32_t)(bA >> 32)) + __popc((uint32_t)(bB >> 32));
            } else {
              count_lo += __popcll(bA) + __popcll(bB);
            }
          }
        } else {
          const bool k0 = (k02 & 0xFFFFu) != 0, k1 = (k13 & 0xFFFFu) != 0, k2 = (k02 >> 16) != 0, k3 = (k13 >> 16) != 0;
          const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1), b2 = __ballot(k2), b3 = __ballot(k3);
          if ((b0 | b1 | b2 | b3) != 0ull) {  // wave-uniform
            int pos = below(b3, below(b2, below(b1, below(b0, half ? count_hi : count_lo))));
            if (HALF) {
              const int lower = __popc((uint32_t)b0) + __popc((uint32_t)b1) + __popc((uint32_t)b2) + __popc((uint32_t)b3);
              pos -= half ? lower : 0;
            }
            // (without NMS cv::FAST_t leaves the response at 0)
            if (k0) {
              if (!HALF || pos < seg_cap) __builtin_amdgcn_raw_buffer_store_b32(yx, seg_rsrc, seg_off + 4u * (uint32_t)pos, 0, 0);
              ++pos;
            }
            if (k1) {
              if (!HALF || pos < seg_cap) __builtin_amdgcn_raw_buffer_store_b32(yx + 1, seg_rsrc, seg_off + 4u * (uint32_t)pos, 0, 0);
              ++pos;
            }
            if (k2) {
              if (!HALF || pos < seg_cap) __builtin_amdgcn_raw_buffer_store_b32(yx + 2, seg_rsrc, seg_off + 4u * (uint32_t)pos, 0, 0);
              ++pos;
            }
            if (k3) {
              if (!HALF || pos < seg_cap) __builtin_amdgcn_raw_buffer_store_b32(yx + 3, seg_rsrc, seg_off + 4u * (uint32_t)pos, 0, 0);
            }
            if (HALF) {
              count_lo += __popcll(b0 & 0xFFFFFFFFull) + __popcll(b1 & 0xFFFFFFFFull) + __popcll(b2 & 0xFFFFFFFFull) +
                          __popcll(b3 & 0xFFFFFFFFull);
              count_hi += __popcll(b0 >> 32) + __popcll(b1 >> 32) + __popcll(b2 >> 32) + __popcll(b3 >> 32);
            } else {
              count_lo += __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
            }
          }
        }
      }
      // rowstart[r + 1] of the cell(s): the running count goes into lane r + 1 of the cell (v_writelane: one instruction;
      // lane 0 keeps rowstart[0] = 0, rowstart[SR] is written from the final count)
      if (r + 1 < SR) {
        // (lane select through M0: the value already takes the instruction's one constant-bus slot)
        asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(my_rs) : "s"(min(count_lo, seg_cap)), "s"(r + 1));
        if (HALF)
          asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(my_rs) : "s"(min(count_hi, seg_cap)), "s"(32 + r + 1));
      }
    }
  };

  // NINE rotating register sets hold image rows: step q reads rows (ys - 4 + q) .. (ys + 2 + q) from sets q .. q + 6
  // (mod 9), set q + 7 already holds the next row's dword and the row after that is requested into set q + 8 -- the set
  // whose row the previous step used last.  The loop body is nine steps, a multiple of the three score-row sets' period, so
  // neither the image rows nor the score rows are ever copied between registers (a period of seven cost 17 v_mov per step).
  // q runs over 0 .. nrows0 + 1 (score rows ys - 1 .. ys + nrows0).
  RowP A0, A1, A2, A3, A4, A5, A6, A7, A8;
  A0.d = load_row(-3), A1.d = load_row(-2), A2.d = load_row(-1), A3.d = load_row(0), A4.d = load_row(1), A5.d = load_row(2),
  A6.d = load_row(3), A7.d = load_row(4);
  prep_row(A0), prep_row(A1), prep_row(A2), prep_row(A3), prep_row(A4), prep_row(A5);
  const int qe = nrows0 + 1;
#define VSF_FAST_STEP(j, r0, r1, r2, r3, r4, r5, r6, ld, su, sm, sd) \
  if (q + (j) > qe) break;                                             \
  ld.d = load_row(q + (j) + 5);                                        \
  step(q + (j), r0, r1, r2, r3, r4, r5, r6, su, sm, sd);
  for (int q = 0;; q += 9) {
    VSF_FAST_STEP(0, A0, A1, A2, A3, A4, A5, A6, A8, S0, S1, S2)
    VSF_FAST_STEP(1, A1, A2, A3, A4, A5, A6, A7, A0, S1, S2, S0)
    VSF_FAST_STEP(2, A2, A3, A4, A5, A6, A7, A8, A1, S2, S0, S1)
    VSF_FAST_STEP(3, A3, A4, A5, A6, A7, A8, A0, A2, S0, S1, S2)
    VSF_FAST_STEP(4, A4, A5, A6, A7, A8, A0, A1, A3, S1, S2, S0)
    VSF_FAST_STEP(5, A5, A6, A7, A8, A0, A1, A2, A4, S2, S0, S1)
    VSF_FAST_STEP(6, A6, A7, A8, A0, A1, A2, A3, A5, S0, S1, S2)
    VSF_FAST_STEP(7, A7, A8, A0, A1, A2, A3, A4, A6, S1, S2, S0)
    VSF_FAST_STEP(8, A8, A0, A1, A2, A3, A4, A5, A7, S2, S0, S1)
  }
#undef VSF_FAST_STEP
  if (valid) {
    if (hl < SR) rs[hl] = (uint16_t)my_rs;                                          // rowstart[0 .. SR-1]
    if (hl == 0) rs[SR] = (uint16_t)min(half ? count_hi : count_lo, seg_cap);       // rowstart[SR] = cell total
  }
}

template <bool HALF, bool NMS>
__global__ __launch_bounds__(256) void fast_march_kernel(FastArgs a, int work0, int nwork) {
  const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (item >= nwork) return;  // wave-uniform
  fast_march_body<HALF, NMS>(a, work0 + item, blockIdx.y);
}

// Full cells and the half-wave cells of the narrow last bands in ONE launch, for batches that leave the chip nearly
// empty: there a launch lasts as long as one cell's march, and two launches in a row last twice that.
template <bool NMS>
__global__ __launch_bounds__(256) void fast_march_both_kernel(FastArgs a, int nfull, int nhalf) {
  const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (item >= nfull + nhalf) return;  // wave-uniform
  if (item < nfull)
    fast_march_body<false, NMS>(a, item, blockIdx.y);
  else
    fast_march_body<true, NMS>(a, item, blockIdx.y);
}

// The same cells walked by ONE resident workgroup per CU (4 x `waves per SIMD` waves; a second one does not fit beside it, so
// every CU gets its share -- four-wave workgroups were packed five to a CU on some CUs and none on others).  For the batched
// calls that run the blur beside FAST: a grid of one workgroup per four cells occupies every register of the chip for as
// long as cells are left and whatever else is queued waits behind it; a fixed set of resident waves leaves the rest of each
// SIMD's registers to the kernel beside it.  Alone, FAST reaches 93 % of the vector ALU's issue rate with five waves per
// SIMD, the same with four, 92 % of that with three and 76 % with two (occupancy sweep, NOTES.md section 6).
template <bool HALF, bool NMS>
__global__ __launch_bounds__(1024) void fast_march_resident_kernel(FastArgs a, int work0, int nwork, int nimages,
                                                                  uint32_t* next_cell) {
  // cells are handed out through a counter (zeroed by the launcher): a wave slowed down by its neighbours takes fewer
  const int total = nwork * nimages;
  while (true) {  // (wave-uniform; the counter passes `total` for every wave)
    // one atomic per wave, issued with the execution mask narrowed to lane 0 inside one asm statement: straight-line code
    // for the compiler (an `if (lane == 0)` in front of a uniform loop exit left it structurising a divergent loop)
    uint32_t r;
    asm volatile(
        "s_mov_b64 s[34:35], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add %0, %1, %2, %3 sc0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, s[34:35]"
        : "=&v"(r)
        : "v"(0u), "v"(1u), "s"(next_cell)
        : "memory", "s34", "s35");
    const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    if (u >= (uint32_t)total) break;
    const int image = (int)u / nwork, cell = (int)u - image * nwork;
    fast_march_body<HALF, NMS>(a, work0 + cell, image);
  }
}

// Standalone FAST detect: unit segments -> contiguous cv::KeyPoint list in raster order.
__global__ __launch_bounds__(256) void fast_emit_kernel(const VsfLevel* __restrict__ levels,
                                                        const uint32_t* __restrict__ cand, uint32_t cand_entries,
                                                        const uint16_t* __restrict__ rowstart, int nunits,
                                                        int max_keypoints, vsf_keypoint* __restrict__ out,
                                                        int32_t* __restrict__ counts, int32_t* __restrict__ status) {
  __shared__ int cellpre[2048 + 8];
  __shared__ __attribute__((aligned(4))) uint16_t rs_lds[64 * VSF_FAST_RS_STRIDE];
  __shared__ int lds4[4];
  const int image = blockIdx.x;
  const VsfLevel L = levels[0];
  const uint16_t* rs_img = rowstart + (size_t)image * nunits * VSF_FAST_RS_STRIDE;
  const uint32_t* cand_img = cand + (size_t)image * cand_entries;
  vsf_keypoint* o = out + (size_t)image * max_keypoints;
  const int n = vsf_gather_level<256>(L, cand_img, rs_img, cellpre, 2048, rs_lds, 64, lds4, [](int) {}, [&](int dst, uint32_t cd) {
    if (dst < max_keypoints) {
      vsf_keypoint kp;
      kp.x = (float)VSF_CAND_X(cd);
      kp.y = (float)VSF_CAND_Y(cd);
      kp.size = 7.f;
      kp.angle = -1.f;
      kp.response = (float)VSF_CAND_SCORE(cd);
      kp.octave = 0;
      kp.class_id = -1;
      o[dst] = kp;
    }
  });
  if (threadIdx.x == 0) {
    counts[image] = n;  // true count; the caller clamps to its capacity
    if (n > max_keypoints) atomicOr(status, 1);
  }
}

}  // namespace

void vsf_launch_fast(const VsfDev& d, const VsfGeom& g, const VsfImages& im, int threshold, int nms, hipStream_t s,
                     int resident_waves_per_simd, int n_cus, uint32_t* d_cell_counters) {
  FastArgs a;
  a.levels = d.levels;
  a.units = d.units;
  a.nunits = g.nunits;
  a.img0 = im.base;
  a.img0_stride = im.image_stride;
  a.img0_pitch = (int)im.row_stride;
  a.pyr = d.pyr;
  a.pyr_bytes = g.pyr_bytes;
  a.cand = d.cand;
  a.cand_entries = g.cand_entries;
  a.rowstart = d.rowstart;
  a.threshold = threshold;
  a.nms = nms;
  if (resident_waves_per_simd > 0 && resident_waves_per_simd <= 4 && nms && d_cell_counters) {
    vsf_note(hipMemsetAsync(d_cell_counters, 0, 2 * sizeof(uint32_t), s));
    const dim3 block(256 * resident_waves_per_simd);
    if (g.nwork_full > 0)
      hipLaunchKernelGGL((fast_march_resident_kernel<false, true>), dim3(n_cus), block, 0, s, a, 0, g.nwork_full, im.n,
                         d_cell_counters);
    if (g.nwork_half > 0)
      hipLaunchKernelGGL((fast_march_resident_kernel<true, true>), dim3(n_cus), block, 0, s, a, g.nwork_full,
                         g.nwork_half, im.n, d_cell_counters + 1);
    return;
  }
  const dim3 gf((g.nwork_full + 3) / 4, im.n), gh((g.nwork_half + 3) / 4, im.n);
  const int both_max = d.tune ? d.tune->fast_both_max : 16;
  if (im.n <= both_max && g.nwork_full > 0 && g.nwork_half > 0) {
    const dim3 gb((g.nwork_full + g.nwork_half + 3) / 4, im.n);
    if (nms)
      hipLaunchKernelGGL((fast_march_both_kernel<true>), gb, dim3(256), 0, s, a, g.nwork_full, g.nwork_half);
    else
      hipLaunchKernelGGL((fast_march_both_kernel<false>), gb, dim3(256), 0, s, a, g.nwork_full, g.nwork_half);
    return;
  }
  if (nms) {
    if (g.nwork_full > 0) hipLaunchKernelGGL((fast_march_kernel<false, true>), gf, dim3(256), 0, s, a, 0, g.nwork_full);
    if (g.nwork_half > 0)
      hipLaunchKernelGGL((fast_march_kernel<true, true>), gh, dim3(256), 0, s, a, g.nwork_full, g.nwork_half);
  } else {
    if (g.nwork_full > 0) hipLaunchKernelGGL((fast_march_kernel<false, false>), gf, dim3(256), 0, s, a, 0, g.nwork_full);
    if (g.nwork_half > 0)
      hipLaunchKernelGGL((fast_march_kernel<true, false>), gh, dim3(256), 0, s, a, g.nwork_full, g.nwork_half);
  }
}

void vsf_launch_fast_emit(const VsfDev& d, const VsfGeom& g, int n_images, int max_keypoints, vsf_keypoint* d_kp,
                          int32_t* d_counts, hipStream_t s) {
  hipLaunchKernelGGL(fast_emit_kernel, dim3(n_images), dim3(256), 0, s, d.levels, d.cand, g.cand_entries, d.rowstart,
                     g.nunits, max_keypoints, d_kp, d_counts, d.status);
}
