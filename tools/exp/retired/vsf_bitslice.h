// vsf_bitslice.h -- the FAST-9/16 segment TEST on bit planes, usable on the host and in HIP device code.
//
// cv::FAST_t<16> (features2d/fast.cpp, reached from ORB's computeKeyPoints and so from slam_frontend.cc:274) calls a
// pixel p a corner at threshold t when 9 contiguous pixels of its 16-pixel Bresenham circle are all brighter than
// I(p) + t or all darker than I(p) - t.  k_fast.hip answers that with the score itself (36 packed min / max per polarity
// and pixel pair: its time is its instruction count).  Here the question is asked of 32 pixels per lane at once: a row
// of 32 pixels is held as eight 32-bit planes (plane b, bit i = bit b of pixel i), a comparison of two rows is eight
// three-input boolean operations (v_bitop3_b32: full issue rate on gfx950), and the arc test is 40 more per polarity.
// The exact score is then only needed where the test says "corner" (k_fastbits.hip).
//
// Everything in here is a pure function of 32-bit words; tests/cpp/test_bitslice.cc checks each piece and the whole
// row test against the per-pixel definition on the host (where bitop3 is evaluated from its truth table).
#ifndef VSF_BITSLICE_H_
#define VSF_BITSLICE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VSF_BS_HD __host__ __device__ __forceinline__
#else
#define VSF_BS_HD inline
#endif

namespace vsf_bs {

// Truth table of a three-input function, in v_bitop3_b32's convention: bit (a << 2 | b << 1 | c) of the table.
#define VSF_BS_TT(expr_of_A_B_C) ((uint32_t)(expr_of_A_B_C) & 0xFFu)
constexpr uint32_t TA = 0xF0, TB = 0xCC, TC = 0xAA;
constexpr uint32_t TT_XOR3 = VSF_BS_TT(TA ^ TB ^ TC);                       // a ^ b ^ c
constexpr uint32_t TT_MAJ = VSF_BS_TT((TA & TB) | (TA & TC) | (TB & TC));   // carry of a + b + c
constexpr uint32_t TT_SEL = VSF_BS_TT((TA & TC) | (TB & ~TC));              // c ? a : b
constexpr uint32_t TT_GT = VSF_BS_TT((TA & ~TB) | (~(TA ^ TB) & TC));       // a > b, or equal and c (bit-serial compare)
constexpr uint32_t TT_AND3 = VSF_BS_TT(TA & TB & TC);
constexpr uint32_t TT_OR3 = VSF_BS_TT(TA | TB | TC);
constexpr uint32_t TT_AND_OR = VSF_BS_TT((TA & TB) | TC);                   // (a & b) | c
constexpr uint32_t TT_OR_AND = VSF_BS_TT((TA | TB) & TC);                   // (a | b) & c
constexpr uint32_t TT_A_ANDN_B = VSF_BS_TT(TA & ~TB);                       // a & ~b (c ignored)
constexpr uint32_t TT_A_ORN_B = VSF_BS_TT(TA | ~TB);                        // a | ~b (c ignored)
constexpr uint32_t TT_XOR3_NB = VSF_BS_TT(TA ^ ~TB ^ TC);                   // a ^ ~b ^ c
constexpr uint32_t TT_MAJ_NB = VSF_BS_TT((TA & ~TB) | (TA & TC) | (~TB & TC));  // carry of a + ~b + c

// A 32-bit constant as an operand of a three-input instruction: held in a vector register (a literal would travel through
// a scalar register, and a scalar operand halves the issue rate of the instruction: profiles/r04/valu_issue_table.json).
#if defined(__HIP_DEVICE_COMPILE__)
#define VSF_BS_VCONST(x) ([] { uint32_t m_ = (x); asm("" : "+v"(m_)); return m_; }())
#else
#define VSF_BS_VCONST(x) ((uint32_t)(x))
#endif

template <uint32_t TT>
VSF_BS_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
  uint32_t r = 0;
  if (TT & 0x01) r |= ~a & ~b & ~c;
  if (TT & 0x02) r |= ~a & ~b & c;
  if (TT & 0x04) r |= ~a & b & ~c;
  if (TT & 0x08) r |= ~a & b & c;
  if (TT & 0x10) r |= a & ~b & ~c;
  if (TT & 0x20) r |= a & ~b & c;
  if (TT & 0x40) r |= a & b & ~c;
  if (TT & 0x80) r |= a & b & c;
  return r;
#endif
}

VSF_BS_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {  // v_perm_b32: selector 0..3 = bytes of lo, 4..7 = of hi
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xFF;
    const uint32_t byte = s < 4 ? (lo >> (8 * s)) & 0xFF : s < 8 ? (hi >> (8 * (s - 4))) & 0xFF : 0u;
    r |= byte << (8 * i);
  }
  return r;
#endif
}

VSF_BS_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) {  // ({hi, lo} >> s)[31:0], s in 0..31
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, s);
#else
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31));
#endif
}

// Eight dwords of a row (w[k] = pixels 4k .. 4k+3, little endian) -> eight planes (p[b] bit i = bit b of pixel i).
// 16 byte permutations bring pixels k, k + 8, k + 16, k + 24 into dword k; three butterfly stages (shift + select)
// then exchange the dword index with the bit index inside each byte.
VSF_BS_HD void transpose_row(const uint32_t (&w)[8], uint32_t (&p)[8]) {
  uint32_t r[8];
#pragma unroll
  for (int a = 0; a < 2; a++) {
    // bytes c and c + 1 of dwords a, a + 2 (t) and a + 4, a + 6 (u)
    const uint32_t t01 = perm(w[a + 2], w[a], 0x05010400u), t23 = perm(w[a + 2], w[a], 0x07030602u);
    const uint32_t u01 = perm(w[a + 6], w[a + 4], 0x05010400u), u23 = perm(w[a + 6], w[a + 4], 0x07030602u);
    r[4 * a + 0] = perm(u01, t01, 0x05040100u);
    r[4 * a + 1] = perm(u01, t01, 0x07060302u);
    r[4 * a + 2] = perm(u23, t23, 0x05040100u);
    r[4 * a + 3] = perm(u23, t23, 0x07060302u);
  }
  // now r[k] byte m = pixel k + 8 m; exchange index bit j of k with bit j of the position inside the byte
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int s = 1 << j;
    const uint32_t M = j == 0 ? VSF_BS_VCONST(0x55555555u) : j == 1 ? VSF_BS_VCONST(0x33333333u) : VSF_BS_VCONST(0x0F0F0F0Fu);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k & s) continue;
      const uint32_t a = r[k], b = r[k | s];
      r[k] = bitop3<TT_SEL>(a, b << s, M);       // (a & M) | ((b << s) & ~M)
      r[k | s] = bitop3<TT_SEL>(a >> s, b, M);   // ((a >> s) & M) | (b & ~M)
    }
  }
#pragma unroll
  for (int b = 0; b < 8; b++) p[b] = r[b];
}

// tm[i] = all ones where bit i of the threshold is set (kept in vector registers: a scalar operand halves the issue rate)
VSF_BS_HD void threshold_masks(int t, uint32_t (&tm)[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) tm[i] = ((t >> i) & 1) ? 0xFFFFFFFFu : 0u;
}

// hi = min(I + t, 255), lo = max(I - t, 0) on planes.
VSF_BS_HD void saturating_add_sub(const uint32_t (&I)[8], const uint32_t (&tm)[8], uint32_t (&hi)[8], uint32_t (&lo)[8]) {
  uint32_t c = I[0] & tm[0];
  hi[0] = I[0] ^ tm[0];
#pragma unroll
  for (int i = 1; i < 8; i++) {
    hi[i] = bitop3<TT_XOR3>(I[i], tm[i], c);
    c = bitop3<TT_MAJ>(I[i], tm[i], c);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) hi[i] |= c;  // overflow: 255
  // I - t = I + ~t + 1: the carry out says "no borrow"
  uint32_t d = bitop3<TT_A_ORN_B>(I[0], tm[0], tm[0]);  // carry of I0 + ~t0 + 1
  lo[0] = I[0] ^ tm[0];                                 // I0 ^ ~t0 ^ 1
#pragma unroll
  for (int i = 1; i < 8; i++) {
    lo[i] = bitop3<TT_XOR3_NB>(I[i], tm[i], d);
    d = bitop3<TT_MAJ_NB>(I[i], tm[i], d);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) lo[i] &= d;  // borrow: 0
}

// mask of the pixels where x > y (unsigned bytes as planes), least significant bit first
VSF_BS_HD uint32_t greater(const uint32_t (&x)[8], const uint32_t (&y)[8]) {
  uint32_t g = bitop3<TT_A_ANDN_B>(x[0], y[0], y[0]);
#pragma unroll
  for (int i = 1; i < 8; i++) g = bitop3<TT_GT>(x[i], y[i], g);
  return g;
}

// A plane moved by dx pixels: result bit i = bit (i - dx) of the row the lanes hold side by side (prev = the lane to the
// left's word, next = the lane to the right's).  dx in -3 .. 3.
template <int DX>
VSF_BS_HD uint32_t shifted(uint32_t cur, uint32_t prev, uint32_t next) {
  if (DX == 0) return cur;
  if (DX > 0) return alignbit(cur, prev, 32 - DX);
  return alignbit(next, cur, -DX);
}

// any 9 contiguous of the 16 masks (circle order) all set
VSF_BS_HD uint32_t arc9(const uint32_t (&m)[16]) {
  uint32_t c3[16];
#pragma unroll
  for (int i = 0; i < 16; i++) c3[i] = bitop3<TT_AND3>(m[i], m[(i + 1) & 15], m[(i + 2) & 15]);
  uint32_t c9[16];
#pragma unroll
  for (int i = 0; i < 16; i++) c9[i] = bitop3<TT_AND3>(c3[i], c3[(i + 3) & 15], c3[(i + 6) & 15]);
  uint32_t o[6];
#pragma unroll
  for (int i = 0; i < 5; i++) o[i] = bitop3<TT_OR3>(c9[3 * i], c9[3 * i + 1], c9[3 * i + 2]);
  o[5] = c9[15];
  return bitop3<TT_OR3>(o[0], o[1], o[2]) | bitop3<TT_OR3>(o[3], o[4], o[5]);
}

// The circle in OpenCV's order (fast_score.cpp makeOffsets, patternSize 16): index -> (dx, dy)
constexpr int CIRCLE_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
constexpr int CIRCLE_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};


// ---- the forward scheme ------------------------------------------------------------------------------------------
// With hi_a = min(I_a + t, 255) and lo_a = max(I_a - t, 0) of the centre row a, a circle pixel at offset d = (dx, dy)
// is "brighter" when I_{a+dy}(x + dx) > hi_a(x) and "darker" when I_{a+dy}(x + dx) < lo_a(x).  The same comparison read
// from the other end says that for the pixel x' = x + dx of row b = a + dy the circle pixel at -d is darker resp.
// brighter.  So only the eight offsets that point forward (dy > 0, or dy = 0 and dx > 0) are compared, in the frame of
// the row below (x' = x + dx: the centre row's planes are the ones that move); a result serves the centre row now,
// moved back by dx, and the row dy steps later as it is.  16 comparisons per row instead of 32, and a window of four
// rows of planes instead of seven.
constexpr int FWD_IDX[8] = {0, 1, 15, 2, 14, 3, 13, 4};  // circle indices of the forward offsets, by dy = 3, 3, 3, 2, 2, 1, 1, 0

template <int DX>
VSF_BS_HD void shift_planes(const uint32_t (&cur)[8], const uint32_t (&prev)[8], const uint32_t (&next)[8], uint32_t (&out)[8]) {
#pragma unroll
  for (int b = 0; b < 8; b++) out[b] = shifted<DX>(cur[b], prev[b], next[b]);
}

// rows[r] = planes of image row a + r (r = 0 .. 3).  mb[f](x') = [I_b(x') > hi_a(x' - dx)], md[f](x') = [I_b(x') < lo_a(x' - dx)].
VSF_BS_HD void forward_masks(const uint32_t (&rows)[4][8], const uint32_t (&hi)[8], const uint32_t (&hi_prev)[8],
                             const uint32_t (&hi_next)[8], const uint32_t (&lo)[8], const uint32_t (&lo_prev)[8],
                             const uint32_t (&lo_next)[8], uint32_t (&mb)[8], uint32_t (&md)[8]) {
  uint32_t h[8], l[8];
  mb[0] = greater(rows[3], hi);  // (0, 3)
  md[0] = greater(lo, rows[3]);
  shift_planes<1>(hi, hi_prev, hi_next, h);
  shift_planes<1>(lo, lo_prev, lo_next, l);
  mb[1] = greater(rows[3], h);   // (1, 3)
  md[1] = greater(l, rows[3]);
  shift_planes<-1>(hi, hi_prev, hi_next, h);
  shift_planes<-1>(lo, lo_prev, lo_next, l);
  mb[2] = greater(rows[3], h);   // (-1, 3)
  md[2] = greater(l, rows[3]);
  shift_planes<2>(hi, hi_prev, hi_next, h);
  shift_planes<2>(lo, lo_prev, lo_next, l);
  mb[3] = greater(rows[2], h);   // (2, 2)
  md[3] = greater(l, rows[2]);
  shift_planes<-2>(hi, hi_prev, hi_next, h);
  shift_planes<-2>(lo, lo_prev, lo_next, l);
  mb[4] = greater(rows[2], h);   // (-2, 2)
  md[4] = greater(l, rows[2]);
  shift_planes<-3>(hi, hi_prev, hi_next, h);
  shift_planes<-3>(lo, lo_prev, lo_next, l);
  mb[6] = greater(rows[1], h);   // (-3, 1)
  md[6] = greater(l, rows[1]);
  shift_planes<3>(hi, hi_prev, hi_next, h);
  shift_planes<3>(lo, lo_prev, lo_next, l);
  mb[5] = greater(rows[1], h);   // (3, 1)
  md[5] = greater(l, rows[1]);
  mb[7] = greater(rows[0], h);   // (3, 0)
  md[7] = greater(l, rows[0]);
}

// Results of the last three centre rows, as the rows below them will read them.
struct FastHistory {
  uint32_t b3[3][3], d3[3][3];  // [f = 0..2][age]: age 0 = the previous centre row
  uint32_t b2[2][2], d2[2][2];  // [f - 3][age]
  uint32_t b1[2], d1[2];        // [f - 5]
  VSF_BS_HD void clear() {
#pragma unroll
    for (int f = 0; f < 3; f++)
#pragma unroll
      for (int g = 0; g < 3; g++) b3[f][g] = d3[f][g] = 0;
#pragma unroll
    for (int f = 0; f < 2; f++) {
      b2[f][0] = b2[f][1] = d2[f][0] = d2[f][1] = 0;
      b1[f] = d1[f] = 0;
    }
  }
  VSF_BS_HD void push(const uint32_t (&mb)[8], const uint32_t (&md)[8]) {
#pragma unroll
    for (int f = 0; f < 3; f++) {
      b3[f][2] = b3[f][1], b3[f][1] = b3[f][0], b3[f][0] = mb[f];
      d3[f][2] = d3[f][1], d3[f][1] = d3[f][0], d3[f][0] = md[f];
    }
#pragma unroll
    for (int f = 0; f < 2; f++) {
      b2[f][1] = b2[f][0], b2[f][0] = mb[3 + f];
      d2[f][1] = d2[f][0], d2[f][0] = md[3 + f];
      b1[f] = mb[5 + f];
      d1[f] = md[5 + f];
    }
  }
};

// Corner masks of the centre row from this row's forward results (with the neighbouring lanes' words) and the history:
// *bright = pixels with 9 contiguous brighter circle pixels, *dark likewise darker (never both: 9 + 9 > 16).
VSF_BS_HD void corner_masks(const uint32_t (&mb)[8], const uint32_t (&mb_prev)[8], const uint32_t (&mb_next)[8],
                            const uint32_t (&md)[8], const uint32_t (&md_prev)[8], const uint32_t (&md_next)[8],
                            const FastHistory& hst, uint32_t* bright, uint32_t* dark) {
  uint32_t B[16], D[16];
  // forward offsets: the comparison sits at x + dx
  B[0] = mb[0], D[0] = md[0];
  B[1] = shifted<-1>(mb[1], mb_prev[1], mb_next[1]), D[1] = shifted<-1>(md[1], md_prev[1], md_next[1]);
  B[15] = shifted<1>(mb[2], mb_prev[2], mb_next[2]), D[15] = shifted<1>(md[2], md_prev[2], md_next[2]);
  B[2] = shifted<-2>(mb[3], mb_prev[3], mb_next[3]), D[2] = shifted<-2>(md[3], md_prev[3], md_next[3]);
  B[14] = shifted<2>(mb[4], mb_prev[4], mb_next[4]), D[14] = shifted<2>(md[4], md_prev[4], md_next[4]);
  B[3] = shifted<-3>(mb[5], mb_prev[5], mb_next[5]), D[3] = shifted<-3>(md[5], md_prev[5], md_next[5]);
  B[13] = shifted<3>(mb[6], mb_prev[6], mb_next[6]), D[13] = shifted<3>(md[6], md_prev[6], md_next[6]);
  B[4] = shifted<-3>(mb[7], mb_prev[7], mb_next[7]), D[4] = shifted<-3>(md[7], md_prev[7], md_next[7]);
  // backward offsets -d: what the row dy above found for d, read from this end (brighter <-> darker)
  B[8] = hst.d3[0][2], D[8] = hst.b3[0][2];     // -(0, 3)
  B[9] = hst.d3[1][2], D[9] = hst.b3[1][2];     // -(1, 3) = (-1, -3)
  B[7] = hst.d3[2][2], D[7] = hst.b3[2][2];     // -(-1, 3) = (1, -3)
  B[10] = hst.d2[0][1], D[10] = hst.b2[0][1];   // -(2, 2)
  B[6] = hst.d2[1][1], D[6] = hst.b2[1][1];     // -(-2, 2)
  B[11] = hst.d1[0], D[11] = hst.b1[0];         // -(3, 1)
  B[5] = hst.d1[1], D[5] = hst.b1[1];           // -(-3, 1)
  B[12] = md[7], D[12] = mb[7];                 // -(3, 0): this row
  *bright = arc9(B);
  *dark = arc9(D);
}

}  // namespace vsf_bs

#endif  // VSF_BITSLICE_H_
