// k_match.hip -- K9 + K10: cv::BFMatcher(NORM_HAMMING)::knnMatch(k = 2) (features2d/matchers.cpp ->
// core/stat.cpp batchDistance + normHamming) and the reference's ratio test, Frontend::GetMatches
// (slam_frontend.cc:521-538).
//
// knn2: Hamming distances on the matrix cores.  A descriptor bit becomes a +-4 FP4 value (train) / -+4 (query), so a train
// row x query column at Hamming distance d multiplies to 8192 d - 2^20; with the row's index riding in the C operand the
// matrix instruction produces the packed sort key itself and the vector ALU only keeps the two smallest per query
// (v_min_i32 + v_med3_i32 on the keys' bits), which is "smaller distance first, ties to the lower train index", exactly
// batchDistance's insertion rule.  A wave owns 32 queries, a workgroup's four waves share each expanded 32-row train tile
// through LDS (double buffered, one barrier per tile); train sets beyond 4096 rows are folded into  d << 20 | index  keys
// per range; with few pairs the train set is split over workgroups and merged by 64-bit CAS.  Bound by the VALU fold beside
// the MFMA pipe; 2000 x 2000 rows move 128 KB, HBM is idle.  (A v_xor / v_bcnt kernel with per-lane queries took 2.7x as
// long and was retired in round 1; round 2's int8 form of this one in round 6: tools/exp/retired/.)
// ratio_compact: per pair, keep  d1 * 2^shift < num * d2  (the double-precision compare of the reference,
// exact in integers) and compact the survivors in ascending query index.
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "vsf_internal.h"

namespace {

constexpr int kTile = 32;          // train rows per MFMA tile
constexpr int kChunkRows = 4096;   // train rows per key range (the row offsets inside a key stay below a distance step)
constexpr int kSplitAlign = 128;   // split chunks are multiples of this many train rows (four tiles)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ inline void merge_top2(uint32_t& m1, uint32_t& m2, uint32_t b1, uint32_t b2) {
  const uint32_t lo = min(m1, b1), hi = max(m1, b1);
  m2 = min(hi, min(m2, b2));
  m1 = lo;
}

// (Round 2's form of the same idea on v_mfma_i32_32x32x32_i8 -- bits as +-64 int8, nine instructions per 256 bits -- is
// in tools/exp/retired/k_match_int8.hip.)
constexpr int kWgQueries = 128;   // 4 waves x 32 queries

__device__ inline int imed3(int a, int b, int c) { return min(max(a, b), max(min(a, b), c)); }  // v_med3_i32
// v_min3_i32.  `a` goes through an empty asm first: min(a, b) is also a subexpression of imed3(a, b, c), and once the compiler
// has merged the two it emits two v_min instead of one v_min3 (the asm holds no instruction and touches no matrix result).
__device__ inline int imin3(int a, int b, int c) {
  asm("" : "+v"(a));
  return min(min(a, b), c);
}

// ---- round 5: the same key out of the FP4 matrix instruction ----
// v_mfma_scale_f32_32x32x64_f8f6f4 with E2M1 operands takes K = 64 per instruction at the cycles v_mfma_i32_32x32x32_i8
// takes for K = 32 (tools/exp/fp4_probe.hip: 132 against 256 cycles per 256-bit 32 x 32 tile).  A descriptor bit becomes
// the FP4 code of +4 or -4 (0x6 / 0xE), both block scales are 2^4 (E8M0 131), so a product is +-4096 exactly as with the
// int8 +-64 operands and the f32 accumulator -- every partial sum a multiple of 4096 below 2^21 -- is exact; 256 bits are
// FOUR instructions, and the row index needs no instruction at all: it rides in the C operand of the first one (sixteen
// constant registers, which the halved operands pay for: a query is 16 registers instead of 32).  C also carries 2^20 +
// 8192, so that a key
//     8192 (d + 1) + (row inside the tile)          -- 65 536 probe keys, all exact
// is a POSITIVE float whatever the distance, and stays positive while the running minima are rebased by 32 per tile over a
// 4096-row chunk: positive floats order like their bit patterns, so the fold is the integer v_min_i32 / v_med3_i32 of the
// int8 kernel on the accumulator's bits (no float min, whose signalling-NaN quieting the compiler would have to add).
// An expanded train row is 128 bytes (144 with the pad that keeps ds_read_b128 conflict-free: 36 dwords per row step, 16
// lanes tile the 64 banks), a thread stages one 16-byte piece per tile.
constexpr int kTileStride4 = 144;
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// 8 descriptor bits -> 8 FP4 codes (bit i -> nibble i): three spreading steps, then the sign (bit 3 of a nibble)
__device__ inline uint32_t fp4_spread8(uint32_t b) {
  uint32_t x = b & 0xFFu;
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x << 3;
}
// TRAIN: set bit -> +4 (0x6), clear -> -4 (0xE); QUERY: set bit -> -4, clear -> +4
template <bool TRAIN>
__device__ inline v4i fp4_expand32(uint32_t bits) {
  v4i r;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t sgn = fp4_spread8(bits >> (8 * k));
    r[k] = (int)(TRAIN ? (0xEEEEEEEEu ^ sgn) : (0x66666666u | sgn));
  }
  return r;
}
__device__ inline v8i fp4_operand(v4i v) { return v8i{v[0], v[1], v[2], v[3], 0, 0, 0, 0}; }  // (FP4 reads four registers)

constexpr int kKeyNoneBits = 0x7F000000;  // 1.7e38f: above every key as a float and as an integer

template <bool SPLIT>
__global__ __launch_bounds__(256, 4) void knn2_fp4_kernel(const uint8_t* __restrict__ desc,
                                                          const int32_t* __restrict__ counts, size_t set_stride,
                                                          const int32_t* __restrict__ q_set,
                                                          const int32_t* __restrict__ t_set, int max_rows,
                                                          int32_t* __restrict__ idx2, int32_t* __restrict__ dist2) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[2][kTile * kTileStride4];
  // byte of descriptor bits -> dword of eight train-side FP4 codes: the staging of a tile is four table reads per thread
  // instead of ~20 vector instructions (the fold, 22 instructions per tile, is what the vector ALU is for here:
  // 0.188 -> 0.173 ms per 256 pairs of 2000 x 2000)
  __shared__ uint32_t lut[256];
  lut[threadIdx.x] = 0xEEEEEEEEu ^ fp4_spread8(threadIdx.x);  // (the first barrier below orders it)
  const int pair = blockIdx.y;
  const int qs = q_set ? q_set[pair] : 2 * pair, ts = t_set ? t_set[pair] : 2 * pair + 1;
  const int nq = min(counts[qs], max_rows), nt = min(counts[ts], max_rows);
  if ((int)blockIdx.x * kWgQueries >= nq) return;  // whole block idle (uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const uint8_t* Q = desc + (size_t)qs * set_stride;
  const uint32_t* T = reinterpret_cast<const uint32_t*>(desc + (size_t)ts * set_stride);
  // query operand: column c of the wave's tile; step s takes dword 2 s + h of the descriptor (32 of the step's 64 bits)
  const int qbase = blockIdx.x * kWgQueries + wave * 32;
  v4i qf[4];
  {
    const int q = min(qbase + c, nq - 1);
    const uint32_t* qw = reinterpret_cast<const uint32_t*>(Q + (size_t)q * 32);
#pragma unroll
    for (int s = 0; s < 4; s++) qf[s] = fp4_expand32<false>(qw[2 * s + h]);
  }
  // C operand of a tile's first product: the row's index inside the tile + 2^20 + 8192 (see above)
  v16f row_c;
#pragma unroll
  for (int i = 0; i < 16; i++) row_c[i] = (float)((i & 3) + 8 * (i >> 2) + 4 * h + (1 << 20) + 8192);
  const int scale = 131;  // E8M0: 2^4 on both operands' blocks
  int t_begin = 0, t_end = nt;
  if (SPLIT) {
    const int chunk = ((nt + (int)gridDim.z - 1) / (int)gridDim.z + kSplitAlign - 1) / kSplitAlign * kSplitAlign;
    t_begin = min((int)blockIdx.z * chunk, nt);
    t_end = min(t_begin + chunk, nt);
  }
  uint32_t g1 = 0xFFFFFFFFu, g2 = 0xFFFFFFFFu;  // distance << 20 | train index
  // staging: thread tid expands dword (tid & 7) of train row (tid >> 3) of the tile: 16 bytes at row * 144 + 16 * dword.
  // Rows past the chunk's end are clamped to its last row (never folded: the last tile's fold checks the row index).
  // (Measured and left out: stages of 64 rows, one barrier per two tiles -- 0.185 against 0.172 ms per 256 pairs of
  // 2000 x 2000; sched_group_barrier groups of 1 MFMA + 8 / 12 / 16 vector instructions -- 0.181 / 0.173 / 0.176: the
  // compiler's own interleaving is as good; a fold that skips groups of four accumulator registers none of whose keys
  // is below a lane's second best -- v_min3 + v_min + one compare + a wave-uniform branch per group, the eight fold
  // instructions for 20-60 % of the groups -- 0.187 ms, and 0.979 against 0.944 ms per 64 pairs of 10 000 x 10 000: the
  // branches cost more than the instructions they save.  Round 6: all of a step's LDS traffic issued at its start -- the
  // next tile's write from registers expanded a step earlier, so that nothing is in flight at the barrier: 0.158 against
  // 0.152 ms and 0.894 against 0.870 ms; two query tiles per wave, half the LDS reads per distance
  // (tools/exp/match_wide.patch): equal.  The counters (tools/exp/match_counters.sh, 64 pairs of 10 000 x 10 000): matrix
  // pipe busy 38 % of the SIMD cycles, vector ALU 50 %, both at once 16 %, the LDS 58 % of the CU's cycles (a third of it
  // bank conflicts of the table reads): no unit is full, the waves of a SIMD take turns.)
  const int st_row = tid >> 3, st_s = tid & 7;
  auto load_bits = [&](int t0, int c_end) -> uint32_t { return T[(size_t)min(t0 + st_row, c_end - 1) * 8 + st_s]; };
  auto stage = [&](int buf, uint32_t bits) {
    v4i r;
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = (int)lut[(bits >> (8 * k)) & 255u];
    *reinterpret_cast<v4i*>(&tile[buf][st_row * kTileStride4 + 16 * st_s]) = r;
  };
  for (int c0 = t_begin; c0 < t_end; c0 += kChunkRows) {
    const int c_end = min(c0 + kChunkRows, t_end);
    const int ntiles = (c_end - c0 + kTile - 1) / kTile;
    float m1 = __int_as_float(kKeyNoneBits), m2 = __int_as_float(kKeyNoneBits);  // key - 32 (tiles since): positive floats
    auto fold = [&](const v16f& a, int t0, bool check) {
      m1 -= (float)kTile;  // (exact: integers below 2^24; the "none" key stays 1.7e38)
      m2 -= (float)kTile;
      int i1 = __float_as_int(m1), i2 = __float_as_int(m2);
      // Two keys at a time: the second smallest of {i1, i2, k0, k1} with i1 <= i2 is min(i2, median(i1, k0, k1)) -- either
      // i2 or the second smallest of the other three -- so a pair costs v_med3 + v_min3 and two pairs share one v_min3 into
      // i2: five instructions per four keys where one key at a time (v_med3 + v_min) takes eight.
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        int k[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bool valid = !check || t0 + ((i + j) & 3) + 8 * ((i + j) >> 2) + 4 * h < c_end;
          k[j] = valid ? __float_as_int(a[i + j]) : kKeyNoneBits;
        }
        const int ta = imed3(i1, k[0], k[1]), ia = imin3(i1, k[0], k[1]);
        const int tb = imed3(ia, k[2], k[3]);
        i1 = imin3(ia, k[2], k[3]);
        i2 = imin3(i2, ta, tb);
      }
      m1 = __int_as_float(i1);
      m2 = __int_as_float(i2);
    };
    constexpr int kAhead = 4;
    uint32_t bits_next[kAhead];
    // One pipeline step = tile t: its four products (into `n`) run beside the fold of tile t - 1's keys (`p`) and the
    // expansion of tile t + 1 into the other LDS buffer: matrix core and vector ALU side by side inside one wave.
    auto step = [&](int t, const v16f& p, v16f& n, bool fold_prev) {
      const int buf = t & 1;
      const uint32_t bits_after = load_bits(c0 + (t + 1 + kAhead) * kTile, c_end);
      const uint8_t* rowp = &tile[buf][c * kTileStride4 + 16 * h];
      v4i tf[4];
#pragma unroll
      for (int s = 0; s < 4; s++) tf[s] = *reinterpret_cast<const v4i*>(rowp + 32 * s);
      n = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_operand(tf[0]), fp4_operand(qf[0]), row_c, 4, 4, 0, scale, 0, scale);
#pragma unroll
      for (int s = 1; s < 4; s++)
        n = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_operand(tf[s]), fp4_operand(qf[s]), n, 4, 4, 0, scale, 0, scale);
      if (fold_prev) fold(p, 0, false);  // (every tile but a chunk's last is full)
      stage(buf ^ 1, bits_next[0]);
#pragma unroll
      for (int k = 0; k + 1 < kAhead; k++) bits_next[k] = bits_next[k + 1];
      bits_next[kAhead - 1] = bits_after;
      __syncthreads();
    };
    __syncthreads();  // the previous chunk's last tile has been read
    stage(0, load_bits(c0, c_end));
#pragma unroll
    for (int k = 0; k < kAhead; k++) bits_next[k] = load_bits(c0 + (1 + k) * kTile, c_end);
    __syncthreads();
    v16f accA, accB = row_c;  // (B is not folded before it is written)
    step(0, accB, accA, false);
    int t = 1;
    for (; t + 1 < ntiles; t += 2) {
      step(t, accA, accB, true);
      step(t + 1, accB, accA, true);
    }
    int t_last;  // the tile m1 / m2 are relative to
    if (t < ntiles) {  // uniform
      step(t, accA, accB, true);
      fold(accB, c0 + t * kTile, true);
      t_last = t;
    } else {
      fold(accA, c0 + (t - 1) * kTile, true);
      t_last = t - 1;
    }
    // chunk keys -> global keys:  key + 4096 = 8192 (d + 1) + (row offset + 4096), the row offset being in (-4096, 32)
    auto global_key = [&](float key) -> uint32_t {
      if (key > 1e30f) return 0xFFFFFFFFu;
      const uint32_t y = (uint32_t)((int)key + 4096);
      return (((y >> 13) - 1u) << 20) | (uint32_t)((int)(y & 8191u) - 4096 + t_last * kTile + c0);
    };
    merge_top2(g1, g2, global_key(m1), global_key(m2));
  }
  // the two lane halves hold the rows 4 h + ... of every tile: merge them
  {
    const uint32_t o1 = __shfl_xor(g1, 32), o2 = __shfl_xor(g2, 32);
    merge_top2(g1, g2, o1, o2);
  }
  if (h != 0) return;
  const int q = qbase + c;
  if (q >= nq) return;
  const uint32_t b1 = g1, b2 = g2;
  if (SPLIT) {
    if (t_begin >= t_end) return;
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(dist2) + (size_t)pair * max_rows + q;
    unsigned long long seen = *slot;
    while (true) {
      const uint32_t a1 = (uint32_t)(seen >> 32), a2 = (uint32_t)seen;
      const uint32_t n1 = min(a1, b1), n2 = min(max(a1, b1), min(a2, b2));
      const unsigned long long merged = ((unsigned long long)n1 << 32) | n2;
      if (merged == seen) break;
      const unsigned long long prev = atomicCAS(slot, seen, merged);
      if (prev == seen) break;
      seen = prev;
    }
  } else {
    const size_t o = ((size_t)pair * max_rows + q) * 2;
    idx2[o] = b1 == 0xFFFFFFFFu ? -1 : (int32_t)(b1 & 0xFFFFFu);
    idx2[o + 1] = b2 == 0xFFFFFFFFu ? -1 : (int32_t)(b2 & 0xFFFFFu);
    dist2[o] = b1 == 0xFFFFFFFFu ? INT_MAX : (int32_t)(b1 >> 20);
    dist2[o + 1] = b2 == 0xFFFFFFFFu ? INT_MAX : (int32_t)(b2 >> 20);
  }
}

// Unpacks the key pairs of the split path into idx2 / dist2 (in place: a thread reads its slot before writing it).
__global__ __launch_bounds__(256) void knn2_finalize_kernel(const int32_t* __restrict__ counts,
                                                            const int32_t* __restrict__ q_set, int max_rows,
                                                            int32_t* __restrict__ idx2, int32_t* __restrict__ dist2) {
  const int pair = blockIdx.y;
  const int qs = q_set ? q_set[pair] : 2 * pair;
  const int nq = min(counts[qs], max_rows);
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const unsigned long long key = (reinterpret_cast<const unsigned long long*>(dist2))[(size_t)pair * max_rows + q];
  const uint32_t b1 = (uint32_t)(key >> 32), b2 = (uint32_t)key;
  const size_t o = ((size_t)pair * max_rows + q) * 2;
  idx2[o] = b1 == 0xFFFFFFFFu ? -1 : (int32_t)(b1 & 0xFFFFFu);
  idx2[o + 1] = b2 == 0xFFFFFFFFu ? -1 : (int32_t)(b2 & 0xFFFFFu);
  dist2[o] = b1 == 0xFFFFFFFFu ? INT_MAX : (int32_t)(b1 >> 20);
  dist2[o + 1] = b2 == 0xFFFFFFFFu ? INT_MAX : (int32_t)(b2 >> 20);
}

__global__ __launch_bounds__(256) void ratio_compact_kernel(const int32_t* __restrict__ counts,
                                                            const int32_t* __restrict__ q_set,
                                                            const int32_t* __restrict__ t_set, int max_rows,
                                                            const int32_t* __restrict__ idx2,
                                                            const int32_t* __restrict__ dist2, uint32_t ratio_num,
                                                            uint32_t ratio_shift, vsf_dmatch* __restrict__ matches,
                                                            int32_t* __restrict__ nmatches) {
  __shared__ int wsum[4];
  __shared__ int s_base;
  const int pair = blockIdx.x;
  const int qs = q_set ? q_set[pair] : 2 * pair, ts = t_set ? t_set[pair] : 2 * pair + 1;
  const int nq = min(counts[qs], max_rows), nt = min(counts[ts], max_rows);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  if (nt >= 2) {  // quirk Q6: with fewer than 2 train rows the reference reads matches[i][1] out of bounds
    for (int q0 = 0; q0 < nq; q0 += 256) {
      const int q = q0 + threadIdx.x;
      bool keep = false;
      int d1 = 0, i1 = 0;
      if (q < nq) {
        const size_t o = ((size_t)pair * max_rows + q) * 2;
        d1 = dist2[o];
        i1 = idx2[o];
        const int d2 = dist2[o + 1];
        keep = ((uint64_t)(uint32_t)d1 << ratio_shift) < (uint64_t)ratio_num * (uint32_t)d2;
      }
      const uint64_t m = __ballot(keep);
      const int within = __popcll(m & ((1ull << lane) - 1));
      if (lane == 0) wsum[wid] = __popcll(m);
      __syncthreads();
      int base = s_base;
      for (int w = 0; w < wid; w++) base += wsum[w];
      if (keep) {
        vsf_dmatch dm;
        dm.queryIdx = q;
        dm.trainIdx = i1;
        dm.imgIdx = 0;
        dm.distance = (float)d1;
        matches[(size_t)pair * max_rows + base + within] = dm;
      }
      __syncthreads();
      if (threadIdx.x == 0) s_base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) nmatches[pair] = s_base;
}

}  // namespace

void vsf_launch_knn2(const uint8_t* d_desc, const int32_t* d_counts, size_t set_stride, const int32_t* d_q_set,
                     const int32_t* d_t_set, int n_pairs, int max_rows, int32_t* d_idx2, int32_t* d_dist2,
                     hipStream_t s, int rows_hint) {
  const int qtiles = (max_rows + kWgQueries - 1) / kWgQueries;
  // A batch of 128 stereo pairs brings ~2000 workgroups, two full rounds of the chip at four workgroups per CU, and runs
  // unsplit; with fewer (one pair of one frame at a time) the train sets are split until about that many workgroups exist
  // (each at least eight 32-row tiles) and merged through the packed key pairs.
  // rows_hint (0: unknown): how many rows the sets are EXPECTED to hold -- the capacity max_rows says nothing about a
  // filtered frame of a few hundred features, whose whole train set is a handful of tiles: no split, and with it no memset in
  // front and no finalize launch behind (a wrong hint costs time, never a result).
  const int rows = rows_hint > 0 ? std::min(rows_hint, max_rows) : max_rows;
  const int qtiles_full = (rows + kWgQueries - 1) / kWgQueries;
  int nsplit = 1;
  if ((long)qtiles_full * n_pairs < 2 * 768) nsplit = (int)std::min<long>(32, 2 * 1024 / ((long)qtiles_full * n_pairs));
  nsplit = std::max(1, std::min(nsplit, rows / (2 * kSplitAlign)));
  if (nsplit <= 1) {
    hipLaunchKernelGGL(knn2_fp4_kernel<false>, dim3(qtiles, n_pairs, 1), dim3(256), 0, s, d_desc, d_counts, set_stride,
                       d_q_set, d_t_set, max_rows, d_idx2, d_dist2);
    return;
  }
  vsf_note(hipMemsetAsync(d_dist2, 0xFF, (size_t)n_pairs * max_rows * 2 * sizeof(int32_t), s));
  hipLaunchKernelGGL(knn2_fp4_kernel<true>, dim3(qtiles, n_pairs, nsplit), dim3(256), 0, s, d_desc, d_counts, set_stride,
                     d_q_set, d_t_set, max_rows, d_idx2, d_dist2);
  hipLaunchKernelGGL(knn2_finalize_kernel, dim3((max_rows + 255) / 256, n_pairs, 1), dim3(256), 0, s, d_counts, d_q_set,
                     max_rows, d_idx2, d_dist2);
}

void vsf_launch_ratio_compact(const int32_t* d_counts, const int32_t* d_q_set, const int32_t* d_t_set, int n_pairs,
                              int max_rows, const int32_t* d_idx2, const int32_t* d_dist2, uint32_t ratio_num,
                              uint32_t ratio_shift, vsf_dmatch* d_matches, int32_t* d_nmatches, int32_t* d_status,
                              hipStream_t s) {
  (void)d_status;
  hipLaunchKernelGGL(ratio_compact_kernel, dim3(n_pairs), dim3(256), 0, s, d_counts, d_q_set, d_t_set, max_rows,
                     d_idx2, d_dist2, ratio_num, ratio_shift, d_matches, d_nmatches);
}
