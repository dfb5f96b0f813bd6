// k_match.hip -- K9 + K10: cv::BFMatcher(NORM_HAMMING)::knnMatch(k = 2) (features2d/matchers.cpp ->
// core/stat.cpp batchDistance + normHamming) and the reference's ratio test, Frontend::GetMatches
// (slam_frontend.cc:521-538).
//
// knn2: one lane per query row keeps its 256-bit descriptor in 8 VGPRs; the train set streams through LDS in
// blocks of 256 rows (coalesced 16-byte loads, then wave-uniform ds_read_b128 broadcasts).  A distance is
// 8 x (v_xor + v_bcnt accumulate); the running top-2 is kept on packed keys  (distance << 20 | train index),
// so "smaller distance first, ties to the lower train index" -- exactly batchDistance's insertion rule -- is an
// unsigned min / med3.  Integer VALU bound (not HBM, not MFMA): 2000 x 2000 rows move 128 KB.
// ratio_compact: per pair, keep  d1 * 2^shift < num * d2  (the double-precision compare of the reference,
// exact in integers) and compact the survivors in ascending query index.
#include <limits.h>

#include <algorithm>

#include "vsf_internal.h"

namespace {

constexpr int kTrainTile = 256;

// SPLIT = false: a workgroup walks the whole train set and writes idx2 / dist2.
// SPLIT = true : gridDim.z workgroups share a query tile, each walks one chunk of the train set and merges its top-2 into
//                the query's packed 64-bit key pair (best << 32 | second) kept in the dist2 slot, with a CAS loop (the
//                merge  m1 = min(a1, b1), m2 = min(max(a1, b1), min(a2, b2))  is associative and commutative);
//                knn2_finalize_kernel then unpacks.
template <bool SPLIT>
__global__ __launch_bounds__(256) void knn2_kernel(const uint8_t* __restrict__ desc,
                                                   const int32_t* __restrict__ counts, size_t set_stride,
                                                   const int32_t* __restrict__ q_set,
                                                   const int32_t* __restrict__ t_set, int max_rows,
                                                   int32_t* __restrict__ idx2, int32_t* __restrict__ dist2) {
  __shared__ uint4 tile[kTrainTile * 2];
  const int pair = blockIdx.y;
  const int qs = q_set ? q_set[pair] : 2 * pair, ts = t_set ? t_set[pair] : 2 * pair + 1;
  const int nq = min(counts[qs], max_rows), nt = min(counts[ts], max_rows);
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= nq) return;  // whole block idle (uniform)
  const uint8_t* Q = desc + (size_t)qs * set_stride;
  const uint8_t* T = desc + (size_t)ts * set_stride;
  uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
  if (q < nq) {
    qa = reinterpret_cast<const uint4*>(Q + (size_t)q * 32)[0];
    qb = reinterpret_cast<const uint4*>(Q + (size_t)q * 32)[1];
  }
  uint32_t b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
  int t_begin = 0, t_end = nt;
  if (SPLIT) {
    const int chunk = ((nt + (int)gridDim.z - 1) / (int)gridDim.z + kTrainTile - 1) / kTrainTile * kTrainTile;
    t_begin = min((int)blockIdx.z * chunk, nt);
    t_end = min(t_begin + chunk, nt);
  }
  for (int t0 = t_begin; t0 < t_end; t0 += kTrainTile) {
    const int cnt = min(kTrainTile, t_end - t0);
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
      const uint4* src = reinterpret_cast<const uint4*>(T + (size_t)(t0 + threadIdx.x) * 32);
      tile[2 * threadIdx.x] = src[0];
      tile[2 * threadIdx.x + 1] = src[1];
    }
    __syncthreads();
    for (int j = 0; j < cnt; j++) {
      const uint4 ta = tile[2 * j], tb = tile[2 * j + 1];
      uint32_t d = __popc(qa.x ^ ta.x);
      d += __popc(qa.y ^ ta.y);
      d += __popc(qa.z ^ ta.z);
      d += __popc(qa.w ^ ta.w);
      d += __popc(qb.x ^ tb.x);
      d += __popc(qb.y ^ tb.y);
      d += __popc(qb.z ^ tb.z);
      d += __popc(qb.w ^ tb.w);
      const uint32_t key = (d << 20) | (uint32_t)(t0 + j);
      b2 = min(b2, max(b1, key));
      b1 = min(b1, key);
    }
  }
  if (SPLIT) {
    if (q < nq && t_begin < t_end) {
      unsigned long long* slot = reinterpret_cast<unsigned long long*>(dist2) + (size_t)pair * max_rows + q;
      unsigned long long seen = *slot;
      while (true) {
        const uint32_t a1 = (uint32_t)(seen >> 32), a2 = (uint32_t)seen;
        const uint32_t m1 = min(a1, b1), m2 = min(max(a1, b1), min(a2, b2));
        const unsigned long long merged = ((unsigned long long)m1 << 32) | m2;
        if (merged == seen) break;
        const unsigned long long prev = atomicCAS(slot, seen, merged);
        if (prev == seen) break;
        seen = prev;
      }
    }
    return;
  }
  if (q < nq) {
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
                     hipStream_t s) {
  const int qtiles = (max_rows + 255) / 256;
  // Aim at ~8000 workgroups: fewer (a single pair of one frame at a time, but also 128 stereo pairs = 1152 query tiles)
  // leave SIMDs with one or two waves, too few to hide the LDS latency of the distance loop; the train sets are then
  // split (at least two tiles per chunk) and merged through the packed key pairs.
  int nsplit = 1;
  if ((long)qtiles * n_pairs < 8192) nsplit = (int)std::min<long>(16, std::max<long>(1, 8192 / ((long)qtiles * n_pairs)));
  nsplit = std::min(nsplit, std::max(1, max_rows / (2 * kTrainTile)));
  if (nsplit <= 1) {
    hipLaunchKernelGGL(knn2_kernel<false>, dim3(qtiles, n_pairs, 1), dim3(256), 0, s, d_desc, d_counts, set_stride, d_q_set,
                       d_t_set, max_rows, d_idx2, d_dist2);
    return;
  }
  (void)hipMemsetAsync(d_dist2, 0xFF, (size_t)n_pairs * max_rows * 2 * sizeof(int32_t), s);
  hipLaunchKernelGGL(knn2_kernel<true>, dim3(qtiles, n_pairs, nsplit), dim3(256), 0, s, d_desc, d_counts, set_stride,
                     d_q_set, d_t_set, max_rows, d_idx2, d_dist2);
  hipLaunchKernelGGL(knn2_finalize_kernel, dim3(qtiles, n_pairs, 1), dim3(256), 0, s, d_counts, d_q_set, max_rows, d_idx2,
                     d_dist2);
}

void vsf_launch_ratio_compact(const int32_t* d_counts, const int32_t* d_q_set, const int32_t* d_t_set, int n_pairs,
                              int max_rows, const int32_t* d_idx2, const int32_t* d_dist2, uint32_t ratio_num,
                              uint32_t ratio_shift, vsf_dmatch* d_matches, int32_t* d_nmatches, int32_t* d_status,
                              hipStream_t s) {
  (void)d_status;
  hipLaunchKernelGGL(ratio_compact_kernel, dim3(n_pairs), dim3(256), 0, s, d_counts, d_q_set, d_t_set, max_rows,
                     d_idx2, d_dist2, ratio_num, ratio_shift, d_matches, d_nmatches);
}
