// k_frontend.hip -- SURVEY.md section 8(f) row f1: the two reference-code steps between the matcher and the outputs,
// on the device, for batches of frames resident in HBM:
//
//   Frontend::RemoveAmbigStereo  (slam_frontend.cc:353-398): epipolar residual |l^T F r| of every stereo match
//       (float, products accumulated left to right as Eigen's 1x3 * 3x3 * 3x1 does), its mean over ALL matches of
//       the frame summed sequentially in match order (float addition is not associative), the threshold chain
//       thr[k] = mean[k-1] + 2 (static stereo_ambig_constraint, cc:353, :392-394; a frame without matches makes it
//       0/0 + 2 = NaN for exactly the next frame, which then keeps nothing -- SURVEY quirk Q3, reproduced), and the
//       rebuild of both frames from the surviving pairs in match order (cc:396-397).
//   Frontend::GetFeatureMatches  (slam_frontend.cc:282-309): std::sort of the ratio-tested matches by
//       DMatch::operator< (distance only; unstable, so the permutation is libstdc++'s introsort + insertion sort,
//       restated in vsf_select.h) and the cut to  int(size * best_percent)  pairs (float product, cc:290).
//
// The residual kernel is embarrassingly parallel except for the ordered sum (one lane, 4 values per LDS read); the
// sort runs the sequential restatement on one lane per (query set, train set) pair with the keys in LDS -- pairs are
// independent, a batch sorts them all concurrently.
#include <cstdlib>
#include <cstring>

#include "vsf_internal.h"
#include "vsf_select.h"
#include "vsf_hoare.h"

#pragma clang fp contract(off)

namespace {

// ---- RemoveAmbigStereo, step 1: residuals + ordered mean per frame ----
// A three-term dot product as the reference's Eigen expression sums it (vsf_params::residual_order): Eigen 3.3's unrolled
// reduction of a fixed-size-3 lazy product adds element 0 to the sum of elements 1 and 2 (Redux.h, redux_novec_unroller
// with HalfLength = 1); order 1 is plain left to right.
__device__ __forceinline__ float dot3(int order, float a0, float b0, float a1, float b1, float a2, float b2) {
  const float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  return order ? (p0 + p1) + p2 : p0 + (p1 + p2);
}

__global__ __launch_bounds__(256) void stereo_residual_kernel(const vsf_keypoint* __restrict__ kp,
                                                              const vsf_dmatch* __restrict__ matches,
                                                              const int32_t* __restrict__ nmatches, int max_rows,
                                                              const float* __restrict__ F,  // 9 floats, row major, or
                                                              VsfF9 Fv,  // NULL: the matrix by value (batched *_dev calls)
                                                              int order,  // vsf_params::residual_order
                                                              float* __restrict__ residual,  // [frames][max_rows]
                                                              float* __restrict__ mean,      // [frames], NaN if empty
                                                              int lds_rows) {
  // lds_rows >= max_rows: the residuals are kept in LDS for the one thread that adds them up in match order (the
  // reference's float accumulation); read back from memory, every term of that chain was a trip to the L2
  extern __shared__ __attribute__((aligned(16))) float s_res[];
  const int f = blockIdx.x;
  const int n = min(nmatches[f], max_rows);
  const vsf_keypoint* left = kp + (size_t)(2 * f) * max_rows;
  const vsf_keypoint* right = kp + (size_t)(2 * f + 1) * max_rows;
  const vsf_dmatch* m = matches + (size_t)f * max_rows;
  float* res = residual + (size_t)f * max_rows;
  const bool in_lds = lds_rows >= max_rows;
  float Fm[9];
#pragma unroll
  for (int i = 0; i < 9; i++) Fm[i] = F ? F[i] : Fv.v[i];
  for (int i = threadIdx.x; i < n; i += 256) {
    const vsf_dmatch dm = m[i];
    const float lx = left[dm.queryIdx].x, ly = left[dm.queryIdx].y;
    const float rx = right[dm.trainIdx].x, ry = right[dm.trainIdx].y;
    float t[3];
#pragma unroll
    for (int j = 0; j < 3; j++) t[j] = dot3(order, lx, Fm[0 * 3 + j], ly, Fm[1 * 3 + j], 1.0f, Fm[2 * 3 + j]);
    const float r = fabsf(dot3(order, t[0], rx, t[1], ry, t[2], 1.0f));
    res[i] = r;
    if (in_lds) s_res[i] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float avg = 0.0f;  // avg_constraint += constraint, in match order
    if (in_lds) {
      int i = 0;
#pragma unroll 4
      for (; i + 4 <= n; i += 4) {  // (the loads run ahead of the chain of additions)
        const float4 v = *reinterpret_cast<const float4*>(s_res + i);
        avg += v.x;
        avg += v.y;
        avg += v.z;
        avg += v.w;
      }
      for (; i < n; i++) avg += s_res[i];
    } else {
      for (int i = 0; i < n; i++) avg += res[i];
    }
    // 0.0f / 0 on the reference's x86 is the default ("real indefinite") NaN, sign bit set; the bits travel into the
    // threshold and the gathered records, so they are reproduced rather than left to this GPU's own default NaN
    mean[f] = n > 0 ? avg / (float)n : __uint_as_float(0xFFC00000u);
  }
}

// ---- step 2: thr[0] = thr_in, thr[k] = mean[k-1] + 2 (NaN when frame k-1 had no match: frame k then keeps nothing, and
// frame k+1 is back to a finite threshold, exactly as the reference's static behaves); thr[n] = value after the batch ----
__global__ void stereo_threshold_chain_kernel(const float* __restrict__ mean, int n, float thr_in,
                                              float* __restrict__ thr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float cur = thr_in;
  for (int k = 0; k < n; k++) {
    thr[k] = cur;
    cur = mean[k] + 2.0f;  // padding_from_average, cc:392-394; NaN after a frame without matches (quirk Q3)
  }
  thr[n] = cur;
}

// ---- step 3: keep residual <= thr, rebuild both frames in match order ----
__global__ __launch_bounds__(256) void stereo_filter_kernel(const vsf_keypoint* __restrict__ kp,
                                                            const uint8_t* __restrict__ desc,
                                                            const vsf_dmatch* __restrict__ matches,
                                                            const int32_t* __restrict__ nmatches, int max_rows,
                                                            const float* __restrict__ residual,
                                                            const float* __restrict__ thr,
                                                            vsf_keypoint* __restrict__ kp_out,   // [2*frames][max_rows]
                                                            uint8_t* __restrict__ desc_out,      // [2*frames][max_rows][32]
                                                            int32_t* __restrict__ counts_out,    // [2*frames]
                                                            // the ObserveImage queue: the rebuilt descriptors of frame f go
                                                            // to the sets out_sets[2f] (left) / out_sets[2f + 1] (right) of
                                                            // desc_out, their number to set_counts[set] as well
                                                            const int32_t* __restrict__ out_sets,
                                                            int32_t* __restrict__ set_counts) {
  __shared__ int wsum[4];
  __shared__ int s_base;
  const int f = blockIdx.x;
  const int n = min(nmatches[f], max_rows);
  const float th = thr[f];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t L = (size_t)(2 * f) * max_rows, R = (size_t)(2 * f + 1) * max_rows;
  const size_t DL = out_sets ? (size_t)out_sets[2 * f] * max_rows : L, DR = out_sets ? (size_t)out_sets[2 * f + 1] * max_rows : R;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    bool keep = false;
    vsf_dmatch dm{0, 0, 0, 0.f};
    if (i < n) {
      dm = matches[(size_t)f * max_rows + i];
      keep = residual[(size_t)f * max_rows + i] <= th;
    }
    const uint64_t b = __ballot(keep);
    const int within = __popcll(b & ((1ull << lane) - 1));
    if (lane == 0) wsum[wid] = __popcll(b);
    __syncthreads();
    int base = s_base;
    for (int w = 0; w < wid; w++) base += wsum[w];
    if (keep) {
      const int o = base + within;
      kp_out[L + o] = kp[L + dm.queryIdx];
      kp_out[R + o] = kp[R + dm.trainIdx];
      const uint4* ls = reinterpret_cast<const uint4*>(desc + (L + dm.queryIdx) * VSF_DESC_BYTES);
      const uint4* rs = reinterpret_cast<const uint4*>(desc + (R + dm.trainIdx) * VSF_DESC_BYTES);
      uint4* ld = reinterpret_cast<uint4*>(desc_out + (DL + o) * VSF_DESC_BYTES);
      uint4* rd = reinterpret_cast<uint4*>(desc_out + (DR + o) * VSF_DESC_BYTES);
      ld[0] = ls[0];
      ld[1] = ls[1];
      rd[0] = rs[0];
      rd[1] = rs[1];
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts_out[2 * f] = s_base;
    counts_out[2 * f + 1] = s_base;
    if (out_sets) {
      set_counts[out_sets[2 * f]] = s_base;
      set_counts[out_sets[2 * f + 1]] = s_base;
    }
  }
}

// ---- RemoveAmbigStereo of ONE frame as one launch (the synchronous ObserveImage: a batch of one): the three steps above
// back to back in one workgroup -- residuals into LDS, the ordered mean and the threshold hand-over by one thread, the
// filter -- with the same float operations in the same order; two dependent launches (~5 us each) fewer in a chain that is
// nothing but launch latencies.  thr_state: the threshold in force before the frame, advanced to mean + 2 (cc:392-394). ----
__global__ __launch_bounds__(256) void stereo_one_frame_kernel(const vsf_keypoint* __restrict__ kp,
                                                               const uint8_t* __restrict__ desc,
                                                               const vsf_dmatch* __restrict__ matches,
                                                               const int32_t* __restrict__ nmatches, int max_rows, VsfF9 Fv,
                                                               int order, float* __restrict__ mean, float* __restrict__ thr,
                                                               float* __restrict__ thr_state,
                                                               vsf_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out,
                                                               int32_t* __restrict__ counts_out,
                                                               const int32_t* __restrict__ out_sets,
                                                               int32_t* __restrict__ set_counts) {
  extern __shared__ __attribute__((aligned(16))) float s_res[];
  __shared__ int wsum[4];
  __shared__ int s_base;
  __shared__ float s_thr;
  const int n = min(nmatches[0], max_rows);
  const vsf_keypoint* left = kp;
  const vsf_keypoint* right = kp + (size_t)max_rows;
  float Fm[9];
#pragma unroll
  for (int i = 0; i < 9; i++) Fm[i] = Fv.v[i];
  for (int i = threadIdx.x; i < n; i += 256) {
    const vsf_dmatch dm = matches[i];
    const float lx = left[dm.queryIdx].x, ly = left[dm.queryIdx].y;
    const float rx = right[dm.trainIdx].x, ry = right[dm.trainIdx].y;
    float t[3];
#pragma unroll
    for (int j = 0; j < 3; j++) t[j] = dot3(order, lx, Fm[0 * 3 + j], ly, Fm[1 * 3 + j], 1.0f, Fm[2 * 3 + j]);
    s_res[i] = fabsf(dot3(order, t[0], rx, t[1], ry, t[2], 1.0f));
  }
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    float avg = 0.0f;  // avg_constraint += constraint, in match order
    int i = 0;
#pragma unroll 4
    for (; i + 4 <= n; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(s_res + i);
      avg += v.x;
      avg += v.y;
      avg += v.z;
      avg += v.w;
    }
    for (; i < n; i++) avg += s_res[i];
    const float m = n > 0 ? avg / (float)n : __uint_as_float(0xFFC00000u);  // (the reference's 0.0f / 0: see above)
    mean[0] = m;
    const float th = *thr_state;
    thr[0] = th;
    *thr_state = m + 2.0f;
    s_thr = th;
  }
  __syncthreads();
  const float th = s_thr;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t L = 0, R = (size_t)max_rows;
  const size_t DL = out_sets ? (size_t)out_sets[0] * max_rows : L, DR = out_sets ? (size_t)out_sets[1] * max_rows : R;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    bool keep = false;
    vsf_dmatch dm{0, 0, 0, 0.f};
    if (i < n) {
      dm = matches[i];
      keep = s_res[i] <= th;
    }
    const uint64_t b = __ballot(keep);
    const int within = __popcll(b & ((1ull << lane) - 1));
    if (lane == 0) wsum[wid] = __popcll(b);
    __syncthreads();
    int base = s_base;
    for (int w = 0; w < wid; w++) base += wsum[w];
    if (keep) {
      const int o = base + within;
      kp_out[L + o] = kp[L + dm.queryIdx];
      kp_out[R + o] = kp[R + dm.trainIdx];
      const uint4* ls = reinterpret_cast<const uint4*>(desc + (L + dm.queryIdx) * VSF_DESC_BYTES);
      const uint4* rs = reinterpret_cast<const uint4*>(desc + (R + dm.trainIdx) * VSF_DESC_BYTES);
      uint4* ld = reinterpret_cast<uint4*>(desc_out + (DL + o) * VSF_DESC_BYTES);
      uint4* rd = reinterpret_cast<uint4*>(desc_out + (DR + o) * VSF_DESC_BYTES);
      ld[0] = ls[0];
      ld[1] = ls[1];
      rd[0] = rs[0];
      rd[1] = rs[1];
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts_out[0] = s_base;
    counts_out[1] = s_base;
    if (out_sets) {
      set_counts[out_sets[0]] = s_base;
      set_counts[out_sets[1]] = s_base;
    }
  }
}

// ---- GetFeatureMatches: std::sort by distance + cut to int(n * best_percent) ----
struct SortKey {
  uint32_t dist;  // Hamming distance (DMatch::distance is (float)int, so the order is the integers')
  uint32_t qt;    // queryIdx | trainIdx << 16
};
struct SortLess {
  __device__ bool operator()(const SortKey& a, const SortKey& b) const { return a.dist < b.dist; }
};

__global__ __launch_bounds__(64) void sort_trim_kernel(const vsf_dmatch* __restrict__ matches,
                                                       const int32_t* __restrict__ nmatches, int max_rows,
                                                       float best_percent,
                                                       const float* __restrict__ best_percent_of,  // per pair, or null
                                                       SortKey* __restrict__ scratch,
                                                       uint64_t* __restrict__ pairs,  // [pairs][max_rows][2]
                                                       int32_t* __restrict__ npairs) {
  extern __shared__ SortKey keys[];  // lds_rows entries, or unused when the pair does not fit
  const int p = blockIdx.x, lane = threadIdx.x;
  const int n = min(nmatches[p], max_rows);
  const vsf_dmatch* m = matches + (size_t)p * max_rows;
  SortKey* a = scratch + (size_t)p * max_rows;  // HBM copy (only used when n exceeds the LDS array)
  const bool in_lds = n <= (int)(VSF_SORT_LDS_ROWS);
  SortKey* arr = in_lds ? keys : a;
  for (int i = lane; i < n; i += 64) {
    const vsf_dmatch dm = m[i];
    arr[i] = SortKey{(uint32_t)(int)dm.distance, (uint32_t)dm.queryIdx | ((uint32_t)dm.trainIdx << 16)};
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) vsf_sel::sort_(arr, n, SortLess());
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // const int num_good_matches = matches.size() * config_.best_percent_;   (size_t -> float, float product, -> int)
  if (best_percent_of) best_percent = best_percent_of[p];
  int good = (int)((float)(size_t)n * best_percent);
  good = min(max(good, 0), n);
  uint64_t* out = pairs + (size_t)p * max_rows * 2;
  for (int i = lane; i < good; i += 64) {
    const SortKey k = arr[i];
    out[2 * i] = (uint64_t)(k.qt & 0xFFFFu);      // FeatureMatch::feature_idx_initial  = queryIdx (cc:295)
    out[2 * i + 1] = (uint64_t)(k.qt >> 16);      // FeatureMatch::feature_idx_current  = trainIdx (cc:296)
  }
  if (lane == 0) npairs[p] = good;
}

// ---- the same, data-parallel: libstdc++'s introsort restated on a whole workgroup ----
// std::sort = __introsort_loop (median-of-3 pivot, __unguarded_partition, recursion on the right part, loop on the left,
// ranges of <= 16 elements left alone, heap sort once 2 lg n levels are used up) + __final_insertion_sort.  The partitions
// of a level work on disjoint ranges and a partition's swaps depend on nothing but its own range, so the order in which
// ranges are processed does not matter; and the final insertion sort never moves an element out of the <= 16-element block
// the partitions left it in (everything in an earlier block is <= everything in a later one, and equal elements do not
// pass each other), so it is an independent stable insertion sort per block.  Hence:
//   A  ranges of more than 256 elements: one Hoare pass each by the whole workgroup (vsf_hoare.h hoare_pass: ballots,
//      block scan, rank -> position tables, pairwise swaps), an explicit stack of ranges in LDS;
//   B  ranges of <= 256 elements: the waves take them from a queue and each wave finishes its range alone (wave_hoare_pass,
//      no workgroup barriers), noting the blocks it leaves;
//   C  one LANE per block: insertion sort.
// Same comparisons on the same elements as the sequential code => the same permutation (tests: random, tie-heavy, sorted,
// reversed, organ-pipe and median-of-3-killer inputs against std::sort through vsf_debug_sort_trim).
// The one-lane kernel above took 90 us for the eleven pairs of a 2000-feature frame and 1 ms for the 8000-row right->left
// match of a 1080p frame.
constexpr int kSortQueue = 1024;     // small ranges waiting for a wave
constexpr int kSortStack = 192;      // phase A's stack of large ranges
constexpr int kSortWaveStack = 40;   // a wave's stack inside its range (depth <= 2 lg 256 + slack)
constexpr int kSortWaveBlocks = 160; // blocks a wave may leave per range (a range of <= 256 elements has <= 128 of >= 2)

struct SortRange {
  int first, last, depth;
};

struct SortShared {  // fixed-size part of the kernel's LDS (after the variable arrays)
  vsf_par::PassCtl ctl;
  SortRange stack[kSortStack];
  SortRange queue[kSortQueue];
  int sp, qn, qhead, overflow;
};

template <int NT>
__global__ __launch_bounds__(NT) void sort_trim_par_kernel(const vsf_dmatch* __restrict__ matches,
                                                           const int32_t* __restrict__ nmatches, int max_rows, int cap,
                                                           float best_percent,
                                                           const float* __restrict__ best_percent_of,
                                                           uint64_t* __restrict__ pairs, int32_t* __restrict__ npairs) {
  using namespace vsf_par;
  extern __shared__ __attribute__((aligned(16))) uint8_t sort_lds[];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NT / 64;
  const int n = min(min(nmatches[p], max_rows), cap);
  // carve the LDS
  const int maxw = (cap + 63) / 64;
  uint8_t* q = sort_lds;
  SortKey* A = reinterpret_cast<SortKey*>(q);
  q += (size_t)cap * sizeof(SortKey);
  PassMem<uint16_t> pm;
  pm.maskL = reinterpret_cast<unsigned long long*>(q);
  q += (size_t)maxw * 8;
  pm.maskR = reinterpret_cast<unsigned long long*>(q);
  q += (size_t)maxw * 8;
  pm.preL = reinterpret_cast<int*>(q);
  q += (size_t)maxw * 4;
  pm.preR = reinterpret_cast<int*>(q);
  q += (size_t)maxw * 4;
  pm.Lp = reinterpret_cast<uint16_t*>(q);
  q += (size_t)cap * 2;
  pm.Rp = reinterpret_cast<uint16_t*>(q);
  q += (size_t)cap * 2;
  q = reinterpret_cast<uint8_t*>(((uintptr_t)q + 15) & ~(uintptr_t)15);
  SortShared& S = *reinterpret_cast<SortShared*>(q);
  q += sizeof(SortShared);
  uint16_t* wtab = reinterpret_cast<uint16_t*>(q + (size_t)wave * kWaveTable);
  q += (size_t)NW * kWaveTable;
  SortRange* wstack = reinterpret_cast<SortRange*>(q) + (size_t)wave * kSortWaveStack;
  q += (size_t)NW * kSortWaveStack * sizeof(SortRange);
  uint16_t* wblocks = reinterpret_cast<uint16_t*>(q) + (size_t)wave * kSortWaveBlocks * 2;
  pm.maxw = maxw;
  pm.c = &S.ctl;
  pm.wbuf = nullptr;
  const vsf_dmatch* m = matches + (size_t)p * max_rows;
  for (int i = tid; i < n; i += NT) {
    const vsf_dmatch dm = m[i];
    A[i] = SortKey{(uint32_t)(int)dm.distance, (uint32_t)dm.queryIdx | ((uint32_t)dm.trainIdx << 16)};
  }
  if (tid == 0) {
    S.sp = 0;
    S.qn = 0;
    S.qhead = 0;
    S.overflow = 0;
    if (n > 1) {
      S.stack[0] = SortRange{0, n, vsf_sel::lg_(n) * 2};
      S.sp = 1;
    }
  }
  __syncthreads();
  const SortLess less;
  // ---- A: large ranges, one workgroup pass each ----
  while (true) {
    const int sp = S.sp;
    if (sp == 0) break;  // (uniform: S.sp only changes between barriers)
    const SortRange r = S.stack[sp - 1];
    __syncthreads();
    if (r.last - r.first <= kWaveCutoff) {
      if (tid == 0) {
        S.sp = sp - 1;
        if (r.last - r.first > 1) {
          if (S.qn < kSortQueue)
            S.queue[S.qn++] = r;
          else
            S.overflow = 1, vsf_sel::sort_from_(A, r.first, r.last, r.depth, less);  // (never seen: kept for safety)
        }
      }
      __syncthreads();
      continue;
    }
    if (r.depth == 0) {  // std::__partial_sort(first, last, last): heap sort, sequential (adversarial inputs only)
      if (tid == 0) {
        vsf_sel::heap_select_(A, r.first, r.last, r.last, less);
        vsf_sel::sort_heap_(A, r.first, r.last, less);
        S.sp = sp - 1;
      }
      __syncthreads();
      continue;
    }
    if (tid == 0) {
      const int mid = r.first + (r.last - r.first) / 2;
      vsf_sel::move_median_to_first_(A, r.first, r.first + 1, mid, r.last - 1, less);
    }
    __syncthreads();
    const SortKey pivot = A[r.first];
    hoare_pass<NT>(
        A, r.first + 1, r.last, [&](const SortKey& x) { return !less(x, pivot); },
        [&](const SortKey& x) { return !less(pivot, x); }, pm);
    if (tid == 0) {
      const int cut = S.ctl.st[4];
      S.stack[sp - 1] = SortRange{r.first, cut, r.depth - 1};
      if (sp < kSortStack) {
        S.stack[sp] = SortRange{cut, r.last, r.depth - 1};
        S.sp = sp + 1;
      } else {  // (the stack cannot grow past ~2 lg n entries; kept for safety)
        S.overflow = 1;
        vsf_sel::sort_from_(A, cut, r.last, r.depth - 1, less);
      }
    }
    __syncthreads();
  }
  // ---- B: small ranges, one wave each ----
  while (true) {
    int qi = 0;
    if (lane == 0) qi = atomicAdd(&S.qhead, 1);
    qi = __builtin_amdgcn_readfirstlane(qi);
    if (qi >= S.qn) break;
    int wsp = 1, nblocks = 0;
    wstack[0] = S.queue[qi];
    wave_fence();
    while (wsp > 0) {
      SortRange r = wstack[wsp - 1];
      --wsp;
      bool sorted = false;
      while (r.last - r.first > 16) {
        if (r.depth == 0) {
          if (lane == 0) {
            vsf_sel::heap_select_(A, r.first, r.last, r.last, less);
            vsf_sel::sort_heap_(A, r.first, r.last, less);
          }
          wave_fence();
          sorted = true;
          break;
        }
        --r.depth;
        // std::__move_median_to_first(first, first + 1, mid, last - 1): uniform reads, one lane stores the swap
        const int pa = r.first + 1, pb = r.first + (r.last - r.first) / 2, pc = r.last - 1;
        const SortKey vf = A[r.first], va = A[pa], vb = A[pb], vc = A[pc];
        int pmid;
        if (less(va, vb))
          pmid = less(vb, vc) ? pb : less(va, vc) ? pc : pa;
        else
          pmid = less(va, vc) ? pa : less(vb, vc) ? pc : pb;
        const SortKey pivot = pmid == pa ? va : pmid == pb ? vb : vc;
        wave_fence();
        if (lane == 0) {
          A[r.first] = pivot;
          A[pmid] = vf;
        }
        wave_fence();
        int total_r, cut;
        wave_hoare_pass(
            A, r.first + 1, r.last, [&](const SortKey& x) { return !less(x, pivot); },
            [&](const SortKey& x) { return !less(pivot, x); }, total_r, cut, wtab);
        if (wsp < kSortWaveStack) {
          if (lane == 0) wstack[wsp] = SortRange{cut, r.last, r.depth};
          ++wsp;
        } else {  // (cannot happen: depth <= 2 lg n bounds the stack)
          if (lane == 0) vsf_sel::sort_from_(A, cut, r.last, r.depth, less);
        }
        wave_fence();
        r.last = cut;
      }
      if (!sorted && r.last - r.first > 1) {
        if (nblocks < kSortWaveBlocks) {
          if (lane == 0) {
            wblocks[2 * nblocks] = (uint16_t)r.first;
            wblocks[2 * nblocks + 1] = (uint16_t)r.last;
          }
          ++nblocks;
        } else {
          if (lane == 0) vsf_sel::insertion_sort_(A, r.first, r.last, less);
        }
      }
    }
    wave_fence();
    // ---- C: the final insertion sort, block by block, one lane each ----
    for (int b = lane; b < nblocks; b += 64) vsf_sel::insertion_sort_(A, (int)wblocks[2 * b], (int)wblocks[2 * b + 1], less);
    wave_fence();
  }
  __syncthreads();
  // const int num_good_matches = matches.size() * config_.best_percent_;   (size_t -> float, float product, -> int)
  if (best_percent_of) best_percent = best_percent_of[p];
  int good = (int)((float)(size_t)n * best_percent);
  good = min(max(good, 0), n);
  uint64_t* out = pairs + (size_t)p * max_rows * 2;
  for (int i = tid; i < good; i += NT) {
    const SortKey k = A[i];
    out[2 * i] = (uint64_t)(k.qt & 0xFFFFu);      // FeatureMatch::feature_idx_initial  = queryIdx (cc:295)
    out[2 * i + 1] = (uint64_t)(k.qt >> 16);      // FeatureMatch::feature_idx_current  = trainIdx (cc:296)
  }
  if (tid == 0) npairs[p] = good;
}

// ---- the ObserveImage queue: everything one ObserveImage returns, compact, written straight into pinned host memory, for
// every frame of a batch (blockIdx.z) ----
// header (16 words) | npairs[n_pairs] padded to 4 words | VisionFeature x nfeat | FeatureMatch x sum(npairs) |
// cv::KeyPoint x nfeat (the filtered left frame) | descriptors x nfeat.  The pairs of frame f in the batch's pair list:
// its temporal factors at tp0 .. tp0 + n_past - 1 (oldest kept frame first), the right -> left matches of
// Calculate3DPoints at index f; they leave in the order the reference books them (temporal first, cc:424-437).
__global__ __launch_bounds__(256) void observe_pack_kernel(VsfObserveArgs a) {
  __shared__ uint32_t s_off[4 + VSF_OBSERVE_MAX_PAIRS];
  const int K = a.max_rows, f = blockIdx.z;
  const VsfObserveFrame fm = a.frames[f];
  const int n_pairs = fm.n_past + 1;
  const int sec = blockIdx.y;
  if (sec >= 3 + n_pairs) return;
  const int nfeat = min(max(a.counts_f[2 * f], 0), K);
  auto pair_index = [&](int p) { return p < fm.n_past ? fm.tp0 + p : f; };
  if (threadIdx.x == 0) {
    uint32_t off = 64u + 4u * (uint32_t)((n_pairs + 3) & ~3);
    s_off[0] = off;  // features
    off += (uint32_t)nfeat * 28u;
    for (int p = 0; p < n_pairs; p++) {
      s_off[3 + p] = off;
      off += (uint32_t)min(max(a.npairs[pair_index(p)], 0), K) * 16u;
    }
    s_off[1] = off;  // keypoints
    off += (uint32_t)nfeat * 28u;
    s_off[2] = off;  // descriptors
    off += (uint32_t)nfeat * 32u;
    s_off[3 + n_pairs] = off;  // total
  }
  __syncthreads();
  const uint32_t total = s_off[3 + n_pairs];
  uint32_t* out = reinterpret_cast<uint32_t*>(a.out + (size_t)fm.out_slot * a.out_stride);
  if (blockIdx.x == 0 && sec == 0) {
    if (threadIdx.x == 0) {
      out[1] = (uint32_t)n_pairs;
      out[2] = (uint32_t)nfeat;
      out[3] = total;
      out[4] = (uint32_t)a.counts_raw[2 * f];
      out[5] = (uint32_t)a.counts_raw[2 * f + 1];
      out[6] = (uint32_t)a.nmatches[f];
      out[7] = (uint32_t)a.npoints[f];
      out[8] = __float_as_uint(a.means[f]);
      out[9] = __float_as_uint(a.thr[f]);
      out[10] = __float_as_uint(a.means[f] + 2.0f);  // the static's value after this frame (cc:392-394)
      out[11] = total > a.out_cap ? 1u : 0u;
      // capacity overflows of THIS frame's two extractions (a status word per image), read and cleared
      out[12] = (uint32_t)((a.status[2 * f] | a.status[2 * f + 1]) & 1);
      a.status[2 * f] = 0;
      a.status[2 * f + 1] = 0;
      out[13] = out[14] = out[15] = 0u;
      out[0] = 0x4F465356u;  // "VSFO"
    }
    if ((int)threadIdx.x < ((n_pairs + 3) & ~3))
      out[16 + threadIdx.x] =
          (int)threadIdx.x < n_pairs ? (uint32_t)min(max(a.npairs[pair_index((int)threadIdx.x)], 0), K) : 0u;
  }
  if (total > a.out_cap) return;
  const uint32_t* src;
  uint32_t words;
  if (sec == 0) {
    src = reinterpret_cast<const uint32_t*>(a.features + (size_t)f * K);
    words = (uint32_t)nfeat * 7u;
  } else if (sec == 1) {
    src = reinterpret_cast<const uint32_t*>(a.kp_f + (size_t)(2 * f) * K);
    words = (uint32_t)nfeat * 7u;
  } else if (sec == 2) {
    src = reinterpret_cast<const uint32_t*>(a.desc_sets + (size_t)fm.left_set * K * VSF_DESC_BYTES);
    words = (uint32_t)nfeat * 8u;
  } else {
    const int p = pair_index(sec - 3);
    src = reinterpret_cast<const uint32_t*>(a.pairs + (size_t)p * K * 2);
    words = (uint32_t)min(max(a.npairs[p], 0), K) * 4u;
  }
  uint32_t* dst = out + s_off[sec] / 4u;
  for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < words; w += gridDim.x * 256u) dst[w] = src[w];
}

}  // namespace

void vsf_launch_stereo_residuals(const vsf_keypoint* d_kp, const vsf_dmatch* d_matches, const int32_t* d_nmatches,
                                 int n_frames, int max_rows, const float* d_F, const float* h_F, int order, float* d_residual,
                                 float* d_mean, hipStream_t s) {
  const int lds_rows = max_rows <= 16000 ? (max_rows + 3) & ~3 : 0;  // (64 KB of LDS without asking for more)
  VsfF9 fv = {};
  if (h_F)  // the matrix rides in the kernel arguments: no copy command, no device buffer shared between calls
    for (int i = 0; i < 9; i++) fv.v[i] = h_F[i];
  hipLaunchKernelGGL(stereo_residual_kernel, dim3(n_frames), dim3(256), (size_t)lds_rows * sizeof(float), s, d_kp,
                     d_matches, d_nmatches, max_rows, h_F ? nullptr : d_F, fv, order, d_residual, d_mean, lds_rows);
}

void vsf_launch_stereo_filter_only(const vsf_keypoint* d_kp, const uint8_t* d_desc, const vsf_dmatch* d_matches,
                                   const int32_t* d_nmatches, int n_frames, int max_rows, const float* d_residual,
                                   const float* d_thr, vsf_keypoint* d_kp_out, uint8_t* d_desc_out,
                                   int32_t* d_counts_out, hipStream_t s, const int32_t* d_out_sets, int32_t* d_set_counts) {
  hipLaunchKernelGGL(stereo_filter_kernel, dim3(n_frames), dim3(256), 0, s, d_kp, d_desc, d_matches, d_nmatches,
                     max_rows, d_residual, d_thr, d_kp_out, d_desc_out, d_counts_out, d_out_sets, d_set_counts);
}

bool vsf_launch_stereo_one_frame(const vsf_keypoint* d_kp, const uint8_t* d_desc, const vsf_dmatch* d_matches,
                                 const int32_t* d_nmatches, int max_rows, const float* h_F, int order, float* d_mean,
                                 float* d_thr, float* d_thr_state, vsf_keypoint* d_kp_out, uint8_t* d_desc_out,
                                 int32_t* d_counts_out, const int32_t* d_out_sets, int32_t* d_set_counts, hipStream_t s) {
  if (max_rows > 16000) return false;  // (64 KB of LDS without asking for more: the three launches take such a frame)
  VsfF9 fv;
  for (int i = 0; i < 9; i++) fv.v[i] = h_F[i];
  const int lds_rows = (max_rows + 3) & ~3;
  hipLaunchKernelGGL(stereo_one_frame_kernel, dim3(1), dim3(256), (size_t)lds_rows * sizeof(float), s, d_kp, d_desc, d_matches,
                     d_nmatches, max_rows, fv, order, d_mean, d_thr, d_thr_state, d_kp_out, d_desc_out, d_counts_out, d_out_sets,
                     d_set_counts);
  return true;
}

void vsf_launch_stereo_filter(const vsf_keypoint* d_kp, const uint8_t* d_desc, const vsf_dmatch* d_matches,
                              const int32_t* d_nmatches, int n_frames, int max_rows, const float* d_F, const float* h_F,
                              int order, const float* d_thr_override, float thr_in, float* d_residual, float* d_mean, float* d_thr,
                              vsf_keypoint* d_kp_out, uint8_t* d_desc_out, int32_t* d_counts_out, hipStream_t s) {
  vsf_launch_stereo_residuals(d_kp, d_matches, d_nmatches, n_frames, max_rows, d_F, h_F, order, d_residual, d_mean, s);
  if (!d_thr_override)
    hipLaunchKernelGGL(stereo_threshold_chain_kernel, dim3(1), dim3(64), 0, s, d_mean, n_frames, thr_in, d_thr);
  vsf_launch_stereo_filter_only(d_kp, d_desc, d_matches, d_nmatches, n_frames, max_rows, d_residual,
                                d_thr_override ? d_thr_override : d_thr, d_kp_out, d_desc_out, d_counts_out, s);
}

void vsf_launch_sort_trim(const vsf_dmatch* d_matches, const int32_t* d_nmatches, int n_pairs, int max_rows,
                          float best_percent, const float* d_best_percent_of, void* d_scratch, uint64_t* d_pairs,
                          int32_t* d_npairs, hipStream_t s, bool force_serial, int lds_limit) {
  // force_serial: the one-lane kernel (VSF_OPT_SORT_SERIAL, A/B measurements); it also takes the pairs beyond the parallel
  // kernel's LDS layout and every pair on a device whose workgroups cannot have that much LDS (lds_limit: queried at
  // vsf_create, where vsf_prepare_sort_kernels raised the kernel's limit -- checked -- to it)
  constexpr int NT = 256;
  const int cap = max_rows;
  const size_t maxw = (size_t)(cap + 63) / 64;
  const size_t lds = (size_t)cap * sizeof(SortKey) + maxw * 2 * sizeof(unsigned long long) + maxw * 2 * sizeof(int) +
                     (size_t)cap * 2 * sizeof(uint16_t) + 16 + sizeof(SortShared) +
                     (size_t)(NT / 64) * (vsf_par::kWaveTable + kSortWaveStack * sizeof(SortRange) +
                                         kSortWaveBlocks * 2 * sizeof(uint16_t));
  if (!force_serial && cap < 65536 && lds <= (size_t)std::max(lds_limit, 0)) {
    hipLaunchKernelGGL((sort_trim_par_kernel<NT>), dim3(n_pairs), dim3(NT), lds, s, d_matches, d_nmatches, max_rows, cap,
                       best_percent, d_best_percent_of, d_pairs, d_npairs);
    return;
  }
  hipLaunchKernelGGL(sort_trim_kernel, dim3(n_pairs), dim3(64), VSF_SORT_LDS_ROWS * sizeof(SortKey), s, d_matches,
                     d_nmatches, max_rows, best_percent, d_best_percent_of, reinterpret_cast<SortKey*>(d_scratch), d_pairs,
                     d_npairs);
}

hipError_t vsf_prepare_sort_kernels(int lds_limit) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_trim_par_kernel<256>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit);
}

void vsf_launch_observe_pack(const VsfObserveArgs& a, int max_pairs_per_frame, hipStream_t s) {
  // sections: 0 VisionFeature, 1 cv::KeyPoint, 2 descriptors, 3.. one per pair; up to 16 chunks of workgroups each; z = frame
  hipLaunchKernelGGL(observe_pack_kernel, dim3(a.n_frames > 8 ? 4 : 16, 3 + max_pairs_per_frame, a.n_frames), dim3(256), 0, s, a);
}
