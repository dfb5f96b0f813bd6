// k_select.hip -- K3 + K4 + K5 of ORB's computeKeyPoints (features2d/orb.cpp, reached from
// slam_frontend.cc:274), one workgroup per (image, level):
//
//   gather   the level's FAST candidates (raster order) from the FAST cells' segments (vsf_gather.h)
//   K3       KeyPointsFilter::retainBest(2 * n_l) on the FAST score           (order-exact)
//   K4       HarrisResponses(blockSize 7, k 0.04f) for the survivors           (int32 sums, 6 float ops, no FMA)
//   K5       KeyPointsFilter::retainBest(n_l) on the Harris response          (order-exact)
//   (K6, ICAngles + fastAtan2, runs per keypoint in k_describe.hip)
//
// Order-exact selection.  retainBest leaves libstdc++'s std::nth_element + std::partition
// permutation, which later stages turn into keypoint indices, so it has to be reproduced exactly.  Both are
// built from Hoare-style passes ("the k-th stopper from the left swaps with the k-th stopper from the right
// until they cross"), and such a pass IS data-parallel:
//   * flag every element as left-stopper / right-stopper against the pivot with wave ballots (bit masks),
//   * prefix-popcount the masks (one block scan),
//   * every stopper writes its position at its rank (rank -> position tables), and thread k swaps the k-th
//     left-stopper with the k-th right-stopper from the right while they have not crossed (no searching),
//   * the split point follows from the table entries next to the last swap.
// Pivot choice (median of first+1 / mid / last-1), the depth limit, the heap-select fallback and the final
// insertion sort are the sequential restatement of vsf_select.h; ranges <= 256 elements are finished by a single
// wave (ballot masks in SGPRs, no workgroup barriers), the rare depth-limit / heap-select case by one lane.
// The arrays live in LDS when they fit and in an HBM scratch area otherwise.
#include "vsf_gather.h"
#include <cstdlib>

#include "vsf_internal.h"
#include "vsf_select.h"
#include "vsf_hoare.h"

#pragma clang fp contract(off)

namespace {

struct SelectArgs {
  const VsfLevel* levels;
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;
  uint32_t pyr_bytes;
  const uint32_t* cand;
  uint32_t cand_entries;
  const uint16_t* rowstart;
  int nunits;
  uint32_t* scratch;  // [image][6 * cand_entries]: stage-1 array, stage-2 pairs (x2), rank tables (x2), masks
  VsfLevelKp* lvlkp;
  int lvlkp_entries;
  int32_t* lvl_count;
  int nlevels;
  int level0;  // first level handled by this launch
  int nlv, nimages;  // levels of this launch, images
  int32_t* status;
  int status_stride;  // 0: one word for the call; 1: a word per image
};

struct ScoreGreater {
  __device__ bool operator()(uint32_t a, uint32_t b) const { return (a >> 24) > (b >> 24); }
};
struct ScoreGe {
  __device__ bool operator()(uint32_t a, uint32_t b) const { return (a >> 24) >= (b >> 24); }
};
struct RespGreater {
  __device__ bool operator()(const uint2& a, const uint2& b) const {
    return __uint_as_float(a.x) > __uint_as_float(b.x);
  }
};
struct RespGe {
  __device__ bool operator()(const uint2& a, const uint2& b) const {
    return __uint_as_float(a.x) >= __uint_as_float(b.x);
  }
};

using namespace vsf_par;

// ---- register-resident passes for ranges of <= 64 elements (one per lane) ----
// A selection spends most of its passes on the short ranges at its end, where a pass through LDS is a chain of seven
// round trips for one wave.  Here the range lives in one VGPR (pair) per lane across all remaining passes: the median
// of three comes from v_readlane, partners are found by pushing lane ids to their rank (ds_permute) and elements move
// with ds_bpermute -- three crossbar hops per pass, no memory round trip.
__device__ __forceinline__ uint32_t lane_get(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint2 lane_get(const uint2& v, int l) {
  return make_uint2((uint32_t)__builtin_amdgcn_readlane((int)v.x, l), (uint32_t)__builtin_amdgcn_readlane((int)v.y, l));
}
__device__ __forceinline__ uint32_t lane_pull(uint32_t v, int src) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
}
__device__ __forceinline__ uint2 lane_pull(const uint2& v, int src) {
  return make_uint2((uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v.x),
                    (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v.y));
}

// One Hoare pass on the lanes flagged l (left-stoppers) / r (right-stoppers); returns the new element of this lane.
template <class T>
__device__ __forceinline__ T reg_hoare_pass(const T& x, bool l, bool r, int& totalR, int& cut_lane) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull, le = (2ull << lane) - 1ull;
  const unsigned long long mL = __ballot(l), mR = __ballot(r);
  const int totalL = __popcll(mL);
  totalR = __popcll(mR);
  const int k = __popcll(mL & lt) + 1;                // 1-based rank from the left (when a left-stopper)
  const int r_le = __popcll(mR & le);                 // right-stoppers at <= lane
  const int rr = totalR - r_le + 1;                   // 1-based rank from the right (when a right-stopper)
  const bool part_l = l && (totalR - r_le >= k);
  const int K = __popcll(__ballot(part_l));
  const bool part_r = r && rr <= K;
  // rank -> lane tables, one entry per lane (entries beyond the stopper counts are never read)
  const int l_of_rank = __builtin_amdgcn_ds_permute((l ? k - 1 : 63) << 2, lane);
  const int r_of_rank = __builtin_amdgcn_ds_permute((r ? rr - 1 : 63) << 2, lane);
  const int pl = __builtin_amdgcn_ds_bpermute((k - 1) << 2, r_of_rank);    // partner of a left-stopper of rank k
  const int pr = __builtin_amdgcn_ds_bpermute((rr - 1) << 2, l_of_rank);   // partner of a right-stopper of rank rr
  const int src = part_l ? pl : part_r ? pr : lane;
  const T y = lane_pull(x, src);
  const unsigned long long ha = __ballot(l && k == K + 1), hb = __ballot(r && rr == K);
  const int aK1 = ha ? (int)__builtin_ctzll(ha) : 0x7FFFFFFF;
  const int bK = (hb && K > 0) ? (int)__builtin_ctzll(hb) : 0x7FFFFFFF;
  cut_lane = (K < totalL && (K == 0 || aK1 < bK)) ? aK1 : bK;
  return y;
}

// std::__introselect continued from (first, last, depth), last - first <= 64, A in LDS; runs to the end (including the
// final insertion sort).  Returns false (nothing done) when the depth limit is hit: the caller's LDS path handles that.
template <class T, class Greater>
__device__ __forceinline__ bool reg_introselect(T* A, int first, int last, int nth, int depth, Greater greater) {
  const int lane = threadIdx.x & 63;
  const int base = first, m0 = last - first;
  T x = A[base + min(lane, m0 - 1)];
  while (last - first > 3) {
    if (depth == 0) {  // heap select on the array itself
      if (lane < m0) A[base + lane] = x;
      wave_fence();
      if (lane == 0) {
        vsf_sel::heap_select_(A, first, nth + 1, last, greater);
        vsf_sel::swap_(A[first], A[nth]);
      }
      wave_fence();
      return true;
    }
    --depth;
    const int lf = first - base, la = lf + 1, lb = lf + (last - first) / 2, lc = last - 1 - base;
    const T vf = lane_get(x, lf), va = lane_get(x, la), vb = lane_get(x, lb), vc = lane_get(x, lc);
    int lm;
    if (greater(va, vb))
      lm = greater(vb, vc) ? lb : greater(va, vc) ? lc : la;
    else
      lm = greater(va, vc) ? la : greater(vb, vc) ? lc : lb;
    const T pivot = lm == la ? va : lm == lb ? vb : vc;
    if (lane == lf) x = pivot;
    if (lane == lm) x = vf;
    const bool in = lane > lf && lane <= lc;
    int total_r, cut_lane;
    x = reg_hoare_pass(x, in && !greater(x, pivot), in && !greater(pivot, x), total_r, cut_lane);
    const int cut = base + cut_lane;
    if (cut <= nth)
      first = cut;
    else
      last = cut;
  }
  // std::__insertion_sort on the (<= 3) remaining elements, on uniform copies
  const int n3 = last - first, l0 = first - base;
  if (n3 >= 2) {
    T t0 = lane_get(x, l0), t1 = lane_get(x, l0 + 1);
    if (greater(t1, t0)) {
      const T s = t0;
      t0 = t1;
      t1 = s;
    }
    T t2 = t1;
    if (n3 == 3) {
      const T v = lane_get(x, l0 + 2);
      if (greater(v, t0)) {
        t2 = t1;
        t1 = t0;
        t0 = v;
      } else if (greater(v, t1)) {
        t2 = t1;
        t1 = v;
      } else {
        t2 = v;
      }
    }
    if (lane == l0) x = t0;
    if (lane == l0 + 1) x = t1;
    if (n3 == 3 && lane == l0 + 2) x = t2;
  }
  if (lane < m0) A[base + lane] = x;
  wave_fence();
  return true;
}

// std::__introselect continued from (first, last, depth) on a range of <= kWaveCutoff elements, A in LDS.
template <class T, class Greater>
__device__ __forceinline__ void wave_introselect(T* A, int first, int last, int nth, int depth, Greater greater, uint16_t* wtab) {
  const int lane = threadIdx.x & 63;
  while (last - first > 3) {
    if (last - first <= 64) {  // the rest of the selection runs in registers
      reg_introselect(A, first, last, nth, depth, greater);
      return;
    }
    if (depth == 0) {
      if (lane == 0) {
        vsf_sel::heap_select_(A, first, nth + 1, last, greater);
        vsf_sel::swap_(A[first], A[nth]);
      }
      wave_fence();
      return;
    }
    --depth;
    // std::__move_median_to_first(first, first + 1, mid, last - 1): every lane reads the four elements (uniform
    // addresses, one round trip) and takes the same decision; lane 0 stores the swap
    const int pa = first + 1, pb = first + (last - first) / 2, pc = last - 1;
    const T vf = A[first], va = A[pa], vb = A[pb], vc = A[pc];
    int pm;
    if (greater(va, vb))
      pm = greater(vb, vc) ? pb : greater(va, vc) ? pc : pa;
    else
      pm = greater(va, vc) ? pa : greater(vb, vc) ? pc : pb;
    const T pivot = pm == pa ? va : pm == pb ? vb : vc;
    if (lane == 0) {
      A[first] = pivot;
      A[pm] = vf;
    }
    wave_fence();
    int total_r, cut;
    wave_hoare_pass(
        A, first + 1, last, [&](const T& x) { return !greater(x, pivot); },
        [&](const T& x) { return !greater(pivot, x); }, total_r, cut, wtab);
    if (cut <= nth)
      first = cut;
    else
      last = cut;
  }
  if (lane == 0) vsf_sel::insertion_sort_(A, first, last, greater);
  wave_fence();
}

// Runs f(B, off) on an LDS image B of A[lo, hi) (hi - lo <= kWaveCutoff), B[i - off] = A[i]: A itself when it lives in
// LDS, else the range is staged through the wave scratch (an HBM-resident array pays a memory round trip per access).
template <bool IN_LDS, class T, class F>
__device__ __forceinline__ void with_lds_range(T* A, int lo, int hi, uint8_t* wbuf, F f) {
  if constexpr (IN_LDS) {
    f(A, 0);
  } else {
    const int lane = threadIdx.x & 63;
    T* B = reinterpret_cast<T*>(wbuf + kWaveTable);
    for (int i = lane; i < hi - lo; i += 64) B[i] = A[lo + i];
    wave_fence();
    f(B, lo);
    for (int i = lane; i < hi - lo; i += 64) A[lo + i] = B[i];
    wave_fence();
  }
}

// Where an HBM-resident selection continues once its range fits: an LDS array of `cap` elements and the pass scratch
// that goes with it (buf == nullptr: stay in HBM).
template <class T>
struct LdsStage {
  T* buf;
  int cap;
  const PassMem<uint16_t>* pm;
};

// std::__introselect from the state (first, last, depth) kept in pm.c->st[0..2] -- all threads call this.
template <int NT, class T, class Greater, class PM>
__device__ __forceinline__ void par_introselect(T* A, int nth, Greater greater, const PM& pm, const LdsStage<T>& stage) {
  PassCtl& s = *pm.c;
  const int tid = threadIdx.x;
  while (true) {
    const int first = s.st[0], last = s.st[1], depth = s.st[2];
    if (last - first <= kWaveCutoff || depth == 0 || last - first - 1 > pm.maxw * 64) break;
    if constexpr (sizeof(*pm.Lp) != 2) {
      // An HBM-resident pass pays a memory round trip in each of its phases (~2.5x an LDS pass on the same range), and
      // the range shrinks to a fraction after the first pass or two: the rest of the selection runs on an LDS copy.
      if (stage.buf && last - first <= stage.cap) {
        __syncthreads();
        for (int i = tid; i < last - first; i += NT) stage.buf[i] = A[first + i];
        if (tid == 0) {
          s.st[0] = 0;
          s.st[1] = last - first;
        }
        __syncthreads();
        par_introselect<NT>(stage.buf, nth - first, greater, *stage.pm, LdsStage<T>{nullptr, 0, nullptr});
        for (int i = tid; i < last - first; i += NT) A[first + i] = stage.buf[i];
        __syncthreads();
        return;
      }
    }
    if (tid == 0) {
      const int mid = first + (last - first) / 2;
      vsf_sel::move_median_to_first_(A, first, first + 1, mid, last - 1, greater);
    }
    __syncthreads();
    const T pivot = A[first];
    hoare_pass<NT>(
        A, first + 1, last, [&](const T& x) { return !greater(x, pivot); },
        [&](const T& x) { return !greater(pivot, x); }, pm);
    if (tid == 0) {
      const int cut = s.st[4];
      s.st[2] = depth - 1;
      if (cut <= nth)
        s.st[0] = cut;
      else
        s.st[1] = cut;
    }
    __syncthreads();
  }
  if (tid < 64) {  // wave 0 finishes the range; the other waves wait at the barrier
    const int first = s.st[0], last = s.st[1], depth = s.st[2];
    if (last - first <= kWaveCutoff)
      with_lds_range<sizeof(*pm.Lp) == 2>(A, first, last, pm.wbuf, [&](T* B, int off) {
        wave_introselect(B, first - off, last - off, nth - off, depth, greater, reinterpret_cast<uint16_t*>(pm.wbuf));
      });
    else if (tid == 0)  // depth limit hit on a large range (heap select), or a range beyond the mask scratch
      vsf_sel::introselect_from_(A, first, last, nth, depth, greater);
  }
  __syncthreads();
}

// std::nth_element(A, A + nth, A + n, greater) -- all threads of the workgroup call this.
template <int NT, class T, class Greater, class PM>
__device__ __forceinline__ void par_nth_element(T* A, int n, int nth, Greater greater, const PM& pm, const LdsStage<T>& stage) {
  PassCtl& s = *pm.c;
  if (n == 0 || nth == n) return;
  if (threadIdx.x == 0) {
    s.st[0] = 0;
    s.st[1] = n;
    s.st[2] = vsf_sel::lg_(n) * 2;
  }
  __syncthreads();
  par_introselect<NT>(A, nth, greater, pm, stage);
}

// std::partition(A + lo, A + hi, pred); returns the split point -- all threads call this.
template <int NT, class T, class Pred, class PM>
__device__ __forceinline__ int par_partition(T* A, int lo, int hi, Pred pred, const PM& pm) {
  PassCtl& s = *pm.c;
  const int tid = threadIdx.x;
  if (hi - lo <= kWaveCutoff || hi - lo > pm.maxw * 64) {
    if (tid < 64) {
      if (hi - lo <= kWaveCutoff) {
        int total_r = 0, cut = 0;
        if (hi > lo)
          with_lds_range<sizeof(*pm.Lp) == 2>(A, lo, hi, pm.wbuf, [&](T* B, int off) {
            wave_hoare_pass(
                B, lo - off, hi - off, [&](const T& x) { return !pred(x); }, [&](const T& x) { return pred(x); }, total_r,
                cut, reinterpret_cast<uint16_t*>(pm.wbuf));
          });
        if (tid == 0) s.st[4] = lo + total_r;
      } else if (tid == 0) {
        s.st[4] = vsf_sel::partition_(A, lo, hi, pred);
      }
    }
    __syncthreads();
    const int r = s.st[4];
    __syncthreads();
    return r;
  }
  hoare_pass<NT>(
      A, lo, hi, [&](const T& x) { return !pred(x); }, [&](const T& x) { return pred(x); }, pm);
  const int r = lo + s.st[6];
  __syncthreads();
  return r;
}

// std::partition(A + lo, A + hi, pred) when only the TRUE side survives -- retainBest truncates the array at the returned
// split, so nothing behind it is ever read again.  libstdc++'s bidirectional partition swaps the k-th false element from the
// left with the k-th true element from the right while they have not crossed; with T true elements in [lo, hi) the split
// is lo + T, the K false elements inside the prefix [lo, lo + T) are exactly the ones that get swapped, and their partners
// are the K rightmost true elements, all of them behind the prefix.  So: one READ pass for the flags (ballot masks + a
// block scan of their popcounts, no rank -> position tables), then every false position of the prefix fetches its partner
// (the true element of 0-based rank T - k from the left: a binary search in the prefix counts + a 64-bit rank select) --
// K reads and K writes instead of a full Hoare pass with its table writes and both-way swaps.  On the HBM-resident
// levels this pass alone moved a fifth of the selection's traffic (profiles/traffic.json, round 3).
// Returns the split (all threads, after a barrier).
template <int NT, class T, class Pred, class PM>
__device__ __forceinline__ int par_partition_keep_true(T* A, int lo, int hi, Pred pred, const PM& s) {
  PassCtl& c = *s.c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = hi - lo, nw = (m + 63) >> 6;
  if (m <= kWaveCutoff || nw > s.maxw) return par_partition<NT>(A, lo, hi, pred, s);
  constexpr int U = 4;
  for (int base = 0; base < m; base += NT * U) {
    T x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = base + u * NT + tid;
      if (i < m) x[u] = A[lo + i];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = base + u * NT + tid;
      const unsigned long long bt = __ballot(i < m && pred(x[u]));
      const int w = ((base + u * NT) >> 6) + wave;
      if (lane == 0 && w < nw) s.maskR[w] = bt;
    }
  }
  __syncthreads();
  const int wpt = (nw + NT - 1) / NT, w0 = tid * wpt;
  int mine = 0;
  for (int j = 0; j < wpt; j++)
    if (w0 + j < nw) mine += __popcll(s.maskR[w0 + j]);
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) c.wsum[wave] = (unsigned long long)inc;
  __syncthreads();
  int base_sum = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    if (w < wave) base_sum += (int)c.wsum[w];
    total += (int)c.wsum[w];
  }
  int run = base_sum + inc - mine;
  for (int j = 0; j < wpt; j++)
    if (w0 + j < nw) {
      s.preR[w0 + j] = run;
      run += __popcll(s.maskR[w0 + j]);
    }
  __syncthreads();
  // the prefix [0, total): a false element at offset i is the k-th false from the left, k = i - (true elements before i) + 1
  for (int base = 0; base < total; base += NT) {
    const int i = base + tid;
    if (i < total) {
      const int w = i >> 6, b = i & 63;
      const unsigned long long mk = s.maskR[w];
      if (!((mk >> b) & 1ull)) {
        const int k = i - (s.preR[w] + __popcll(mk & ((1ull << b) - 1ull))) + 1;
        const int rho = total - k;  // 0-based rank, from the left, of the k-th true element from the right
        int a = 0, e = nw - 1;      // last word whose prefix count is <= rho
        while (a < e) {
          const int mid = (a + e + 1) >> 1;
          if (s.preR[mid] <= rho)
            a = mid;
          else
            e = mid - 1;
        }
        A[lo + i] = A[lo + a * 64 + select64(s.maskR[a], rho - s.preR[a])];  // (the partner lies behind the prefix)
      }
    }
  }
  __syncthreads();
  return lo + total;
}

// cv::KeyPointsFilter::retainBest(A[0..n), n_points); returns the new size -- all threads call this.
template <int NT, class T, class Greater, class GreaterEq, class PM>
__device__ __forceinline__ int par_retain_best(T* A, int n, int n_points, Greater greater, GreaterEq ge, const PM& s,
                               const LdsStage<T>& stage = LdsStage<T>{nullptr, 0, nullptr}) {
  if (n_points >= 0 && n > n_points) {
    if (n_points == 0) return 0;
    if (n <= kWaveCutoff) {  // short array: wave 0 runs selection and partition back to back, one barrier at the end
      PassCtl& c = *s.c;
      if (threadIdx.x < 64) {
        uint16_t* wtab = reinterpret_cast<uint16_t*>(s.wbuf);
        with_lds_range<sizeof(*s.Lp) == 2>(A, 0, n, s.wbuf, [&](T* B, int) {
          wave_introselect(B, 0, n, n_points, vsf_sel::lg_(n) * 2, greater, wtab);
          const T ambiguous = B[n_points - 1];
          int total_r = 0, cut = 0;
          wave_hoare_pass(
              B, n_points, n, [&](const T& x) { return !ge(x, ambiguous); },
              [&](const T& x) { return ge(x, ambiguous); }, total_r, cut, wtab);
          if (threadIdx.x == 0) c.st[4] = n_points + total_r;
        });
      }
      __syncthreads();
      const int r = c.st[4];
      __syncthreads();
      return r;
    }
    par_nth_element<NT>(A, n, n_points, greater, s, stage);
    const T ambiguous = A[n_points - 1];
    return par_partition_keep_true<NT>(
        A, n_points, n, [&](const T& x) { return ge(x, ambiguous); }, s);
  }
  return n;
}

// HarrisResponses for one keypoint at integer (x0, y0): the 9x9 patch (7x7 block + Sobel reach) is fetched first as
// 9 x 3 aligned dwords -- 27 independent loads in flight, one memory round trip -- and shifted into place with
// v_alignbyte; the arithmetic then runs from registers.
__device__ __forceinline__ float harris_response(const uint8_t* __restrict__ img, int pitch, int x0, int y0) {
  const int xs = x0 - 4;
  const uint8_t* base = img + (size_t)(y0 - 4) * pitch + (xs & ~3);
  const uint32_t sh = (uint32_t)(xs & 3);
  uint32_t lo[9], mid[9], hi[9];
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (size_t)r * pitch);
    lo[r] = p[0];
    mid[r] = p[1];
    hi[r] = p[2];
  }
  int px[9][9];
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const uint32_t a0 = __builtin_amdgcn_alignbyte(mid[r], lo[r], sh);  // patch columns 0..3
    const uint32_t a1 = __builtin_amdgcn_alignbyte(hi[r], mid[r], sh);  // 4..7
    const uint32_t a2 = __builtin_amdgcn_alignbyte(0u, hi[r], sh);      // 8
#pragma unroll
    for (int c = 0; c < 4; c++) {
      px[r][c] = (int)((a0 >> (8 * c)) & 255u);
      px[r][4 + c] = (int)((a1 >> (8 * c)) & 255u);
    }
    px[r][8] = (int)(a2 & 255u);
  }
  int a = 0, b = 0, c = 0;
#pragma unroll
  for (int i = 1; i <= 7; i++) {
#pragma unroll
    for (int j = 1; j <= 7; j++) {
      const int Ix = (px[i][j + 1] - px[i][j - 1]) * 2 + (px[i - 1][j + 1] - px[i - 1][j - 1]) +
                     (px[i + 1][j + 1] - px[i + 1][j - 1]);
      const int Iy = (px[i + 1][j] - px[i - 1][j]) * 2 + (px[i + 1][j - 1] - px[i - 1][j - 1]) +
                     (px[i + 1][j + 1] - px[i - 1][j + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  }
  const float scale = 1.f / ((1 << 2) * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  return (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale_sq_sq;
}

// ENTRIES: stage-1 candidates kept in LDS; STAGE2: stage-2 pairs kept in LDS; MAXW: mask words of a parallel pass.
// LDS_TABLES = false (round 4, the class of the widest levels): the ARRAY of a level with up to ENTRIES candidates lives in
// LDS from the gather on, and only the rank -> position tables of its passes (16-bit, written once and read once per pass)
// go through the level's HBM scratch -- instead of the array itself, its masks and 32-bit tables living there until the
// selection range has shrunk to the common class's 3 072 entries (two or three passes over 4 ... 8 000 elements each).
template <int NT, int ENTRIES, int STAGE2, int MAXW, bool LDS_TABLES = true, int WGS = (NT == 256 ? 5 : 1)>
__global__ __launch_bounds__(NT, WGS) void orb_select_kernel(SelectArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t sA[ENTRIES];
  constexpr int POS16 = LDS_TABLES ? 2 * ENTRIES : 4096;  // 16-bit words of sPos
  __shared__ uint16_t sPos[POS16];  // rank -> position tables of the LDS-resident passes (and the gather's scratch)
  __shared__ __attribute__((aligned(16))) uint2 sB[STAGE2];
  __shared__ unsigned long long sMask[2 * MAXW];
  __shared__ int sPre[2 * MAXW];
  __shared__ PassCtl ctl;
  __shared__ int lds4[16];
  static_assert(ENTRIES <= 65536 && ENTRIES / 64 <= MAXW && STAGE2 <= ENTRIES && ENTRIES >= 1025, "scratch sizes");
  // the gather borrows the position tables: three quarters for the cell prefix array, one for the row-start tables
  constexpr int kCellCap = (POS16 / 2) * 3 / 4, kRsUnits = (POS16 / 8) * 2 / VSF_FAST_RS_STRIDE;
  static_assert(kCellCap >= VSF_FAST_STRIP_ROWS * 16 + 1 && kRsUnits >= 8, "gather scratch");
  int* cellpre = reinterpret_cast<int*>(sPos);
  uint16_t* rs_lds = reinterpret_cast<uint16_t*>(cellpre + kCellCap);
  const int tid = threadIdx.x;
  // image -> XCD affinity (workgroups go to the 8 XCDs round-robin by linear id): an image's candidates, pixels and
  // scratch stay in one L2 (speed only)
  // Dispatch order = level-major: level 0 of every image first, then level 1, ... -- the heaviest workgroups (the widest
  // levels: most candidates) start first and the light ones fill the tail of the launch (longest job first), instead of
  // the last images' widest levels starting when the chip has already drained.
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int groups = (a.nimages + 7) >> 3;
  const int image = xcd + 8 * (seq - (seq / groups) * groups), level = a.level0 + seq / groups;
  if (image >= a.nimages) return;
  const VsfLevel L = a.levels[level];
  const uint8_t* img;
  int pitch;
  if (level == 0) {
    img = a.img0 + (size_t)image * a.img0_stride;
    pitch = a.img0_pitch;
  } else {
    img = a.pyr + (size_t)image * a.pyr_bytes + L.offset;
    pitch = L.pitch;
  }

  // ---- gather: merge the FAST units' segments into the level's raster order (vsf_gather.h) ----
  const uint16_t* rs_img = a.rowstart + (size_t)image * a.nunits * VSF_FAST_RS_STRIDE;
  uint32_t* gscratch = a.scratch + (size_t)image * 6 * a.cand_entries;
  uint32_t* gA = gscratch + L.cand_offset;
  uint2* gB = reinterpret_cast<uint2*>(gscratch + a.cand_entries) + L.cand_offset;
  // wave scratch: stage 1 borrows the (still unused) stage-2 array, stage 2 the stage-1 array (consumed by then)
  static_assert(sizeof(uint2) * STAGE2 >= kWaveScratch && sizeof(uint32_t) * ENTRIES >= kWaveScratch, "wave scratch");
  uint8_t* wbuf1 = reinterpret_cast<uint8_t*>(sB);
  uint8_t* wbuf2 = reinterpret_cast<uint8_t*>(sA);
  // (LDS_TABLES = false: the two 16-bit tables sit in the level's slices of scratch blocks 3 and 4 -- level_cap 32-bit
  // words each, i.e. room for 2 level_cap positions -- where the HBM-resident passes keep their 32-bit ones)
  uint16_t* tabL = LDS_TABLES ? sPos : reinterpret_cast<uint16_t*>(gscratch + 3 * a.cand_entries + L.cand_offset);
  uint16_t* tabR = LDS_TABLES ? sPos + ENTRIES : reinterpret_cast<uint16_t*>(gscratch + 4 * a.cand_entries + L.cand_offset);
  const PassMem<uint16_t> pm_lds{sMask, sMask + MAXW, sPre, sPre + MAXW, tabL, tabR, ENTRIES / 64, &ctl, wbuf1};
  // HBM-resident passes keep their masks in HBM as well (24 bytes per 64 elements, carved from the level's slice of
  // the sixth scratch block), so a level of any candidate count runs the parallel passes
  const int level_cap = L.seg_cap * L.nbands * L.nstrips, hbm_w = (level_cap + 63) / 64;
  unsigned long long* gmask = reinterpret_cast<unsigned long long*>(
      (reinterpret_cast<uintptr_t>(gscratch + 5 * a.cand_entries + L.cand_offset) + 7) & ~(uintptr_t)7);
  int* gpre = reinterpret_cast<int*>(gmask + 2 * hbm_w);
  const bool hbm_masks = 6 * hbm_w + 2 <= level_cap;  // (always true for real levels; else the LDS masks bound the range)
  const PassMem<uint32_t> pm_hbm{hbm_masks ? gmask : sMask,
                                 hbm_masks ? gmask + hbm_w : sMask + MAXW,
                                 hbm_masks ? gpre : sPre,
                                 hbm_masks ? gpre + hbm_w : sPre + MAXW,
                                 gscratch + 3 * a.cand_entries + L.cand_offset,
                                 gscratch + 4 * a.cand_entries + L.cand_offset,
                                 hbm_masks ? hbm_w : MAXW,
                                 &ctl,
                                 wbuf1};
  PassMem<uint16_t> pm_lds2 = pm_lds;
  PassMem<uint32_t> pm_hbm2 = pm_hbm;
  pm_lds2.wbuf = pm_hbm2.wbuf = wbuf2;
  bool a_in_lds = true;
  const uint32_t* cand_img = a.cand + (size_t)image * a.cand_entries;
  const int n = vsf_gather_level<NT>(
      L, cand_img, rs_img, cellpre, kCellCap, rs_lds, kRsUnits, lds4, [&](int total) { a_in_lds = total <= ENTRIES; },
      [&](int dst, uint32_t cd) {
        if (a_in_lds)  // (workgroup-uniform)
          sA[dst] = cd;
        else
          gA[dst] = cd;
      });
  __syncthreads();

  // ---- K3: retainBest(2 * n_l) on the FAST score ----
  int m1;
  if (a_in_lds)
    m1 = par_retain_best<NT>(sA, n, 2 * L.nfeatures, ScoreGreater(), ScoreGe(), pm_lds);
  else
    m1 = par_retain_best<NT>(gA, n, 2 * L.nfeatures, ScoreGreater(), ScoreGe(), pm_hbm,
                             LdsStage<uint32_t>{sA, ENTRIES, &pm_lds});  // (sA is idle while the array lives in HBM)
  __syncthreads();

  // ---- K4: Harris responses (one lane per keypoint) ----
  const bool b_in_lds = m1 <= STAGE2;
  for (int i = tid; i < m1; i += NT) {
    const uint32_t cd = a_in_lds ? sA[i] : gA[i];
    const float r = harris_response(img, pitch, VSF_CAND_X(cd), VSF_CAND_Y(cd));
    const uint2 e = make_uint2(__float_as_uint(r), cd & 0xFFFFFFu);
    if (b_in_lds)
      sB[i] = e;
    else
      gB[i] = e;
  }
  __syncthreads();

  // ---- K5: retainBest(n_l) on the Harris response ----
  int m2;
  if (b_in_lds)
    m2 = par_retain_best<NT>(sB, m1, L.nfeatures, RespGreater(), RespGe(), pm_lds2);
  else
    m2 = par_retain_best<NT>(gB, m1, L.nfeatures, RespGreater(), RespGe(), pm_hbm2);
  __syncthreads();

  // ---- survivors in retainBest order; K6 (ICAngles) runs in k_describe.hip, one wave per keypoint ----
  VsfLevelKp* out = a.lvlkp + (size_t)image * a.lvlkp_entries + L.kp_offset;
  const int m_out = min(m2, L.kp_cap);
  for (int i = tid; i < m_out; i += NT) {
    const uint2 e = b_in_lds ? sB[i] : gB[i];
    VsfLevelKp kp;
    kp.xy = e.y;
    kp.response = __uint_as_float(e.x);
    kp.angle = -1.f;
    kp.ca = 1.f;
    kp.sb = 0.f;
    out[i] = kp;
  }
  if (tid == 0) {
    a.lvl_count[(size_t)image * a.nlevels + level] = m_out;
    if (m2 > L.kp_cap) atomicOr(a.status + (size_t)image * a.status_stride, 1);
  }
}

// Test hook: retainBest on (float key, id) pairs, one workgroup, arrays in HBM or LDS.
template <int NT, int MAXW>
__global__ __launch_bounds__(NT) void retain_best_test_kernel(uint2* data, uint32_t* tables, int n, int n_points,
                                                              int use_lds, int mode, int* out_n) {
  constexpr int kBuf = 4096;
  __shared__ uint2 buf[kBuf];
  __shared__ uint16_t sPos[2 * kBuf];
  __shared__ unsigned long long sMask[2 * MAXW];
  __shared__ int sPre[2 * MAXW];
  __shared__ PassCtl ctl;
  __shared__ __attribute__((aligned(8))) uint8_t wbuf[kWaveScratch];
  const PassMem<uint16_t> pm_lds{sMask, sMask + MAXW, sPre, sPre + MAXW, sPos, sPos + kBuf, kBuf / 64, &ctl, wbuf};
  const PassMem<uint32_t> pm_hbm{sMask, sMask + MAXW, sPre, sPre + MAXW, tables, tables + n, MAXW, &ctl, wbuf};
  int m;
  if (mode == 0) {  // float keys
    if (use_lds && n <= kBuf) {
      for (int i = threadIdx.x; i < n; i += NT) buf[i] = data[i];
      __syncthreads();
      m = par_retain_best<NT>(buf, n, n_points, RespGreater(), RespGe(), pm_lds);
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += NT) data[i] = buf[i];
    } else {
      m = par_retain_best<NT>(data, n, n_points, RespGreater(), RespGe(), pm_hbm);
    }
  } else {  // packed candidates: compare the top byte of .x only
    struct G {
      __device__ bool operator()(const uint2& a, const uint2& b) const { return (a.x >> 24) > (b.x >> 24); }
    };
    struct GE {
      __device__ bool operator()(const uint2& a, const uint2& b) const { return (a.x >> 24) >= (b.x >> 24); }
    };
    m = par_retain_best<NT>(data, n, n_points, G(), GE(), pm_hbm);
  }
  if (threadIdx.x == 0) *out_n = m;
}

}  // namespace

void vsf_launch_select(const VsfDev& d, const VsfGeom& g, const VsfLevel* h_levels, const VsfImages& im,
                       hipStream_t s) {
  SelectArgs a;
  a.levels = d.levels;
  a.img0 = im.base;
  a.img0_stride = im.image_stride;
  a.img0_pitch = (int)im.row_stride;
  a.pyr = d.pyr;
  a.pyr_bytes = g.pyr_bytes;
  a.cand = d.cand;
  a.cand_entries = g.cand_entries;
  a.rowstart = d.rowstart;
  a.nunits = g.nunits;
  a.scratch = d.scratch;
  a.lvlkp = d.lvlkp;
  a.lvlkp_entries = g.lvlkp_entries;
  a.lvl_count = d.lvl_count;
  a.nlevels = g.nlevels;
  a.status = d.status;
  a.status_stride = d.status_stride;
  // Levels are split by the area in which keypoints may sit; every class falls back to HBM scratch when a level has
  // more candidates than its LDS array holds, so the split only affects speed.
  int nbig = 0;
  while (nbig < g.nlevels) {
    const VsfLevel& L = h_levels[nbig];
    const long area = (long)(L.x_hi - L.x_lo) * (L.y_hi - L.y_lo);
    if (area < 100000) break;
    ++nbig;
  }
  // ... and a "tiny" class (single-wave workgroups: a barrier is one s_barrier of one wave, a dozen workgroups
  // share a CU) for the many small top levels, where the selection is pure latency
  int ntiny0 = nbig;
  while (ntiny0 < g.nlevels) {
    const VsfLevel& L = h_levels[ntiny0];
    const long area = (long)(L.x_hi - L.x_lo) * (L.y_hi - L.y_lo);
    if (area < 20000) break;
    ++ntiny0;
  }
  // Large and mid levels share one launch of 256-thread workgroups with a 3072-entry LDS array (30 KB, and 96 VGPRs by
  // launch bounds: FIVE workgroups per CU -- the kernel is a chain of short dependent phases, so resident workgroups
  // are throughput).  Levels with more candidates start on HBM-resident arrays and move to the LDS array once the
  // selection range fits (par_introselect).  Per 128-frame step: one 1024-thread / 134 KB workgroup per CU 1.13 ms;
  // 5120 entries, three per CU 0.58 ms; 4096, four per CU 0.53 ms; 3072, five per CU 0.50 ms; 2560, six per CU 0.50 ms.
  a.nimages = im.n;
  const int n8 = (im.n + 7) / 8 * 8;
  // A frame or two leaves the chip nearly empty and the stage lasts as long as its slowest workgroup -- level 0 of a
  // 640x480 image: 8 500 candidates, 92 us of selection passes with 256 threads on an HBM-resident array, 39 us with 1 024
  // threads on an LDS array that holds the whole level (105 KB, one workgroup per CU: at most 256 of them).
  const bool wide_off = d.tune && !d.tune->select_wide;
  if (ntiny0 > 0 && !wide_off && (long)ntiny0 * im.n <= 256) {
    if ((long)g.nlevels * im.n <= 256) ntiny0 = g.nlevels;  // ... and the small top levels ride along (one launch)
    a.level0 = 0;
    a.nlv = ntiny0;
    hipLaunchKernelGGL((orb_select_kernel<1024, 12288, 512, 192>), dim3(a.nlv * n8), dim3(1024), 0, s, a);
  } else if (ntiny0 > 0) {
    // Levels whose area holds more than 3 072 but (on a busy scene: one candidate per ~30 pixels) no more than 9 216
    // candidates take the 9 216-entry class, three per CU: their array lives in LDS from the gather on.  Larger levels
    // (1920x1080's first twenty-odd: up to 70 000 candidates) would fall back to HBM-resident passes in ANY class and keep
    // the common one, whose five workgroups per CU hide those passes' latency better (1080p, 96 frames per step: 2.03 ms
    // against 2.40 ms with every wide level in the 9 216-entry class).
    int nhuge = 0;
    while (nhuge < ntiny0) {
      const VsfLevel& L = h_levels[nhuge];
      if ((long)(L.x_hi - L.x_lo) * (L.y_hi - L.y_lo) <= 300000) break;
      ++nhuge;
    }
    // ... and only when there are enough of them to fill its three slots per CU a few times over (64 images at 1080p are
    // 900 such workgroups for 768 slots: one round and a long tail -- 5 450 -> 5 090 frames/s)
    int nwide_end = (d.tune && !d.tune->select_big_class) ? nhuge : std::max(nhuge, std::min(nbig, ntiny0));
    if ((long)(nwide_end - nhuge) * im.n < 1536) nwide_end = nhuge;
    auto common = [&](int l0, int l1) {
      if (l1 <= l0) return;
      a.level0 = l0;
      a.nlv = l1 - l0;
      hipLaunchKernelGGL((orb_select_kernel<256, 3072, 512, 64>), dim3(a.nlv * n8), dim3(256), 0, s, a);
    };
    if (nwide_end == nhuge) nhuge = nwide_end = 0;  // (no wide class this time: ONE launch of the common class, as before)
    common(0, nhuge);
    if (nwide_end > nhuge) {
      a.level0 = nhuge;
      a.nlv = nwide_end - nhuge;
      hipLaunchKernelGGL((orb_select_kernel<256, 9216, 512, 144, false, 3>), dim3(a.nlv * n8), dim3(256), 0, s, a);
    }
    // (one after the other on the stream: beside each other on two streams the pair took 0.4 ms longer per step)
    common(nwide_end, ntiny0);
  }
  if (ntiny0 < g.nlevels) {
    a.level0 = ntiny0;
    a.nlv = g.nlevels - ntiny0;
    hipLaunchKernelGGL((orb_select_kernel<64, 1536, 512, 64>), dim3(a.nlv * n8), dim3(64), 0, s, a);
  }
}

void vsf_launch_retain_best_test(uint2* d_data, uint32_t* d_tables, int n, int n_points, int use_lds, int mode,
                                 int* d_out_n, hipStream_t s) {
  // d_tables: 2 * n uint32 of scratch (rank -> position tables of the HBM-resident passes)
  hipLaunchKernelGGL((retain_best_test_kernel<256, 1024>), dim3(1), dim3(256), 0, s, d_data, d_tables, n, n_points,
                     use_lds, mode, d_out_n);
}
