// vsf_hoare.h -- data-parallel Hoare partition passes (device code): the building block of the order-exact parallel
// restatements of libstdc++'s introselect (k_select.hip: cv::KeyPointsFilter::retainBest) and introsort (k_frontend.hip:
// the std::sort of Frontend::GetFeatureMatches, slam_frontend.cc:289).  A pass reproduces std::__unguarded_partition's
// swaps exactly: the k-th left-stopper (from the left) is exchanged with the k-th right-stopper (from the right) while they
// have not crossed, and the returned cut is where the sequential scan would have stopped.
#ifndef VSF_HOARE_H_
#define VSF_HOARE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vsf_par {

constexpr int kWaveCutoff = 256;  // ranges up to this size are finished by ONE wave without workgroup barriers

// Scratch of the parallel passes.  Masks / prefix counts cover maxw * 64 elements; Lp / Rp are the rank -> position
// tables of the left- and right-stoppers (P = uint16_t in LDS, uint32_t in HBM scratch).
struct PassCtl {
  unsigned long long wsum[16];
  int st[8];  // 0 first, 1 last, 2 depth, 3 K, 4 cut, 5 totalL, 6 totalR
};
template <class P>
struct PassMem {
  unsigned long long* maskL;
  unsigned long long* maskR;
  int* preL;
  int* preR;
  P* Lp;
  P* Rp;
  int maxw;
  PassCtl* c;
  uint8_t* wbuf;  // LDS scratch of the single-wave passes: kWaveScratch bytes (rank table + staging of an HBM range)
};
constexpr int kWaveTable = 2 * kWaveCutoff;               // u16 rank -> position table
constexpr int kWaveScratch = kWaveTable + 8 * kWaveCutoff;  // + up to 256 staged 8-byte elements

__device__ __forceinline__ int select64(unsigned long long x, int r) {
  // position of the r-th (0-based) set bit of x
  int pos = 0;
  uint32_t v = (uint32_t)x;
  int c = __popc(v);
  if (r >= c) {
    r -= c;
    pos = 32;
    v = (uint32_t)(x >> 32);
  }
  c = __popc(v & 0xFFFFu);
  if (r >= c) {
    r -= c;
    pos += 16;
    v >>= 16;
  }
  c = __popc(v & 0xFFu);
  if (r >= c) {
    r -= c;
    pos += 8;
    v >>= 8;
  }
  c = __popc(v & 0xFu);
  if (r >= c) {
    r -= c;
    pos += 4;
    v >>= 4;
  }
  c = __popc(v & 0x3u);
  if (r >= c) {
    r -= c;
    pos += 2;
    v >>= 2;
  }
  if (r >= (int)(v & 1u)) pos += 1;
  return pos;
}

// One Hoare pass over A[lo, hi): left-stoppers are elements with FL(x), right-stoppers those with FR(x).
// Swaps the k-th left-stopper (from the left) with the k-th right-stopper (from the right) for every k with
// left position < right position.  Steps (a workgroup barrier between them):
//   1  flags -> ballot masks (loads batched four deep);  2  block scan of the mask popcounts;
//   3  every stopper writes its position at its rank: Lp[rank from the left], Rp[rank from the left];
//   4  thread k swaps A[Lp[k]] <-> A[Rp[totalR - 1 - k]] while Lp[k] < Rp[...] (no searching);
//   5  K = number of swaps, cut = where a sequential scan would have stopped.
// On return (all threads, after a barrier) c.st[3] = K, c.st[5] / c.st[6] = stopper totals, c.st[4] = cut
// (__unguarded_partition's return value when at least one stopper of each kind exists).
template <int NT, class T, class FL, class FR, class PM>
__device__ __forceinline__ void hoare_pass(T* A, int lo, int hi, FL fl, FR fr, const PM& s) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = hi - lo;
  const int nw = (m + 63) >> 6;
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
      bool l = false, r = false;
      if (i < m) {
        l = fl(x[u]);
        r = fr(x[u]);
      }
      const unsigned long long bl = __ballot(l), br = __ballot(r);
      const int w = ((base + u * NT) >> 6) + wave;
      if (lane == 0 && w < nw) {
        s.maskL[w] = bl;
        s.maskR[w] = br;
      }
    }
  }
  __syncthreads();
  // exclusive prefix popcounts, words [t*wpt, (t+1)*wpt) per thread
  const int wpt = (nw + NT - 1) / NT;
  const int w0 = tid * wpt;
  unsigned long long mine = 0;  // low 32: left count, high 32: right count
  for (int j = 0; j < wpt; j++) {
    const int w = w0 + j;
    if (w < nw) mine += (unsigned long long)__popcll(s.maskL[w]) | ((unsigned long long)__popcll(s.maskR[w]) << 32);
  }
  unsigned long long inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s.c->wsum[wave] = inc;
  __syncthreads();
  unsigned long long base_sum = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    if (w < wave) base_sum += s.c->wsum[w];
    total += s.c->wsum[w];
  }
  unsigned long long run = base_sum + inc - mine;
  for (int j = 0; j < wpt; j++) {
    const int w = w0 + j;
    if (w < nw) {
      s.preL[w] = (int)(uint32_t)run;
      s.preR[w] = (int)(uint32_t)(run >> 32);
      run += (unsigned long long)__popcll(s.maskL[w]) | ((unsigned long long)__popcll(s.maskR[w]) << 32);
    }
  }
  const int totalL = (int)(uint32_t)total, totalR = (int)(uint32_t)(total >> 32);
  __syncthreads();
  // rank -> position tables
  typedef decltype(s.Lp[0] + 0) PosInt;  // (promoted) element type of the tables
  for (int base = 0; base < m; base += NT) {
    const int i = base + tid;
    if (i < m) {
      const int w = i >> 6, b = i & 63;
      const unsigned long long ml = s.maskL[w], mr = s.maskR[w], below = (1ull << b) - 1ull;
      if ((ml >> b) & 1ull) s.Lp[s.preL[w] + __popcll(ml & below)] = (PosInt)i;
      if ((mr >> b) & 1ull) s.Rp[s.preR[w] + __popcll(mr & below)] = (PosInt)i;
    }
  }
  __syncthreads();
  // swaps: the k-th left-stopper with the k-th right-stopper from the right, while they have not crossed
  const int kmax = min(totalL, totalR);
  int nswap = 0;
  for (int base = 0; base < kmax; base += NT * U) {
    int pi[U], pj[U];
    T xi[U], xj[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = base + u * NT + tid;
      pi[u] = 0;
      pj[u] = 0;
      if (k < kmax) {
        pi[u] = (int)s.Lp[k];
        pj[u] = (int)s.Rp[totalR - 1 - k];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
      if (pi[u] < pj[u]) {
        xi[u] = A[lo + pi[u]];
        xj[u] = A[lo + pj[u]];
      }
#pragma unroll
    for (int u = 0; u < U; u++)
      if (pi[u] < pj[u]) {
        A[lo + pi[u]] = xj[u];
        A[lo + pj[u]] = xi[u];
        ++nswap;
      }
  }
  // K = total swaps
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nswap += __shfl_xor(nswap, o, 64);
  if (lane == 0) s.c->wsum[wave] = (unsigned long long)nswap;
  __syncthreads();
  if (tid == 0) {
    int K = 0;
    for (int w = 0; w < NT / 64; w++) K += (int)s.c->wsum[w];
    s.c->st[3] = K;
    s.c->st[5] = totalL;
    s.c->st[6] = totalR;
    // where the left scan stops after K swaps: the (K+1)-th left-stopper if it lies before the K-th right-stopper
    // from the right (or no swap happened), else that right-stopper's position (now holding a left-stopper).
    const int aK1 = K < totalL ? (int)s.Lp[K] : 0x7FFFFFFF;
    const int bK = K > 0 ? (int)s.Rp[totalR - K] : 0x7FFFFFFF;
    s.c->st[4] = (K < totalL && (K == 0 || aK1 < bK)) ? lo + aK1 : lo + bK;
  }
  __syncthreads();
}

// ---- single-wave versions for ranges of <= kWaveCutoff elements (4 per lane) ----
// Same Hoare pass as above, but the stopper masks are four 64-bit ballots held in scalar registers, ranks are
// v_mbcnt prefix counts and nothing needs a workgroup barrier.  All 64 lanes of ONE wave call these.
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

template <int E>
__device__ __forceinline__ int select_rank(const unsigned long long (&m)[E], const int (&pre)[E + 1], int t) {
  int w = 0;
#pragma unroll
  for (int e = 1; e < E; e++) w += (t >= pre[e]);
  unsigned long long mk = m[0];
  int p = pre[0];
#pragma unroll
  for (int e = 1; e < E; e++)
    if (w == e) mk = m[e], p = pre[e];
  return w * 64 + select64(mk, t - p);
}

// One pass over A[lo, hi), hi - lo <= 64 E, A in LDS.  Returns (uniform) the right-stopper total and `cut`.
// The right-stoppers publish their positions in a rank -> position table (rank counted from the right); a left-stopper
// of rank k that takes part reads its partner's position there (two LDS round trips instead of a 64-bit rank-select per
// element), then all partner elements are read before any is written (a position is touched by at most one swap).
// E = elements per lane: a single wave works through these passes alone, so their cost is instruction latency, and
// most passes of a selection run on the short ranges at its end.
template <int E, class T, class FL, class FR>
__device__ __forceinline__ void wave_hoare_pass_e(T* A, int lo, int hi, FL fl, FR fr, int& totalR, int& cut,
                                                  uint16_t* wtab) {
  const int lane = threadIdx.x & 63;
  const int m = hi - lo;
  const unsigned long long lt = (1ull << lane) - 1ull, le = (2ull << lane) - 1ull;
  unsigned long long mL[E], mR[E];
  T x[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = e * 64 + lane;
    if (i < m) x[e] = A[lo + i];
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = e * 64 + lane;
    bool l = false, r = false;
    if (i < m) {
      l = fl(x[e]);
      r = fr(x[e]);
    }
    mL[e] = __ballot(l);
    mR[e] = __ballot(r);
  }
  int preL[E + 1], preR[E + 1];
  preL[0] = preR[0] = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    preL[e + 1] = preL[e] + __popcll(mL[e]);
    preR[e + 1] = preR[e] + __popcll(mR[e]);
  }
  const int totalL = preL[E];
  totalR = preR[E];
  bool part[E];
  int kk[E], rr[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const bool is_l = (mL[e] >> lane) & 1ull, is_r = (mR[e] >> lane) & 1ull;
    const int k = preL[e] + __popcll(mL[e] & lt) + 1;    // 1-based rank from the left
    const int r_le = preR[e] + __popcll(mR[e] & le);     // right-stoppers at <= i
    if (is_r) wtab[totalR - r_le] = (uint16_t)(e * 64 + lane);  // 0-based rank from the right
    part[e] = is_l && (totalR - r_le >= k);
    kk[e] = k;
    rr[e] = totalR - r_le + 1;                           // 1-based rank from the right (when a right-stopper)
  }
  wave_fence();
  int jj[E];
#pragma unroll
  for (int e = 0; e < E; e++) jj[e] = part[e] ? (int)wtab[kk[e] - 1] : 0;
  T xj[E];
#pragma unroll
  for (int e = 0; e < E; e++)
    if (part[e]) xj[e] = A[lo + jj[e]];
  int K = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    if (part[e]) {
      A[lo + e * 64 + lane] = xj[e];
      A[lo + jj[e]] = x[e];
    }
    K += __popcll(__ballot(part[e]));
  }
  // the (K+1)-th left-stopper and the K-th right-stopper from the right: the lane that holds it raises its hand
  int aK1 = 0x7FFFFFFF, bK = 0x7FFFFFFF;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const unsigned long long ha = __ballot(((mL[e] >> lane) & 1ull) && kk[e] == K + 1);
    const unsigned long long hb = __ballot(((mR[e] >> lane) & 1ull) && rr[e] == K);
    if (ha) aK1 = e * 64 + (int)__builtin_ctzll(ha);
    if (hb && K > 0) bK = e * 64 + (int)__builtin_ctzll(hb);
  }
  cut = (K < totalL && (K == 0 || aK1 < bK)) ? lo + aK1 : lo + bK;
  wave_fence();
}

template <class T, class FL, class FR>
__device__ __forceinline__ void wave_hoare_pass(T* A, int lo, int hi, FL fl, FR fr, int& totalR, int& cut,
                                                uint16_t* wtab) {
  if (hi - lo <= 64)  // wave-uniform
    wave_hoare_pass_e<1>(A, lo, hi, fl, fr, totalR, cut, wtab);
  else
    wave_hoare_pass_e<4>(A, lo, hi, fl, fr, totalR, cut, wtab);
}

}  // namespace vsf_par
#endif  // VSF_HOARE_H_
