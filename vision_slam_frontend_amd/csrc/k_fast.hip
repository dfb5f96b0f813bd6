// k_fast.hip -- K2 (+ the border half of K3): FAST-9/16 segment test, cornerScore, 3x3 non-max suppression and
// raster-order compaction, for every level of every image in ONE launch.
//
// Restates cv::FAST_t<16> and cornerScore<16> (features2d/fast.cpp, fast_score.cpp) as called by
// ORB's computeKeyPoints (threshold 20; slam_frontend.cc:274, parameters :205-213) and by
// FastFeatureDetector::detect (threshold 10, slam_frontend.cc:191,271), plus
// KeyPointsFilter::runByImageBorder(31) which is folded into the evaluated rectangle.
//
// Work decomposition: one workgroup owns a full-width strip of `strip_rows` rows of one level of one image.
//   phase 0  coalesced 16-byte loads of the strip (+4 halo rows) into an LDS image tile
//   phase A  every pixel: high-speed reject on the 2 vertical circle pixels, then the 16-pixel segment
//            test on bit masks; corners are appended to an LDS list (dense work for phase B)
//   phase B  one lane per listed corner: score = max(t, max_arc min(v-p), max_arc min(p-v)) - 1 -> LDS score tile
//   phase C  strict 8-neighbour NMS on the score tile and an order-preserving compaction (each lane owns a
//            contiguous raster run; block-wide exclusive scan) into the strip's candidate segment in HBM.
// Strips of a level are ordered by y, so concatenating the segments gives OpenCV's raster order.
#include "vsf_internal.h"

namespace {

struct FastArgs {
  const VsfLevel* levels;
  const uint32_t* strips;
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;
  uint32_t pyr_bytes;
  uint32_t* cand;
  uint32_t cand_entries;
  int32_t* strip_count;
  int nstrips;
  int threshold;
  int nms;
  int strip_rows;
  int tile_pitch_max;
  int score_pitch_max;
};

__device__ __forceinline__ bool has9(uint32_t m) {
  // m: 16 circle flags; true iff 9 circularly contiguous bits are set.
  m |= m << 16;
  uint32_t a = m & (m >> 1);
  a &= a >> 2;
  a &= a >> 4;
  a &= m >> 8;
  return (a & 0xFFFFu) != 0;
}

// Block-wide exclusive scan over 256 threads (4 waves of 64).
__device__ __forceinline__ int block_excl_scan_256(int v, int* lds4, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds4[wid] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wid; w++) base += lds4[w];
  *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return base + inc - v;
}

#define VSF_CIRCLE16(F)                                                                                        \
  F(0, 0, 3) F(1, 1, 3) F(2, 2, 2) F(3, 3, 1) F(4, 3, 0) F(5, 3, -1) F(6, 2, -2) F(7, 1, -3) F(8, 0, -3)        \
      F(9, -1, -3) F(10, -2, -2) F(11, -3, -1) F(12, -3, 0) F(13, -3, 1) F(14, -2, 2) F(15, -1, 3)

__global__ __launch_bounds__(VSF_FAST_THREADS) void fast_strip_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const uint32_t sdesc = a.strips[blockIdx.x];
  const int level = (int)(sdesc >> 16), ys = (int)(sdesc & 0xFFFFu);
  const VsfLevel L = a.levels[level];
  const int image = blockIdx.y;
  const int SR = a.strip_rows;
  const int ye = min(ys + SR, L.y_hi);
  const uint8_t* src;
  int pitch;
  if (level == 0) {
    src = a.img0 + (size_t)image * a.img0_stride;
    pitch = a.img0_pitch;
  } else {
    src = a.pyr + (size_t)image * a.pyr_bytes + L.offset;
    pitch = L.pitch;
  }
  uint8_t* tile = smem;
  uint8_t* score = tile + (SR + 8) * a.tile_pitch_max;
  uint16_t* list = reinterpret_cast<uint16_t*>(score + (SR + 2) * a.score_pitch_max);
  int* misc = reinterpret_cast<int*>(list + (SR + 2) * a.score_pitch_max);

  const int cx0 = max(L.x_lo - 4, 0) & ~15;
  const int cx1 = min((L.x_hi + 4 + 15) & ~15, pitch);
  const int tp = cx1 - cx0;
  const int ry0 = max(ys - 4, 0), ry1 = min(ye + 4, L.h);
  const int nrows = ry1 - ry0;
  const int sp = (L.ncols + 15) & ~15;
  const int nsr = ye - ys + 2;

  // phase 0: image tile + zeroed score tile
  {
    const int cpr = tp >> 4;
    const int nchunks = nrows * cpr;
    for (int i = tid; i < nchunks; i += VSF_FAST_THREADS) {
      const int r = i / cpr, c = i - r * cpr;
      const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(ry0 + r) * pitch + cx0 + c * 16);
      *reinterpret_cast<uint4*>(tile + r * tp + c * 16) = v;
    }
    const int nz = (nsr * sp) >> 4;
    for (int i = tid; i < nz; i += VSF_FAST_THREADS) reinterpret_cast<uint4*>(score)[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) misc[4] = 0;
  }
  __syncthreads();

  // phase A: segment test
  const int ncols = L.ncols;
  const int total = nsr * ncols;
  const int t = a.threshold;
  for (int idx = tid; idx < total; idx += VSF_FAST_THREADS) {
    const int r = (int)__umulhi((uint32_t)idx, L.ncols_magic);
    const int c = idx - r * ncols;
    const int y = ys - 1 + r, x = L.x_lo - 1 + c;
    bool corner = false;
    if (y >= 3 && y < L.h - 3 && x >= 3 && x < L.w - 3) {
      const uint8_t* p = tile + (y - ry0) * tp + (x - cx0);
      const int v = p[0];
      const int lo = v - t, hi = v + t;
      const int p0 = p[3 * tp], p8 = p[-3 * tp];
      // A 9-arc of the 16-circle always contains pixel 0 or pixel 8.
      if ((p0 < lo) | (p8 < lo) | (p0 > hi) | (p8 > hi)) {
        uint32_t dark = 0, bright = 0;
#define VSF_F(k, dx, dy)                     \
  {                                          \
    const int pk = p[(dy)*tp + (dx)];        \
    dark |= (uint32_t)(pk < lo) << (k);      \
    bright |= (uint32_t)(pk > hi) << (k);    \
  }
        VSF_CIRCLE16(VSF_F)
#undef VSF_F
        corner = has9(dark) || has9(bright);
      }
    }
    if (corner) {
      const int pos = atomicAdd(&misc[4], 1);
      list[pos] = (uint16_t)idx;
    }
  }
  __syncthreads();

  // phase B: corner score
  const int nlist = misc[4];
  for (int i = tid; i < nlist; i += VSF_FAST_THREADS) {
    const int idx = list[i];
    const int r = (int)__umulhi((uint32_t)idx, L.ncols_magic);
    const int c = idx - r * ncols;
    const int y = ys - 1 + r, x = L.x_lo - 1 + c;
    int s = 1;  // marker when NMS is off (OpenCV then leaves the response at 0)
    if (a.nms) {
      const uint8_t* p = tile + (y - ry0) * tp + (x - cx0);
      const int v = p[0];
      int d[16];
#define VSF_F(k, dx, dy) d[k] = v - (int)p[(dy)*tp + (dx)];
      VSF_CIRCLE16(VSF_F)
#undef VSF_F
      int mn[16], mx[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        mn[k] = min(d[k], d[(k + 1) & 15]);
        mx[k] = max(d[k], d[(k + 1) & 15]);
      }
      int mn2[16], mx2[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        mn2[k] = min(mn[k], mn[(k + 2) & 15]);
        mx2[k] = max(mx[k], mx[(k + 2) & 15]);
      }
      int a0 = t, b0 = -t;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int m9 = min(min(mn2[k], mn2[(k + 4) & 15]), d[(k + 8) & 15]);
        const int x9 = max(max(mx2[k], mx2[(k + 4) & 15]), d[(k + 8) & 15]);
        a0 = max(a0, m9);
        b0 = min(b0, x9);
      }
      // cornerScore<16>: a0 = max(t, max_arc min d); b0 = min(-a0, min_arc max d); result -b0 - 1.
      b0 = min(b0, -a0);
      s = -b0 - 1;
    }
    score[r * sp + c] = (uint8_t)s;
  }
  __syncthreads();

  // phase C: NMS + raster-order compaction over rows [ys, ye) x cols [x_lo, x_hi)
  const int vw = L.x_hi - L.x_lo;
  const int npx = (ye - ys) * vw;
  const int chunk = (npx + VSF_FAST_THREADS - 1) / VSF_FAST_THREADS;
  const int beg = min(tid * chunk, npx), end = min(beg + chunk, npx);
  const int r_beg = beg / vw, c_beg = beg - r_beg * vw;
  auto is_kp = [&](int r, int c) -> int {
    const uint8_t* q = score + (r + 1) * sp + (c + 1);
    const int s = q[0];
    if (s == 0) return -1;
    if (!a.nms) return 0;
    const bool keep = s > q[-1] && s > q[1] && s > q[-sp - 1] && s > q[-sp] && s > q[-sp + 1] && s > q[sp - 1] &&
                      s > q[sp] && s > q[sp + 1];
    return keep ? s : -1;
  };
  int count = 0;
  {
    int r = r_beg, c = c_beg;
    for (int i = beg; i < end; i++) {
      count += is_kp(r, c) >= 0;
      if (++c == vw) c = 0, ++r;
    }
  }
  int total_kp;
  int pos = block_excl_scan_256(count, misc, &total_kp);
  const int strip_local = (int)blockIdx.x - L.strip0;
  uint32_t* seg = a.cand + (size_t)image * a.cand_entries + L.cand_offset + (size_t)strip_local * L.seg_cap;
  if (count) {
    int r = r_beg, c = c_beg;
    for (int i = beg; i < end; i++) {
      const int s = is_kp(r, c);
      if (s >= 0) {
        if (pos < L.seg_cap) seg[pos] = VSF_CAND_PACK(L.x_lo + c, ys + r, s);
        ++pos;
      }
      if (++c == vw) c = 0, ++r;
    }
  }
  if (tid == 0) a.strip_count[(size_t)image * a.nstrips + blockIdx.x] = min(total_kp, L.seg_cap);
}

// Standalone FAST detect: candidate segments -> contiguous cv::KeyPoint list (raster order).
__global__ __launch_bounds__(256) void fast_emit_kernel(const VsfLevel* __restrict__ levels,
                                                        const uint32_t* __restrict__ cand, uint32_t cand_entries,
                                                        const int32_t* __restrict__ strip_count, int nstrips,
                                                        int max_keypoints, vsf_keypoint* __restrict__ out,
                                                        int32_t* __restrict__ counts, int32_t* __restrict__ status) {
  __shared__ int lds4[8];
  __shared__ int s_base;
  const int image = blockIdx.x;
  const VsfLevel L = levels[0];
  const int32_t* sc = strip_count + (size_t)image * nstrips;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int s0 = 0; s0 < L.nstrips; s0 += 256) {
    const int s = s0 + threadIdx.x;
    const int n = s < L.nstrips ? sc[L.strip0 + s] : 0;
    int tot;
    const int off = block_excl_scan_256(n, lds4, &tot) + s_base;
    if (s < L.nstrips) {
      const uint32_t* seg = cand + (size_t)image * cand_entries + L.cand_offset + (size_t)s * L.seg_cap;
      for (int i = 0; i < n; i++) {
        const int o = off + i;
        if (o < max_keypoints) {
          const uint32_t cd = seg[i];
          vsf_keypoint kp;
          kp.x = (float)VSF_CAND_X(cd);
          kp.y = (float)VSF_CAND_Y(cd);
          kp.size = 7.f;
          kp.angle = -1.f;
          kp.response = (float)VSF_CAND_SCORE(cd);
          kp.octave = 0;
          kp.class_id = -1;
          out[(size_t)image * max_keypoints + o] = kp;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts[image] = s_base;  // true count; the caller clamps to its capacity
    if (s_base > max_keypoints) atomicOr(status, 1);
  }
}

}  // namespace

void vsf_launch_fast(const VsfDev& d, const VsfGeom& g, const VsfImages& im, int threshold, hipStream_t s) {
  FastArgs a;
  a.levels = d.levels;
  a.strips = d.strips;
  a.img0 = im.base;
  a.img0_stride = im.image_stride;
  a.img0_pitch = (int)im.row_stride;
  a.pyr = d.pyr;
  a.pyr_bytes = g.pyr_bytes;
  a.cand = d.cand;
  a.cand_entries = g.cand_entries;
  a.strip_count = d.strip_count;
  a.nstrips = g.nstrips;
  a.threshold = threshold & 0xFFFF;
  a.nms = (threshold >> 16) ? 0 : 1;  // bit 16 of `threshold` disables NMS (standalone FAST only)
  a.strip_rows = g.strip_rows;
  a.tile_pitch_max = g.max_tile_pitch;
  a.score_pitch_max = g.max_score_pitch;
  const size_t lds = (size_t)(g.strip_rows + 8) * g.max_tile_pitch +
                     (size_t)(g.strip_rows + 2) * g.max_score_pitch * 3 + 64;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fast_strip_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(g.nstrips, im.n, 1);
  hipLaunchKernelGGL(fast_strip_kernel, grid, dim3(VSF_FAST_THREADS), lds, s, a);
}

void vsf_launch_fast_emit(const VsfDev& d, const VsfGeom& g, int n_images, int max_keypoints, vsf_keypoint* d_kp,
                          int32_t* d_counts, hipStream_t s) {
  hipLaunchKernelGGL(fast_emit_kernel, dim3(n_images), dim3(256), 0, s, d.levels, d.cand, g.cand_entries,
                     d.strip_count, g.nstrips, max_keypoints, d_kp, d_counts, d.status);
}
