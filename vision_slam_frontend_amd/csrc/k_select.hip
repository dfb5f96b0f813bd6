// k_select.hip -- K3 + K4 + K5 + K6 of ORB's computeKeyPoints (features2d/orb.cpp, reached from
// slam_frontend.cc:274), one workgroup per (image, level):
//
//   gather   the level's FAST candidates (raster order) from the strip segments
//   K3       KeyPointsFilter::retainBest(2 * n_l) on the FAST score           (order-exact, vsf_select.h)
//   K4       HarrisResponses(blockSize 7, k 0.04f) for the survivors           (int32 sums, 6 float ops, no FMA)
//   K5       KeyPointsFilter::retainBest(n_l) on the Harris response          (order-exact)
//   K6       ICAngles (intensity-centroid moments over the radius-15 disc, one wave per keypoint) + fastAtan2
//
// The arrays being permuted live in LDS when they fit (16 K candidates / 2 K survivors) and in an HBM scratch
// area otherwise.  The permutation must equal libstdc++'s, so the selection itself is the
// sequential restatement in vsf_select.h executed by one lane; everything around it is data-parallel.
#include "vsf_internal.h"
#include "vsf_select.h"

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
  const int32_t* strip_count;
  int nstrips;
  uint32_t* scratch;  // [image][3 * cand_entries]
  VsfLevelKp* lvlkp;
  int lvlkp_entries;
  int32_t* lvl_count;
  int nlevels;
  int32_t* status;
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

// cv::fastAtan2 (core/mathfuncs.cpp), degrees.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// HarrisResponses for one keypoint at integer (x0, y0).
__device__ __forceinline__ float harris_response(const uint8_t* __restrict__ img, int pitch, int x0, int y0) {
  const uint8_t* base = img + (size_t)(y0 - 3) * pitch + (x0 - 3);
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < 7; i++) {
    const uint8_t* pm = base + (i - 1) * pitch;
    const uint8_t* p0 = base + i * pitch;
    const uint8_t* pp = base + (i + 1) * pitch;
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const int Ix = ((int)p0[j + 1] - (int)p0[j - 1]) * 2 + ((int)pm[j + 1] - (int)pm[j - 1]) +
                     ((int)pp[j + 1] - (int)pp[j - 1]);
      const int Iy = ((int)pp[j] - (int)pm[j]) * 2 + ((int)pp[j - 1] - (int)pm[j - 1]) +
                     ((int)pp[j + 1] - (int)pm[j + 1]);
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

__device__ __forceinline__ int umax31(int v) {
  // ORB's umax table for patchSize 31 (rows of the radius-15 disc): checked against the formula on the host.
  const uint64_t lo = 0xDDEEEFFFFull;  // v = 0..8 : 15,15,15,15,14,14,14,13,13 (4 bits each)
  const uint64_t hi = 0x3689ABCull;    // v = 9..15: 12,11,10,9,8,6,3
  return v < 9 ? (int)((lo >> (4 * v)) & 15) : (int)((hi >> (4 * (v - 9))) & 15);
}

__global__ __launch_bounds__(VSF_SELECT_THREADS) void orb_select_kernel(SelectArgs a) {
  __shared__ uint32_t sA[VSF_SELECT_LDS_ENTRIES];
  __shared__ uint2 sB[VSF_SELECT_LDS_STAGE2];
  __shared__ int soff[520];
  __shared__ int misc[8];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int level = blockIdx.x, image = blockIdx.y;
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

  // ---- gather: exclusive scan of the strip counts, then a coalesced copy of every segment ----
  const int32_t* sc = a.strip_count + (size_t)image * a.nstrips + L.strip0;
  if (tid == 0) {
    int acc = 0;
    for (int s = 0; s < L.nstrips; s++) {
      soff[s] = acc;
      acc += sc[s];
    }
    soff[L.nstrips] = acc;
  }
  __syncthreads();
  const int n = soff[L.nstrips];
  uint32_t* gscratch = a.scratch + (size_t)image * 3 * a.cand_entries;
  const bool a_in_lds = n <= VSF_SELECT_LDS_ENTRIES;
  uint32_t* A = a_in_lds ? sA : gscratch + L.cand_offset;
  const uint32_t* segs = a.cand + (size_t)image * a.cand_entries + L.cand_offset;
  for (int s = 0; s < L.nstrips; s++) {
    const int cnt = soff[s + 1] - soff[s];
    const uint32_t* seg = segs + (size_t)s * L.seg_cap;
    for (int i = tid; i < cnt; i += VSF_SELECT_THREADS) A[soff[s] + i] = seg[i];
  }
  __threadfence_block();
  __syncthreads();

  // ---- K3: retainBest(2 * n_l) on the FAST score ----
  if (tid == 0) {
    int m1;
    if (a_in_lds)
      m1 = vsf_sel::retain_best_(sA, n, 2 * L.nfeatures, ScoreGreater(), ScoreGe());
    else
      m1 = vsf_sel::retain_best_(gscratch + L.cand_offset, n, 2 * L.nfeatures, ScoreGreater(), ScoreGe());
    misc[0] = m1;
  }
  __threadfence_block();
  __syncthreads();
  const int m1 = misc[0];

  // ---- K4: Harris responses (one lane per keypoint) ----
  const bool b_in_lds = m1 <= VSF_SELECT_LDS_STAGE2;
  uint2* B = b_in_lds ? sB : reinterpret_cast<uint2*>(gscratch + a.cand_entries) + L.cand_offset;
  for (int i = tid; i < m1; i += VSF_SELECT_THREADS) {
    const uint32_t cd = A[i];
    const float r = harris_response(img, pitch, VSF_CAND_X(cd), VSF_CAND_Y(cd));
    B[i] = make_uint2(__float_as_uint(r), cd & 0xFFFFFFu);
  }
  __threadfence_block();
  __syncthreads();

  // ---- K5: retainBest(n_l) on the Harris response ----
  if (tid == 0) {
    int m2;
    if (b_in_lds)
      m2 = vsf_sel::retain_best_(sB, m1, L.nfeatures, RespGreater(), RespGe());
    else
      m2 = vsf_sel::retain_best_(reinterpret_cast<uint2*>(gscratch + a.cand_entries) + L.cand_offset, m1,
                                 L.nfeatures, RespGreater(), RespGe());
    misc[1] = m2;
  }
  __threadfence_block();
  __syncthreads();
  const int m2 = misc[1];

  // ---- K6: IC angle, one wave per keypoint, two disc rows per step (32 lanes each) ----
  VsfLevelKp* out = a.lvlkp + (size_t)image * a.lvlkp_entries + L.kp_offset;
  const int m_out = min(m2, L.kp_cap);
  for (int i = wid; i < m_out; i += VSF_SELECT_THREADS / 64) {
    const uint2 e = B[i];
    const int x0 = (int)(e.y & 0xFFFu), y0 = (int)(e.y >> 12);
    const uint8_t* center = img + (size_t)y0 * pitch + x0;
    const int u = (lane & 31) - 15;
    int m10 = 0, m01 = 0;
    for (int v0 = -15; v0 <= 15; v0 += 2) {
      const int v = v0 + (lane >> 5);
      const int av = v < 0 ? -v : v;
      if (v <= 15 && u <= 15) {
        const int d = umax31(av);
        if (u >= -d && u <= d) {
          const int val = center[v * pitch + u];
          m10 += u * val;
          m01 += v * val;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      m10 += __shfl_xor(m10, o, 64);
      m01 += __shfl_xor(m01, o, 64);
    }
    if (lane == 0) {
      VsfLevelKp kp;
      kp.xy = e.y;
      kp.response = __uint_as_float(e.x);
      kp.angle = fast_atan2_deg((float)m01, (float)m10);
      out[i] = kp;
    }
  }
  if (tid == 0) {
    a.lvl_count[(size_t)image * a.nlevels + level] = m_out;
    if (m2 > L.kp_cap) atomicOr(a.status, 1);
  }
}

}  // namespace

void vsf_launch_select(const VsfDev& d, const VsfGeom& g, const VsfImages& im, hipStream_t s) {
  SelectArgs a;
  a.levels = d.levels;
  a.img0 = im.base;
  a.img0_stride = im.image_stride;
  a.img0_pitch = (int)im.row_stride;
  a.pyr = d.pyr;
  a.pyr_bytes = g.pyr_bytes;
  a.cand = d.cand;
  a.cand_entries = g.cand_entries;
  a.strip_count = d.strip_count;
  a.nstrips = g.nstrips;
  a.scratch = d.scratch;
  a.lvlkp = d.lvlkp;
  a.lvlkp_entries = g.lvlkp_entries;
  a.lvl_count = d.lvl_count;
  a.nlevels = g.nlevels;
  a.status = d.status;
  hipLaunchKernelGGL(orb_select_kernel, dim3(g.nlevels, im.n), dim3(VSF_SELECT_THREADS), 0, s, a);
}
