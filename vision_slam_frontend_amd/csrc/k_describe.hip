// k_describe.hip -- K6 + K8 and assembly of the final cv::KeyPoint / descriptor arrays (features2d/orb.cpp,
// reached from slam_frontend.cc:274), ONE WAVE PER KEYPOINT:
//
//   K6  ICAngles: m10 = sum u*I, m01 = sum v*I over the radius-15 disc (749 px, rows |v| <= 15 with half-width
//       umax[|v|]) of the UNBLURRED level, angle = fastAtan2(m01, m10).  The disc is cut into 31 rows x 9 aligned
//       dwords = 279 (row, dword) items, 5 per lane; an item's four byte weights come from a table indexed by the
//       byte phase of x0 - 15 (built on the host: vsf_api.hip build_ic_table), so one item costs one 4-byte pixel
//       load, one 8-byte table load and two v_dot4_u32_u8 (sum (u+16)*I and sum I over the in-disc bytes).  Integer
//       sums are order-free, so the wave reduction gives exactly OpenCV's m10 / m01.
//   K8  computeOrbDescriptors (WTA_K = 2, 256-bit rotated BRIEF) on the BLURRED level: lane l evaluates pattern
//       pairs l, l+64, l+128, l+192, so each 64-lane ballot IS eight consecutive descriptor bytes (bit i of byte k
//       = pair 8k+i, LSB first, exactly OpenCV's packing).  Per sample: x = px*a - py*b, y = px*b + py*a as
//       separately rounded float ops (no FMA), cvRound (round-half-even), one byte load.
// A level's keypoints are dealt round-robin to kSplit workgroups x 4 waves; the level's slot in the image's
// level-major output is the sum of the preceding levels' counts.
#include "vsf_internal.h"

#pragma clang fp contract(off)

namespace {

__constant__ int8_t c_pattern31[256 * 4] = {
#include "orb_pattern31.inc"
};

constexpr int kSplit = 2;  // workgroups per (image, level)

struct DescribeArgs {
  const VsfLevel* levels;
  const uint8_t* img0;  // unblurred level 0 = the caller's images
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;   // unblurred levels >= 1
  const uint8_t* blur;
  uint32_t pyr_bytes;
  const uint2* ic_table;  // [4][VSF_IC_ITEMS]: .x = (u + 16) byte weights, .y = 0/1 byte mask
  VsfLevelKp* lvlkp;
  int lvlkp_entries;
  const int32_t* lvl_count;
  int nlevels;
  int max_keypoints;
  vsf_keypoint* kp_out;
  uint8_t* desc_out;
  int32_t* counts;
  int32_t* status;
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

// cos / sin of x in [0, 2 pi + eps] in double precision, error < 1 ulp(double) (the reference takes the libm value
// and rounds it to float): Cody-Waite reduction by multiples of pi/2 (k <= 5, so two constant terms are
// exact to ~1e-33 relative to k) and the fdlibm __kernel_sin / __kernel_cos minimax polynomials on [-pi/4, pi/4].
// Hand-rolled instead of the device libm's sin()/cos() because those carry a Payne-Hanek path whose registers
// (106 VGPRs for this kernel) halve the occupancy of a latency-bound kernel.
__device__ __forceinline__ void sincos_2pi(double x, double* s_out, double* c_out) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632679489655800e+00;  // first 53 bits of pi/2
  const double pio2_lo = 6.12323399573676603587e-17;  // pi/2 - pio2_hi
  const double kd = __builtin_rint(x * two_over_pi);
  const int k = (int)kd;
  const double r = __builtin_fma(-kd, pio2_lo, __builtin_fma(-kd, pio2_hi, x));
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2), S1);
  const double sr = __builtin_fma(z * r, ps, r);
  const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * z * pc);
  // quadrant: x = r + k pi/2
  const bool swap = k & 1;
  const double sv = swap ? cr : sr, cv = swap ? sr : cr;
  *s_out = (k & 2) ? -sv : sv;
  *c_out = ((k + 1) & 2) ? -cv : cv;
}

// K6: one wave per keypoint (rounds of up to 64 keypoints per wave, lane k <-> keypoint k of the round).
__global__ __launch_bounds__(256) void orb_angle_kernel(DescribeArgs a) {
  const int level = blockIdx.x / kSplit, part = blockIdx.x - level * kSplit, image = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const VsfLevel L = a.levels[level];
  const int n = a.lvl_count[(size_t)image * a.nlevels + level];
  const uint8_t* raw;
  int rpitch;
  if (level == 0) {
    raw = a.img0 + (size_t)image * a.img0_stride;
    rpitch = a.img0_pitch;
  } else {
    raw = a.pyr + (size_t)image * a.pyr_bytes + L.offset;
    rpitch = L.pitch;
  }
  VsfLevelKp* kps = a.lvlkp + (size_t)image * a.lvlkp_entries + L.kp_offset;
  // this lane's five disc items: row index (0..30 <-> v = -15..15) and dword index (0..8); items >= 279 carry
  // zero weights and are pointed at row 30 so that they stay inside the image
  int item_off[5], item_v[5];
#pragma unroll
  for (int it = 0; it < 5; it++) {
    const int item = it * 64 + lane;
    const int r = min(item / 9, 30), j = item - (item / 9) * 9;
    item_off[it] = r * rpitch + 4 * j;
    item_v[it] = r - 15;
  }
  const int stride = 4 * kSplit;
  for (int base = part * 4 + wid; base < n; base += 64 * stride) {
    const int cnt = min(64, (n - base + stride - 1) / stride);  // keypoints of this round (wave-uniform)
    uint32_t my_xy = 0;  // lane k holds keypoint k: one load for all records of the round
    if (lane < cnt) my_xy = kps[base + lane * stride].xy;
    float my_angle = 0.f;
    for (int kk = 0; kk < cnt; kk++) {
      const uint32_t xy = (uint32_t)__builtin_amdgcn_readlane((int)my_xy, kk);
      const int x0 = (int)(xy & 0xFFFu), y0 = (int)(xy >> 12);
      const int xs = x0 - 15;
      const uint8_t* abase = raw + (size_t)(y0 - 15) * rpitch + (xs & ~3);
      const uint2* tab = a.ic_table + (xs & 3) * VSF_IC_ITEMS;
      int m10 = 0, m01 = 0;
#pragma unroll
      for (int it = 0; it < 5; it++) {
        const uint32_t px = *reinterpret_cast<const uint32_t*>(abase + (uint32_t)item_off[it]);
        const uint2 t = tab[(uint32_t)(it * 64 + lane)];
        const int sw = (int)__builtin_amdgcn_udot4(px, t.x, 0u, false);
        const int sm = (int)__builtin_amdgcn_udot4(px, t.y, 0u, false);
        m10 += sw - 16 * sm;
        m01 += item_v[it] * sm;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        m10 += __shfl_xor(m10, off, 64);
        m01 += __shfl_xor(m01, off, 64);
      }
      const float kp_angle = fast_atan2_deg((float)m01, (float)m10);
      if (lane == kk) my_angle = kp_angle;
    }
    // cos / sin, one keypoint per lane (the float nearest to the double-precision value)
    const float my_rad = my_angle * (float)(3.14159265358979323846 / 180.f);
    double sd, cd;
    sincos_2pi((double)my_rad, &sd, &cd);
    if (lane < cnt) {
      VsfLevelKp* k = kps + base + lane * stride;
      k->angle = my_angle;
      k->ca = (float)cd;
      k->sb = (float)sd;
    }
  }
}

// K8 + output assembly: one wave per keypoint.
__global__ __launch_bounds__(256) void orb_describe_kernel(DescribeArgs a) {
  const int level = blockIdx.x / kSplit, part = blockIdx.x - level * kSplit, image = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int32_t* lc = a.lvl_count + (size_t)image * a.nlevels;
  // base = sum of counts of the levels before this one; total = all levels (wave-parallel, nlevels <= 64)
  const int mine = lane < a.nlevels ? lc[lane] : 0;
  int before = lane < level ? mine : 0, total = mine;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o, 64);
    total += __shfl_xor(total, o, 64);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.counts[image] = total;
    if (total > a.max_keypoints) atomicOr(a.status, 1);
  }
  const VsfLevel L = a.levels[level];
  const int n = lc[level];
  const uint8_t* img = a.blur + (size_t)image * a.pyr_bytes + L.offset;
  const int pitch = L.pitch;
  const VsfLevelKp* kps = a.lvlkp + (size_t)image * a.lvlkp_entries + L.kp_offset;
  const float lscale = L.scale;
  const float inv = 1.f / lscale;
  // this lane's four pattern pairs (loop invariant)
  float pat[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int pair = lane + 64 * j;
#pragma unroll
    for (int c = 0; c < 4; c++) pat[j][c] = (float)c_pattern31[4 * pair + c];
  }
  const int stride = 4 * kSplit;
  for (int base = part * 4 + wid; base < n; base += 64 * stride) {
    const int cnt = min(64, (n - base + stride - 1) / stride);  // keypoints of this round (wave-uniform)
    const int cnt_out = min(cnt, (a.max_keypoints - (before + base) + stride - 1) / stride);  // that fit the output
    if (cnt_out <= 0) break;
    VsfLevelKp mine_kp{0u, 0.f, 0.f, 1.f, 0.f};  // lane k holds keypoint k: one load for all records of the round
    if (lane < cnt_out) mine_kp = kps[base + lane * stride];
    for (int kk = 0; kk < cnt_out; kk++) {
      const int o = before + base + kk * stride;
      const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine_kp.ca), kk));
      const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine_kp.sb), kk));
      const uint32_t xy = (uint32_t)__builtin_amdgcn_readlane((int)mine_kp.xy, kk);
      const float fx = (float)(int)(xy & 0xFFFu) * lscale, fy = (float)(int)(xy >> 12) * lscale;
      const int cx = __float2int_rn(fx * inv), cy = __float2int_rn(fy * inv);
      // patch origin (centre - 19 rows - 19 columns; |rotated pattern| <= 18.4): lane offsets are non-negative
      const uint8_t* corner = img + (size_t)(cy - 19) * pitch + (cx - 19);
      uint64_t w[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float x0f = pat[j][0], y0f = pat[j][1], x1f = pat[j][2], y1f = pat[j][3];
        const int ix0 = __float2int_rn(x0f * ca - y0f * sb), iy0 = __float2int_rn(x0f * sb + y0f * ca);
        const int ix1 = __float2int_rn(x1f * ca - y1f * sb), iy1 = __float2int_rn(x1f * sb + y1f * ca);
        const int t0 = corner[(uint32_t)((iy0 + 19) * pitch + ix0 + 19)];
        const int t1 = corner[(uint32_t)((iy1 + 19) * pitch + ix1 + 19)];
        w[j] = __ballot(t0 < t1);
      }
      if (lane < 4) {
        const uint64_t v = lane == 0 ? w[0] : lane == 1 ? w[1] : lane == 2 ? w[2] : w[3];
        reinterpret_cast<uint64_t*>(a.desc_out + ((size_t)image * a.max_keypoints + o) * VSF_DESC_BYTES)[lane] = v;
      }
    }
    // the round's cv::KeyPoint records, one per lane
    if (lane < cnt_out) {
      vsf_keypoint kp;
      kp.x = (float)(int)(mine_kp.xy & 0xFFFu) * lscale;  // KeyPoint::pt *= scale
      kp.y = (float)(int)(mine_kp.xy >> 12) * lscale;
      kp.size = 31 * lscale;
      kp.angle = mine_kp.angle;
      kp.response = mine_kp.response;
      kp.octave = level;
      kp.class_id = -1;
      a.kp_out[(size_t)image * a.max_keypoints + before + base + lane * stride] = kp;
    }
  }
}

}  // namespace

void vsf_launch_describe(const VsfDev& d, const VsfGeom& g, const VsfImages& im, int max_keypoints,
                         vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts, hipStream_t s) {
  DescribeArgs a;
  a.levels = d.levels;
  a.img0 = im.base;
  a.img0_stride = im.image_stride;
  a.img0_pitch = (int)im.row_stride;
  a.pyr = d.pyr;
  a.ic_table = d.ic_table;
  a.blur = d.blur;
  a.pyr_bytes = g.pyr_bytes;
  a.lvlkp = d.lvlkp;
  a.lvlkp_entries = g.lvlkp_entries;
  a.lvl_count = d.lvl_count;
  a.nlevels = g.nlevels;
  a.max_keypoints = max_keypoints;
  a.kp_out = d_kp;
  a.desc_out = d_desc;
  a.counts = d_counts;
  a.status = d.status;
  hipLaunchKernelGGL(orb_angle_kernel, dim3(g.nlevels * kSplit, im.n), dim3(256), 0, s, a);
  hipLaunchKernelGGL(orb_describe_kernel, dim3(g.nlevels * kSplit, im.n), dim3(256), 0, s, a);
}
