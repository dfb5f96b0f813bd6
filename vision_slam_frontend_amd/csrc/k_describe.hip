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
// Both run in one kernel over the image's keypoints in output order (level-major; a level's slot is the sum of the
// preceding levels' counts), four consecutive keypoints per wave.
#include <algorithm>

#include "vsf_internal.h"

#pragma clang fp contract(off)

namespace {

__constant__ int8_t c_pattern31[256 * 4] = {
#include "orb_pattern31.inc"
};

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
  int status_stride;  // 0: one word for the call; 1: a word per image (the ObserveImage queue: an overflow stays on its own frame)
  int nblocks, nimages;  // workgroups per image, images
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

typedef float f2 __attribute__((ext_vector_type(2)));

// Consecutive output keypoints per wave (template parameter KPW).  At 2000 keypoints per image, per 256-frame step: 2 -> 0.89
// ms, 4 -> 0.83, 8 -> 0.85, 16 -> 0.89, 32 -> 0.97: the kernel lives on waves in flight, not on amortised prologues -- FOUR.
// At the reference's 10 000 there are waves enough either way and the prologue counts: 8 -> 3.07 ms against 3.20 (the step
// 11.77 -> 11.51 ms; at 2000 the step LOSES 2 % with 8) -- EIGHT for large batches of >= 4096 keypoints per image (the
// launcher decides).

// K6 + K8 + output assembly.  The image's keypoints are numbered in output order (level-major, retainBest order
// inside a level); wave w of the image takes numbers [4 w, 4 w + 4), whatever levels they belong to: lane k < 4
// looks up keypoint k's level (binary search in the wave-scanned level counts), record and level geometry once, and
// the wave then walks its keypoints with v_readlane broadcasts.  (One workgroup per (level, image) left most waves
// with two or three keypoints and a prologue longer than their work.)
template <int kKpPerWave>
__global__ __launch_bounds__(256) void orb_orient_describe_kernel(DescribeArgs a) {
  constexpr int kWinRows = 39, kWinPitch = 64, kWinBytes = kWinRows * kWinPitch;
  __shared__ __attribute__((aligned(16))) uint8_t s_win[4][2][kWinBytes];
  // Workgroups are dealt to the 8 XCDs round-robin by their linear id; all workgroups of an image are given ids of
  // one residue class so that an image's pyramid levels are pulled into ONE XCD's L2 (speed only, not correctness).
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int image = xcd + 8 * (seq / a.nblocks), block = seq - (seq / a.nblocks) * a.nblocks;
  if (image >= a.nimages) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int32_t* lc = a.lvl_count + (size_t)image * a.nlevels;
  // inclusive scan of the level counts over the lanes (nlevels <= 64)
  const int mine = lane < a.nlevels ? lc[lane] : 0;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  const int total = __shfl(incl, 63, 64);
  if (block == 0 && threadIdx.x == 0) {
    a.counts[image] = total;
    if (total > a.max_keypoints) atomicOr(a.status + (size_t)image * a.status_stride, 1);
  }
  const int n_out = min(total, a.max_keypoints);
  const int o_begin = (block * 4 + wid) * kKpPerWave;
  if (o_begin >= n_out) return;  // wave-uniform
  const int cnt = min(kKpPerWave, n_out - o_begin);

  // ---- lane k <-> keypoint o_begin + k: level, record, level geometry ----
  const int o_k = o_begin + min(lane, cnt - 1);
  int level_k = 0;  // number of levels that end at or before o_k
#pragma unroll
  for (int step = 32; step > 0; step >>= 1) {
    const int t = level_k + step;
    const int v = __shfl(incl, t - 1, 64);  // (lanes >= nlevels hold `total` > o_k)
    if (v <= o_k) level_k = t;
  }
  const int idx_k = o_k - __shfl(incl - mine, level_k, 64);
  const VsfLevel Lk = a.levels[level_k];
  const VsfLevelKp* rec_p = a.lvlkp + (size_t)image * a.lvlkp_entries + Lk.kp_offset + idx_k;
  const uint32_t xy_k = rec_p->xy;
  const float resp_k = rec_p->response;
  const float scale_k = Lk.scale, inv_k = 1.f / scale_k;
  const uint32_t off_k = Lk.offset;
  const int pitch_k = Lk.pitch;

  // ---- K6: ICAngles on the unblurred level.  The lane's five disc items: row (0..30 <-> v = -15..15) and dword
  // (0..8); items >= 279 carry zero weights and are pointed at row 30 so that they stay inside the image.
  int item_r[5], item_j4[5];
#pragma unroll
  for (int it = 0; it < 5; it++) {
    const int item = it * 64 + lane;
    item_r[it] = min(item / 9, 30);
    item_j4[it] = 4 * (item - (item / 9) * 9);
  }
  float my_angle = 0.f;
  // The five pixel dwords and table entries of keypoint k + 1 are requested before keypoint k is reduced: the loop
  // is a chain of memory round trips otherwise (the levels do not fit the caches).
  uint32_t px[5];
  uint2 tw[5];
  auto request = [&](int kk) {
    const uint32_t xy = (uint32_t)__builtin_amdgcn_readlane((int)xy_k, kk);
    const int lvl = __builtin_amdgcn_readlane(level_k, kk);
    const uint8_t* raw;
    int rpitch;
    if (lvl == 0) {
      raw = a.img0 + (size_t)image * a.img0_stride;
      rpitch = a.img0_pitch;
    } else {
      raw = a.pyr + (size_t)image * a.pyr_bytes + (uint32_t)__builtin_amdgcn_readlane((int)off_k, kk);
      rpitch = __builtin_amdgcn_readlane(pitch_k, kk);
    }
    const int xs = (int)(xy & 0xFFFu) - 15, y0 = (int)(xy >> 12);
    const uint8_t* abase = raw + (size_t)(y0 - 15) * rpitch + (xs & ~3);
    const uint2* tab = a.ic_table + (xs & 3) * VSF_IC_ITEMS;
#pragma unroll
    for (int it = 0; it < 5; it++) {
      px[it] = *reinterpret_cast<const uint32_t*>(abase + (__umul24((uint32_t)item_r[it], (uint32_t)rpitch) + (uint32_t)item_j4[it]));  // (24-bit multiply: full rate)
      tw[it] = tab[(uint32_t)(it * 64 + lane)];
    }
  };
  // sum over the 64 lanes without LDS: four DPP steps leave every lane of a 16-lane row with the row's sum, the four
  // row sums meet on the scalar unit
  auto wave_sum = [](int v) -> int {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
  };
  request(0);
  for (int kk = 0; kk < cnt; kk++) {
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int it = 0; it < 5; it++) {
      const int sw = (int)__builtin_amdgcn_udot4(px[it], tw[it].x, 0u, false);
      const int sm = (int)__builtin_amdgcn_udot4(px[it], tw[it].y, 0u, false);
      m10 += sw - 16 * sm;
      m01 += (item_r[it] - 15) * sm;
    }
    if (kk + 1 < cnt) request(kk + 1);
    const float kp_angle = fast_atan2_deg((float)wave_sum(m01), (float)wave_sum(m10));
    if (lane == kk) my_angle = kp_angle;
  }
  // cos / sin, one keypoint per lane (the float nearest to the double-precision value)
  const float my_rad = my_angle * (float)(3.14159265358979323846 / 180.f);
  double sd, cd;
  sincos_2pi((double)my_rad, &sd, &cd);
  const float ca_k = (float)cd, sb_k = (float)sd;
  if (lane < cnt) {  // (kept in the level record for vsf_debug_level_keypoints)
    VsfLevelKp* k = a.lvlkp + (size_t)image * a.lvlkp_entries + Lk.kp_offset + idx_k;
    k->angle = my_angle;
    k->ca = ca_k;
    k->sb = sb_k;
  }

  // ---- K8: rotated BRIEF on the blurred level ----
  // this lane's four pattern pairs
  f2 patx[4], paty[4];  // (x0, x1), (y0, y1) of pair lane + 64 j
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int pair = lane + 64 * j;
    patx[j] = f2{(float)c_pattern31[4 * pair + 0], (float)c_pattern31[4 * pair + 2]};
    paty[j] = f2{(float)c_pattern31[4 * pair + 1], (float)c_pattern31[4 * pair + 3]};
  }
  // A keypoint's 512 pattern points fall in the 39 x 39 window around it (|rotated pattern| <= 18.4).  Gathering them
  // straight from memory costs the L1 one tag lookup per distinct cache line per load instruction (~40 lines x 8
  // loads per keypoint: the kernel ran at that rate); instead the window is copied once (39 rows x 64 bytes from a
  // 16-byte aligned column: 156 16-byte pieces, three per lane) into a wave-private LDS tile and the points are
  // gathered from LDS.  The blurred levels are stored in 4 x 32 tiles (k_blur.hip), so the window is ~24 cache lines.
  // The copy of keypoint k + 1 is in flight while keypoint k is gathered (two tiles).
  // the lane's 16-byte pieces of a window: piece p = lane + 64 r -> row p / 4, segment p % 4
  const int pc0 = lane, pc1 = lane + 64, pc2 = min(lane + 128, kWinRows * 4 - 1);  // (spare lanes repeat the last piece)
  const int prow0 = pc0 >> 2, prow1 = pc1 >> 2, prow2 = pc2 >> 2;
  const int pseg0 = pc0 & 3, pseg1 = pc1 & 3, pseg2 = pc2 & 3;
  const int pdst0 = prow0 * kWinPitch + 16 * pseg0, pdst1 = prow1 * kWinPitch + 16 * pseg1,
            pdst2 = prow2 * kWinPitch + 16 * pseg2;
  // the window centres, one keypoint per lane (KeyPoint::pt scaled up and back down, as computeDescriptors sees it)
  const int cx_k = __float2int_rn((float)(int)(xy_k & 0xFFFu) * scale_k * inv_k);
  const int cy_k = __float2int_rn((float)(int)(xy_k >> 12) * scale_k * inv_k);
  auto centre = [&](int kk, int* cx, int* cy) {
    *cx = __builtin_amdgcn_readlane(cx_k, kk);
    *cy = __builtin_amdgcn_readlane(cy_k, kk);
  };
  uint4 q0, q1, q2;
  auto fetch = [&](int kk) {
    int cx, cy;
    centre(kk, &cx, &cy);
    const int pitch = __builtin_amdgcn_readlane(pitch_k, kk);
    const uint8_t* lvl = a.blur + (size_t)image * a.pyr_bytes + (uint32_t)__builtin_amdgcn_readlane((int)off_k, kk);
    const int ya = cy - 19, tx = (cx - 19) >> 4, tx_max = (pitch >> 4) - 1;  // wave-uniform (16-byte segments)
    auto piece = [&](int prow, int pseg) -> uint4 {
      const int y = ya + prow, t = min(tx + pseg, tx_max);  // (a segment past the row end is never gathered)
      // VSF_BLUR_TILE_OFFSET(pitch, 16 t, y) with a full-rate 24-bit multiply
      const uint32_t off = __umul24((uint32_t)(y >> 2), (uint32_t)(pitch * 4)) + ((uint32_t)(t >> 1) << 7) +
                           ((uint32_t)(y & 3) << 5) + ((uint32_t)(t & 1) << 4);
      return *reinterpret_cast<const uint4*>(lvl + off);
    };
    q0 = piece(prow0, pseg0);
    q1 = piece(prow1, pseg1);
    q2 = piece(prow2, pseg2);
  };
  fetch(0);
  for (int kk = 0; kk < cnt; kk++) {
    uint8_t* win = s_win[wid][kk & 1];
    *reinterpret_cast<uint4*>(win + pdst0) = q0;
    *reinterpret_cast<uint4*>(win + pdst1) = q1;
    *reinterpret_cast<uint4*>(win + pdst2) = q2;
    if (kk + 1 < cnt) fetch(kk + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca_k), kk));
    const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sb_k), kk));
    int cx, cy;
    centre(kk, &cx, &cy);
    // tile origin = (centre - 19 rows, (centre - 19 columns) rounded down to 16): lane offsets are non-negative.
    // The two points of a pair are rotated together on packed floats (v_pk_mul_f32 / v_pk_add_f32: every product and
    // sum rounded by itself, as in the scalar code); cvRound = round to nearest even comes out of one more addition:
    // |coordinate| < 2^22, so  c + 1.5 * 2^23  is rounded to an integer by the addition itself and its bit pattern is
    // 0x4B400000 + cvRound(c).  Row * 64 + column of the two patterns, less 65 * 0x4B400000, is the window offset.
    // (formed as a 32-bit LDS address that wraps around, not as an index into the window array: out of range as an index)
    typedef __attribute__((address_space(3))) const uint8_t lds_u8;
    const uint32_t corner = (uint32_t)(size_t)(lds_u8*)win + (uint32_t)(19 * kWinPitch + 19 + ((cx - 19) & 15)) -
                            65u * 0x4B400000u;
    const f2 ca2 = {ca, ca}, sb2 = {sb, sb};
    const f2 magic = {12582912.0f, 12582912.0f};
    uint64_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f2 xa = patx[j] * ca2, yb = paty[j] * sb2, xb = patx[j] * sb2, ya = paty[j] * ca2;
      const f2 fx = (xa - yb) + magic, fy = (xb + ya) + magic;
      // (__float_as_uint of a copy: __builtin_bit_cast applied to a vector ELEMENT comes out undefined with this compiler)
      const float fx0 = fx.x, fx1 = fx.y, fy0 = fy.x, fy1 = fy.y;
      const uint32_t o0 = (__float_as_uint(fy0) << 6) + __float_as_uint(fx0) + corner;
      const uint32_t o1 = (__float_as_uint(fy1) << 6) + __float_as_uint(fx1) + corner;
      const int t0 = *(lds_u8*)(size_t)o0;
      const int t1 = *(lds_u8*)(size_t)o1;
      w[j] = __ballot(t0 < t1);
    }
    if (lane < 4) {
      const uint64_t v = lane == 0 ? w[0] : lane == 1 ? w[1] : lane == 2 ? w[2] : w[3];
      reinterpret_cast<uint64_t*>(a.desc_out + ((size_t)image * a.max_keypoints + o_begin + kk) * VSF_DESC_BYTES)[lane] = v;
    }
  }
  // the cv::KeyPoint records, one per lane
  if (lane < cnt) {
    vsf_keypoint kp;
    kp.x = (float)(int)(xy_k & 0xFFFu) * scale_k;  // KeyPoint::pt *= scale
    kp.y = (float)(int)(xy_k >> 12) * scale_k;
    kp.size = 31 * scale_k;
    kp.angle = my_angle;
    kp.response = resp_k;
    kp.octave = level_k;
    kp.class_id = -1;
    a.kp_out[(size_t)image * a.max_keypoints + o_begin + lane] = kp;
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
  a.status_stride = d.status_stride;
  // (eight per wave needs a launch that still fills the chip with waves: 512 images x 10 000 keypoints; 64-192 images of
  // 8 000 -- 1920x1080 -- run 1-2 % faster with four)
  const int kpw = (max_keypoints >= 4096 && (long)im.n * max_keypoints >= 3000000L) ? 8 : 4;
  a.nblocks = std::max((max_keypoints + 4 * kpw - 1) / (4 * kpw), 1);
  a.nimages = im.n;
  if (kpw == 8)
    hipLaunchKernelGGL(orb_orient_describe_kernel<8>, dim3(a.nblocks * ((im.n + 7) / 8 * 8)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(orb_orient_describe_kernel<4>, dim3(a.nblocks * ((im.n + 7) / 8 * 8)), dim3(256), 0, s, a);
}
