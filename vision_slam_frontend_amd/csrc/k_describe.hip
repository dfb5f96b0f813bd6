// k_describe.hip -- K8: computeOrbDescriptors (WTA_K = 2, 256-bit rotated BRIEF) and assembly of the final
// cv::KeyPoint / descriptor arrays (features2d/orb.cpp; reached from slam_frontend.cc:274).
//
// One workgroup per (image, level); its four waves walk the level's keypoints, one wave per keypoint.
// Lane l evaluates pattern pairs l, l+64, l+128, l+192, so each 64-lane ballot IS eight consecutive
// descriptor bytes (bit i of byte k = pair 8k+i, LSB first, exactly OpenCV's packing) -- no shuffles, no LDS.
// Per sample: x = px*a - py*b, y = px*b + py*a as separately rounded float ops (no FMA), cvRound
// (round-half-even), one byte load from the blurred level (L2-resident 39x39 neighbourhood).
// The level's slot in the image's level-major output is the sum of the preceding levels' counts.
#include "vsf_internal.h"

#pragma clang fp contract(off)

namespace {

__constant__ int8_t c_pattern31[256 * 4] = {
#include "orb_pattern31.inc"
};

struct DescribeArgs {
  const VsfLevel* levels;
  const uint8_t* blur;
  uint32_t pyr_bytes;
  const VsfLevelKp* lvlkp;
  int lvlkp_entries;
  const int32_t* lvl_count;
  int nlevels;
  int max_keypoints;
  vsf_keypoint* kp_out;
  uint8_t* desc_out;
  int32_t* counts;
  int32_t* status;
};

__global__ __launch_bounds__(256) void orb_describe_kernel(DescribeArgs a) {
  const int level = blockIdx.x, image = blockIdx.y;
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
  if (level == 0 && threadIdx.x == 0) {
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
  for (int i = wid; i < n; i += 4) {
    const int o = before + i;
    if (o >= a.max_keypoints) break;
    const VsfLevelKp k = kps[i];
    const float fx = (float)(int)(k.xy & 0xFFFu) * lscale;  // KeyPoint::pt *= scale
    const float fy = (float)(int)(k.xy >> 12) * lscale;
    float angle = k.angle;
    angle *= (float)(3.14159265358979323846 / 180.f);
    // cos/sin of the float angle taken as the correctly rounded float (double evaluation).
    const float ca = (float)cos((double)angle), sb = (float)sin((double)angle);
    const int cx = __float2int_rn(fx * inv), cy = __float2int_rn(fy * inv);
    const uint8_t* center = img + (size_t)cy * pitch + cx;
    uint64_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int pair = lane + 64 * j;
      const float x0 = (float)c_pattern31[4 * pair + 0], y0 = (float)c_pattern31[4 * pair + 1];
      const float x1 = (float)c_pattern31[4 * pair + 2], y1 = (float)c_pattern31[4 * pair + 3];
      const int ix0 = __float2int_rn(x0 * ca - y0 * sb), iy0 = __float2int_rn(x0 * sb + y0 * ca);
      const int ix1 = __float2int_rn(x1 * ca - y1 * sb), iy1 = __float2int_rn(x1 * sb + y1 * ca);
      const int t0 = center[iy0 * pitch + ix0], t1 = center[iy1 * pitch + ix1];
      w[j] = __ballot(t0 < t1);
    }
    if (lane < 4) {
      const uint64_t v = lane == 0 ? w[0] : lane == 1 ? w[1] : lane == 2 ? w[2] : w[3];
      reinterpret_cast<uint64_t*>(a.desc_out + ((size_t)image * a.max_keypoints + o) * VSF_DESC_BYTES)[lane] = v;
    }
    if (lane == 4) {
      vsf_keypoint kp;
      kp.x = fx;
      kp.y = fy;
      kp.size = 31 * lscale;
      kp.angle = k.angle;
      kp.response = k.response;
      kp.octave = level;
      kp.class_id = -1;
      a.kp_out[(size_t)image * a.max_keypoints + o] = kp;
    }
  }
}

}  // namespace

void vsf_launch_describe(const VsfDev& d, const VsfGeom& g, const VsfImages& im, int max_keypoints,
                         vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts, hipStream_t s) {
  DescribeArgs a;
  a.levels = d.levels;
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
  hipLaunchKernelGGL(orb_describe_kernel, dim3(g.nlevels, im.n), dim3(256), 0, s, a);
}
