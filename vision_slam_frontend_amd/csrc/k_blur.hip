// k_blur.hip -- K7: GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of every pyramid level (CV_8UC1), the
// image the rBRIEF tests sample.  ORB_Impl::detectAndCompute (features2d/orb.cpp) blurs each level in place
// before computeOrbDescriptors; reached from slam_frontend.cc:274.
//
// Restates the 8-bit fixed-point separable path of imgproc/smooth.cpp + filter.cpp: kernel
// round(256 * gauss) = [18 34 49 55 49 34 18] per axis (sum 257, not renormalised); row pass R = sum k_i p_i
// (fits 16 bit); column pass N = sum k_j R_j; result = N / 65536 rounded half-to-even on the columns OpenCV's
// SSE2 SymmColumnVec_32s8u handles ([0, w - w%4)) and half-up on the scalar tail, saturated to 255.
//
#include "vsf_internal.h"

namespace {

__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - 2 - p;
  return p < 0 ? 0 : (p >= len ? len - 1 : p);  // clamp only reachable for len < 4 (one reflection suffices otherwise)
}

// ---------------------------------------------------------------------------------------------------------------
// The blur on the MATRIX cores.  (Round 2's streaming kernel -- a 7-row register window per wave, packed 16-bit column
// sums, v_dot2 row sums -- issued vector-ALU instructions at 79 % of the chip's rate, 86 wave-instructions per 256 pixels;
// it lives on in tools/exp/retired/k_blur_march.hip.)  Both passes are banded-Toeplitz products, so they leave the vector
// ALU altogether:
//   pass 1 (rows of the separable filter = horizontal):  R[y][x] = sum_c  (Img[y][c] - 128) * T[c][x]  + 128 * sum k
//            v_mfma_i32_32x32x32_i8: A = 32 image rows x 32 columns (16 bytes per lane, v_xor 0x80 makes them signed),
//            B = the 32 x 32 band of T (a constant operand: BORDER_REFLECT_101 of the columns is folded into the weights of
//            the level's first / last tiles on the host), C = the bias, so R = sum k_j p_j exactly, 0 .. 65535.  An output
//            tile of 32 columns takes the two operands that cover columns [32 c - 16, 32 c + 48).
//   pass 2 (columns = vertical):  the accumulator layout of pass 1 (column on the lane, 16 rows in the registers) IS
//            the A-operand layout of the next MFMA (cdna guide, "an accumulator tile as the next MFMA's operand"), so R
//            never moves between lanes: its low and high bytes are zero-extended to 16 bit with one v_perm_b32 per pair
//            of values -- a byte b in a 16-bit half is the f16 DENORMAL b * 2^-24, which v_mfma_f32_32x32x16_f16 takes
//            without flushing (tools/exp/mfma_probe.hip: 204 800 sums, all exact) -- and multiplied by the taps k_i
//            (low bytes) resp. 256 k_i (high bytes), both exact f16 numbers; the f32 accumulator starts at -0.5 and ends
//            as N * 2^-24 - 0.5 with N = sum k_i R_i < 2^24.01: every partial sum is an integer multiple of 2^-24 below
//            2^24 of them in magnitude, i.e. exact.
//   rounding: t = acc * 256 + 128 = N / 65536 exactly (below 256; above it saturates), v_cvt_pk_u8_f32 rounds to nearest
//            even, saturates and inserts the byte (probe: 2 083 values incl. every tie) = cvtps2dq + packus of OpenCV's
//            SSE2 column pass; the <= 3 scalar-tail columns of a level (half-up) take floor(t + 0.5) first.
//   layout:  pass 2 leaves an output row on a lane (4 x 4 columns per half wave); two v_permlane32_swap give each lane 16
//            contiguous bytes and the tile is written with ONE 16-byte store per lane: eight whole 128-byte lines of the
//            tiled blurred level.
// A wave computes 26 output rows x 64 columns (two tiles) per step from 32 rows x 96 columns of the source (row reflection
// by address), statelessly: no ring of row sums in registers.  The source reaches the A operands through LDS: loading them
// straight from memory (lane = row: 32 rows x 32 bytes per instruction) kept the texture addresser busy with 32 cache lines
// per instruction and fetched every line 2.7 times (measured: 0.85 ms per step for loads + arithmetic against 0.47 for the
// arithmetic alone).  A workgroup of four waves = 2 bands x 2 steps shares a block of 58 rows x 256 bytes, fetched in
// whole 128-byte rows by buffer_load_dwordx4 ... lds (8 rows per instruction, no registers, no vector-ALU work), double
// buffered, one workgroup barrier per block; the image is XOR-swizzled through the SOURCE address (slot (rho, s) of a piece
// holds chunk s ^ rho of row rho) and the pieces of odd piece rows start 128 bytes later, which makes every ds_read_b128
// of an operand conflict-free (checked for all lane groups).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

constexpr int kMmaRows = VSF_BLUR_MMA_ROWS;  // output rows per step (32 loaded rows - 6)
constexpr int kPieceRowStride = 2 * 1024 + 128;  // LDS bytes per piece row (two 1-KB pieces; odd rows shifted by half a bank row)
constexpr int kBlockBytes = 8 * kPieceRowStride;  // one staged block: 64 rows x 256 bytes

struct BlurMmaArgs {
  const VsfLevel* levels;
  const uint32_t* units;  // level << 24 | band pair << 16 | first double step << 8 | double steps
  int nunits, nimages;
  const uint8_t* img0;
  size_t img0_stride;
  int img0_pitch;
  const uint8_t* pyr;
  uint8_t* blur;
  uint32_t pyr_bytes;
  const uint4* tcol;  // [level's blur_tcol + 4 band + {L0, R0, L1, R1}][64 lanes]: B operands of pass 1
  const uint4* tv;    // [{lo k-step 0, lo k-step 1, hi k-step 0, hi k-step 1}][64 lanes]: B operands of pass 2
  int bias;           // 128 * sum of the taps
};

__device__ __forceinline__ v4i as_v4i(uint4 v) { return __builtin_bit_cast(v4i, v); }

typedef __attribute__((address_space(3))) uint8_t lds_u8;

// TAIL: the band holds columns of the level's scalar tail (x >= blur_vec_end, rounded half-up); wave-uniform, a template
// parameter so that the other bands' code carries none of it.
template <bool TAIL>
__device__ __forceinline__ void blur_mma_body(const BlurMmaArgs& a, const VsfLevel& L, lds_u8* lds, int level, int bp,
                                              int ds0, int nds, int image) {
  const uint8_t* src;
  int pitch;
  if (level == 0) {
    src = a.img0 + (size_t)image * a.img0_stride;
    pitch = a.img0_pitch;
  } else {
    src = a.pyr + (size_t)image * a.pyr_bytes + L.offset;
    pitch = L.pitch;
  }
  uint8_t* dst = a.blur + (size_t)image * a.pyr_bytes + L.offset;
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wx = wave & 1, wy = wave >> 1;
  const int band = 2 * bp + wx;
  const int h = L.h;
  const int nsteps = (h + kMmaRows - 1) / kMmaRows;
  const bool has_band = band * 64 < L.w;  // (an odd number of bands: the pair's second wave only helps loading)
  const __amdgpu_buffer_rsrc_t src_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, pitch * h, 0x00020000);
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, L.pitch * ((h + 7) & ~7), 0x00020000);
  // constant operands (the table is padded to an even number of bands)
  const uint4* tc = a.tcol + ((size_t)L.blur_tcol + (size_t)band * 4) * 64 + lane;
  const v4i TL0 = as_v4i(tc[0]), TR0 = as_v4i(tc[64]), TL1 = as_v4i(tc[128]), TR1 = as_v4i(tc[192]);
  const v8h VL0 = __builtin_bit_cast(v8h, a.tv[lane]), VL1 = __builtin_bit_cast(v8h, a.tv[64 + lane]),
            VH0 = __builtin_bit_cast(v8h, a.tv[128 + lane]), VH1 = __builtin_bit_cast(v8h, a.tv[192 + lane]);
  v16i bias;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    bias[i] = a.bias;
    asm volatile("" : "+v"(bias[i]));  // stays in its registers (the compiler would otherwise re-create it per MFMA)
  }
  const int vec_end = L.blur_vec_end;
  const uint32_t pitch4 = (uint32_t)(L.pitch * 4);

  // ---- staging: this wave loads piece rows 2 wave, 2 wave + 1 (block rows 16 wave .. 16 wave + 15), both piece columns.
  // Lane = slot (rho = lane >> 3, s = lane & 7) of a piece; it fetches chunk s ^ rho of the piece's row rho.
  const int rho = lane >> 3;
  const int src_col = bp * 128 - 64 + 16 * ((lane & 7) ^ rho);  // (left of the image the offset wraps: the load returns 0)
  auto stage = [&](int dstep, int buf) {
    const int y0 = dstep * (2 * kMmaRows) - 3 + 16 * wave + rho;
    const uint32_t o0 = (uint32_t)(reflect101(y0, h) * pitch + src_col);
    const uint32_t o1 = (uint32_t)(reflect101(y0 + 8, h) * pitch + src_col);
    lds_u8* base = lds + buf * kBlockBytes + (2 * wave) * kPieceRowStride;
    // (the second piece column's +128 is added to the lane offset: an instruction offset would move the LDS address too,
    // and a scalar offset is not part of the range check -- row 0's wrapped negative lane offsets must stay out of range
    // for the first piece column only)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, base, 16, o0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, base + 1024, 16, o0 + 128u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, base + kPieceRowStride, 16, o1, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, base + kPieceRowStride + 1024, 16, o1 + 128u, 0, 0, 0);
  };
  // ---- operand reads: block row j = 26 wy + r, chunk 4 wx + 3 + 2 q + hh of the 16 (columns 64 band - 16 + 32 q + 16 hh)
  uint32_t qaddr[3];
  {
    const int j = kMmaRows * wy + r, pr = j >> 3, rj = j & 7;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int chunk = 4 * wx + 3 + 2 * q + hh, pc = chunk >> 3, cc = chunk & 7;
      qaddr[q] = (uint32_t)(pr * kPieceRowStride + pc * 1024 + (rj * 8 + (cc ^ rj)) * 16);
    }
  }
  auto sgn = [](v4i q) -> v4i { return q ^ (int)0x80808080; };
  // pass 2 + rounding + store of one tile: racc = row sums (lane = column, registers = rows)
  auto finish = [&](const v16i& racc, uint32_t row_off, int tile) {
    v4i lo0, lo1, hi0, hi1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      lo0[i] = (int)__builtin_amdgcn_perm((uint32_t)racc[2 * i + 1], (uint32_t)racc[2 * i], 0x0C040C00u);
      hi0[i] = (int)__builtin_amdgcn_perm((uint32_t)racc[2 * i + 1], (uint32_t)racc[2 * i], 0x0C050C01u);
      lo1[i] = (int)__builtin_amdgcn_perm((uint32_t)racc[8 + 2 * i + 1], (uint32_t)racc[8 + 2 * i], 0x0C040C00u);
      hi1[i] = (int)__builtin_amdgcn_perm((uint32_t)racc[8 + 2 * i + 1], (uint32_t)racc[8 + 2 * i], 0x0C050C01u);
    }
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, lo0), VL0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, lo1), VL1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, hi0), VH0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, hi1), VH1, acc, 0, 0, 0);
    // lane = output row, register i = column (i & 3) + 8 (i >> 2) + 4 hh of the tile
    uint32_t d[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float t = acc[4 * j + i] * 256.f;
        if (TAIL) {
          const int col = band * 64 + tile * 32 + 8 * j + 4 * hh + i;
          t = col >= vec_end ? __builtin_floorf(t + 0.5f) : t;
        }
        // (the builtin, not inline asm: v_permlane32_swap must not read a register a vector instruction wrote in the two
        // slots before it, and the compiler's hazard recognizer does not look into asm statements -- an asm here left the
        // swap one slot behind the last v_cvt_pk and lanes 12..15 of a tile's first dwords came out stale now and then)
        w = __builtin_amdgcn_cvt_pk_u8_f32(t, (uint32_t)i, i == 0 ? 0u : w);
      }
      d[j] = w;
    }
    auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
    auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
    // (no branch around the store: a lane without an output row carries an offset beyond the buffer, and the range
    // check drops its write)
    typedef unsigned int u4 __attribute__((__vector_size__(16)));
    const u4 v = {s02[0], s02[1], s13[0], s13[1]};
    // The tile's offset rides in the lane offset (band) and the instruction offset (tile), NOT in the scalar offset: the
    // compiler assumes that a 16-byte store with a scalar offset register may be followed at once by a vector write to
    // its data registers; on gfx950 that overwrote the first data dword of lanes 12..15 / 44..47 now and then (the next
    // tile's first v_perm_b32 landed in it).  Without a scalar offset register it keeps the wait state.
    __builtin_amdgcn_raw_buffer_store_b128(v, dst_rsrc, row_off + (tile << 7), 0, 0);
  };
  auto compute = [&](int buf, int step) {
    v4i q[3];
    const uint32_t bo = (uint32_t)(uintptr_t)lds + (uint32_t)(buf * kBlockBytes);
    // (inline asm: the compiler would put s_waitcnt vmcnt(0) in front of an LDS read it can see while an LDS-DMA is in
    // flight, i.e. wait for the NEXT block's loads.  Reads and their wait are ONE statement: the compiler takes an asm's
    // outputs as available when the statement ends.)
    asm volatile(
        "ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2])
        : "v"(bo + qaddr[0]), "v"(bo + qaddr[1]), "v"(bo + qaddr[2])
        : "memory");
    const v4i q0 = sgn(q[0]), q1 = sgn(q[1]), q2 = sgn(q[2]);
    v16i r0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(q0, TL0, bias, 0, 0, 0);
    r0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(q1, TR0, r0, 0, 0, 0);
    v16i r1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(q1, TL1, bias, 0, 0, 0);
    r1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(q2, TR1, r1, 0, 0, 0);
    const int yo = step * kMmaRows + r;
    const bool writer = r < kMmaRows && yo < h;
    const uint32_t row_off =
        writer ? (uint32_t)(yo >> 2) * pitch4 + (uint32_t)((yo & 3) << 5) + (uint32_t)(16 * hh) + (uint32_t)(band << 8)
               : 0x80000000u;
    finish(r0, row_off, 0);
    finish(r1, row_off, 1);
  };

  // Blocks are double buffered: block d + 1 is requested right after the barrier that says every wave has block d (and
  // has therefore finished reading block d - 1, whose buffer the request overwrites).  vmcnt counts in issue order: the
  // two stores of a step are younger than the four loads issued before it, so "at most 2 outstanding" = loads landed.
  stage(ds0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int d = 0; d < nds; d++) {
    __builtin_amdgcn_s_barrier();
    if (d + 1 < nds) stage(ds0 + d + 1, (d + 1) & 1);  // wave-uniform
    const int step = 2 * (ds0 + d) + wy;
    if (has_band && step < nsteps) compute(d & 1, step);  // wave-uniform
    if (d + 1 < nds) {
      if (has_band && step < nsteps)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

__global__ __launch_bounds__(256, 2) void blur_mma_kernel(BlurMmaArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[2 * kBlockBytes];
  // Workgroups are dealt to the 8 XCDs round-robin by their linear id; all workgroups of an image get ids of one residue
  // class, so the halo rows and columns neighbouring blocks fetch twice come out of ONE XCD's L2 the second time (with the
  // blocks spread over the XCDs, 70 % of the L2 requests missed and the kernel fetched 3.6 GB per 2-GB pyramid pass).
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int image = xcd + 8 * (seq / a.nunits), unit = seq - (seq / a.nunits) * a.nunits;
  if (image >= a.nimages) return;
  const uint32_t ud = a.units[unit];
  const int level = (int)(ud >> 24), bp = (int)((ud >> 16) & 0xFF), ds0 = (int)((ud >> 8) & 0xFF), nds = (int)(ud & 0xFF);
  const VsfLevel L = a.levels[level];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lds_u8* lds = (lds_u8*)lds_raw;
  if ((2 * bp + (wave & 1)) * 64 + 64 > L.blur_vec_end)  // (wave-uniform) the band reaches into the scalar-tail columns
    blur_mma_body<true>(a, L, lds, level, bp, ds0, nds, image);
  else
    blur_mma_body<false>(a, L, lds, level, bp, ds0, nds, image);
}

}  // namespace

void vsf_launch_blur_mma(const VsfDev& d, const VsfGeom& g, const VsfImages& im, const uint32_t* d_units, int nunits,
                         const uint4* d_tcol, const uint4* d_tv, int bias, hipStream_t s) {
  BlurMmaArgs a;
  a.levels = d.levels;
  a.units = d_units;
  a.nunits = nunits;
  a.img0 = im.base;
  a.img0_stride = im.image_stride;
  a.img0_pitch = (int)im.row_stride;
  a.pyr = d.pyr;
  a.blur = d.blur;
  a.pyr_bytes = g.pyr_bytes;
  a.tcol = d_tcol;
  a.tv = d_tv;
  a.bias = bias;
  a.nimages = im.n;
  hipLaunchKernelGGL(blur_mma_kernel, dim3(nunits * ((im.n + 7) / 8 * 8)), dim3(256), 0, s, a);
}
