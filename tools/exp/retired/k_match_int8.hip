// k_match_int8.hip -- RETIRED from the product in round 6 (it was VSF_OPT_MATCH_INT8): round 2's matcher on
// v_mfma_i32_32x32x32_i8, bit-identical to the FP4 form of csrc/k_match.hip and slower (0.26 against 0.17 ms per 256 pairs
// of 2000 x 2000; records: profiles/r05).  Kept for reference: the kernel as it stood inside csrc/k_match.hip (it used that
// file's expand16, merge_top2, imed3, kTile, kTileStride, kChunkRows, kWgQueries, kKeyNone).
// Hamming distances as int8 matrix products on the matrix cores.  With train bits expanded to +64 / -64 and query bits
// to -64 / +64, a train row and a query column at Hamming distance d multiply to  8192 d - 2^20.  A NINTH product per
// tile -- the row's index inside the tile in one k-slot of the train operand against a 1 in the same slot of the query
// operand, on a zero accumulator -- puts that index into every column, so the accumulator ends as the (signed) sort key
//     8192 d - 2^20 + (row inside the tile)
// without a vector instruction: the matrix core produces the key itself and the vector ALU only keeps the two smallest
// per query (v_min_i32 + v_med3_i32 per element).  Keys are kept RELATIVE TO THE CURRENT TILE: before a tile is folded
// the two running minima move down by 32 (rows of earlier tiles become negative offsets; a chunk is at most 4096 rows,
// less than the 8192 a distance step is worth, so the order -- smaller distance first, ties to the lower train index,
// batchDistance's insertion rule -- is the integer order).  Per 32-row tile a wave issues 9 MFMAs (288 cycles of its
// SIMD's matrix pipe) and ~70 vector instructions (280 cycles).
// One v_mfma_i32_32x32x32_i8 covers 32 train rows x 32 queries x 32 bits; a wave owns 32 queries (expanded once into 32
// VGPRs) and walks the train set 32 rows at a time; the workgroup's four waves share each expanded train tile through
// LDS (double buffered, one barrier per tile).  118 registers: four waves per SIMD, and a workgroup finds room beside
// another kernel's waves (round 2's form, 64 queries per wave in 253 registers, waited for whole CUs to drain whenever
// it shared the chip, and was no faster alone: 0.27 against 0.25 ms per 256 pairs of 2000 x 2000).
//   C/D layout (cdna guide): column = lane & 31 (a query), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (a train row).
//   A/B layout: lane (r, h) holds 16 of the 32 k-values of row / column r; which 16 does not matter here as long as
//   both operands are expanded the same way (they are: expand16 of bits [16 h, 16 h + 16) of dword s for step s).
// SPLIT = false: a workgroup walks the whole train set and writes idx2 / dist2.
// SPLIT = true : gridDim.z workgroups share a query tile, each walks one chunk of the train set and merges its top-2 into
//                the query's packed 64-bit key pair (best << 32 | second) kept in the dist2 slot, with a CAS loop (the
//                merge  m1 = min(a1, b1), m2 = min(max(a1, b1), min(a2, b2))  is associative and commutative);
//                knn2_finalize_kernel then unpacks.
// Keys leave the kernel as  distance << 20 | train index.

template <bool SPLIT>
__global__ __launch_bounds__(256, 4) void knn2_kernel(const uint8_t* __restrict__ desc,
                                                      const int32_t* __restrict__ counts, size_t set_stride,
                                                      const int32_t* __restrict__ q_set,
                                                      const int32_t* __restrict__ t_set, int max_rows,
                                                      int32_t* __restrict__ idx2, int32_t* __restrict__ dist2) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[2][kTile * kTileStride];
  const int pair = blockIdx.y;
  const int qs = q_set ? q_set[pair] : 2 * pair, ts = t_set ? t_set[pair] : 2 * pair + 1;
  const int nq = min(counts[qs], max_rows), nt = min(counts[ts], max_rows);
  if ((int)blockIdx.x * kWgQueries >= nq) return;  // whole block idle (uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const uint8_t* Q = desc + (size_t)qs * set_stride;
  const uint32_t* T = reinterpret_cast<const uint32_t*>(desc + (size_t)ts * set_stride);
  // query operand: column c of the wave's tile, bits [16 h, 16 h + 16) of each descriptor dword
  const int qbase = blockIdx.x * kWgQueries + wave * 32;
  v4i qf[8];
  {
    const int q = min(qbase + c, nq - 1);
    const uint4 lo = reinterpret_cast<const uint4*>(Q + (size_t)q * 32)[0];
    const uint4 hi = reinterpret_cast<const uint4*>(Q + (size_t)q * 32)[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int s = 0; s < 8; s++) qf[s] = expand16(w[s] >> (16 * h), 0x000040C0u);  // set bit -> -64, clear -> +64
  }
  // the ninth product's operands: k-slot 0 of the lanes h = 0 carries the row's index (train side) and a 1 (query side)
  const v4i row_a = {h == 0 ? c : 0, 0, 0, 0}, one_b = {h == 0 ? 1 : 0, 0, 0, 0};
  int t_begin = 0, t_end = nt;
  if (SPLIT) {
    const int chunk = ((nt + (int)gridDim.z - 1) / (int)gridDim.z + kSplitAlign - 1) / kSplitAlign * kSplitAlign;
    t_begin = min((int)blockIdx.z * chunk, nt);
    t_end = min(t_begin + chunk, nt);
  }
  uint32_t g1 = 0xFFFFFFFFu, g2 = 0xFFFFFFFFu;  // distance << 20 | train index
  // staging: thread tid expands dword (tid & 7) of train row (tid >> 3) of the tile: 32 bytes at row * 272 + 32 s.
  // Rows past the chunk's end are clamped to its last row (never folded: the last tile's fold checks the row index).
  const int st_row = tid >> 3, st_s = tid & 7;
  auto load_bits = [&](int t0, int c_end) -> uint32_t { return T[(size_t)min(t0 + st_row, c_end - 1) * 8 + st_s]; };
  auto stage = [&](int buf, uint32_t bits) {
    uint8_t* dst = &tile[buf][st_row * kTileStride + 32 * st_s];
    *reinterpret_cast<v4i*>(dst) = expand16(bits, 0x0000C040u);             // set bit -> +64, clear -> -64
    *reinterpret_cast<v4i*>(dst + 16) = expand16(bits >> 16, 0x0000C040u);
  };
  for (int c0 = t_begin; c0 < t_end; c0 += kChunkRows) {
    const int c_end = min(c0 + kChunkRows, t_end);
    const int ntiles = (c_end - c0 + kTile - 1) / kTile;
    int m1 = kKeyNone, m2 = kKeyNone;  // 8192 d - 2^20 + (row - first row of the tile folded last)
    // fold: the running minima move to the new tile's frame, then take its 16 keys
    auto fold = [&](const v16i& a, int t0, bool check) {
      m1 -= kTile;  // (kKeyNone stays far above every key: at most 128 tiles per chunk)
      m2 -= kTile;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const bool valid = !check || t0 + (i & 3) + 8 * (i >> 2) + 4 * h < c_end;
        const int k = valid ? a[i] : kKeyNone;
        m2 = imed3(m1, m2, k);
        m1 = min(m1, k);
      }
    };
    // One pipeline step = tile t: its nine MFMAs (into `n`) are issued one at a time with eight vector-ALU
    // instructions behind each -- the fold of tile t - 1's keys (`p`) and the expansion of tile t + 1 into the other LDS
    // buffer -- so the matrix core and the vector ALU of a SIMD run side by side inside one wave (waves that alternate
    // whole MFMA and VALU phases fall into step with each other and serialise).
    constexpr int kAhead = 4;    // descriptor dwords are fetched this many tiles ahead (L2 latency ~ 2 steps)
    uint32_t bits_next[kAhead];  // tiles t + 1 .. t + kAhead: this thread's descriptor dword of each
    auto step = [&](int t, const v16i& p, v16i& n, bool fold_prev) {
      const int buf = t & 1;
      const uint32_t bits_after = load_bits(c0 + (t + 1 + kAhead) * kTile, c_end);
      const uint8_t* rowp = &tile[buf][c * kTileStride + 16 * h];
      v4i tf[8];
#pragma unroll
      for (int s = 0; s < 8; s++) tf[s] = *reinterpret_cast<const v4i*>(rowp + 32 * s);
      const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      n = __builtin_amdgcn_mfma_i32_32x32x32_i8(row_a, one_b, zero, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 8; s++) n = __builtin_amdgcn_mfma_i32_32x32x32_i8(tf[s], qf[s], n, 0, 0, 0);
      if (fold_prev) fold(p, 0, false);  // (every tile but a chunk's last is full)
      stage(buf ^ 1, bits_next[0]);
#pragma unroll
      for (int k = 0; k + 1 < kAhead; k++) bits_next[k] = bits_next[k + 1];
      bits_next[kAhead - 1] = bits_after;
#pragma unroll
      for (int g = 0; g < 9; g++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // its share of the vector-ALU work
      }
      __syncthreads();
    };
    __syncthreads();  // the previous chunk's last tile has been read
    stage(0, load_bits(c0, c_end));
#pragma unroll
    for (int k = 0; k < kAhead; k++) bits_next[k] = load_bits(c0 + (1 + k) * kTile, c_end);
    __syncthreads();
    v16i accA, accB = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (B is not folded before it is written)
    step(0, accB, accA, false);
    int t = 1;
    for (; t + 1 < ntiles; t += 2) {
      step(t, accA, accB, true);
      step(t + 1, accB, accA, true);
    }
    int t_last;  // the tile m1 / m2 are relative to
    if (t < ntiles) {  // uniform
      step(t, accA, accB, true);
      fold(accB, c0 + t * kTile, true);
      t_last = t;
    } else {
      fold(accA, c0 + (t - 1) * kTile, true);
      t_last = t - 1;
    }
    // chunk keys -> global keys, merged into the running best two:  key + 2^20 + 4096 = 8192 d + (row offset + 4096),
    // the row offset being in (-4096, 32)
    auto global_key = [&](int key) -> uint32_t {
      if (key >= kKeyNone - kChunkRows) return 0xFFFFFFFFu;
      const uint32_t y = (uint32_t)(key + (1 << 20) + 4096);
      return ((y >> 13) << 20) | (uint32_t)((int)(y & 8191u) - 4096 + t_last * kTile + c0);
    };
    merge_top2(g1, g2, global_key(m1), global_key(m2));
  }
  // the two lane halves hold the rows 4 h + ... of every tile: merge them
  {
    const uint32_t o1 = __shfl_xor(g1, 32), o2 = __shfl_xor(g2, 32);
    merge_top2(g1, g2, o1, o2);
  }
  if (h != 0) return;
  const int q = qbase + c;
  if (q >= nq) return;
  const uint32_t b1 = g1, b2 = g2;
  if (SPLIT) {
    if (t_begin >= t_end) return;
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(dist2) + (size_t)pair * max_rows + q;
    unsigned long long seen = *slot;
    while (true) {
      const uint32_t a1 = (uint32_t)(seen >> 32), a2 = (uint32_t)seen;
      const uint32_t n1 = min(a1, b1), n2 = min(max(a1, b1), min(a2, b2));
      const unsigned long long merged = ((unsigned long long)n1 << 32) | n2;
      if (merged == seen) break;
      const unsigned long long prev = atomicCAS(slot, seen, merged);
      if (prev == seen) break;
      seen = prev;
    }
  } else {
    const size_t o = ((size_t)pair * max_rows + q) * 2;
    idx2[o] = b1 == 0xFFFFFFFFu ? -1 : (int32_t)(b1 & 0xFFFFFu);
    idx2[o + 1] = b2 == 0xFFFFFFFFu ? -1 : (int32_t)(b2 & 0xFFFFFu);
    dist2[o] = b1 == 0xFFFFFFFFu ? INT_MAX : (int32_t)(b1 >> 20);
    dist2[o + 1] = b2 == 0xFFFFFFFFu ? INT_MAX : (int32_t)(b2 >> 20);
  }
}

