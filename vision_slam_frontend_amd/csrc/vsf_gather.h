// vsf_gather.h -- device helpers shared by k_fast.hip and k_select.hip: block scan and the merge of the FAST march
// kernel's per-unit candidate segments into ONE raster-ordered list per (image, level).
//
// cv::FAST emits keypoints in raster order (y, then x) and everything downstream (retainBest's permutation)
// depends on that order.  The march kernel works on units = (240-column band) x (32-row strip); a unit's segment
// is in raster order inside the unit and carries the start offset of each of its rows (rowstart[0..32]).  The global
// order interleaves the bands row by row, so the destination of a segment element is
//   prefix[cell(row, band)] + (index inside its row),   cell order = row-major, band-minor,
// with the prefix obtained by a block-wide exclusive scan over the cell counts.
#ifndef VSF_GATHER_H_
#define VSF_GATHER_H_

#include "vsf_internal.h"

// Block-wide exclusive scan over NT threads (NT / 64 waves). lds4 needs NT / 64 ints.
template <int NT>
__device__ __forceinline__ int vsf_block_excl_scan(int v, int* lds4, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds4[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    const int c = lds4[w];
    if (w < wid) base += c;
    tot += c;
  }
  *total = tot;
  __syncthreads();
  return base + inc - v;
}

// Number of candidates of level L of one image (all NT threads call; same value returned to all).
template <int NT>
__device__ __forceinline__ int vsf_level_candidate_count(const VsfLevel& L, const uint16_t* __restrict__ rs_img,
                                                         int* lds4) {
  const int nu = L.nbands * L.nstrips;
  int s = 0;
  for (int u = threadIdx.x; u < nu; u += NT)
    s += rs_img[(size_t)(L.unit0 + u) * VSF_FAST_RS_STRIDE + VSF_FAST_STRIP_ROWS];
  int total;
  vsf_block_excl_scan<NT>(s, lds4, &total);
  return total;
}

// Merges the level's unit segments into raster order; store(dst_index, entry) receives every candidate once, after
// on_total(n) has told every thread the level's candidate count (returned as well).
// cellpre: LDS int array with cellcap >= VSF_FAST_STRIP_ROWS * nbands + 1 entries; rs_lds: LDS copy of the row-start
// tables of the units of one chunk, room for rs_units >= nbands tables.  All NT threads call.
// The level is walked in chunks of whole strips.  Per chunk the units' row-start tables come into LDS with one
// coalesced read, the per-cell counts and their prefix sums are formed from LDS, and the segments are copied one unit
// per wave: the only dependent global read left in front of the copy is the segment itself.
template <int NT, class OnTotal, class Store>
__device__ __forceinline__ int vsf_gather_level(const VsfLevel& L, const uint32_t* __restrict__ cand_img,
                                                const uint16_t* __restrict__ rs_img, int* cellpre, int cellcap,
                                                uint16_t* rs_lds, int rs_units, int* lds4, OnTotal on_total,
                                                Store store) {
  constexpr int SR = VSF_FAST_STRIP_ROWS, RS = VSF_FAST_RS_STRIDE;
  const int tid = threadIdx.x;
  const int nrows = L.y_hi - L.y_lo, nb = L.nbands;
  if (nrows <= 0 || nb <= 0) {
    on_total(0);
    return 0;
  }
  int strips_per_chunk = min(((cellcap - 1) / nb) / SR, rs_units / nb);
  if (strips_per_chunk < 1) strips_per_chunk = 1;  // (capacities are sized so that this cannot happen)
  const int rows_per_chunk = strips_per_chunk * SR;
  // A level that fits one chunk (all but the widest) learns its count from the chunk's own prefix sums; otherwise the
  // count takes one more pass over the row-start tables in memory.
  const bool one_chunk = rows_per_chunk >= nrows;
  int n_level = 0;
  if (!one_chunk) {
    n_level = vsf_level_candidate_count<NT>(L, rs_img, lds4);
    on_total(n_level);
  }
  int base = 0;
  for (int row0 = 0; row0 < nrows; row0 += rows_per_chunk) {
    const int nr = min(rows_per_chunk, nrows - row0);
    const int nc = nr * nb;
    const int s0 = row0 / SR, s1 = (row0 + nr + SR - 1) / SR;
    const int nu = (s1 - s0) * nb;
    // the chunk's row-start tables (contiguous in memory: units s0 * nb .. s1 * nb - 1), as dwords
    {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(rs_img + (size_t)(L.unit0 + s0 * nb) * RS);
      uint32_t* dst = reinterpret_cast<uint32_t*>(rs_lds);
      for (int i = tid; i < nu * (RS / 2); i += NT) dst[i] = src[i];
    }
    __syncthreads();
    // cell (row, band) -> number of candidates, then exclusive prefix sums over the cells
    for (int c = tid; c < nc; c += NT) {
      const int row = row0 + c / nb, b = c - (c / nb) * nb;
      const int s = row / SR, r = row - s * SR;
      const uint16_t* rs = rs_lds + ((s - s0) * nb + b) * RS;
      cellpre[c] = (int)rs[r + 1] - (int)rs[r];
    }
    __syncthreads();
    const int cpt = (nc + NT - 1) / NT;
    const int c_beg = min(tid * cpt, nc), c_end = min(c_beg + cpt, nc);
    int local = 0;
    for (int c = c_beg; c < c_end; c++) local += cellpre[c];
    int total;
    int run = vsf_block_excl_scan<NT>(local, lds4, &total);
    for (int c = c_beg; c < c_end; c++) {
      const int v = cellpre[c];
      cellpre[c] = run;
      run += v;
    }
    if (one_chunk) {
      n_level = total;
      on_total(total);
    }
    __syncthreads();
    // one wave per unit (round robin)
    for (int u = s0 * nb + (tid >> 6); u < s1 * nb; u += NT / 64) {
      const int s = u / nb, b = u - s * nb;
      const uint16_t* rs = rs_lds + (u - s0 * nb) * RS;
      const int tot = rs[SR];
      const uint32_t* seg = cand_img + L.cand_offset + (size_t)u * L.seg_cap;
      // four segment reads in flight per lane (a unit rarely holds more than 256 candidates)
      for (int e0 = tid & 63; e0 < tot; e0 += 256) {
        uint32_t entry[4];
#pragma unroll
        for (int k = 0; k < 4; k++) entry[k] = e0 + 64 * k < tot ? seg[e0 + 64 * k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int e = e0 + 64 * k;
          if (e < tot) {
            const int row = VSF_CAND_Y(entry[k]) - L.y_lo;
            const int r = row - s * SR;
            const int c = (row - row0) * nb + b;
            store(base + cellpre[c] + (e - (int)rs[r]), entry[k]);
          }
        }
      }
    }
    base += total;
    __syncthreads();
  }
  return n_level;
}

#endif  // VSF_GATHER_H_
