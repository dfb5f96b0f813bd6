// Host check of vision_slam_frontend_amd/csrc/vsf_bitslice.h: every piece against its per-pixel definition, then the
// whole forward scheme (a march over the rows of an image, lanes side by side) against the FAST-9/16 segment test as
// cv::FAST_t<16> states it (9 contiguous circle pixels all > v + t or all < v - t).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vsf_bitslice.h"

using namespace vsf_bs;

static uint32_t rng_state = 12345;
static uint32_t rnd() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return rng_state >> 8;
}

static int fails = 0;
#define CHECK(c, ...)                                  \
  do {                                                 \
    if (!(c)) {                                        \
      if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } \
      ++fails;                                         \
    }                                                  \
  } while (0)

static void planes_of(const uint8_t* px, uint32_t (&p)[8]) {  // definition: plane b bit i = bit b of pixel i
  for (int b = 0; b < 8; b++) {
    p[b] = 0;
    for (int i = 0; i < 32; i++) p[b] |= (uint32_t)((px[i] >> b) & 1) << i;
  }
}

static void test_transpose() {
  for (int it = 0; it < 2000; it++) {
    uint8_t px[32];
    for (int i = 0; i < 32; i++) px[i] = it < 8 ? (uint8_t)(1u << it) * (i == (it * 5) % 32) : (uint8_t)rnd();
    uint32_t w[8], p[8], want[8];
    memcpy(w, px, 32);
    transpose_row(w, p);
    planes_of(px, want);
    for (int b = 0; b < 8; b++) CHECK(p[b] == want[b], "transpose it %d plane %d: %08x != %08x", it, b, p[b], want[b]);
  }
}

static void test_add_sub_greater() {
  for (int t = 0; t < 256; t++) {
    uint32_t tm[8];
    threshold_masks(t, tm);
    for (int it = 0; it < 40; it++) {
      uint8_t a[32], b[32];
      for (int i = 0; i < 32; i++) {
        a[i] = (uint8_t)rnd();
        b[i] = (it & 1) ? (uint8_t)(a[i] + (int)(rnd() % 5) - 2) : (uint8_t)rnd();
      }
      if (it == 0)
        for (int i = 0; i < 32; i++) a[i] = (uint8_t)(i * 8 + (i & 7)), b[i] = (uint8_t)(255 - i);
      uint32_t A[8], B[8], hi[8], lo[8];
      planes_of(a, A);
      planes_of(b, B);
      saturating_add_sub(A, tm, hi, lo);
      uint8_t whi[32], wlo[32];
      for (int i = 0; i < 32; i++) {
        whi[i] = (uint8_t)(a[i] + t > 255 ? 255 : a[i] + t);
        wlo[i] = (uint8_t)(a[i] - t < 0 ? 0 : a[i] - t);
      }
      uint32_t WH[8], WL[8];
      planes_of(whi, WH);
      planes_of(wlo, WL);
      for (int k = 0; k < 8; k++) {
        CHECK(hi[k] == WH[k], "sat add t %d plane %d", t, k);
        CHECK(lo[k] == WL[k], "sat sub t %d plane %d", t, k);
      }
      uint32_t g = greater(A, B), wg = 0;
      for (int i = 0; i < 32; i++) wg |= (uint32_t)(a[i] > b[i]) << i;
      CHECK(g == wg, "greater");
    }
  }
}

static void test_arc9() {
  for (int it = 0; it < 20000; it++) {
    uint32_t m[16];
    for (int k = 0; k < 16; k++) m[k] = (rnd() << 16) ^ rnd() ^ ((it & 3) ? ((rnd() << 16) | rnd()) : 0);
    if (it & 1)
      for (int k = 0; k < 16; k++) m[k] |= (rnd() << 16) | rnd();  // denser
    const uint32_t got = arc9(m);
    uint32_t want = 0;
    for (int i = 0; i < 32; i++) {
      bool any = false;
      for (int s = 0; s < 16 && !any; s++) {
        bool all = true;
        for (int k = 0; k < 9; k++) all = all && ((m[(s + k) & 15] >> i) & 1);
        any = all;
      }
      want |= (uint32_t)any << i;
    }
    CHECK(got == want, "arc9");
  }
}

// the whole scheme on an image of NL lanes x 32 pixels
static void test_image(int W, int H, int t, int kind) {
  const int NL = W / 32;
  std::vector<uint8_t> img((size_t)W * H);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int v;
      switch (kind) {
        case 0: v = rnd() & 255; break;
        case 1: v = 128 + (int)(rnd() % 61) - 30; break;
        case 2: v = ((x / 5 + y / 7) & 1) ? 200 + (int)(rnd() % 7) : 40 + (int)(rnd() % 7); break;
        case 3: v = (rnd() % 10 == 0) ? (int)(rnd() & 255) : 100; break;
        default: v = ((x ^ y) & 1) ? 255 : 0; break;
      }
      img[(size_t)y * W + x] = (uint8_t)v;
    }
  uint32_t tm[8];
  threshold_masks(t, tm);
  // planes of every row and lane
  std::vector<uint32_t> P((size_t)H * NL * 8);
  for (int y = 0; y < H; y++)
    for (int l = 0; l < NL; l++) {
      uint32_t w[8], p[8];
      memcpy(w, &img[(size_t)y * W + 32 * l], 32);
      transpose_row(w, p);
      memcpy(&P[((size_t)y * NL + l) * 8], p, 32);
    }
  std::vector<FastHistory> hist(NL);
  for (auto& h : hist) h.clear();
  std::vector<uint32_t> HI(NL * 8), LO(NL * 8), MB(NL * 8), MD(NL * 8);
  const uint32_t zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = 0; a + 3 < H; a++) {
    for (int l = 0; l < NL; l++) {
      uint32_t I[8], hi[8], lo[8];
      memcpy(I, &P[((size_t)a * NL + l) * 8], 32);
      saturating_add_sub(I, tm, hi, lo);
      memcpy(&HI[l * 8], hi, 32);
      memcpy(&LO[l * 8], lo, 32);
    }
    auto arr = [&](std::vector<uint32_t>& v, int l) -> const uint32_t(&)[8] {
      if (l < 0 || l >= NL) return zero8;
      return *reinterpret_cast<const uint32_t(*)[8]>(&v[l * 8]);
    };
    for (int l = 0; l < NL; l++) {
      uint32_t rows[4][8], mb[8], md[8];
      for (int r = 0; r < 4; r++) memcpy(rows[r], &P[((size_t)(a + r) * NL + l) * 8], 32);
      forward_masks(rows, arr(HI, l), arr(HI, l - 1), arr(HI, l + 1), arr(LO, l), arr(LO, l - 1), arr(LO, l + 1), mb, md);
      memcpy(&MB[l * 8], mb, 32);
      memcpy(&MD[l * 8], md, 32);
    }
    for (int l = 0; l < NL; l++) {
      uint32_t br, dk;
      corner_masks(arr(MB, l), arr(MB, l - 1), arr(MB, l + 1), arr(MD, l), arr(MD, l - 1), arr(MD, l + 1), hist[l], &br, &dk);
      if (a >= 3) {
        for (int i = 0; i < 32; i++) {
          const int x = 32 * l + i;
          if (x < 3 || x >= W - 3) continue;
          const int v = img[(size_t)a * W + x];
          bool wb = false, wd = false;
          for (int s = 0; s < 16; s++) {
            bool ab = true, ad = true;
            for (int k = 0; k < 9; k++) {
              const int c = (s + k) & 15;
              const int q = img[(size_t)(a + CIRCLE_DY[c]) * W + x + CIRCLE_DX[c]];
              ab = ab && q > v + t;
              ad = ad && q < v - t;
            }
            wb = wb || ab;
            wd = wd || ad;
          }
          CHECK(((br >> i) & 1) == (uint32_t)wb, "bright W %d t %d kind %d at (%d, %d)", W, t, kind, x, a);
          CHECK(((dk >> i) & 1) == (uint32_t)wd, "dark W %d t %d kind %d at (%d, %d)", W, t, kind, x, a);
          CHECK(!(wb && wd), "both polarities at (%d, %d)", x, a);
        }
      }
      uint32_t mb[8], md[8];
      memcpy(mb, &MB[l * 8], 32);
      memcpy(md, &MD[l * 8], 32);
      hist[l].push(mb, md);
    }
  }
}

int main() {
  test_transpose();
  test_add_sub_greater();
  test_arc9();
  const int ts[] = {0, 1, 7, 10, 20, 21, 64, 127, 128, 200, 254, 255};
  for (int kind = 0; kind < 5; kind++)
    for (int t : ts) test_image(96, 40, t, kind);
  test_image(256, 64, 20, 0);
  test_image(256, 64, 20, 1);
  if (fails) {
    printf("%d checks failed\n", fails);
    return 1;
  }
  printf("bitslice ok\n");
  return 0;
}
