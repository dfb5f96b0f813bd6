// Checks csrc/vsf_select.h (the order-exact restatement used on the GPU) against the host libstdc++:
// std::nth_element, std::partition, std::sort and the retainBest composition must produce the SAME
// permutation, element for element, on random, tie-heavy, structured and adversarial inputs.
#define VSF_SELECT_TRACE 1
int vsf_sel_trace_fallbacks = 0;
#include "../../vision_slam_frontend_amd/csrc/vsf_select.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

struct E {
  float key;
  int id;
};
static bool operator==(const E& a, const E& b) { return a.key == b.key && a.id == b.id; }
struct Greater {
  bool operator()(const E& a, const E& b) const { return a.key > b.key; }
};
struct GreaterEq {
  bool operator()(const E& a, const E& b) const { return a.key >= b.key; }
};
struct Less {
  bool operator()(const E& a, const E& b) const { return a.key < b.key; }
};

static int g_fail = 0;
static long g_cases = 0;

static void check_same(const std::vector<E>& a, const std::vector<E>& b, const char* what, int n, int k) {
  ++g_cases;
  if (a.size() != b.size() || !std::equal(a.begin(), a.end(), b.begin())) {
    if (g_fail < 10) std::printf("MISMATCH %s n=%d k=%d\n", what, n, k);
    ++g_fail;
  }
}

static void run_all(const std::vector<float>& keys) {
  const int n = (int)keys.size();
  std::vector<E> base(n);
  for (int i = 0; i < n; i++) base[i] = E{keys[i], i};
  // nth_element at several positions
  std::vector<int> ks = {0, 1, n / 7, n / 3, n / 2, n - 2, n - 1, n};
  for (int k : ks) {
    if (k < 0 || k > n) continue;
    std::vector<E> a = base, b = base;
    std::nth_element(a.begin(), a.begin() + k, a.end(), Greater());
    vsf_sel::nth_element_(b.data(), n, k, Greater());
    check_same(a, b, "nth_element", n, k);
    // retainBest(k)
    a = base;
    b = base;
    if (k >= 0 && n > k) {
      if (k == 0) {
        a.clear();
      } else {
        std::nth_element(a.begin(), a.begin() + k, a.end(), Greater());
        const float amb = a[k - 1].key;
        auto ne = std::partition(a.begin() + k, a.end(), [&](const E& e) { return e.key >= amb; });
        a.resize(ne - a.begin());
      }
    }
    const int m = vsf_sel::retain_best_(b.data(), n, k, Greater(), GreaterEq());
    b.resize(m);
    check_same(a, b, "retain_best", n, k);
  }
  {
    std::vector<E> a = base, b = base;
    std::sort(a.begin(), a.end(), Less());
    vsf_sel::sort_(b.data(), n, Less());
    check_same(a, b, "sort", n, 0);
  }
  if (n > 0) {
    std::vector<E> a = base, b = base;
    const float thr = keys[n / 2];
    auto it = std::partition(a.begin(), a.end(), [&](const E& e) { return e.key >= thr; });
    const int cut = vsf_sel::partition_(b.data(), 0, n, [&](const E& e) { return e.key >= thr; });
    if ((int)(it - a.begin()) != cut) {
      ++g_fail;
      std::printf("partition cut mismatch n=%d\n", n);
    }
    check_same(a, b, "partition", n, 0);
  }
}

// McIlroy's adaptive adversary ("A Killer Adversary for Quicksort"): drives the host libstdc++ itself into
// its depth-limit fallback and records the concrete input that does it.
struct Adversary {
  std::vector<int> val;
  int nsolid = 0, candidate = 0, gas;
  explicit Adversary(int n) : val(n, n - 1), gas(n - 1) {}
  bool less(int x, int y) {
    if (val[x] == gas && val[y] == gas) {
      if (x == candidate)
        val[x] = nsolid++;
      else
        val[y] = nsolid++;
    }
    if (val[x] == gas)
      candidate = x;
    else if (val[y] == gas)
      candidate = y;
    return val[x] < val[y];
  }
};

static std::vector<float> killer_for_sort(int n) {
  Adversary adv(n);
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(), [&](int x, int y) { return adv.less(x, y); });
  std::vector<float> k(n);
  for (int i = 0; i < n; i++) k[i] = (float)adv.val[i];
  return k;
}

static std::vector<float> killer_for_nth(int n, int nth) {
  Adversary adv(n);
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  // comparator "greater": x before y if val[x] > val[y]  <=> less(y, x)
  std::nth_element(idx.begin(), idx.begin() + nth, idx.end(), [&](int x, int y) { return adv.less(y, x); });
  std::vector<float> k(n);
  for (int i = 0; i < n; i++) k[i] = (float)adv.val[i];
  return k;
}

int main() {
  std::mt19937 rng(12345);
  const int sizes[] = {0, 1, 2, 3, 4, 5, 7, 15, 16, 17, 18, 31, 33, 64, 100, 181, 257, 1000, 2866, 8525};
  for (int n : sizes) {
    for (int rep = 0; rep < (n < 300 ? 40 : 6); rep++) {
      std::vector<float> k(n);
      // uniform floats
      for (auto& v : k) v = std::uniform_real_distribution<float>(-1.f, 1.f)(rng);
      run_all(k);
      // small-int ties (FAST scores 20..60)
      for (auto& v : k) v = (float)std::uniform_int_distribution<int>(20, 20 + (rep % 5) * 10 + 1)(rng);
      run_all(k);
      // Hamming distances as floats
      for (auto& v : k) v = (float)std::binomial_distribution<int>(256, 0.2)(rng);
      run_all(k);
    }
    std::vector<float> k(n);
    for (int i = 0; i < n; i++) k[i] = (float)i;
    run_all(k);
    for (int i = 0; i < n; i++) k[i] = (float)(n - i);
    run_all(k);
    for (int i = 0; i < n; i++) k[i] = 3.f;
    run_all(k);
    for (int i = 0; i < n; i++) k[i] = (float)std::min(i, n - 1 - i);
    run_all(k);
    for (int i = 0; i < n; i++) k[i] = (float)(i % 2 ? i : -i);
    run_all(k);
  }
  const int before = vsf_sel_trace_fallbacks;
  for (int n : {64, 200, 1000, 4000}) {
    run_all(killer_for_sort(n));
    for (int nth : {1, n / 2, n - 2}) run_all(killer_for_nth(n, nth));
  }
  const int fallbacks = vsf_sel_trace_fallbacks - before;
  std::printf("cases=%ld failures=%d heap_fallbacks_on_killers=%d\n", g_cases, g_fail, fallbacks);
  if (fallbacks == 0) {
    std::printf("adversarial inputs did not reach the depth-limit fallback\n");
    return 2;
  }
  return g_fail ? 1 : 0;
}
