// vsf_select.h -- order-exact selection and sorting, usable on host and in HIP device code.
//
// cv::KeyPointsFilter::retainBest (features2d/keypoint.cpp, reached from ORB's computeKeyPoints and so from
// slam_frontend.cc:274) leaves its survivors in whatever permutation libstdc++'s std::nth_element +
// std::partition produce, and Frontend::GetFeatureMatches (slam_frontend.cc:289-291) keeps the first 30 % of
// an (unstable) std::sort.  "Bit-exact keypoint indices / match pairs" therefore needs those permutations
// (SURVEY.md section 7 H1, Appendix A.10).  A permutation depends only on the sequence of comparison outcomes, so
// this header restates the algorithms (introselect with median-of-3 + unguarded Hoare partition + heap-select
// fallback; bidirectional partition; introsort + final insertion sort) step for step on plain arrays.
// tests/cpp/test_select.cc checks every routine against the host libstdc++ on random, tie-heavy and
// adversarial inputs.
#ifndef VSF_SELECT_H_
#define VSF_SELECT_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define VSF_HD __host__ __device__ __forceinline__
#else
#define VSF_HD inline
#endif

#if defined(VSF_SELECT_TRACE)
extern int vsf_sel_trace_fallbacks;  // test hook: counts depth-limit (heap) fallbacks taken
#define VSF_SEL_TRACE_FALLBACK() (++vsf_sel_trace_fallbacks)
#else
#define VSF_SEL_TRACE_FALLBACK() ((void)0)
#endif

namespace vsf_sel {

template <class T>
VSF_HD void swap_(T& a, T& b) {
  T t = a;
  a = b;
  b = t;
}

VSF_HD int lg_(int n) {  // std::__lg: floor(log2(n)), n > 0
  int r = 0;
  while (n > 1) {
    n >>= 1;
    ++r;
  }
  return r;
}

// ---- heap primitives (bits/stl_heap.h) ----
template <class T, class Comp>
VSF_HD void push_heap_(T* first, int hole, int top, T value, Comp comp) {
  int parent = (hole - 1) / 2;
  while (hole > top && comp(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

template <class T, class Comp>
VSF_HD void adjust_heap_(T* first, int hole, int len, T value, Comp comp) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (comp(first[second], first[second - 1])) second--;
    first[hole] = first[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    first[hole] = first[second - 1];
    hole = second - 1;
  }
  push_heap_(first, hole, top, value, comp);
}

template <class T, class Comp>
VSF_HD void make_heap_(T* first, int len, Comp comp) {
  if (len < 2) return;
  int parent = (len - 2) / 2;
  while (true) {
    T value = first[parent];
    adjust_heap_(first, parent, len, value, comp);
    if (parent == 0) return;
    parent--;
  }
}

// __pop_heap(first, last, result): heap is [first, first+len), result may lie outside it.
template <class T, class Comp>
VSF_HD void pop_heap_(T* first, int len, T* result, Comp comp) {
  T value = *result;
  *result = *first;
  adjust_heap_(first, 0, len, value, comp);
}

template <class T, class Comp>
VSF_HD void heap_select_(T* a, int first, int middle, int last, Comp comp) {
  make_heap_(a + first, middle - first, comp);
  for (int i = middle; i < last; ++i)
    if (comp(a[i], a[first])) pop_heap_(a + first, middle - first, a + i, comp);
}

template <class T, class Comp>
VSF_HD void sort_heap_(T* a, int first, int last, Comp comp) {
  while (last - first > 1) {
    --last;
    pop_heap_(a + first, last - first, a + last, comp);
  }
}

// ---- insertion sort (bits/stl_algo.h) ----
template <class T, class Comp>
VSF_HD void unguarded_linear_insert_(T* a, int last, Comp comp) {
  T val = a[last];
  int next = last - 1;
  while (comp(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

template <class T, class Comp>
VSF_HD void insertion_sort_(T* a, int first, int last, Comp comp) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (comp(a[i], a[first])) {
      T val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];  // std::move_backward(first, i, i + 1)
      a[first] = val;
    } else {
      unguarded_linear_insert_(a, i, comp);
    }
  }
}

template <class T, class Comp>
VSF_HD void unguarded_insertion_sort_(T* a, int first, int last, Comp comp) {
  for (int i = first; i != last; ++i) unguarded_linear_insert_(a, i, comp);
}

// ---- pivot + unguarded partition ----
template <class T, class Comp>
VSF_HD void move_median_to_first_(T* arr, int result, int a, int b, int c, Comp comp) {
  if (comp(arr[a], arr[b])) {
    if (comp(arr[b], arr[c]))
      swap_(arr[result], arr[b]);
    else if (comp(arr[a], arr[c]))
      swap_(arr[result], arr[c]);
    else
      swap_(arr[result], arr[a]);
  } else if (comp(arr[a], arr[c])) {
    swap_(arr[result], arr[a]);
  } else if (comp(arr[b], arr[c])) {
    swap_(arr[result], arr[c]);
  } else {
    swap_(arr[result], arr[b]);
  }
}

template <class T, class Comp>
VSF_HD int unguarded_partition_(T* a, int first, int last, int pivot, Comp comp) {
  while (true) {
    while (comp(a[first], a[pivot])) ++first;
    --last;
    while (comp(a[pivot], a[last])) --last;
    if (!(first < last)) return first;
    swap_(a[first], a[last]);
    ++first;
  }
}

template <class T, class Comp>
VSF_HD int unguarded_partition_pivot_(T* a, int first, int last, Comp comp) {
  const int mid = first + (last - first) / 2;
  move_median_to_first_(a, first, first + 1, mid, last - 1, comp);
  return unguarded_partition_(a, first + 1, last, first, comp);
}

// ---- std::__introselect continued from a given (first, last, depth_limit) state ----
template <class T, class Comp>
VSF_HD void introselect_from_(T* a, int first, int last, int nth, int depth, Comp comp) {
  while (last - first > 3) {
    if (depth == 0) {
      VSF_SEL_TRACE_FALLBACK();
      heap_select_(a, first, nth + 1, last, comp);
      swap_(a[first], a[nth]);
      return;
    }
    --depth;
    const int cut = unguarded_partition_pivot_(a, first, last, comp);
    if (cut <= nth)
      first = cut;
    else
      last = cut;
  }
  insertion_sort_(a, first, last, comp);
}

// ---- std::nth_element(a, a + nth, a + n, comp) ----
template <class T, class Comp>
VSF_HD void nth_element_(T* a, int n, int nth, Comp comp) {
  if (n == 0 || nth == n) return;
  int first = 0, last = n;
  int depth = lg_(n) * 2;
  while (last - first > 3) {
    if (depth == 0) {
      VSF_SEL_TRACE_FALLBACK();
      heap_select_(a, first, nth + 1, last, comp);
      swap_(a[first], a[nth]);
      return;
    }
    --depth;
    const int cut = unguarded_partition_pivot_(a, first, last, comp);
    if (cut <= nth)
      first = cut;
    else
      last = cut;
  }
  insertion_sort_(a, first, last, comp);
}

// ---- std::partition(a + first, a + last, pred) for bidirectional iterators; returns the split point ----
template <class T, class Pred>
VSF_HD int partition_(T* a, int first, int last, Pred pred) {
  while (true) {
    while (true) {
      if (first == last) return first;
      if (pred(a[first]))
        ++first;
      else
        break;
    }
    --last;
    while (true) {
      if (first == last) return first;
      if (!pred(a[last]))
        --last;
      else
        break;
    }
    swap_(a[first], a[last]);
    ++first;
  }
}

// ---- std::sort(a, a + n, comp): introsort (threshold 16) + final insertion sort ----
template <class T, class Comp>
VSF_HD void sort_(T* a, int n, Comp comp) {
  if (n <= 0) return;
  // __introsort_loop with the recursion on [cut, last) unrolled onto an explicit stack.
  int stack_first[64], stack_last[64], stack_depth[64];
  int sp = 0;
  stack_first[sp] = 0;
  stack_last[sp] = n;
  stack_depth[sp] = lg_(n) * 2;
  ++sp;
  while (sp > 0) {
    --sp;
    int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
    while (last - first > 16) {
      if (depth == 0) {
        VSF_SEL_TRACE_FALLBACK();
        // __partial_sort(first, last, last)
        heap_select_(a, first, last, last, comp);
        sort_heap_(a, first, last, comp);
        break;
      }
      --depth;
      const int cut = unguarded_partition_pivot_(a, first, last, comp);
      // The real code recurses into [cut, last) first and then continues with [first, cut); the two ranges
      // are disjoint, so finishing [first, cut) first yields the same permutation.
      stack_first[sp] = cut;
      stack_last[sp] = last;
      stack_depth[sp] = depth;
      ++sp;
      last = cut;
    }
  }
  if (n > 16) {
    insertion_sort_(a, 0, 16, comp);
    unguarded_insertion_sort_(a, 16, n, comp);
  } else {
    insertion_sort_(a, 0, n, comp);
  }
}

// ---- the part of std::sort that concerns a[first, last) once __introsort_loop has reached it with `depth` levels left:
// the remaining partitions of the range and the (block-local) final insertion sort.  Sequential fallback of the parallel
// restatement in k_frontend.hip.
template <class T, class Comp>
VSF_HD void sort_from_(T* a, int first0, int last0, int depth0, Comp comp) {
  int stack_first[64], stack_last[64], stack_depth[64];
  int sp = 0;
  stack_first[sp] = first0;
  stack_last[sp] = last0;
  stack_depth[sp] = depth0;
  ++sp;
  while (sp > 0) {
    --sp;
    int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
    bool sorted = false;
    while (last - first > 16) {
      if (depth == 0) {
        heap_select_(a, first, last, last, comp);
        sort_heap_(a, first, last, comp);
        sorted = true;
        break;
      }
      --depth;
      const int cut = unguarded_partition_pivot_(a, first, last, comp);
      stack_first[sp] = cut;
      stack_last[sp] = last;
      stack_depth[sp] = depth;
      ++sp;
      last = cut;
    }
    if (!sorted) insertion_sort_(a, first, last, comp);  // the final insertion sort never leaves this block
  }
}

// ---- cv::KeyPointsFilter::retainBest(kps, n_points) on an array; returns the new size ----
// greater(a, b): a.response > b.response;  ge(x, y): x.response >= y.response
template <class T, class Greater, class GreaterEq>
VSF_HD int retain_best_(T* a, int n, int n_points, Greater greater, GreaterEq ge) {
  if (n_points >= 0 && n > n_points) {
    if (n_points == 0) return 0;
    nth_element_(a, n, n_points, greater);
    const T ambiguous = a[n_points - 1];
    return partition_(a, n_points, n, [&](const T& x) { return ge(x, ambiguous); });
  }
  return n;
}

}  // namespace vsf_sel
#endif  // VSF_SELECT_H_
