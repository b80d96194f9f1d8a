// Restatement of two libstdc++ (GCC 11, the reference's build-host standard library) algorithms
// whose exact behaviour the depth path's results depend on. libstdc++ is a third-party
// dependency of the reference, not vendored in its tree; the algorithms are restated from
// their published form (bits/stl_algo.h, bits/stl_heap.h, bits/random.h, bits/random.tcc).
//
//  * std::nth_element on std::pair<float,float> (Derp.cpp:210). The permutation it leaves in
//    [0, keep) fixes the order in which Derp.cpp:211-214 sums float costs, so it is reproduced
//    step for step: introselect (median-of-3 + unguarded partition, depth limit 2*lg(n),
//    heap-select fallback) finished by insertion sort on the last <= 3 elements.
//  * std::default_random_engine (= minstd_rand0: x <- 16807 x mod 2^31-1) and
//    std::uniform_real_distribution<float> (Derp.cpp:757-758,806-808), with O(log n) jump-ahead
//    so that a pixel can find its position in the row's stream without walking the row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef DERP_HD
#define DERP_HD __host__ __device__ __forceinline__
#endif

namespace derp {

struct alignas(8) SsdPair {  // 8-byte aligned: one ds_read_b64 / ds_write_b64 per LDS slot access
  float first, second;
};

// std::pair operator<: a.first < b.first || (!(b.first < a.first) && a.second < b.second).
// SSD pairs are sums of squares: non-negative, never NaN, never -0 — for such floats the IEEE order
// is the order of the bit patterns, so the lexicographic pair order is one unsigned 64-bit compare.
DERP_HD unsigned long long pair_key(const SsdPair& a) {
  unsigned int hi, lo;
  __builtin_memcpy(&hi, &a.first, 4);
  __builtin_memcpy(&lo, &a.second, 4);
  return ((unsigned long long)hi << 32) | lo;
}
DERP_HD bool pair_less(const SsdPair& a, const SsdPair& b) {
  return pair_key(a) < pair_key(b);
}

// A = accessor with get(i) / set(i, v); indices are absolute positions in the array.
template <typename A>
struct GccSelect {
  A& a;
  DERP_HD explicit GccSelect(A& acc) : a(acc) {}

  DERP_HD bool lt(int i, int j) {
    return pair_less(a.get(i), a.get(j));
  }
  DERP_HD void swap(int i, int j) {
    const SsdPair t = a.get(i);
    a.set(i, a.get(j));
    a.set(j, t);
  }

  // __move_median_to_first(result, a, b, c)
  // returns the key of the element it moved to `result`
  DERP_HD unsigned long long move_median_to_first(int result, int ia, int ib, int ic) {
    // three reads, then the same decision tree on register values
    const SsdPair pa = a.get(ia), pb = a.get(ib), pc = a.get(ic);
    const unsigned long long ka = pair_key(pa), kb = pair_key(pb), kc = pair_key(pc);
    int m;
    if (ka < kb) {
      m = (kb < kc) ? ib : (ka < kc) ? ic : ia;
    } else {
      m = (ka < kc) ? ia : (kb < kc) ? ic : ib;
    }
    const SsdPair pm = (m == ia) ? pa : (m == ib) ? pb : pc;
    const SsdPair pr = a.get(result);
    a.set(result, pm);
    a.set(m, pr);
    return pair_key(pm);
  }

  // __unguarded_partition(first, last, pivot)
  // `kp` = key of the pivot, which sits below `first` and is never swapped here
  DERP_HD int unguarded_partition(int first, int last, unsigned long long kp) {
    for (;;) {
      SsdPair pf = a.get(first);
      while (pair_key(pf) < kp) {
        ++first;
        pf = a.get(first);
      }
      --last;
      SsdPair pl = a.get(last);
      while (kp < pair_key(pl)) {
        --last;
        pl = a.get(last);
      }
      if (!(first < last)) {
        return first;
      }
      a.set(first, pl);  // iter_swap(first, last) with both values already in registers
      a.set(last, pf);
      ++first;
    }
  }

  DERP_HD int unguarded_partition_pivot(int first, int last) {
    const int mid = first + (last - first) / 2;
    const unsigned long long kp = move_median_to_first(first, first + 1, mid, last - 1);
    return unguarded_partition(first + 1, last, kp);
  }

  // __insertion_sort(first, last)
  DERP_HD void insertion_sort(int first, int last) {
    if (first == last) {
      return;
    }
    for (int i = first + 1; i != last; ++i) {
      const SsdPair val = a.get(i);
      if (pair_less(val, a.get(first))) {
        for (int k = i; k > first; --k) {  // move_backward(first, i, i + 1)
          a.set(k, a.get(k - 1));
        }
        a.set(first, val);
      } else {  // __unguarded_linear_insert
        int hole = i;
        int next = i - 1;
        while (pair_less(val, a.get(next))) {
          a.set(hole, a.get(next));
          hole = next;
          --next;
        }
        a.set(hole, val);
      }
    }
  }

  // ---- heap helpers (bits/stl_heap.h), heap occupies [first, first + len) ----
  DERP_HD void push_heap(int first, int hole, int top, const SsdPair& value) {
    int parent = (hole - 1) / 2;
    while (hole > top && pair_less(a.get(first + parent), value)) {
      a.set(first + hole, a.get(first + parent));
      hole = parent;
      parent = (hole - 1) / 2;
    }
    a.set(first + hole, value);
  }
  DERP_HD void adjust_heap(int first, int hole, int len, const SsdPair& value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (lt(first + child, first + (child - 1))) {
        child--;
      }
      a.set(first + hole, a.get(first + child));
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      a.set(first + hole, a.get(first + (child - 1)));
      hole = child - 1;
    }
    push_heap(first, hole, top, value);
  }
  DERP_HD void make_heap(int first, int last) {
    const int len = last - first;
    if (len < 2) {
      return;
    }
    int parent = (len - 2) / 2;
    for (;;) {
      const SsdPair value = a.get(first + parent);
      adjust_heap(first, parent, len, value);
      if (parent == 0) {
        return;
      }
      parent--;
    }
  }
  // __heap_select(first, middle, last)
  DERP_HD void heap_select(int first, int middle, int last) {
    make_heap(first, middle);
    for (int i = middle; i < last; ++i) {
      if (lt(i, first)) {
        // __pop_heap(first, middle, i)
        const SsdPair value = a.get(i);
        a.set(i, a.get(first));
        adjust_heap(first, 0, middle - first, value);
      }
    }
  }

  // std::nth_element(first = 0, nth, last = n)
  DERP_HD void nth_element(int nth, int n) {
    int first = 0, last = n;
    if (first == last || nth == last) {
      return;
    }
    int depth_limit = 2 * (31 - __builtin_clz((unsigned)n));  // std::__lg(n) * 2
    while (last - first > 3) {
      if (depth_limit == 0) {
        heap_select(first, nth + 1, last);
        swap(first, nth);
        return;
      }
      --depth_limit;
      const int cut = unguarded_partition_pivot(first, last);
      if (cut <= nth) {
        first = cut;
      } else {
        last = cut;
      }
    }
    insertion_sort3(first, last);
  }

  // __insertion_sort over the at most three elements introselect leaves: elements move on strict
  // "less" only, i.e. the result is the stable order of the keys — computed on register copies
  DERP_HD void insertion_sort3(int first, int last) {
    const int cnt = last - first;
    if (cnt < 2) {
      return;
    }
    SsdPair p0 = a.get(first), p1 = a.get(first + 1);
    if (pair_key(p1) < pair_key(p0)) {
      const SsdPair t = p0;
      p0 = p1;
      p1 = t;
    }
    if (cnt == 3) {
      const SsdPair p2 = a.get(first + 2);
      const unsigned long long k2 = pair_key(p2);
      if (k2 < pair_key(p0)) {
        a.set(first + 2, p1);
        p1 = p0;
        p0 = p2;
      } else if (k2 < pair_key(p1)) {
        a.set(first + 2, p1);
        p1 = p2;
      }
    }
    a.set(first, p0);
    a.set(first + 1, p1);
  }
};

// ---- minstd_rand0 ------------------------------------------------------------------------
static constexpr uint32_t kMinstdA = 16807u;
static constexpr uint32_t kMinstdM = 2147483647u;

// x * y mod (2^31 - 1) for x, y < 2^31 - 1. 2^31 = 1 (mod m), so the 62-bit product folds as
// (p & m) + (p >> 31) twice, then one conditional subtraction — no 64-bit division.
DERP_HD uint32_t minstd_mulmod(uint32_t x, uint32_t y) {
  const uint64_t p = (uint64_t)x * (uint64_t)y;
  uint64_t t = (p & kMinstdM) + (p >> 31);  // < 2^32
  t = (t & kMinstdM) + (t >> 31);           // <= m + 1
  return (uint32_t)(t >= kMinstdM ? t - kMinstdM : t);
}
// linear_congruential_engine::seed(s): s mod m, 0 -> 1 (random.tcc:114-124)
DERP_HD uint32_t minstd_seed(int s) {
  const uint32_t v = (uint32_t)((uint64_t)(uint32_t)s % kMinstdM);
  return v == 0 ? 1u : v;
}
// state after `n` draws from `state`
DERP_HD uint32_t minstd_jump(uint32_t state, uint64_t n) {
  uint32_t mult = 1u, base = kMinstdA;
  while (n) {
    if (n & 1) {
      mult = minstd_mulmod(mult, base);
    }
    base = minstd_mulmod(base, base);
    n >>= 1;
  }
  return minstd_mulmod(state, mult);
}
// one uniform_real_distribution<float>(a, b)(engine) draw; advances `state`.
// generate_canonical<float,24>: one engine call; (float)(x - 1) / 2147483648.0f
// (the range 2147483646 rounds to 2^31 in float), results >= 1 -> nextafter(1, 0).
DERP_HD float minstd_uniform(uint32_t& state, float a, float b) {
  state = minstd_mulmod(state, kMinstdA);
  float u = (float)(state - 1u) / 2147483648.0f;
  if (u >= 1.0f) {
    u = 0.99999994f;  // nextafterf(1.f, 0.f)
  }
  return u * (b - a) + a;
}

}  // namespace derp
