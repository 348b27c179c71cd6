// goheap.hpp -- Go container/heap sift rules over (value, weight) arrays, usable on host and
// device.  The reference's top-k containers (common/heap/filter.go, common/heap/pq.go) are thin
// wrappers over container/heap with a STRICT comparison (pq.go:42-48), so which of two equal
// weights survives / comes first is decided by these sift rules; reproducing them is what makes
// the rank lists index-exact, ties included.
//   up(j):      i=(j-1)/2; if i==j || !less(j,i) break; swap(i,j); j=i
//   down(i0,n): j1=2i+1; if j1>=n break; j=j1; if j2=j1+1<n && less(j2,j1) j=j2;
//               if !less(j,i) break; swap(i,j); i=j
//   Push = append, up(n-1).   Pop = swap(0,n-1), down(0,n-1), take last.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GORSE_HD __host__ __device__
#else
#define GORSE_HD
#endif

namespace gorse {

template <bool DESC>
struct GoHeap {
    int32_t *v;
    float *w;
    int n;
    GORSE_HD GoHeap(int32_t *v_, float *w_) : v(v_), w(w_), n(0) {}
    GORSE_HD bool less(int i, int j) const { return DESC ? w[i] > w[j] : w[i] < w[j]; }
    GORSE_HD void swap(int i, int j) {
        int32_t tv = v[i];
        v[i] = v[j];
        v[j] = tv;
        float tw = w[i];
        w[i] = w[j];
        w[j] = tw;
    }
    GORSE_HD void up(int j) {
        for (;;) {
            int i = (j - 1) / 2;
            if (i == j || !less(j, i)) break;
            swap(i, j);
            j = i;
        }
    }
    GORSE_HD void down(int i0, int m) {
        int i = i0;
        for (;;) {
            int j1 = 2 * i + 1;
            if (j1 >= m || j1 < 0) break;
            int j = j1, j2 = j1 + 1;
            if (j2 < m && less(j2, j1)) j = j2;
            if (!less(j, i)) break;
            swap(i, j);
            i = j;
        }
    }
    GORSE_HD void push(int32_t val, float wt) {
        v[n] = val;
        w[n] = wt;
        n++;
        up(n - 1);
    }
    // removes the root; it is left at position n (just past the new end)
    GORSE_HD void pop() {
        int m = n - 1;
        swap(0, m);
        down(0, m);
        n = m;
    }
};

}  // namespace gorse
