// bpr_bins.hpp -- the geometry of the BPR chunk preparation by user bins (csrc/bpr.hip: bpr_bin_count / _offsets / _scatter / _sort
// kernels), shared by the launch code and by the CPU cover test (host library hook gh_test_bpr_bins_*, tests/test_launch_geometry_cpu.py).
// Pure host C++: no device code, no HIP types.  Reference semantics of what is being grouped: model/cf/model.go:449-468 (the samples
// of an epoch, users drawn uniformly); the order in which a Hogwild epoch applies them is free (common/parallel/parallel.go:44-68).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace gorse {

constexpr int kMaxBins = 8192;    // bins of a chunk at most + 1 (LDS of the count / scatter kernels: one counter each)
constexpr int kMaxBinShift = 11;  // user ids per bin at most 2^11 (LDS of the sort kernel: one counter each)

// bins of the binned preparation for a chunk of n samples: user ids per bin 2^shift, (U >> shift) + 1 bins (key U = no user)
struct PrepBins {
    int shift, nbins;
    int64_t tile;  // samples per workgroup of the count / scatter kernels
    bool ok;       // false: more user ids per bin than the sort kernel's LDS holds (U > 16M) -> the preparation without bins
};
inline PrepBins prep_bins(int64_t U, int64_t n) {
    PrepBins b;
    int64_t target = std::min<int64_t>(std::max<int64_t>(n / 4096, 512), kMaxBins - 1);
    // (a small chunk of a handle with millions of users -- the tail of an epoch -- takes as many bins as the sort kernel's LDS asks for,
    // not the few its samples would: found by tests/test_launch_geometry_cpu.py, which met shift 15 at 10M users and a one-sample chunk)
    const int64_t need = (U >> kMaxBinShift) + 1;
    if (need <= kMaxBins - 1) target = std::max(target, need);
    b.shift = 0;
    while ((U >> b.shift) + 1 > target) b.shift++;
    b.nbins = (int)((U >> b.shift) + 1);
    int64_t t = 4096;
    while (t < 65536 && t * 512 < n) t *= 2;
    b.tile = t;
    b.ok = b.shift <= kMaxBinShift;
    return b;
}
// words of the tile x bin count matrix that serve every chunk of at most `cap` samples: prep_bins gives a chunk at most
// max(512, cap / 65536) tiles and min(kMaxBins - 1, U + 1) bins
inline size_t prep_matrix_words(int64_t U, int64_t cap) {
    return (size_t)std::max<int64_t>(512, (cap + 65535) / 65536) * (size_t)std::min<int64_t>(kMaxBins - 1, U + 1);
}

// CPU restatement of the four passes over given user keys (key[s] in [0, U), or < 0 = no user could be drawn), in the kernels' own
// arithmetic: count matrix per (tile, bin) -> column prefixes + bin totals -> bin starts -> (sample, key) pairs by bin -> run offsets
// bucket[0 .. U + 1] and the pairs in run order.  (The kernels place the samples of a (tile, bin) cell and of a run in order of
// arrival; here: ascending sample id.)  Returns false where prep_bins says the unbinned form must run.
inline bool prep_bins_emulate(int64_t U, const int32_t *key, int64_t n, std::vector<int32_t> &bucket, std::vector<int32_t> &pair_s,
                              std::vector<int32_t> &pair_u) {
    const PrepBins pb = prep_bins(U, n);
    if (!pb.ok) return false;
    const int64_t tiles = (n + pb.tile - 1) / pb.tile;
    auto bin_of = [&](int32_t k) { return (int)((k < 0 ? U : (int64_t)k) >> pb.shift); };
    std::vector<int32_t> H((size_t)tiles * pb.nbins, 0), bin_count((size_t)pb.nbins, 0), bin_start((size_t)pb.nbins + 1, 0);
    for (int64_t s = 0; s < n; s++) H[(size_t)(s / pb.tile) * pb.nbins + bin_of(key[s])]++;          // bpr_bin_count_kernel
    for (int b = 0; b < pb.nbins; b++) {                                                             // bpr_bin_offsets_kernel
        int32_t run = 0;
        for (int64_t t = 0; t < tiles; t++) {
            const int32_t v = H[(size_t)t * pb.nbins + b];
            H[(size_t)t * pb.nbins + b] = run;
            run += v;
        }
        bin_count[(size_t)b] = run;
    }
    for (int b = 0; b < pb.nbins; b++) bin_start[(size_t)b + 1] = bin_start[(size_t)b] + bin_count[(size_t)b];
    std::vector<int32_t> bs((size_t)n), bu((size_t)n), cur(H.size());                                // bpr_bin_scatter_kernel
    for (int64_t t = 0; t < tiles; t++)
        for (int b = 0; b < pb.nbins; b++) cur[(size_t)t * pb.nbins + b] = bin_start[(size_t)b] + H[(size_t)t * pb.nbins + b];
    for (int64_t s = 0; s < n; s++) {
        const int32_t pos = cur[(size_t)(s / pb.tile) * pb.nbins + bin_of(key[s])]++;
        bs[(size_t)pos] = (int32_t)s;
        bu[(size_t)pos] = key[s];
    }
    bucket.assign((size_t)U + 2, 0);                                                                 // bpr_bin_sort_kernel
    pair_s.assign((size_t)n, 0);
    pair_u.assign((size_t)n, 0);
    const int ub = 1 << pb.shift;
    std::vector<int32_t> cnt((size_t)ub);
    for (int b = 0; b < pb.nbins; b++) {
        const int64_t ulo = (int64_t)b << pb.shift;
        const int32_t b0 = bin_start[(size_t)b], b1 = bin_start[(size_t)b + 1];
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int32_t e = b0; e < b1; e++) cnt[(size_t)((bu[(size_t)e] < 0 ? U : (int64_t)bu[(size_t)e]) - ulo)]++;
        int32_t run = 0;
        for (int k = 0; k < ub; k++) {
            const int32_t c = cnt[(size_t)k];
            cnt[(size_t)k] = run;
            if (ulo + k <= U) bucket[(size_t)(ulo + k)] = b0 + run;
            if (ulo + k == U) bucket[(size_t)U + 1] = b1;
            run += c;
        }
        for (int32_t e = b0; e < b1; e++) {
            const int32_t p = b0 + cnt[(size_t)((bu[(size_t)e] < 0 ? U : (int64_t)bu[(size_t)e]) - ulo)]++;
            pair_s[(size_t)p] = bs[(size_t)e];
            pair_u[(size_t)p] = bu[(size_t)e];
        }
    }
    return true;
}

}  // namespace gorse
