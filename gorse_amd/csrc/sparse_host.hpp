// sparse_host.hpp -- host-side part of the index construction of the sparse top-k (no HIP in here): CSR validation, the
// scratch numbering of the rows, the list of distinct indices.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

namespace gorse {
namespace sparse {

// CSR sanity: indptr non-decreasing from a non-negative start, indices strictly ascending inside a row (what
// slices.Sort + a set give the reference's writers, logics/item_to_item.go:187-193).  Returns "" or the complaint.
inline std::string validate_csr(int64_t rows, const int64_t *indptr, const uint32_t *indices) {
    if (rows < 0) return "negative row count";
    if (rows == 0) return "";
    if (!indptr) return "indptr is NULL";
    if (indptr[0] < 0) return "indptr[0] is negative";
    for (int64_t r = 0; r < rows; r++) {
        if (indptr[r + 1] < indptr[r]) return "indptr decreases at row " + std::to_string(r);
        if (indptr[r + 1] > indptr[r] && !indices) return "indices is NULL";
        for (int64_t e = indptr[r] + 1; e < indptr[r + 1]; e++)
            if (indices[e] <= indices[e - 1]) return "indices of row " + std::to_string(r) + " are not strictly ascending";
    }
    return "";
}

// Scratch numbering of the stored rows: longest row first (ties by row).  Rows with many entries are the ones most queries
// reach; numbering them first puts them into the first tiles (sparse_kernels.hpp), whose accumulators are then dense.
// Results, masks and exclusions keep the caller's row ids (orig_of translates back).
// Round 6: the FRONT.  The rows longer than `front_cut` (the ones a pass answers as long / heavy queries) get a row group of their
// own when there are fewer of them than a group holds: `pad` phantom ids follow them (orig_of = -1: no entries, never a query,
// never a result), so that the next row starts group 1.  The symmetric all-pairs pass then lets the front DELIVER its scores and
// every other row leaves group 0 out of its walk (sparse_kernels.hpp, FrontArgs).
struct RowOrder {
    std::vector<int32_t> new_of;   // caller's row -> scratch id
    std::vector<int32_t> orig_of;  // scratch id -> caller's row, or -1 (a phantom id)
    int64_t Np = 0;                // scratch ids: N + pad
    int32_t n_front = 0, pad = 0;  // pad > 0: ids [0, n_front) are the front, [n_front, n_front + pad) phantoms
    int64_t rank_of(int64_t sid) const { return sid < n_front ? sid : sid - pad; }       // position in the longest-first order
    int64_t sid_of_rank(int64_t rank) const { return rank < n_front ? rank : rank + pad; }
};
inline RowOrder order_rows(int64_t N, const int64_t *indptr, int64_t front_cut = 0, int64_t group = 0) {
    RowOrder o;
    std::vector<int32_t> by_len((size_t)N);
    o.new_of.resize((size_t)N);
    std::iota(by_len.begin(), by_len.end(), 0);
    std::stable_sort(by_len.begin(), by_len.end(), [&](int32_t a, int32_t b) {
        return indptr[a + 1] - indptr[a] > indptr[b + 1] - indptr[b];
    });
    int64_t n_front = 0;
    if (front_cut > 0 && group > 0)
        while (n_front < N && indptr[by_len[(size_t)n_front] + 1] - indptr[by_len[(size_t)n_front]] > front_cut) n_front++;
    if (n_front > 0 && n_front < group && n_front < N) o.n_front = (int32_t)n_front, o.pad = (int32_t)(group - n_front);
    o.Np = N + o.pad;
    o.orig_of.assign((size_t)o.Np, -1);
    for (int64_t t = 0; t < N; t++) {
        const int64_t sid = o.sid_of_rank(t);
        o.orig_of[(size_t)sid] = by_len[(size_t)t];
        o.new_of[(size_t)by_len[(size_t)t]] = (int32_t)sid;
    }
    return o;
}

// The sorted distinct indices of the stored entries: the posting-list directory has one row per index that occurs, so
// the raw index space (up to 2^32) costs nothing.  A bitmap where the space is small enough, a sort otherwise.
inline std::vector<uint32_t> distinct_indices(const uint32_t *indices, int64_t nnz) {
    std::vector<uint32_t> out;
    if (nnz <= 0) return out;
    uint32_t top = 0;
    for (int64_t t = 0; t < nnz; t++) top = indices[t] > top ? indices[t] : top;
    if ((uint64_t)top < ((uint64_t)1 << 31)) {  // <= 256 MB of bits
        std::vector<uint64_t> bits(((size_t)top >> 6) + 1, 0);
        for (int64_t t = 0; t < nnz; t++) bits[indices[t] >> 6] |= (uint64_t)1 << (indices[t] & 63);
        size_t n = 0;
        for (uint64_t w : bits) n += (size_t)__builtin_popcountll(w);
        out.reserve(n);
        for (size_t w = 0; w < bits.size(); w++)
            for (uint64_t b = bits[w]; b; b &= b - 1) out.push_back((uint32_t)((w << 6) + (size_t)__builtin_ctzll(b)));
    } else {
        out.assign(indices, indices + nnz);
        std::sort(out.begin(), out.end());
        out.erase(std::unique(out.begin(), out.end()), out.end());
    }
    return out;
}

// smallest non-zero finite |value| (0 when there is none): sparse.hip uses ds_add_f32 only where no product of a stored and a
// query value -- hence no partial sum -- can come near the subnormal range
inline float smallest_magnitude(const float *v, int64_t n) {
    float m = 0.0f;
    for (int64_t t = 0; t < n; t++) {
        const float x = v[t] < 0 ? -v[t] : v[t];
        if (x > 0.0f && x <= 3.4028234e38f && (m == 0.0f || x < m)) m = x;
    }
    return m;
}

// LDS ranking buffer of sparse_tile_kernel: the smallest instantiated KP >= k (0 = k too large)
inline int pick_kp(int k) {
    for (int kp = 128; kp <= 1024; kp <<= 1)  // 128: two more waves per CU than 256 for the k = 100 of the similarity refreshes
        if (k <= kp) return kp;
    return 0;
}

}  // namespace sparse
}  // namespace gorse
