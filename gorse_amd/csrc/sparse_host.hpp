// sparse_host.hpp -- host-side index construction of the sparse top-k (no HIP in here: sparse.hip uses it before the
// upload, tests/emu/sparse_emu.cpp uses the very same code in front of the emulated kernel).
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

namespace gorse {
namespace sparse {

constexpr int64_t kMaxDims = (int64_t)1 << 30;  // posting-list directory: 8 bytes per possible index

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
// reach, so the first H scratch ids are where an LDS-resident slice of the accumulators pays; results, masks and
// exclusions keep the caller's row ids (orig_of translates back).
struct RowOrder {
    std::vector<int32_t> new_of;   // caller's row -> scratch id
    std::vector<int32_t> orig_of;  // scratch id -> caller's row
};
inline RowOrder order_rows(int64_t N, const int64_t *indptr) {
    RowOrder o;
    o.orig_of.resize((size_t)N);
    o.new_of.resize((size_t)N);
    std::iota(o.orig_of.begin(), o.orig_of.end(), 0);
    std::stable_sort(o.orig_of.begin(), o.orig_of.end(), [&](int32_t a, int32_t b) {
        return indptr[a + 1] - indptr[a] > indptr[b + 1] - indptr[b];
    });
    for (int64_t t = 0; t < N; t++) o.new_of[(size_t)o.orig_of[(size_t)t]] = (int32_t)t;
    return o;
}

// Postings (the transposed CSR) by counting sort: list t holds the (scratch ids of the) rows that contain index t, in
// ascending order of the caller's row ids.
// D = largest index + 1 (0 without entries).
struct Postings {
    int64_t D = 0;
    std::vector<int64_t> ptr;   // D + 1
    std::vector<int32_t> row;   // nnz
    std::vector<float> val;     // nnz
};
// `new_of` (may be null = identity): the id stored for row r
inline std::string build_postings(int64_t N, const int64_t *indptr, const uint32_t *indices, const float *values,
                                  Postings &out, const int32_t *new_of = nullptr) {
    const int64_t b = N > 0 ? indptr[0] : 0, e = N > 0 ? indptr[N] : 0;
    int64_t D = 0;
    for (int64_t t = b; t < e; t++) D = indices[t] >= D ? (int64_t)indices[t] + 1 : D;
    if (D > kMaxDims) return "largest index " + std::to_string(D - 1) + " exceeds the supported index space";
    out.D = D;
    out.ptr.assign((size_t)D + 1, 0);
    out.row.resize((size_t)(e - b));
    out.val.resize((size_t)(e - b));
    for (int64_t t = b; t < e; t++) out.ptr[(size_t)indices[t] + 1]++;
    for (int64_t t = 0; t < D; t++) out.ptr[(size_t)t + 1] += out.ptr[(size_t)t];
    std::vector<int64_t> cur(out.ptr.begin(), out.ptr.end() - 1);
    for (int64_t r = 0; r < N; r++)
        for (int64_t t = indptr[r]; t < indptr[r + 1]; t++) {
            const int64_t at = cur[indices[t]]++;
            out.row[(size_t)at] = new_of ? new_of[r] : (int32_t)r;
            out.val[(size_t)at] = values[t];
        }
    return "";
}

// LDS ranking buffer of sparse_query_kernel: the smallest instantiated KP >= k (0 = k too large); 256 at least, because a
// lane ranks four candidates between two votes of its 64-lane workgroup (sparse_kernels.hpp rank_candidates)
inline int pick_kp(int k) {
    for (int kp = 256; kp <= 1024; kp <<= 1)
        if (k <= kp) return kp;
    return 0;
}

}  // namespace sparse
}  // namespace gorse
