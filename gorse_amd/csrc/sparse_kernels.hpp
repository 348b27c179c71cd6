// sparse_kernels.hpp -- device code of the exact sparse top-k (sparse.hip): the sparse collections of
// vectors.Database (storage/vectors/database.go:90-97, xvec.go:241-247: Dot over Indices / Values, exact Flat index)
// that the IDF item-to-item / user-to-user writers fill (logics/vector_writer.go:192-209).
//
// One workgroup answers one query at a time.  The stored rows are held as POSTINGS (one list of (row, value) per
// index); the query's indices are walked in ascending order and, for each, the workgroup's lanes stream that posting
// list (coalesced 4-byte rows + 4-byte values = the 8 algorithmic bytes per multiply-add) and update a per-workgroup
// accumulator (the high word of `cell[row]`).  A row occurs at most once per posting list, so inside one list no two lanes touch the same
// accumulator, and the barrier between lists makes every row's sum run in ascending index order: the float32 result
// is the merge-order sparse dot of the oracle bit for bit, with no atomics on data.  Rows reached for the first time
// (the low word of cell[row] != serial of this query) are appended to a `touched` list; only those are ranked, so nothing of size N
// is cleared or scanned per query.
//
// Ranking: 64-bit keys (order-preserving score bits, ~row) are distinct, so "the k largest keys, descending" is one
// well-defined answer whatever order the lanes append in.  Keys above the running threshold go to an LDS buffer of
// 2*KP entries; when it overflows it is bitonic-sorted, cut to KP entries and the threshold becomes the k-th key.
// The reference ranks ALL admissible documents (one sharing no index scores 0), cuts to topK and THEN drops Score == 0
// (xvec.go:419-421), so zero-score documents use up slots: only the non-zero rows are ranked here, and the number of
// results follows from the counts of positive / negative rows and the number of admissible rows (see `written`).
//
// Only constructs that tests/emu/hip_emu.hpp can also run on the CPU are used here (threadIdx/blockIdx, static
// __shared__, __syncthreads, __syncthreads_or, integer atomicAdd): the kernel's control flow is exercised without a GPU
// by tests/test_sparse_kernel_emu_cpu.py.  That emulation is test infrastructure; the product runs this file on gfx950.
#pragma once
#include <cstdint>

namespace gorse {
namespace sparse {

constexpr int kBlock = 64;  // one wavefront per workgroup: the per-list barrier costs a wave-local s_barrier

// One scratch cell per (workgroup, stored row): low word = serial of the last query that reached the row, high word = the
// bits of its running inner product.  Kept as ONE 64-bit integer so that a posting costs one 8-byte load and one 8-byte
// store (as a two-field struct the compiler loads the stamp, branches, and loads the sum in a second round trip).
using Cell = unsigned long long;
__device__ inline uint32_t cell_stamp(Cell c) { return (uint32_t)c; }
__device__ inline float cell_acc(Cell c) { return __uint_as_float((uint32_t)(c >> 32)); }
__device__ inline Cell make_cell(uint32_t stamp, float acc) { return ((Cell)__float_as_uint(acc) << 32) | (Cell)stamp; }

struct QueryArgs {
    // postings of the N stored rows: list of index t = p_row / p_val [p_ptr[t], p_ptr[t+1]), D lists
    const int64_t *p_ptr;
    const int32_t *p_row;
    const float *p_val;
    int64_t D;
    // queries: CSR rows q_first .. q_first + nq of (q_ptr, q_idx, q_val); indices strictly ascending per row
    const int64_t *q_ptr;
    const uint32_t *q_idx;
    const float *q_val;
    int64_t q_first, nq;
    const int64_t *exclude;  // per query: a stored row left out of its result (-1 = none); may be null
    int exclude_self;        // all pairs: query t is stored row q_first + t, left out of its own result
    const uint8_t *mask;     // admissible[row] or null
    int64_t n_admissible;    // number of admissible rows (N without a mask)
    int64_t N;
    // scratch, N entries per workgroup each: (stamp, accumulator) pairs -- one 8-byte access per posting -- and the list
    // of rows the current query has reached
    Cell *cell;
    int32_t *touched;
    uint32_t serial_base;  // stamps of this launch are serial_base + 1 ..; never reused for a scratch slot
    int k;
    int32_t *out_idx;   // nq x k, padded with -1
    float *out_score;   // nq x k, padded with -inf
    int32_t *out_cnt;   // nq
    unsigned long long *stat;  // [0] += postings walked, [1] += rows hit
};

constexpr uint32_t kZeroOrd = 0x80000000u;  // ordered bits of +0
// order-preserving bits of a score: larger float <=> larger unsigned; -0 counts as +0
__device__ inline uint32_t score_ord(float score) {
    uint32_t u = __float_as_uint(score);
    if ((u << 1) == 0) u = 0;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline unsigned long long make_key(uint32_t ord, int32_t row) {
    return ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)row);
}
// results the reference returns (xvec.go:379-446): it ranks every admissible document, cuts to k, drops Score == 0.
// pos / neg = admissible rows scoring above / below zero, adm = admissible rows; the rest score zero.
__device__ inline int written(long long pos, long long neg, long long adm, int k) {
    if (pos >= k) return k;
    const long long zeros = adm - pos - neg;
    long long n = pos;
    if (pos + zeros < k) n += neg < k - pos - zeros ? neg : k - pos - zeros;
    return (int)n;
}
__device__ inline float key_score(unsigned long long key) {
    uint32_t u = (uint32_t)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}
__device__ inline int32_t key_row(unsigned long long key) { return (int32_t)(0xFFFFFFFFu - (uint32_t)key); }

// bitonic sort of b[0..CAP) into descending order by all threads of the workgroup; ends with a barrier
template <int CAP>
__device__ inline void sort_desc(unsigned long long *b, int tid, int nt) {
    for (int size = 2; size <= CAP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < CAP; i += nt) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const unsigned long long x = b[i], y = b[j];
                    if (desc ? x < y : x > y) {
                        b[i] = y;
                        b[j] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <int KP>
__global__ __launch_bounds__(kBlock) void sparse_query_kernel(QueryArgs a) {
    constexpr int CAP = 2 * KP;
    __shared__ unsigned long long s_buf[CAP];
    __shared__ unsigned long long s_thr;  // keys <= s_thr cannot be among the k best
    __shared__ int s_cnt;                 // rows touched by the current query
    __shared__ int s_bcnt;                // slots handed out in s_buf
    __shared__ int s_pos, s_neg;          // admissible rows of the current query scoring above / below zero
    const int tid = threadIdx.x, nt = blockDim.x;
    Cell *cell = a.cell + (int64_t)blockIdx.x * a.N;
    int32_t *touched = a.touched + (int64_t)blockIdx.x * a.N;
    uint32_t serial = a.serial_base;
    for (int64_t t = blockIdx.x; t < a.nq; t += gridDim.x) {
        serial++;
        if (tid == 0) {
            s_cnt = 0;
            s_bcnt = 0;
            s_thr = 0;
            s_pos = 0;
            s_neg = 0;
        }
        __syncthreads();
        // ---- accumulate: one posting list per query index, ascending ----
        const int64_t qr = a.q_first + t;
        const int64_t qs = a.q_ptr[qr], qe = a.q_ptr[qr + 1];
        unsigned long long walked = 0;
        for (int64_t e = qs; e < qe; e++) {  // every condition below is uniform over the workgroup
            const uint32_t dim = a.q_idx[e];
            if ((int64_t)dim >= a.D) continue;
            const int64_t ps = a.p_ptr[dim], pe = a.p_ptr[dim + 1];
            if (ps == pe) continue;
            const float qv = a.q_val[e];
            walked += (unsigned long long)(pe - ps);
            for (int64_t p = ps + tid; p < pe; p += nt) {
                const int32_t row = a.p_row[p];
                const float term = __fmul_rn(qv, a.p_val[p]);
                const Cell c = cell[row];
                const bool first = cell_stamp(c) != serial;
                cell[row] = make_cell(serial, __fadd_rn(first ? 0.0f : cell_acc(c), term));
                if (first) touched[atomicAdd(&s_cnt, 1)] = row;
            }
            __syncthreads();
        }
        // ---- rank the touched rows ----
        const int T = s_cnt;
        const int64_t ex = a.exclude ? a.exclude[t] : (a.exclude_self ? qr : (int64_t)-1);
        int my_pos = 0, my_neg = 0;
        for (int base = 0; base < T; base += nt) {
            unsigned long long key = 0;  // 0 is below every real key
            const int i = base + tid;
            if (i < T) {
                const int32_t row = touched[i];
                if ((int64_t)row != ex && (!a.mask || a.mask[row])) {
                    const uint32_t ord = score_ord(cell_acc(cell[row]));
                    if (ord != kZeroOrd) {  // a zero score is dropped by the reference's wrapper
                        key = make_key(ord, row);
                        my_pos += ord > kZeroOrd;
                        my_neg += ord < kZeroOrd;
                    }
                }
            }
            bool want = key > s_thr;
            while (true) {
                bool over = false;
                if (want) {
                    const int slot = atomicAdd(&s_bcnt, 1);
                    if (slot < CAP) {
                        s_buf[slot] = key;
                        want = false;
                    } else {
                        over = true;
                    }
                }
                if (!__syncthreads_or(over ? 1 : 0)) break;  // everybody found a slot
                // some lane drew a slot >= CAP, so slots 0..CAP-1 are all written: keep the KP best
                sort_desc<CAP>(s_buf, tid, nt);
                if (tid == 0) {
                    s_bcnt = KP;
                    s_thr = s_buf[a.k - 1];
                }
                __syncthreads();
                want = want && key > s_thr;
            }
        }
        if (my_pos) atomicAdd(&s_pos, my_pos);
        if (my_neg) atomicAdd(&s_neg, my_neg);
        __syncthreads();
        const int n = s_bcnt;  // <= CAP: an overflow is always followed by the cut to KP
        if (n > 0) {
            for (int i = n + tid; i < CAP; i += nt) s_buf[i] = 0;
            __syncthreads();
            sort_desc<CAP>(s_buf, tid, nt);  // positive scores first, then the negative ones
        }
        const bool ex_counts = ex >= 0 && ex < a.N && (!a.mask || a.mask[ex]);
        const int cnt = written(s_pos, s_neg, a.n_admissible - (ex_counts ? 1 : 0), a.k);
        for (int i = tid; i < a.k; i += nt) {
            const unsigned long long key = i < cnt ? s_buf[i] : 0;
            a.out_idx[t * a.k + i] = i < cnt ? key_row(key) : -1;
            a.out_score[t * a.k + i] = i < cnt ? key_score(key) : __uint_as_float(0xff800000u);
        }
        if (tid == 0) {
            a.out_cnt[t] = cnt;
            if (a.stat) {
                atomicAdd(&a.stat[0], walked);
                atomicAdd(&a.stat[1], (unsigned long long)T);
            }
        }
        __syncthreads();  // s_buf / s_cnt are reused by the next query
    }
}

// ---- postings on the device (counting sort of the CSR entries by index) -------------------------------------------------
// Not the library's default yet (sparse.hip builds the postings on the host unless gorse_hip_test_set_sparse_build(1)):
// written without a GPU like the rest of this file, exercised through the emulation.  The order of the entries INSIDE a
// posting list depends on the atomics' order; no result depends on it (a row occurs once per list, and a row's sum runs
// over the lists in the query's index order).
constexpr int kScanBlock = 1024;

struct BuildArgs {
    const int64_t *r_ptr;  // N + 1
    const uint32_t *r_idx;
    const float *r_val;
    int64_t N, nnz, D;
    unsigned long long *p_ptr;   // D + 1, zeroed before the count kernel; = the posting directory after the scan
    unsigned long long *cursor;  // D: next free slot of every list during the scatter
    int32_t *p_row;
    float *p_val;
};

// p_ptr[t + 1] = number of entries with index t
__global__ void sparse_count_kernel(BuildArgs a) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.nnz; e += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&a.p_ptr[(int64_t)a.r_idx[e] + 1], 1ull);
}

// in-place inclusive scan of p_ptr[1 .. D] by ONE workgroup (the directory has at most 2^30 entries and is scanned once per
// index build), then cursor[t] = p_ptr[t]: every thread sums a contiguous chunk, thread 0 scans the chunk sums, every
// thread rewrites its chunk
__global__ __launch_bounds__(kScanBlock) void sparse_scan_kernel(BuildArgs a) {
    __shared__ unsigned long long s_part[kScanBlock];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t chunk = (a.D + nt - 1) / nt;
    const int64_t lo = 1 + (int64_t)tid * chunk, hi = lo + chunk < a.D + 1 ? lo + chunk : a.D + 1;
    unsigned long long sum = 0;
    for (int64_t t = lo; t < hi; t++) sum += a.p_ptr[t];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < nt; t++) {
            const unsigned long long x = s_part[t];
            s_part[t] = run;
            run += x;
        }
    }
    __syncthreads();
    unsigned long long run = s_part[tid];
    for (int64_t t = lo; t < hi; t++) {
        run += a.p_ptr[t];
        a.p_ptr[t] = run;
    }
    __syncthreads();  // the directory is complete (one workgroup: a barrier orders its global writes for its own reads)
    for (int64_t t = tid; t < a.D; t += nt) a.cursor[t] = a.p_ptr[t];
}

// entry e of the CSR goes to the next free slot of its index's list; its row = the r_ptr interval that holds e
__global__ void sparse_scatter_kernel(BuildArgs a) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.nnz; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = a.N;  // largest row with r_ptr[row] <= e
        while (hi - lo > 1) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (a.r_ptr[mid] <= e)
                lo = mid;
            else
                hi = mid;
        }
        const unsigned long long at = atomicAdd(&a.cursor[a.r_idx[e]], 1ull);
        a.p_row[at] = (int32_t)lo;
        a.p_val[at] = a.r_val[e];
    }
}

}  // namespace sparse
}  // namespace gorse
