// sparse_kernels.hpp -- device code of the exact sparse top-k (sparse.hip): the sparse collections of
// vectors.Database (storage/vectors/database.go:90-97, xvec.go:241-247: Dot over Indices / Values, exact Flat index)
// that the IDF item-to-item / user-to-user writers fill (logics/vector_writer.go:192-209).
//
// sparse_query_kernel: one workgroup (one wave) answers one query at a time.  The stored rows are held as POSTINGS (one
// list of (row, value) per index); the query's indices are walked in ascending order and, for each, the lanes stream that
// posting list (coalesced 4-byte rows + 4-byte values = the 8 algorithmic bytes per multiply-add) and update a
// per-workgroup accumulator (the high word of `cell[row]`).  A row occurs at most once per posting list, so inside one
// list no two lanes touch the same accumulator, and the barrier between lists makes every row's sum run in ascending
// index order: the float32 result is the merge-order sparse dot of the oracle bit for bit, with no atomics on data.
// Rows reached for the first time (the low word of cell[row] != serial of this query) are appended to a `touched`
// list; only those are ranked, so nothing of size N is cleared or scanned per query.  Queries with very many entries go
// to the row-streaming kernels further down instead (sparse_heavy_*).
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
constexpr int kRankUnroll = 4;   // candidates a lane of sparse_query_kernel ranks between two votes (KP >= 4 * kBlock)
constexpr int kHeavyRankBlock = 1024;  // sparse_heavy_rank_kernel: sixteen waves stream the N rows of one heavy query

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
    int64_t heavy_dims;      // queries with more entries than this are left to the row-streaming kernels below
    const uint8_t *mask;     // admissible[row] or null
    int64_t n_admissible;    // number of admissible rows (N without a mask)
    int64_t N;
    // posting lists and scratch use SCRATCH ids (longest stored row first, sparse_host.hpp order_rows); orig_of translates
    // back to the caller's row ids, which masks, exclusions and results use
    const int32_t *orig_of;
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

// The ranking shared by both query paths: candidates i = 0 .. count-1, key_at(i) = its 64-bit key or 0 for "not a candidate"
// (called once per i by the lane that owns it).  Every lane takes U candidates between two votes of the workgroup (a vote
// is a barrier, and a query that reaches most of 200,000 rows would otherwise pay 3000 of them on its one wave).  Leaves
// the min(buffered, CAP) best keys sorted descending in s_buf and returns how many are buffered (> 0 only).  s_thr / s_bcnt
// must be 0 on entry; needs U * blockDim.x <= KP (after a cut to KP the CAP - KP free slots take every candidate still
// waiting).  All lanes of the workgroup call it together.
template <int KP, int U, typename KeyAt>
__device__ inline int rank_candidates(int64_t count, KeyAt key_at, int k, unsigned long long *s_buf, unsigned long long *s_thr,
                                      int *s_bcnt, int tid, int nt) {
    constexpr int CAP = 2 * KP;
    for (int64_t base = 0; base < count; base += (int64_t)nt * U) {
        unsigned long long key[U];
        bool want[U];
        const unsigned long long thr = *s_thr;
        for (int u = 0; u < U; u++) {
            const int64_t i = base + (int64_t)u * nt + tid;
            key[u] = i < count ? key_at(i) : 0;  // 0 is below every real key
            want[u] = key[u] > thr;
        }
        while (true) {
            bool over = false;
            for (int u = 0; u < U; u++)
                if (want[u]) {
                    const int slot = atomicAdd(s_bcnt, 1);
                    if (slot < CAP) {
                        s_buf[slot] = key[u];
                        want[u] = false;
                    } else {
                        over = true;
                    }
                }
            if (!__syncthreads_or(over ? 1 : 0)) break;  // every candidate found a slot
            // some candidate drew a slot >= CAP, so slots 0..CAP-1 are all written: keep the KP best
            sort_desc<CAP>(s_buf, tid, nt);
            if (tid == 0) {
                *s_bcnt = KP;
                *s_thr = s_buf[k - 1];
            }
            __syncthreads();
            for (int u = 0; u < U; u++) want[u] = want[u] && key[u] > *s_thr;
        }
    }
    __syncthreads();
    const int n = *s_bcnt;  // <= CAP: an overflow is always followed by the cut to KP
    if (n > 0) {
        for (int i = n + tid; i < CAP; i += nt) s_buf[i] = 0;
        __syncthreads();
        sort_desc<CAP>(s_buf, tid, nt);  // positive scores first, then the negative ones
    }
    return n;
}

// result row t from the sorted buffer: cnt entries, the rest padded
__device__ inline void write_result(const unsigned long long *s_buf, int cnt, int k, int64_t t, int32_t *out_idx,
                                    float *out_score, int32_t *out_cnt, int tid, int nt) {
    for (int i = tid; i < k; i += nt) {
        const unsigned long long key = i < cnt ? s_buf[i] : 0;
        out_idx[t * k + i] = i < cnt ? key_row(key) : -1;
        out_score[t * k + i] = i < cnt ? key_score(key) : __uint_as_float(0xff800000u);
    }
    if (tid == 0) out_cnt[t] = cnt;
}

// HOT > 0 (probe, off by default): the cells of scratch ids < HOT -- the longest stored rows, which under a popularity
// law take most of the hits -- live in LDS instead of the workgroup's global scratch row.
template <int KP, int HOT>
__global__ __launch_bounds__(kBlock) void sparse_query_kernel(QueryArgs a) {
    constexpr int CAP = 2 * KP;
    __shared__ unsigned long long s_buf[CAP];
    __shared__ Cell s_hot[HOT > 0 ? HOT : 1];
    __shared__ unsigned long long s_thr;  // keys <= s_thr cannot be among the k best
    __shared__ int s_cnt;                 // rows touched by the current query
    __shared__ int s_bcnt;                // slots handed out in s_buf
    __shared__ int s_pos, s_neg;          // admissible rows of the current query scoring above / below zero
    const int tid = threadIdx.x, nt = blockDim.x;
    Cell *cell = a.cell + (int64_t)blockIdx.x * a.N;
    int32_t *touched = a.touched + (int64_t)blockIdx.x * a.N;
    uint32_t serial = a.serial_base;
    if (HOT > 0) {  // LDS does not survive a launch: stamp 0 = "reached by no query" (serials start at 1)
        for (int i = tid; i < HOT; i += nt) s_hot[i] = 0;
        __syncthreads();
    }
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
        if (qe - qs > a.heavy_dims) continue;  // uniform; answered by sparse_heavy_score_kernel / sparse_heavy_rank_kernel
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
                Cell *at = (HOT > 0 && row < HOT) ? &s_hot[row] : &cell[row];
                const Cell c = *at;
                const bool first = cell_stamp(c) != serial;
                *at = make_cell(serial, __fadd_rn(first ? 0.0f : cell_acc(c), term));
                if (first) touched[atomicAdd(&s_cnt, 1)] = row;
            }
            __syncthreads();
        }
        // ---- rank the touched rows ----
        const int T = s_cnt;
        const int64_t ex = a.exclude ? a.exclude[t] : (a.exclude_self ? qr : (int64_t)-1);
        int my_pos = 0, my_neg = 0;
        rank_candidates<KP, kRankUnroll>(
            T,
            [&](int64_t i) -> unsigned long long {
                const int32_t sid = touched[i];
                const int32_t row = a.orig_of[sid];
                if ((int64_t)row == ex || (a.mask && !a.mask[row])) return 0;
                const uint32_t ord = score_ord(cell_acc((HOT > 0 && sid < HOT) ? s_hot[sid] : cell[sid]));
                if (ord == kZeroOrd) return 0;  // a zero score is dropped by the reference's wrapper
                my_pos += ord > kZeroOrd;
                my_neg += ord < kZeroOrd;
                return make_key(ord, row);
            },
            a.k, s_buf, &s_thr, &s_bcnt, tid, nt);
        if (my_pos) atomicAdd(&s_pos, my_pos);
        if (my_neg) atomicAdd(&s_neg, my_neg);
        __syncthreads();
        const bool ex_counts = ex >= 0 && ex < a.N && (!a.mask || a.mask[ex]);
        const int cnt = written(s_pos, s_neg, a.n_admissible - (ex_counts ? 1 : 0), a.k);
        write_result(s_buf, cnt, a.k, t, a.out_idx, a.out_score, a.out_cnt, tid, nt);
        if (tid == 0) {
            if (a.stat) {
                atomicAdd(&a.stat[0], walked);
                atomicAdd(&a.stat[1], (unsigned long long)T);
            }
        }
        __syncthreads();  // s_buf / s_cnt are reused by the next query
    }
}

// ---- heavy queries: row streaming instead of posting lists -----------------------------------------------------------------
// A query with very many entries (a popular item's user set under a Zipf law: tens of thousands) would walk that many
// posting lists one after the other in sparse_query_kernel -- milliseconds on ONE wave while the rest of the launch has long
// finished.  Such a query reaches most stored rows anyway, so it is answered the other way round: every stored row r is
// merged against the query by ONE lane (the row's entries in ascending order, each looked up in the query's sorted index
// list by binary search), which needs no accumulators, no barriers and no order bookkeeping -- the sum runs over the
// common indices in ascending order by construction, the oracle's merge order.  HBM-bound on the stored CSR (read once per
// batch of kHeavyBatch queries); the query's own arrays stay in L2.
constexpr int kHeavyBatch = 8;

struct HeavyArgs {
    // the stored rows (CSR, indices ascending) and the queries (CSR rows q_first + t)
    const int64_t *r_ptr;
    const uint32_t *r_idx;
    const float *r_val;
    int64_t N;
    const int64_t *q_ptr;
    const uint32_t *q_idx;
    const float *q_val;
    int64_t q_first;
    int64_t hq[kHeavyBatch];  // the heavy queries of this batch (indices t into the call's queries)
    int nb;
    float *score;     // nb x N: the inner product of row r with heavy query b
    uint8_t *common;  // nb x N: 1 when they share an index
    // ranking (sparse_heavy_rank_kernel)
    const int64_t *exclude;
    int exclude_self;
    const uint8_t *mask;
    int64_t n_admissible;
    int k;
    int32_t *out_idx;
    float *out_score;
    int32_t *out_cnt;
    unsigned long long *stat;  // [0] += row entries looked up, [1] += rows sharing an index
};

__global__ void sparse_heavy_score_kernel(HeavyArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.N; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rs = a.r_ptr[r], re = a.r_ptr[r + 1];
        for (int b = 0; b < a.nb; b++) {
            const int64_t qr = a.q_first + a.hq[b];
            const int64_t qs = a.q_ptr[qr], qe = a.q_ptr[qr + 1];
            float sum = 0.0f;
            bool any = false;
            int64_t from = qs;  // both lists ascend: the search for the next entry starts behind the last match
            for (int64_t e = rs; e < re && from < qe; e++) {
                const uint32_t idx = a.r_idx[e];
                int64_t lo = from, hi = qe;  // first query entry with index >= idx
                while (lo < hi) {
                    const int64_t mid = lo + (hi - lo) / 2;
                    if (a.q_idx[mid] < idx)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                from = lo;
                if (lo < qe && a.q_idx[lo] == idx) {
                    sum = __fadd_rn(sum, __fmul_rn(a.q_val[lo], a.r_val[e]));
                    any = true;
                    from = lo + 1;
                }
            }
            a.score[(int64_t)b * a.N + r] = sum;
            a.common[(int64_t)b * a.N + r] = any ? 1 : 0;
        }
    }
}

// one workgroup of kHeavyRankBlock lanes per heavy query of the batch: the same ranking as sparse_query_kernel over all N rows
__global__ __launch_bounds__(kHeavyRankBlock) void sparse_heavy_rank_kernel(HeavyArgs a) {
    constexpr int KP = kHeavyRankBlock, CAP = 2 * KP;
    __shared__ unsigned long long s_buf[CAP];
    __shared__ unsigned long long s_thr;
    __shared__ int s_bcnt, s_pos, s_neg, s_hit;
    const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
    if (tid == 0) {
        s_bcnt = 0;
        s_thr = 0;
        s_pos = 0;
        s_neg = 0;
        s_hit = 0;
    }
    __syncthreads();
    const int64_t t = a.hq[b], qr = a.q_first + t;
    const int64_t ex = a.exclude ? a.exclude[t] : (a.exclude_self ? qr : (int64_t)-1);
    const float *score = a.score + (int64_t)b * a.N;
    const uint8_t *common = a.common + (int64_t)b * a.N;
    int my_pos = 0, my_neg = 0, my_hit = 0;
    rank_candidates<KP, 1>(
        a.N,
        [&](int64_t row) -> unsigned long long {
            if (!common[row]) return 0;
            my_hit++;
            if (row == ex || (a.mask && !a.mask[row])) return 0;
            const uint32_t ord = score_ord(score[row]);
            if (ord == kZeroOrd) return 0;
            my_pos += ord > kZeroOrd;
            my_neg += ord < kZeroOrd;
            return make_key(ord, (int32_t)row);
        },
        a.k, s_buf, &s_thr, &s_bcnt, tid, nt);
    if (my_pos) atomicAdd(&s_pos, my_pos);
    if (my_neg) atomicAdd(&s_neg, my_neg);
    if (my_hit) atomicAdd(&s_hit, my_hit);
    __syncthreads();
    const bool ex_counts = ex >= 0 && ex < a.N && (!a.mask || a.mask[ex]);
    const int cnt = written(s_pos, s_neg, a.n_admissible - (ex_counts ? 1 : 0), a.k);
    write_result(s_buf, cnt, a.k, t, a.out_idx, a.out_score, a.out_cnt, tid, nt);
    if (tid == 0 && a.stat) {
        atomicAdd(&a.stat[0], (unsigned long long)(a.r_ptr[a.N] - a.r_ptr[0]));
        atomicAdd(&a.stat[1], (unsigned long long)s_hit);
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
    const int32_t *new_of;  // scratch id of every row (what the posting lists store)
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
        a.p_row[at] = a.new_of[lo];
        a.p_val[at] = a.r_val[e];
    }
}

}  // namespace sparse
}  // namespace gorse
