// sparse.hip -- exact top-k over SPARSE vectors on gfx950: the sparse collections of vectors.Database
// (storage/vectors/database.go:90-97 Vector.Indices / Values; xvec.go:241-247: dimension 0 = sparse, distance Dot,
// Flat = exact index) that the IDF item-to-item / user-to-user writers fill (logics/vector_writer.go:192-209) and
// QueryItemToItem / QueryUserToUser read (logics/item_to_item.go:50-88, user_to_user.go:50-88).
// Host side of the gorse_sparse_* entry points of include/gorse_hip.h; the kernel is in sparse_kernels.hpp, the index
// construction (CSR validation, postings) in sparse_host.hpp.
#include <algorithm>

#include "common.hpp"
#include "sparse_host.hpp"
#include "sparse_kernels.hpp"

using namespace gorse;
using gorse::sparse::QueryArgs;

struct gorse_sparse {
    int device = 0;
    int64_t N = 0, nnz = 0, D = 0;
    hipStream_t stream = nullptr;
    // stored rows as CSR (the queries of all_pairs) and as postings (what every query walks)
    DevBuf<int64_t> r_ptr, p_ptr;
    DevBuf<uint32_t> r_idx;
    DevBuf<int32_t> p_row, orig_of;  // posting lists hold scratch ids (longest row first); orig_of maps them back
    DevBuf<float> r_val, p_val;
    DevBuf<uint8_t> mask;
    bool has_mask = false;
    int64_t n_admissible = 0;  // rows with mask != 0 (N without a mask)
    // per-workgroup scratch (slots x N each); stamps are never reused for a slot until the wrap-around clear
    DevBuf<sparse::Cell> cell;
    DevBuf<int32_t> touched;
    int64_t slots = 0;
    uint32_t serial = 0;
    // staging of one call
    DevBuf<int64_t> q_ptr, q_excl;
    DevBuf<uint32_t> q_idx;
    DevBuf<float> q_val, out_score;
    DevBuf<int32_t> out_idx, out_cnt;
    DevBuf<unsigned long long> stat;
    // heavy queries (row streaming): host copy of the stored rows' offsets (the lengths of all_pairs' queries), per-batch
    // score / shared-index rows
    std::vector<int64_t> r_ptr_host;
    DevBuf<float> hscore;
    DevBuf<uint8_t> hcommon;
    KernelProfile prof{1};
    int64_t last_postings = 0, last_hits = 0;
    int32_t use() const {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return fail(GORSE_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return GORSE_OK;
    }
};

namespace {

// Queries with more entries than this take the row-streaming path.  A posting-list walk costs its query about a microsecond
// per list on one wave (16K lists = a tail of ~16 ms), a row-streaming pass serves eight queries for one read of the stored
// CSR plus a binary search per (entry, query): worth it for the handful of longest queries of a popularity-skewed
// collection only (64 of the 200,000 items of the C3 shard exceed 16K users; 622 exceed 2K).  To be tuned on a device.
int64_t g_sparse_heavy_dims = 16384;
int g_sparse_hot = 0;  // probe: 512 / 1024 = that many of the longest rows keep their accumulators in LDS (0 = none)
int g_sparse_build = 0;  // 0 = postings built on the host (default), 1 = by the device kernels (test hook, see gorse_hip.h)

constexpr int64_t kScratchBudget = (int64_t)16 << 30;  // bytes of accumulator scratch (of 288 GB of HBM)
constexpr int64_t kMaxSlots = 8192;                    // 256 CUs x 32 single-wave workgroups

int64_t g_sparse_max_slots = 0;  // probe: fewer resident workgroups = a smaller live scratch footprint (0 = kMaxSlots)

int64_t slot_cap(int64_t N) {
    const int64_t most = g_sparse_max_slots > 0 ? std::min(g_sparse_max_slots, kMaxSlots) : kMaxSlots;
    return std::max<int64_t>(1, std::min<int64_t>(most, kScratchBudget / (12 * N)));
}

int32_t ensure_scratch(gorse_sparse *h, int64_t want) {
    if (want <= h->slots) return GORSE_OK;
    GORSE_TRY(h->cell.alloc((size_t)want * h->N));
    GORSE_TRY(h->touched.alloc((size_t)want * h->N));
    GORSE_HIP_CHECK(hipMemsetAsync(h->cell.p, 0, (size_t)want * h->N * sizeof(sparse::Cell), h->stream));
    h->slots = want;
    h->serial = 0;
    return GORSE_OK;
}

template <int KP>
void launch_query(const QueryArgs &a, unsigned grid, hipStream_t s) {
    if (g_sparse_hot == 512)
        sparse::sparse_query_kernel<KP, 512><<<dim3(grid), dim3(sparse::kBlock), 0, s>>>(a);
    else if (g_sparse_hot == 1024)
        sparse::sparse_query_kernel<KP, 1024><<<dim3(grid), dim3(sparse::kBlock), 0, s>>>(a);
    else
        sparse::sparse_query_kernel<KP, 0><<<dim3(grid), dim3(sparse::kBlock), 0, s>>>(a);
}

// nq queries = CSR rows q_first .. of device arrays (qp, qi, qv); results into the handle's out_* buffers and, where
// given, the host arrays
// q_len_host: q_len_host[t + 1] - q_len_host[t] = number of entries of query t (host copy of the offsets)
int32_t run_queries(gorse_sparse *h, const int64_t *qp, const uint32_t *qi, const float *qv, int64_t q_first, int64_t nq,
                    const int64_t *q_len_host, const int64_t *excl_dev, int exclude_self, int k, int32_t *idx_out,
                    float *score_out, int32_t *cnt_out) {
    const int kp = sparse::pick_kp(k);
    if (!kp) return fail(GORSE_ERR_INVALID, "k = %d: must be in 1..1024", k);
    const int64_t grid = std::min<int64_t>(nq, slot_cap(h->N));
    GORSE_TRY(ensure_scratch(h, grid));
    GORSE_TRY(h->out_idx.ensure((size_t)nq * k));
    GORSE_TRY(h->out_score.ensure((size_t)nq * k));
    GORSE_TRY(h->out_cnt.ensure((size_t)nq));
    GORSE_TRY(h->stat.ensure(2));
    const int64_t per_slot = ceil_div(nq, grid);  // queries (= stamps) one workgroup consumes in this launch
    if ((uint64_t)h->serial + (uint64_t)per_slot >= 0xFFFFFFFFull) {
        GORSE_HIP_CHECK(hipMemsetAsync(h->cell.p, 0, (size_t)h->slots * h->N * sizeof(sparse::Cell), h->stream));
        h->serial = 0;
    }
    GORSE_HIP_CHECK(hipMemsetAsync(h->stat.p, 0, 2 * sizeof(unsigned long long), h->stream));
    QueryArgs a;
    a.p_ptr = h->p_ptr.p, a.p_row = h->p_row.p, a.p_val = h->p_val.p, a.D = h->D;
    a.q_ptr = qp, a.q_idx = qi, a.q_val = qv, a.q_first = q_first, a.nq = nq;
    a.exclude = excl_dev, a.exclude_self = exclude_self;
    a.heavy_dims = g_sparse_heavy_dims > 0 ? g_sparse_heavy_dims : INT64_MAX;
    a.mask = h->has_mask ? h->mask.p : nullptr;
    a.n_admissible = h->has_mask ? h->n_admissible : h->N;
    a.N = h->N;
    a.cell = h->cell.p, a.touched = h->touched.p;
    a.orig_of = h->orig_of.p;
    a.serial_base = h->serial;
    a.k = k;
    a.out_idx = h->out_idx.p, a.out_score = h->out_score.p, a.out_cnt = h->out_cnt.p;
    a.stat = h->stat.p;
    h->serial += (uint32_t)per_slot;
    const int tok = h->prof.begin(0, h->stream);
    switch (kp) {
        case 256: launch_query<256>(a, (unsigned)grid, h->stream); break;
        case 512: launch_query<512>(a, (unsigned)grid, h->stream); break;
        default: launch_query<1024>(a, (unsigned)grid, h->stream); break;
    }
    GORSE_HIP_CHECK(hipGetLastError());
    // the queries the kernel above skipped: row streaming, kHeavyBatch of them per pass over the stored rows
    std::vector<int64_t> heavy;
    for (int64_t t = 0; t < nq; t++)
        if (q_len_host[t + 1] - q_len_host[t] > a.heavy_dims) heavy.push_back(t);
    if (!heavy.empty()) {
        GORSE_TRY(h->hscore.ensure((size_t)sparse::kHeavyBatch * h->N));
        GORSE_TRY(h->hcommon.ensure((size_t)sparse::kHeavyBatch * h->N));
        sparse::HeavyArgs ha;
        ha.r_ptr = h->r_ptr.p, ha.r_idx = h->r_idx.p, ha.r_val = h->r_val.p, ha.N = h->N;
        ha.q_ptr = qp, ha.q_idx = qi, ha.q_val = qv, ha.q_first = q_first;
        ha.score = h->hscore.p, ha.common = h->hcommon.p;
        ha.exclude = excl_dev, ha.exclude_self = exclude_self, ha.mask = a.mask, ha.n_admissible = a.n_admissible;
        ha.k = k, ha.out_idx = a.out_idx, ha.out_score = a.out_score, ha.out_cnt = a.out_cnt, ha.stat = a.stat;
        const unsigned sgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, ceil_div(h->N, 256)));
        for (size_t at = 0; at < heavy.size(); at += sparse::kHeavyBatch) {
            ha.nb = (int)std::min<size_t>(sparse::kHeavyBatch, heavy.size() - at);
            for (int b = 0; b < sparse::kHeavyBatch; b++) ha.hq[b] = b < ha.nb ? heavy[at + b] : 0;
            sparse::sparse_heavy_score_kernel<<<dim3(sgrid), dim3(256), 0, h->stream>>>(ha);
            sparse::sparse_heavy_rank_kernel<<<dim3((unsigned)ha.nb), dim3(sparse::kHeavyRankBlock), 0, h->stream>>>(ha);
        }
        GORSE_HIP_CHECK(hipGetLastError());
    }
    h->prof.end(tok, h->stream);
    unsigned long long st[2] = {0, 0};
    GORSE_HIP_CHECK(hipMemcpyAsync(st, h->stat.p, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    if (idx_out)
        GORSE_HIP_CHECK(hipMemcpyAsync(idx_out, h->out_idx.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
    if (score_out)
        GORSE_HIP_CHECK(hipMemcpyAsync(score_out, h->out_score.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
    if (cnt_out) GORSE_HIP_CHECK(hipMemcpyAsync(cnt_out, h->out_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->last_postings = (int64_t)st[0];
    h->last_hits = (int64_t)st[1];
    return GORSE_OK;
}

}  // namespace

extern "C" int32_t gorse_sparse_create(gorse_sparse **out, int32_t device, int64_t N, const int64_t *indptr,
                                       const uint32_t *indices, const float *values) {
    if (!out) return fail(GORSE_ERR_INVALID, "handle pointer is NULL");
    *out = nullptr;
    if (N <= 0 || !indptr) return fail(GORSE_ERR_INVALID, "N must be positive and indptr non-NULL");
    if (N > INT32_MAX) return fail(GORSE_ERR_INVALID, "N must fit int32");
    const std::string bad = sparse::validate_csr(N, indptr, indices);
    if (!bad.empty()) return fail(GORSE_ERR_INVALID, "stored vectors: %s", bad.c_str());
    const int64_t base = indptr[0], nnz = indptr[N] - indptr[0];
    if (nnz > 0 && !values) return fail(GORSE_ERR_INVALID, "values is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GORSE_ERR_NO_DEVICE, "no HIP device visible (libgorse_hip needs an MI355X / gfx950)");
    if (device < 0 || device >= ndev) return fail(GORSE_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    const bool on_device = g_sparse_build == 1;
    const sparse::RowOrder order = sparse::order_rows(N, indptr);
    sparse::Postings post;
    if (on_device) {  // only the index space is needed from the host
        for (int64_t t = base; t < base + nnz; t++) post.D = indices[t] >= post.D ? (int64_t)indices[t] + 1 : post.D;
        if (post.D > sparse::kMaxDims)
            return fail(GORSE_ERR_INVALID, "largest index %lld exceeds the supported index space", (long long)(post.D - 1));
    } else {
        const std::string why = sparse::build_postings(N, indptr, indices, values, post, order.new_of.data());
        if (!why.empty()) return fail(GORSE_ERR_INVALID, "%s", why.c_str());
    }
    gorse_sparse *h = new (std::nothrow) gorse_sparse();
    if (!h) return fail(GORSE_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->N = N;
    h->nnz = nnz;
    h->D = post.D;
    int32_t rc = [&]() -> int32_t {
        GORSE_TRY(h->use());
        GORSE_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        std::vector<int64_t> &ptr0 = h->r_ptr_host;
        ptr0.resize((size_t)N + 1);
        for (int64_t r = 0; r <= N; r++) ptr0[(size_t)r] = indptr[r] - base;
        GORSE_TRY(h->r_ptr.alloc((size_t)N + 1));
        GORSE_TRY(h->r_idx.alloc((size_t)nnz));
        GORSE_TRY(h->r_val.alloc((size_t)nnz));
        GORSE_TRY(h->p_ptr.alloc((size_t)post.D + 1));
        GORSE_TRY(h->p_row.alloc((size_t)nnz));
        GORSE_TRY(h->orig_of.alloc((size_t)N));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->orig_of.p, order.orig_of.data(), (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
        GORSE_TRY(h->p_val.alloc((size_t)nnz));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->r_ptr.p, ptr0.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice, h->stream));
        if (nnz > 0) {
            GORSE_HIP_CHECK(hipMemcpyAsync(h->r_idx.p, indices + base, (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
            GORSE_HIP_CHECK(hipMemcpyAsync(h->r_val.p, values + base, (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
        }
        if (on_device) {  // counting sort of the uploaded CSR entries by index: count, scan (one workgroup), scatter
            DevBuf<unsigned long long> cursor;
            DevBuf<int32_t> new_of;
            GORSE_TRY(cursor.alloc((size_t)post.D));
            GORSE_TRY(new_of.alloc((size_t)N));
            GORSE_HIP_CHECK(hipMemcpyAsync(new_of.p, order.new_of.data(), (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
            GORSE_HIP_CHECK(hipMemsetAsync(h->p_ptr.p, 0, ((size_t)post.D + 1) * 8, h->stream));
            sparse::BuildArgs b;
            b.r_ptr = h->r_ptr.p, b.r_idx = h->r_idx.p, b.r_val = h->r_val.p;
            b.N = N, b.nnz = nnz, b.D = post.D;
            b.p_ptr = reinterpret_cast<unsigned long long *>(h->p_ptr.p), b.cursor = cursor.p;
            b.p_row = h->p_row.p, b.p_val = h->p_val.p, b.new_of = new_of.p;
            const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, ceil_div(nnz, 256)));
            sparse::sparse_count_kernel<<<dim3(grid), dim3(256), 0, h->stream>>>(b);
            sparse::sparse_scan_kernel<<<dim3(1), dim3(sparse::kScanBlock), 0, h->stream>>>(b);
            sparse::sparse_scatter_kernel<<<dim3(grid), dim3(256), 0, h->stream>>>(b);
            GORSE_HIP_CHECK(hipGetLastError());
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // `cursor` and `new_of` are freed when this scope ends
        } else {
            GORSE_HIP_CHECK(hipMemcpyAsync(h->p_ptr.p, post.ptr.data(), ((size_t)post.D + 1) * 8, hipMemcpyHostToDevice,
                                           h->stream));
            if (nnz > 0) {
                GORSE_HIP_CHECK(hipMemcpyAsync(h->p_row.p, post.row.data(), (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
                GORSE_HIP_CHECK(hipMemcpyAsync(h->p_val.p, post.val.data(), (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
            }
        }
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // the host staging vectors die with this scope
        return GORSE_OK;
    }();
    if (rc != GORSE_OK) {
        std::string keep = last_error();
        gorse_sparse_destroy(h);
        last_error() = keep;
        return rc;
    }
    *out = h;
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_destroy(gorse_sparse *h) {
    if (!h) return GORSE_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_set_mask(gorse_sparse *h, const uint8_t *admissible) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    if (!admissible) {
        h->has_mask = false;
        return GORSE_OK;
    }
    GORSE_TRY(h->mask.ensure((size_t)h->N));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->mask.p, admissible, (size_t)h->N, hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->n_admissible = 0;
    for (int64_t r = 0; r < h->N; r++) h->n_admissible += admissible[r] != 0;
    h->has_mask = true;
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_search(gorse_sparse *h, int64_t nq, const int64_t *q_indptr, const uint32_t *q_indices,
                                       const float *q_values, const int64_t *exclude, int32_t k, int32_t *idx_out,
                                       float *score_out, int32_t *count_out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (nq < 0 || k <= 0 || (nq > 0 && !q_indptr)) return fail(GORSE_ERR_INVALID, "bad arguments");
    if (nq == 0) return GORSE_OK;
    const std::string bad = sparse::validate_csr(nq, q_indptr, q_indices);
    if (!bad.empty()) return fail(GORSE_ERR_INVALID, "queries: %s", bad.c_str());
    const int64_t base = q_indptr[0], qnnz = q_indptr[nq] - base;
    if (qnnz > 0 && !q_values) return fail(GORSE_ERR_INVALID, "q_values is NULL");
    if (exclude)
        for (int64_t t = 0; t < nq; t++)
            if (exclude[t] < -1 || exclude[t] >= h->N)
                return fail(GORSE_ERR_RANGE, "exclude[%lld] = %lld out of range", (long long)t, (long long)exclude[t]);
    GORSE_TRY(h->use());
    std::vector<int64_t> ptr0((size_t)nq + 1);
    for (int64_t t = 0; t <= nq; t++) ptr0[(size_t)t] = q_indptr[t] - base;
    GORSE_TRY(h->q_ptr.ensure((size_t)nq + 1));
    GORSE_TRY(h->q_idx.ensure((size_t)qnnz));
    GORSE_TRY(h->q_val.ensure((size_t)qnnz));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->q_ptr.p, ptr0.data(), ((size_t)nq + 1) * 8, hipMemcpyHostToDevice, h->stream));
    if (qnnz > 0) {
        GORSE_HIP_CHECK(hipMemcpyAsync(h->q_idx.p, q_indices + base, (size_t)qnnz * 4, hipMemcpyHostToDevice, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->q_val.p, q_values + base, (size_t)qnnz * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (exclude) {
        GORSE_TRY(h->q_excl.ensure((size_t)nq));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->q_excl.p, exclude, (size_t)nq * 8, hipMemcpyHostToDevice, h->stream));
    }
    // run_queries ends with a stream synchronisation, which also covers the uploads from ptr0
    return run_queries(h, h->q_ptr.p, h->q_idx.p, h->q_val.p, 0, nq, q_indptr, exclude ? h->q_excl.p : nullptr, 0, k, idx_out,
                       score_out, count_out);
}

extern "C" int32_t gorse_sparse_all_pairs(gorse_sparse *h, int64_t q_begin, int64_t q_end, int32_t k, int32_t exclude_self,
                                          int32_t *idx_out, float *score_out, int32_t *count_out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (q_begin < 0 || q_end > h->N || q_begin > q_end || k <= 0) return fail(GORSE_ERR_RANGE, "bad query range");
    if (q_begin == q_end) return GORSE_OK;
    GORSE_TRY(h->use());
    return run_queries(h, h->r_ptr.p, h->r_idx.p, h->r_val.p, q_begin, q_end - q_begin, h->r_ptr_host.data() + q_begin, nullptr,
                       exclude_self != 0, k, idx_out, score_out, count_out);
}

extern "C" int32_t gorse_sparse_synchronize(gorse_sparse *h) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_set_profiling(gorse_sparse *h, int32_t on) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    h->prof.on = on != 0;
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_get_profile(gorse_sparse *h, int64_t *launches, double *total_ms) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->prof.resolve();
    if (launches) *launches = h->prof.launches[0];
    if (total_ms) *total_ms = h->prof.ms[0];
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_last_stats(gorse_sparse *h, int64_t *postings, int64_t *hits) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (postings) *postings = h->last_postings;
    if (hits) *hits = h->last_hits;
    return GORSE_OK;
}

extern "C" void gorse_hip_test_set_sparse_build(int32_t mode) { g_sparse_build = mode; }
extern "C" void gorse_hip_test_set_sparse_slots(int64_t max_slots) { g_sparse_max_slots = max_slots; }
extern "C" void gorse_hip_test_set_sparse_heavy(int64_t dims) { g_sparse_heavy_dims = dims; }
extern "C" void gorse_hip_test_set_sparse_hot(int32_t rows) { g_sparse_hot = rows; }
extern "C" int32_t gorse_hip_test_sparse_set_serial(gorse_sparse *h, uint32_t serial) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    h->serial = serial;
    return GORSE_OK;
}
