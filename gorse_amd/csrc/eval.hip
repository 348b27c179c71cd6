// eval.hip -- the sampling half of cf.Evaluate on the device.
// Reference: dataset.SampleUserNegatives (dataset/dataset.go:242-253) through RandomGenerator.SampleInt32
// (common/util/random.go:108-132), consumed by Evaluate (model/cf/evaluator.go:35-72): for every user numCandidates items
// outside (the user's test feedback union the user's train feedback), and for every user WITH test feedback the candidate list
// "test items, then the negatives" that Rank (evaluator.go:162-169) scores.  A production split arrives without preloaded
// negatives (master/tasks.go:232: SplitCF + SampleUserNegatives), so every Fit paid this as a host loop before epoch 0; here one
// thread per user draws from the user's own Philox stream (seed, "neg", user) through Go's Int31n -- the reference's ONE
// sequential math/rand stream cannot be reproduced anyway -- and the candidate CSR stays resident for
// gorse_mf_rank_resident, so an Evaluate between epochs uploads nothing.
#include "goheap.hpp"
#include "mf_internal.hpp"

using namespace gorse;

namespace {

constexpr uint64_t kNegStream = 0x6e6567ull;  // "neg": the epoch word of the negatives' streams (the oracle's ORC_NEG_STREAM)
constexpr int64_t kNegMaxDraws = (int64_t)1 << 24;  // per user; the reference spins forever

__device__ __forceinline__ bool sorted_has(const int32_t *__restrict__ row, int64_t n, int32_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (row[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < n && row[lo] == x;
}
__device__ __forceinline__ bool linear_has(const int32_t *__restrict__ row, int64_t n, int32_t x) {
    for (int64_t t = 0; t < n; t++)
        if (row[t] == x) return true;
    return false;
}

// one thread per user: SampleInt32(0, I, n, test row, train row)
__global__ __launch_bounds__(128) void sample_negatives_kernel(int64_t U, int32_t I, const int64_t *__restrict__ train_ptr,
                                                               const int32_t *__restrict__ train_sorted,
                                                               const int64_t *__restrict__ test_ptr,
                                                               const int32_t *__restrict__ test_idx, int32_t n, uint64_t seed,
                                                               int32_t *__restrict__ out, int32_t *__restrict__ len,
                                                               int32_t *__restrict__ fail_count) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int32_t *tr = train_sorted + train_ptr[u];
    const int64_t ntr = train_ptr[u + 1] - train_ptr[u];
    const int32_t *te = test_idx + test_ptr[u];
    const int64_t nte = test_ptr[u + 1] - test_ptr[u];
    int64_t card = 0;  // |test set U train set|, duplicates once (mapset semantics)
    for (int64_t t = 0; t < ntr; t++) card += t == 0 || tr[t] != tr[t - 1];
    for (int64_t t = 0; t < nte; t++) card += !sorted_has(tr, ntr, te[t]) && !linear_has(te, t, te[t]);
    int32_t *o = out + u * (int64_t)n;
    int32_t got = 0;
    if ((int64_t)n >= (int64_t)I - card) {  // random.go:115-121: everything that is left, ascending
        for (int32_t i = 0; i < I && got < n; i++)
            if (!sorted_has(tr, ntr, i) && !linear_has(te, nte, i)) o[got++] = i;
    } else {
        Philox g;
        g.init(seed, kNegStream, (uint64_t)u);
        int64_t draws = 0;
        while (got < n && draws < kNegMaxDraws) {
            const int32_t v = g.int31n(I);
            draws++;
            if (!sorted_has(tr, ntr, v) && !linear_has(te, nte, v) && !linear_has(o, got, v)) o[got++] = v;
        }
        if (got < n) atomicAdd(fail_count, 1);
    }
    len[u] = got;
    for (int32_t t = got; t < n; t++) o[t] = -1;
}

// users with test feedback, ascending (the order Evaluate visits them in), and the lengths of their candidate lists
__global__ void eval_flag_kernel(int64_t U, const int64_t *__restrict__ test_ptr, const int32_t *__restrict__ neg_len,
                                 int32_t *__restrict__ has, int32_t *__restrict__ clen) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int64_t nte = test_ptr[u + 1] - test_ptr[u];
    has[u] = nte > 0;
    clen[u] = nte > 0 ? (int32_t)nte + neg_len[u] : 0;
}
// candidates = append(testSet.GetUserFeedback()[user], negatives[user]...) (evaluator.go:49-52)
__global__ void eval_fill_kernel(int64_t U, const int64_t *__restrict__ test_ptr, const int32_t *__restrict__ test_idx,
                                 const int32_t *__restrict__ neg, const int32_t *__restrict__ neg_len, int32_t n,
                                 const int64_t *__restrict__ upos, const int64_t *__restrict__ cpos,
                                 int32_t *__restrict__ users, int64_t *__restrict__ cand_ptr, int32_t *__restrict__ cand) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int64_t nte = test_ptr[u + 1] - test_ptr[u];
    if (nte <= 0) return;
    const int64_t k = upos[u], c0 = cpos[u];
    users[k] = (int32_t)u;
    cand_ptr[k] = c0;
    for (int64_t t = 0; t < nte; t++) cand[c0 + t] = test_idx[test_ptr[u] + t];
    for (int32_t t = 0; t < neg_len[u]; t++) cand[c0 + nte + t] = neg[u * (int64_t)n + t];
}

}  // namespace

// exclusive prefix sums on the host would do as well (U + 1 words); the two scans run once per split
static void host_exclusive(const std::vector<int32_t> &v, std::vector<int64_t> &out) {
    out.resize(v.size() + 1);
    out[0] = 0;
    for (size_t t = 0; t < v.size(); t++) out[t + 1] = out[t] + v[t];
}

extern "C" int32_t gorse_mf_sample_user_negatives(gorse_mf *h, const int64_t *test_indptr, const int32_t *test_indices,
                                                  int32_t num_candidates, uint64_t seed, int32_t *neg_out, int32_t *neg_len) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (!test_indptr) return fail(GORSE_ERR_INVALID, "test_indptr is NULL");
    if (num_candidates < 0) return fail(GORSE_ERR_INVALID, "num_candidates < 0");
    const int64_t U = h->U, I = h->I;
    if (test_indptr[0] != 0) return fail(GORSE_ERR_INVALID, "test_indptr[0] != 0");
    for (int64_t u = 0; u < U; u++)
        if (test_indptr[u + 1] < test_indptr[u]) return fail(GORSE_ERR_INVALID, "test_indptr not monotone at user %lld", (long long)u);
    const int64_t nt = test_indptr[U];
    if (nt > 0 && !test_indices) return fail(GORSE_ERR_INVALID, "test_indices is NULL");
    for (int64_t t = 0; t < nt; t++)
        if (test_indices[t] < 0 || test_indices[t] >= I) return fail(GORSE_ERR_RANGE, "test item %d out of range", test_indices[t]);
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    const int32_t n = num_candidates;
    GORSE_TRY(h->ev_tptr.alloc((size_t)U + 1));
    GORSE_TRY(h->ev_tidx.alloc((size_t)std::max<int64_t>(nt, 1)));
    GORSE_TRY(h->ev_neg.alloc((size_t)std::max<int64_t>(U * (int64_t)n, 1)));
    GORSE_TRY(h->ev_neglen.alloc((size_t)U));
    GORSE_TRY(h->ev_has.alloc((size_t)U));
    GORSE_TRY(h->ev_clen.alloc((size_t)U));
    h->ev_valid = false;
    GORSE_HIP_CHECK(hipMemcpyAsync(h->ev_tptr.p, test_indptr, (size_t)(U + 1) * 8, hipMemcpyHostToDevice, h->stream));
    if (nt > 0) GORSE_HIP_CHECK(hipMemcpyAsync(h->ev_tidx.p, test_indices, (size_t)nt * 4, hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipMemsetAsync(h->fail_count.p, 0, sizeof(int32_t), h->stream));
    const unsigned blocks = (unsigned)ceil_div(U, 128);
    sample_negatives_kernel<<<dim3(blocks), dim3(128), 0, h->stream>>>(U, (int32_t)I, h->uptr.p, h->uidx_sorted.p, h->ev_tptr.p,
                                                                       h->ev_tidx.p, n, seed, h->ev_neg.p, h->ev_neglen.p,
                                                                       h->fail_count.p);
    eval_flag_kernel<<<dim3(blocks), dim3(128), 0, h->stream>>>(U, h->ev_tptr.p, h->ev_neglen.p, h->ev_has.p, h->ev_clen.p);
    GORSE_HIP_CHECK(hipGetLastError());
    std::vector<int32_t> has((size_t)U), clen((size_t)U);
    int32_t failed = 0;
    GORSE_HIP_CHECK(hipMemcpyAsync(has.data(), h->ev_has.p, (size_t)U * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(clen.data(), h->ev_clen.p, (size_t)U * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(&failed, h->fail_count.p, 4, hipMemcpyDeviceToHost, h->stream));
    if (neg_out && U * (int64_t)n > 0)
        GORSE_HIP_CHECK(hipMemcpyAsync(neg_out, h->ev_neg.p, (size_t)U * n * 4, hipMemcpyDeviceToHost, h->stream));
    if (neg_len) GORSE_HIP_CHECK(hipMemcpyAsync(neg_len, h->ev_neglen.p, (size_t)U * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (failed) return fail(GORSE_ERR_INVALID, "%d users found no %d negatives within the draw limit", failed, n);
    std::vector<int64_t> upos, cpos;
    host_exclusive(has, upos);
    host_exclusive(clen, cpos);
    h->ev_users_n = upos[(size_t)U];
    h->ev_cand_n = cpos[(size_t)U];
    GORSE_TRY(h->ev_upos.alloc((size_t)U + 1));
    GORSE_TRY(h->ev_cpos.alloc((size_t)U + 1));
    GORSE_TRY(h->ev_users.alloc((size_t)std::max<int64_t>(h->ev_users_n, 1)));
    GORSE_TRY(h->ev_cptr.alloc((size_t)h->ev_users_n + 1));
    GORSE_TRY(h->ev_cand.alloc((size_t)std::max<int64_t>(h->ev_cand_n, 1)));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->ev_upos.p, upos.data(), (size_t)(U + 1) * 8, hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->ev_cpos.p, cpos.data(), (size_t)(U + 1) * 8, hipMemcpyHostToDevice, h->stream));
    eval_fill_kernel<<<dim3(blocks), dim3(128), 0, h->stream>>>(U, h->ev_tptr.p, h->ev_tidx.p, h->ev_neg.p, h->ev_neglen.p, n,
                                                                h->ev_upos.p, h->ev_cpos.p, h->ev_users.p, h->ev_cptr.p,
                                                                h->ev_cand.p);
    GORSE_HIP_CHECK(hipGetLastError());
    GORSE_HIP_CHECK(hipMemcpyAsync(h->ev_cptr.p + h->ev_users_n, &h->ev_cand_n, 8, hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // upos / cpos are host temporaries
    h->ev_valid = true;
    h->ev_generation++;  // every sampling replaces the resident lists: a caller that kept the old number knows they are not its own
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_resident_candidates(gorse_mf *h, int64_t *n_users, int64_t *n_candidates) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (!h->ev_valid) return fail(GORSE_ERR_INVALID, "no resident candidate lists (gorse_mf_sample_user_negatives first)");
    if (n_users) *n_users = h->ev_users_n;
    if (n_candidates) *n_candidates = h->ev_cand_n;
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_resident_generation(gorse_mf *h, uint64_t *generation) {
    if (!h || !generation) return fail(GORSE_ERR_INVALID, "NULL argument");
    *generation = h->ev_valid ? h->ev_generation : 0;
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_rank_resident(gorse_mf *h, int32_t topk, int32_t *users_out, int32_t *rank_out, int32_t *rank_len) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (!h->ev_valid) return fail(GORSE_ERR_INVALID, "no resident candidate lists (gorse_mf_sample_user_negatives first)");
    if (topk <= 0) return fail(GORSE_ERR_INVALID, "topk <= 0");
    if (h->ev_users_n == 0) return GORSE_OK;
    if (!rank_out || !rank_len) return fail(GORSE_ERR_INVALID, "NULL argument");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    GORSE_TRY(mf_rank_device(h, h->ev_users_n, h->ev_users.p, h->ev_cptr.p, h->ev_cand.p, h->ev_cand_n, topk, rank_out, rank_len));
    if (users_out)
        GORSE_HIP_CHECK(hipMemcpyAsync(users_out, h->ev_users.p, (size_t)h->ev_users_n * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}
