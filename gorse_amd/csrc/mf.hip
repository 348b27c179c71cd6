// mf.hip -- gorse_mf lifetime, factor transfer, internalPredict (score), Rank, exchange helpers.
// Reference: model/cf/model.go:118-203 (BaseMatrixFactorization), evaluator.go:162-169 (Rank),
// common/heap/filter.go:23-59 (TopKFilter).
#include <algorithm>
#include <chrono>
#include <cstdlib>

#include "goheap.hpp"
#include <time.h>

#include "mf_internal.hpp"

using namespace gorse;

extern "C" int32_t gorse_hip_abi_version(void) { return GORSE_HIP_ABI_VERSION; }
extern "C" const char *gorse_hip_last_error(void) { return last_error().c_str(); }
extern "C" int32_t gorse_hip_device_count(int32_t *n) {
    if (!n) return fail(GORSE_ERR_INVALID, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(GORSE_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return GORSE_OK;
}

namespace {

// each CSR row sorted ascending, on the host (one-off at create time; the rows are cut over the host's cores)
void sort_rows(const int64_t *ptr, const int32_t *idx, int64_t rows, std::vector<int32_t> &out) {
    out.resize((size_t)ptr[rows]);
    parallel_rows(rows, ptr, [&](int, int64_t r0, int64_t r1) {
        std::copy(idx + ptr[r0], idx + ptr[r1], out.data() + ptr[r0]);
        for (int64_t r = r0; r < r1; r++) {
            int32_t *b = out.data() + ptr[r], *e = out.data() + ptr[r + 1];
            if (e - b > 1 && !std::is_sorted(b, e)) std::sort(b, e);
        }
    });
}

int32_t validate_csr(const char *name, const int64_t *ptr, const int32_t *idx, int64_t rows, int64_t cols) {
    if (ptr[0] != 0) return fail(GORSE_ERR_INVALID, "%s_indptr[0] != 0", name);
    for (int64_t r = 0; r < rows; r++)
        if (ptr[r + 1] < ptr[r]) return fail(GORSE_ERR_INVALID, "%s_indptr not monotone at row %lld", name, (long long)r);
    std::vector<int64_t> bad(64, -1);  // first offending entry seen by each worker
    parallel_rows(rows, ptr, [&](int t, int64_t r0, int64_t r1) {
        for (int64_t e = ptr[r0]; e < ptr[r1]; e++)
            if (idx[e] < 0 || idx[e] >= cols) {
                bad[(size_t)t] = e;
                return;
            }
    });
    for (int64_t e : bad)
        if (e >= 0)
            return fail(GORSE_ERR_INVALID, "%s_indices[%lld] = %d out of range [0,%lld)", name, (long long)e, idx[e],
                        (long long)cols);
    return GORSE_OK;
}

// ---- internalPredict ---------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(kBlock) void mf_score_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                                                          const int32_t *__restrict__ us,
                                                          const int32_t *__restrict__ is, int64_t n, int d,
                                                          float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & (kGroup - 1);
    const int gib = threadIdx.x / kGroup;
    const int64_t group = (int64_t)blockIdx.x * kGroupsPerBlock + gib;
    const int64_t ngroups = (int64_t)gridDim.x * kGroupsPerBlock;
    const VecShape vs(d);
    for (int64_t t = group; t < n; t += ngroups) {
        const int u = us[t], i = is[t];
        float r = 0.0f;
        if (u >= 0 && i >= 0) {
            const float *pu = P + (int64_t)u * d, *qi = Q + (int64_t)i * d;
            if constexpr (NC > 0) {
                float a[NC], b[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    a[c] = pu[16 * c + lane];
                    b[c] = qi[16 * c + lane];
                }
                r = dot512_regs<NC>(a, b);
            } else {
                float *sa = smem + (size_t)gib * 2 * d, *sb = sa + d;
                for (int e = lane; e < d; e += kGroup) {
                    sa[e] = pu[e];
                    sb[e] = qi[e];
                }
                __builtin_amdgcn_wave_barrier();
                r = dot512_lds(sa, sb, vs, lane);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) out[t] = r;
    }
}

// ---- heap.TopKFilter per user over precomputed scores ------------------------------------------
__global__ void mf_rank_kernel(const int64_t *__restrict__ cand_ptr, const int32_t *__restrict__ cand,
                               const float *__restrict__ score, int64_t n_users, int topk, int32_t *heap_v,
                               float *heap_w, int32_t *__restrict__ rank_out, int32_t *__restrict__ rank_len) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_users) return;
    GoHeap<false> h(heap_v + t * (topk + 1), heap_w + t * (topk + 1));
    for (int64_t c = cand_ptr[t]; c < cand_ptr[t + 1]; c++) {
        h.push(cand[c], score[c]);
        if (h.n > topk) h.pop();
    }
    const int cnt = h.n;
    rank_len[t] = cnt;
    for (int k = 0; k < topk; k++) rank_out[t * topk + k] = -1;
    for (int k = cnt - 1; k >= 0; k--) {
        h.pop();  // root moved to position h.n
        rank_out[t * topk + k] = h.v[h.n];
    }
}

__global__ void expand_users_kernel(const int64_t *__restrict__ cand_ptr, const int32_t *__restrict__ users,
                                    int64_t n_users, int32_t *__restrict__ pair_u) {
    int64_t t = blockIdx.x;
    for (; t < n_users; t += gridDim.x)
        for (int64_t c = cand_ptr[t] + threadIdx.x; c < cand_ptr[t + 1]; c += blockDim.x) pair_u[c] = users[t];
}

__global__ void delta_export_kernel(const float4 *__restrict__ q, const float4 *__restrict__ qs, float4 *__restrict__ dst,
                                    int64_t n4) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (int64_t)gridDim.x * blockDim.x) {
        float4 a = q[t], b = qs[t];
        dst[t] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    }
}
__global__ void delta_import_kernel(float4 *__restrict__ q, float4 *__restrict__ qs, const float4 *__restrict__ src,
                                    int64_t n4) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (int64_t)gridDim.x * blockDim.x) {
        float4 b = qs[t], s = src[t];
        float4 r = make_float4(b.x + s.x, b.y + s.y, b.z + s.z, b.w + s.w);
        q[t] = r;
        qs[t] = r;
    }
}
__global__ void delta_export_tail(const float *q, const float *qs, float *dst, int64_t begin, int64_t n) {
    int64_t t = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = q[t] - qs[t];
}
__global__ void delta_import_tail(float *q, float *qs, const float *src, int64_t begin, int64_t n) {
    int64_t t = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        float r = qs[t] + src[t];
        q[t] = r;
        qs[t] = r;
    }
}

}  // namespace

namespace gorse {

int32_t mf_sync_streams(gorse_mf *h) {
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream2));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

// ---- epoch pacing ---------------------------------------------------------------------------------------------------------
// A Fit that enqueues its epochs between two evaluations (gorse_bpr_epoch_enqueue) neither knows how long an epoch took on the
// device -- the reference logs that as fit_time (model.go:496-503) -- nor sees a cancelled context before its next synchronous
// call (the reference checks per sample, model.go:449).  Every epoch is therefore bracketed by a (begin, end) event pair on the update
// stream: begin = the stream reaches the epoch (= the end of the previous epoch's last update kernel when epochs follow each other),
// end = its last update kernel is done.  gorse_mf_epoch_throttle waits on them with the cancel flag in hand, gorse_mf_epoch_times
// adds up their spans.
int32_t mf_epoch_begin(gorse_mf *h, bool chained) {
    if (!h->ep_events) {
        for (int i = 0; i < gorse_mf::kEpochRing; i++) {
            GORSE_HIP_CHECK(hipEventCreate(&h->ev_ep_begin[i]));
            GORSE_HIP_CHECK(hipEventCreate(&h->ev_ep_end[i]));
        }
        h->ep_events = true;
    }
    // (one slot of margin: an epoch's begin may be the END event of the slot before it, which must outlive the epoch's own harvest)
    if (h->ep_seq - h->ep_done >= (uint64_t)gorse_mf::kEpochRing - 1) {  // the slot about to be reused was never read
        GORSE_TRY(mf_epoch_harvest(h, false));
        while (h->ep_seq - h->ep_done >= (uint64_t)gorse_mf::kEpochRing - 1) {
            h->ep_done++;
            h->ep_untimed++;
        }
    }
    const int slot = (int)(h->ep_seq % gorse_mf::kEpochRing);
    // chained (gorse_mf::ep_begin_prev): the previous epoch was the last thing issued on this handle and has not ended yet -- the update
    // stream reaches this epoch when that one ends
    bool from_prev = false;
    if (chained && h->ep_seq > 0) {
        const hipError_t q = hipEventQuery(h->ev_ep_end[(h->ep_seq - 1) % gorse_mf::kEpochRing]);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            from_prev = true;
        } else {
            GORSE_HIP_CHECK(q);
        }
    }
    h->ep_begin_prev[slot] = from_prev;
    if (!from_prev) GORSE_HIP_CHECK(hipEventRecord(h->ev_ep_begin[slot], h->stream));
    return GORSE_OK;
}
int32_t mf_epoch_end(gorse_mf *h) {
    GORSE_HIP_CHECK(hipEventRecord(h->ev_ep_end[h->ep_seq % gorse_mf::kEpochRing], h->stream));
    h->ep_seq++;
    return GORSE_OK;
}
int32_t mf_epoch_harvest(gorse_mf *h, bool wait) {
    while (h->ep_done < h->ep_seq) {
        const int slot = (int)(h->ep_done % gorse_mf::kEpochRing);
        if (wait) {
            GORSE_HIP_CHECK(hipEventSynchronize(h->ev_ep_end[slot]));
        } else {
            const hipError_t q = hipEventQuery(h->ev_ep_end[slot]);
            if (q == hipErrorNotReady) {
                (void)hipGetLastError();
                break;
            }
            GORSE_HIP_CHECK(q);
        }
        float ms = 0.0f;
        const hipEvent_t begin = h->ep_begin_prev[slot] ? h->ev_ep_end[(slot + gorse_mf::kEpochRing - 1) % gorse_mf::kEpochRing] : h->ev_ep_begin[slot];
        GORSE_HIP_CHECK(hipEventElapsedTime(&ms, begin, h->ev_ep_end[slot]));
        h->ep_ms += ms;
        h->ep_timed++;
        h->ep_done++;
    }
    return GORSE_OK;
}

// score launcher shared with the rank path (device pointers)
int32_t mf_score_device(gorse_mf *h, const int32_t *us, const int32_t *is, int64_t n, float *out) {
    if (n == 0) return GORSE_OK;
    const int d = h->d;
    int64_t blocks = ceil_div(n, kGroupsPerBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks), block(kBlock);
    if (d == 16)
        mf_score_kernel<1><<<grid, block, 0, h->stream>>>(h->P.p, h->Q.p, us, is, n, d, out);
    else if (d == 32)
        mf_score_kernel<2><<<grid, block, 0, h->stream>>>(h->P.p, h->Q.p, us, is, n, d, out);
    else if (d == 64)
        mf_score_kernel<4><<<grid, block, 0, h->stream>>>(h->P.p, h->Q.p, us, is, n, d, out);
    else if (d == 128)
        mf_score_kernel<8><<<grid, block, 0, h->stream>>>(h->P.p, h->Q.p, us, is, n, d, out);
    else
        mf_score_kernel<0><<<grid, block, (size_t)kGroupsPerBlock * 2 * d * sizeof(float), h->stream>>>(
            h->P.p, h->Q.p, us, is, n, d, out);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

}  // namespace gorse

int g_mf_flat_streams = 1;  // 1 = both streams of a handle at the same priority; 0 (probe) = the update stream ahead
extern "C" void gorse_hip_test_set_stream_priorities(int32_t on) { g_mf_flat_streams = on ? 0 : 1; }
int g_mf_prep_cu_stride = 0;  // probe: > 1 = the preparation stream of a handle created afterwards runs on every n-th CU only
extern "C" void gorse_hip_test_set_prep_cu_stride(int32_t n) { g_mf_prep_cu_stride = n; }
// 32768: no item of a catalogue smaller than that is cold (S-ml1m: every update stays an atomic; with 2048 there, a third of
// the cold rows' updates were overwritten); at C3 98 % of the items are
constexpr int64_t kDefaultColdWindow = 32768;
int64_t g_mf_cold_window = kDefaultColdWindow;  // items expected to be touched less than once per this many samples are "cold" (0 = none)
extern "C" void gorse_hip_test_set_bpr_cold_window(int64_t samples) { g_mf_cold_window = samples < 0 ? kDefaultColdWindow : samples; }

// the cold classes of one handle for another window: warm <-> cold only, the hot slots stay (include/gorse_hip.h)
extern "C" int32_t gorse_mf_set_bpr_cold_window(gorse_mf *h, int64_t samples, int64_t *n_cold) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (samples < 0) return fail(GORSE_ERR_INVALID, "cold window < 0");
    GORSE_TRY(h->use());
    if ((int64_t)h->h_hot_slot.size() != h->I) return fail(GORSE_ERR_INVALID, "this handle has no item classes");
    GORSE_TRY(mf_sync_streams(h));  // no epoch may be reading the classes
    h->cold_window = samples;
    h->n_cold = 0;
    for (int64_t i = 0; i < h->I; i++) {
        int32_t &s = h->h_hot_slot[(size_t)i];
        if (s >= 0) continue;  // hot: replica slot
        const bool cold = samples > 0 && h->nnz > 0 &&
                          ((double)h->h_item_count[(size_t)i] / (double)h->nnz + 1.0 / (double)h->I) * (double)samples < 1.0;
        s = cold ? -2 : -1;
        h->n_cold += cold;
    }
    GORSE_HIP_CHECK(hipMemcpyAsync(h->hot_slot.p, h->h_hot_slot.data(), (size_t)h->I * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (n_cold) *n_cold = h->n_cold;
    return GORSE_OK;
}

// GORSE_MF_TRACE=1: the phases of gorse_mf_create / gorse_mf_destroy in milliseconds on stderr (where a short Fit's time goes)
namespace {
struct CreateTrace {
    bool on = getenv("GORSE_MF_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gorse_mf] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};
}  // namespace

extern "C" int32_t gorse_mf_create(gorse_mf **out, int32_t device, int64_t U, int64_t I, int32_t d,
                                   const int64_t *user_indptr, const int32_t *user_indices,
                                   const int64_t *item_indptr, const int32_t *item_indices) {
    if (!out) return fail(GORSE_ERR_INVALID, "handle pointer is NULL");
    *out = nullptr;
    if (U <= 0 || I <= 0 || d <= 0) return fail(GORSE_ERR_INVALID, "U, I, d must be positive (got %lld, %lld, %d)",
                                                 (long long)U, (long long)I, d);
    if (U > INT32_MAX || I > INT32_MAX) return fail(GORSE_ERR_INVALID, "U and I must fit int32 (dataset indices are int32)");
    // the generic kernels stage rows in LDS: bpr_update_kernel<0> 192 bytes per factor (16 groups x 3 rows), the score / rank
    // kernels less; what must launch decides the limit (64 KB of LDS without opting into more: 341; kept at a round 256)
    if (d > 256) return fail(GORSE_ERR_INVALID, "nFactors %d > 256 unsupported (the generic BPR update stages 192 B of LDS per factor)", d);
    if (!user_indptr || !user_indices) return fail(GORSE_ERR_INVALID, "user CSR is NULL");
    if ((item_indptr == nullptr) != (item_indices == nullptr))
        return fail(GORSE_ERR_INVALID, "item_indptr and item_indices must both be given or both NULL");
    CreateTrace trace;
    GORSE_TRY(validate_csr("user", user_indptr, user_indices, U, I));
    if (item_indptr) GORSE_TRY(validate_csr("item", item_indptr, item_indices, I, U));
    trace.mark("create: validate");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GORSE_ERR_NO_DEVICE, "no HIP device visible (libgorse_hip needs an MI355X / gfx950)");
    if (device < 0 || device >= ndev) return fail(GORSE_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    gorse_mf *h = new (std::nothrow) gorse_mf();
    if (!h) return fail(GORSE_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->U = U;
    h->I = I;
    h->d = d;
    h->nnz = user_indptr[U];
    h->has_item_csr = item_indptr != nullptr;
    for (int64_t r = 0; r < U; r++) h->max_user_row = std::max(h->max_user_row, user_indptr[r + 1] - user_indptr[r]);
    if (item_indptr)
        for (int64_t r = 0; r < I; r++) h->max_item_row = std::max(h->max_item_row, item_indptr[r + 1] - item_indptr[r]);
    int32_t rc = [&]() -> int32_t {
        GORSE_TRY(h->use());
        // Both streams at the same priority.  Probe (gorse_hip_test_set_stream_priorities(1)): the update stream at the highest
        // stream priority and the sampler / sort stream at the lowest changes nothing at C2 (0.697 vs 0.702 ms per epoch) and
        // costs 3 % at the C3 shard (13.24 vs 12.80 ms): profiles/r02_ak_probe_stream_prio.txt.
        if (g_mf_prep_cu_stride > 1) {
            // probe: the preparation stream on every n-th CU only (hipExtStreamCreateWithCUMask): the chunk's preparation then
            // disturbs the update kernel it runs beside on 1 / n of the chip (gorse_hip_test_set_prep_cu_stride)
            int ncu = 256;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int c = 0; c < ncu; c += g_mf_prep_cu_stride) mask[(size_t)c / 32] |= 1u << (c % 32);
            GORSE_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
            GORSE_HIP_CHECK(hipExtStreamCreateWithCUMask(&h->stream2, (uint32_t)mask.size(), mask.data()));
        } else if (g_mf_flat_streams) {
            // (Round 5 kept the streams and events of a destroyed handle for the next one -- creating and destroying them costs ~8 ms
            // per handle, a third of a whole BPR.Fit at the reference's own test shape.  Reverted: the idle streams keep hardware queues,
            // the runtime has four per process, and the two streams of the NEXT handle -- the sparse index's list walk and its
            // heavy-query kernel in the same bench process -- then shared one and ran one after the other: 36.9 -> 43.7 ms,
            // profiles/r05_zy_*.)
            GORSE_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
            GORSE_HIP_CHECK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        } else {
            int prio_lo = 0, prio_hi = 0;
            GORSE_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));  // lo = least urgent (largest number)
            GORSE_HIP_CHECK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi));
            GORSE_HIP_CHECK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_lo));
        }
        for (int b = 0; b < 2; b++) {
            GORSE_HIP_CHECK(hipEventCreateWithFlags(&h->ev_sampled[b], hipEventDisableTiming));
            GORSE_HIP_CHECK(hipEventCreateWithFlags(&h->ev_consumed[b], hipEventDisableTiming));
        }
        trace.mark("create: streams, events");
        // one row more than the matrix has, and it stays zero: the ALS gathers send the entries past a row's end there (row id U
        // resp. I) instead of selecting an address per load
        GORSE_TRY(h->P.alloc((size_t)(U + 1) * d));
        GORSE_TRY(h->Q.alloc((size_t)(I + 1) * d));
        GORSE_HIP_CHECK(hipMemsetAsync(h->P.p, 0, (size_t)(U + 1) * d * sizeof(float), h->stream));
        GORSE_HIP_CHECK(hipMemsetAsync(h->Q.p, 0, (size_t)(I + 1) * d * sizeof(float), h->stream));
        GORSE_TRY(h->uptr.alloc((size_t)U + 1));
        GORSE_TRY(h->uidx.alloc((size_t)h->nnz));
        GORSE_TRY(h->uidx_sorted.alloc((size_t)h->nnz));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->uptr.p, user_indptr, (size_t)(U + 1) * sizeof(int64_t), hipMemcpyHostToDevice,
                                       h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->uidx.p, user_indices, (size_t)h->nnz * sizeof(int32_t), hipMemcpyHostToDevice,
                                       h->stream));
        trace.mark("create: P, Q, user CSR");
        std::vector<int32_t> sorted;
        sort_rows(user_indptr, user_indices, U, sorted);
        trace.mark("create: sort_rows");
        GORSE_HIP_CHECK(hipMemcpyAsync(h->uidx_sorted.p, sorted.data(), (size_t)h->nnz * sizeof(int32_t),
                                       hipMemcpyHostToDevice, h->stream));
        if (h->has_item_csr) {
            int64_t innz = item_indptr[I];
            GORSE_TRY(h->iptr.alloc((size_t)I + 1));
            GORSE_TRY(h->iidx.alloc((size_t)innz));
            GORSE_HIP_CHECK(hipMemcpyAsync(h->iptr.p, item_indptr, (size_t)(I + 1) * sizeof(int64_t),
                                           hipMemcpyHostToDevice, h->stream));
            GORSE_HIP_CHECK(hipMemcpyAsync(h->iidx.p, item_indices, (size_t)innz * sizeof(int32_t), hipMemcpyHostToDevice,
                                           h->stream));
            h->h_uptr.assign(user_indptr, user_indptr + U + 1);
            h->h_iptr.assign(item_indptr, item_indptr + I + 1);
            h->als_hi[0] = U;
            h->als_hi[1] = I;
            GORSE_TRY(als_build_plan(h, 0, user_indptr, U, 0, U));
            GORSE_TRY(als_build_plan(h, 1, item_indptr, I, 0, I));
        }
        trace.mark("create: item CSR, ALS plan");
        {   // hot items: share of the training feedback >= 1/8192 (and >= 64 feedbacks), at most 1024 and a quarter of the items (bpr.hip, HotRows)
            std::vector<int64_t> cnt((size_t)I, 0);
            if (h->nnz < ((int64_t)1 << 26) || I > ((int64_t)1 << 22)) {
                for (int64_t t = 0; t < h->nnz; t++) cnt[user_indices[t]]++;
            } else {  // a billion entries: per-worker counters, added up afterwards
                std::vector<std::vector<int32_t>> part(64);
                parallel_rows(U, user_indptr, [&](int t, int64_t r0, int64_t r1) {
                    part[(size_t)t].assign((size_t)I, 0);
                    int32_t *c = part[(size_t)t].data();
                    for (int64_t e = user_indptr[r0]; e < user_indptr[r1]; e++) c[user_indices[e]]++;
                });
                for (auto &pc : part)
                    if (!pc.empty())
                        for (int64_t i = 0; i < I; i++) cnt[(size_t)i] += pc[(size_t)i];
            }
            // 1/2048 until round 3: with the negatives routed through the replicas too, 1/8192 is 6 % faster at C2
            // (profiles/r03_zx_probe_bpr_hot.txt).  A replica's content reaches Q one folder pass late, so the hot items stay a
            // minority: at least 64 feedbacks, at most a quarter of the items (S-ml100k with two thirds of its items hot lost
            // 0.011 of NDCG@10 in the per-sample schedule; with every item hot S-ml1m's fit diverges)
            const int64_t hdiv = 8192;
            const size_t hcap = (size_t)std::min<int64_t>(1024, std::max<int64_t>(1, I / 4));
            const int64_t thr = std::max<int64_t>(64, (h->nnz + hdiv - 1) / hdiv);
            std::vector<int32_t> hot;
            for (int64_t i = 0; i < I; i++)
                if (cnt[i] >= thr) hot.push_back((int32_t)i);
            if (hot.size() > hcap) {
                std::nth_element(hot.begin(), hot.begin() + hcap, hot.end(),
                                 [&](int32_t a, int32_t b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; });
                hot.resize(hcap);
                std::sort(hot.begin(), hot.end());
            }
            std::vector<int32_t> slot((size_t)I, -1);
            // cold items (class -2, bpr.hip kCold): a sample touches item i with probability share(i) as its positive and 1 / I as
            // its negative; where fewer than one touch is expected per `cold window` samples the row's update may be a plain
            // write-through store (the reference's own unlocked write) instead of d atomic dwords -- in GORSE_BPR_HOGWILD_STORES
            // only.  The window is the handle's own (gorse_mf_set_bpr_cold_window re-classifies from the counts kept here).
            h->h_item_count.resize((size_t)I);
            for (int64_t i = 0; i < I; i++) h->h_item_count[(size_t)i] = (int32_t)std::min<int64_t>(cnt[(size_t)i], INT32_MAX);
            h->cold_window = g_mf_cold_window;
            h->n_cold = 0;
            if (h->cold_window > 0 && h->nnz > 0) {
                const double w = (double)h->cold_window;
                for (int64_t i = 0; i < I; i++)
                    if (((double)cnt[(size_t)i] / (double)h->nnz + 1.0 / (double)I) * w < 1.0) {
                        slot[(size_t)i] = -2;
                        h->n_cold++;
                    }
            }
            for (size_t k = 0; k < hot.size(); k++) slot[hot[k]] = (int32_t)k;
            h->h_hot_slot = slot;
            h->n_hot = (int)hot.size();
            GORSE_TRY(h->hot_slot.alloc((size_t)I));
            GORSE_TRY(h->hot_items.alloc(hot.size()));
            GORSE_TRY(h->hot_rep.alloc(hot.size() * GORSE_HOT_REPLICAS_ALLOC * (size_t)d));
            GORSE_TRY(h->hot_done.alloc((size_t)GORSE_HOT_DONE_STRIPES * GORSE_HOT_DONE_STRIDE));  // (bpr.hip worker_done)
            GORSE_HIP_CHECK(hipMemsetAsync(h->hot_done.p, 0, (size_t)GORSE_HOT_DONE_STRIPES * GORSE_HOT_DONE_STRIDE * sizeof(int32_t), h->stream));
            GORSE_HIP_CHECK(hipMemcpyAsync(h->hot_slot.p, slot.data(), (size_t)I * sizeof(int32_t), hipMemcpyHostToDevice,
                                           h->stream));
            if (!hot.empty())
                GORSE_HIP_CHECK(hipMemcpyAsync(h->hot_items.p, hot.data(), hot.size() * sizeof(int32_t),
                                               hipMemcpyHostToDevice, h->stream));
            GORSE_HIP_CHECK(hipMemsetAsync(h->hot_rep.p, 0, std::max<size_t>(1, hot.size() * GORSE_HOT_REPLICAS_ALLOC * (size_t)d) * sizeof(float),
                                           h->stream));
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // slot / hot are host temporaries
        }
        trace.mark("create: item classes");
        GORSE_TRY(h->loss.alloc(1));
        GORSE_TRY(h->fail_count.alloc(1));
        GORSE_HIP_CHECK(hipMemsetAsync(h->fail_count.p, 0, sizeof(int32_t), h->stream));
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // host staging vectors go out of scope
        return GORSE_OK;
    }();
    if (rc != GORSE_OK) {
        std::string keep = last_error();
        gorse_mf_destroy(h);
        last_error() = keep;
        return rc;
    }
    trace.mark("create: last sync");
    *out = h;
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_destroy(gorse_mf *h) {
    if (!h) return GORSE_OK;
    CreateTrace trace;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    for (int b = 0; b < 2; b++) {
        if (h->ev_sampled[b]) (void)hipEventDestroy(h->ev_sampled[b]);
        if (h->ev_consumed[b]) (void)hipEventDestroy(h->ev_consumed[b]);
    }
    if (h->ep_events)
        for (int i = 0; i < gorse_mf::kEpochRing; i++) {
            (void)hipEventDestroy(h->ev_ep_begin[i]);
            (void)hipEventDestroy(h->ev_ep_end[i]);
        }
    trace.mark("destroy: sync, events");
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    trace.mark("destroy: streams");
    delete h;
    trace.mark("destroy: buffers");
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_set_factors(gorse_mf *h, const float *P, const float *Q) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    if (P) GORSE_HIP_CHECK(hipMemcpyAsync(h->P.p, P, (size_t)h->U * h->d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (Q) GORSE_HIP_CHECK(hipMemcpyAsync(h->Q.p, Q, (size_t)h->I * h->d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_get_factors(gorse_mf *h, float *P, float *Q) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    if (P) GORSE_HIP_CHECK(hipMemcpyAsync(P, h->P.p, (size_t)h->U * h->d * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (Q) GORSE_HIP_CHECK(hipMemcpyAsync(Q, h->Q.p, (size_t)h->I * h->d * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_score(gorse_mf *h, const int32_t *u, const int32_t *items, int64_t n, float *out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (n < 0 || (n > 0 && (!u || !items || !out))) return fail(GORSE_ERR_INVALID, "bad arguments");
    if (n == 0) return GORSE_OK;
    for (int64_t t = 0; t < n; t++)
        if (u[t] >= h->U || items[t] >= h->I)
            return fail(GORSE_ERR_RANGE, "pair %lld (%d,%d) out of range", (long long)t, u[t], items[t]);
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    size_t bytes = (size_t)n * (2 * sizeof(int32_t) + sizeof(float));
    GORSE_TRY(h->stage.ensure(bytes));
    int32_t *du = (int32_t *)h->stage.p, *di = du + n;
    float *dout = (float *)(di + n);
    GORSE_HIP_CHECK(hipMemcpyAsync(du, u, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(di, items, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    GORSE_TRY(mf_score_device(h, du, di, n, dout));
    GORSE_HIP_CHECK(hipMemcpyAsync(out, dout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_rank(gorse_mf *h, int64_t n_users, const int32_t *users, const int64_t *cand_indptr,
                                 const int32_t *cand, int32_t topk, int32_t *rank_out, int32_t *rank_len) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (n_users < 0 || topk <= 0) return fail(GORSE_ERR_INVALID, "n_users < 0 or topk <= 0");
    if (n_users == 0) return GORSE_OK;
    if (!users || !cand_indptr || !rank_out || !rank_len) return fail(GORSE_ERR_INVALID, "NULL argument");
    const int64_t nc = cand_indptr[n_users];
    if (cand_indptr[0] != 0 || nc < 0 || (nc > 0 && !cand)) return fail(GORSE_ERR_INVALID, "bad candidate CSR");
    for (int64_t t = 0; t < n_users; t++) {
        if (users[t] < 0 || users[t] >= h->U) return fail(GORSE_ERR_RANGE, "user %d out of range", users[t]);
        if (cand_indptr[t + 1] < cand_indptr[t]) return fail(GORSE_ERR_INVALID, "cand_indptr not monotone");
    }
    for (int64_t c = 0; c < nc; c++)
        if (cand[c] < 0 || cand[c] >= h->I) return fail(GORSE_ERR_RANGE, "candidate %d out of range", cand[c]);
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    // inputs to the device (their own buffer: mf_rank_device carves h->stage for its scratch)
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_ptr = carve((size_t)(n_users + 1) * 8), o_users = carve((size_t)n_users * 4), o_cand = carve((size_t)std::max<int64_t>(nc, 1) * 4);
    GORSE_TRY(h->rank_in.ensure(off));
    char *base = h->rank_in.p;
    int64_t *d_ptr = (int64_t *)(base + o_ptr);
    int32_t *d_users = (int32_t *)(base + o_users), *d_cand = (int32_t *)(base + o_cand);
    GORSE_HIP_CHECK(hipMemcpyAsync(d_ptr, cand_indptr, (size_t)(n_users + 1) * 8, hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(d_users, users, (size_t)n_users * 4, hipMemcpyHostToDevice, h->stream));
    if (nc > 0) GORSE_HIP_CHECK(hipMemcpyAsync(d_cand, cand, (size_t)nc * 4, hipMemcpyHostToDevice, h->stream));
    GORSE_TRY(mf_rank_device(h, n_users, d_users, d_ptr, d_cand, nc, topk, rank_out, rank_len));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

int32_t gorse::mf_rank_device(gorse_mf *h, int64_t n_users, const int32_t *d_users, const int64_t *d_ptr, const int32_t *d_cand,
                              int64_t nc, int32_t topk, int32_t *rank_out, int32_t *rank_len) {
    // scratch layout: pair_u | score | heap_v | heap_w | rank | len
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_pu = carve((size_t)std::max<int64_t>(nc, 1) * 4), o_score = carve((size_t)std::max<int64_t>(nc, 1) * 4),
                 o_hv = carve((size_t)n_users * (topk + 1) * 4), o_hw = carve((size_t)n_users * (topk + 1) * 4),
                 o_rank = carve((size_t)n_users * topk * 4), o_len = carve((size_t)n_users * 4);
    GORSE_TRY(h->stage.ensure(off));
    char *base = h->stage.p;
    int32_t *d_pu = (int32_t *)(base + o_pu);
    float *d_score = (float *)(base + o_score);
    int32_t *d_hv = (int32_t *)(base + o_hv);
    float *d_hw = (float *)(base + o_hw);
    int32_t *d_rank = (int32_t *)(base + o_rank), *d_len = (int32_t *)(base + o_len);
    if (nc > 0) {
        int64_t eb = n_users < 4096 ? n_users : 4096;
        expand_users_kernel<<<dim3((unsigned)eb), dim3(64), 0, h->stream>>>(d_ptr, d_users, n_users, d_pu);
        GORSE_HIP_CHECK(hipGetLastError());
        GORSE_TRY(mf_score_device(h, d_pu, d_cand, nc, d_score));
    }
    mf_rank_kernel<<<dim3((unsigned)ceil_div(n_users, 64)), dim3(64), 0, h->stream>>>(d_ptr, d_cand, d_score, n_users, topk,
                                                                                     d_hv, d_hw, d_rank, d_len);
    GORSE_HIP_CHECK(hipGetLastError());
    GORSE_HIP_CHECK(hipMemcpyAsync(rank_out, d_rank, (size_t)n_users * topk * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(rank_len, d_len, (size_t)n_users * 4, hipMemcpyDeviceToHost, h->stream));
    return GORSE_OK;
}

// ---- multi-GPU exchange ---------------------------------------------------------------------
extern "C" int32_t gorse_mf_item_sync_mark(gorse_mf *h) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_TRY(h->Qsync.ensure((size_t)h->I * h->d));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->Qsync.p, h->Q.p, (size_t)h->I * h->d * sizeof(float), hipMemcpyDeviceToDevice,
                                   h->stream));
    return GORSE_OK;
}

// the two halves of the item-factor exchange, enqueued on the handle's stream (no host synchronisation): comm.hip puts the
// RCCL all-reduce between them on the same stream
int32_t gorse::mf_delta_export_async(gorse_mf *h, float *dst) {
    if (!h->Qsync.p) return fail(GORSE_ERR_INVALID, "gorse_mf_item_sync_mark was never called");
    const int64_t n = h->I * (int64_t)h->d, n4 = (((uintptr_t)dst & 15) == 0) ? n / 4 : 0;
    if (n4 > 0) {
        int64_t blocks = std::min<int64_t>(ceil_div(n4, 256), 2048);
        delta_export_kernel<<<dim3((unsigned)blocks), dim3(256), 0, h->stream>>>((const float4 *)h->Q.p,
                                                                                 (const float4 *)h->Qsync.p, (float4 *)dst, n4);
    }
    if (n4 * 4 < n)
        delta_export_tail<<<dim3((unsigned)ceil_div(n - n4 * 4, 256)), dim3(256), 0, h->stream>>>(h->Q.p, h->Qsync.p, dst,
                                                                                                 n4 * 4, n);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

int32_t gorse::mf_delta_import_async(gorse_mf *h, const float *src) {
    if (!h->Qsync.p) return fail(GORSE_ERR_INVALID, "gorse_mf_item_sync_mark was never called");
    const int64_t n = h->I * (int64_t)h->d, n4 = (((uintptr_t)src & 15) == 0) ? n / 4 : 0;
    if (n4 > 0) {
        int64_t blocks = std::min<int64_t>(ceil_div(n4, 256), 2048);
        delta_import_kernel<<<dim3((unsigned)blocks), dim3(256), 0, h->stream>>>((float4 *)h->Q.p, (float4 *)h->Qsync.p,
                                                                                 (const float4 *)src, n4);
    }
    if (n4 * 4 < n)
        delta_import_tail<<<dim3((unsigned)ceil_div(n - n4 * 4, 256)), dim3(256), 0, h->stream>>>(h->Q.p, h->Qsync.p, src,
                                                                                                 n4 * 4, n);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_item_delta_export(gorse_mf *h, float *dst) {
    if (!h || !dst) return fail(GORSE_ERR_INVALID, "NULL argument");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_delta_export_async(h, dst));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // the caller's collective runs on its own stream
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_item_delta_import(gorse_mf *h, const float *src) {
    if (!h || !src) return fail(GORSE_ERR_INVALID, "NULL argument");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_delta_import_async(h, src));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" int32_t gorse_mf_device_ptrs(gorse_mf *h, float **P, float **Q) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (P) *P = h->P.p;
    if (Q) *Q = h->Q.p;
    return GORSE_OK;
}

// ---- stream / measurement -----------------------------------------------------------------------
extern "C" int32_t gorse_mf_synchronize(gorse_mf *h) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    return mf_sync_streams(h);
}
extern "C" int32_t gorse_mf_epoch_throttle(gorse_mf *h, int32_t max_in_flight, const volatile int32_t *cancel) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (max_in_flight < 0) return fail(GORSE_ERR_INVALID, "max_in_flight < 0");
    const bool chain = h->ep_chain;  // (this call issues nothing on the handle's streams: two epochs around it still follow each other)
    GORSE_TRY(h->use());
    h->ep_chain = chain;
    // the epochs are finished in order: wait for the oldest until no more than max_in_flight are left, the cancel flag in hand
    for (;;) {
        GORSE_TRY(mf_epoch_harvest(h, false));
        if (h->ep_seq - h->ep_done <= (uint64_t)max_in_flight) return GORSE_OK;
        if (cancel && *cancel) return fail(GORSE_ERR_CANCELLED, "cancelled");
        if (!cancel) {  // nobody to listen for: block on the oldest epoch
            GORSE_HIP_CHECK(hipEventSynchronize(h->ev_ep_end[h->ep_done % gorse_mf::kEpochRing]));
        } else {
            struct timespec ts = {0, 20000};  // 20 us between two looks at the event and the flag
            nanosleep(&ts, nullptr);
        }
    }
}
extern "C" int32_t gorse_mf_epoch_times(gorse_mf *h, int64_t *epochs, double *total_ms, int64_t *in_flight, int32_t reset) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    const bool chain = h->ep_chain;
    GORSE_TRY(h->use());
    h->ep_chain = chain;
    GORSE_TRY(mf_epoch_harvest(h, false));
    if (epochs) *epochs = h->ep_timed;
    if (total_ms) *total_ms = h->ep_ms;
    if (in_flight) *in_flight = (int64_t)(h->ep_seq - h->ep_done);
    if (reset) {
        h->ep_ms = 0.0;
        h->ep_timed = 0;
        h->ep_untimed = 0;
    }
    return GORSE_OK;
}
extern "C" int32_t gorse_mf_set_profiling(gorse_mf *h, int32_t on) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    h->prof.resolve();
    h->prof.on = on != 0;
    return GORSE_OK;
}
extern "C" int32_t gorse_mf_get_profile(gorse_mf *h, int32_t cls, int64_t *launches, double *total_ms) {
    if (!h || cls < 0 || cls >= GORSE_PROF_NCLASSES) return fail(GORSE_ERR_INVALID, "bad kernel class");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    h->prof.resolve();
    if (launches) *launches = h->prof.launches[cls];
    if (total_ms) *total_ms = h->prof.ms[cls];
    return GORSE_OK;
}
extern "C" int32_t gorse_mf_reset_profile(gorse_mf *h) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    h->prof.reset();
    return GORSE_OK;
}
