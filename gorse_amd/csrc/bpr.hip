// bpr.hip -- BPR training step on gfx950.  Reference: model/cf/model.go:446-494.
//
// Two kernels per chunk of samples:
//   bpr_sample_kernel : model.go:449-468 -- one thread per sample draws (u, i, j) from a
//                       counter-based Philox stream; runs AHEAD on its own stream.
//   bpr_update_kernel : model.go:469-488 -- one 16-lane group per sample gathers the three
//                       factor rows (64-byte contiguous segments per load), two AVX512-order dot
//                       products by DPP rotate-adds, sigmoid, and three scaled row updates.
// HBM-bound: algorithmic bytes per sample = 6*d*4 (three rows read + three written) + 12 (indices).
#include <algorithm>

#include "mf_internal.hpp"

using namespace gorse;

namespace {

int g_variant = 0;  // probe-only ablation bits (1 plain loads, 2/4/8 skip P/Qi/Qj writes)
constexpr int MODE_ATOMIC = GORSE_BPR_HOGWILD_ATOMIC;
constexpr int MODE_EXACT = GORSE_BPR_SEQUENTIAL;
constexpr int MODE_RACY = GORSE_BPR_HOGWILD_RACY;

// ---- sampling ------------------------------------------------------------------------------
__device__ __forceinline__ bool row_contains(const int32_t *__restrict__ row, int64_t n, int32_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (row[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < n && row[lo] == x;
}

__global__ __launch_bounds__(256) void bpr_sample_kernel(int32_t U, int32_t I, const int64_t *__restrict__ uptr,
                                                         const int32_t *__restrict__ uidx,
                                                         const int32_t *__restrict__ usorted, uint64_t seed,
                                                         uint64_t epoch, int64_t sample_base, int64_t n,
                                                         int32_t *__restrict__ us, int32_t *__restrict__ is,
                                                         int32_t *__restrict__ js, int32_t *__restrict__ fail_count) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        Philox g;
        g.init(seed, epoch, (uint64_t)(sample_base + s));
        int32_t u = -1;
        int64_t beg = 0, cnt = 0;
        for (int t = 0; t < kMaxDraws; t++) {
            int32_t cu = g.int31n(U);
            beg = uptr[cu];
            cnt = uptr[cu + 1] - beg;
            if (cnt > 0) {
                u = cu;
                break;
            }
        }
        int32_t pi = -1, nj = -1;
        if (u >= 0) {
            pi = uidx[beg + g.int31n((int32_t)cnt)];
            for (int t = 0; t < kMaxDraws; t++) {
                int32_t c = g.int31n(I);
                if (!row_contains(usorted + beg, cnt, c)) {
                    nj = c;
                    break;
                }
            }
        }
        if (u < 0 || nj < 0) {
            atomicAdd(fail_count, 1);
            u = pi = nj = -1;
        }
        us[s] = u;
        is[s] = pi;
        js[s] = nj;
    }
}

// ---- memory access flavours ------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float load_row(const float *p, int variant = 0) {
    if (MODE == MODE_EXACT || (variant & 1))
        return *p;
    else  // agent-scope load: served by L2 (never stale for written-back data), bypasses the CU's L1
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// row[e] <- fma(t, lr, snapshot) in the flavour of MODE
template <int MODE>
__device__ __forceinline__ void apply(float *p, float snap, float t, float lr, bool fused) {
    if constexpr (MODE == MODE_ATOMIC) {
        (void)snap;
        (void)fused;
        __hip_atomic_fetch_add(p, t * lr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        float r = fused ? fmaf(t, lr, snap) : t * lr + snap;
        if constexpr (MODE == MODE_EXACT)
            *p = r;
        else  // write-through (sc1) store: visible to the other XCDs' L2s, like a CPU store
            __hip_atomic_store(p, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ float bpr_exp(float x, int exp_mode) {
    return exp_mode == 1 ? exp_restated(x) : (exp_mode == 2 ? __expf(x) : expf(x));
}

// one element of the three updates of model.go:473-488 (operation order of SURVEY.md A2)
template <int MODE>
__device__ __forceinline__ void update_elem(float *pu, float *qi, float *qj, int e, float p, float a, float b, float grad,
                                            float nreg, float lr, bool fused, bool same_item, int variant = 0) {
    float t1 = p * grad;
    t1 = fused ? fmaf(a, nreg, t1) : a * nreg + t1;
    float t2 = p * (-grad);
    t2 = fused ? fmaf(b, nreg, t2) : b * nreg + t2;
    float t3 = a - b;
    t3 = t3 * grad;
    t3 = fused ? fmaf(p, nreg, t3) : p * nreg + t3;
    if constexpr (MODE == MODE_EXACT) {
        float qi_new = fused ? fmaf(t1, lr, a) : t1 * lr + a;
        qi[e] = qi_new;
        // i == j cannot come out of the sampler; a hand-made stream applies both updates in order
        float base = same_item ? qi_new : b;
        qj[e] = fused ? fmaf(t2, lr, base) : t2 * lr + base;
        pu[e] = fused ? fmaf(t3, lr, p) : t3 * lr + p;
    } else {
        if (!(variant & 4)) apply<MODE>(qi + e, a, t1, lr, fused);
        if (!(variant & 8)) apply<MODE>(qj + e, b, t2, lr, fused);
        if (!(variant & 2)) apply<MODE>(pu + e, p, t3, lr, fused);
    }
}

template <int NC, int MODE>
__global__ __launch_bounds__(kBlock) void bpr_update_kernel(float *P, float *Q, const int32_t *__restrict__ us,
                                                            const int32_t *__restrict__ is,
                                                            const int32_t *__restrict__ js,
                                                            const int32_t *__restrict__ order, int64_t begin,
                                                            int64_t end, int d, float lr, float reg, int exp_mode,
                                                            double *loss, int variant) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & (kGroup - 1);
    const int gib = threadIdx.x / kGroup;
    const int64_t group = (int64_t)blockIdx.x * kGroupsPerBlock + gib;
    const int64_t ngroups = (int64_t)gridDim.x * kGroupsPerBlock;
    const VecShape vs(d);
    const float nreg = -reg;
    double my_loss = 0.0;
    for (int64_t s = begin + group; s < end; s += ngroups) {
        const int64_t t = order ? (int64_t)order[s] : s;
        const int u = us[t], i = is[t], j = js[t];
        if ((u | i | j) < 0) continue;
        float *pu = P + (int64_t)u * d, *qi = Q + (int64_t)i * d, *qj = Q + (int64_t)j * d;
        if constexpr (NC > 0) {
            float p[NC], a[NC], b[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                p[c] = load_row<MODE>(pu + 16 * c + lane, variant);
                a[c] = load_row<MODE>(qi + 16 * c + lane, variant);
                b[c] = load_row<MODE>(qj + 16 * c + lane, variant);
            }
            const float diff = dot512_regs<NC>(p, a) - dot512_regs<NC>(p, b);
            const float ex = bpr_exp(-diff, exp_mode);
            const float grad = ex / (1.0f + ex);
            if (loss && lane == 0) my_loss += (double)log1pf(ex);
#pragma unroll
            for (int c = 0; c < NC; c++)
                update_elem<MODE>(pu, qi, qj, 16 * c + lane, p[c], a[c], b[c], grad, nreg, lr, true, i == j, variant);
        } else {
            float *sp = smem + (size_t)gib * 3 * d, *sa = sp + d, *sb = sa + d;
            for (int e = lane; e < d; e += kGroup) {
                sp[e] = load_row<MODE>(pu + e);
                sa[e] = load_row<MODE>(qi + e);
                sb[e] = load_row<MODE>(qj + e);
            }
            __builtin_amdgcn_wave_barrier();
            const float diff = dot512_lds(sp, sa, vs, lane) - dot512_lds(sp, sb, vs, lane);
            const float ex = bpr_exp(-diff, exp_mode);
            const float grad = ex / (1.0f + ex);
            if (loss && lane == 0) my_loss += (double)log1pf(ex);
            for (int e = lane; e < d; e += kGroup)
                update_elem<MODE>(pu, qi, qj, e, sp[e], sa[e], sb[e], grad, nreg, lr, !vs.unfused(e), i == j);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (loss && lane == 0 && my_loss != 0.0) atomicAdd(loss, my_loss);
}

template <int MODE>
int32_t launch_update_mode(gorse_mf *h, const int32_t *us, const int32_t *is, const int32_t *js, const int32_t *order,
                           int64_t begin, int64_t end, float lr, float reg, int exp_mode, double *loss,
                           hipStream_t st) {
    const int64_t n = end - begin;
    if (n <= 0) return GORSE_OK;
    const int d = h->d;
    int64_t blocks = ceil_div(n, kGroupsPerBlock);
    const int64_t cap = 256 * 16;  // 16 workgroups of 4 waves per CU: grid-stride beyond that
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks), block(kBlock);
#define LAUNCH(NC, SH)                                                                                               \
    bpr_update_kernel<NC, MODE><<<grid, block, SH, st>>>(h->P.p, h->Q.p, us, is, js, order, begin, end, d, lr, reg, \
                                                         exp_mode, loss, g_variant)
    if (d == 16)
        LAUNCH(1, 0);
    else if (d == 32)
        LAUNCH(2, 0);
    else if (d == 64)
        LAUNCH(4, 0);
    else if (d == 128)
        LAUNCH(8, 0);
    else
        LAUNCH(0, (size_t)kGroupsPerBlock * 3 * d * sizeof(float));
#undef LAUNCH
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

int32_t launch_update(gorse_mf *h, int mode, const int32_t *us, const int32_t *is, const int32_t *js,
                      const int32_t *order, int64_t begin, int64_t end, float lr, float reg, int exp_mode, double *loss,
                      hipStream_t st) {
    switch (mode) {
    case MODE_ATOMIC:
        return launch_update_mode<MODE_ATOMIC>(h, us, is, js, order, begin, end, lr, reg, exp_mode, loss, st);
    case MODE_EXACT:
        return launch_update_mode<MODE_EXACT>(h, us, is, js, order, begin, end, lr, reg, exp_mode, loss, st);
    case MODE_RACY:
        return launch_update_mode<MODE_RACY>(h, us, is, js, order, begin, end, lr, reg, exp_mode, loss, st);
    }
    return fail(GORSE_ERR_INVALID, "unknown BPR mode %d", mode);
}

int32_t launch_sampler(gorse_mf *h, uint64_t seed, uint64_t epoch, int64_t base, int64_t n, int32_t *trip, size_t cap,
                       hipStream_t st) {
    if (n <= 0) return GORSE_OK;
    int64_t blocks = std::min<int64_t>(ceil_div(n, 256), 256 * 8);
    bpr_sample_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((int32_t)h->U, (int32_t)h->I, h->uptr.p, h->uidx.p,
                                                                    h->uidx_sorted.p, seed, epoch, base, n, trip,
                                                                    trip + cap, trip + 2 * cap, h->fail_count.p);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

int32_t ensure_trip(gorse_mf *h, int64_t want) {
    size_t cap = (size_t)std::min<int64_t>(std::max<int64_t>(want, 1), (int64_t)1 << 22);
    if (cap <= h->trip_cap) return GORSE_OK;
    GORSE_TRY(mf_sync_streams(h));
    for (int b = 0; b < 2; b++) GORSE_TRY(h->trip[b].alloc(cap * 3));
    h->trip_cap = cap;
    return GORSE_OK;
}

// Dependency levels: level(s) = 1 + max(level of the last earlier sample touching P[u], Q[i], Q[j]).
// Samples of one level touch pairwise disjoint rows, and every pair of conflicting samples keeps its
// stream order across levels, so running level after level reproduces the sequential (Jobs = 1,
// common/parallel/parallel.go:34-43) result exactly.
void build_levels(int64_t U, int64_t I, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n,
                  std::vector<int32_t> &order, std::vector<int64_t> &level_ptr) {
    std::vector<int32_t> lastP((size_t)U, 0), lastQ((size_t)I, 0), lvl((size_t)n, 0);
    int32_t maxl = 0;
    for (int64_t s = 0; s < n; s++) {
        if ((u[s] | i[s] | j[s]) < 0) continue;  // skipped sample: level 0 = never launched
        int32_t l = std::max(lastP[u[s]], std::max(lastQ[i[s]], lastQ[j[s]])) + 1;
        lvl[s] = l;
        lastP[u[s]] = l;
        lastQ[i[s]] = l;
        lastQ[j[s]] = l;
        maxl = std::max(maxl, l);
    }
    level_ptr.assign((size_t)maxl + 2, 0);
    for (int64_t s = 0; s < n; s++) level_ptr[lvl[s] + 1]++;
    for (int32_t l = 0; l <= maxl; l++) level_ptr[l + 1] += level_ptr[l];
    order.resize((size_t)n);
    std::vector<int64_t> cur(level_ptr.begin(), level_ptr.end() - 1);
    for (int64_t s = 0; s < n; s++) order[cur[lvl[s]]++] = (int32_t)s;
}

// sequential schedule over device-resident triplets (us/is/js) whose host copy is (hu, hi, hj)
int32_t run_sequential(gorse_mf *h, const int32_t *d_us, const int32_t *d_is, const int32_t *d_js, const int32_t *hu,
                       const int32_t *hi, const int32_t *hj, int64_t n, float lr, float reg, int exp_mode,
                       const volatile int32_t *cancel, double *d_loss) {
    std::vector<int32_t> order;
    std::vector<int64_t> level_ptr;
    build_levels(h->U, h->I, hu, hi, hj, n, order, level_ptr);
    GORSE_TRY(h->order.ensure((size_t)n));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->order.p, order.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    const size_t nlev = level_ptr.size() - 1;
    for (size_t l = 1; l < nlev; l++) {  // level 0 holds skipped samples
        if (cancel && *cancel && (l & 255) == 0) {
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
            return fail(GORSE_ERR_CANCELLED, "cancelled");
        }
        GORSE_TRY(launch_update(h, MODE_EXACT, d_us, d_is, d_js, h->order.p, level_ptr[l], level_ptr[l + 1], lr, reg,
                                exp_mode, d_loss, h->stream));
    }
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // `order` (host vector) was copied asynchronously
    return GORSE_OK;
}

int g_exp_mode_exact = 0;  // exp flavour of the sequential schedule; tests flip it to 1 for bit parity

int32_t check_mode(int mode) {
    if (mode != MODE_ATOMIC && mode != MODE_EXACT && mode != MODE_RACY)
        return fail(GORSE_ERR_INVALID, "unknown BPR mode %d", mode);
    return GORSE_OK;
}

int32_t epoch_impl(gorse_mf *h, int64_t n_samples, float lr, float reg, uint64_t seed, uint64_t epoch, int64_t base,
                   int mode, const volatile int32_t *cancel, double *loss_out, bool sync) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (n_samples < 0) return fail(GORSE_ERR_INVALID, "n_samples < 0");
    GORSE_TRY(check_mode(mode));
    GORSE_TRY(h->use());
    if (n_samples == 0) {
        if (loss_out) *loss_out = 0;
        return GORSE_OK;
    }
    GORSE_TRY(ensure_trip(h, n_samples));
    double *d_loss = loss_out ? h->loss.p : nullptr;
    if (d_loss) GORSE_HIP_CHECK(hipMemsetAsync(d_loss, 0, sizeof(double), h->stream));
    const int64_t cap = (int64_t)h->trip_cap;
    if (mode == MODE_EXACT) {
        std::vector<int32_t> host((size_t)cap * 3);
        for (int64_t s0 = 0; s0 < n_samples; s0 += cap) {
            const int64_t m = std::min(cap, n_samples - s0);
            int32_t *tb = h->trip[0].p;
            GORSE_TRY(launch_sampler(h, seed, epoch, base + s0, m, tb, (size_t)cap, h->stream));
            GORSE_HIP_CHECK(hipMemcpyAsync(host.data(), tb, (size_t)cap * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
            GORSE_TRY(run_sequential(h, tb, tb + cap, tb + 2 * cap, host.data(), host.data() + cap, host.data() + 2 * cap, m,
                                     lr, reg, g_exp_mode_exact, cancel, d_loss));
        }
    } else {
        // two-stream pipeline: stream2 samples chunk c+1 while stream applies chunk c
        int64_t c = 0;
        for (int64_t s0 = 0; s0 < n_samples; s0 += cap, c++) {
            const int b = (int)(c & 1);
            const int64_t m = std::min(cap, n_samples - s0);
            if (cancel && *cancel) {
                GORSE_TRY(mf_sync_streams(h));
                return fail(GORSE_ERR_CANCELLED, "cancelled");
            }
            int32_t *tb = h->trip[b].p;
            if (c >= 2) GORSE_HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_consumed[b], 0));
            int tok = h->prof.begin(GORSE_PROF_BPR_SAMPLE, h->stream2);
            GORSE_TRY(launch_sampler(h, seed, epoch, base + s0, m, tb, (size_t)cap, h->stream2));
            h->prof.end(tok, h->stream2);
            GORSE_HIP_CHECK(hipEventRecord(h->ev_sampled[b], h->stream2));
            GORSE_HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_sampled[b], 0));
            tok = h->prof.begin(GORSE_PROF_BPR_UPDATE, h->stream);
            GORSE_TRY(launch_update(h, mode, tb, tb + cap, tb + 2 * cap, nullptr, 0, m, lr, reg, 0, d_loss, h->stream));
            h->prof.end(tok, h->stream);
            GORSE_HIP_CHECK(hipEventRecord(h->ev_consumed[b], h->stream));
            if (cancel && (c & 7) == 7) GORSE_TRY(mf_sync_streams(h));
        }
        // the next call's sampler must not overwrite a buffer still being applied
        GORSE_HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_consumed[0], 0));
        if (c >= 2) GORSE_HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_consumed[1], 0));
    }
    if (sync || loss_out) {
        if (loss_out)
            GORSE_HIP_CHECK(hipMemcpyAsync(loss_out, h->loss.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        GORSE_TRY(mf_sync_streams(h));
    }
    return GORSE_OK;
}

}  // namespace

extern "C" void gorse_hip_test_set_exact_exp(int32_t mode) { g_exp_mode_exact = mode; }
extern "C" void gorse_hip_test_set_variant(int32_t v) { g_variant = v; }

extern "C" int32_t gorse_bpr_epoch(gorse_mf *h, int64_t n_samples, float lr, float reg, uint64_t seed, uint64_t epoch,
                                   int64_t sample_base, int32_t mode, const volatile int32_t *cancel, double *loss_out) {
    return epoch_impl(h, n_samples, lr, reg, seed, epoch, sample_base, mode, cancel, loss_out, true);
}

extern "C" int32_t gorse_bpr_epoch_enqueue(gorse_mf *h, int64_t n_samples, float lr, float reg, uint64_t seed,
                                           uint64_t epoch, int64_t sample_base, int32_t mode) {
    if (mode == MODE_EXACT) return fail(GORSE_ERR_INVALID, "the sequential schedule cannot be enqueued asynchronously");
    return epoch_impl(h, n_samples, lr, reg, seed, epoch, sample_base, mode, nullptr, nullptr, false);
}

extern "C" int32_t gorse_bpr_sample_triplets(gorse_mf *h, int64_t n, uint64_t seed, uint64_t epoch, int64_t sample_base,
                                             int32_t *u, int32_t *i, int32_t *j) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (n < 0 || (n > 0 && (!u || !i || !j))) return fail(GORSE_ERR_INVALID, "bad arguments");
    if (n == 0) return GORSE_OK;
    GORSE_TRY(h->use());
    GORSE_TRY(ensure_trip(h, n));
    GORSE_TRY(mf_sync_streams(h));
    const int64_t cap = (int64_t)h->trip_cap;
    for (int64_t s0 = 0; s0 < n; s0 += cap) {
        const int64_t m = std::min(cap, n - s0);
        int32_t *tb = h->trip[0].p;
        GORSE_TRY(launch_sampler(h, seed, epoch, sample_base + s0, m, tb, (size_t)cap, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(u + s0, tb, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(i + s0, tb + cap, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(j + s0, tb + 2 * cap, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    return GORSE_OK;
}

extern "C" int32_t gorse_bpr_apply_triplets(gorse_mf *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n,
                                            float lr, float reg, int32_t mode) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (n < 0 || (n > 0 && (!u || !i || !j))) return fail(GORSE_ERR_INVALID, "bad arguments");
    GORSE_TRY(check_mode(mode));
    if (n == 0) return GORSE_OK;
    for (int64_t s = 0; s < n; s++)
        if (u[s] >= h->U || i[s] >= h->I || j[s] >= h->I)
            return fail(GORSE_ERR_RANGE, "triplet %lld (%d,%d,%d) out of range", (long long)s, u[s], i[s], j[s]);
    GORSE_TRY(h->use());
    GORSE_TRY(ensure_trip(h, n));
    GORSE_TRY(mf_sync_streams(h));
    const int64_t cap = (int64_t)h->trip_cap;
    for (int64_t s0 = 0; s0 < n; s0 += cap) {
        const int64_t m = std::min(cap, n - s0);
        int32_t *tb = h->trip[0].p;
        GORSE_HIP_CHECK(hipMemcpyAsync(tb, u + s0, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(tb + cap, i + s0, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(tb + 2 * cap, j + s0, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
        if (mode == MODE_EXACT) {
            GORSE_TRY(run_sequential(h, tb, tb + cap, tb + 2 * cap, u + s0, i + s0, j + s0, m, lr, reg, g_exp_mode_exact,
                                     nullptr, nullptr));
        } else {
            GORSE_TRY(launch_update(h, mode, tb, tb + cap, tb + 2 * cap, nullptr, 0, m, lr, reg, 0, nullptr, h->stream));
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    }
    return GORSE_OK;
}
