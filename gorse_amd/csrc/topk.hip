// topk.hip -- exact nearest-neighbour search (ann.Index / ann.Bruteforce) on gfx950.
// Reference: common/ann/ann.go:21-25, common/ann/bruteforce.go:24-83, common/heap/pq.go.
//
// Path A (this file, always exact, any d / dtype / metric; the arbiter for every query path B of
// topk_mfma.hip cannot decide): score one block of queries against
// all N stored vectors in the reference's own arithmetic order (one 16-lane group per pair, the
// query block resident in LDS), then run the reference's heap selection.  The selection is the
// literal container/heap procedure (goheap.hpp), so ties come out exactly as in Go.
#include <algorithm>
#include <cmath>
#include <limits>

#include "goheap.hpp"
#include "topk_internal.hpp"

using namespace gorse;

namespace {

__global__ void expand_bf16_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = __uint_as_float((uint32_t)in[t] << 16);
}

// norm2[i] = floats.Dot(x_i, x_i)
__global__ __launch_bounds__(kBlock) void norm2_kernel(const float *__restrict__ X, int64_t n, int d,
                                                       float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & (kGroup - 1), gib = threadIdx.x / kGroup;
    const VecShape vs(d);
    const int gpb = blockDim.x / kGroup;  // groups per workgroup: 16, fewer when d is large (scan_groups)
    float *sa = smem + (size_t)gib * d;
    for (int64_t t = (int64_t)blockIdx.x * gpb + gib; t < n; t += (int64_t)gridDim.x * gpb) {
        for (int e = lane; e < d; e += kGroup) sa[e] = X[t * d + e];
        __builtin_amdgcn_wave_barrier();
        float r = dot512_lds(sa, sa, vs, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) out[t] = r;
    }
}

// dist[qb][i] for one query (blockIdx.y) against candidates i; the query row sits in LDS
__global__ __launch_bounds__(kBlock) void dist_kernel(const float *__restrict__ X, const float *__restrict__ norm2,
                                                      const float *__restrict__ Qv, const float *__restrict__ qnorm2,
                                                      int64_t N, int d, int metric, float *__restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & (kGroup - 1), gib = threadIdx.x / kGroup;
    const VecShape vs(d);
    float *sq = smem;                             // d
    float *sx = smem + d + (size_t)gib * d;       // per-group candidate row
    const int64_t q = blockIdx.y;
    for (int e = threadIdx.x; e < d; e += blockDim.x) sq[e] = Qv[q * d + e];
    __syncthreads();
    const float qq = metric == GORSE_METRIC_COSINE ? qnorm2[q] : 0.0f;
    const int gpb = blockDim.x / kGroup;  // groups per workgroup: 16, fewer when d is large (scan_groups)
    for (int64_t i = (int64_t)blockIdx.x * gpb + gib; i < N; i += (int64_t)gridDim.x * gpb) {
        for (int e = lane; e < d; e += kGroup) sx[e] = X[i * d + e];
        __builtin_amdgcn_wave_barrier();
        float r;
        if (metric == GORSE_METRIC_EUCLIDEAN || metric == kMetricEuclidBf16) {
            r = euclid_any_lds(metric, sq, sx, vs, lane);
        } else {
            float ab = dot512_lds(sq, sx, vs, lane);
            if (metric == GORSE_METRIC_NEG_DOT)
                r = -ab;
            else
                r = 1.0f - ab / (sqrtf(qq) * sqrtf(norm2[i]));
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) dist[q * N + i] = r;
    }
}

__global__ void gather_rows_kernel(const float *__restrict__ X, const float *__restrict__ norm2,
                                   const int64_t *__restrict__ qidx, int d, float *__restrict__ Qv,
                                   float *__restrict__ qn) {
    const int64_t q = blockIdx.x;
    const int64_t src = qidx[q];
    for (int e = threadIdx.x; e < d; e += blockDim.x) Qv[q * d + e] = X[src * d + e];
    if (threadIdx.x == 0 && norm2) qn[q] = norm2[src];
}

// Bruteforce selection (bruteforce.go:45-62 / 67-82), one thread per query, literal heaps.
__global__ void select_kernel(const float *__restrict__ dist, const int64_t *__restrict__ qidx, int64_t nq, int64_t N,
                              int k, int prune0, int32_t *heap_v, float *heap_w, int32_t *__restrict__ out_idx,
                              float *__restrict__ out_dist, int32_t *__restrict__ out_cnt,
                              const uint8_t *__restrict__ mask, const int32_t *__restrict__ only) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    if (only && !only[q]) return;  // answered by select_fast_kernel
    const int64_t skip = qidx ? qidx[q] : -1;
    int32_t *hv = heap_v + q * 2 * (k + 1);
    float *hw = heap_w + q * 2 * (k + 1);
    GoHeap<true> mx(hv, hw);
    const float *dq = dist + q * N;
    for (int64_t i = 0; i < N; i++) {
        if (i == skip || (mask && !mask[i])) continue;  // a masked row is never pushed: Bruteforce over the admissible rows
        mx.push((int32_t)i, dq[i]);
        if (mx.n > k) mx.pop();
    }
    GoHeap<false> mn(hv + (k + 1), hw + (k + 1));  // Reverse(): re-push in array order
    for (int t = 0; t < mx.n; t++) mn.push(hv[t], hw[t]);
    int cnt = 0;
    while (mn.n > 0) {
        mn.pop();
        const int32_t v = mn.v[mn.n];
        const float w = mn.w[mn.n];
        if (!prune0 || w > 0) {
            out_idx[q * k + cnt] = v;
            out_dist[q * k + cnt] = w;
            cnt++;
        }
    }
    out_cnt[q] = cnt;
    for (int t = cnt; t < k; t++) {
        out_idx[q * k + t] = -1;
        out_dist[q * k + t] = __int_as_float(0x7f800000);
    }
}

// The same selection for the queries whose answer does not depend on the heap's history: when the k + 1 smallest distances of
// a query are pairwise distinct (and none is NaN), Bruteforce's bounded max-heap ends with the k smallest and its Reverse()
// pops them in ascending order -- whatever was pushed in between.  One workgroup per query finds them in two passes over the
// query's row of the distance slab: (1) every thread keeps the 4 smallest keys of its stride; the (k+1)-th smallest of those
// 4096 keys bounds the true (k+1)-th smallest from above; (2) the rows up to that bound (k + 1 of them, give or take) are
// collected, sorted and checked for equal neighbours.  Anything else -- ties among the first k + 1, a NaN, more rows at the
// bound than the buffer holds -- is flagged and left to the literal select_kernel.  (The literal kernel alone, one thread per
// query, answered ONE query against 1,000,000 vectors in 3.3 s: profiles/r02_u_probe_query_latency.txt.)
constexpr int kSelThreads = 1024, kSelKeep = 4, kSelCap = 2048;
__device__ inline uint32_t ascending_key(float w) {  // smaller distance <=> smaller key; -0 = +0
    uint32_t u = __float_as_uint(w);
    if ((u << 1) == 0) u = 0;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key_distance(uint32_t key) {
    const uint32_t u = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
    return __uint_as_float(u);
}
__global__ __launch_bounds__(kSelThreads) void select_fast_kernel(const float *__restrict__ dist, const int64_t *__restrict__ qidx,
                                                                  int64_t N, int k, int prune0, int32_t *__restrict__ out_idx,
                                                                  float *__restrict__ out_dist, int32_t *__restrict__ out_cnt,
                                                                  const uint8_t *__restrict__ mask, int32_t *__restrict__ literal) {
    __shared__ unsigned long long s_sel[kSelCap];
    __shared__ int s_count, s_adm, s_bad;
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t skip = qidx ? qidx[q] : -1;
    const float *dq = dist + q * N;
    if (tid == 0) s_count = 0, s_adm = 0, s_bad = 0;
    __syncthreads();
    uint32_t best[kSelKeep];
#pragma unroll
    for (int j = 0; j < kSelKeep; j++) best[j] = 0xFFFFFFFFu;
    int adm = 0;
    bool bad = false;
    for (int64_t i = tid; i < N; i += kSelThreads) {
        if (i == skip || (mask && !mask[i])) continue;
        const float w = dq[i];
        bad = bad || (__float_as_uint(w) & 0x7FFFFFFFu) > 0x7F800000u;  // NaN: the heap's comparisons are all false
        uint32_t key = ascending_key(w);
        adm++;
#pragma unroll
        for (int j = 0; j < kSelKeep; j++) {  // insertion into the ascending four
            const uint32_t lo = min(best[j], key);
            key = max(best[j], key);
            best[j] = lo;
        }
    }
    atomicAdd(&s_adm, adm);
    if (bad) s_bad = 1;
    __syncthreads();
    const int A = s_adm, need = min(k + 1, A);
    if (s_bad || k + 1 > kSelCap / 2) {
        if (tid == 0) literal[q] = 1;
        return;
    }
    if (tid == 0) literal[q] = 0;
    // the need-th smallest of the kept keys, bit by bit: the smallest t with count(keys <= t) >= need
    uint32_t tau = 0;
    if (need > 0) {
        for (int b = 31; b >= 0; --b) {
            const uint32_t t = tau | (((uint32_t)1 << b) - 1);
            int c = 0;
#pragma unroll
            for (int j = 0; j < kSelKeep; j++) c += best[j] <= t && best[j] != 0xFFFFFFFFu;
            // (a real key is never 0xFFFFFFFF: that would be a NaN's)
            c = __syncthreads_count(c > 0) + __syncthreads_count(c > 1) + __syncthreads_count(c > 2) + __syncthreads_count(c > 3);
            if (c < need) tau |= (uint32_t)1 << b;
        }
        for (int64_t i = tid; i < N; i += kSelThreads) {
            if (i == skip || (mask && !mask[i])) continue;
            const uint32_t key = ascending_key(dq[i]);
            if (key <= tau) {
                const int slot = atomicAdd(&s_count, 1);
                if (slot < kSelCap) s_sel[slot] = ((unsigned long long)key << 32) | (unsigned long long)(uint32_t)i;
            }
        }
    }
    __syncthreads();
    const int cnt = s_count;
    if (cnt > kSelCap) {  // a plateau of equal distances at the bound
        if (tid == 0) literal[q] = 1;
        return;
    }
    for (int i = cnt + tid; i < kSelCap; i += kSelThreads) s_sel[i] = ~0ull;
    __syncthreads();
    for (int size = 2; size <= kSelCap; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < kSelCap; i += kSelThreads) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool up = (i & size) == 0;
                    const unsigned long long x = s_sel[i], y = s_sel[j];
                    if (up ? x > y : x < y) s_sel[i] = y, s_sel[j] = x;
                }
            }
            __syncthreads();
        }
    bool tie = false;
    for (int i = tid; i + 1 < need; i += kSelThreads) tie = tie || (uint32_t)(s_sel[i] >> 32) == (uint32_t)(s_sel[i + 1] >> 32);
    if (__syncthreads_or(tie)) {
        if (tid == 0) literal[q] = 1;
        return;
    }
    if (tid == 0) {
        const int take = min(k, A);
        int n = 0;
        for (int i = 0; i < take; i++) {
            const float w = key_distance((uint32_t)(s_sel[i] >> 32));
            if (!prune0 || w > 0) {
                out_idx[q * k + n] = (int32_t)(uint32_t)s_sel[i];
                out_dist[q * k + n] = w;
                n++;
            }
        }
        out_cnt[q] = n;
        for (int t = n; t < k; t++) {
            out_idx[q * k + t] = -1;
            out_dist[q * k + t] = __int_as_float(0x7f800000);
        }
    }
}

constexpr int kTopkMaxDim = 16384;
// 16-lane groups per workgroup of dist_kernel / norm2_kernel: as many as keep (1 + g) rows of d floats inside 144 KB of LDS
int scan_groups(int d) {
    int g = kGroupsPerBlock;
    while (g > 1 && (size_t)(1 + g) * (size_t)d * sizeof(float) > (size_t)144 * 1024) g >>= 1;
    return g;
}
constexpr int64_t kDistBudget = (int64_t)1 << 28;  // floats in the distance slab (1 GiB)
int g_scan_literal_only = 0;  // test hook: every query of the scan through the literal heap kernel
constexpr int64_t kMinRerouteQueries = (int64_t)1 << 40;  // "enough queries": asks topk_mfma_usable about the index and k only

}  // namespace

namespace gorse {

// queries already on the device in h->qbuf (nq x d, fp32) [+ h->qnorm]; qidx_dev = exclude list or null
int32_t topk_scan_block(gorse_topk *h, int64_t nq, const int64_t *qidx_dev, const int64_t *qidx_host, int k, int prune0,
                        int32_t *idx_out, float *dist_out, int32_t *cnt_out, bool reroute) {
    const int d = h->d;
    GORSE_TRY(h->dist.ensure((size_t)nq * h->N));
    GORSE_TRY(h->heap_v.ensure((size_t)nq * 2 * (k + 1)));
    GORSE_TRY(h->heap_w.ensure((size_t)nq * 2 * (k + 1)));
    GORSE_TRY(h->out_idx.ensure((size_t)nq * k));
    GORSE_TRY(h->out_dist.ensure((size_t)nq * k));
    GORSE_TRY(h->out_cnt.ensure((size_t)nq));
    const int gpb = scan_groups(d);
    int64_t bx = std::min<int64_t>(ceil_div(h->N, gpb), 1024);
    int tok = h->prof.begin(GORSE_PROF_TOPK_SCORE, h->stream);
    const size_t dist_lds = (size_t)(1 + gpb) * d * sizeof(float);
    GORSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&dist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dist_lds));
    dist_kernel<<<dim3((unsigned)bx, (unsigned)nq), dim3(gpb * kGroup), dist_lds, h->stream>>>(h->X.p, h->norm2.p, h->qbuf.p, h->qnorm.p,
                                                                                              h->N, d, h->kernel_metric(), h->dist.p);
    GORSE_HIP_CHECK(hipGetLastError());
    h->prof.end(tok, h->stream);
    tok = h->prof.begin(GORSE_PROF_TOPK_RESCORE, h->stream);
    const uint8_t *mask = h->has_mask ? h->mask.p : nullptr;
    int32_t *literal = nullptr;
    // The queries the fast selection leaves (ties among the k + 1 smallest distances, NaN): the literal heap kernel is one
    // thread per query over all N rows -- 3.3 s for a million rows, however few queries there are -- so where the MFMA path
    // can take them (its history sweep + heap replay answer a tie query in tens of milliseconds) they go there.
    // (not with a mask: the replay counts the rows between two recorded ones, masked rows included, so the MFMA path sends a
    // masked tie query back to the literal kernel anyway -- after a full sweep)
    reroute = reroute && !g_scan_literal_only && !h->has_mask && topk_mfma_usable(h, kMinRerouteQueries, k);
    if (!g_scan_literal_only) {
        GORSE_TRY(h->scan_literal.ensure((size_t)nq));
        literal = h->scan_literal.p;
        select_fast_kernel<<<dim3((unsigned)nq), dim3(kSelThreads), 0, h->stream>>>(h->dist.p, qidx_dev, h->N, k, prune0, h->out_idx.p,
                                                                                 h->out_dist.p, h->out_cnt.p, mask, literal);
        GORSE_HIP_CHECK(hipGetLastError());
    }
    if (!reroute) {
        select_kernel<<<dim3((unsigned)ceil_div(nq, 64)), dim3(64), 0, h->stream>>>(h->dist.p, qidx_dev, nq, h->N, k, prune0,
                                                                                  h->heap_v.p, h->heap_w.p, h->out_idx.p,
                                                                                  h->out_dist.p, h->out_cnt.p, mask, literal);
        GORSE_HIP_CHECK(hipGetLastError());
    }
    h->prof.end(tok, h->stream);
    std::vector<int32_t> flags;
    if (reroute) {
        flags.resize((size_t)nq);
        GORSE_HIP_CHECK(hipMemcpyAsync(flags.data(), literal, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream));
    }
    if (idx_out) GORSE_HIP_CHECK(hipMemcpyAsync(idx_out, h->out_idx.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
    if (dist_out) GORSE_HIP_CHECK(hipMemcpyAsync(dist_out, h->out_dist.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
    if (cnt_out) GORSE_HIP_CHECK(hipMemcpyAsync(cnt_out, h->out_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (!reroute) return GORSE_OK;
    std::vector<int64_t> todo;
    for (int64_t t = 0; t < nq; t++)
        if (flags[(size_t)t]) todo.push_back(t);
    if (todo.empty()) return GORSE_OK;
    const size_t nf = todo.size();
    std::vector<int64_t> ids(nf);
    DevBuf<float> qf;
    if (qidx_host) {
        for (size_t r = 0; r < nf; r++) ids[r] = qidx_host[todo[r]];
    } else {  // their vectors, packed
        GORSE_TRY(qf.alloc(nf * (size_t)d));
        for (size_t r = 0; r < nf; r++)
            GORSE_HIP_CHECK(hipMemcpyAsync(qf.p + r * d, h->qbuf.p + todo[r] * d, (size_t)d * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    std::vector<int32_t> ti(nf * (size_t)k), tc(nf);
    std::vector<float> td(nf * (size_t)k);
    const int64_t keep_fallback = h->n_fallback;
    GORSE_TRY(topk_mfma_search(h, qidx_host ? ids.data() : nullptr, -1, qidx_host ? nullptr : qf.p, (int64_t)nf, k, prune0, ti.data(),
                               td.data(), tc.data()));
    h->n_fallback += keep_fallback;
    for (size_t r = 0; r < nf; r++) {
        const int64_t t = todo[r];
        if (idx_out) memcpy(idx_out + t * k, ti.data() + r * k, (size_t)k * 4);
        if (dist_out) memcpy(dist_out + t * k, td.data() + r * k, (size_t)k * 4);
        if (cnt_out) cnt_out[t] = tc[r];
    }
    return GORSE_OK;
}

int64_t topk_scan_block_queries(const gorse_topk *h) {
    int64_t b = kDistBudget / std::max<int64_t>(h->N, 1);
    return std::max<int64_t>(1, std::min<int64_t>(b, 65535));
}

int32_t topk_compute_norms(gorse_topk *h, const float *V, int64_t n, float *out) {
    const int gpb = scan_groups(h->d);
    int64_t bx = std::min<int64_t>(ceil_div(n, gpb), 4096);
    const size_t lds = (size_t)gpb * h->d * sizeof(float);
    GORSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&norm2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    norm2_kernel<<<dim3((unsigned)bx), dim3(gpb * kGroup), lds, h->stream>>>(V, n, h->d, out);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

}  // namespace gorse

extern "C" int32_t gorse_topk_create(gorse_topk **out, int32_t device, int64_t N, int32_t d, int32_t dtype,
                                     int32_t metric, const void *X) {
    if (!out) return fail(GORSE_ERR_INVALID, "handle pointer is NULL");
    *out = nullptr;
    if (N <= 0 || d <= 0 || !X) return fail(GORSE_ERR_INVALID, "N, d must be positive and X non-NULL");
    if (N > INT32_MAX) return fail(GORSE_ERR_INVALID, "N must fit int32");
    // dist_kernel stages 1 + g rows of d floats per workgroup in LDS, g = 16 groups down to 1 as d grows (scan_groups):
    // the reference's shipped configuration uses embeddings of 1024 dimensions (config.toml: embedding_dimensions), 1536 and
    // 3072 are common
    if (d > kTopkMaxDim) return fail(GORSE_ERR_INVALID, "d %d > %d unsupported (the scan stages two rows of d floats in 160 KB of LDS)", d, kTopkMaxDim);
    if (dtype != GORSE_DTYPE_F32 && dtype != GORSE_DTYPE_BF16) return fail(GORSE_ERR_INVALID, "unknown dtype %d", dtype);
    if (metric < 0 || metric > 3) return fail(GORSE_ERR_INVALID, "unknown metric %d", metric);
    if (metric == GORSE_METRIC_EUCLIDEAN_BF16 && dtype != GORSE_DTYPE_BF16)
        return fail(GORSE_ERR_INVALID, "GORSE_METRIC_EUCLIDEAN_BF16 (bfloats.Euclidean) needs a bf16 index");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GORSE_ERR_NO_DEVICE, "no HIP device visible (libgorse_hip needs an MI355X / gfx950)");
    if (device < 0 || device >= ndev) return fail(GORSE_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    gorse_topk *h = new (std::nothrow) gorse_topk();
    if (!h) return fail(GORSE_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->N = N;
    h->d = d;
    h->dtype = dtype;
    h->metric = metric == GORSE_METRIC_EUCLIDEAN_BF16 ? GORSE_METRIC_EUCLIDEAN : metric;
    h->bf16_order = metric == GORSE_METRIC_EUCLIDEAN_BF16;
    int32_t rc = [&]() -> int32_t {
        GORSE_TRY(h->use());
        GORSE_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        GORSE_TRY(h->X.alloc((size_t)N * d));
        if (dtype == GORSE_DTYPE_BF16) {
            GORSE_TRY(h->Xb.alloc((size_t)N * d));
            GORSE_HIP_CHECK(hipMemcpyAsync(h->Xb.p, X, (size_t)N * d * 2, hipMemcpyHostToDevice, h->stream));
            expand_bf16_kernel<<<dim3(2048), dim3(256), 0, h->stream>>>(h->Xb.p, h->X.p, N * (int64_t)d);
            GORSE_HIP_CHECK(hipGetLastError());
        } else {
            GORSE_HIP_CHECK(hipMemcpyAsync(h->X.p, X, (size_t)N * d * 4, hipMemcpyHostToDevice, h->stream));
        }
        GORSE_TRY(h->norm2.alloc((size_t)N));
        GORSE_TRY(topk_compute_norms(h, h->X.p, N, h->norm2.p));
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        GORSE_TRY(topk_mfma_prepare(h));  // path B operands (dot / cosine, d within the register budget)
        return GORSE_OK;
    }();
    if (rc != GORSE_OK) {
        std::string keep = last_error();
        gorse_topk_destroy(h);
        last_error() = keep;
        return rc;
    }
    *out = h;
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_destroy(gorse_topk *h) {
    if (!h) return GORSE_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->stream);
    }
    topk_mfma_release(h);
    delete h;
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_search_index(gorse_topk *h, const int64_t *q, int64_t nq, int32_t k, int32_t prune0,
                                           int32_t *idx_out, float *dist_out, int32_t *count_out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (nq < 0 || k <= 0 || (nq > 0 && !q)) return fail(GORSE_ERR_INVALID, "bad arguments");
    for (int64_t t = 0; t < nq; t++)
        if (q[t] < 0 || q[t] >= h->N) return fail(GORSE_ERR_RANGE, "index out of range: %lld", (long long)q[t]);
    if (nq == 0) return GORSE_OK;
    GORSE_TRY(h->use());
    h->n_fallback = 0;
    if (topk_mfma_usable(h, nq, k)) return topk_mfma_search(h, q, -1, nullptr, nq, k, prune0, idx_out, dist_out, count_out);
    const int64_t bq = topk_scan_block_queries(h);
    GORSE_TRY(h->qidx.ensure((size_t)std::min(bq, nq)));
    GORSE_TRY(h->qbuf.ensure((size_t)std::min(bq, nq) * h->d));
    GORSE_TRY(h->qnorm.ensure((size_t)std::min(bq, nq)));
    for (int64_t q0 = 0; q0 < nq; q0 += bq) {
        const int64_t m = std::min(bq, nq - q0);
        GORSE_HIP_CHECK(hipMemcpyAsync(h->qidx.p, q + q0, (size_t)m * 8, hipMemcpyHostToDevice, h->stream));
        gather_rows_kernel<<<dim3((unsigned)m), dim3(64), 0, h->stream>>>(h->X.p, h->norm2.p, h->qidx.p, h->d, h->qbuf.p,
                                                                         h->qnorm.p);
        GORSE_HIP_CHECK(hipGetLastError());
        GORSE_TRY(topk_scan_block(h, m, h->qidx.p, q + q0, k, prune0, idx_out ? idx_out + q0 * k : nullptr,
                               dist_out ? dist_out + q0 * k : nullptr, count_out ? count_out + q0 : nullptr));
    }
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_search_vector(gorse_topk *h, const void *qv, int64_t nq, int32_t k, int32_t prune0,
                                            int32_t *idx_out, float *dist_out, int32_t *count_out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (nq < 0 || k <= 0 || (nq > 0 && !qv)) return fail(GORSE_ERR_INVALID, "bad arguments");
    if (nq == 0) return GORSE_OK;
    GORSE_TRY(h->use());
    h->n_fallback = 0;
    if (topk_mfma_usable(h, nq, k)) {  // all query vectors to the device as fp32, then path B
        GORSE_TRY(h->qf32.ensure((size_t)nq * h->d));
        if (h->dtype == GORSE_DTYPE_BF16) {
            DevBuf<uint16_t> q16;
            GORSE_TRY(q16.alloc((size_t)nq * h->d));
            GORSE_HIP_CHECK(hipMemcpyAsync(q16.p, qv, (size_t)nq * h->d * 2, hipMemcpyHostToDevice, h->stream));
            expand_bf16_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>(q16.p, h->qf32.p, nq * (int64_t)h->d);
            GORSE_HIP_CHECK(hipGetLastError());
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        } else {
            GORSE_HIP_CHECK(hipMemcpyAsync(h->qf32.p, qv, (size_t)nq * h->d * 4, hipMemcpyHostToDevice, h->stream));
        }
        return topk_mfma_search(h, nullptr, -1, h->qf32.p, nq, k, prune0, idx_out, dist_out, count_out);
    }
    const int64_t bq = topk_scan_block_queries(h);
    const int64_t mb = std::min(bq, nq);
    GORSE_TRY(h->qbuf.ensure((size_t)mb * h->d));
    GORSE_TRY(h->qnorm.ensure((size_t)mb));
    DevBuf<uint16_t> qb16;
    if (h->dtype == GORSE_DTYPE_BF16) GORSE_TRY(qb16.alloc((size_t)mb * h->d));
    for (int64_t q0 = 0; q0 < nq; q0 += bq) {
        const int64_t m = std::min(bq, nq - q0);
        if (h->dtype == GORSE_DTYPE_BF16) {
            GORSE_HIP_CHECK(hipMemcpyAsync(qb16.p, (const uint16_t *)qv + q0 * h->d, (size_t)m * h->d * 2,
                                           hipMemcpyHostToDevice, h->stream));
            expand_bf16_kernel<<<dim3(256), dim3(256), 0, h->stream>>>(qb16.p, h->qbuf.p, m * (int64_t)h->d);
            GORSE_HIP_CHECK(hipGetLastError());
        } else {
            GORSE_HIP_CHECK(hipMemcpyAsync(h->qbuf.p, (const float *)qv + q0 * h->d, (size_t)m * h->d * 4,
                                           hipMemcpyHostToDevice, h->stream));
        }
        if (h->metric == GORSE_METRIC_COSINE) GORSE_TRY(topk_compute_norms(h, h->qbuf.p, m, h->qnorm.p));
        GORSE_TRY(topk_scan_block(h, m, nullptr, nullptr, k, prune0, idx_out ? idx_out + q0 * k : nullptr,
                               dist_out ? dist_out + q0 * k : nullptr, count_out ? count_out + q0 : nullptr));
    }
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_all_pairs(gorse_topk *h, int64_t q_begin, int64_t q_end, int32_t k, int32_t *idx_out,
                                        float *dist_out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (q_begin < 0 || q_end > h->N || q_begin > q_end || k <= 0) return fail(GORSE_ERR_RANGE, "bad query range");
    const int64_t nq = q_end - q_begin;
    if (nq == 0) return GORSE_OK;
    GORSE_TRY(h->use());
    h->n_fallback = 0;
    if (topk_mfma_usable(h, nq, k)) return topk_mfma_search(h, nullptr, q_begin, nullptr, nq, k, 0, idx_out, dist_out, nullptr);
    const int64_t bq = topk_scan_block_queries(h);
    const int64_t mb = std::min(bq, nq);
    GORSE_TRY(h->qidx.ensure((size_t)mb));
    GORSE_TRY(h->qbuf.ensure((size_t)mb * h->d));
    GORSE_TRY(h->qnorm.ensure((size_t)mb));
    std::vector<int64_t> ids((size_t)mb);
    for (int64_t q0 = 0; q0 < nq; q0 += bq) {
        const int64_t m = std::min(bq, nq - q0);
        for (int64_t t = 0; t < m; t++) ids[t] = q_begin + q0 + t;
        GORSE_HIP_CHECK(hipMemcpyAsync(h->qidx.p, ids.data(), (size_t)m * 8, hipMemcpyHostToDevice, h->stream));
        gather_rows_kernel<<<dim3((unsigned)m), dim3(64), 0, h->stream>>>(h->X.p, h->norm2.p, h->qidx.p, h->d, h->qbuf.p,
                                                                         h->qnorm.p);
        GORSE_HIP_CHECK(hipGetLastError());
        GORSE_TRY(topk_scan_block(h, m, h->qidx.p, ids.data(), k, 0, idx_out ? idx_out + q0 * k : nullptr,
                               dist_out ? dist_out + q0 * k : nullptr, nullptr));
    }
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_synchronize(gorse_topk *h) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}
extern "C" int32_t gorse_topk_set_profiling(gorse_topk *h, int32_t on) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->prof.resolve();
    h->prof.on = on != 0;
    return GORSE_OK;
}
extern "C" int32_t gorse_topk_get_profile(gorse_topk *h, int32_t cls, int64_t *launches, double *total_ms) {
    if (!h || cls < 0 || cls >= GORSE_PROF_TOPK_NCLASSES) return fail(GORSE_ERR_INVALID, "bad kernel class");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->prof.resolve();
    if (launches) *launches = h->prof.launches[cls];
    if (total_ms) *total_ms = h->prof.ms[cls];
    return GORSE_OK;
}
extern "C" int32_t gorse_topk_last_stats(gorse_topk *h, int64_t *n_fallback, int64_t *n_tie_resolved) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (n_fallback) *n_fallback = h->n_fallback;
    if (n_tie_resolved) *n_tie_resolved = h->n_tie;
    return GORSE_OK;
}

// rscale_m[i] = admissible[i] ? base[i] : NaN -- a NaN score fails every comparison of the sweep; the kTopkRowPad entries
// behind row n - 1 are NaN as well (the sweep's tiles read them for the rows past N)
__global__ void masked_scale_kernel(const uint8_t *__restrict__ mask, const float *__restrict__ base, int64_t n, int64_t n_padded,
                                    float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_padded; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (i < n && mask[i]) ? base[i] : __builtin_nanf("");
}

extern "C" int32_t gorse_topk_set_mask(gorse_topk *h, const uint8_t *admissible) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    if (!admissible) {
        h->has_mask = false;
        return GORSE_OK;
    }
    GORSE_TRY(h->mask.ensure((size_t)h->N));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->mask.p, admissible, (size_t)h->N, hipMemcpyHostToDevice, h->stream));
    h->n_admissible = 0;
    for (int64_t r = 0; r < h->N; r++) h->n_admissible += admissible[r] != 0;
    if (h->mfma_ok) {
        GORSE_TRY(h->rscale_m.ensure((size_t)h->N + gorse::kTopkRowPad));
        masked_scale_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>(h->mask.p, h->rscale.p, h->N, h->N + gorse::kTopkRowPad, h->rscale_m.p);
        GORSE_HIP_CHECK(hipGetLastError());
    }
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->has_mask = true;
    return GORSE_OK;
}

// probe: queries of the last search whose warm-started threshold could not be verified and that were swept again
extern "C" int32_t gorse_hip_test_topk_resweeps(gorse_topk *h, int64_t *n) {
    if (!h || !n) return fail(GORSE_ERR_INVALID, "NULL argument");
    *n = h->n_resweep;
    return GORSE_OK;
}

// probe: did the main sweep of the last search (its last chunk) take the symmetric form (csrc/topk_mfma.hip, SYM)
extern "C" int32_t gorse_hip_test_topk_last_symmetric(gorse_topk *h, int32_t *sym) {
    if (!h || !sym) return fail(GORSE_ERR_INVALID, "NULL argument");
    *sym = h->last_sym ? 1 : 0;
    return GORSE_OK;
}

// probe: the four counters of the last symmetric search: queries the pilot left without a threshold, warm starts the rescoring could
// not verify, foreign lists that overflowed, hits beyond a wave's staging area
extern "C" int32_t gorse_hip_test_topk_sym_stats(gorse_topk *h, uint64_t *out4) {
    if (!h || !out4) return fail(GORSE_ERR_INVALID, "NULL argument");
    for (int i = 0; i < 4; i++) out4[i] = 0;
    if (h->sym_stats.n < 4) return GORSE_OK;
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipMemcpyAsync(out4, h->sym_stats.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

// probe: the warm-start thresholds of the last search's last chunk (n of them, as the pilot -- or, for a symmetric sweep,
// sym_thresholds_kernel -- left them)
extern "C" int32_t gorse_hip_test_topk_get_thresholds(gorse_topk *h, float *out, int64_t n) {
    if (!h || !out) return fail(GORSE_ERR_INVALID, "NULL argument");
    if ((int64_t)h->f0.n < n) return fail(GORSE_ERR_INVALID, "no warm start of that size has run on this handle");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipMemcpyAsync(out, h->f0.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

// test hook: the per-query flags of the last MFMA search's last chunk as the host read them behind the rescoring (non-zero = the
// query went on to the tie path: ties among its k + 1 best, a warm start that could not be verified, no pilot threshold, a list
// that overflowed), and -- after a symmetric sweep -- how many entries the other workgroups appended to every query's foreign list
// (more than the list's 512 slots = it overflowed)
extern "C" int32_t gorse_hip_test_topk_get_flags(gorse_topk *h, uint8_t *flags, int64_t n) {
    if (!h || !flags) return fail(GORSE_ERR_INVALID, "NULL argument");
    if ((int64_t)h->cflag.n < n) return fail(GORSE_ERR_INVALID, "no MFMA search of that size has run on this handle");
    GORSE_TRY(h->use());  // the flag bytes stay on the device behind the rescoring: the tie path lists them there (flag_compact_kernel)
    GORSE_HIP_CHECK(hipMemcpyAsync(flags, h->cflag.p, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}
extern "C" int32_t gorse_hip_test_topk_get_foreign_counts(gorse_topk *h, int32_t *counts, int64_t n) {
    if (!h || !counts) return fail(GORSE_ERR_INVALID, "NULL argument");
    if ((int64_t)h->fcnt.n < n) return fail(GORSE_ERR_INVALID, "no symmetric sweep of that size has run on this handle");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipMemcpyAsync(counts, h->fcnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

// probe (variant bit 24): the pilot's per-query flags and list lengths
extern "C" int32_t gorse_hip_test_topk_get_pilot_state(gorse_topk *h, uint8_t *flags, int32_t *counts, int64_t n) {
    if (!h || !flags || !counts) return fail(GORSE_ERR_INVALID, "NULL argument");
    if ((int64_t)h->dbg_flags.size() < n) return fail(GORSE_ERR_INVALID, "no pilot-only search of that size has run on this handle");
    memcpy(flags, h->dbg_flags.data(), (size_t)n);
    memcpy(counts, h->dbg_counts.data(), (size_t)n * 4);
    return GORSE_OK;
}

// test hook: 0 = automatic path choice, 1 = path A only (literal scan), 2 = path B whenever its operands exist
extern "C" void gorse_hip_test_set_topk_path(int32_t path) { gorse::g_topk_force_path = path; }
extern "C" void gorse_hip_test_set_topk_variant(int32_t v) { gorse::g_topk_variant = v; }
// probe: the 8 phase counters of the last instrumented sweep (variant bit 4) of this handle
extern "C" int32_t gorse_hip_test_get_sweep_profile(gorse_topk *h, uint64_t *out16) {
    if (!h || !out16) return fail(GORSE_ERR_INVALID, "NULL argument");
    if (h->sweep_prof.n < 16) return fail(GORSE_ERR_INVALID, "no instrumented sweep has run on this handle");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipMemcpyAsync(out16, h->sweep_prof.p, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" void gorse_hip_test_set_scan_literal(int32_t on) { g_scan_literal_only = on != 0; }
