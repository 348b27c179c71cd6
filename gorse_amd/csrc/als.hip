// als.hip -- ALS / eALS epoch on gfx950.  Reference: model/cf/model.go:641-738.
//
//   als_gram_*        S = sum over rows with feedback of x x^T   (model.go:645-658, 693-706)
//   als_sweep_kernel  one workgroup per user (item): element-wise coordinate descent over the
//                     d factors with the row's feedback rows staged in LDS (model.go:659-690,
//                     707-738).  HBM-bound: algorithmic bytes per half-sweep = nnz*d*4 gathered
//                     + rows*d*4*2.
// Rows are independent inside a half-sweep, so unlike BPR the result does not depend on
// scheduling; sums are tree-reduced (the reference adds sequentially) => parity within 1e-4
// relative, as BASELINE.md section 2 states.
#include <algorithm>

#include "mf_internal.hpp"

using namespace gorse;

namespace {

constexpr int kGramRows = 8;  // rows staged per barrier pair

// partial[g][d*d] = sum over this block's rows (with feedback) of x x^T
__global__ __launch_bounds__(256) void als_gram_partial_kernel(const float *__restrict__ F,
                                                               const int64_t *__restrict__ ptr, int64_t rows, int d,
                                                               float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // kGramRows * d
    __shared__ int has[kGramRows];
    const int dd = d * d;
    constexpr int MAXE = 64;  // entries per thread (d <= 128)
    float acc[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; e++) acc[e] = 0.0f;
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    for (int64_t rb = r0; rb < r1; rb += kGramRows) {
        const int nr = (int)(r1 - rb < kGramRows ? r1 - rb : kGramRows);
        for (int t = threadIdx.x; t < nr * d; t += blockDim.x) smem[t] = F[rb * d + t];
        if (threadIdx.x < kGramRows)
            has[threadIdx.x] = threadIdx.x < nr && (ptr[rb + threadIdx.x + 1] - ptr[rb + threadIdx.x]) > 0;
        __syncthreads();
        for (int r = 0; r < nr; r++) {
            if (!has[r]) continue;
            const float *x = smem + r * d;
#pragma unroll
            for (int e = 0; e < MAXE; e++) {
                int idx = threadIdx.x + e * 256;
                if (idx < dd) acc[e] += x[idx / d] * x[idx % d];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < MAXE; e++) {
        int idx = threadIdx.x + e * 256;
        if (idx < dd) partial[(size_t)blockIdx.x * dd + idx] = acc[e];
    }
}

__global__ void als_gram_reduce_kernel(const float *__restrict__ partial, int nparts, int dd, float *__restrict__ S) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= dd) return;
    float s = 0.0f;
    for (int g = 0; g < nparts; g++) s += partial[(size_t)g * dd + e];  // fixed order: deterministic
    S[e] = s;
}

// generic (any d) fallback gram: one thread per (i,j) entry, sequential over rows (exactly the
// reference's accumulation order); used when d > 128
__global__ void als_gram_naive_kernel(const float *__restrict__ F, const int64_t *__restrict__ ptr, int64_t rows, int d,
                                      float *__restrict__ S) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d * d) return;
    const int i = e / d, j = e % d;
    float s = 0.0f;
    for (int64_t r = 0; r < rows; r++)
        if (ptr[r + 1] - ptr[r] > 0) s += F[r * d + i] * F[r * d + j];
    S[e] = s;
}

struct SweepShared {
    int pred_cap;  // entries of pred kept in LDS
    int q_cap;     // feedback rows staged in LDS (stride d+1)
};

// block-wide sum of three floats; result broadcast to every thread
__device__ __forceinline__ void block_sum3(float &a, float &b, float &c, float *red /*3*4*/) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
        c += __shfl_down(c, off);
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) {
        red[w * 3 + 0] = a;
        red[w * 3 + 1] = b;
        red[w * 3 + 2] = c;
    }
    __syncthreads();
    a = (red[0] + red[3]) + (red[6] + red[9]);
    b = (red[1] + red[4]) + (red[7] + red[10]);
    c = (red[2] + red[5]) + (red[8] + red[11]);
    __syncthreads();
}

// A: matrix being solved (rows x d), B: the other side, ptr/idx: this side's feedback CSR, S: d x d gram of B
__global__ __launch_bounds__(256) void als_sweep_kernel(float *__restrict__ A, const float *__restrict__ B,
                                                        const int64_t *__restrict__ ptr,
                                                        const int32_t *__restrict__ idx, const float *__restrict__ S,
                                                        int64_t rows, int d, float w, float reg, int pred_cap,
                                                        int q_cap, float *__restrict__ scratch, int64_t scratch_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sp = smem;                      // d          current row of A
    float *red = sp + ((d + 3) & ~3);      // 12
    float *tmp = red + 12;                 // 2*16*d     dot staging for generic d
    const int dotbytes = 0;
    (void)dotbytes;
    float *spred = tmp + 2 * kGroupsPerBlock * d;  // pred_cap
    float *sq = spred + pred_cap;                   // q_cap * (d+1)
    const int lane = threadIdx.x & (kGroup - 1), gib = threadIdx.x / kGroup;
    const VecShape vs(d);
    const float one_w = 1 - w;
    float *gpred = scratch + (size_t)blockIdx.x * scratch_stride;
    for (int64_t u = blockIdx.x; u < rows; u += gridDim.x) {
        const int64_t beg = ptr[u];
        const int n = (int)(ptr[u + 1] - beg);
        const int32_t *fb = idx + beg;
        float *pu = A + u * d;
        for (int e = threadIdx.x; e < d; e += blockDim.x) sp[e] = pu[e];
        // stage the first q_cap feedback rows of B
        const int nq = n < q_cap ? n : q_cap;
        for (int t = gib; t < nq; t += kGroupsPerBlock) {
            const float *row = B + (int64_t)fb[t] * d;
            for (int e = lane; e < d; e += kGroup) sq[t * (d + 1) + e] = row[e];
        }
        __syncthreads();
        float *pred = n <= pred_cap ? spred : gpred;
        // predictions: internalPredict (floats.Dot, AVX512 order) -- model.go:661-663
        float *ta = tmp + (size_t)gib * 2 * d, *tb = ta + d;
        for (int t = gib; t < n; t += kGroupsPerBlock) {
            const float *row = B + (int64_t)fb[t] * d;
            for (int e = lane; e < d; e += kGroup) {
                ta[e] = sp[e];
                tb[e] = t < nq ? sq[t * (d + 1) + e] : row[e];
            }
            __builtin_amdgcn_wave_barrier();
            float r = dot512_lds(ta, tb, vs, lane);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) pred[t] = r;
        }
        __syncthreads();
        for (int f = 0; f < d; f++) {
            const float pf = sp[f];
            float a = 0.0f, b = 0.0f, c = 0.0f;
            for (int t = threadIdx.x; t < n; t += blockDim.x) {
                const float q = t < nq ? sq[t * (d + 1) + f] : B[(int64_t)fb[t] * d + f];
                const float res = pred[t] - pf * q;
                pred[t] = res;
                a += (1 - one_w * res) * q;
                c += one_w * q * q;
            }
            for (int k = threadIdx.x; k < d; k += blockDim.x)
                if (k != f) b += w * sp[k] * S[k * d + f];
            block_sum3(a, b, c, red);
            const float nf = (a - b) / (c + w * S[f * d + f] + reg);
            if (threadIdx.x == 0) sp[f] = nf;
            for (int t = threadIdx.x; t < n; t += blockDim.x) {
                const float q = t < nq ? sq[t * (d + 1) + f] : B[(int64_t)fb[t] * d + f];
                pred[t] = pred[t] + nf * q;
            }
            __syncthreads();
        }
        for (int e = threadIdx.x; e < d; e += blockDim.x) pu[e] = sp[e];
        __syncthreads();
    }
}

int32_t run_gram(gorse_mf *h, const float *F, const int64_t *ptr, int64_t rows, int tok_cls) {
    const int d = h->d, dd = d * d;
    GORSE_TRY(h->gram.ensure((size_t)dd));
    int tok = h->prof.begin(tok_cls, h->stream);
    if (d <= 128) {
        int nparts = (int)std::min<int64_t>(1024, ceil_div(rows, kGramRows));
        GORSE_TRY(h->gram_partial.ensure((size_t)nparts * dd));
        als_gram_partial_kernel<<<dim3(nparts), dim3(256), (size_t)kGramRows * d * sizeof(float), h->stream>>>(
            F, ptr, rows, d, h->gram_partial.p);
        GORSE_HIP_CHECK(hipGetLastError());
        als_gram_reduce_kernel<<<dim3((unsigned)ceil_div(dd, 256)), dim3(256), 0, h->stream>>>(h->gram_partial.p, nparts,
                                                                                              dd, h->gram.p);
    } else {
        als_gram_naive_kernel<<<dim3((unsigned)ceil_div(dd, 256)), dim3(256), 0, h->stream>>>(F, ptr, rows, d, h->gram.p);
    }
    GORSE_HIP_CHECK(hipGetLastError());
    h->prof.end(tok, h->stream);
    return GORSE_OK;
}

int32_t run_sweep(gorse_mf *h, float *A, const float *B, const int64_t *ptr, const int32_t *idx, int64_t rows,
                  int64_t max_row, float w, float reg) {
    const int d = h->d;
    const size_t fixed = ((size_t)((d + 3) & ~3) + 12 + 2 * (size_t)kGroupsPerBlock * d) * sizeof(float);
    const size_t budget = 64 * 1024;
    if (fixed + 1024 > 150 * 1024) return fail(GORSE_ERR_INVALID, "nFactors %d too large for the ALS kernel", d);
    int pred_cap = 4096, q_cap = 0;
    if (fixed + (size_t)pred_cap * 4 < budget) q_cap = (int)((budget - fixed - (size_t)pred_cap * 4) / ((size_t)(d + 1) * 4));
    while (fixed + (size_t)pred_cap * 4 + (size_t)q_cap * (d + 1) * 4 > 150 * 1024 && pred_cap > 256) pred_cap /= 2;
    const size_t shmem = fixed + (size_t)pred_cap * 4 + (size_t)q_cap * (d + 1) * 4;
    int blocks = (int)std::min<int64_t>(rows, 256 * 8);
    const int64_t stride = max_row > pred_cap ? max_row : 1;
    GORSE_TRY(h->als_scratch.ensure((size_t)blocks * stride));
    GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)shmem));
    int tok = h->prof.begin(GORSE_PROF_ALS_SWEEP, h->stream);
    als_sweep_kernel<<<dim3(blocks), dim3(256), shmem, h->stream>>>(A, B, ptr, idx, h->gram.p, rows, d, w, reg, pred_cap,
                                                                   q_cap, h->als_scratch.p, stride);
    GORSE_HIP_CHECK(hipGetLastError());
    h->prof.end(tok, h->stream);
    return GORSE_OK;
}

}  // namespace

extern "C" int32_t gorse_als_epoch(gorse_mf *h, float weight, float reg, const volatile int32_t *cancel) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (!h->has_item_csr) return fail(GORSE_ERR_INVALID, "ALS needs the item feedback CSR (item_indptr/item_indices)");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    const int64_t max_u = h->max_user_row, max_i = h->max_item_row;
    if (cancel && *cancel) return fail(GORSE_ERR_CANCELLED, "cancelled");
    GORSE_TRY(run_gram(h, h->Q.p, h->iptr.p, h->I, GORSE_PROF_ALS_GRAM));                          // model.go:645-658
    GORSE_TRY(run_sweep(h, h->P.p, h->Q.p, h->uptr.p, h->uidx.p, h->U, max_u, weight, reg));       // model.go:659-690
    if (cancel && *cancel) {
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        return fail(GORSE_ERR_CANCELLED, "cancelled");
    }
    GORSE_TRY(run_gram(h, h->P.p, h->uptr.p, h->U, GORSE_PROF_ALS_GRAM));                          // model.go:693-706
    GORSE_TRY(run_sweep(h, h->Q.p, h->P.p, h->iptr.p, h->iidx.p, h->I, max_i, weight, reg));       // model.go:707-738
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}
