// sgemm.hip -- floats.MM / blas.SGEMM on gfx950.
// Reference: common/floats/floats.go:241, mm.go:19-49, floats_amd64.go:188-197,
// src/floats_avx512.c:443-480, common/blas/blas_openblas.go:23-26.
//
// Semantics are the reference's (NOT BLAS beta=0): the NN, TN and TT cases accumulate into C
// through an l-ascending FMA chain per element (clang contracts `c += a*b` in _mm512_mm), the NT
// case overwrites C with floats.Dot of two rows in AVX512 lane order.  The f32 chain is exactly
// what one MFMA f32 accumulator does, so results are bit-identical to the AVX512 build.
#include <algorithm>

#include "cf_device.hpp"

using namespace gorse;

namespace {

constexpr int TS = 16;

// NN / TN / TT: C[i][j] = fma chain over l ascending, starting from C's previous value.
template <bool TA, bool TB>
__global__ __launch_bounds__(TS *TS) void sgemm_chain_kernel(int m, int n, int k, const float *__restrict__ a, int lda,
                                                             const float *__restrict__ b, int ldb,
                                                             float *__restrict__ c, int ldc) {
    __shared__ float sa[TS][TS + 1], sb[TS][TS + 1];
    const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
    const int i = blockIdx.y * TS + ty, j = blockIdx.x * TS + tx;
    float acc = (i < m && j < n) ? c[(int64_t)i * ldc + j] : 0.0f;
    for (int l0 = 0; l0 < k; l0 += TS) {
        // sa[ty][tx] = A(i_block + ty, l0 + tx) ; sb[ty][tx] = B(l0 + ty, j_block + tx)
        const int ai = blockIdx.y * TS + ty, al = l0 + tx;
        sa[ty][tx] = (ai < m && al < k) ? (TA ? a[(int64_t)al * lda + ai] : a[(int64_t)ai * lda + al]) : 0.0f;
        const int bl = l0 + ty, bj = blockIdx.x * TS + tx;
        sb[ty][tx] = (bl < k && bj < n) ? (TB ? b[(int64_t)bj * ldb + bl] : b[(int64_t)bl * ldb + bj]) : 0.0f;
        __syncthreads();
        const int lim = k - l0 < TS ? k - l0 : TS;
        for (int l = 0; l < lim; l++) acc = fmaf(sa[ty][l], sb[l][tx], acc);
        __syncthreads();
    }
    if (i < m && j < n) c[(int64_t)i * ldc + j] = acc;
}

// NT: C[i][j] = floats.Dot(A row i, B row j) in AVX512 order; one 16-lane group per element.
__global__ __launch_bounds__(kBlock) void sgemm_nt_kernel(int m, int n, int k, const float *__restrict__ a, int lda,
                                                          const float *__restrict__ b, int ldb, float *__restrict__ c,
                                                          int ldc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & (kGroup - 1), gib = threadIdx.x / kGroup;
    const VecShape vs(k);
    float *sa = smem + (size_t)gib * 2 * k, *sb = sa + k;
    const int64_t total = (int64_t)m * n;
    for (int64_t t = (int64_t)blockIdx.x * kGroupsPerBlock + gib; t < total; t += (int64_t)gridDim.x * kGroupsPerBlock) {
        const int i = (int)(t / n), j = (int)(t % n);
        for (int e = lane; e < k; e += kGroup) {
            sa[e] = a[(int64_t)i * lda + e];
            sb[e] = b[(int64_t)j * ldb + e];
        }
        __builtin_amdgcn_wave_barrier();
        float r = dot512_lds(sa, sb, vs, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) c[(int64_t)i * ldc + j] = r;
    }
}

}  // namespace

extern "C" int32_t gorse_hip_sgemm(int32_t device, int32_t transA, int32_t transB, int32_t m, int32_t n, int32_t k,
                                   const float *a, int32_t lda, const float *b, int32_t ldb, float *c, int32_t ldc) {
    if (m < 0 || n < 0 || k < 0) return fail(GORSE_ERR_INVALID, "negative dimension");
    if (m == 0 || n == 0) return GORSE_OK;
    if (!a || !b || !c) return fail(GORSE_ERR_INVALID, "NULL matrix");
    const int a_rows = transA ? k : m, a_cols = transA ? m : k;
    const int b_rows = transB ? n : k, b_cols = transB ? k : n;
    if (lda < a_cols || ldb < b_cols || ldc < n) return fail(GORSE_ERR_INVALID, "leading dimension too small");
    if (k > 8192 && !transA && transB) return fail(GORSE_ERR_INVALID, "k %d > 8192 unsupported in the NT case", k);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GORSE_ERR_NO_DEVICE, "no HIP device visible (libgorse_hip needs an MI355X / gfx950)");
    if (device < 0 || device >= ndev) return fail(GORSE_ERR_INVALID, "device %d out of range", device);
    GORSE_HIP_CHECK(hipSetDevice(device));
    const size_t na = (size_t)(a_rows > 0 ? (a_rows - 1) : 0) * lda + a_cols;
    const size_t nb = (size_t)(b_rows > 0 ? (b_rows - 1) : 0) * ldb + b_cols;
    const size_t nc = (size_t)(m - 1) * ldc + n;
    DevBuf<float> da, db, dc;
    GORSE_TRY(da.alloc(na));
    GORSE_TRY(db.alloc(nb));
    GORSE_TRY(dc.alloc(nc));
    hipStream_t st = nullptr;  // one-shot call: the null stream is fine
    if (k > 0) {
        GORSE_HIP_CHECK(hipMemcpyAsync(da.p, a, na * 4, hipMemcpyHostToDevice, st));
        GORSE_HIP_CHECK(hipMemcpyAsync(db.p, b, nb * 4, hipMemcpyHostToDevice, st));
    }
    GORSE_HIP_CHECK(hipMemcpyAsync(dc.p, c, nc * 4, hipMemcpyHostToDevice, st));
    if (!transA && transB) {
        int64_t blocks = std::min<int64_t>(ceil_div((int64_t)m * n, kGroupsPerBlock), 8192);
        sgemm_nt_kernel<<<dim3((unsigned)blocks), dim3(kBlock), (size_t)kGroupsPerBlock * 2 * std::max(k, 1) * 4, st>>>(
            m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);
    } else if (k > 0) {
        dim3 grid((unsigned)ceil_div(n, TS), (unsigned)ceil_div(m, TS)), block(TS * TS);
        if (!transA && !transB)
            sgemm_chain_kernel<false, false><<<grid, block, 0, st>>>(m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);
        else if (transA && !transB)
            sgemm_chain_kernel<true, false><<<grid, block, 0, st>>>(m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);
        else
            sgemm_chain_kernel<true, true><<<grid, block, 0, st>>>(m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);
    }
    GORSE_HIP_CHECK(hipGetLastError());
    GORSE_HIP_CHECK(hipMemcpyAsync(c, dc.p, nc * 4, hipMemcpyDeviceToHost, st));
    GORSE_HIP_CHECK(hipStreamSynchronize(st));
    return GORSE_OK;
}
