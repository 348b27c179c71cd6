// sgemm.hip -- floats.MM / blas.SGEMM on gfx950.
// Reference: common/floats/floats.go:241, mm.go:19-49, floats_amd64.go:188-197,
// src/floats_avx512.c:443-480, common/blas/blas_openblas.go:23-26.
//
// Semantics are the reference's (NOT BLAS beta=0): the NN, TN and TT cases accumulate into C
// through an l-ascending FMA chain per element (clang contracts `c += a*b` in _mm512_mm), the NT
// case overwrites C with floats.Dot of two rows in AVX512 lane order.  The f32 chain is exactly
// what one accumulator of v_mfma_f32_32x32x2_f32 does -- checked bit for bit on the device
// (scripts/probe_mfma_f32_chain.hip, profiles/r04_q_probe_mfma_f32_chain.txt: wide exponent ranges, signed
// zeros) -- so the three chain cases run on the matrix cores (sgemm_mfma_kernel, round 4) and stay
// bit-identical to the AVX512 build; sgemm_chain_kernel is the same chain on the vector ALU (tiny shapes).
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

// NN / TN / TT on the fp32 MFMA.  A workgroup of four waves owns a 128 x 128 (or 64 x 64) tile of C, a wave a quarter of it as 2 x 2
// (or one) accumulators of 32 x 32; the operands pass through LDS in blocks of 16 l: As[i][l] (row stride 17 words: the 32 lanes of a fragment read 32
// different banks), Bs[l][j].  One v_mfma_f32_32x32x2_f32 advances every element of an accumulator by TWO steps of its chain
// (l, then l + 1); an odd last l is one fmaf per element afterwards -- never a padded zero step, which would turn an accumulated
// -0 into +0.  Rows / columns past m / n are loaded as zeros and not stored.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMmKB = 16, kMmAs = kMmKB + 1;

// W = 2: a 128 x 128 tile per workgroup (2 x 2 accumulators per wave); W = 1: 64 x 64 (one accumulator per wave) for shapes that
// would otherwise leave most of the 256 CUs without a tile.  The next block's operands are loaded into registers while the
// current block's MFMAs run (global latency behind the matrix pipe), then stored to LDS between two barriers.
template <bool TA, bool TB, int W>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(int m, int n, int k, const float *__restrict__ a, int lda,
                                                         const float *__restrict__ b, int ldb, float *__restrict__ c, int ldc) {
    constexpr int T = 64 * W;       // tile edge
    constexpr int PT = T * kMmKB / 256;  // elements of each operand tile per thread (8 or 4)
    __shared__ float As[T * kMmAs];
    __shared__ float Bs[kMmKB * T];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = blockIdx.y * T, j0 = blockIdx.x * T;
    const int wi = 32 * W * (wv >> 1), wj = 32 * W * (wv & 1);  // the wave's quarter inside the tile
    f32x16 acc[W][W];
#pragma unroll
    for (int bi = 0; bi < W; bi++)
#pragma unroll
        for (int bj = 0; bj < W; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = i0 + wi + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = j0 + wj + 32 * bj + (lane & 31);
                acc[bi][bj][r] = (i < m && j < n) ? c[(int64_t)i * ldc + j] : 0.0f;
            }
    const int k2 = k & ~1;  // the l that come in pairs
    // element e of this thread in the A tile is (ti, tl), in the B tile (ul, uj): contiguous along the operand's storage order
    auto a_at = [&](int e, int &ti, int &tl) {
        const int x = tid * PT + e;
        if (TA) {  // a[l * lda + i]: contiguous in i
            tl = x / T;
            ti = x % T;
        } else {   // a[i * lda + l]: contiguous in l
            ti = x / kMmKB;
            tl = x % kMmKB;
        }
    };
    auto b_at = [&](int e, int &ul, int &uj) {
        const int x = tid * PT + e;
        if (TB) {  // b[j * ldb + l]: contiguous in l
            uj = x / kMmKB;
            ul = x % kMmKB;
        } else {   // b[l * ldb + j]: contiguous in j
            ul = x / T;
            uj = x % T;
        }
    };
    float ra[PT], rb[PT];
    auto gload = [&](int l0) {
#pragma unroll
        for (int e = 0; e < PT; e++) {
            int ti, tl, ul, uj;
            a_at(e, ti, tl);
            b_at(e, ul, uj);
            const int gi = i0 + ti, gl = l0 + tl, gj = j0 + uj, hl = l0 + ul;
            ra[e] = (gi < m && gl < k2) ? (TA ? a[(int64_t)gl * lda + gi] : a[(int64_t)gi * lda + gl]) : 0.0f;
            rb[e] = (gj < n && hl < k2) ? (TB ? b[(int64_t)gj * ldb + hl] : b[(int64_t)hl * ldb + gj]) : 0.0f;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int e = 0; e < PT; e++) {
            int ti, tl, ul, uj;
            a_at(e, ti, tl);
            b_at(e, ul, uj);
            As[ti * kMmAs + tl] = ra[e];
            Bs[ul * T + uj] = rb[e];
        }
    };
    if (k2 > 0) {
        gload(0);
        lstore();
        __syncthreads();
    }
    for (int l0 = 0; l0 < k2; l0 += kMmKB) {
        const bool more = l0 + kMmKB < k2;
        if (more) gload(l0 + kMmKB);
        const int steps = min(kMmKB, k2 - l0);  // even
        for (int kk = 0; kk < steps; kk += 2) {
            float fa[W], fb[W];
#pragma unroll
            for (int bi = 0; bi < W; bi++) fa[bi] = As[(wi + 32 * bi + (lane & 31)) * kMmAs + kk + (lane >> 5)];
#pragma unroll
            for (int bj = 0; bj < W; bj++) fb[bj] = Bs[(kk + (lane >> 5)) * T + wj + 32 * bj + (lane & 31)];
#pragma unroll
            for (int bi = 0; bi < W; bi++)
#pragma unroll
                for (int bj = 0; bj < W; bj++) acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[bi], fb[bj], acc[bi][bj], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }
    const bool odd = (k & 1) != 0;
#pragma unroll
    for (int bi = 0; bi < W; bi++)
#pragma unroll
        for (int bj = 0; bj < W; bj++) {
            const int j = j0 + wj + 32 * bj + (lane & 31);
            float bl = 0.0f;
            if (odd && j < n) bl = TB ? b[(int64_t)j * ldb + (k - 1)] : b[(int64_t)(k - 1) * ldb + j];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = i0 + wi + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (i < m && j < n) {
                    float v = acc[bi][bj][r];
                    if (odd) v = fmaf(TA ? a[(int64_t)(k - 1) * lda + i] : a[(int64_t)i * lda + (k - 1)], bl, v);
                    c[(int64_t)i * ldc + j] = v;
                }
            }
        }
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

int g_sgemm_valu = 0;  // test hook: 1 = the vector-ALU chain whatever the shape (gorse_hip_test_set_sgemm_valu)
double g_sgemm_last_ms = 0.0;  // kernel time of the last call (hipEvents around the launch): gorse_hip_test_sgemm_last_ms

}  // namespace

extern "C" void gorse_hip_test_set_sgemm_valu(int32_t on) { g_sgemm_valu = on; }
extern "C" double gorse_hip_test_sgemm_last_ms(void) { return g_sgemm_last_ms; }

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
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    (void)hipEventCreate(&ev0);
    (void)hipEventCreate(&ev1);
    (void)hipEventRecord(ev0, st);
    if (!transA && transB) {
        int64_t blocks = std::min<int64_t>(ceil_div((int64_t)m * n, kGroupsPerBlock), 8192);
        sgemm_nt_kernel<<<dim3((unsigned)blocks), dim3(kBlock), (size_t)kGroupsPerBlock * 2 * std::max(k, 1) * 4, st>>>(
            m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);
    } else if (k > 0 && (int64_t)m * n >= 64 * 64 && !g_sgemm_valu) {  // the matrix cores (a tile is 128 x 128: below 64 x 64 the vector ALU form)
        // 128 x 128 tiles where they give every CU work (>= 512 of them: two per CU), 64 x 64 tiles otherwise
        const bool big = (int64_t)ceil_div(n, 128) * ceil_div(m, 128) >= 512;
        const int T = big ? 128 : 64;
        dim3 grid((unsigned)ceil_div(n, T), (unsigned)ceil_div(m, T)), block(256);
#define MM(TA_, TB_)                                                                                                   \
    do {                                                                                                               \
        if (big)                                                                                                       \
            sgemm_mfma_kernel<TA_, TB_, 2><<<grid, block, 0, st>>>(m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);          \
        else                                                                                                           \
            sgemm_mfma_kernel<TA_, TB_, 1><<<grid, block, 0, st>>>(m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);          \
    } while (0)
        if (!transA && !transB)
            MM(false, false);
        else if (transA && !transB)
            MM(true, false);
        else
            MM(true, true);
#undef MM
    } else if (k > 0) {
        dim3 grid((unsigned)ceil_div(n, TS), (unsigned)ceil_div(m, TS)), block(TS * TS);
        if (!transA && !transB)
            sgemm_chain_kernel<false, false><<<grid, block, 0, st>>>(m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);
        else if (transA && !transB)
            sgemm_chain_kernel<true, false><<<grid, block, 0, st>>>(m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);
        else
            sgemm_chain_kernel<true, true><<<grid, block, 0, st>>>(m, n, k, da.p, lda, db.p, ldb, dc.p, ldc);
    }
    const hipError_t launched = hipGetLastError();
    (void)hipEventRecord(ev1, st);
    hipError_t rc = launched;
    if (rc == hipSuccess) rc = hipMemcpyAsync(c, dc.p, nc * 4, hipMemcpyDeviceToHost, st);
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    float ms = 0.0f;
    if (rc == hipSuccess && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) g_sgemm_last_ms = ms;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    if (rc != hipSuccess) return fail(GORSE_ERR_HIP, "gorse_hip_sgemm: %s", hipGetErrorString(rc));
    return GORSE_OK;
}
