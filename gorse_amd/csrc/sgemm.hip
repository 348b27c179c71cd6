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
// current block's MFMAs run (global latency behind the matrix pipe) and stored into the OTHER LDS buffer: one barrier per block.
// The operands are PADDED by gorse_hip_sgemm: m and n are multiples of the tile edge (rows / columns past the caller's are computed
// and never copied back), and the l range of A and B is ALLOCATED up to a multiple of 16 -- read, never multiplied: the MFMA loop
// stops at k2.  So the kernel has no edge: a thread's PT elements of an operand block are contiguous in the operand's storage, ONE
// address per operand and block, the elements at immediate offsets.  (With sixteen separately clamped 64-bit addresses the kernel
// held 236 registers = two waves per SIMD, the matrix pipe 43 % busy: 61-65 TFLOP/s at 4096^3, profiles/r04_zd_pmc_SQ_mm.txt.)
// Round 6, measured and not kept (profiles/r06_k_ab_mm_pipelined.txt, r06_l_ab_mm_eight_waves.txt, r06_m_pmc_SQ_mm.txt; every form bit-equal):
//   * the fragment reads of step s + 1 issued before the MFMAs of step s (unrolled block, the LDS round trip hidden inside the wave):
//     needs a second fragment set = 168 registers = three workgroups per CU -- 96.7 TFLOP/s against 112.6 at 4096^3;
//   * eight waves per workgroup (64 x 32 of the tile each, 54 registers, eight waves per SIMD), four or three workgroups per CU:
//     112.3-112.6 against 113.6 -- the kernel is not short of ready waves either.
// What the counters say about the shipped form at 4096^3: SQ_VALU_MFMA_BUSY_CYCLES 2^31 = 64 cycles x every MFMA; GRBM_GUI_ACTIVE / 8 XCDs
// / 1.199 ms = 2.05 GHz -- the clock the chip sustains under this load (MI355X_MICROARCH.md, DVFS), not the 2.4 GHz the 157.3 TFLOP/s
// peak is quoted at -- and the matrix pipe is busy in 0.85 of those cycles: 114.6 TFLOP/s is 0.73 of the nominal peak and 0.85 of the
// 134.6 TFLOP/s this clock allows.  126 TFLOP/s (0.80 nominal) would need the pipe 0.94 busy at 2.05 GHz.  SQ_LDS_BANK_CONFLICT reads
// 2^25 cycles (5 % of a CU's time); a conflict-free assignment of the A tile's stores (the two halves of a row of 16 l to lanes whose
// rows never differ by 8) changed neither that counter nor the time (114.9 against 114.4, r06_n_ab_mm_astore.txt): not kept.
#ifndef GORSE_MM_WAVES
#define GORSE_MM_WAVES 4  // waves per workgroup of the 128 x 128 tile form: 4 (a 64 x 64 quarter each) or 8 (half a quarter: 64 x 32)
#endif
#ifndef GORSE_MM_WGS8
#define GORSE_MM_WGS8 4   // workgroups per CU the eight-wave form is compiled for (4: 64 registers per wave; 3: 85)
#endif
constexpr int mm_waves(int w) { return w == 2 ? GORSE_MM_WAVES : 4; }
template <bool TA, bool TB, int W>
__global__ __launch_bounds__(64 * mm_waves(W), W == 2 ? (mm_waves(W) == 8 ? GORSE_MM_WGS8 : 4) : 2) void sgemm_mfma_kernel(int m, int n, int k, const float *__restrict__ a, int lda,
                                                                         const float *__restrict__ b, int ldb, float *__restrict__ c,
                                                                         int ldc) {
    constexpr int T = 64 * W;       // tile edge
    constexpr int NWV = mm_waves(W), NT = 64 * NWV;
    constexpr int CW = NWV == 8 ? W / 2 : W;  // 32-column blocks per wave (eight waves: each takes half of a quarter's columns)
    constexpr int PT = T * kMmKB / NT;  // elements of each operand tile per thread (8 or 4)
    __shared__ float As[2][T * kMmAs];
    __shared__ float Bs[2][kMmKB * T];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = blockIdx.y * T, j0 = blockIdx.x * T;
    const int qv = NWV == 8 ? wv >> 1 : wv;  // the wave's quarter inside the tile, and (eight waves) its half of the quarter's columns
    const int wi = 32 * W * (qv >> 1), wj = 32 * W * (qv & 1) + (NWV == 8 ? 32 * CW * (wv & 1) : 0);
    (void)m, (void)n;
    uint32_t c_lane = ((uint32_t)(i0 + wi + 4 * (lane >> 5)) * (uint32_t)ldc + (uint32_t)(j0 + wj + (lane & 31))) * 4u;
    auto c_off = [&](int bi, int bj, int r) {
        return c_lane + ((uint32_t)(32 * bi + (r & 3) + 8 * (r >> 2)) * (uint32_t)ldc + (uint32_t)(32 * bj)) * 4u;
    };
    f32x16 acc[W][CW];
#pragma unroll
    for (int bi = 0; bi < W; bi++)
#pragma unroll
        for (int bj = 0; bj < CW; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++)  // (C spans less than 4 GB: gorse_hip_sgemm -- one 32-bit offset per element from the scalar base)
                acc[bi][bj][r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(c) + c_off(bi, bj, r));
    const int k2 = k & ~1;  // the l that come in pairs
    // the thread's first element of the A block is (ati, atl), of the B block (bul, buj); the other PT - 1 follow in storage order
    const int xa = tid * PT;
    const int ati = TA ? xa % T : xa / kMmKB, atl = TA ? xa / T : xa % kMmKB;
    const int buj = TB ? xa / kMmKB : xa % T, bul = TB ? xa % kMmKB : xa / T;
    const float *pa = TA ? a + (int64_t)atl * lda + (i0 + ati) : a + (int64_t)(i0 + ati) * lda + atl;
    const float *pb = TB ? b + (int64_t)(j0 + buj) * ldb + bul : b + (int64_t)bul * ldb + (j0 + buj);
    const int64_t sa = TA ? (int64_t)kMmKB * lda : kMmKB, sb = TB ? kMmKB : (int64_t)kMmKB * ldb;  // one block of l further
    float ra[PT], rb[PT];
    auto gload = [&]() {
#pragma unroll
        for (int e = 0; e < PT; e++) ra[e] = pa[e], rb[e] = pb[e];
        pa += sa, pb += sb;
    };
    auto lstore = [&](int buf) {  // As[i][l] (stride kMmAs), Bs[l][j] (stride T)
#pragma unroll
        for (int e = 0; e < PT; e++) {
            const int ti = TA ? ati + e : ati, tl = TA ? atl : atl + e;
            const int uj = TB ? buj : buj + e, ul = TB ? bul + e : bul;
            As[buf][ti * kMmAs + tl] = ra[e];
            Bs[buf][ul * T + uj] = rb[e];
        }
    };
    if (k2 > 0) {
        gload();
        lstore(0);
        __syncthreads();
    }
    int buf = 0;
    for (int l0 = 0; l0 < k2; l0 += kMmKB) {
        const bool more = l0 + kMmKB < k2;
        if (more) gload();
        const int steps = min(kMmKB, k2 - l0);  // even
        const float *as = As[buf], *bs = Bs[buf];
        for (int kk = 0; kk < steps; kk += 2) {
            float fa[W], fb[CW];
#pragma unroll
            for (int bi = 0; bi < W; bi++) fa[bi] = as[(wi + 32 * bi + (lane & 31)) * kMmAs + kk + (lane >> 5)];
#pragma unroll
            for (int bj = 0; bj < CW; bj++) fb[bj] = bs[(kk + (lane >> 5)) * T + wj + 32 * bj + (lane & 31)];
#pragma unroll
            for (int bi = 0; bi < W; bi++)
#pragma unroll
                for (int bj = 0; bj < CW; bj++) acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[bi], fb[bj], acc[bi][bj], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);  // (the other buffer: everybody left it before the barrier of the block before)
        __syncthreads();
        buf ^= 1;
    }
    // (the lane's offset goes through an opaque move: the 64 element offsets are then formed again here instead of being kept in 64
    // registers from the loads at the top to these stores -- 162 spilled registers at four waves per SIMD)
    asm volatile("" : "+v"(c_lane));
#pragma unroll
    for (int bi = 0; bi < W; bi++)
#pragma unroll
        for (int bj = 0; bj < CW; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                *reinterpret_cast<float *>(reinterpret_cast<char *>(c) + c_off(bi, bj, r)) = acc[bi][bj][r];
}

// an odd last l: one fmaf per element after the pairs -- never a padded zero step in the MFMA loop (it would turn an accumulated -0
// into +0).  Its own launch: inside the tile kernel its 64 operand loads per thread cost the registers of a fourth wave per SIMD.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_last_step_kernel(int m, int n, int l, const float *__restrict__ a, int lda,
                                                              const float *__restrict__ b, int ldb, float *__restrict__ c, int ldc) {
    const int64_t total = (int64_t)m * n;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / n), j = (int)(t % n);
        const float av = TA ? a[(int64_t)l * lda + i] : a[(int64_t)i * lda + l];
        const float bv = TB ? b[(int64_t)j * ldb + l] : b[(int64_t)l * ldb + j];
        c[(int64_t)i * ldc + j] = fmaf(av, bv, c[(int64_t)i * ldc + j]);
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

namespace {
// the pair of events around the kernel(s) of one call: released on every path out of it
struct EventPair {
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    EventPair() {
        (void)hipEventCreate(&ev0);
        (void)hipEventCreate(&ev1);
    }
    EventPair(const EventPair &) = delete;
    EventPair &operator=(const EventPair &) = delete;
    ~EventPair() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

int32_t sgemm_check(int32_t device, int32_t transA, int32_t transB, int32_t m, int32_t n, int32_t k, const void *a, int32_t lda,
                    const void *b, int32_t ldb, const void *c, int32_t ldc) {
    if (m < 0 || n < 0 || k < 0) return fail(GORSE_ERR_INVALID, "negative dimension");
    if (m == 0 || n == 0) return GORSE_OK;
    if (!a || !b || !c) return fail(GORSE_ERR_INVALID, "NULL matrix");
    const int a_cols = transA ? m : k, b_cols = transB ? k : n;
    if (lda < a_cols || ldb < b_cols || ldc < n) return fail(GORSE_ERR_INVALID, "leading dimension too small");
    if (k > 8192 && !transA && transB) return fail(GORSE_ERR_INVALID, "k %d > 8192 unsupported in the NT case", k);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GORSE_ERR_NO_DEVICE, "no HIP device visible (libgorse_hip needs an MI355X / gfx950)");
    if (device < 0 || device >= ndev) return fail(GORSE_ERR_INVALID, "device %d out of range", device);
    GORSE_HIP_CHECK(hipSetDevice(device));
    return GORSE_OK;
}

// C (+)= A B on matrices that lie in device memory; everything is enqueued on st, the events of `ev` bracket the kernel(s)
int32_t sgemm_on_device(hipStream_t st, EventPair &ev, int32_t transA, int32_t transB, int32_t m, int32_t n, int32_t k, const float *da,
                        int32_t lda, const float *db, int32_t ldb, float *dc, int32_t ldc) {
    const int a_rows = transA ? k : m, a_cols = transA ? m : k;
    const int b_rows = transB ? n : k, b_cols = transB ? k : n;
    (void)hipEventRecord(ev.ev0, st);
    if (!transA && transB) {
        int64_t blocks = std::min<int64_t>(ceil_div((int64_t)m * n, kGroupsPerBlock), 8192);
        sgemm_nt_kernel<<<dim3((unsigned)blocks), dim3(kBlock), (size_t)kGroupsPerBlock * 2 * std::max(k, 1) * 4, st>>>(
            m, n, k, da, lda, db, ldb, dc, ldc);
    } else if (k > 0 && (int64_t)m * n >= 64 * 64 && (int64_t)(m + 128) * (n + 128) * 4 < ((int64_t)1 << 32) && !g_sgemm_valu) {
        // the matrix cores (a tile is 128 x 128: below 64 x 64 the vector ALU form; C below 4 GB: the kernel addresses it by 32-bit offsets)
        // 128 x 128 tiles where they give every CU work (>= 512 of them: two per CU), 64 x 64 tiles otherwise.  The kernel has no edge
        // handling: the operands are re-laid here with m and n rounded up to the tile and the l range allocated up to a multiple of
        // 16 (zeros; the rows and columns past the caller's are computed and dropped, the l past k are read and never multiplied).
        const bool big = (int64_t)ceil_div(n, 128) * ceil_div(m, 128) >= 512;
        const int T = big ? 128 : 64;
        const int mp = (int)ceil_div(m, T) * T, np = (int)ceil_div(n, T) * T, kp = (int)ceil_div(k, kMmKB) * kMmKB;
        const int pa_rows = transA ? kp : mp, pa_cols = transA ? mp : kp, pb_rows = transB ? np : kp, pb_cols = transB ? kp : np;
        // (operands that are whole tiles already are used where they lie)
        const bool whole = mp == m && np == n && kp == k && (int64_t)m * ldc * 4 < ((int64_t)1 << 32);
        DevBuf<float> pa, pb, pc;
        const float *ka = da, *kb = db;
        float *kc = dc;
        int klda = lda, kldb = ldb, kldc = ldc;
        if (!whole) {
            GORSE_TRY(pa.alloc((size_t)pa_rows * pa_cols));
            GORSE_TRY(pb.alloc((size_t)pb_rows * pb_cols));
            GORSE_TRY(pc.alloc((size_t)mp * np));
            GORSE_HIP_CHECK(hipMemsetAsync(pa.p, 0, (size_t)pa_rows * pa_cols * 4, st));
            GORSE_HIP_CHECK(hipMemsetAsync(pb.p, 0, (size_t)pb_rows * pb_cols * 4, st));
            GORSE_HIP_CHECK(hipMemsetAsync(pc.p, 0, (size_t)mp * np * 4, st));
            GORSE_HIP_CHECK(hipMemcpy2DAsync(pa.p, (size_t)pa_cols * 4, da, (size_t)lda * 4, (size_t)a_cols * 4, a_rows, hipMemcpyDeviceToDevice, st));
            GORSE_HIP_CHECK(hipMemcpy2DAsync(pb.p, (size_t)pb_cols * 4, db, (size_t)ldb * 4, (size_t)b_cols * 4, b_rows, hipMemcpyDeviceToDevice, st));
            GORSE_HIP_CHECK(hipMemcpy2DAsync(pc.p, (size_t)np * 4, dc, (size_t)ldc * 4, (size_t)n * 4, m, hipMemcpyDeviceToDevice, st));
            ka = pa.p, kb = pb.p, kc = pc.p, klda = pa_cols, kldb = pb_cols, kldc = np;
        }
        (void)hipEventRecord(ev.ev0, st);  // (the figure of gorse_hip_test_sgemm_last_ms: the kernel alone)
        dim3 grid((unsigned)(np / T), (unsigned)(mp / T)), block(256);
#define MM(TA_, TB_)                                                                                                   \
    do {                                                                                                               \
        if (big)                                                                                                       \
            sgemm_mfma_kernel<TA_, TB_, 2><<<grid, dim3(64 * mm_waves(2)), 0, st>>>(mp, np, k, ka, klda, kb, kldb, kc, kldc); \
        else                                                                                                           \
            sgemm_mfma_kernel<TA_, TB_, 1><<<grid, block, 0, st>>>(mp, np, k, ka, klda, kb, kldb, kc, kldc);          \
    } while (0)
        if (!transA && !transB)
            MM(false, false);
        else if (transA && !transB)
            MM(true, false);
        else
            MM(true, true);
#undef MM
        if (k & 1) {
            const unsigned lb = (unsigned)std::min<int64_t>(ceil_div((int64_t)mp * np, 256), 4096);
            if (!transA && !transB)
                sgemm_last_step_kernel<false, false><<<dim3(lb), dim3(256), 0, st>>>(mp, np, k - 1, ka, klda, kb, kldb, kc, kldc);
            else if (transA && !transB)
                sgemm_last_step_kernel<true, false><<<dim3(lb), dim3(256), 0, st>>>(mp, np, k - 1, ka, klda, kb, kldb, kc, kldc);
            else
                sgemm_last_step_kernel<true, true><<<dim3(lb), dim3(256), 0, st>>>(mp, np, k - 1, ka, klda, kb, kldb, kc, kldc);
        }
        (void)hipEventRecord(ev.ev1, st);
        GORSE_HIP_CHECK(hipGetLastError());
        if (!whole) {
            GORSE_HIP_CHECK(hipMemcpy2DAsync(dc, (size_t)ldc * 4, pc.p, (size_t)np * 4, (size_t)n * 4, m, hipMemcpyDeviceToDevice, st));
            GORSE_HIP_CHECK(hipStreamSynchronize(st));  // pa / pb / pc are released at the end of this scope
        }
        return GORSE_OK;
    } else if (k > 0) {
        dim3 grid((unsigned)ceil_div(n, TS), (unsigned)ceil_div(m, TS)), block(TS * TS);
        if (!transA && !transB)
            sgemm_chain_kernel<false, false><<<grid, block, 0, st>>>(m, n, k, da, lda, db, ldb, dc, ldc);
        else if (transA && !transB)
            sgemm_chain_kernel<true, false><<<grid, block, 0, st>>>(m, n, k, da, lda, db, ldb, dc, ldc);
        else
            sgemm_chain_kernel<true, true><<<grid, block, 0, st>>>(m, n, k, da, lda, db, ldb, dc, ldc);
    }
    GORSE_HIP_CHECK(hipGetLastError());
    (void)hipEventRecord(ev.ev1, st);
    return GORSE_OK;
}
}  // namespace

extern "C" int32_t gorse_hip_sgemm(int32_t device, int32_t transA, int32_t transB, int32_t m, int32_t n, int32_t k,
                                   const float *a, int32_t lda, const float *b, int32_t ldb, float *c, int32_t ldc) {
    GORSE_TRY(sgemm_check(device, transA, transB, m, n, k, a, lda, b, ldb, c, ldc));
    if (m == 0 || n == 0) return GORSE_OK;
    const int a_rows = transA ? k : m, a_cols = transA ? m : k;
    const int b_rows = transB ? n : k, b_cols = transB ? k : n;
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
    EventPair ev;
    GORSE_TRY(sgemm_on_device(st, ev, transA, transB, m, n, k, da.p, lda, db.p, ldb, dc.p, ldc));
    GORSE_HIP_CHECK(hipMemcpyAsync(c, dc.p, nc * 4, hipMemcpyDeviceToHost, st));
    GORSE_HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev.ev0, ev.ev1) == hipSuccess) g_sgemm_last_ms = ms;
    return GORSE_OK;
}

// The same product on matrices that already lie in the memory of `device` (a caller that keeps its operands resident -- the
// reference's common/nn layers call floats.MM in a loop -- pays no PCIe transfer: 13 ms of them around a 1.2 ms kernel at 4096^3).
// Synchronous: the result is complete when the call returns.
extern "C" int32_t gorse_hip_sgemm_device(int32_t device, int32_t transA, int32_t transB, int32_t m, int32_t n, int32_t k,
                                          const float *a_dev, int32_t lda, const float *b_dev, int32_t ldb, float *c_dev, int32_t ldc) {
    GORSE_TRY(sgemm_check(device, transA, transB, m, n, k, a_dev, lda, b_dev, ldb, c_dev, ldc));
    if (m == 0 || n == 0) return GORSE_OK;
    hipStream_t st = nullptr;
    EventPair ev;
    GORSE_TRY(sgemm_on_device(st, ev, transA, transB, m, n, k, a_dev, lda, b_dev, ldb, c_dev, ldc));
    GORSE_HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev.ev0, ev.ev1) == hipSuccess) g_sgemm_last_ms = ms;
    return GORSE_OK;
}
