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
#include <cstdlib>
#include <cstring>

#include "als_plan.hpp"
#include "mf_internal.hpp"
#include <type_traits>

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
                // (an element past d * d accumulates element d * d - 1 and is never stored: a bound check here is 64 masks that
                // hipcc keeps in scalar registers across the row loop -- 65 spills)
                const int idx = threadIdx.x + e * 256 < dd ? threadIdx.x + e * 256 : dd - 1;
                acc[e] += x[idx / d] * x[idx % d];
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

__global__ void als_gram_reduce_kernel(const float *__restrict__ partial, int nparts, int dd, float *__restrict__ S,
                                       int64_t stride) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= dd) return;
    float s = 0.0f;
    int g = 0;
    for (; g + 15 < nparts; g += 16) {  // sixteen loads in flight, added in the same fixed order (one load at a time: 128 us for C5's 489 partials)
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = partial[(size_t)(g + j) * stride + e];
#pragma unroll
        for (int j = 0; j < 16; j++) s += x[j];
    }
    for (; g < nparts; g++) s += partial[(size_t)g * stride + e];  // fixed order: deterministic
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
                                                        int64_t row_begin, int64_t rows, int d, float w, float reg,
                                                        int pred_cap, int q_cap, float *__restrict__ scratch,
                                                        int64_t scratch_stride, int min_n) {
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
    for (int64_t u = row_begin + blockIdx.x; u < rows; u += gridDim.x) {  // rows = end of the row range
        const int64_t beg = ptr[u];
        const int n = (int)(ptr[u + 1] - beg);
        if (n < min_n) continue;  // a row of als_wide_kernel's
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


// ---- Gram-form row solve for 64 < nFactors <= 128 ------------------------------------------------------------------------
// The same substitution as below (M = (1 - w) G + w S per row, one Gauss-Seidel sweep), for factor widths whose M no longer
// fits one wave's registers as columns.  One 256-thread workgroup per row (or per chunk of a long row): G = sum q q^T by fused
// multiply-adds, thread (ti, tj) owning the 8 x 8 block of rows 8 ti .. and columns 8 tj .. (64 accumulators), the entries
// staged 16 at a time through LDS with the next batch's gathers in flight; M goes to LDS (128 x 129 floats, zero past d: a
// padded coordinate solves to 0 and changes nothing), and one wave runs the sweep with two coordinates per lane and the
// columns read from LDS.  A row of more than 4096 entries is cut into chunks (the row plan of the d <= 64 form): every chunk
// leaves its G and column sums in global memory, als_wide_long_kernel adds them in chunk order and solves.
constexpr int kWideLd = 129, kWideBatch = 16, kWidePartial = 128 * 128 + 128;  // floats per chunk: G row-major, then the sums
typedef float f32x16 __attribute__((ext_vector_type(16)));

// G and the column sums of entries idx[beg .. beg + n) of B, in this thread's 8 x 8 block / (ti == 0) 8 columns
__device__ __forceinline__ void wide_accumulate(const float *__restrict__ B, const int32_t *__restrict__ idx, int64_t beg, int n,
                                                int d, float *sq, float (&acc)[8][8], float (&cs)[8]) {
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int lr = tid >> 4, lc = (tid & 15) * 8;  // gather: entry lr of the batch, columns lc .. lc + 7
    float g[8];
#pragma unroll
    for (int a = 0; a < 8; a++) {
        cs[a] = 0.0f;
#pragma unroll
        for (int b = 0; b < 8; b++) acc[a][b] = 0.0f;
    }
    auto gather = [&](int e0) {
        const int e = e0 + lr;
        const float *row = B + (int64_t)idx[beg + (e < n ? e : 0)] * d;
#pragma unroll
        for (int i = 0; i < 8; i++) g[i] = (e < n && lc + i < d) ? row[lc + i < d ? lc + i : 0] : 0.0f;
    };
    if (n > 0) gather(0);
    for (int e0 = 0; e0 < n; e0 += kWideBatch) {
#pragma unroll
        for (int i = 0; i < 8; i++) sq[lr * 128 + lc + i] = g[i];
        __syncthreads();
        if (e0 + kWideBatch < n) gather(e0 + kWideBatch);
        const int m = n - e0 < kWideBatch ? n - e0 : kWideBatch;
        for (int e = 0; e < m; e++) {
            float r[8], c[8];
#pragma unroll
            for (int a = 0; a < 8; a++) {
                r[a] = sq[e * 128 + 8 * ti + a];
                c[a] = sq[e * 128 + 8 * tj + a];
            }
#pragma unroll
            for (int a = 0; a < 8; a++)
#pragma unroll
                for (int b = 0; b < 8; b++) acc[a][b] = fmaf(r[a], c[b], acc[a][b]);
            if (ti == 0) {
#pragma unroll
                for (int b = 0; b < 8; b++) cs[b] += c[b];
            }
        }
        __syncthreads();  // the batch is consumed: sq may be rewritten
    }
}

__device__ __forceinline__ void wide_sweep(float *__restrict__ a, int d, float reg, const float *sM, const float *ss,
                                           const float *sy = nullptr);

// M = (1 - w) G + w S into LDS, then the sweep of row `a` by the first wave; ends with a barrier (sM / ss are free again)
__device__ __forceinline__ void wide_solve(float *__restrict__ a, const float *__restrict__ S, int d, float w, float reg,
                                           float *sM, float *ss, const float (&acc)[8][8], const float (&cs)[8]) {
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const float one_w = 1 - w;
    // (the 64 bound checks below are invariant across the callers' row loops; with d as written hipcc keeps their masks in scalar
    // registers over the loop -- 86 / 100 spills in als_wide_long_kernel / als_wide_kernel<false, 0>: an opaque copy of d forms them here)
    int dq = d;
    asm volatile("" : "+s"(dq));
#pragma unroll
    for (int x = 0; x < 8; x++)
#pragma unroll
        for (int y = 0; y < 8; y++) {
            const int i = 8 * ti + x, j = 8 * tj + y;
            sM[i * kWideLd + j] = (i < dq && j < dq) ? one_w * acc[x][y] + w * S[i * d + j] : 0.0f;
        }
    if (ti == 0) {
#pragma unroll
        for (int y = 0; y < 8; y++) ss[8 * tj + y] = 8 * tj + y < d ? cs[y] : 0.0f;
    }
    __syncthreads();
    wide_sweep(a, d, reg, sM, ss);
}

// the Gauss-Seidel sweep of row `a` over M (LDS, 128 x kWideLd, zero past d) and the column sums ss by the first wave, two
// coordinates per lane; ends with a barrier (sM / ss are free again)
// sy != null: y = M p (the row's current factors) has been formed by the whole workgroup (wide_form_y: two partial sums per
// coordinate, even and odd rows of M, at sy[k] and sy[128 + k]) with p at sy[256 + k]; else this wave forms it here.
__device__ __forceinline__ void wide_sweep(float *__restrict__ a, int d, float reg, const float *sM, const float *ss, const float *sy) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        // the chain's operations must not queue behind the 64-cycle MFMAs of the other workgroup's wave on this SIMD (as in
        // als_row_kernel: at equal priority the arbiter lets one of them through per MFMA)
        __builtin_amdgcn_s_setprio(3);
        const int lane = tid, k0 = lane, k1 = lane + 64;
        const float p0_lo = sy ? sy[256 + k0] : (k0 < d ? a[k0] : 0.0f), p0_hi = sy ? sy[256 + k1] : (k1 < d ? a[k1] : 0.0f);
        const float diag_lo = sM[k0 * kWideLd + k0], diag_hi = sM[k1 * kWideLd + k1];
        // a padded coordinate (k >= d: zero row and column of M) keeps inv = 0: it solves to 0 whatever reg is
        const float inv_lo = k0 < d ? __builtin_amdgcn_rcpf(diag_lo + reg) : 0.0f;
        const float inv_hi = k1 < d ? __builtin_amdgcn_rcpf(diag_hi + reg) : 0.0f;
        const float base_lo = (ss[k0] + p0_lo * diag_lo) * inv_lo, base_hi = (ss[k1] + p0_hi * diag_hi) * inv_hi;
        float y_lo = 0.0f, y_hi = 0.0f;
        auto lane_of = [](float v, int f) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), f)); };
        // (the two halves of the coordinates are separate loops: which register a step's scalars come from is then known
        // at compile time instead of being a uniform branch in every step)
        if (sy) {
            y_lo = sy[k0] + sy[128 + k0];
            y_hi = sy[k1] + sy[128 + k1];
        } else {
#pragma unroll 4
            for (int f = 0; f < 64; f++) {
                const float pf = lane_of(p0_lo, f);
                y_lo = fmaf(sM[k0 * kWideLd + f], pf, y_lo);
                y_hi = fmaf(sM[k1 * kWideLd + f], pf, y_hi);
            }
#pragma unroll 4
            for (int f = 0; f < 64; f++) {
                const float pf = lane_of(p0_hi, f);
                y_lo = fmaf(sM[k0 * kWideLd + 64 + f], pf, y_lo);
                y_hi = fmaf(sM[k1 * kWideLd + 64 + f], pf, y_hi);
            }
        }
        // The sweep is one dependent chain through y; per step it is kept to three operations: every lane forms its own
        // would-be step delta_k = (base_k - p_k) - y_k inv_k (one fused operation on the chain), lane f's is broadcast
        // (v_readlane), and y += delta_f M[:, f].  The new coordinate p_f' = base_f - y_f inv_f is formed beside the chain, the
        // columns of M for the next eight steps are read from LDS while the current eight run.  (profiles/r03_zc_probe_als_wide.txt: 22.5K
        // cycles per row = 176 per step with base, y, inv and p0 of coordinate f broadcast separately in every step.)
        // (Round 6: the new coordinate is no longer formed in every step -- a multiply, a subtraction, a compare and a select beside
        // the chain's three operations.  A lane keeps y as it stands at ITS OWN step, one compare and one select per step; every lane
        // has exactly one step per half, so p' = base - y inv is formed once behind the half, the same two roundings.)
        const float g_lo = base_lo - p0_lo, g_hi = base_hi - p0_hi;
        float ys_lo = 0.0f, ys_hi = 0.0f;
        auto sweep_half = [&](const int hi) {
            const float *c0 = sM + k0 * kWideLd + 64 * hi, *c1 = sM + k1 * kWideLd + 64 * hi;
            float cl[8], ch[8];
#pragma unroll
            for (int j = 0; j < 8; j++) cl[j] = c0[j], ch[j] = c1[j];
#pragma unroll 1
            for (int fb = 0; fb < 64; fb += 8) {
                float nl[8], nh[8];
                const int nb = fb + 8 < 64 ? fb + 8 : fb;
#pragma unroll
                for (int j = 0; j < 8; j++) nl[j] = c0[nb + j], nh[j] = c1[nb + j];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int f = fb + j;
                    const float yk = hi ? y_hi : y_lo;
                    const float dk = fmaf(-yk, hi ? inv_hi : inv_lo, hi ? g_hi : g_lo);
                    const float delta = lane_of(dk, f);
                    y_lo = fmaf(delta, cl[j], y_lo);
                    y_hi = fmaf(delta, ch[j], y_hi);
                    if (hi)
                        ys_hi = lane == f ? yk : ys_hi;
                    else
                        ys_lo = lane == f ? yk : ys_lo;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) cl[j] = nl[j], ch[j] = nh[j];
            }
        };
        sweep_half(0);
        sweep_half(1);
        const float p_lo = base_lo - ys_lo * inv_lo, p_hi = base_hi - ys_hi * inv_hi;
        if (k0 < d) a[k0] = p_lo;
        if (k1 < d) a[k1] = p_hi;
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
}


// ---- three bf16 values per float (the bf16-MFMA forms of the Gram: wide_gram_b3 below, gram_accumulate_b3 further down) -----------
typedef __bf16 als_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 als_bf16x2 __attribute__((ext_vector_type(2)));
typedef float als_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // {bf16(a) in bits 0..15, bf16(b) in bits 16..31}
    const als_f32x2 v = {a, b};
    const als_bf16x2 p = __builtin_convertvector(v, als_bf16x2);
    return *reinterpret_cast<const uint32_t *>(&p);
}
// x = hi + mid + lo exactly (round to nearest even each time); the three are returned as the upper halves of floats
__device__ __forceinline__ void split_pair_bf16(float x0, float x1, uint32_t &ph, uint32_t &pm, uint32_t &pl) {
    // (v_dot2c_f32_bf16 with a (-1, 0) / (0, -1) operand would subtract a packed half in one instruction instead of a shift or mask and
    // a subtraction; as hipcc emits it from __builtin_amdgcn_fdot2_f32_bf16 the results were wrong, and the epoch no faster: r04_zh)
    ph = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(ph << 16), r1 = x1 - __uint_as_float(ph & 0xffff0000u);
    pm = pack_bf16(r0, r1);
    const float l0 = r0 - __uint_as_float(pm << 16), l1 = r1 - __uint_as_float(pm & 0xffff0000u);
    pl = pack_bf16(l0, l1);
}

// ---- G of a wide row on the fp32 MFMA ---------------------------------------------------------------------------------
// The 128 x 128 Gram matrix is 4 x 4 blocks of 32 x 32; its upper triangle is ten v_mfma_f32_32x32x2_f32 tiles, shared out
// among the four waves of the workgroup 3 + 3 + 2 + 2 (the diagonal blocks' waves also keep the column sums of their two
// column blocks).  The gathered rows go through LDS 32 entries at a time -- two buffers inside the region that holds M
// afterwards, so one barrier per batch: a buffer is rewritten two batches after it was read -- and an MFMA consumes two
// entries: lane (half, col) supplies q[entry 2 p + half][32 b + col] as the A operand of block b and as the B operand.
// 192 MFMA cycles per entry pair on the busiest SIMD against ~512 cycles of fused multiply-adds per pair and thread before.
__device__ __forceinline__ constexpr int wide_tiles(int wv) { return wv < 2 ? 3 : 2; }
__device__ __forceinline__ constexpr int wide_tile_bi(int wv, int t) { return wv == 0 ? (t == 2 ? 1 : 0) : (wv == 1 ? (t == 2 ? 3 : 2) : wv - 2); }
__device__ __forceinline__ constexpr int wide_tile_bj(int wv, int t) { return wv == 0 ? (t == 0 ? 0 : 1) : (wv == 1 ? (t == 0 ? 2 : 3) : 2 + t); }
constexpr int kWideMfmaBatch = 32;

// A thread's share of the gather pipeline: the batch that is consumed next (two entries x eight columns) and the row ids of the
// batch after it.  It lives across rows: a row's first batch is gathered while the row before it is still being accumulated,
// formed and swept -- a gather is two dependent reads (entry -> row id -> row), ~10K cycles in front of a 100-entry row's
// ~12K cycles of MFMA when it starts cold (profiles/r03_zb_probe_als_wide.txt: 44K cycles of accumulation per such row).
struct WideStage {
    float g[2][8];
    int ix[2];
};
__device__ __forceinline__ void wide_load_ids(const int32_t *__restrict__ idx, int64_t beg, int n, int e0, int (&ix)[2]) {
    const int lr = threadIdx.x >> 4;
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
        const int e = e0 + lr + 16 * h2;
        ix[h2] = idx[beg + (e < n ? e : 0)];
    }
}
__device__ __forceinline__ void wide_gather(const float *__restrict__ B, int n, int e0, int d, const int (&ix)[2], float (&g)[2][8]) {
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 8;  // entries lr and lr + 16 of the batch, columns lc .. lc + 7
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
        const int e = e0 + lr + 16 * h2;
        const float *row = B + (int64_t)ix[h2] * d;
#pragma unroll
        for (int i = 0; i < 8; i++) g[h2][i] = (e < n && lc + i < d) ? row[lc + i < d ? lc + i : 0] : 0.0f;
    }
}
// the pipeline's state for a workgroup's first item
__device__ __forceinline__ void wide_first_batch(const float *__restrict__ B, const int32_t *__restrict__ idx, int64_t beg, int n, int d,
                                                 WideStage &st) {
    wide_load_ids(idx, beg, n, 0, st.ix);
    wide_gather(B, n, 0, d, st.ix, st.g);
    wide_load_ids(idx, beg, n, kWideMfmaBatch, st.ix);
}

// G of item (beg, n) into this wave's tiles; on entry st holds the item's first batch and the ids of its second, on exit those of
// the next item (nbeg, nn)
template <int W>
__device__ __forceinline__ void wide_gram_mfma(const float *__restrict__ B, const int32_t *__restrict__ idx, int64_t beg, int n,
                                               int64_t nbeg, int nn, int d, float *sbuf, WideStage &st, f32x16 (&tl)[3],
                                               float (&cs)[2]) {
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int lr = tid >> 4, lc = (tid & 15) * 8;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) tl[t][r] = 0.0f;
    cs[0] = cs[1] = 0.0f;
    int ixn[2];  // the next item's first ids: read now, used when this item's last batch is being multiplied
    wide_load_ids(idx, nbeg, nn, 0, ixn);
    auto next_item = [&]() {
        wide_gather(B, nn, 0, d, ixn, st.g);
        wide_load_ids(idx, nbeg, nn, kWideMfmaBatch, st.ix);
    };
    int buf = 0;
    for (int e0 = 0; e0 < n; e0 += kWideMfmaBatch) {
        float *sq = sbuf + buf * (kWideMfmaBatch * 128);
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {  // (16-byte stores: as dwords the eight-column shares of 16 lanes met in four banks)
            float4 *dst = reinterpret_cast<float4 *>(sq + (lr + 16 * h2) * 128 + lc);
            dst[0] = make_float4(st.g[h2][0], st.g[h2][1], st.g[h2][2], st.g[h2][3]);
            dst[1] = make_float4(st.g[h2][4], st.g[h2][5], st.g[h2][6], st.g[h2][7]);
        }
        __syncthreads();
        if (e0 + kWideMfmaBatch < n) {
            wide_gather(B, n, e0 + kWideMfmaBatch, d, st.ix, st.g);
            wide_load_ids(idx, beg, n, e0 + 2 * kWideMfmaBatch, st.ix);
        } else {
            next_item();
        }
        const int m = n - e0 < kWideMfmaBatch ? n - e0 : kWideMfmaBatch;
        const int pairs = (m + 1) >> 1;  // (an odd batch end: the second entry of the last pair is a row of zeros)
        const float *mine = sq + half * 128 + col;
#pragma unroll 2
        for (int p = 0; p < pairs; p++) {
            float fr[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                bool used = false;
#pragma unroll
                for (int t = 0; t < wide_tiles(W); t++) used |= wide_tile_bi(W, t) == b || wide_tile_bj(W, t) == b;
                fr[b] = used ? mine[p * 256 + 32 * b] : 0.0f;
            }
#pragma unroll
            for (int t = 0; t < wide_tiles(W); t++)
                tl[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[wide_tile_bi(W, t)], fr[wide_tile_bj(W, t)], tl[t], 0, 0, 0);
            if (W < 2) {
                cs[0] += fr[2 * W];
                cs[1] += fr[2 * W + 1];
            }
        }
        buf ^= 1;
    }
    if (n <= 0) next_item();
    __syncthreads();  // the batches are consumed: the region is M's now
}

// ---- the same on the bf16 MFMA over three-way split values (gram_accumulate_b3 has the arithmetic) ------------------------------------
// The split is done ONCE, by the thread that gathers: thread (column c = tid & 127, half eg = tid >> 7) gathers column c of the sixteen
// entries 16 eg .. 16 eg + 15 of a 32-entry batch (a wave reads 256 consecutive bytes of one row per load), keeps the column's sum,
// splits its sixteen floats and stores the three planes as bf16, entry-contiguous: plane p, column c, entries 0..31 = 64 bytes at
// p * 10240 + c * 80 (the 80-byte column stride keeps the 16-byte stores and fragment reads of sixteen neighbouring columns on
// different banks).  A wave's MFMA fragment -- column 32 b + col, entries 8 half .. 8 half + 7 of a 16-entry group -- is ONE 16-byte
// read.  Two buffers of 30 KB inside M's region; six MFMAs of 32 cycles per tile and 16 entries against eight of 64 cycles.
struct WideStageB3 {
    float g[16];  // column c of the sixteen entries of this thread's half of the batch consumed next
    int id;       // lane & 15: the row id of that entry of the batch AFTER it
};
constexpr int kWideB3Col = 80, kWideB3Plane = 128 * kWideB3Col, kWideB3Buf = 3 * kWideB3Plane;  // bytes
__device__ __forceinline__ int wide_load_id_b3(const int32_t *__restrict__ idx, int64_t beg, int n, int e0) {
    const int e = e0 + 16 * (threadIdx.x >> 7) + (threadIdx.x & 15);
    return idx[beg + (e < n ? e : 0)];
}
__device__ __forceinline__ void wide_gather_b3(const float *__restrict__ B, int n, int e0, int d, int id, float (&g)[16]) {
    const int c = threadIdx.x & 127, eb = e0 + 16 * (threadIdx.x >> 7);
    const uint32_t coff = (uint32_t)(c < d ? c : 0) * 4u, rowbytes = (uint32_t)d * 4u;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        // the wave's sixteen entries: lanes 0..15 hold their ids.  Through the LDS crossbar and as a 32-bit offset from the scalar
        // base (the form is only taken for factor matrices below 4 GB and 2^24 rows): as scalars -- v_readlane, a 64-bit multiply
        // and add per load -- the sixteen addresses cost the kernel its scalar registers (323 spills)
        const uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * j, id);
        // RAW: an entry past the row's end / a column past d holds some other element -- masked where the batch is consumed (a
        // select here is a use of the load, and the wait for it lands in front of the MFMAs the gather is meant to run under)
        g[j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + (__umul24(r, rowbytes) + coff));
    }
    (void)eb;
    (void)n;
}
__device__ __forceinline__ void wide_first_batch_b3(const float *__restrict__ B, const int32_t *__restrict__ idx, int64_t beg, int n, int d,
                                                    WideStageB3 &st) {
    st.id = wide_load_id_b3(idx, beg, n, 0);
    wide_gather_b3(B, n, 0, d, st.id, st.g);
    st.id = wide_load_id_b3(idx, beg, n, kWideMfmaBatch);
}
// G of item (beg, n) into this wave's tiles, the column sums into `sums` (128 floats: LDS ss, or the chunk's partial in global
// memory); on entry st holds the item's first batch and the ids of its second, on exit those of the next item (nbeg, nn)
template <int W>
__device__ __forceinline__ void wide_gram_b3(const float *__restrict__ B, const int32_t *__restrict__ idx, int64_t beg, int n,
                                             int64_t nbeg, int nn, int d, float *sbuf, float *ss, float *__restrict__ sums,
                                             WideStageB3 &st, f32x16 (&tl)[3]) {
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int c = tid & 127, eg = tid >> 7;
    unsigned char *base = reinterpret_cast<unsigned char *>(sbuf);
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) tl[t][r] = 0.0f;
    float colsum = 0.0f;
    int idn = wide_load_id_b3(idx, nbeg, nn, 0);  // the next item's first ids: read now, used when this item's last batch is multiplied
    auto next_item = [&]() {
        wide_gather_b3(B, nn, 0, d, idn, st.g);
        st.id = wide_load_id_b3(idx, nbeg, nn, kWideMfmaBatch);
    };
    int buf = 0;
    for (int e0 = 0; e0 < n; e0 += kWideMfmaBatch) {
        unsigned char *sq = base + buf * kWideB3Buf;
        {
            uint32_t ph[8], pm[8], pl[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int e = e0 + 16 * eg + 2 * j;
                const float x0 = (e < n && c < d) ? st.g[2 * j] : 0.0f, x1 = (e + 1 < n && c < d) ? st.g[2 * j + 1] : 0.0f;
                colsum += x0;
                colsum += x1;
                split_pair_bf16(x0, x1, ph[j], pm[j], pl[j]);
            }
            uint4 *dh = reinterpret_cast<uint4 *>(sq + c * kWideB3Col + 32 * eg);
            uint4 *dm = reinterpret_cast<uint4 *>(sq + kWideB3Plane + c * kWideB3Col + 32 * eg);
            uint4 *dl = reinterpret_cast<uint4 *>(sq + 2 * kWideB3Plane + c * kWideB3Col + 32 * eg);
            dh[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            dh[1] = make_uint4(ph[4], ph[5], ph[6], ph[7]);
            dm[0] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
            dm[1] = make_uint4(pm[4], pm[5], pm[6], pm[7]);
            dl[0] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            dl[1] = make_uint4(pl[4], pl[5], pl[6], pl[7]);
        }
        __syncthreads();
        if (e0 + kWideMfmaBatch < n) {
            wide_gather_b3(B, n, e0 + kWideMfmaBatch, d, st.id, st.g);
            st.id = wide_load_id_b3(idx, beg, n, e0 + 2 * kWideMfmaBatch);
        } else {
            next_item();
        }
#pragma unroll
        for (int gq = 0; gq < 2; gq++) {
            if (e0 + 16 * gq < n) {  // (a batch's second half past the row's end: sixteen zero rows, skipped by the whole workgroup)
                als_bf16x8 fh[4], fm[4], fl[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    bool used = false;
#pragma unroll
                    for (int t = 0; t < wide_tiles(W); t++) used |= wide_tile_bi(W, t) == b || wide_tile_bj(W, t) == b;
                    if (used) {
                        const unsigned char *f = sq + (32 * b + col) * kWideB3Col + 32 * gq + 16 * half;
                        fh[b] = *reinterpret_cast<const als_bf16x8 *>(f);
                        fm[b] = *reinterpret_cast<const als_bf16x8 *>(f + kWideB3Plane);
                        fl[b] = *reinterpret_cast<const als_bf16x8 *>(f + 2 * kWideB3Plane);
                    }
                }
#pragma unroll
                for (int pr = 0; pr < 6; pr++)
#pragma unroll
                    for (int t = 0; t < wide_tiles(W); t++) {
                        const int bi = wide_tile_bi(W, t), bj = wide_tile_bj(W, t);
                        const als_bf16x8 &a = pr == 0 ? fm[bi] : (pr == 2 ? fl[bi] : (pr == 4 ? fm[bi] : fh[bi]));
                        const als_bf16x8 &x = pr == 0 ? fm[bj] : (pr == 1 ? fl[bj] : (pr == 3 ? fm[bj] : fh[bj]));
                        tl[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x, tl[t], 0, 0, 0);
                    }
            }
        }
        buf ^= 1;
    }
    if (n <= 0) next_item();
    if (eg == 1) ss[c] = colsum;  // (ss is not read before the barriers that follow the Gram)
    __syncthreads();              // the batches are consumed: the region is M's now
    if (eg == 0) sums[c] = c < d ? colsum + ss[c] : 0.0f;
}

// this wave's tiles -> M = (1 - w) G + w S in LDS (both triangles, zero past d), its column sums -> ss.  w S comes from the wave's
// registers in the layout of its tiles (wide_load_wS_tiles: read once per workgroup, the same S for every row of the half-sweep), so M
// is complete when it is written: the separate pass that added w S to every element in LDS (64 reads and 64 writes per thread and
// row, 27 % of a 100-entry row's time in the phase counters of round 4) is gone, the elements are the same in every bit -- the
// two products are rounded before the sum either way.  (With the loads of S in here -- 96 of them, each its own 64-bit address --
// the kernel needed 300 registers.)
template <int W, bool SUMS = true>  // SUMS = false: the column sums are already in ss (wide_gram_b3)
__device__ __forceinline__ void wide_form_mfma(const f32x16 (&tl)[3], const float (&cs)[2], int d, float w, float *sM, float *ss,
                                               const float (&ws)[64]) {
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const float one_w = 1 - w;
    int o = 0;
#pragma unroll
    for (int t = 0; t < wide_tiles(W); t++) {
        const int bi = wide_tile_bi(W, t), bj = wide_tile_bj(W, t);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * half, j = 32 * bj + col;
            // No `i < d && j < d` select: the gathers zero every column past d (wide_gather, wide_gram_b3), so the rows and columns of G
            // past d ARE zero, and so are those of w S as loaded.  The 160 selects' masks are invariant across the row loop: hipcc kept
            // them in scalar registers over it (311 / 335 spills into vector lanes in als_wide_kernel<false, 2 / 1>).
            const float v = one_w * tl[t][r];
            sM[i * kWideLd + j] = v + ws[o + r];
            if (bi != bj) sM[j * kWideLd + i] = v + ws[o + 16 + r];
        }
        o += bi != bj ? 32 : 16;
    }
    if (SUMS && W < 2) {
#pragma unroll
        for (int b2 = 0; b2 < 2; b2++) {
            const float other = __shfl_xor(cs[b2], 32, 64);
            const int e = 32 * (2 * W + b2) + col;
            if (lane < 32) ss[e] = cs[b2] + other;  // (a column past d: a sum of zeros)
        }
    }
    (void)d;
}

// w S in the layout of wave W's tiles (wide_form_mfma): element r of tile t, and behind it -- for a tile off the diagonal -- the
// mirrored element S[j][i] (S is symmetric as the wide Gram kernels leave it; read anyway: any S gives the M of the formula);
// zero past d.  64 registers in every wave: a diagonal tile takes 16, the others 32.
template <int W>
__device__ __forceinline__ void wide_load_wS_tiles(const float *__restrict__ S, int d, float w, float (&ws)[64]) {
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    int o = 0;
#pragma unroll
    for (int t = 0; t < wide_tiles(W); t++) {
        const int bi = wide_tile_bi(W, t), bj = wide_tile_bj(W, t);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * half, j = 32 * bj + col;
            const bool in = i < d && j < d;
            const float v = S[in ? i * d + j : 0];
            ws[o + r] = in ? w * v : 0.0f;
            if (bi != bj) {
                const float vm = S[in ? j * d + i : 0];
                ws[o + 16 + r] = in ? w * vm : 0.0f;
            }
        }
        o += bi != bj ? 32 : 16;
    }
#pragma unroll
    for (int q = o; q < 64; q++) ws[q] = 0.0f;
}
// the sweep's starting y = M p over the complete M: thread (i0 = tid >> 7, j = tid & 127) sums M[i][j] p_i over the rows i = i0 + 2 q
// of column j (M is symmetric: that is its part of y_j); p is at sy[256 ..], the two partial sums of coordinate j go to sy[j] (even
// rows) and sy[128 + j] (odd rows) -- the sums and their order are those of the pass that also added w S (rounds 3-5)
__device__ __forceinline__ void wide_form_y(const float *sM, float *sy) {
    const int i0 = threadIdx.x >> 7, j = threadIdx.x & 127;
    const float *col = sM + i0 * kWideLd + j;
    const float *p = sy + 256 + i0;
    float y = 0.0f;
#pragma unroll
    for (int q = 0; q < 64; q++) y = fmaf(col[2 * q * kWideLd], p[2 * q], y);
    sy[128 * i0 + j] = y;
}

// this wave's tiles -> the chunk's partial G (row-major 128 x 128, both triangles) and column sums in global memory
template <int W, bool SUMS = true>
__device__ __forceinline__ void wide_partial_mfma(const f32x16 (&tl)[3], const float (&cs)[2], float *__restrict__ dst) {
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int t = 0; t < wide_tiles(W); t++) {
        const int bi = wide_tile_bi(W, t), bj = wide_tile_bj(W, t);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * half, j = 32 * bj + col;
            dst[i * 128 + j] = tl[t][r];
            if (bi != bj) dst[j * 128 + i] = tl[t][r];
        }
    }
    if (SUMS && W < 2) {
#pragma unroll
        for (int b2 = 0; b2 < 2; b2++) {
            const float other = __shfl_xor(cs[b2], 32, 64);
            if (lane < 32) dst[128 * 128 + 32 * (2 * W + b2) + col] = cs[b2] + other;
        }
    }
}

template <int W, bool CHUNKS>
__device__ __forceinline__ void wide_item_mfma(const float *__restrict__ B, const int32_t *__restrict__ idx, int64_t beg, int n,
                                               int64_t nbeg, int nn, int d, WideStage &st, float w, float *sM, float *ss,
                                               float *__restrict__ dst, const float (&ws)[64]) {
    f32x16 tl[3];
    float cs[2];
    wide_gram_mfma<W>(B, idx, beg, n, nbeg, nn, d, sM, st, tl, cs);
    if (CHUNKS)
        wide_partial_mfma<W>(tl, cs, dst);
    else
        wide_form_mfma<W>(tl, cs, d, w, sM, ss, ws);
}

template <int W, bool CHUNKS>
__device__ __forceinline__ void wide_item_b3(const float *__restrict__ B, const int32_t *__restrict__ idx, int64_t beg, int n,
                                             int64_t nbeg, int nn, int d, WideStageB3 &st, float w, float *sM, float *ss,
                                             float *__restrict__ dst, const float (&ws)[64]) {
    f32x16 tl[3];
    const float cs[2] = {0.0f, 0.0f};
    wide_gram_b3<W>(B, idx, beg, n, nbeg, nn, d, sM, ss, CHUNKS ? dst + 128 * 128 : ss, st, tl);
    if (CHUNKS)
        wide_partial_mfma<W, false>(tl, cs, dst);
    else
        wide_form_mfma<W, false>(tl, cs, d, w, sM, ss, ws);
}

// CHUNKS = false: the rows of `rows` (n_items of them), accumulated and solved.  CHUNKS = true: the chunks of the long rows
// (chunk_beg / chunk_cnt, n_items of them), each leaving its G and sums in `partial`.  MFMA = false: the round-2 form (G by
// fused multiply-adds, a thread per 8 x 8 block), kept as the probe's comparison (gorse_hip_test_set_als_path(8)).
template <bool CHUNKS, int MFMA>  // MFMA: 0 = fused multiply-adds, 1 = fp32 MFMA, 2 = bf16 MFMA over three-way split values
__global__ __launch_bounds__(256, 2) void als_wide_kernel(float *__restrict__ A, const float *__restrict__ B,
                                                       const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                       const float *__restrict__ S, const int32_t *__restrict__ rows,
                                                       const int64_t *__restrict__ chunk_beg, const int32_t *__restrict__ chunk_cnt,
                                                       int64_t n_items, int d, float w, float reg, float *__restrict__ partial,
                                                       int probe, unsigned long long *prof) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // probe only (prof != null, MFMA rows): s_memtime ticks of the workgroup's first wave in [0] G, [1] M to LDS + S, [2] sweep;
    // [3] rows, [4] entries, [5] kernel ticks, [6] workgroups
    unsigned long long c_acc = 0, c_m = 0, c_solve = 0, c_rows = 0, c_ent = 0, t_begin = 0;
#ifndef GORSE_PROBE
    prof = nullptr;  // the phase counters exist in `make probe-lib` builds only: here every `if (prof)` folds away
#endif
    if (prof) t_begin = __builtin_amdgcn_s_memtime();
    float *sM = smem;                           // 128 x 129
    float *sq = sM + 128 * kWideLd;             // 16 x 128: one batch of gathered rows
    float *ss = sq + kWideBatch * 128;          // 128 column sums
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    if constexpr (MFMA != 0) {
        auto item = [&](int64_t t, int64_t &u, int64_t &beg, int &n) {
            u = 0, beg = 0, n = 0;
            if (t >= n_items) return;
            if (CHUNKS) {
                beg = chunk_beg[t];
                n = chunk_cnt[t];
            } else {
                u = rows[t];
                beg = ptr[u];
                n = (int)(ptr[u + 1] - beg);
            }
        };
        int64_t u, beg;
        int n;
        item(blockIdx.x, u, beg, n);
        std::conditional_t<MFMA == 2, WideStageB3, WideStage> st;
        if constexpr (MFMA == 2)
            wide_first_batch_b3(B, idx, beg, n, d, st);
        else
            wide_first_batch(B, idx, beg, n, d, st);
        float ws[64];  // (CHUNKS: never read, folds away)
        if constexpr (!CHUNKS) {
            switch (tid >> 6) {
            case 0: wide_load_wS_tiles<0>(S, d, w, ws); break;
            case 1: wide_load_wS_tiles<1>(S, d, w, ws); break;
            case 2: wide_load_wS_tiles<2>(S, d, w, ws); break;
            default: wide_load_wS_tiles<3>(S, d, w, ws); break;
            }
        }
        for (int64_t t = blockIdx.x; t < n_items; t += gridDim.x) {
            int64_t u2, beg2;
            int n2;
            item(t + gridDim.x, u2, beg2, n2);
            float *dst = CHUNKS ? partial + t * kWidePartial : nullptr;
            // the row's current factors: read now, needed (in LDS) when M is complete
            const float pa = (!CHUNKS && tid < d) ? A[u * d + tid] : 0.0f;
            unsigned long long t0 = 0;
            if (prof) t0 = __builtin_amdgcn_s_memtime();
            if constexpr (MFMA == 2) {
                switch (tid >> 6) {  // wave-uniform; every branch meets the same barriers
                case 0: wide_item_b3<0, CHUNKS>(B, idx, beg, n, beg2, n2, d, st, w, sM, ss, dst, ws); break;
                case 1: wide_item_b3<1, CHUNKS>(B, idx, beg, n, beg2, n2, d, st, w, sM, ss, dst, ws); break;
                case 2: wide_item_b3<2, CHUNKS>(B, idx, beg, n, beg2, n2, d, st, w, sM, ss, dst, ws); break;
                default: wide_item_b3<3, CHUNKS>(B, idx, beg, n, beg2, n2, d, st, w, sM, ss, dst, ws); break;
                }
            } else {
                switch (tid >> 6) {
                case 0: wide_item_mfma<0, CHUNKS>(B, idx, beg, n, beg2, n2, d, st, w, sM, ss, dst, ws); break;
                case 1: wide_item_mfma<1, CHUNKS>(B, idx, beg, n, beg2, n2, d, st, w, sM, ss, dst, ws); break;
                case 2: wide_item_mfma<2, CHUNKS>(B, idx, beg, n, beg2, n2, d, st, w, sM, ss, dst, ws); break;
                default: wide_item_mfma<3, CHUNKS>(B, idx, beg, n, beg2, n2, d, st, w, sM, ss, dst, ws); break;
                }
            }
            if (!CHUNKS) {
                if (tid < 128) sq[256 + tid] = pa;  // (the batch region: free in this form of the kernel)
                __syncthreads();
                unsigned long long t1 = 0;
                if (prof) t1 = __builtin_amdgcn_s_memtime();
                if constexpr (!CHUNKS) {
                    if (!(probe & 2)) wide_form_y(sM, sq);
                }
                __syncthreads();
                unsigned long long t2 = 0;
                if (prof) t2 = __builtin_amdgcn_s_memtime();
                if (!(probe & 1)) wide_sweep(A + u * d, d, reg, sM, ss, sq);
                if (prof) {
                    c_acc += t1 - t0;
                    c_m += t2 - t1;
                    c_solve += __builtin_amdgcn_s_memtime() - t2;
                    c_rows++;
                    c_ent += n;
                }
            }
            u = u2, beg = beg2, n = n2;
        }
        if (prof && threadIdx.x == 0) {
            atomicAdd(prof + 0, c_acc);
            atomicAdd(prof + 1, c_m);
            atomicAdd(prof + 2, c_solve);
            atomicAdd(prof + 3, c_rows);
            atomicAdd(prof + 4, c_ent);
            atomicAdd(prof + 5, (unsigned long long)__builtin_amdgcn_s_memtime() - t_begin);
            atomicAdd(prof + 6, 1ull);
        }
        return;
    }
    for (int64_t t = blockIdx.x; t < n_items; t += gridDim.x) {
        float acc[8][8], cs[8];
        if (CHUNKS) {
            wide_accumulate(B, idx, chunk_beg[t], chunk_cnt[t], d, sq, acc, cs);
            float *dst = partial + t * kWidePartial;
#pragma unroll
            for (int x = 0; x < 8; x++)
#pragma unroll
                for (int y = 0; y < 8; y++) dst[(8 * ti + x) * 128 + 8 * tj + y] = acc[x][y];
            if (ti == 0) {
#pragma unroll
                for (int y = 0; y < 8; y++) dst[128 * 128 + 8 * tj + y] = cs[y];
            }
        } else {
            const int64_t u = rows[t];
            const int64_t beg = ptr[u];
            wide_accumulate(B, idx, beg, (int)(ptr[u + 1] - beg), d, sq, acc, cs);
            wide_solve(A + u * d, S, d, w, reg, sM, ss, acc, cs);
        }
    }
}

// the long rows: the partial G / sums of a row's chunks added one after the other (a fixed order: deterministic, and the same
// whichever process solves the row), then the solve
__global__ __launch_bounds__(256) void als_wide_long_kernel(float *__restrict__ A, const float *__restrict__ S,
                                                            const int32_t *__restrict__ rows, const int32_t *__restrict__ first,
                                                            const int32_t *__restrict__ nch, int64_t n_rows, int d, float w,
                                                            float reg, const float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sM = smem;
    float *ss = sM + 128 * kWideLd + kWideBatch * 128;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    for (int64_t t = blockIdx.x; t < n_rows; t += gridDim.x) {
        float acc[8][8], cs[8];
#pragma unroll
        for (int x = 0; x < 8; x++) {
            cs[x] = 0.0f;
#pragma unroll
            for (int y = 0; y < 8; y++) acc[x][y] = 0.0f;
        }
        for (int c = 0; c < nch[t]; c++) {
            const float *src = partial + (int64_t)(first[t] + c) * kWidePartial;
#pragma unroll
            for (int x = 0; x < 8; x++)
#pragma unroll
                for (int y = 0; y < 8; y++) acc[x][y] += src[(8 * ti + x) * 128 + 8 * tj + y];
            if (ti == 0) {
#pragma unroll
                for (int y = 0; y < 8; y++) cs[y] += src[128 * 128 + 8 * tj + y];
            }
        }
        wide_solve(A + (int64_t)rows[t] * d, S, d, w, reg, sM, ss, acc, cs);
    }
}

// ---- Gram-form row solve (d <= 64) --------------------------------------------------------------
// model.go:659-690 keeps a residual per feedback entry and re-walks the row's n entries for every one
// of the d coordinates.  Substituting the residual, r^f_t = sum_{k != f} p_k q_tk, turns the three sums into
//     a - b = s_f - sum_{k != f} p_k M_kf,   c + w S_ff = M_ff,   M = (1 - w) G + w S,
//     G = sum_t q_t q_t^T,   s = sum_t q_t        (t over the row's feedback entries)
// i.e. the same Gauss-Seidel sweep over f on a d x d system per row.  G is built by
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fmaf chain over the entries) straight from the
// gathered rows: a half-wave reads 32 consecutive floats of one feedback row (128 B), so one MFMA
// consumes two entries.  Same mathematics, different summation order than the reference: parity is
// the 1e-4 relative bar of BASELINE.md (the reference's own fp32 residual recurrence is no closer to
// the exact solution: tests/test_gpu_cf_parity.py::test_als_*).
int g_als_long_row = 0;  // rows longer than this are cut into chunks; 0 = by the side's size (als_build_plan); test hook: gorse_hip_test_set_als_plan
int g_als_chunk = 0;     // feedback entries per chunk of a long row; 0 = the threshold
bool g_als_solve_adds = false;  // tests (path | 2048): no als_partial_reduce_kernel in front of the long-row solve
int g_als_path = 0;         // 0 auto (Gram form: MFMA kernels for d <= 64, als_wide_kernel for d <= 128; else the residual
                            // sweep), 1 force the residual sweep, 2 force the MFMA Gram form (d <= 64)
int g_als_wide_fma = 0;     // als_wide_kernel: G by fused multiply-adds (round 2) instead of the fp32 MFMA (probe: path | 8)
bool g_als_prof = false;    // probe: 8 counters per side in h->als_prof (gorse_hip_test_als_profile)
int g_als_slow_gather = 0;  // als_row_kernel / als_chunk_kernel: the first form of the gather stage whatever the shape (probe: path | 64)
int g_als_wide_probe = 0;   // timing probes of als_wide_kernel (results are garbage): path | 16 = no sweep, path | 32 = S not added
int g_als_nob3 = 0;         // no bf16 x 3 Gram: d = 32 / 64 take the fp32 16 x 16 tiles (probe: path | 1024)
int g_als_waves8 = 0;       // als_row_kernel in 16 x 16 tiles: 8 waves per workgroup even where 12 fit (probe: path | 256)
int g_als_tile32 = 0;       // als_row_kernel / als_chunk_kernel: 32 x 32 MFMA tiles even where d = 16 NB takes 16 x 16 ones (probe: path | 128)
int g_als_phased = 0;       // als_row_kernel: the waves of a workgroup accumulate together and solve together (probe: path | 4)
constexpr int kAlsDP = 65;          // LDS row stride of the per-wave M matrix
constexpr int kAlsWaves = 4;        // waves per workgroup of the chunk kernels
constexpr int kAlsRowWaves = 8;     // waves per workgroup of als_row_kernel: ONE workgroup per CU (2 waves per SIMD) so that its LDS
                                    // holds the 8 per-wave M buffers (133 KB) next to one copy of S (16 KB)


__device__ __forceinline__ float wave_sum64(float v) {
    v = group_tree16(v);  // every lane of a 16-lane row holds its row's total
    int m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __int_as_float(m);
    m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    v += __int_as_float(m);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int NB>
struct GramAcc {
    static constexpr int NT = NB * (NB + 1) / 2;
    f32x16 t[NT];   // upper-triangular 32x32 tiles (bi <= bj), row-major over (bi, bj)
    float sum[NB];  // this lane's share of the column sums
    __device__ __forceinline__ static constexpr int tile(int bi, int bj) { return bi * NB - bi * (bi - 1) / 2 + (bj - bi); }
};

// (Rounds 1-4 accumulated these tiles on the fp32 MFMA, v_mfma_f32_32x32x2_f32 over gathered entry pairs -- gram_accumulate, three forms of
// its gather stage; the 16 x 16 tile forms below and the bf16 form replaced it for every shape in round 5.)

// ---- the same accumulation on the bf16 MFMA, every fp32 value split three ways (d = 32 NB with the fast gather stage) --------------
// The fp32 MFMA runs at 1/16 of the bf16 rate.  A float is EXACTLY hi + mid + lo with hi = bf16(x), mid = bf16(x - hi),
// lo = x - hi - mid (round to nearest even; 8 + 8 + 8 significand bits and a float's exponent range: lo IS a bf16 value;
// tests/test_als_split_cpu.py), so x y = (hi + mid + lo)(hi' + mid' + lo'); the six products hi hi', hi mid', mid hi', hi lo', lo hi',
// mid mid' are formed exactly by v_mfma_f32_32x32x16_bf16 and added in fp32; what is dropped (mid lo', lo mid', lo lo') is below
// 2^-23 of |x y| (|mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|) -- an fp32 product's own rounding is 2^-24.  Six bf16 MFMAs of 16 entries (32 cycles each) per 32 x 32 tile: 36 cycles per gathered row at d = 64
// against 80 for the fp32 tiles.  Fragment: lane l holds column (l & 31) of the EIGHT entries 8 (l >> 5) .. + 7 of a 16-entry stage
// (the MFMA's k index), two values per register; the accumulators and their D layout are those of the fp32 32 x 32 form.
constexpr int kAlsOct = 8;  // entries per lane and stage
#ifndef GORSE_ALS_ROW_DEEP
#define GORSE_ALS_ROW_DEEP 1
#endif
constexpr bool kAlsRowDeep = GORSE_ALS_ROW_DEEP != 0;  // the row kernel's bf16 form: four gather stages and no pairing of sweeps (the registers of
                                                       // one or the other): C5 5.69-5.71 against 5.78 ms (profiles/r04_zq_ab_row_ring.txt)
constexpr bool kAlsRowPair = !kAlsRowDeep;

template <int NB>
__device__ __forceinline__ void gram_load_stage_b3(const float *__restrict__ B, uint32_t rowbytes, int idx, int first, int lane,
                                                   float (&fr)[kAlsOct][NB]) {
    const int half = lane >> 5, col = lane & 31;
    const int sel = (first + 8 * half) * 4;  // ds_bpermute address of this lane's first entry of the stage
#pragma unroll
    for (int k = 0; k < kAlsOct; k++) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute(sel + 4 * k, idx);
        const uint32_t off = __umul24(r, rowbytes) + (uint32_t)col * 4u;
        const float *row = reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + off);
#pragma unroll
        for (int b = 0; b < NB; b++) fr[k][b] = row[32 * b];
    }
}

template <int NB, bool DEEP = true>  // DEEP: four gather stages in the ring (long runs of stages: the chunk kernel), else two
__device__ __forceinline__ void gram_accumulate_b3(const float *__restrict__ B, const int32_t *__restrict__ fb, int n, int d, int lane,
                                                   GramAcc<NB> &g, int idx0, int idx1, int zero_row) {
#pragma unroll
    for (int t = 0; t < GramAcc<NB>::NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) g.t[t][r] = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; b++) g.sum[b] = 0.0f;
    if (n <= 0) return;
    const int nstages = (n + 15) / 16;
    constexpr int kStagesPerBatch = 4;
    int idx_cur = lane < n ? idx0 : zero_row;
    int idx_nxt = 64 + lane < n ? idx1 : zero_row;
    int loaded = 0;
    auto issue = [&](float (&fr)[kAlsOct][NB]) {
        const int sb = loaded % kStagesPerBatch;
        if (sb == 0 && loaded > 0) {
            idx_cur = idx_nxt;
            const int64_t nb = (int64_t)(loaded / kStagesPerBatch + 1) * 64 + lane;
            idx_nxt = nb < n ? fb[nb] : zero_row;
        }
        gram_load_stage_b3<NB>(B, (uint32_t)d * 4u, idx_cur, sb * 16, lane, fr);
        loaded++;
    };
    auto consume = [&](const float (&fr)[kAlsOct][NB]) {
        als_bf16x8 hi[NB], mid[NB], lo[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            uint32_t ph[4], pm[4], pl[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
                const float x0 = fr[2 * k2][b], x1 = fr[2 * k2 + 1][b];
                asm("v_add_f32 %0, %1, %0" : "+v"(g.sum[b]) : "v"(x0));
                asm("v_add_f32 %0, %1, %0" : "+v"(g.sum[b]) : "v"(x1));
                split_pair_bf16(x0, x1, ph[k2], pm[k2], pl[k2]);  // eleven vector instructions per pair of values
            }
            hi[b] = *reinterpret_cast<als_bf16x8 *>(ph);
            mid[b] = *reinterpret_cast<als_bf16x8 *>(pm);
            lo[b] = *reinterpret_cast<als_bf16x8 *>(pl);
        }
        // product after product over the tiles (consecutive MFMAs write different accumulators), the small terms first
#pragma unroll
        for (int pr = 0; pr < 6; pr++)
#pragma unroll
            for (int bi = 0; bi < NB; bi++)
#pragma unroll
                for (int bj = bi; bj < NB; bj++) {
                    f32x16 &acc = g.t[GramAcc<NB>::tile(bi, bj)];
                    const als_bf16x8 &a = pr == 0 ? mid[bi] : (pr == 2 ? lo[bi] : (pr == 4 ? mid[bi] : hi[bi]));
                    const als_bf16x8 &c = pr == 0 ? mid[bj] : (pr == 1 ? lo[bj] : (pr == 3 ? mid[bj] : hi[bj]));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, c, acc, 0, 0, 0);
                }
    };
    // DEEP: three stages in flight, a fourth being consumed (a ring of four buffers, unrolled: nothing moves) -- for the chunk kernel,
    // whose wave walks 256 stages on end: the S Gram 0.27 -> 0.22 ms and the long rows' share of the sweeps with it (C5 6.1 -> 5.8 ms,
    // profiles/r04_zo_probe_als_tiles.txt).  In the row kernel (a 100-entry row is seven stages) the deeper ring is worth 1.5 %, and only
    // in place of the pairing of two rows' sweeps -- with both the registers spill (kAlsRowDeep).
    if constexpr (DEEP) {
        float f0[kAlsOct][NB], f1[kAlsOct][NB], f2[kAlsOct][NB], f3[kAlsOct][NB];
        issue(f0);
        issue(f1);
        issue(f2);
        int s = 0;
        for (; s + 4 <= nstages; s += 4) {
            issue(f3);
            consume(f0);
            issue(f0);
            consume(f1);
            issue(f1);
            consume(f2);
            issue(f2);
            consume(f3);
        }
        if (s < nstages) consume(f0);
        if (s + 1 < nstages) consume(f1);
        if (s + 2 < nstages) consume(f2);
        return;
    } else {
    float f0[kAlsOct][NB], f1[kAlsOct][NB];
    issue(f0);
    int s = 0;
    for (; s + 2 <= nstages; s += 2) {
        issue(f1);
        consume(f0);
        issue(f0);
        consume(f1);
    }
    if (s < nstages) consume(f0);
    }
}

// ---- the same accumulation in 16 x 16 tiles (d = 16 NB with the fast gather stage) -------------------------------------------------
// A 32 x 32 diagonal tile computes both triangles of its block: 3072 multiply-adds per gathered row at d = 64 for the 2080 of the
// strict triangle.  v_mfma_f32_16x16x4_f32 has the same rate (1024 multiply-adds in 32 cycles) on a quarter of the area: ten tiles
// cover the upper triangle of 64 x 64 with 2560 multiply-adds per row (six of 48 x 48: 1536 against 3072; three of 32 x 32: 768
// against 1024; one of 16 x 16: 256 against 1024).  A / B fragment: lane l holds q[entry l >> 4 of the quad][16 b + (l & 15)];
// D: register r of lane l is element (4 (l >> 4) + r, l & 15).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kAlsQuads = 4;  // feedback-entry quads per pipeline stage (16 entries, as kAlsPairs pairs)

template <int NB>
struct GramAcc16 {
    static constexpr int NT = NB * (NB + 1) / 2;
    f32x4 t[NT];    // upper-triangular 16x16 tiles (bi <= bj), row-major over (bi, bj)
    float sum[NB];  // this lane's share of the column sums
    __device__ __forceinline__ static constexpr int tile(int bi, int bj) { return bi * NB - bi * (bi - 1) / 2 + (bj - bi); }
};

// FULL: d = 16 NB; else (round 5: any d <= 16 NB -- the reference's own test width 8 among them) the lanes of the columns past d read word
// 0 of the zero row (one select per load, on the offset).  O64: the matrix spans 4 GB and more, or 2^24 rows: a 64-bit address per
// entry instead of the 24 x 24-bit offset.
template <int NB, bool FULL = true, bool O64 = false>
__device__ __forceinline__ void gram_load_stage16(const float *__restrict__ B, uint32_t rowbytes, int idx, int first, int lane,
                                                  float (&fr)[kAlsQuads][NB], int d = 16 * NB, int zero_row = 0) {
    const int slot = lane >> 4, col = lane & 15;
    const int sel = (first + slot) * 4;  // ds_bpermute address of this lane's entry of quad 0
    if constexpr (O64) {
        const char *zero = reinterpret_cast<const char *>(B) + (uint64_t)(uint32_t)zero_row * rowbytes;
#pragma unroll
        for (int j = 0; j < kAlsQuads; j++) {
            const uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute(sel + 16 * j, idx);
            const char *row = reinterpret_cast<const char *>(B) + (uint64_t)r * rowbytes + (uint32_t)col * 4u;
#pragma unroll
            for (int b = 0; b < NB; b++) fr[j][b] = *reinterpret_cast<const float *>((FULL || 16 * b + col < d) ? row + 64 * b : zero);
        }
    } else {
        const uint32_t zero_off = __umul24((uint32_t)zero_row, rowbytes);
#pragma unroll
        for (int j = 0; j < kAlsQuads; j++) {
            const uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute(sel + 16 * j, idx);
            const uint32_t off = __umul24(r, rowbytes) + (uint32_t)col * 4u;
            if constexpr (FULL) {
                const float *row = reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + off);
#pragma unroll
                for (int b = 0; b < NB; b++) fr[j][b] = row[16 * b];  // (the block's 64 bytes: the load's immediate offset)
            } else {
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const uint32_t o = 16 * b + col < d ? off + 64u * b : zero_off;
                    fr[j][b] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + o);
                }
            }
        }
    }
}

// idx0 / idx1: lane l's entries l and 64 + l of the row as loaded by the caller (any value past the row's end); zero_row: the row id of
// the zero row behind B (what an entry past the row's end reads); DEEP: as gram_accumulate_b3
template <int NB, bool DEEP = false, bool FULL = true, bool O64 = false>
__device__ __forceinline__ void gram_accumulate16(const float *__restrict__ B, const int32_t *__restrict__ fb, int n, int d, int lane,
                                                  GramAcc16<NB> &g, int idx0, int idx1, int zero_row) {
#pragma unroll
    for (int t = 0; t < GramAcc16<NB>::NT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) g.t[t][r] = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; b++) g.sum[b] = 0.0f;
    if (n <= 0) return;
    const int nstages = (n + 4 * kAlsQuads - 1) / (4 * kAlsQuads);
    constexpr int kStagesPerBatch = 64 / (4 * kAlsQuads);
    int idx_cur = lane < n ? idx0 : zero_row;
    int idx_nxt = 64 + lane < n ? idx1 : zero_row;
    int loaded = 0;
    auto issue = [&](float (&fr)[kAlsQuads][NB]) {
        const int sb = loaded % kStagesPerBatch;
        if (sb == 0 && loaded > 0) {
            idx_cur = idx_nxt;
            const int64_t nb = (int64_t)(loaded / kStagesPerBatch + 1) * 64 + lane;
            idx_nxt = nb < n ? fb[nb] : zero_row;
        }
        gram_load_stage16<NB, FULL, O64>(B, (uint32_t)d * 4u, idx_cur, sb * 4 * kAlsQuads, lane, fr, d, zero_row);
        loaded++;
    };
    // (Skipping the MFMAs of the padding -- a stage and a half of the eight of a 100-entry row -- behind a wave-uniform branch per quad
    // or per stage cost more than it saved: the branch keeps the next stage's gathers from being issued among the MFMAs.  C5 7.28
    // and 7.52-7.60 ms against 7.06-7.09 unguarded; profiles/r04_u_probe_als_tiles_skip.txt.)
    auto consume = [&](const float (&fr)[kAlsQuads][NB]) {
#pragma unroll
        for (int j = 0; j < kAlsQuads; j++) {
#pragma unroll
            for (int b = 0; b < NB; b++) asm("v_add_f32 %0, %1, %0" : "+v"(g.sum[b]) : "v"(fr[j][b]));
#pragma unroll
            for (int bi = 0; bi < NB; bi++)
#pragma unroll
                for (int bj = bi; bj < NB; bj++)
                    g.t[GramAcc16<NB>::tile(bi, bj)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        fr[j][bi], fr[j][bj], g.t[GramAcc16<NB>::tile(bi, bj)], 0, 0, 0);
        }
    };
    // ping-pong over pairs of stages; an odd last stage (already gathered into f0) is consumed behind the loop instead of being
    // paired with sixteen zero rows (C5: 7.24-7.26 against 7.46-7.49 ms in one session, profiles/r04_u_probe_als_tiles_peel.txt)
    if constexpr (DEEP) {
        float f0[kAlsQuads][NB], f1[kAlsQuads][NB], f2[kAlsQuads][NB], f3[kAlsQuads][NB];
        issue(f0);
        issue(f1);
        issue(f2);
        int s = 0;
        for (; s + 4 <= nstages; s += 4) {
            issue(f3);
            consume(f0);
            issue(f0);
            consume(f1);
            issue(f1);
            consume(f2);
            issue(f2);
            consume(f3);
        }
        if (s < nstages) consume(f0);
        if (s + 1 < nstages) consume(f1);
        if (s + 2 < nstages) consume(f2);
    } else {
        float f0[kAlsQuads][NB], f1[kAlsQuads][NB];
        issue(f0);
        int s = 0;
        for (; s + 2 <= nstages; s += 2) {
            issue(f1);
            consume(f0);
            issue(f0);
            consume(f1);
        }
        if (s < nstages) consume(f0);
    }
}

// the callback gets the compile-time parts of the coordinates: i = ci + 4 * (lane >> 4), j = cj + (lane & 15)
template <int NB, typename F>
__device__ __forceinline__ void gram_foreach16(const GramAcc16<NB> &g, F &&f) {
#pragma unroll
    for (int bi = 0; bi < NB; bi++)
#pragma unroll
        for (int bj = bi; bj < NB; bj++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = g.t[GramAcc16<NB>::tile(bi, bj)][r];
                f(16 * bi + r, 16 * bj, v, false);
                if (bi != bj) f(16 * bi + r, 16 * bj, v, true);
            }
}

// column sums: the four 16-lane rows of the wave hold the four entry slots' shares of column 16 b + (l & 15)
template <int NB>
__device__ __forceinline__ void gram_store_sums16(const GramAcc16<NB> &g, int lane, float *dst, int d = 16 * NB) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
        float v = g.sum[b];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16 && 16 * b + lane < d) dst[16 * b + lane] = v;
    }
}

// visit every value of the full symmetric G held as upper-triangular tiles; C layout of the 32x32 MFMA: lane holds
// column lane & 31, rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).  The callback gets the COMPILE-TIME parts (ci, cj)
// of the coordinates: i = ci + 4 * (lane >> 5), j = cj + (lane & 31); `mirror` says the value is being handed over a
// second time for the transposed position (j, i) of an off-diagonal tile.
template <int NB, typename F>
__device__ __forceinline__ void gram_foreach(const GramAcc<NB> &g, F &&f) {
#pragma unroll
    for (int bi = 0; bi < NB; bi++)
#pragma unroll
        for (int bj = bi; bj < NB; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ci = 32 * bi + (r & 3) + 8 * (r >> 2), cj = 32 * bj;
                const float v = g.t[GramAcc<NB>::tile(bi, bj)][r];
                f(ci, cj, v, false);
                if (bi != bj) f(ci, cj, v, true);
            }
}

// column sums: lanes l and l + 32 hold the two halves of column (32 b + l & 31)
template <int NB>
__device__ __forceinline__ void gram_store_sums(const GramAcc<NB> &g, int d, int lane, float *dst) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const float other = __shfl_xor(g.sum[b], 32, 64);
        const int e = 32 * b + (lane & 31);
        if (lane < 32 && e < d) dst[e] = g.sum[b] + other;
    }
}

// one Gauss-Seidel sweep over the d coordinates of row `a` against M (LDS, stride kAlsDP) and s (LDS)
// One Gauss-Seidel sweep over the d coordinates of row `a`, M held in registers: lane k keeps column k of the
// symmetric M (= row k), so step f needs M[f][k] = mcol[f] -- a register picked by the unrolled loop index -- and the
// only cross-lane traffic is the wave sum and two v_readlane.  (The first version re-read M from LDS every step: 475
// cycles per step, latency bound; profiles/r01_n_probe_als_prof.txt.)
//   FORM: mcol[i] = (1 - w) * G[i][lane] + w * S[i][lane] from the raw Gram in LDS and S in global memory (L1-resident);
//   !FORM: sM already holds M.
//   FULL: d == DMAX is known at compile time -- no bound checks, and the step loop is straight-line code.
// Every load of the set-up is unconditional (clamped index, value masked afterwards): a load inside `if (i < d && lane < d)`
// becomes its own exec-masked region that waits for its own LDS round trip -- 64 of them in a row cost 11.7K of the solve's
// 29K cycles per row (profiles/r02_x_probe_als_phased.txt, which also shows that the sibling wave's MFMAs are NOT what slows
// the solve: it takes as long when all eight waves of the workgroup solve together).
// The d steps of the sweep, unrolled at compile time (the lane a step writes is an immediate of v_writelane_b32).
// what a row's sweep needs once its M is formed: column `lane` of M, the running y = M p, 1 / (M_kk + reg), base_k - p_k, p_k, and where
// the row goes
template <int DMAX>
struct SolveState {
    float mcol[DMAX];
    float y, inv, gk, p0;
    float *a;
    bool lane_in;
    int steps;  // lane f: the bits of delta_f once step f has run
};
template <int F, int DMAX, bool FULL>
struct SolveSteps {
    static __device__ __forceinline__ void run(SolveState<DMAX> &x, int d) {
        if (FULL || F < d) {  // uniform
            const float dk = fmaf(-x.y, x.inv, x.gk);
            const int delta = __builtin_amdgcn_readlane(__float_as_int(dk), F);
            x.y = fmaf(__int_as_float(delta), x.mcol[F], x.y);
            asm("v_writelane_b32 %0, %1, %2" : "+v"(x.steps) : "s"(delta), "n"(F));
        }
        SolveSteps<F + 1, DMAX, FULL>::run(x, d);
    }
    // two rows' sweeps, step by step: each chain is three dependent operations per step and every link waits ~45 cycles next to the
    // sibling wave's MFMA stream -- two independent chains take the same slots twice as well
    static __device__ __forceinline__ void run2(SolveState<DMAX> &x, SolveState<DMAX> &z, int d) {
        if (FULL || F < d) {
            const float dkx = fmaf(-x.y, x.inv, x.gk);
            const float dkz = fmaf(-z.y, z.inv, z.gk);
            const int dx = __builtin_amdgcn_readlane(__float_as_int(dkx), F);
            const int dz = __builtin_amdgcn_readlane(__float_as_int(dkz), F);
            x.y = fmaf(__int_as_float(dx), x.mcol[F], x.y);
            z.y = fmaf(__int_as_float(dz), z.mcol[F], z.y);
            asm("v_writelane_b32 %0, %1, %2" : "+v"(x.steps) : "s"(dx), "n"(F));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(z.steps) : "s"(dz), "n"(F));
        }
        SolveSteps<F + 1, DMAX, FULL>::run2(x, z, d);
    }
};
template <int DMAX, bool FULL>
struct SolveSteps<DMAX, DMAX, FULL> {
    static __device__ __forceinline__ void run(SolveState<DMAX> &, int) {}
    static __device__ __forceinline__ void run2(SolveState<DMAX> &, SolveState<DMAX> &, int) {}
};
// everything in front of the chain: the columns of M into registers, the diagonal, y = M p
// PAD (with FULL): the system is DMAX x DMAX with zeros past dvalid -- M, S (stride DMAX) and the sums are zero there -- so the DMAX
// steps run as straight-line code and the steps past dvalid change nothing; only the lanes past dvalid are masked (ONE mask: a
// bound check per column of M kept 2 x DMAX scalar registers alive across the row loop, 392 scalar spills at DMAX = 64).
template <int DMAX, bool FORM, bool FULL = false, int DP = kAlsDP, bool PAD = false>  // DP: row stride of sM
__device__ __forceinline__ void als_solve_prepare(SolveState<DMAX> &x, float *__restrict__ a, const float *sM, const float *ss,
                                                  const float *__restrict__ S, int d, float one_w, float w, float reg, int lane,
                                                  unsigned long long *c_load = nullptr, int dvalid = 0) {
    static_assert(!PAD || FULL, "a padded system is solved with all DMAX steps");
    unsigned long long t_in = 0;
    if (c_load) t_in = __builtin_amdgcn_s_memtime();
    if (FULL) d = DMAX;
    float (&mcol)[DMAX] = x.mcol;
    if (FORM) {  // an opaque zero offset per call: keeps the 64 loads of S inside the row loop instead of 64 registers
        int z;   // hoisted across the whole kernel (they are LDS / L1 hits; the registers are needed by the accumulation)
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));
        S += z;
    }
    // lanes past d hold nothing (DMAX = 32 leaves half of the wave idle): masked even in the FULL form, where the test is a
    // compile-time fact for DMAX = 64
    const bool lane_in = PAD ? lane < dvalid : ((FULL && DMAX >= 64) || lane < d);
    const int lane_c = PAD ? min(lane, dvalid - 1) : ((FULL && DMAX >= 64) ? lane : min(lane, d - 1));
#pragma unroll
    for (int i0 = 0; i0 < DMAX; i0 += 16) {  // 16 columns' worth of loads in flight at a time (register pressure)
#pragma unroll
        for (int i = i0; i < i0 + 16; i++) {
            float m = sM[i * DP + (DP == kAlsDP ? lane : lane_c)];  // (DP = kAlsDP: sM is 64 x kAlsDP and zero past d)
            if (FORM) m = one_w * m + w * S[(FULL ? i : min(i, d - 1)) * d + lane_c];
            mcol[i] = ((FULL || i < d) && lane_in) ? m : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // Gauss-Seidel with a running y = M p instead of one wave-wide dot product per coordinate:
    //     sum_{k != f} p_k M_kf = y_f - p_f M_ff,   p_f' = (s_f - y_f + p_f M_ff) / (M_ff + reg),   y += (p_f' - p_f) M[:, f]
    // Lane k keeps y_k, and column f of the symmetric M is mcol[f] of every lane, so a step is one v_readlane of y, two
    // VALU operations on wave-uniform values and ONE fused multiply-add per lane -- no cross-lane reduction in the chain
    // (the wave sum took ~650 cycles per step next to a sibling wave's MFMAs: 42K cycles per row against 27K for the Gram
    // accumulation, profiles/r01_p_probe_als_prof.txt).  Same recurrence, products summed in a different order; agreement
    // with the float64 recurrence 6e-7 of the row's scale on random systems (well inside the 1e-4 bar).
    if (c_load) *c_load += __builtin_amdgcn_s_memtime() + (__float_as_int(mcol[DMAX - 1]) & 0) - t_in;  // probe: columns of M in registers
    const float p0_raw = a[lane_c], sv_raw = ss[lane_c];
    float diag = sM[lane_c * DP + lane_c];
    if (FORM) diag = one_w * diag + w * S[lane_c * d + lane_c];
    const float p0 = lane_in ? p0_raw : 0.0f;
    const float sv = lane_in ? sv_raw : 0.0f;
    if (!lane_in) diag = 0.0f;
    float inv = __builtin_amdgcn_rcpf(diag + reg);  // 1 ulp; the parity bar of ALS is 1e-4 relative
    if (PAD && !lane_in) inv = 0.0f;  // (reg may be 0: the step of a padding coordinate must come out as 0, not 0 / 0)
    const float base = (sv + p0 * diag) * inv;
    // y = M p as four interleaved partial sums: one chain of DMAX dependent operations cost ~45 cycles per link next to the sibling
    // wave's MFMA stream (a fifth of the solve, profiles/r04_u_probe_als_tiles_peel.txt)
    float yq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int f = 0; f < DMAX; f++)
        if (FULL || f < d) yq[f & 3] = fmaf(mcol[f], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), f)), yq[f & 3]);
    x.y = (yq[0] + yq[1]) + (yq[2] + yq[3]);
    // Three operations on the chain per step (wide_sweep has the same form): every lane forms its own would-be step
    // delta_k = (base_k - p_k) - y_k inv_k with one fused operation, lane f's is broadcast, y += delta_f M[:, f]; the new
    // coordinate base_f - y_f inv_f is formed beside the chain.  (Four broadcasts and two more dependent operations per step
    // before: 235 cycles per step next to the sibling wave's MFMAs, profiles/r03_zi_probe_als_prof.txt.)
    x.inv = inv;
    x.gk = base - p0;
    x.p0 = p0;
    x.a = a;
    x.lane_in = lane_in;
    // Lane f's new coordinate is p0_f + delta_f, and delta_f is already in a scalar register (the broadcast of the chain): it is
    // written into lane f of a vector of steps (v_writelane, one instruction beside the chain) and added to p0 once after the loop.
    // (Until round 4 every step selected `lane == f ? p0 + dk : p`: hipcc kept the 64 masks in 128 scalar registers, spilled them
    // into vector registers once per row and reloaded two words per step -- the 64 v_writelane + ~130 v_readlane per row of the
    // kernel's 416 "SGPR spills".)
    x.steps = 0;
}
template <int DMAX>
__device__ __forceinline__ void als_solve_finish(const SolveState<DMAX> &x, int lane) {
    const float p = x.p0 + __int_as_float(x.steps);  // p_f' = p_f + delta_f; lanes past d: 0 + 0
    if (x.lane_in) x.a[lane] = p;
}
// one Gauss-Seidel sweep over the d coordinates of row `a` against M (LDS, stride DP) and s (LDS)
template <int DMAX, bool FORM, bool FULL = false, int DP = kAlsDP>
__device__ __forceinline__ void als_solve_row(float *__restrict__ a, const float *sM, const float *ss,
                                              const float *__restrict__ S, int d, float one_w, float w, float reg,
                                              int lane, unsigned long long *c_load = nullptr) {
    SolveState<DMAX> x;
    als_solve_prepare<DMAX, FORM, FULL, DP>(x, a, sM, ss, S, d, one_w, w, reg, lane, c_load);
    SolveSteps<0, DMAX, FULL>::run(x, FULL ? DMAX : d);
    als_solve_finish<DMAX>(x, lane);
}

// A: side being solved, B: the other side, S: d x d Gram of B over rows with feedback
// MODE 1: NB counts 16-column blocks, d = 16 NB, fp32 MFMA in 16 x 16 tiles.  MODE 2: d = 32 NB, the bf16 MFMA on three-way split
// values (gram_accumulate_b3).  MODE 3 (round 5): ANY d <= 16 NB in 16 x 16 fp32 tiles -- the columns past d are gathered from the
// zero row, M is 16 NB x (16 NB + 1) with zeros past d, the sweep runs d steps -- which is what the reference's test width 8 and
// every nFactors that is not a multiple of 16 take.  MODE 4: mode 3 with 64-bit gather addresses (factor matrices of 4 GB and
// more, or 2^24 rows).  (The 32 x 32 fp32 tile form of rounds 1-4, MODE 0, served those shapes with 124 registers + 175 scalar spills
// for d = 8; it is gone.)  All modes gather through the zero row behind the matrix (zero_row), keep M per wave in LDS and pair the
// rows' sweeps.  WAVES: waves per workgroup (one workgroup per CU).
template <int NB, int MODE = 1, int WAVES = kAlsRowWaves>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void als_row_kernel(float *__restrict__ A, const float *__restrict__ B,
                                                                 const int64_t *__restrict__ ptr,
                                                                 const int32_t *__restrict__ idx,
                                                                 const float *__restrict__ S,
                                                                 const int32_t *__restrict__ rows, int64_t n_rows, int d,
                                                                 float w, float reg, const float *__restrict__ zeros,
                                                                 unsigned long long *prof, int phased, int zero_row) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // probe only (prof != null): s_memtime ticks per wave in [0] Gram accumulation, [1] M to LDS, [2] solve; [3] rows,
    // [4] feedback entries, [5] kernel ticks, [6] waves
    unsigned long long c_acc = 0, c_m = 0, c_solve = 0, c_rows = 0, c_ent = 0, t_begin = 0, c_load = 0;
#ifndef GORSE_PROBE
    prof = nullptr;  // the phase counters exist in `make probe-lib` builds only: here every `if (prof)` folds away
#endif
    if (prof) t_begin = __builtin_amdgcn_s_memtime();
    // S (d x d, the same for every row of the half-sweep) is copied to LDS once: read from global memory inside the solve, its
    // 64 loads per row queued behind the sibling waves' gathers -- 24.6K of the solve's 40.5K cycles per row
    // (profiles/r02_i_probe_als_prof.txt)
    float *sS = smem;
    if constexpr (MODE == 3 || MODE == 4) {  // S padded to the tiles' 16 NB x 16 NB with zeros (als_solve_prepare, PAD)
        for (int e = threadIdx.x; e < 16 * NB * 16 * NB; e += blockDim.x) {
            const int i = e / (16 * NB), j = e % (16 * NB);
            sS[e] = i < d && j < d ? S[i * d + j] : 0.0f;
        }
    } else {
        for (int e = threadIdx.x; e < d * d; e += blockDim.x) sS[e] = S[e];
    }
    __syncthreads();
    static_assert(MODE >= 1 && MODE <= 4, "als_row_kernel modes");
    constexpr bool T16 = MODE != 2;
    constexpr bool FULLD = MODE == 1 || MODE == 2;       // d is DD itself
    constexpr int DD = MODE == 2 ? 32 * NB : 16 * NB;    // modes 1, 2: d itself; 3, 4: d rounded up to whole 16-column blocks
    constexpr int DP = DD + 1, MROWS = DD;
    float *sM = smem + (size_t)(FULLD ? d * d : DD * DD) + (size_t)wv * (MROWS * DP + MROWS);
    float *ss = sM + MROWS * DP;
    const float one_w = 1 - w;
    const int64_t wave = (int64_t)blockIdx.x * WAVES + wv, nwaves = (int64_t)gridDim.x * WAVES;
    // A row starts with three dependent reads (row id -> row pointer -> the first 128 indices) before its first gather can be
    // issued: ~3 memory latencies in front of ~25 us of work.  They are taken off the path: the next row's id is read at the
    // top of a row, its pointer after the accumulation, its first indices before the solve -- each is in flight while the
    // current row computes.  (idx[0] exists even without feedback: DevBuf never allocates less than one element.)
    auto first_indices = [&](int64_t beg_, int n_, int &i0, int &i1) {
        i0 = idx[lane < n_ ? beg_ + lane : 0];
        i1 = idx[64 + lane < n_ ? beg_ + 64 + lane : 0];
    };
    int64_t u = 0, beg = 0;
    int n = 0, idx0 = 0, idx1 = 0;
    if (wave < n_rows) {
        u = rows[wave];
        beg = ptr[u];
        n = (int)(ptr[u + 1] - beg);
        first_indices(beg, n, idx0, idx1);
    }
    // phased: the eight waves of the workgroup accumulate together and solve together (a barrier in between and one after): a
    // solving wave then never shares its SIMD with a wave that streams 64-cycle fp32 MFMAs.  Every wave runs the same number of
    // iterations; one without a row of its own (the tail) only keeps the barriers.
    const int64_t t_end = phased ? (n_rows + nwaves - 1) / nwaves * nwaves : n_rows;
    SolveState<DD> held;  // the first row of a pair, waiting for the second
    bool have_held = false;
    for (int64_t t = wave; t < t_end; t += nwaves) {
        if (t >= n_rows) {
            __syncthreads();
            __syncthreads();
            continue;
        }
        const int64_t u_next = rows[t + nwaves < n_rows ? t + nwaves : t];
        std::conditional_t<T16, GramAcc16<NB>, GramAcc<NB>> g;
        unsigned long long t0 = 0;
        if (prof) t0 = __builtin_amdgcn_s_memtime();
        if constexpr (T16)
            gram_accumulate16<NB, true, FULLD, MODE == 4>(B, idx + beg, n, d, lane, g, idx0, idx1, zero_row);
        else
            gram_accumulate_b3<NB, kAlsRowDeep>(B, idx + beg, n, d, lane, g, idx0, idx1, zero_row);
        const int64_t beg_next = ptr[u_next];
        const int64_t end_next = ptr[u_next + 1];
        if (prof) {
            // the accumulators are only complete once they are read: touch one so that the stamp waits for the MFMAs
            const unsigned long long t1 = __builtin_amdgcn_s_memtime() + (__float_as_int(g.t[0][0]) & 0);
            c_acc += t1 - t0;
            t0 = t1;
        }
        {   // rows / columns past d are zeros of the padded gathers: stored unconditionally (sM is DD x DP), with
            // immediate offsets from two lane-dependent bases
            auto put = [&](float *direct, float *mirror) {
                return [=](int ci, int cj, float v, bool mir) {
                    if (mir)
                        mirror[cj * DP + ci] = v;
                    else
                        direct[ci * DP + cj] = v;
                };
            };
            if constexpr (T16)
                gram_foreach16<NB>(g, put(sM + 4 * (lane >> 4) * DP + (lane & 15), sM + (lane & 15) * DP + 4 * (lane >> 4)));
            else
                gram_foreach<NB>(g, put(sM + 4 * (lane >> 5) * DP + (lane & 31), sM + (lane & 31) * DP + 4 * (lane >> 5)));
        }
        if constexpr (T16)
            gram_store_sums16<NB>(g, lane, ss);  // (ss holds DD words: the sums of the padding columns are zeros)
        else
            gram_store_sums<NB>(g, d, lane, ss);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (prof) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            c_m += t1 - t0;
            t0 = t1;
        }
        // The solve is a chain of ~1100 dependent VALU operations; the sibling wave of this SIMD is meanwhile streaming fp32
        // MFMAs, and at equal priority the arbiter lets one VALU operation through per 64-cycle MFMA (840 cycles per solve
        // step, profiles/r02_g_probe_als_prof.txt).  At raised priority the chain issues at its own pace and the MFMA stream
        // takes the slots in between -- it needs one issue per 64 cycles.
        const int n_next = (int)(end_next - beg_next);
        int idx0_next, idx1_next;
        first_indices(beg_next, n_next, idx0_next, idx1_next);
        if (phased) __syncthreads();
        __builtin_amdgcn_s_setprio(3);
        {
            // two rows' sweeps run together: the first row of a pair is taken as far as the chain (its M in registers, the LDS
            // buffer free for the second row's), the chains of both then advance step by step (SolveSteps::run2)
            SolveState<DD> cur;
            als_solve_prepare<DD, true, true, DP, !FULLD>(cur, A + u * d, sM, ss, sS, DD, one_w, w, reg, lane, prof ? &c_load : nullptr, d);
            if (have_held) {
                SolveSteps<0, DD, true>::run2(held, cur, d);
                als_solve_finish<DD>(held, lane);
                als_solve_finish<DD>(cur, lane);
                have_held = false;
            } else if (kAlsRowPair && !phased && t + nwaves < n_rows) {
                held = cur;
                have_held = true;
            } else {
                SolveSteps<0, DD, true>::run(cur, d);
                als_solve_finish<DD>(cur, lane);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_wave_barrier();
        if (prof) {
            c_solve += __builtin_amdgcn_s_memtime() - t0;
            c_rows++;
            c_ent += n;
        }
        if (phased) __syncthreads();
        u = u_next, beg = beg_next, n = n_next, idx0 = idx0_next, idx1 = idx1_next;
    }
    if (prof && lane == 0) {
        atomicAdd(prof + 0, c_acc);
        atomicAdd(prof + 1, c_m);
        atomicAdd(prof + 2, c_solve);
        atomicAdd(prof + 3, c_rows);
        atomicAdd(prof + 4, c_ent);
        atomicAdd(prof + 5, (unsigned long long)__builtin_amdgcn_s_memtime() - t_begin);
        atomicAdd(prof + 6, 1ull);
        atomicAdd(prof + 7, c_load);
    }
}

// long rows, stage 1: one wave per chunk -> partial[c] = [G (d x d, full) | s (d)]
template <int NB, int MODE = 1>  // (modes: als_row_kernel)
__global__ __launch_bounds__(64 * kAlsWaves, 2) void als_chunk_kernel(const float *__restrict__ B,
                                                                   const int32_t *__restrict__ idx,
                                                                   const int64_t *__restrict__ chunk_beg,
                                                                   const int32_t *__restrict__ chunk_cnt,
                                                                   int64_t n_chunks, int d, float *__restrict__ partial,
                                                                   const float *__restrict__ zeros, int zero_row,
                                                                   const int32_t *__restrict__ order) {
    static_assert(MODE >= 1 && MODE <= 4, "als_chunk_kernel modes");
    constexpr bool T16 = MODE != 2;
    constexpr bool FULLD = MODE == 1 || MODE == 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * kAlsWaves + wv, nwaves = (int64_t)gridDim.x * kAlsWaves;
    const int64_t stride = (int64_t)d * d + d;
    // `order` (may be null): the chunks longest first.  A wave takes the chunks wave, wave + nwaves, ... of that list: the 16K-entry
    // chunks of the most popular rows (a row has at most 256 chunks) start first instead of wherever their rows' numbers put them
    for (int64_t ci = wave; ci < n_chunks; ci += nwaves) {
        const int64_t c = order ? order[ci] : ci;
        std::conditional_t<T16, GramAcc16<NB>, GramAcc<NB>> g;
        const int32_t *fb = idx + chunk_beg[c];
        const int cn = chunk_cnt[c];
        if constexpr (T16)
            gram_accumulate16<NB, true, FULLD, MODE == 4>(B, fb, cn, d, lane, g, fb[lane < cn ? lane : 0], fb[64 + lane < cn ? 64 + lane : 0], zero_row);
        else
            gram_accumulate_b3<NB>(B, fb, cn, d, lane, g, fb[lane < cn ? lane : 0], fb[64 + lane < cn ? 64 + lane : 0], zero_row);
        float *dst = partial + c * stride;
        if constexpr (FULLD) {
            // d is a compile-time fact in these forms: the chunk's base + two lane offsets, every element at an immediate offset
            // (with an address per element the kernel held 227 registers; four waves per SIMD instead of two, which 128 allow, are
            // no faster: 0.455 against 0.413 ms per launch at C5, profiles/r04_zi_kernel_stats_als.txt -- the bound stays two)
            constexpr int DD = MODE == 1 ? 16 * NB : 32 * NB, SH = MODE == 1 ? 4 : 5, MK = (1 << SH) - 1;
            float *direct = dst + 4 * (lane >> SH) * DD + (lane & MK), *mirror = dst + (lane & MK) * DD + 4 * (lane >> SH);
            auto put = [&](int ci, int cj, float v, bool mir) {
                if (mir)
                    mirror[cj * DD + ci] = v;
                else
                    direct[ci * DD + cj] = v;
            };
            if constexpr (T16) {
                gram_foreach16<NB>(g, put);
                gram_store_sums16<NB>(g, lane, dst + DD * DD);
            } else {
                gram_foreach<NB>(g, put);
                gram_store_sums<NB>(g, DD, lane, dst + DD * DD);
            }
        } else {  // modes 3, 4: d x d of the 16 NB x 16 NB tiles
            // (an opaque copy of d: the stores' bound checks are invariant across the chunk loop, and with d as written their masks
            // stay in scalar registers over the accumulation -- 34 / 38 spills at NB = 4)
            int dq = d;
            asm volatile("" : "+s"(dq));
            gram_foreach16<NB>(g, [&](int ci, int cj, float v, bool mir) {
                const int i = ci + 4 * (lane >> 4), j = cj + (lane & 15);
                if (i < dq && j < dq) dst[mir ? j * d + i : i * d + j] = v;
            });
            gram_store_sums16<NB>(g, lane, dst + (int64_t)d * d, d);
        }
    }
}

// long rows, stage 2: one workgroup per row adds the row's partials in chunk order, wave 0 solves.  DMAX = d rounded up to a
// multiple of 16, FULL: d == DMAX (round 5: the one 64-step form with its bound checks held 300 scalar spills)
template <int DMAX>
__global__ __launch_bounds__(256) void als_long_solve_kernel(float *__restrict__ A, const float *__restrict__ S,
                                                             const int32_t *__restrict__ rows,
                                                             const int32_t *__restrict__ first,
                                                             const int32_t *__restrict__ nch, int64_t n_rows, int d,
                                                             float w, float reg, const float *__restrict__ partial) {
    __shared__ float sM[DMAX * kAlsDP];
    __shared__ float ss[DMAX];
    const float one_w = 1 - w;
    const int64_t stride = (int64_t)d * d + d;
    // M and the sums padded with zeros to DMAX (als_solve_prepare, PAD): the words past d are written once
    for (int e = threadIdx.x; e < DMAX * kAlsDP; e += blockDim.x) sM[e] = 0.0f;
    if (threadIdx.x < DMAX) ss[threadIdx.x] = 0.0f;
    __syncthreads();
    for (int64_t t = blockIdx.x; t < n_rows; t += gridDim.x) {
        const float *src = partial + (int64_t)first[t] * stride;
        const int nc = nch[t];
        for (int e = threadIdx.x; e < (int)stride; e += blockDim.x) {
            // chunk order, four independent chains (chunks c = 0, 1, 2, 3 mod 4) so that four loads are in flight;
            // the grouping is fixed, hence deterministic and the same whichever process solves the row
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            int c = 0;
            for (; c + 3 < nc; c += 4) {
                a0 += src[(int64_t)c * stride + e];
                a1 += src[(int64_t)(c + 1) * stride + e];
                a2 += src[(int64_t)(c + 2) * stride + e];
                a3 += src[(int64_t)(c + 3) * stride + e];
            }
            for (; c < nc; c++) a0 += src[(int64_t)c * stride + e];
            const float acc = (a0 + a1) + (a2 + a3);
            if (e < d * d)
                sM[(e / d) * kAlsDP + e % d] = one_w * acc + w * S[e];
            else
                ss[e - d * d] = acc;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            SolveState<DMAX> x;
            als_solve_prepare<DMAX, false, true, kAlsDP, true>(x, A + (int64_t)rows[t] * d, sM, ss, S, DMAX, one_w, w, reg, threadIdx.x, nullptr, d);
            SolveSteps<0, DMAX, true>::run(x, DMAX);
            als_solve_finish<DMAX>(x, threadIdx.x);
        }
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
                                                                                              dd, h->gram.p, dd);
    } else {
        als_gram_naive_kernel<<<dim3((unsigned)ceil_div(dd, 256)), dim3(256), 0, h->stream>>>(F, ptr, rows, d, h->gram.p);
    }
    GORSE_HIP_CHECK(hipGetLastError());
    h->prof.end(tok, h->stream);
    return GORSE_OK;
}

int als_zero_row(const gorse_mf *h, const float *F);
bool als_split_products();
int32_t run_sweep(gorse_mf *h, float *A, const float *B, const int64_t *ptr, const int32_t *idx, int64_t row_begin,
                  int64_t row_end, int64_t max_row, float w, float reg) {
    const int d = h->d;
    // 64 < nFactors <= 128, the product's choice: the Gram form of als_wide_kernel over the row plan (short rows whole, long rows
    // by chunks + als_wide_long_kernel); the residual sweep below stays as the forced path 1
    if (d > 64 && d <= 128 && g_als_path == 0) {
        gorse_mf::AlsPlan &pl = h->als_plan[A == h->P.p ? 0 : 1];
        const size_t wlds = ((size_t)128 * kWideLd + (size_t)kWideBatch * 128 + 128) * sizeof(float);
        // G by fused multiply-adds / the fp32 MFMA / the bf16 MFMA over split values (32-bit gather offsets: als_zero_row's condition)
        const int form = g_als_wide_fma ? 0 : ((!als_split_products() || als_zero_row(h, B) < 0) ? 1 : 2);
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_wide_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_wide_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_wide_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_wide_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_wide_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_wide_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_wide_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        unsigned long long *wprof = nullptr;
        if (g_als_prof) {
            const int side = A == h->P.p ? 0 : 1;
            GORSE_TRY(h->als_prof.ensure(16));
            GORSE_HIP_CHECK(hipMemsetAsync(h->als_prof.p + 8 * side, 0, 8 * sizeof(unsigned long long), h->stream));
            wprof = h->als_prof.p + 8 * side;
        }
        const int tokw = h->prof.begin(GORSE_PROF_ALS_SWEEP, h->stream);
        if (pl.n_short > 0) {
            auto k = form == 2 ? als_wide_kernel<false, 2> : (form == 1 ? als_wide_kernel<false, 1> : als_wide_kernel<false, 0>);
            k<<<dim3((unsigned)std::min<int64_t>(pl.n_short, 512)), dim3(256), wlds, h->stream>>>(
                A, B, ptr, idx, h->gram.p, pl.short_rows.p, nullptr, nullptr, pl.n_short, d, w, reg, nullptr, g_als_wide_probe, wprof);
            GORSE_HIP_CHECK(hipGetLastError());
        }
        if (pl.n_long > 0) {
            GORSE_TRY(h->als_partial.ensure((size_t)pl.n_chunks * kWidePartial));
            auto k = form == 2 ? als_wide_kernel<true, 2> : (form == 1 ? als_wide_kernel<true, 1> : als_wide_kernel<true, 0>);
            k<<<dim3((unsigned)std::min<int64_t>(pl.n_chunks, 2048)), dim3(256), wlds, h->stream>>>(
                A, B, ptr, idx, h->gram.p, nullptr, pl.chunk_beg.p, pl.chunk_cnt.p, pl.n_chunks, d, w, reg, h->als_partial.p, 0, nullptr);
            GORSE_HIP_CHECK(hipGetLastError());
            als_wide_long_kernel<<<dim3((unsigned)std::min<int64_t>(pl.n_long, 512)), dim3(256), wlds, h->stream>>>(
                A, h->gram.p, pl.long_rows.p, pl.long_first.p, pl.long_nch.p, pl.n_long, d, w, reg, h->als_partial.p);
            GORSE_HIP_CHECK(hipGetLastError());
        }
        h->prof.end(tokw, h->stream);
        return GORSE_OK;
    }
    const size_t fixed = ((size_t)((d + 3) & ~3) + 12 + 2 * (size_t)kGroupsPerBlock * d) * sizeof(float);
    const size_t budget = 64 * 1024;
    if (fixed + 1024 > 150 * 1024) return fail(GORSE_ERR_INVALID, "nFactors %d too large for the ALS kernel", d);
    int pred_cap = 4096, q_cap = 0;
    if (fixed + (size_t)pred_cap * 4 < budget) q_cap = (int)((budget - fixed - (size_t)pred_cap * 4) / ((size_t)(d + 1) * 4));
    while (fixed + (size_t)pred_cap * 4 + (size_t)q_cap * (d + 1) * 4 > 150 * 1024 && pred_cap > 256) pred_cap /= 2;
    const size_t shmem = fixed + (size_t)pred_cap * 4 + (size_t)q_cap * (d + 1) * 4;
    if (row_end <= row_begin) return GORSE_OK;
    int blocks = (int)std::min<int64_t>(row_end - row_begin, 256 * 8);
    const int64_t stride = max_row > pred_cap ? max_row : 1;
    GORSE_TRY(h->als_scratch.ensure((size_t)blocks * stride));
    GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)shmem));
    int tok = h->prof.begin(GORSE_PROF_ALS_SWEEP, h->stream);
    als_sweep_kernel<<<dim3(blocks), dim3(256), shmem, h->stream>>>(A, B, ptr, idx, h->gram.p, row_begin, row_end, d, w, reg,
                                                                   pred_cap, q_cap, h->als_scratch.p, stride, 0);
    GORSE_HIP_CHECK(hipGetLastError());
    h->prof.end(tok, h->stream);
    return GORSE_OK;
}


// probe: 8 counters per side in h->als_prof when the hook is on
unsigned long long *als_prof_slot(gorse_mf *h, int side) { return g_als_prof && h->als_prof.n >= 16 ? h->als_prof.p + 8 * side : nullptr; }

// the row id of the zero row behind matrix F (mf.hip) when 32-bit gather offsets apply (offsets in 32 bits, row ids in 24); else -1
int als_zero_row(const gorse_mf *h, const float *F) {
    const int64_t rows = F == h->P.p ? h->U : h->I;
    const int d = h->d;
    if (g_als_slow_gather || rows + 1 >= ((int64_t)1 << 24) || (rows + 1) * d * 4 >= ((int64_t)1 << 32)) return -1;
    return (int)rows;
}
// the zero row itself (every factor matrix has one behind its last row: mf.hip)
int als_pad_row(const gorse_mf *h, const float *F) { return (int)(F == h->P.p ? h->U : h->I); }

// how the Gram of a row is accumulated (als_row_kernel / als_chunk_kernel MODE), d <= 64: 2 = bf16 MFMA on three-way split values
// (d = 32 or 64), 1 = fp32 MFMA in 16 x 16 tiles (d = 16, 48), 3 = the same tiles for any other d (columns past d from the zero row),
// 4 = mode 3 with 64-bit gather addresses (als_zero_row < 0: a matrix of 4 GB and more, or the test hook 64).  Hook 128: mode 3 for
// every d (the generic form against the specialised ones).
// The environment variable GORSE_ALS_GRAM = "fp32" keeps every product of the Gram on the fp32 MFMA (read ONCE, when the library first
// needs it -- it cannot be changed per handle afterwards); the default lets nFactors 32 / 64 and 65..128 form them from three-way
// split floats on the bf16 MFMA (gram_accumulate_b3).
bool als_split_products() {
    static const bool fp32_only = [] {
        const char *e = getenv("GORSE_ALS_GRAM");
        return e && !strcmp(e, "fp32");
    }();
    return !fp32_only && !g_als_nob3;
}
int als_gram_mode(int d, int zrow) {
    if (zrow < 0) return 4;
    if (g_als_tile32) return 3;
    if (als_split_products() && d % 32 == 0) return 2;
    if (d % 16 == 0) return 1;
    return 3;
}

void launch_chunks(const float *B, const int32_t *idx, const int64_t *chunk_beg, const int32_t *chunk_cnt, int64_t n_chunks, int d,
                   float *partial, const float *zeros, int zrow, int pad_row, unsigned grid, hipStream_t st, const int32_t *order = nullptr) {
#define CHUNK_LAUNCH(...) \
    als_chunk_kernel<__VA_ARGS__><<<dim3(grid), dim3(64 * kAlsWaves), 0, st>>>(B, idx, chunk_beg, chunk_cnt, n_chunks, d, partial, zeros, pad_row, order)
    const int mode = als_gram_mode(d, zrow);
    switch (mode * 10 + (mode == 2 ? d / 32 : (d + 15) / 16)) {
    case 11: CHUNK_LAUNCH(1, 1); break;
    case 12: CHUNK_LAUNCH(2, 1); break;
    case 13: CHUNK_LAUNCH(3, 1); break;
    case 14: CHUNK_LAUNCH(4, 1); break;
    case 21: CHUNK_LAUNCH(1, 2); break;
    case 22: CHUNK_LAUNCH(2, 2); break;
    case 31: CHUNK_LAUNCH(1, 3); break;
    case 32: CHUNK_LAUNCH(2, 3); break;
    case 33: CHUNK_LAUNCH(3, 3); break;
    case 34: CHUNK_LAUNCH(4, 3); break;
    case 41: CHUNK_LAUNCH(1, 4); break;
    case 42: CHUNK_LAUNCH(2, 4); break;
    case 43: CHUNK_LAUNCH(3, 4); break;
    default: CHUNK_LAUNCH(4, 4); break;
    }
#undef CHUNK_LAUNCH
}

// long rows, stage 1 1/2: the partials of a row added up by ONE THREAD PER ELEMENT, for rows with many chunks.  als_long_solve_kernel
// sums a row's partials with the 256 threads of its one workgroup, seventeen elements per thread one after the other, four loads in
// flight each: the most popular row of C5's item side (256 partials of 16.6 KB) was 0.52 ms of a 5.6 ms epoch, the rest of the chip
// waiting for that workgroup.  Here every element of every long row has its own thread (the row's 256 partials = 64 dependent steps,
// once), the solve then reads ONE partial per row.  The same four chains by chunk number mod 4 and the same (a0 + a1) + (a2 + a3) as
// in als_long_solve_kernel, which then adds 0 to the result: the row's solution is the same in every bit.
__global__ __launch_bounds__(256) void als_partial_reduce_kernel(const float *__restrict__ partial, const int32_t *__restrict__ first,
                                                                 const int32_t *__restrict__ nch, int64_t n_rows, int stride,
                                                                 float *__restrict__ out) {
    const int nb = (stride + 255) / 256;
    for (int64_t b = blockIdx.x; b < n_rows * nb; b += gridDim.x) {
        const int64_t t = b / nb;
        const int e = (int)(b - t * nb) * 256 + threadIdx.x;
        if (e >= stride) continue;
        const float *src = partial + (int64_t)first[t] * stride + e;
        const int nc = nch[t];
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int c = 0;
        for (; c + 15 < nc; c += 16) {  // sixteen loads in flight; every chain still adds its chunks in order
            float x[16];
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = src[(int64_t)(c + j) * stride];
#pragma unroll
            for (int j = 0; j < 16; j += 4) a0 += x[j], a1 += x[j + 1], a2 += x[j + 2], a3 += x[j + 3];
        }
        for (; c + 3 < nc; c += 4) {
            a0 += src[(int64_t)c * stride];
            a1 += src[(int64_t)(c + 1) * stride];
            a2 += src[(int64_t)(c + 2) * stride];
            a3 += src[(int64_t)(c + 3) * stride];
        }
        for (; c < nc; c++) a0 += src[(int64_t)c * stride];
        out[t * stride + e] = (a0 + a1) + (a2 + a3);
    }
}

void launch_long_solve(float *A, const float *S, const int32_t *rows, const int32_t *first, const int32_t *nch, int64_t n_rows, int d,
                       float w, float reg, const float *partial, hipStream_t st) {
#define LONG_LAUNCH(DMAX_) \
    als_long_solve_kernel<DMAX_><<<dim3((unsigned)std::min<int64_t>(n_rows, 1024)), dim3(256), 0, st>>>(A, S, rows, first, nch, n_rows, d, w, reg, partial)
    switch ((d + 15) / 16) {
    case 1: LONG_LAUNCH(16); break;
    case 2: LONG_LAUNCH(32); break;
    case 3: LONG_LAUNCH(48); break;
    default: LONG_LAUNCH(64); break;
    }
#undef LONG_LAUNCH
}

int32_t run_side_gram(gorse_mf *h, int side, float *A, const float *B, const int64_t *ptr, const int32_t *idx, float w,
                      float reg) {
    const int d = h->d;
    const int zrow = als_zero_row(h, B), pad_row = als_pad_row(h, B);
    gorse_mf::AlsPlan &pl = h->als_plan[side];
    if (g_als_prof) {
        GORSE_TRY(h->als_prof.ensure(16));
        GORSE_HIP_CHECK(hipMemsetAsync(h->als_prof.p + 8 * side, 0, 8 * sizeof(unsigned long long), h->stream));
    }
    int tok = h->prof.begin(GORSE_PROF_ALS_SWEEP, h->stream);
    if (pl.n_short > 0) {
#define ROW_LAUNCH(WAVES_, LDS_, ...)                                                                                  \
    do {                                                                                                               \
        const unsigned grid_ = (unsigned)std::min<int64_t>(ceil_div(pl.n_short, WAVES_), 256); /* one workgroup per CU */ \
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)als_row_kernel<__VA_ARGS__, WAVES_>,                         \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_)));                 \
        als_row_kernel<__VA_ARGS__, WAVES_><<<dim3(grid_), dim3(64 * WAVES_), (LDS_), h->stream>>>(                    \
            A, B, ptr, idx, h->gram.p, pl.short_rows.p, pl.n_short, d, w, reg, h->als_zeros.p, als_prof_slot(h, side), \
            g_als_phased, pad_row);                                                                                    \
    } while (0)
        const int mode = als_gram_mode(d, zrow);
        {
            // DD x (DD + 1) + DD words of M and sums per wave next to S (DD = d, or d rounded up to whole 16-column blocks in modes 3, 4).
            // Twelve waves per CU where they fit and pay: d = 16 0.86 -> 0.77 ms, d = 32 1.30 -> 1.26 (C5 shard / 4); d = 48 is no
            // faster with twelve, and d = 64 (M packed as the block rows of its upper triangle so that twelve buffers fit) was slower:
            // 3.06 against 2.89 ms (profiles/r04_t_probe_als_tiles*.txt)
            const int dd = mode == 2 ? d : (d + 15) / 16 * 16;
            const int wv16 = (dd <= 32 && !g_als_waves8 && mode != 4) ? 12 : 8;
            const size_t lds16 = ((size_t)dd * dd + (size_t)wv16 * ((size_t)dd * (dd + 1) + dd)) * sizeof(float);
            switch (mode * 1000 + (mode == 2 ? d / 32 * 2 : dd / 16) * 100 + wv16) {
            case 1108: ROW_LAUNCH(8, lds16, 1, 1); break;
            case 1112: ROW_LAUNCH(12, lds16, 1, 1); break;
            case 1208: ROW_LAUNCH(8, lds16, 2, 1); break;
            case 1212: ROW_LAUNCH(12, lds16, 2, 1); break;
            case 1308: ROW_LAUNCH(8, lds16, 3, 1); break;
            case 1408: ROW_LAUNCH(8, lds16, 4, 1); break;
            case 2208: ROW_LAUNCH(8, lds16, 1, 2); break;
            case 2212: ROW_LAUNCH(12, lds16, 1, 2); break;
            case 2408: ROW_LAUNCH(8, lds16, 2, 2); break;
            case 3108: ROW_LAUNCH(8, lds16, 1, 3); break;
            case 3112: ROW_LAUNCH(12, lds16, 1, 3); break;
            case 3208: ROW_LAUNCH(8, lds16, 2, 3); break;
            case 3212: ROW_LAUNCH(12, lds16, 2, 3); break;
            case 3308: ROW_LAUNCH(8, lds16, 3, 3); break;
            case 3408: ROW_LAUNCH(8, lds16, 4, 3); break;
            case 4108: ROW_LAUNCH(8, lds16, 1, 4); break;
            case 4208: ROW_LAUNCH(8, lds16, 2, 4); break;
            case 4308: ROW_LAUNCH(8, lds16, 3, 4); break;
            case 4408: ROW_LAUNCH(8, lds16, 4, 4); break;
            default: return fail(GORSE_ERR_INVALID, "no ALS row kernel for nFactors %d (mode %d)", d, mode);
            }
        }
#undef ROW_LAUNCH
        GORSE_HIP_CHECK(hipGetLastError());
    }
    if (pl.n_long > 0) {
        GORSE_TRY(h->als_partial.ensure((size_t)pl.n_chunks * ((size_t)d * d + d)));
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(pl.n_chunks, kAlsWaves), 2048);
        launch_chunks(B, idx, pl.chunk_beg.p, pl.chunk_cnt.p, pl.n_chunks, d, h->als_partial.p, h->als_zeros.p, zrow, pad_row, grid, h->stream,
                      pl.chunk_order.p);
        GORSE_HIP_CHECK(hipGetLastError());
        if (pl.max_nch > 8 && !g_als_solve_adds) {  // (path | 2048: the solve kernel adds the partials itself whatever their number -- tests)
            const int stride = d * d + d;
            GORSE_TRY(h->als_reduced.ensure((size_t)pl.n_long * stride));
            const int64_t blocks = std::min<int64_t>(pl.n_long * ceil_div(stride, 256), (int64_t)1 << 20);
            als_partial_reduce_kernel<<<dim3((unsigned)blocks), dim3(256), 0, h->stream>>>(h->als_partial.p, pl.long_first.p, pl.long_nch.p,
                                                                                        pl.n_long, stride, h->als_reduced.p);
            GORSE_HIP_CHECK(hipGetLastError());
            launch_long_solve(A, h->gram.p, pl.long_rows.p, pl.long_ident.p, pl.long_one.p, pl.n_long, d, w, reg, h->als_reduced.p, h->stream);
        } else {
            launch_long_solve(A, h->gram.p, pl.long_rows.p, pl.long_first.p, pl.long_nch.p, pl.n_long, d, w, reg, h->als_partial.p, h->stream);
        }
        GORSE_HIP_CHECK(hipGetLastError());
    }
    h->prof.end(tok, h->stream);
    return GORSE_OK;
}

// S = sum over the rows of F with feedback of x x^T on the fp32 MFMA: the rows-with-feedback list of `side` is the
// "feedback list", cut into chunks; partial Gram matrices are added in chunk order (deterministic, bitwise symmetric)
int32_t run_gram_mfma(gorse_mf *h, const float *F, int side) {
    const int d = h->d, dd = d * d;
    const int zrow = als_zero_row(h, F), pad_row = als_pad_row(h, F);
    gorse_mf::AlsPlan &pl = h->als_plan[side];
    GORSE_TRY(h->gram.ensure((size_t)dd));
    int tok = h->prof.begin(GORSE_PROF_ALS_GRAM, h->stream);
    if (pl.n_gchunks == 0) {
        GORSE_HIP_CHECK(hipMemsetAsync(h->gram.p, 0, (size_t)dd * sizeof(float), h->stream));
    } else {
        const int64_t stride = (int64_t)dd + d;
        GORSE_TRY(h->gram_partial.ensure((size_t)pl.n_gchunks * stride));
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(pl.n_gchunks, kAlsWaves), 2048);
        launch_chunks(F, pl.fb_rows.p, pl.g_beg.p, pl.g_cnt.p, pl.n_gchunks, d, h->gram_partial.p, h->als_zeros.p, zrow, pad_row, grid, h->stream);
        GORSE_HIP_CHECK(hipGetLastError());
        als_gram_reduce_kernel<<<dim3((unsigned)ceil_div(dd, 256)), dim3(256), 0, h->stream>>>(
            h->gram_partial.p, (int)pl.n_gchunks, dd, h->gram.p, stride);
        GORSE_HIP_CHECK(hipGetLastError());
    }
    h->prof.end(tok, h->stream);
    return GORSE_OK;
}

// the same for 64 < nFactors <= 128: the chunks go through als_wide_kernel's MFMA form (partial G of 128 x 128 + sums per chunk),
// added in chunk order
__global__ void als_gram_reduce_wide_kernel(const float *__restrict__ partial, int nparts, int d, float *__restrict__ S) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d * d) return;
    const int i = e / d, j = e % d;
    float acc = 0.0f;
    for (int c = 0; c < nparts; c++) acc += partial[(int64_t)c * kWidePartial + i * 128 + j];
    S[e] = acc;
}
int32_t run_gram_mfma_wide(gorse_mf *h, const float *F, int side) {
    const int d = h->d, dd = d * d;
    gorse_mf::AlsPlan &pl = h->als_plan[side];
    GORSE_TRY(h->gram.ensure((size_t)dd));
    int tok = h->prof.begin(GORSE_PROF_ALS_GRAM, h->stream);
    if (pl.n_gchunks == 0) {
        GORSE_HIP_CHECK(hipMemsetAsync(h->gram.p, 0, (size_t)dd * sizeof(float), h->stream));
    } else {
        GORSE_TRY(h->gram_partial.ensure((size_t)pl.n_gchunks * kWidePartial));
        const size_t wlds = ((size_t)128 * kWideLd + (size_t)kWideBatch * 128 + 128) * sizeof(float);
        auto gk = (!als_split_products() || als_zero_row(h, F) < 0) ? als_wide_kernel<true, 1> : als_wide_kernel<true, 2>;
        GORSE_HIP_CHECK(hipFuncSetAttribute((const void *)gk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        gk<<<dim3((unsigned)std::min<int64_t>(pl.n_gchunks, 2048)), dim3(256), wlds, h->stream>>>(
            nullptr, F, nullptr, pl.fb_rows.p, nullptr, nullptr, pl.g_beg.p, pl.g_cnt.p, pl.n_gchunks, d, 0.0f, 0.0f, h->gram_partial.p, 0,
            nullptr);
        GORSE_HIP_CHECK(hipGetLastError());
        als_gram_reduce_wide_kernel<<<dim3((unsigned)ceil_div(dd, 256)), dim3(256), 0, h->stream>>>(h->gram_partial.p, (int)pl.n_gchunks,
                                                                                                   d, h->gram.p);
        GORSE_HIP_CHECK(hipGetLastError());
    }
    h->prof.end(tok, h->stream);
    return GORSE_OK;
}

bool use_gram_form(const gorse_mf *h) { return g_als_path == 2 || (g_als_path == 0 && h->d <= 64); }
bool use_wide_mfma(const gorse_mf *h) { return g_als_path == 0 && h->d > 64 && h->d <= 128 && !g_als_wide_fma; }

}  // namespace

namespace {
// one half of model.go:641-738: S over ALL rows (with feedback) of the other side, then the coordinate sweep of this
// handle's row range of `side` (0: user factors from item factors, 1: item factors from user factors)
int32_t half_epoch(gorse_mf *h, int side, float weight, float reg) {
    const bool gram_form = use_gram_form(h);
    float *A = side == 0 ? h->P.p : h->Q.p;
    const float *B = side == 0 ? h->Q.p : h->P.p;
    const int64_t *ptr = side == 0 ? h->uptr.p : h->iptr.p;
    const int32_t *idx = side == 0 ? h->uidx.p : h->iidx.p;
    if (gram_form) {
        GORSE_TRY(run_gram_mfma(h, B, 1 - side));
        GORSE_TRY(run_side_gram(h, side, A, B, ptr, idx, weight, reg));
    } else {
        if (use_wide_mfma(h))
            GORSE_TRY(run_gram_mfma_wide(h, B, 1 - side));
        else
            GORSE_TRY(run_gram(h, B, side == 0 ? h->iptr.p : h->uptr.p, side == 0 ? h->I : h->U, GORSE_PROF_ALS_GRAM));
        GORSE_TRY(run_sweep(h, A, B, ptr, idx, h->als_lo[side], h->als_hi[side],
                            side == 0 ? h->max_user_row : h->max_item_row, weight, reg));
    }
    return GORSE_OK;
}
int32_t als_check(gorse_mf *h) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (!h->has_item_csr) return fail(GORSE_ERR_INVALID, "ALS needs the item feedback CSR (item_indptr/item_indices)");
    if (g_als_path == 2 && h->d > 64) return fail(GORSE_ERR_INVALID, "the Gram-form ALS kernels cover nFactors <= 64 (got %d)", h->d);
    return GORSE_OK;
}
}  // namespace

extern "C" int32_t gorse_als_epoch(gorse_mf *h, float weight, float reg, const volatile int32_t *cancel) {
    GORSE_TRY(als_check(h));
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    if (cancel && *cancel) return fail(GORSE_ERR_CANCELLED, "cancelled");
    GORSE_TRY(half_epoch(h, 0, weight, reg));  // model.go:645-690
    if (cancel && *cancel) {
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        return fail(GORSE_ERR_CANCELLED, "cancelled");
    }
    GORSE_TRY(half_epoch(h, 1, weight, reg));  // model.go:693-738
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" int32_t gorse_als_half_epoch(gorse_mf *h, int32_t side, float weight, float reg) {
    GORSE_TRY(als_check(h));
    if (side != 0 && side != 1) return fail(GORSE_ERR_INVALID, "side must be 0 (users) or 1 (items)");
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    GORSE_TRY(half_epoch(h, side, weight, reg));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

// the same, only enqueued on the handle's stream: one process driving N handles (integration/go/model/cf/rccl_hip.go) enqueues
// the half-sweep of every device, then gorse_mf_rows_allgather (stream-ordered), and synchronises once per epoch -- through
// the synchronous call the N devices would solve their row ranges one after the other
extern "C" int32_t gorse_als_half_epoch_enqueue(gorse_mf *h, int32_t side, float weight, float reg) {
    GORSE_TRY(als_check(h));
    if (side != 0 && side != 1) return fail(GORSE_ERR_INVALID, "side must be 0 (users) or 1 (items)");
    GORSE_TRY(h->use());
    return half_epoch(h, side, weight, reg);
}

extern "C" int32_t gorse_als_set_ranges(gorse_mf *h, int64_t u_begin, int64_t u_end, int64_t i_begin, int64_t i_end) {
    GORSE_TRY(als_check(h));
    if (u_begin < 0 || u_end > h->U || u_begin > u_end || i_begin < 0 || i_end > h->I || i_begin > i_end)
        return fail(GORSE_ERR_RANGE, "bad row ranges [%lld,%lld) x [%lld,%lld)", (long long)u_begin, (long long)u_end,
                    (long long)i_begin, (long long)i_end);
    GORSE_TRY(h->use());
    GORSE_TRY(mf_sync_streams(h));
    h->als_lo[0] = u_begin;
    h->als_hi[0] = u_end;
    h->als_lo[1] = i_begin;
    h->als_hi[1] = i_end;
    GORSE_TRY(als_build_plan(h, 0, h->h_uptr.data(), h->U, u_begin, u_end));
    GORSE_TRY(als_build_plan(h, 1, h->h_iptr.data(), h->I, i_begin, i_end));
    return GORSE_OK;
}

// factor rows [begin, end) of one side <-> a caller-owned device buffer (the blocks an all-gather moves)
extern "C" int32_t gorse_mf_rows_export(gorse_mf *h, int32_t side, int64_t begin, int64_t end, float *dst) {
    if (!h || !dst) return fail(GORSE_ERR_INVALID, "NULL argument");
    const int64_t rows = side == 0 ? h->U : h->I;
    if ((side != 0 && side != 1) || begin < 0 || end > rows || begin > end) return fail(GORSE_ERR_RANGE, "bad row range");
    GORSE_TRY(h->use());
    const float *src = (side == 0 ? h->P.p : h->Q.p) + begin * h->d;
    GORSE_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)(end - begin) * h->d * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}
extern "C" int32_t gorse_mf_rows_import(gorse_mf *h, int32_t side, int64_t begin, int64_t end, const float *src) {
    if (!h || !src) return fail(GORSE_ERR_INVALID, "NULL argument");
    const int64_t rows = side == 0 ? h->U : h->I;
    if ((side != 0 && side != 1) || begin < 0 || end > rows || begin > end) return fail(GORSE_ERR_RANGE, "bad row range");
    GORSE_TRY(h->use());
    float *dst = (side == 0 ? h->P.p : h->Q.p) + begin * h->d;
    GORSE_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)(end - begin) * h->d * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" void gorse_hip_test_set_als_path(int32_t path) {
    g_als_path = path & 3;
    g_als_phased = (path & 4) != 0;
    g_als_wide_fma = (path & 8) != 0;
    g_als_wide_probe = (path >> 4) & 3;
    g_als_slow_gather = (path & 64) != 0;
    g_als_tile32 = (path & 128) != 0;
    g_als_waves8 = (path & 256) != 0;
    g_als_nob3 = (path & 1024) != 0;
    g_als_solve_adds = (path & 2048) != 0;
}
// probe: phase counters of als_row_kernel for the last half-sweep of each side (16 values: users, items)
extern "C" int32_t gorse_hip_test_als_profile(gorse_mf *h, int32_t enable, uint64_t *out16) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    g_als_prof = enable != 0;
    if (!out16) return GORSE_OK;
    if (h->als_prof.n < 16) return fail(GORSE_ERR_INVALID, "no profiled ALS sweep has run on this handle");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipMemcpyAsync(out16, h->als_prof.p, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}
// takes effect for handles created afterwards (the row plan is built in gorse_mf_create)
extern "C" void gorse_hip_test_set_als_plan(int32_t long_row, int32_t chunk) {
    g_als_long_row = long_row > 0 ? long_row : 0;  // 0: chosen per side by als_build_plan
    g_als_chunk = chunk > 0 ? chunk : 0;
}

namespace gorse {
// rows of one side -> short-row list (longest first) + chunks of the long rows; host CSR pointers only
int32_t als_build_plan(gorse_mf *h, int side, const int64_t *ptr, int64_t rows, int64_t lo, int64_t hi) {
    gorse_mf::AlsPlan &pl = h->als_plan[side];
    if (h->als_zeros.n < 64) {
        GORSE_TRY(h->als_zeros.alloc(64));
        GORSE_HIP_CHECK(hipMemsetAsync(h->als_zeros.p, 0, 64 * sizeof(float), h->stream));
    }
    std::vector<int32_t> shorts, lrows, lfirst, lnch, crow, ccnt;
    std::vector<int64_t> cbeg;
    // the threshold follows the side's size (csrc/als_plan.hpp, shared with the CPU test), taken from the WHOLE side, not from [lo, hi)
    int64_t long_row = g_als_long_row, chunk_len = g_als_chunk;
    if (long_row <= 0) long_row = als_long_row_threshold(ptr[rows] - ptr[0], h->d);
    if (chunk_len <= 0) chunk_len = long_row;
    for (int64_t r = lo; r < hi; r++) {  // the rows this handle solves; the Gram list below covers ALL rows
        const int64_t n = ptr[r + 1] - ptr[r];
        if (n <= long_row) {
            shorts.push_back((int32_t)r);
            continue;
        }
        lrows.push_back((int32_t)r);
        lfirst.push_back((int32_t)crow.size());
        // at most 256 chunks per row: very long rows get proportionally longer chunks, in multiples of one pipeline stage (16 entries).
        // (1024 chunks per row, so that the 4.1M-entry row of C5's item side comes in chunks of 4096 entries like everybody else's
        // instead of 16K: measured, no change -- the chunk kernel runs at the row kernel's rate per entry, not at its longest chunk's.)
        const int64_t chunk = std::max<int64_t>(chunk_len, (ceil_div(n, 256) + 15) / 16 * 16);
        int nc = 0;
        for (int64_t b = 0; b < n; b += chunk, nc++) {
            crow.push_back((int32_t)r);
            cbeg.push_back(ptr[r] + b);
            ccnt.push_back((int32_t)std::min<int64_t>(chunk, n - b));
        }
        lnch.push_back(nc);
    }
    std::vector<int32_t> fbr, gcnt;
    std::vector<int64_t> gbeg;
    for (int64_t r = 0; r < rows; r++)
        if (ptr[r + 1] > ptr[r]) fbr.push_back((int32_t)r);
    constexpr int64_t kGramChunk = 1024;
    for (int64_t b = 0; b < (int64_t)fbr.size(); b += kGramChunk) {
        gbeg.push_back(b);
        gcnt.push_back((int32_t)std::min<int64_t>(kGramChunk, (int64_t)fbr.size() - b));
    }
    pl.n_fb_rows = (int64_t)fbr.size();
    pl.n_gchunks = (int64_t)gbeg.size();
    std::stable_sort(shorts.begin(), shorts.end(),
                     [&](int32_t a, int32_t b) { return ptr[a + 1] - ptr[a] > ptr[b + 1] - ptr[b]; });
    pl.n_short = (int64_t)shorts.size();
    pl.n_long = (int64_t)lrows.size();
    pl.n_chunks = (int64_t)crow.size();
    auto up32 = [&](DevBuf<int32_t> &b, const std::vector<int32_t> &v) -> int32_t {
        GORSE_TRY(b.alloc(v.size()));
        if (!v.empty())
            GORSE_HIP_CHECK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        return GORSE_OK;
    };
    GORSE_TRY(up32(pl.short_rows, shorts));
    GORSE_TRY(up32(pl.long_rows, lrows));
    GORSE_TRY(up32(pl.long_first, lfirst));
    GORSE_TRY(up32(pl.long_nch, lnch));
    {   // the row list of a solve that reads ONE (already added up) partial per row: als_partial_reduce_kernel
        std::vector<int32_t> ident(lrows.size()), one(lrows.size(), 1);
        for (size_t k = 0; k < ident.size(); k++) ident[k] = (int32_t)k;
        GORSE_TRY(up32(pl.long_ident, ident));
        GORSE_TRY(up32(pl.long_one, one));
        pl.max_nch = 0;
        for (int32_t v : lnch) pl.max_nch = std::max(pl.max_nch, v);
    }
    GORSE_TRY(up32(pl.chunk_row, crow));
    GORSE_TRY(up32(pl.chunk_cnt, ccnt));
    {   // the chunks longest first (als_chunk_kernel's `order`)
        std::vector<int32_t> order(ccnt.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = (int32_t)k;
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return ccnt[a] > ccnt[b]; });
        GORSE_TRY(up32(pl.chunk_order, order));
    }
    GORSE_TRY(up32(pl.fb_rows, fbr));
    GORSE_TRY(up32(pl.g_cnt, gcnt));
    GORSE_TRY(pl.g_beg.alloc(gbeg.size()));
    if (!gbeg.empty())
        GORSE_HIP_CHECK(hipMemcpyAsync(pl.g_beg.p, gbeg.data(), gbeg.size() * sizeof(int64_t), hipMemcpyHostToDevice,
                                       h->stream));
    GORSE_TRY(pl.chunk_beg.alloc(cbeg.size()));
    if (!cbeg.empty())
        GORSE_HIP_CHECK(hipMemcpyAsync(pl.chunk_beg.p, cbeg.data(), cbeg.size() * sizeof(int64_t), hipMemcpyHostToDevice,
                                       h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // host vectors go out of scope
    return GORSE_OK;
}
}  // namespace gorse
