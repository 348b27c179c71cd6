// bpr.hip -- BPR training step on gfx950.  Reference: model/cf/model.go:446-494.
//
// Kernels per chunk of samples (the user-run schedule: the production Hogwild form for >= 4096 users and nFactors 8/16/32/64/128):
//   bpr_sample_user_kernel : model.go:452-458 -- sample s draws its USER from the counter-based Philox stream of sample s and
//                            takes its rank in that user's run; runs AHEAD on its own stream, like the rest of the preparation.
//   scan + bpr_scatter_ids : counting sort of the chunk's SAMPLE IDS by user (the order in which a Hogwild epoch applies its
//                            samples is free: parallel.go:44-68).
//   bpr_sample_items_kernel: model.go:459-468 -- one thread per sorted position replays its sample's stream and draws the
//                            positive and the negative; (i, j) are written in sorted order.
//   bpr_update_user_kernel : model.go:469-488 -- one 16-lane group applies ALL samples of one user: p_u is loaded
//                            once, updated in registers and stored once; q_i / q_j are gathered ahead (64-byte
//                            contiguous segments per load) and updated with fp32 atomics (hot items through replica
//                            rows) -- or, the negative of a COLD item, by one write-through store of fma(t, lr, row).
// Other schedules:
//   bpr_sample_kernel      : model.go:449-468 -- one thread per sample draws the whole triplet (the same stream): the
//                            per-sample and sequential schedules, gorse_bpr_sample_triplets.
//   bpr_update_kernel      : the per-sample form (one group per sample, three rows updated in place): the
//                            sequential parity schedule, the racy diagnostic, and the Hogwild schedule of shapes
//                            with few users or another factor width.
// HBM-bound: algorithmic bytes per sample = 6*d*4 (three rows read + three written) + 12 (indices).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "bpr_bins.hpp"
#include "mf_internal.hpp"
#include <utility>

using namespace gorse;

namespace {

int g_variant = 0;  // schedule switches of the tests and probes (gorse_hip_test_set_variant); nothing of it reaches a kernel
constexpr int MODE_ATOMIC = GORSE_BPR_HOGWILD_ATOMIC;
constexpr int MODE_EXACT = GORSE_BPR_SEQUENTIAL;
constexpr int MODE_RACY = GORSE_BPR_HOGWILD_RACY;
constexpr int MODE_STORES = GORSE_BPR_HOGWILD_STORES;  // MODE_ATOMIC with the cold negatives by store (user-run schedule only)

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
                                                         int32_t *__restrict__ js, int32_t *__restrict__ fail_count,
                                                         int32_t *__restrict__ bucket, int32_t *__restrict__ rank) {
    // bucket != null (user-run schedule): the sample also draws its arrival rank inside its user's run -- the first pass of
    // the counting sort by user, done here so that no separate pass over the triplets is needed (skipped samples: key U)
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
        if (bucket) rank[s] = atomicAdd(&bucket[u < 0 ? U : u], 1);
    }
}

// ---- sampling for the user-run schedule: user first, items by run -----------------------------
// The same triplets as bpr_sample_kernel (sample s = the Philox stream keyed by (seed, epoch, sample_base + s)), produced in two
// passes so that what is random about the memory accesses shrinks from ~13 cache lines per sample to ~3:
//   bpr_sample_user_kernel  : sample s draws its USER only (model.go:452-458; one look at the row pointers) and takes its
//                             arrival rank in that user's run (the count pass of the counting sort by user);
//   scan + bpr_scatter_ids  : position of sample s in the user-sorted order -> perm[position] = s, su[position] = its user
//                             (8 bytes scattered per sample instead of the whole triplet);
//   bpr_sample_items_kernel : one thread per SORTED position replays the stream of its sample up to the user draw (no memory:
//                             the first non-empty row drawn IS the user) and draws the positive and the negative
//                             (model.go:459-468) -- neighbouring threads hold samples of the same user, whose two item rows are
//                             read while they sit in the cache -- and writes (i, j) straight to the sorted position.
// A sample whose negative cannot be found (kMaxDraws rejections: a user holding nearly every item) keeps its place in the run
// with j = -1; bpr_update_user_kernel skips it.
__global__ __launch_bounds__(256) void bpr_sample_user_kernel(int32_t U, const int64_t *__restrict__ uptr, uint64_t seed,
                                                              uint64_t epoch, int64_t sample_base, int64_t n,
                                                              int32_t *__restrict__ key, int32_t *__restrict__ fail_count,
                                                              int32_t *__restrict__ bucket, int32_t *__restrict__ rank) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        Philox g;
        g.init(seed, epoch, (uint64_t)(sample_base + s));
        int32_t u = -1;
        for (int t = 0; t < kMaxDraws; t++) {
            const int32_t cu = g.int31n(U);
            if (uptr[cu + 1] > uptr[cu]) {
                u = cu;
                break;
            }
        }
        if (u < 0) atomicAdd(fail_count, 1);
        key[s] = u;
        rank[s] = atomicAdd(&bucket[u < 0 ? U : u], 1);
    }
}

__global__ __launch_bounds__(256) void bpr_scatter_ids_kernel(const int32_t *__restrict__ key, const int32_t *__restrict__ rank,
                                                              const int32_t *__restrict__ bucket, int64_t n, int32_t U,
                                                              int2 *__restrict__ pairs) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        const int32_t k = key[s] < 0 ? U : key[s];
        const int64_t pos = (int64_t)bucket[k] + rank[s];
        pairs[pos] = make_int2((int32_t)s, key[s]);  // (sample id, user) at the sorted position: one 8-byte store
    }
}

// the item draws of one sample from the start of its stream (model.go:452-468): the path of a sample whose first negative candidate
// was one of the user's own items -- the batch in bpr_sample_items_kernel only carries the first candidate
__device__ __forceinline__ int32_t draw_negative_again(int32_t U, int32_t I, const int32_t *__restrict__ row, int64_t cnt, int32_t u,
                                                    uint64_t seed, uint64_t epoch, uint64_t sample) {
    Philox g;
    g.init(seed, epoch, sample);
    for (int k = 0; k < kMaxDraws; k++)
        if (g.int31n(U) == u) break;
    (void)g.int31n((int32_t)cnt);  // the positive's draw
    for (int k = 0; k < kMaxDraws; k++) {
        const int32_t c = g.int31n(I);
        if (!row_contains(row, cnt, c)) return c;
    }
    return -1;
}

__global__ __launch_bounds__(256) void bpr_sample_items_kernel(int32_t U, int32_t I, const int64_t *__restrict__ uptr,
                                                               const int32_t *__restrict__ uidx,
                                                               const int32_t *__restrict__ usorted, uint64_t seed,
                                                               uint64_t epoch, int64_t sample_base, int64_t n,
                                                               const int2 *__restrict__ pairs, int32_t *__restrict__ si,
                                                               int32_t *__restrict__ sj, int32_t *__restrict__ fail_count) {
    // one thread per SORTED position: neighbouring threads hold samples of the same user, so the user's row pointers and its
    // two item rows are shared by the lanes of a wave (and by the waves that follow) while they sit in the cache
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int2 pr = pairs[t];  // (sample id, user)
        const int32_t u = pr.y;
        if (u < 0) continue;  // no user could be drawn: the positions behind the last run, never read
        const int64_t rbeg = uptr[u];
        const int64_t cnt = uptr[u + 1] - rbeg;
        Philox g;
        g.init(seed, epoch, (uint64_t)(sample_base + pr.x));
        // the user draw again: rows drawn before u were empty (u is the first non-empty one), no look-up needed
        for (int k = 0; k < kMaxDraws; k++)
            if (g.int31n(U) == u) break;
        int32_t pi = uidx[rbeg + g.int31n((int32_t)cnt)], nj = -1;
        for (int k = 0; k < kMaxDraws; k++) {
            const int32_t c = g.int31n(I);
            if (!row_contains(usorted + rbeg, cnt, c)) {
                nj = c;
                break;
            }
        }
        if (nj < 0) {
            atomicAdd(fail_count, 1);
            pi = -1;
        }
        si[t] = pi;
        sj[t] = nj;
    }
}

__global__ __launch_bounds__(256) void bpr_sample_items_batch_kernel(int32_t U, int32_t I, const int64_t *__restrict__ uptr,
                                                               const int32_t *__restrict__ uidx,
                                                               const int32_t *__restrict__ usorted, uint64_t seed,
                                                               uint64_t epoch, int64_t sample_base, int64_t n,
                                                               const int2 *__restrict__ pairs, int32_t *__restrict__ si,
                                                               int32_t *__restrict__ sj, int32_t *__restrict__ fail_count) {
    // bpr_sample_items_kernel with four positions per thread, in lock step: a sample is a chain of ~12 dependent loads (row pointers, the
    // positive, the binary search of the negative candidate in the sorted row); where the rows do not sit in the L2 (C3 shard: 12.5M
    // feedbacks) four chains per thread take 100 -> 81 us per 4M samples; where they do (S-ml1m) the one-sample kernel is the
    // faster one (37 against 47 us per million: more steps per wave, the redo of the samples whose first candidate is rejected).
    // The search is the same lower bound, one step of all four per round.
    constexpr int B = 4;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t tb = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tb < n; tb += B * nthreads) {
        int32_t u[B], sid[B];
#pragma unroll
        for (int e = 0; e < B; e++) {
            const int64_t t = tb + e * nthreads;
            const int2 pr = t < n ? pairs[t] : make_int2(0, -1);  // (sample id, user)
            u[e] = pr.y;  // u < 0: no user could be drawn (the positions behind the last run, never read)
            sid[e] = pr.x;
        }
        int64_t rbeg[B];
        int32_t cnt[B];
#pragma unroll
        for (int e = 0; e < B; e++) {
            const int32_t uu = u[e] < 0 ? 0 : u[e];
            rbeg[e] = uptr[uu];
            cnt[e] = (int32_t)(uptr[uu + 1] - rbeg[e]);
            if (u[e] < 0 || cnt[e] <= 0) cnt[e] = 0;
        }
        int32_t pi[B], c[B];
#pragma unroll
        for (int e = 0; e < B; e++) {
            Philox g;
            g.init(seed, epoch, (uint64_t)(sample_base + sid[e]));
            // the user draw again: rows drawn before u were empty (u is the first non-empty one), no look-up needed
            for (int k = 0; k < kMaxDraws; k++)
                if (cnt[e] == 0 || g.int31n(U) == u[e]) break;
            const int32_t r = g.int31n(cnt[e] > 0 ? cnt[e] : 1);
            c[e] = g.int31n(I);
            pi[e] = cnt[e] > 0 ? uidx[rbeg[e] + r] : -1;
        }
        int32_t lo[B], len[B];
#pragma unroll
        for (int e = 0; e < B; e++) lo[e] = 0, len[e] = cnt[e];
        while ((len[0] | len[1] | len[2] | len[3]) != 0) {  // (lengths are >= 0)
            int32_t v[B];
#pragma unroll
            for (int e = 0; e < B; e++) v[e] = len[e] > 0 ? usorted[rbeg[e] + lo[e] + (len[e] >> 1)] : 0;
#pragma unroll
            for (int e = 0; e < B; e++) {
                if (len[e] > 0) {
                    const int32_t half = len[e] >> 1;
                    if (v[e] < c[e]) {
                        lo[e] += half + 1;
                        len[e] -= half + 1;
                    } else {
                        len[e] = half;
                    }
                }
            }
        }
        int32_t at[B];
#pragma unroll
        for (int e = 0; e < B; e++) at[e] = lo[e] < cnt[e] ? usorted[rbeg[e] + lo[e]] : -1;
#pragma unroll
        for (int e = 0; e < B; e++) {
            const int64_t t = tb + e * nthreads;
            if (t >= n || u[e] < 0) continue;
            int32_t nj = c[e];
            if (at[e] == c[e])  // the candidate is one of the user's items: the rest of the draws, one at a time
                nj = draw_negative_again(U, I, usorted + rbeg[e], cnt[e], u[e], seed, epoch, (uint64_t)(sample_base + sid[e]));
            if (nj < 0) atomicAdd(fail_count, 1);
            si[t] = nj < 0 ? -1 : pi[e];
            sj[t] = nj;
        }
    }
}

// ---- the preparation by user BINS (round 5) ---------------------------------------------------
// What the three kernels above put on the memory side per sample -- one returning atomic on the user's counter and two 4-byte stores
// to random places of the sorted order, every 64-byte line of which is written in sixteen pieces by workgroups of different XCDs --
// is what the update kernel, which runs beside the preparation of the next chunk, is bound by (its item-row atomics).  The binned form
// sorts in two levels so that no global atomic and no lone store is left per sample:
//   bpr_bin_count_kernel   : a workgroup draws the users of one TILE of samples (as bpr_sample_user_kernel does), stores the keys and
//                            counts them per BIN (a range of 2^shift user ids) in LDS and writes its row of the tile x bin matrix;
//   bpr_bin_offsets_kernel : the matrix's columns -> prefix over the tiles, and the bins' totals;
//   bpr_bin_scatter_kernel : scans the bin totals (every workgroup for itself: < 8192 bins), adds its row of the matrix = its tile's
//                            place in every bin, and writes (sample id, user) there -- runs of a tile's samples of one bin, written
//                            by ONE workgroup, merge in its L2;
//   bpr_bin_sort_kernel    : a workgroup per bin counts the bin's samples per user in LDS, scans, writes the run offsets of its users
//                            (bucket[] as the update kernel reads it) and places (sample id, user) at the sorted positions -- stores
//                            inside the bin's own window;
//   bpr_sample_items_kernel: unchanged.
// The order of the samples inside a run is the order of arrival as before (no order is promised: the runs are multisets).
constexpr int kBinThreads = 512;     // workgroup of the count / scatter kernels
constexpr int kBinBatch = 8;         // samples of a thread whose loads are in flight together (scatter / sort kernels)

template <int NW>
__device__ __forceinline__ int32_t block_exclusive_scan(int32_t v, int32_t *lds /* NW + 1 */, int32_t *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    int32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int32_t t = lds[w];
        if (w < wid) base += t;
        all += t;
    }
    if (total) *total = all;
    __syncthreads();
    return base + x - v;
}

// the user draw of a sample whose FIRST draw met a user without feedback: the stream again from its start (model.go:452-458)
__device__ __forceinline__ int32_t draw_user_again(int32_t U, const int64_t *__restrict__ uptr, uint64_t seed, uint64_t epoch,
                                                uint64_t sample, int32_t *__restrict__ fail_count) {
    Philox g;
    g.init(seed, epoch, sample);
    for (int t = 0; t < kMaxDraws; t++) {
        const int32_t cu = g.int31n(U);
        if (uptr[cu + 1] > uptr[cu]) return cu;
    }
    atomicAdd(fail_count, 1);
    return -1;
}

__global__ __launch_bounds__(kBinThreads) void bpr_bin_count_kernel(int32_t U, const int64_t *__restrict__ uptr, uint64_t seed,
                                                                    uint64_t epoch, int64_t sample_base, int64_t n, int64_t tile,
                                                                    int shift, int nbins, int32_t *__restrict__ key,
                                                                    int32_t *__restrict__ fail_count,
                                                                    int32_t *__restrict__ H) {
    __shared__ int32_t hist[kMaxBins];
    for (int b = threadIdx.x; b < nbins; b += kBinThreads) hist[b] = 0;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * tile, s1 = s0 + tile < n ? s0 + tile : n;
    // four samples per thread at a time: their first user draws' row pointers are in flight together (the draw that finds an
    // empty row -- rare -- goes on alone, same stream, same order of draws)
    constexpr int B = 4;
    for (int64_t sb = s0 + threadIdx.x; sb < s1; sb += B * kBinThreads) {
        int32_t cu[B];
        int64_t r0[B], r1[B];
#pragma unroll
        for (int e = 0; e < B; e++) {
            const int64_t s = sb + (int64_t)e * kBinThreads;
            Philox g;
            g.init(seed, epoch, (uint64_t)(sample_base + (s < s1 ? s : s0)));
            cu[e] = g.int31n(U);
            r0[e] = uptr[cu[e]];
            r1[e] = uptr[cu[e] + 1];
        }
#pragma unroll
        for (int e = 0; e < B; e++) {
            const int64_t s = sb + (int64_t)e * kBinThreads;
            if (s < s1) {
                int32_t u = r1[e] > r0[e] ? cu[e] : -1;
                if (u < 0) u = draw_user_again(U, uptr, seed, epoch, (uint64_t)(sample_base + s), fail_count);
                key[s] = u;
                atomicAdd(&hist[(u < 0 ? U : u) >> shift], 1);
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += kBinThreads) H[(int64_t)blockIdx.x * nbins + b] = hist[b];
}

// H[t][b] = tile t's count of bin b -> the exclusive prefix over the tiles, in place; bin_count[b] = the bin's total.  (The first form
// of the binned preparation reserved a tile's share of a bin with a returning atomic on the bin's cursor: a few hundred tiles on
// one address take ~0.4 us each, one after the other -- 88 us at S-ml1m, the whole gain.  r05_zd_bpr_serial_*.txt)
constexpr int kOffWaves = 16;
__global__ __launch_bounds__(kOffWaves * 64) void bpr_bin_offsets_kernel(int32_t *__restrict__ H, int tiles, int nbins,
                                                                         int32_t *__restrict__ bin_count) {
    __shared__ int32_t seg[kOffWaves][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    const int per = (tiles + kOffWaves - 1) / kOffWaves, t0 = w * per, t1 = t0 + per < tiles ? t0 + per : tiles;
    int32_t sum = 0;
    if (b < nbins)
        for (int t = t0; t < t1; t++) sum += H[(int64_t)t * nbins + b];
    seg[w][lane] = sum;
    __syncthreads();
    int32_t run = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kOffWaves; k++) {
        const int32_t v = seg[k][lane];
        if (k < w) run += v;
        total += v;
    }
    if (b < nbins) {
        for (int t = t0; t < t1; t++) {
            const int32_t v = H[(int64_t)t * nbins + b];
            H[(int64_t)t * nbins + b] = run;
            run += v;
        }
        if (w == 0) bin_count[b] = total;
    }
}

__global__ __launch_bounds__(kBinThreads) void bpr_bin_scatter_kernel(int32_t U, const int32_t *__restrict__ key, int64_t n, int64_t tile,
                                                                      int shift, int nbins, const int32_t *__restrict__ bin_count,
                                                                      const int32_t *__restrict__ H, int32_t *__restrict__ bin_start,
                                                                      int2 *__restrict__ bp) {
    __shared__ int32_t cur[kMaxBins];  // where this tile's next sample of a bin goes
    __shared__ int32_t wsum[kBinThreads / 64 + 1];
    constexpr int PER = kMaxBins / kBinThreads;  // bins per thread of the scan
    {
        // exclusive scan of the chunk's bin counts (every workgroup its own: < 8192 words); a bin's start + the counts of the tiles
        // in front of this one = this tile's place in the bin
        int32_t c[PER], v = 0;
#pragma unroll
        for (int e = 0; e < PER; e++) {
            const int b = threadIdx.x * PER + e;
            c[e] = b < nbins ? bin_count[b] : 0;
            v += c[e];
        }
        int32_t run = block_exclusive_scan<kBinThreads / 64>(v, wsum, nullptr);
#pragma unroll
        for (int e = 0; e < PER; e++) {
            const int b = threadIdx.x * PER + e;
            if (blockIdx.x == 0 && b <= nbins) bin_start[b] = run;  // bin_start[nbins] = n (nbins < kMaxBins)
            if (b < nbins) cur[b] = run + H[(int64_t)blockIdx.x * nbins + b];
            run += c[e];
        }
    }
    __syncthreads();
    // (eight keys per thread in flight: one load per iteration made the loop a chain of memory latencies -- 96 -> us per 4M samples)
    const int64_t s0 = (int64_t)blockIdx.x * tile, s1 = s0 + tile < n ? s0 + tile : n;
    for (int64_t sb = s0 + threadIdx.x; sb < s1; sb += kBinBatch * kBinThreads) {
        int32_t u[kBinBatch];
#pragma unroll
        for (int e = 0; e < kBinBatch; e++) {
            const int64_t s = sb + (int64_t)e * kBinThreads;
            u[e] = s < s1 ? key[s] : 0;
        }
#pragma unroll
        for (int e = 0; e < kBinBatch; e++) {
            const int64_t s = sb + (int64_t)e * kBinThreads;
            if (s < s1) {
                const int32_t pos = atomicAdd(&cur[(u[e] < 0 ? U : u[e]) >> shift], 1);
                bp[pos] = make_int2((int32_t)s, u[e]);  // (sample id, user): one 8-byte store
            }
        }
    }
}

__global__ __launch_bounds__(256) void bpr_bin_sort_kernel(int32_t U, int shift, int nbins, const int32_t *__restrict__ bin_start,
                                                           const int2 *__restrict__ bp, int32_t *__restrict__ bucket,
                                                           int2 *__restrict__ pairs) {
    __shared__ int32_t cnt[1 << kMaxBinShift];
    __shared__ int32_t wsum[5];
    const int ub = 1 << shift;
    for (int b = blockIdx.x; b < nbins; b += gridDim.x) {
        const int32_t ulo = b << shift;  // keys ulo .. ulo + ub - 1 (key U = the samples without a user, sorted last)
        const int32_t b0 = bin_start[b], b1 = bin_start[b + 1];
        for (int k = threadIdx.x; k < ub; k += 256) cnt[k] = 0;
        __syncthreads();
        for (int32_t eb = b0 + threadIdx.x; eb < b1; eb += kBinBatch * 256) {
            int32_t u[kBinBatch];
#pragma unroll
            for (int e = 0; e < kBinBatch; e++) u[e] = eb + e * 256 < b1 ? bp[eb + e * 256].y : 0;
#pragma unroll
            for (int e = 0; e < kBinBatch; e++)
                if (eb + e * 256 < b1) atomicAdd(&cnt[(u[e] < 0 ? U : u[e]) - ulo], 1);
        }
        __syncthreads();
        // exclusive scan of cnt[0 .. ub): eight counters per thread and round
        int32_t carry = 0;
        for (int k0 = 0; k0 < ub; k0 += 2048) {
            int32_t c[8], v = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int k = k0 + threadIdx.x * 8 + e;
                c[e] = k < ub ? cnt[k] : 0;
                v += c[e];
            }
            int32_t total;
            int32_t run = carry + block_exclusive_scan<4>(v, wsum, &total);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int k = k0 + threadIdx.x * 8 + e;
                if (k < ub) {
                    cnt[k] = run;
                    if (ulo + k <= U) bucket[ulo + k] = b0 + run;
                    if (ulo + k == U) bucket[U + 1] = b1;
                }
                run += c[e];
            }
            carry += total;
        }
        __syncthreads();
        for (int32_t eb = b0 + threadIdx.x; eb < b1; eb += kBinBatch * 256) {
            int2 pr[kBinBatch];
#pragma unroll
            for (int e = 0; e < kBinBatch; e++) pr[e] = eb + e * 256 < b1 ? bp[eb + e * 256] : make_int2(0, 0);
#pragma unroll
            for (int e = 0; e < kBinBatch; e++) {
                if (eb + e * 256 < b1) {
                    const int32_t p = b0 + atomicAdd(&cnt[(pr[e].y < 0 ? U : pr[e].y) - ulo], 1);
                    pairs[p] = pr[e];
                }
            }
        }
        __syncthreads();
    }
}

// ---- memory access flavours ------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float load_row(const float *p) {
    if (MODE == MODE_EXACT)
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
                                            float nreg, float lr, bool fused, bool same_item) {
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
        apply<MODE>(qi + e, a, t1, lr, fused);
        apply<MODE>(qj + e, b, t2, lr, fused);
        apply<MODE>(pu + e, p, t3, lr, fused);
    }
}

// Hot-row replicas (GORSE_BPR_HOGWILD_ATOMIC only).  Device-scope atomics and L1-bypassing loads are
// served memory-side, line by line, in order: a row that a popularity-skewed epoch keeps hitting builds a
// deep atomic queue, and every gather of that row -- and the retirement of every wave that updated it --
// waits in it (measured: positive-item atomics redirected away from the rows being read = 2.0x, spread
// over 8 rows each = 2.7x on S-ml1m; profiles/r01_b_probe_rep.txt).  So the positive-item update of a HOT
// item (share of the training feedback >= 1/8192, chosen at create time) -- and, in the user-run schedule, the update of a
// hot item drawn as the NEGATIVE -- lands in one of kHotReplicas
// private rows picked by the group id, and FOLDER workgroups of the same launch keep draining the
// replicas into the real rows with one combined atomic per element and pass.  Q stays the only source of
// truth for every reader; a hot row's update becomes visible one folder pass (tens of microseconds) late,
// the same order as the latency of the atomics themselves.  A fold kernel after the launch leaves every
// replica zero, so nothing outside bpr.hip ever sees them.
constexpr int kHotReplicas = GORSE_HOT_REPLICAS;
constexpr int kFolderBlocks = 32;

struct HotRows {
    const int32_t *slot;   // I entries: replica slot of an item, or -1
    const int32_t *items;  // n_hot entries: item of a slot
    float *rep;            // n_hot x kHotReplicas x d, all zero outside an update launch
    int32_t *done;         // worker workgroups that have finished, counted in kDoneStripes words (zeroed before the launch)
    int n_hot;
    int64_t stride_s, stride_r;  // replica r of slot s starts at rep + s * stride_s + r * stride_r
#ifdef GORSE_PROBE
    float *warm_scratch = nullptr;  // timing probe (variant bit 23): the atomics of the items WITHOUT replicas land here instead of on Q (results garbage)
#endif
};

// The arrival counter of the worker workgroups, in kDoneStripes words 256 bytes apart: one 64-byte line serves 88 M atomics/s
// (scripts/probe_atomics4.hip), and the per-sample kernel's 4096 workgroups, which all finish within a few microseconds of each other,
// queued 46 us on a single word at the end of every launch (S-ml100k: a third of the kernel).
constexpr int kDoneStripes = GORSE_HOT_DONE_STRIPES, kDoneStride = GORSE_HOT_DONE_STRIDE;
__device__ __forceinline__ void worker_done(const HotRows &hot) {
    __hip_atomic_fetch_add(hot.done + (blockIdx.x & (kDoneStripes - 1)) * kDoneStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool workers_done(const HotRows &hot, int workers) {  // (called by whole waves)
    const int lane = threadIdx.x & 63;
    int v = lane < kDoneStripes ? __hip_atomic_load(hot.done + lane * kDoneStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v >= workers;
}

// one folder pass: (slot, element) pairs strided over the folder threads
__device__ __forceinline__ void fold_pass(const HotRows &hot, float *Q, int d, int64_t tid, int64_t nthreads) {
    const int64_t work = (int64_t)hot.n_hot * d;
    for (int64_t w = tid; w < work; w += nthreads) {
        const int64_t slot = w / d;
        const int e = (int)(w - slot * d);
        float *r0 = hot.rep + slot * hot.stride_s + e;
        float v[kHotReplicas];
#pragma unroll
        for (int r = 0; r < kHotReplicas; r++)
            v[r] = __hip_atomic_exchange(r0 + r * hot.stride_r, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < kHotReplicas; r++) sum += v[r];
        if (sum != 0.0f)
            __hip_atomic_fetch_add(Q + (int64_t)hot.items[slot] * d + e, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NC, int MODE>
__global__ __launch_bounds__(kBlock) void bpr_update_kernel(float *P, float *Q, const int32_t *__restrict__ us,
                                                            const int32_t *__restrict__ is,
                                                            const int32_t *__restrict__ js,
                                                            const int32_t *__restrict__ order, int64_t begin,
                                                            int64_t end, int d, float lr, float reg, int exp_mode,
                                                            double *loss, HotRows hot, int folders) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (MODE == MODE_ATOMIC && (int)blockIdx.x < folders) {
        // folder workgroups: drain the replicas until every worker workgroup has finished.  Nobody waits
        // for a folder, and the pass count is bounded, so an early exit only leaves more to the fold kernel.
        const int workers = (int)gridDim.x - folders;
        const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)folders * blockDim.x;
        for (int pass = 0; pass < (1 << 16); pass++) {
            fold_pass(hot, Q, d, tid, nthreads);
            if (workers_done(hot, workers)) break;
            __builtin_amdgcn_s_sleep(8);
        }
        return;
    }
    const int lane = threadIdx.x & (kGroup - 1);
    const int gib = threadIdx.x / kGroup;
    const int64_t group = (int64_t)((int)blockIdx.x - folders) * kGroupsPerBlock + gib;
    const int64_t ngroups = (int64_t)((int)gridDim.x - folders) * kGroupsPerBlock;
    const VecShape vs(d);
    const float nreg = -reg;
    double my_loss = 0.0;
    for (int64_t s = begin + group; s < end; s += ngroups) {
        const int64_t t = order ? (int64_t)order[s] : s;
        const int u = us[t], i = is[t], j = js[t];
        if ((u | i | j) < 0) continue;
        float *pu = P + (int64_t)u * d, *qi = Q + (int64_t)i * d, *qj = Q + (int64_t)j * d;
        float *qiw = qi;  // where the positive item's update lands
        if (MODE == MODE_ATOMIC && hot.n_hot > 0) {
            const int slot = hot.slot[i];
            if (slot >= 0) qiw = hot.rep + (int64_t)slot * hot.stride_s + (int64_t)(group & (kHotReplicas - 1)) * hot.stride_r;
        }
        if constexpr (NC > 0) {
            float p[NC], a[NC], b[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                p[c] = load_row<MODE>(pu + 16 * c + lane);
                a[c] = load_row<MODE>(qi + 16 * c + lane);
                b[c] = load_row<MODE>(qj + 16 * c + lane);
            }
            const float diff = dot512_regs<NC>(p, a) - dot512_regs<NC>(p, b);
            const float ex = bpr_exp(-diff, exp_mode);
            const float grad = ex / (1.0f + ex);
            if (loss && lane == 0) my_loss += (double)log1pf(ex);
#pragma unroll
            for (int c = 0; c < NC; c++)
                update_elem<MODE>(pu, qiw, qj, 16 * c + lane, p[c], a[c], b[c], grad, nreg, lr, true, i == j);
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
                update_elem<MODE>(pu, qiw, qj, e, sp[e], sa[e], sb[e], grad, nreg, lr, !vs.unfused(e), i == j);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (loss && lane == 0 && my_loss != 0.0) atomicAdd(loss, my_loss);
    if (MODE == MODE_ATOMIC && folders > 0) {
        __syncthreads();
        if (threadIdx.x == 0) worker_done(hot);
    }
}

// after an update launch: Q[item] += sum of its replicas, replicas <- 0 (nothing else runs on the stream)
__global__ __launch_bounds__(256) void bpr_fold_kernel(HotRows hot, float *Q, int d) {
    const int64_t work = (int64_t)hot.n_hot * d;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < work; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = w / d;
        const int e = (int)(w - slot * d);
        float *r0 = hot.rep + slot * hot.stride_s + e;
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < kHotReplicas; r++) {
            sum += r0[r * hot.stride_r];
            r0[r * hot.stride_r] = 0.0f;
        }
        if (sum != 0.0f) Q[(int64_t)hot.items[slot] * d + e] += sum;
    }
    // every worker of the update launch has finished (stream order): the arrival counter goes back to zero for the next launch
    if (blockIdx.x == 0 && threadIdx.x < kDoneStripes) hot.done[threadIdx.x * kDoneStride] = 0;
}

// ---- counting sort by user ---------------------------------------------------------------------
constexpr int kScanTile = 2048;  // elements per workgroup of the scan (256 threads x 8)

// first pass of the counting sort for callers whose sampler did not rank (gorse_bpr_apply_triplets, the stable-rank test hook):
// rank[s] = arrival rank of sample s among the samples of its key; keys < 0 (skipped samples) count under key K
__global__ __launch_bounds__(256) void bpr_rank_kernel(const int32_t *__restrict__ key, int64_t n, int32_t K,
                                                       int32_t *__restrict__ bucket, int32_t *__restrict__ rank) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x)
        rank[s] = atomicAdd(&bucket[key[s] < 0 ? K : key[s]], 1);
}

// exclusive scan of data[0..m) in place, three launches: tile sums, scan of the tile sums, tile rescans
__device__ __forceinline__ int32_t block_exclusive_scan_256(int32_t v, int32_t *lds, int32_t *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    int32_t base = 0;
    for (int w = 0; w < wid; w++) base += lds[w];
    if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + x - v;
}

__global__ __launch_bounds__(256) void scan_tile_sums_kernel(const int32_t *__restrict__ data, int64_t m,
                                                             int32_t *__restrict__ sums) {
    __shared__ int32_t lds[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 8;
    int32_t v = 0;
#pragma unroll
    for (int e = 0; e < 8; e++)
        if (base + e < m) v += data[base + e];
    int32_t total;
    (void)block_exclusive_scan_256(v, lds, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scan_sums_kernel(int32_t *__restrict__ sums, int64_t nt) {
    __shared__ int32_t lds[4];
    int32_t carry = 0;
    for (int64_t t0 = 0; t0 < nt; t0 += 256) {
        const int64_t t = t0 + threadIdx.x;
        const int32_t v = t < nt ? sums[t] : 0;
        int32_t total;
        const int32_t ex = block_exclusive_scan_256(v, lds, &total);
        if (t < nt) sums[t] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(int32_t *__restrict__ data, int64_t m,
                                                         const int32_t *__restrict__ sums) {
    __shared__ int32_t lds[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 8;
    int32_t x[8], v = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        x[e] = base + e < m ? data[base + e] : 0;
        v += x[e];
    }
    int32_t run = sums[blockIdx.x] + block_exclusive_scan_256(v, lds, nullptr);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if (base + e < m) data[base + e] = run;
        run += x[e];
    }
}


// scatter by an arbitrary key array (one window): keys < 0 sort behind every real key
__global__ __launch_bounds__(256) void bpr_scatter_by_kernel(const int32_t *__restrict__ key, const int32_t *__restrict__ us,
                                                             const int32_t *__restrict__ is, const int32_t *__restrict__ js,
                                                             int64_t n, int32_t K, const int32_t *__restrict__ bucket,
                                                             const int32_t *__restrict__ rank, int32_t *__restrict__ su,
                                                             int32_t *__restrict__ si, int32_t *__restrict__ sj) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        const int32_t kk = key[s] < 0 ? K : key[s];
        const int64_t pos = (int64_t)bucket[kk] + rank[s];
        su[pos] = us[s];
        si[pos] = is[s];
        sj[pos] = js[s];
    }
}

// ---- user-run schedule ---------------------------------------------------------------------------
// The chunk's triplets are counting-sorted by USER (the order in which a Hogwild epoch applies its samples is free:
// parallel.go:44-68) and one 16-lane group walks ALL samples of one user: p_u is loaded once, updated in registers
// sample after sample -- exact sequential SGD on the user side, no lost or delayed update -- and stored once, so a
// third of the fp32 atomics (the unit this kernel is bound by: ~1 dword/clk per L2 channel) and a third of the
// gathers disappear.  q_i / q_j are gathered one sample ahead of the arithmetic and updated with atomics exactly as
// in bpr_update_kernel (hot positive items through the replicas).  Users are drawn uniformly (model.go:452-458), so
// the runs are Poisson(N / U)-sized: balanced without any work splitting.
// Item classes (hot.slot): >= 0 = replica slot of a HOT item, kWarm = fp32 atomics straight onto the row, kCold = an item whose
// row is touched so rarely (expected touches per `cold window` samples < 1, gorse_mf_create) that the reference's own unlocked
// load / fma / store (model.go:473-488 under parallel.go:44-81) loses next to nothing: its update is ONE write-through store of
// fma(t, lr, row) instead of d atomic dwords.  ST selects which side may take that route: ST_NEG the negative, ST_POS the
// positive, ST_LIVE re-reads the row just before the store instead of adding to the snapshot gathered two samples earlier
// (diagnostic: a shorter window in which another group's update can be overwritten, one more gather per row).
constexpr int kCold = -2;  // (-1 = "warm": fp32 atomics straight onto the row)
constexpr int ST_NEG = 1, ST_POS = 2, ST_LIVE = 4;

// D8: nFactors = 8 (the width of model_test.go:35-48): lanes 0..7 of the group own the eight elements -- the unfused 8-lane tail of
// the AVX512 kernels (floats_avx512.c:350-358, VecShape::unfused) -- and lanes 8..15 mirror them (the reduction needs the products
// replicated there); only lanes 0..7 write.
// SEG (round 5, nFactors <= 32; `make probe-lib` builds only -- a measured dead end, kept so that the measurement can be repeated): a
// user's run is cut into `segs` consecutive segments, each taken by its own group: every segment starts from the row as it stood
// before the launch and adds its own change to it with atomics at the end.  With 6040 users (S-ml1m) a group per user leaves the chip
// at 1.5 waves per SIMD walking ~165 dependent samples each; the segments are the reference's own race -- two of its workers that drew
// the same user both update p_u from what they read (model.go:449-488) -- made regular.  Result (profiles/r05_k_probe_gpu_probe_bpr_
// segments.txt: S-ml1m, 30 epochs, three seeds): 3 segments take 5 % off the epoch (0.456 -> 0.432 ms at nFactors 8, 0.350 -> 0.332 at
// 16, nothing at 32) for 0.001-0.003 of NDCG@10 -- the epoch is not the per-user chain but the chunk's preparation running beside
// it -- and with 4 / 8 segments a fit DIVERGED (NaN) in two of 18 runs: a heavy user's segments each apply the whole run's
// regularisation shrink to the same starting row, and the summed changes overshoot (n_u lr reg > 1 for 2000 feedbacks).
#ifndef GORSE_BPR_D8_PAIRS
#define GORSE_BPR_D8_PAIRS 1  // A/B: 0 = nFactors 8 with lanes 8..15 of a group mirroring lanes 0..7 (rounds 4-5)
#endif
template <int NC, int ST, bool D8 = false, int G = 2, int IA = 3, bool SEG = false>
__global__ __launch_bounds__(kBlock) void bpr_update_user_kernel(float *P, float *Q, const int32_t *__restrict__ si,
                                                                 const int32_t *__restrict__ sj,
                                                                 const int32_t *__restrict__ off, int32_t U, int d,
                                                                 float lr, float reg, int exp_mode, double *loss,
                                                                 HotRows hot, int folders, int neg_replicas, int segs = 1) {
    if ((int)blockIdx.x < folders) {
        const int workers = (int)gridDim.x - folders;
        const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)folders * blockDim.x;
        for (int pass = 0; pass < (1 << 16); pass++) {
            fold_pass(hot, Q, d, tid, nthreads);
            if (workers_done(hot, workers)) break;
            __builtin_amdgcn_s_sleep(8);
        }
        return;
    }
    // with the store route open for negatives their rows are gathered ONE sample ahead instead of two: what a store can
    // overwrite is what other groups added between the gather and the store, and that window halves
    constexpr bool NEG1 = (ST & ST_NEG) != 0;
    const int glane = threadIdx.x & (kGroup - 1);
    const int lane = D8 ? (glane & 7) : glane;   // element owned inside a 16-float chunk
    // D8 (round 6): the two halves of a 16-lane group take a user run EACH (eight lanes own a row of eight: until round 5 lanes 8..15
    // mirrored lanes 0..7 and only wrote nothing) -- eight runs per wave instead of four
    constexpr int SUBS = D8 && GORSE_BPR_D8_PAIRS ? 2 : 1;
    const bool writer = !D8 || SUBS == 2 || glane < 8;
    auto mad = [](float x, float y, float z) { return D8 ? x * y + z : fmaf(x, y, z); };
    auto tree8 = [](float v) { return SUBS == 2 ? group_tree8_halves(v) : group_tree8(v); };
    const int gib = threadIdx.x / kGroup;
    const int64_t group = ((int64_t)((int)blockIdx.x - folders) * kGroupsPerBlock + gib) * SUBS + (SUBS == 2 ? glane >> 3 : 0);
    const int64_t ngroups = (int64_t)((int)gridDim.x - folders) * kGroupsPerBlock * SUBS;
    const float nreg = -reg;
    double my_loss = 0.0;
    // which class look-ups this launch needs (every path issues the same loads from a valid address: a look-up nobody needs
    // reads a word that is in cache anyway)
    const bool look_i = hot.n_hot > 0 || (ST & ST_POS);
    const bool look_j = (hot.n_hot > 0 && neg_replicas) || (ST & ST_NEG);
    const int32_t *slot_of = (look_i || look_j) ? hot.slot : si;
    const int64_t rep_r = (int64_t)(group & (kHotReplicas - 1)) * hot.stride_r;
    // the items of this group's last two samples: a row one of them wrote is NOT in the snapshot of the current sample (gathered
    // two samples ago), so its update goes through an atomic whatever its class -- a store would overwrite the group's own work
    int im1 = -1, jm1 = -1, im2 = -1, jm2 = -1;
    const int64_t runs = SEG ? (int64_t)U * segs : (int64_t)U;
    for (int64_t run = group; run < runs; run += ngroups) {
        const int64_t u = SEG ? run / segs : run;
        int beg = off[u], end = off[u + 1];
        if constexpr (SEG) {  // segment run - u * segs of the user's run
            const int64_t len = end - beg, sg = run - u * segs;
            end = beg + (int)(len * (sg + 1) / segs);
            beg = beg + (int)(len * sg / segs);
        }
        if (beg >= end) continue;
        float *pu = P + u * d;
        // item rows are gathered G samples ahead of the arithmetic (ra[0] / rb[0]: this sample, ra[k]: sample s + k, the row of sample
        // s + G in flight), their indices IA ahead and the class of an item G ahead; positions past the run's end re-read the
        // last sample (result unused).  Nothing is used in the iteration that loads it: the counter the waits go by also counts the
        // atomics and returns in order, so a wait for a load issued after them is a wait for their acknowledgement from the memory
        // side.  What a load waits behind is therefore the atomics of min(G, IA - G) iterations ago: with few groups per SIMD
        // (C2: 6040 groups on 1024 SIMDs) that distance IS the time of an iteration.
        static_assert(IA > G && G >= 2, "indices ahead of the rows they address");
        static_assert(ST == 0 || G == 2, "the own-history rule of the store route looks two samples back");
        constexpr int GB = NEG1 ? 1 : G;  // the negative's gather distance
        float p[NC], ra[G][NC], rb[GB][NC];
        const int last = end - 1;
        auto at = [&](int s) { return s <= last ? s : last; };
        auto cl = [](int x) { return x < 0 ? 0 : x; };  // a skipped sample (i = j = -1) gathers row 0, its results are not used
        auto idx_i = [&](int i_, int j_) { return look_i ? cl(i_) : (look_j ? cl(j_) : beg); };
        auto idx_j = [&](int i_, int j_) { return look_j ? cl(j_) : (look_i ? cl(i_) : beg); };
        int ii[IA], jj[IA], sli[G], slj[G];
#pragma unroll
        for (int k = 0; k < IA; k++) ii[k] = si[at(beg + k)], jj[k] = sj[at(beg + k)];
#pragma unroll
        for (int k = 0; k < G; k++) sli[k] = slot_of[idx_i(ii[k], jj[k])], slj[k] = slot_of[idx_j(ii[k], jj[k])];
        float p_in[SEG ? NC : 1];  // SEG: the row as this segment found it
#pragma unroll
        for (int c = 0; c < NC; c++) {
            p[c] = load_row<MODE_ATOMIC>(pu + 16 * c + lane);
            if (SEG) p_in[SEG ? c : 0] = p[c];
#pragma unroll
            for (int k = 0; k < G; k++) ra[k][c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(ii[k]) * d + 16 * c + lane);
#pragma unroll
            for (int k = 0; k < GB; k++) rb[k][c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(jj[k]) * d + 16 * c + lane);
        }
        for (int s = beg; s < end; s++) {
            const int in_ = si[at(s + IA)], jn_ = sj[at(s + IA)];
            const int sli_n = slot_of[idx_i(ii[G], jj[G])];
            const int slj_n = slot_of[idx_j(ii[G], jj[G])];
            const int i = ii[0], j = jj[0], slot = sli[0], slotj = slj[0];
            float(&a)[NC] = ra[0];
            float(&b)[NC] = rb[0];
            float al[NC], bl[NC], ran[NC], rbn[NC];
            if constexpr ((ST & ST_LIVE) != 0) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    al[c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(i) * d + 16 * c + lane);
                    bl[c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(j) * d + 16 * c + lane);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                ran[c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(ii[G]) * d + 16 * c + lane);
                rbn[c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(jj[GB]) * d + 16 * c + lane);
            }
            const bool valid = j >= 0;  // j < 0: the sampler found no negative for this sample (bpr_sample_items_kernel)
            float *qi = Q + (int64_t)cl(i) * d, *qj = Q + (int64_t)cl(j) * d;
            if (hot.n_hot > 0 && slot >= 0) qi = hot.rep + (int64_t)slot * hot.stride_s + rep_r;
            if (hot.n_hot > 0 && neg_replicas && slotj >= 0) qj = hot.rep + (int64_t)slotj * hot.stride_s + rep_r;
#ifdef GORSE_PROBE
            if (hot.warm_scratch && !(hot.n_hot > 0 && slot >= 0)) qi = hot.warm_scratch + (int64_t)cl(i) * d;
            if (hot.warm_scratch && !(hot.n_hot > 0 && neg_replicas && slotj >= 0)) qj = hot.warm_scratch + (int64_t)cl(j) * d;
#endif
            bool st_i = false, st_j = false;
            if constexpr ((ST & (ST_POS | ST_NEG)) != 0) {
                const bool own_i = i == j || i == im1 || i == jm1 || i == im2 || i == jm2;
                const bool own_j = i == j || j == im1 || j == jm1 || j == im2 || j == jm2;
                st_i = (ST & ST_POS) && slot == kCold && !own_i;
                st_j = (ST & ST_NEG) && slotj == kCold && !own_j;
            }
            const float diff = D8 ? tree8(p[0] * a[0]) - tree8(p[0] * b[0]) : dot512_regs<NC>(p, a) - dot512_regs<NC>(p, b);
            const float ex = bpr_exp(-diff, exp_mode);
            const float grad = ex / (1.0f + ex);
            if (loss && (SUBS == 2 ? (glane & 7) == 0 : glane == 0) && valid) my_loss += (double)log1pf(ex);
            float t1[NC], t2[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                t1[c] = mad(a[c], nreg, p[c] * grad);
                t2[c] = mad(b[c], nreg, p[c] * (-grad));
                const float t3 = mad(p[c], nreg, (a[c] - b[c]) * grad);
                p[c] = valid ? mad(t3, lr, p[c]) : p[c];
            }
            if (!valid || !writer) {
                // nothing to write
            } else if (st_i) {
#pragma unroll
                for (int c = 0; c < NC; c++)
                    __hip_atomic_store(qi + 16 * c + lane, mad(t1[c], lr, (ST & ST_LIVE) ? al[c] : a[c]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            } else {
#pragma unroll
                for (int c = 0; c < NC; c++)
                    __hip_atomic_fetch_add(qi + 16 * c + lane, t1[c] * lr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!valid || !writer) {
            } else if (st_j) {
#pragma unroll
                for (int c = 0; c < NC; c++)
                    __hip_atomic_store(qj + 16 * c + lane, mad(t2[c], lr, (ST & ST_LIVE) ? bl[c] : b[c]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            } else {
#pragma unroll
                for (int c = 0; c < NC; c++)
                    __hip_atomic_fetch_add(qj + 16 * c + lane, t2[c] * lr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
#pragma unroll
                for (int k = 0; k + 1 < G; k++) ra[k][c] = ra[k + 1][c];
                ra[G - 1][c] = ran[c];
#pragma unroll
                for (int k = 0; k + 1 < GB; k++) rb[k][c] = rb[k + 1][c];
                rb[GB - 1][c] = rbn[c];
            }
            im2 = im1, jm2 = jm1, im1 = i, jm1 = j;
#pragma unroll
            for (int k = 0; k + 1 < IA; k++) ii[k] = ii[k + 1], jj[k] = jj[k + 1];
            ii[IA - 1] = in_, jj[IA - 1] = jn_;
#pragma unroll
            for (int k = 0; k + 1 < G; k++) sli[k] = sli[k + 1], slj[k] = slj[k + 1];
            sli[G - 1] = sli_n, slj[G - 1] = slj_n;
        }
        if (writer) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (SEG)  // one of `segs` writers of this row: its own change, added
                    __hip_atomic_fetch_add(pu + 16 * c + lane, p[c] - p_in[SEG ? c : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else  // the only writer of this row in the launch
                    __hip_atomic_store(pu + 16 * c + lane, p[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (loss && (SUBS == 2 ? (glane & 7) == 0 : glane == 0) && my_loss != 0.0) atomicAdd(loss, my_loss);
    if (folders > 0) {
        __syncthreads();
        if (threadIdx.x == 0) worker_done(hot);
    }
}

#ifdef GORSE_PROBE
// ---- PROBE BUILD ONLY (make probe-lib; scripts/gpu_probe_bpr_depth.py, gpu_probe_bpr_c2_limits.py): the user-run kernel without a
// register rotation (atomics only).  A measured dead end, kept so that the measurement can be repeated:
// bpr_update_user_kernel keeps its gathered rows in a, a1, a2 and moves them down at the end of every iteration: the move of a2 is
// a USE of the row loaded in that very iteration, the wait in front of it counts (in order) everything issued since -- and because
// the atomics sit behind a lane-dependent branch the compiler has to assume they were not issued: `s_waitcnt vmcnt(0)` at the end
// of every sample.  Here the ring is addressed by a compile-time step index (the loop is unrolled R times, no value ever moves), the
// atomics are issued unconditionally (a sample past the run's end or without a negative adds 0.0f), the smallest wait in the loop
// is vmcnt(18): no atomic of the last sample is ever waited for.  Result (profiles/r04_s_probe_bpr_depth.txt, _c2_limits.txt):
// C2 0.615-0.629 ms against 0.619-0.629, d = 16 0.320 against 0.336, d = 8 0.415 against 0.436, C3 shard 11.5-12.4 against 11.1 --
// the waits were never what the kernel is bound by.  What is (profiles/r04_s_probe_atomics3.txt): the atomic unit itself, which
// sustains 318 G dwords/s on a 3704-row table when nothing reads it and 243 G/s when the same rows are also LOADED (as every sample
// must); on a 30K-row table 302-317 G/s either way.
template <int N, class F, int... K>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, K...>) {
    (f(std::integral_constant<int, K>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

template <int NC, bool D8, int R, int D>
__global__ __launch_bounds__(kBlock) void bpr_update_user_ring_kernel(float *P, float *Q, const int32_t *__restrict__ si,
                                                                      const int32_t *__restrict__ sj,
                                                                      const int32_t *__restrict__ off, int32_t U, int d, float lr,
                                                                      float reg, int exp_mode, double *loss, HotRows hot,
                                                                      int folders, int neg_replicas) {
    if ((int)blockIdx.x < folders) {
        const int workers = (int)gridDim.x - folders;
        const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)folders * blockDim.x;
        for (int pass = 0; pass < (1 << 16); pass++) {
            fold_pass(hot, Q, d, tid, nthreads);
            if (workers_done(hot, workers)) break;
            __builtin_amdgcn_s_sleep(8);
        }
        return;
    }
    static_assert(D >= 1 && R - D >= 2, "ids D steps ahead of the rows they address, rows at least two steps ahead of their use");
    constexpr int G = R - D;  // rows and classes are gathered G samples ahead, ids R ahead
    const int glane = threadIdx.x & (kGroup - 1);
    const int lane = D8 ? (glane & 7) : glane;
    auto mad = [](float x, float y, float z) { return D8 ? x * y + z : fmaf(x, y, z); };
    // gpw groups of a wave work (4 = all; fewer = more waves for the same runs, the other lanes idle), the launch may use smaller
    // workgroups than kBlock
    const int gpw = (neg_replicas >> 4) & 7, wave = threadIdx.x >> 6, giw = (threadIdx.x & 63) / kGroup;
    const int gpb = ((int)blockDim.x >> 6) * gpw;
    int64_t group = (int64_t)((int)blockIdx.x - folders) * gpb + wave * gpw + giw;
    const int64_t ngroups = (int64_t)((int)gridDim.x - folders) * gpb;
    if (giw >= gpw) group = U;  // an idle group: no run
    const float nreg = -reg;
    double my_loss = 0.0;
    const bool look_i = hot.n_hot > 0;
    const bool look_j = hot.n_hot > 0 && (neg_replicas & 1);
#ifdef GORSE_PROBE
    const bool no_atomics = (neg_replicas & 2) != 0;  // probe: what the kernel costs without its item updates
#else
    constexpr bool no_atomics = false;
#endif
    const int32_t *slot_of = look_i ? hot.slot : si;
    const int64_t rep_r = (int64_t)(group & (kHotReplicas - 1)) * hot.stride_r;
    for (int64_t u = group; u < U; u += ngroups) {
        const int beg = off[u], end = off[u + 1];
        if (beg >= end) continue;
        float *pu = P + u * d;
        const int last = end - 1;
        auto at = [&](int s) { return s <= last ? s : last; };
        auto cl = [](int x) { return x < 0 ? 0 : x; };
        auto idx_i = [&](int i_) { return look_i ? cl(i_) : beg; };
        auto idx_j = [&](int i_, int j_) { return look_j ? cl(j_) : idx_i(i_); };
        // ring slot K holds sample beg + m with m % R == K: its item ids, the items' classes, their rows
        float p[NC], ra[R][NC], rb[R][NC];
        int ii[R], jj[R], sli[R], slj[R];
#pragma unroll
        for (int k = 0; k < R; k++) ii[k] = si[at(beg + k)], jj[k] = sj[at(beg + k)];
#pragma unroll
        for (int k = 0; k < G; k++) sli[k] = slot_of[idx_i(ii[k])], slj[k] = slot_of[idx_j(ii[k], jj[k])];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            p[c] = load_row<MODE_ATOMIC>(pu + 16 * c + lane);
#pragma unroll
            for (int k = 0; k < G; k++) {
                ra[k][c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(ii[k]) * d + 16 * c + lane);
                rb[k][c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(jj[k]) * d + 16 * c + lane);
            }
        }
        // nothing pending on entry: what the loop's waits count is then what the loop itself issued (a pending load of the
        // prologue would be "a few instructions old" at the head of every iteration as far as the compiler can tell)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        for (int s0 = beg; s0 < end; s0 += R) {
            static_for<R>([&](auto KC) {
                constexpr int K = decltype(KC)::value, KP = (K + G) % R;
                const int s = s0 + K;
                const int i = ii[K], j = jj[K], slot = sli[K], slotj = slj[K];
                const bool valid = j >= 0 && s < end;  // j < 0: the sampler found no negative (bpr_sample_items_kernel)
                // sample s + R's ids into the slot this sample leaves; class and rows of sample s + G (its ids were loaded D steps ago)
                ii[K] = si[at(s + R)], jj[K] = sj[at(s + R)];
                sli[KP] = slot_of[idx_i(ii[KP])], slj[KP] = slot_of[idx_j(ii[KP], jj[KP])];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    ra[KP][c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(ii[KP]) * d + 16 * c + lane);
                    rb[KP][c] = load_row<MODE_ATOMIC>(Q + (int64_t)cl(jj[KP]) * d + 16 * c + lane);
                }
                float(&a)[NC] = ra[K];
                float(&b)[NC] = rb[K];
                float *qi = Q + (int64_t)cl(i) * d, *qj = Q + (int64_t)cl(j) * d;
                if (hot.n_hot > 0 && slot >= 0) qi = hot.rep + (int64_t)slot * hot.stride_s + rep_r;
                if (look_j && slotj >= 0) qj = hot.rep + (int64_t)slotj * hot.stride_s + rep_r;
                const float diff =
                    D8 ? group_tree8(p[0] * a[0]) - group_tree8(p[0] * b[0]) : dot512_regs<NC>(p, a) - dot512_regs<NC>(p, b);
                const float ex = bpr_exp(-diff, exp_mode);
                const float grad = ex / (1.0f + ex);
                if (loss && glane == 0 && valid) my_loss += (double)log1pf(ex);
                const float step = valid ? lr : 0.0f;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const float t1 = mad(a[c], nreg, p[c] * grad);
                    const float t2 = mad(b[c], nreg, p[c] * (-grad));
                    const float t3 = mad(p[c], nreg, (a[c] - b[c]) * grad);
                    p[c] = valid ? mad(t3, lr, p[c]) : p[c];
                    if (no_atomics) continue;
                    if constexpr (D8) {  // lanes 0..7 carry the positive's row, their mirrors 8..15 the negative's: one instruction
                        __hip_atomic_fetch_add((glane < 8 ? qi : qj) + lane, (glane < 8 ? t1 : t2) * step, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        __hip_atomic_fetch_add(qi + 16 * c + lane, t1 * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(qj + 16 * c + lane, t2 * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            });
        }
        if (!D8 || glane < 8) {
#pragma unroll
            for (int c = 0; c < NC; c++)  // the only writer of this row in the launch
                __hip_atomic_store(pu + 16 * c + lane, p[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (loss && glane == 0 && my_loss != 0.0) atomicAdd(loss, my_loss);
    if (folders > 0) {
        __syncthreads();
        if (threadIdx.x == 0) worker_done(hot);
    }
}
#endif  // GORSE_PROBE

// the whole scan by ONE workgroup, tile after tile (a few thousand counters: two launches and two dependencies fewer)
__global__ __launch_bounds__(256) void scan_small_kernel(int32_t *__restrict__ data, int64_t m) {
    __shared__ int32_t lds[4];
    int32_t carry = 0;
    for (int64_t t0 = 0; t0 < m; t0 += kScanTile) {
        const int64_t base = t0 + (int64_t)threadIdx.x * 8;
        int32_t x[8], v = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            x[e] = base + e < m ? data[base + e] : 0;
            v += x[e];
        }
        int32_t total;
        int32_t run = carry + block_exclusive_scan_256(v, lds, &total);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (base + e < m) data[base + e] = run;
            run += x[e];
        }
        carry += total;
    }
}

int32_t exclusive_scan_i32(int32_t *data, int64_t m, int32_t *tmp, hipStream_t st) {
    const int64_t nt = ceil_div(m, kScanTile);
    if (nt <= 16) {
        scan_small_kernel<<<dim3(1), dim3(256), 0, st>>>(data, m);
        GORSE_HIP_CHECK(hipGetLastError());
        return GORSE_OK;
    }
    scan_tile_sums_kernel<<<dim3((unsigned)nt), dim3(256), 0, st>>>(data, m, tmp);
    scan_sums_kernel<<<dim3(1), dim3(256), 0, st>>>(tmp, nt);
    scan_apply_kernel<<<dim3((unsigned)nt), dim3(256), 0, st>>>(data, m, tmp);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}


// counting sort of the chunk by user: `sorted` receives su | si | sj, bucket[0..U] the run offsets
// ranked: the sampler already counted the runs into `bucket` and wrote every sample's rank (launch_sampler with a bucket)
int32_t launch_user_sort(gorse_mf *h, const int32_t *trip, int32_t *sorted, int32_t *bucket, int32_t *rank, int64_t n, size_t cap,
                         hipStream_t st, bool ranked) {
    const int64_t m = h->U + 2;  // one counter per user + one for skipped samples (sorted last) + the end offset
    const int64_t blocks = std::min<int64_t>(ceil_div(n, 256), 256 * 8);
    if (!ranked) {
        GORSE_HIP_CHECK(hipMemsetAsync(bucket, 0, (size_t)m * sizeof(int32_t), st));
        // key = user, skipped samples (u < 0) get key U
        // test hook (variant bit 29): one thread ranks the samples in stream order -> every run keeps the stream's order
        if (g_variant & (1 << 29))
            bpr_rank_kernel<<<dim3(1), dim3(1), 0, st>>>(trip, n, (int32_t)h->U, bucket, rank);
        else
            bpr_rank_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(trip, n, (int32_t)h->U, bucket, rank);
    }
    GORSE_TRY(exclusive_scan_i32(bucket, m, h->scan_tmp2.p, st));
    bpr_scatter_by_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(trip, trip, trip + cap, trip + 2 * cap, n,
                                                                       (int32_t)h->U, bucket, rank, sorted,
                                                                       sorted + cap, sorted + 2 * cap);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}
int32_t ensure_user_sort(gorse_mf *h) {
    const int64_t m = h->U + 2;
    // the tile x bin matrix of the binned preparation: prep_bins gives a chunk at most max(512, cap / 65536) tiles and min(kMaxBins - 1, U + 1) bins
    const size_t mat = prep_matrix_words(h->U, (int64_t)h->trip_cap);
    if ((size_t)m <= h->ubucket[0].n && h->scan_tmp2.n >= (size_t)ceil_div(m, kScanTile) && h->urank[0].n >= 2 * h->trip_cap &&
        h->ubins.n >= (size_t)2 * kMaxBins && h->ubinmat.n >= mat)
        return GORSE_OK;
    GORSE_TRY(mf_sync_streams(h));
    if (h->ubins.n < (size_t)2 * kMaxBins) GORSE_TRY(h->ubins.alloc((size_t)2 * kMaxBins));
    if (h->ubinmat.n < mat) GORSE_TRY(h->ubinmat.alloc(mat));
    for (int b = 0; b < 2; b++) {
        GORSE_TRY(h->ubucket[b].alloc((size_t)m));
        // per buffer (the sampler of chunk c + 1 ranks while chunk c is applied); twice the chunk: the binned preparation's (sample id,
        // user) pairs in sorted order
        GORSE_TRY(h->urank[b].alloc(2 * h->trip_cap));
    }
    if (h->scan_tmp2.n < (size_t)ceil_div(m, kScanTile)) GORSE_TRY(h->scan_tmp2.alloc((size_t)ceil_div(m, kScanTile)));
    return GORSE_OK;
}
// user runs need enough users to fill the chip with one 16-lane group each (4096 groups = one wave per SIMD) and a
// register-resident factor width; otherwise the per-sample schedule is the faster one (S-ml100k: 943 users)
bool user_runs_supported(const gorse_mf *h) {
    return (h->d == 8 || h->d == 16 || h->d == 32 || h->d == 64 || h->d == 128) && (h->U >= 4096 || (g_variant & 128));
}

// which item updates may take the store route (gorse_hip_test_set_bpr_store_mode)
// ST_NEG: measured at C3 whole (profiles/r04_b_probe_bpr_stores_*.txt): update kernel 89 -> 60 ms per epoch, 4 % of the updates of cold rows
// overwritten, NDCG@10 0.6100 against 0.6096 with atomics only (sequential oracle 0.6126); the positive side stays atomic -- by store
// it costs 0.003-0.005 of NDCG for 6 % more speed
constexpr int kDefaultStoreMode = 1;
int g_store_mode = kDefaultStoreMode;  // ST_* bits of bpr_update_user_kernel
#ifdef GORSE_PROBE
constexpr bool kProbeBuild = true;
#else
constexpr bool kProbeBuild = false;
#endif
int g_user_segments = 0;               // segments per user run of the SEG form (gorse_hip_test_set_bpr_user_segments): 0 = the library's choice
// segments per user run: the probe build's hook, nFactors <= 32
int user_segments(const gorse_mf *h) {
#ifdef GORSE_PROBE
    if (h->d <= 32 && g_user_segments > 0) return std::min(g_user_segments, 8);
#endif
    return 1;  // the shipped library runs one group per user run (see bpr_update_user_kernel, SEG)
}
int g_user_gpw = 4;                      // probe builds: groups of a wave that work in the ring kernel (4 = all)
int g_user_block = kBlock;               // probe builds: threads per workgroup of the ring kernel
int g_user_depth = 0;                  // probe builds: which (G, IA) pipeline of the atomics-only kernel (gorse_hip_test_set_bpr_user_depth)

HotRows make_hot(const gorse_mf *h) {
    // the eight replica rows of a slot lie next to each other (r04_a: 4 KB apart or n_hot rows apart makes no difference)
    return HotRows{h->hot_slot.p, h->hot_items.p, h->hot_rep.p, h->hot_done.p, 0, (int64_t)kHotReplicas * h->d, h->d};
}

int32_t launch_update_users(gorse_mf *h, const int32_t *sorted, const int32_t *bucket, size_t cap, float lr, float reg,
                            int exp_mode, double *loss, hipStream_t st, bool stores) {
    const int d = h->d;
    const int segs = user_segments(h);
    int64_t blocks = ceil_div(h->U * segs, (int64_t)kGroupsPerBlock * (d == 8 && GORSE_BPR_D8_PAIRS ? 2 : 1));  // nFactors 8: two runs per group
    const int64_t capb = 256 * 16;
    if (blocks > capb) blocks = capb;
    HotRows hot = make_hot(h);
    int folders = 0;
    if (h->n_hot > 0 && !(g_variant & 32)) {
        hot.n_hot = h->n_hot;
        folders = kFolderBlocks;  // hot_done is zero: gorse_mf_create, and every fold kernel leaves it so
    }
    blocks += folders;
    // the negative's slot look-up is one more gather per sample: only where a draw has a fair chance of meeting a hot item (C2: a
    // quarter of the items are hot; at the 10M x 1M set one in ten thousand, and the look-up cost 4 % of the epoch)
    const int neg_rep = !(g_variant & (1 << 25)) && (int64_t)hot.n_hot * 64 >= h->I ? 1 : 0;
#ifdef GORSE_PROBE
    if (g_variant & (1 << 23)) {
        static DevBuf<float> scratch;  // (a probe: never released)
        if (scratch.n < (size_t)h->I * d) GORSE_TRY(scratch.alloc((size_t)h->I * d));
        hot.warm_scratch = scratch.p;
    }
#endif
    dim3 grid((unsigned)blocks), block(kBlock);
#ifdef GORSE_PROBE
    const int64_t rblocks = std::min<int64_t>(ceil_div(h->U, g_user_block / 64 * g_user_gpw), capb * (kBlock / g_user_block)) + folders;
    dim3 rgrid((unsigned)rblocks), rblock(g_user_block);
#endif
    // cold items by store only in GORSE_BPR_HOGWILD_STORES and where the handle has any (n_cold) -- the atomics-only instantiation otherwise
    const int store_mode = stores && h->n_cold > 0 ? g_store_mode : 0;
    // the library ships two forms of the kernel (atomics only; cold negatives by store); the positive-side and re-reading forms of
    // the ablation (profiles/r04_*_probe_bpr_stores_*.txt) exist in `make probe-lib` builds only
#ifdef GORSE_PROBE
#define PROBE_CASES(NC) case 3: LAUNCH2(NC, 3, 2, 3); break; case 5: LAUNCH2(NC, 5, 2, 3); break; case 7: LAUNCH2(NC, 7, 2, 3); break;
#define PROBE_DEPTHS(NC)                                                                                               \
    case 1: LAUNCH2(NC, 0, 2, 4); break;                                                                               \
    case 2: LAUNCH2(NC, 0, 3, 5); break;                                                                               \
    case 3: LAUNCH2(NC, 0, 3, 6); break;                                                                               \
    case 4: LAUNCH2(NC, 0, 4, 6); break;                                                                               \
    case 5: LAUNCH2(NC, 0, 4, 8); break;                                                                               \
    case 6: LAUNCH2(NC, 0, 6, 9); break;                                                                               \
    case 10: LAUNCHR(NC, 3, 1); break;                                                                                 \
    case 11: LAUNCHR(NC, 4, 2); break;                                                                                 \
    case 12: LAUNCHR(NC, 6, 3); break;                                                                                 \
    case 13: LAUNCHR(NC, 8, 4); break;                                                                                 \
    case 14: LAUNCHR(NC, 6, 2); break;
#else
#define PROBE_CASES(NC)
#define PROBE_DEPTHS(NC)
#endif
#define LAUNCH2(NC, ST, G, IA)                                                                                         \
    do {                                                                                                               \
        if constexpr (NC <= 2 && G == 2 && IA == 3 && kProbeBuild) {                                                   \
            if (segs > 1) {                                                                                            \
                bpr_update_user_kernel<(NC == 0 ? 1 : NC), ST, NC == 0, G, IA, true><<<grid, block, 0, st>>>(          \
                    h->P.p, h->Q.p, sorted + cap, sorted + 2 * cap, bucket, (int32_t)h->U, d, lr, reg, exp_mode, loss, hot, folders, \
                    neg_rep, segs);                                                                                    \
                break;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        bpr_update_user_kernel<(NC == 0 ? 1 : NC), ST, NC == 0, G, IA><<<grid, block, 0, st>>>(                        \
            h->P.p, h->Q.p, sorted + cap, sorted + 2 * cap, bucket, (int32_t)h->U, d, lr, reg, exp_mode, loss, hot, folders, neg_rep); \
    } while (0)
#ifdef GORSE_PROBE
#define LAUNCHR(NC, R, D)                                                                                              \
    bpr_update_user_ring_kernel<(NC == 0 ? 1 : NC), NC == 0, R, D><<<rgrid, rblock, 0, st>>>(                          \
        h->P.p, h->Q.p, sorted + cap, sorted + 2 * cap, bucket, (int32_t)h->U, d, lr, reg, exp_mode, loss, hot, folders, \
        neg_rep | ((g_variant >> 23) & 2) | (g_user_gpw << 4))
#endif
#define LAUNCH(NC)                                                                                                     \
    do {                                                                                                               \
        switch (store_mode) {                                                                                          \
        case 1: LAUNCH2(NC, 1, 2, 3); break;                                                                           \
        PROBE_CASES(NC)                                                                                                \
        default:                                                                                                       \
            switch (g_user_depth) {                                                                                    \
            PROBE_DEPTHS(NC)                                                                                           \
            default: LAUNCH2(NC, 0, 2, 3); break;                                                                      \
            }                                                                                                          \
            break;                                                                                                     \
        }                                                                                                              \
    } while (0)
    if (d == 8)
        LAUNCH(0);
    else if (d == 16)
        LAUNCH(1);
    else if (d == 32)
        LAUNCH(2);
    else if (d == 64)
        LAUNCH(4);
    else
        LAUNCH(8);
#undef LAUNCH2
#undef LAUNCHR
#undef LAUNCH
#undef PROBE_CASES
#undef PROBE_DEPTHS
    GORSE_HIP_CHECK(hipGetLastError());
    if (folders > 0) {
        const int64_t fb = std::min<int64_t>(ceil_div((int64_t)hot.n_hot * d, 256), 512);
        bpr_fold_kernel<<<dim3((unsigned)fb), dim3(256), 0, st>>>(hot, h->Q.p, d);
        GORSE_HIP_CHECK(hipGetLastError());
    }
    return GORSE_OK;
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
    HotRows hot = make_hot(h);
    int folders = 0;
    if (MODE == MODE_ATOMIC && h->n_hot > 0 && !(g_variant & 32)) {
        hot.n_hot = h->n_hot;
        folders = kFolderBlocks;  // hot_done is zero: gorse_mf_create, and every fold kernel leaves it so
    }
    blocks += folders;
    dim3 grid((unsigned)blocks), block(kBlock);
#define LAUNCH(NC, SH)                                                                                               \
    bpr_update_kernel<NC, MODE><<<grid, block, SH, st>>>(h->P.p, h->Q.p, us, is, js, order, begin, end, d, lr, reg, \
                                                         exp_mode, loss, hot, folders)
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
    if (folders > 0) {
        const int64_t fb = std::min<int64_t>(ceil_div((int64_t)hot.n_hot * d, 256), 512);
        bpr_fold_kernel<<<dim3((unsigned)fb), dim3(256), 0, st>>>(hot, h->Q.p, d);
        GORSE_HIP_CHECK(hipGetLastError());
    }
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
                       hipStream_t st, int32_t *bucket = nullptr, int32_t *rank = nullptr) {
    if (n <= 0) return GORSE_OK;
    int64_t blocks = std::min<int64_t>(ceil_div(n, 256), 256 * 8);
    if (bucket) GORSE_HIP_CHECK(hipMemsetAsync(bucket, 0, (size_t)(h->U + 2) * sizeof(int32_t), st));
    bpr_sample_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((int32_t)h->U, (int32_t)h->I, h->uptr.p, h->uidx.p,
                                                                    h->uidx_sorted.p, seed, epoch, base, n, trip,
                                                                    trip + cap, trip + 2 * cap, h->fail_count.p, bucket, rank);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

// The user-run schedule's preparation of one chunk (see bpr_sample_user_kernel): user draws + ranks, scan of the run counters,
// sample ids scattered to their sorted positions, item draws by run.  `trip` (3 x cap ints, otherwise the unsorted triplets)
// holds the keys and (sample id, user) pairs; `rank` (2 x cap) the ranks or the sorted pairs; `sorted` receives si at cap, sj at
// 2 cap; bucket[0..U + 1] the run offsets.
// (PrepBins / prep_bins: csrc/bpr_bins.hpp, shared with the CPU cover test)
int32_t launch_prepare_users(gorse_mf *h, uint64_t seed, uint64_t epoch, int64_t base, int64_t n, int32_t *trip, int32_t *sorted,
                             int32_t *bucket, int32_t *rank, size_t cap, hipStream_t st) {
    if (n <= 0) return GORSE_OK;
    const int64_t m = h->U + 2;
    const int64_t blocks = std::min<int64_t>(ceil_div(n, 256), 256 * 8);
    const PrepBins pb = prep_bins(h->U, n);
    // the item draws: four positions per thread where the users' rows cannot all sit in the L2s (see bpr_sample_items_batch_kernel)
    auto launch_items = [&](const int2 *pairs) {
        if (h->uidx.n > ((size_t)8 << 20))
            bpr_sample_items_batch_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((int32_t)h->U, (int32_t)h->I, h->uptr.p, h->uidx.p,
                                                                                        h->uidx_sorted.p, seed, epoch, base, n, pairs,
                                                                                        sorted + cap, sorted + 2 * cap, h->fail_count.p);
        else
            bpr_sample_items_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((int32_t)h->U, (int32_t)h->I, h->uptr.p, h->uidx.p,
                                                                                  h->uidx_sorted.p, seed, epoch, base, n, pairs,
                                                                                  sorted + cap, sorted + 2 * cap, h->fail_count.p);
    };
    if (pb.ok && !(g_variant & (1 << 21))) {  // variant bit 21 (probes, tests): the preparation without bins
        int32_t *key = trip;
        int2 *bp = reinterpret_cast<int2 *>(trip + cap), *pairs = reinterpret_cast<int2 *>(rank);  // (rank: 2 x cap words, ensure_user_sort)
        int32_t *bin_count = h->ubins.p, *bin_start = h->ubins.p + kMaxBins, *H = h->ubinmat.p;
        const unsigned tiles = (unsigned)ceil_div(n, pb.tile);
        if ((size_t)tiles * pb.nbins > h->ubinmat.n) return fail(GORSE_ERR_INVALID, "tile x bin matrix smaller than %u x %d", tiles, pb.nbins);
        int tok = h->prof.begin(GORSE_PROF_BPR_SAMPLE, st);
        bpr_bin_count_kernel<<<dim3(tiles), dim3(kBinThreads), 0, st>>>((int32_t)h->U, h->uptr.p, seed, epoch, base, n, pb.tile, pb.shift,
                                                                        pb.nbins, key, h->fail_count.p, H);
        h->prof.end(tok, st);
        tok = h->prof.begin(GORSE_PROF_BPR_SORT, st);
        bpr_bin_offsets_kernel<<<dim3((unsigned)ceil_div(pb.nbins, 64)), dim3(kOffWaves * 64), 0, st>>>(H, (int)tiles, pb.nbins, bin_count);
        bpr_bin_scatter_kernel<<<dim3(tiles), dim3(kBinThreads), 0, st>>>((int32_t)h->U, key, n, pb.tile, pb.shift, pb.nbins, bin_count,
                                                                          H, bin_start, bp);
        bpr_bin_sort_kernel<<<dim3((unsigned)pb.nbins), dim3(256), 0, st>>>((int32_t)h->U, pb.shift, pb.nbins, bin_start, bp, bucket, pairs);
        h->prof.end(tok, st);
        tok = h->prof.begin(GORSE_PROF_BPR_SAMPLE, st);
        launch_items(pairs);
        h->prof.end(tok, st);
        GORSE_HIP_CHECK(hipGetLastError());
        return GORSE_OK;
    }
    int32_t *key = trip;
    int2 *pairs = reinterpret_cast<int2 *>(trip + cap);
    int tok = h->prof.begin(GORSE_PROF_BPR_SAMPLE, st);
    GORSE_HIP_CHECK(hipMemsetAsync(bucket, 0, (size_t)m * sizeof(int32_t), st));
    bpr_sample_user_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((int32_t)h->U, h->uptr.p, seed, epoch, base, n, key,
                                                                         h->fail_count.p, bucket, rank);
    h->prof.end(tok, st);
    tok = h->prof.begin(GORSE_PROF_BPR_SORT, st);
    GORSE_TRY(exclusive_scan_i32(bucket, m, h->scan_tmp2.p, st));
    bpr_scatter_ids_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(key, rank, bucket, n, (int32_t)h->U, pairs);
    h->prof.end(tok, st);
    tok = h->prof.begin(GORSE_PROF_BPR_SAMPLE, st);
    launch_items(pairs);
    h->prof.end(tok, st);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

// Samples per chunk.  The user-run schedule loads p_u once per RUN (a user's samples inside one chunk), so a chunk should hold
// a few dozen samples per user: 4M samples serve the S-ml1m and C3-shard shapes (<= 125K users); the full C3 set (1M users)
// takes 32M, the 10M-user set 128M -- 64 bytes of triplet / pair / sort buffers per sample of capacity, 8.2 GB of the 288 at most.
int64_t g_chunk_override = 0;  // probe: gorse_hip_test_set_bpr_chunk
int64_t chunk_capacity(const gorse_mf *h) {
    if (g_chunk_override > 0) return g_chunk_override;
    const int64_t lo = (int64_t)1 << 22, hi = (int64_t)1 << 27;
    return std::min(std::max(lo, 32 * h->U), hi);
}

int32_t ensure_trip(gorse_mf *h, int64_t want) {
    size_t cap = (size_t)std::min<int64_t>(std::max<int64_t>(want, 1), chunk_capacity(h));
    cap = (cap + 1) & ~(size_t)1;  // even: trip + cap is read as (sample id, user) pairs of 8 bytes
    if (cap <= h->trip_cap) return GORSE_OK;
    GORSE_TRY(mf_sync_streams(h));
    for (int b = 0; b < 2; b++) GORSE_TRY(h->trip[b].alloc(cap * 3));
    for (int b = 0; b < 2; b++) GORSE_TRY(h->sorted[b].alloc(cap * 3));
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

// Hogwild schedule: 1 = user runs (bpr_update_user_kernel), 0 = per-sample groups (bpr_update_kernel).  The
// environment variable GORSE_BPR_SCHEDULE = "users" | "samples" overrides the compiled default (read once);
// the probe bits of gorse_hip_test_set_variant override both.
int g_user_runs = 1;
bool user_runs_default() {
    static const int v = [] {
        const char *e = getenv("GORSE_BPR_SCHEDULE");
        if (e && !strcmp(e, "users")) return 1;
        if (e && !strcmp(e, "samples")) return 0;
        return g_user_runs;
    }();
    return v != 0;
}
bool user_runs_enabled() { return (g_variant & 128) ? true : ((g_variant & (1 << 28)) ? false : user_runs_default()); }
int g_exp_mode_exact = 0;  // exp flavour of the sequential schedule; tests flip it to 1 for bit parity

int32_t check_mode(int mode) {
    if (mode != MODE_ATOMIC && mode != MODE_EXACT && mode != MODE_RACY && mode != MODE_STORES)
        return fail(GORSE_ERR_INVALID, "unknown BPR mode %d", mode);
    return GORSE_OK;
}

int32_t epoch_impl(gorse_mf *h, int64_t n_samples, float lr, float reg, uint64_t seed, uint64_t epoch, int64_t base,
                   int mode, const volatile int32_t *cancel, double *loss_out, bool sync) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (n_samples < 0) return fail(GORSE_ERR_INVALID, "n_samples < 0");
    GORSE_TRY(check_mode(mode));
    const bool chained = h->ep_chain;  // the previous epoch was the last thing issued on this handle (mf_internal.hpp: ep_begin_prev)
    GORSE_TRY(h->use());
    if (n_samples == 0) {
        if (loss_out) *loss_out = 0;
        return GORSE_OK;
    }
    GORSE_TRY(ensure_trip(h, n_samples));
    GORSE_TRY(mf_epoch_begin(h, chained));  // epoch pacing: the (begin, end) pair gorse_mf_epoch_throttle / _times read (csrc/mf.hip)
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
        // two-stream pipeline: stream2 samples (and item-sorts) chunk c+1 while stream applies chunk c; the
        // buffer parity runs on across calls so that back-to-back enqueued epochs overlap as well
        const bool uruns = (mode == MODE_ATOMIC || mode == MODE_STORES) && user_runs_enabled() && user_runs_supported(h);
        if (uruns) GORSE_TRY(ensure_user_sort(h));
        int64_t c = 0;
        for (int64_t s0 = 0; s0 < n_samples; s0 += cap, c++) {
            const int b = (int)(h->chunk_seq & 1);
            h->chunk_seq++;
            const int64_t m = std::min(cap, n_samples - s0);
            if (cancel && *cancel) {
                GORSE_TRY(mf_sync_streams(h));
                return fail(GORSE_ERR_CANCELLED, "cancelled");
            }
            int32_t *tb = h->trip[b].p;
            // variant bit 22 (probes): the preparation on the update stream, so that a kernel timeline shows every kernel alone
            const hipStream_t prep = (g_variant & (1 << 22)) ? h->stream : h->stream2;
            // a never-recorded event is complete: the first two chunks of a handle do not wait
            GORSE_HIP_CHECK(hipStreamWaitEvent(prep, h->ev_consumed[b], 0));
            // user runs: the whole preparation of chunk c + 1 (launch_prepare_users) runs on the sampler stream under the update
            // kernel of chunk c.  Variant bit 27: the round-3 preparation (whole triplets sampled per sample, then scattered) with the sort on the
            // update stream; bit 26: the same with the sort on the sampler stream.
            const bool fused = uruns && !(g_variant & (1 << 29)) && !(g_variant & (1 << 27));
            const bool by_run = fused && !(g_variant & (1 << 26));
            if (by_run) {
                GORSE_TRY(launch_prepare_users(h, seed, epoch, base + s0, m, tb, h->sorted[b].p, h->ubucket[b].p, h->urank[b].p,
                                               (size_t)cap, prep));
            } else {
                int tok = h->prof.begin(GORSE_PROF_BPR_SAMPLE, prep);
                GORSE_TRY(launch_sampler(h, seed, epoch, base + s0, m, tb, (size_t)cap, prep, fused ? h->ubucket[b].p : nullptr,
                                         fused ? h->urank[b].p : nullptr));
                h->prof.end(tok, prep);
                if (fused) {
                    tok = h->prof.begin(GORSE_PROF_BPR_SORT, prep);
                    GORSE_TRY(launch_user_sort(h, tb, h->sorted[b].p, h->ubucket[b].p, h->urank[b].p, m, (size_t)cap, prep, true));
                    h->prof.end(tok, prep);
                }
            }
            GORSE_HIP_CHECK(hipEventRecord(h->ev_sampled[b], prep));
            GORSE_HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_sampled[b], 0));
            int tok;
            if (uruns && !fused) {
                tok = h->prof.begin(GORSE_PROF_BPR_SORT, h->stream);
                GORSE_TRY(launch_user_sort(h, tb, h->sorted[b].p, h->ubucket[b].p, h->urank[b].p, m, (size_t)cap, h->stream, false));
                h->prof.end(tok, h->stream);
            }
            tok = h->prof.begin(GORSE_PROF_BPR_UPDATE, h->stream);
            if (uruns)
                GORSE_TRY(launch_update_users(h, h->sorted[b].p, h->ubucket[b].p, (size_t)cap, lr, reg, g_exp_mode_exact, d_loss, h->stream,
                                              mode == MODE_STORES));
            else  // (the per-sample schedule has no store route: mode 3 is mode 0 there)
                GORSE_TRY(launch_update(h, mode == MODE_STORES ? MODE_ATOMIC : mode, tb, tb + cap, tb + 2 * cap, nullptr, 0, m, lr, reg, 0, d_loss, h->stream));
            h->prof.end(tok, h->stream);
            GORSE_HIP_CHECK(hipEventRecord(h->ev_consumed[b], h->stream));
            if (cancel && (c & 7) == 7) GORSE_TRY(mf_sync_streams(h));
        }
    }
    GORSE_TRY(mf_epoch_end(h));
    h->ep_chain = true;
    if (sync || loss_out) {
        if (loss_out)
            GORSE_HIP_CHECK(hipMemcpyAsync(loss_out, h->loss.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        GORSE_TRY(mf_sync_streams(h));
        GORSE_TRY(mf_epoch_harvest(h, true));  // every epoch issued so far is done: their device times are read now
    }
    return GORSE_OK;
}

}  // namespace

extern "C" void gorse_hip_test_set_exact_exp(int32_t mode) { g_exp_mode_exact = mode; }

extern "C" int32_t gorse_mf_bpr_schedule(gorse_mf *h, int32_t *user_runs) {
    if (!h || !user_runs) return fail(GORSE_ERR_INVALID, "NULL argument");
    *user_runs = (user_runs_enabled() && user_runs_supported(h)) ? 1 : 0;
    return GORSE_OK;
}
extern "C" void gorse_hip_test_set_variant(int32_t v) { g_variant = v; }
extern "C" void gorse_hip_test_set_bpr_user_depth(int32_t v) {
    g_user_depth = v & 0xff;
    g_user_block = ((v >> 8) & 0xfff) ? ((v >> 8) & 0xfff) : kBlock;  // bits 8..19: threads per workgroup of the ring kernel (64 / 128 / 256)
    g_user_gpw = (v >> 20) ? (v >> 20) : 4;                           // bits 20..: working groups per wave (1 / 2 / 4)
}
extern "C" void gorse_hip_test_set_bpr_chunk(int64_t samples) { g_chunk_override = samples; }
extern "C" void gorse_hip_test_set_bpr_user_segments(int32_t segments) { g_user_segments = segments < 0 ? 0 : segments; }
extern "C" void gorse_hip_test_set_bpr_store_mode(int32_t store_mode) {
    g_store_mode = store_mode < 0 ? kDefaultStoreMode : store_mode;
#ifndef GORSE_PROBE
    if (g_store_mode & ~1) g_store_mode &= 1;  // forms the shipped library does not carry fall back to the nearest one it does
#endif
}
extern "C" int32_t gorse_hip_test_probe_build(void) {
#ifdef GORSE_PROBE
    return 1;
#else
    return 0;
#endif
}

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

extern "C" int32_t gorse_hip_test_bpr_prepare_chunk(gorse_mf *h, int64_t n, uint64_t seed, uint64_t epoch, int64_t sample_base,
                                                    int32_t *off /*U + 2*/, int32_t *si /*n*/, int32_t *sj /*n*/) {
    if (!h || !off || !si || !sj) return fail(GORSE_ERR_INVALID, "NULL argument");
    GORSE_TRY(h->use());
    GORSE_TRY(ensure_trip(h, n));
    if (n < 0 || n > (int64_t)h->trip_cap) return fail(GORSE_ERR_INVALID, "n outside one chunk (%zu samples)", h->trip_cap);
    if (!user_runs_supported(h)) return fail(GORSE_ERR_INVALID, "this handle does not run the user-run schedule");
    GORSE_TRY(ensure_user_sort(h));
    GORSE_TRY(mf_sync_streams(h));
    const size_t cap = h->trip_cap;
    GORSE_TRY(launch_prepare_users(h, seed, epoch, sample_base, n, h->trip[0].p, h->sorted[0].p, h->ubucket[0].p, h->urank[0].p, cap,
                                   h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(off, h->ubucket[0].p, (size_t)(h->U + 2) * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(si, h->sorted[0].p + cap, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipMemcpyAsync(sj, h->sorted[0].p + 2 * cap, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
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
        } else if ((mode == MODE_ATOMIC || mode == MODE_STORES) && user_runs_enabled() && user_runs_supported(h)) {
            GORSE_TRY(ensure_user_sort(h));
            GORSE_TRY(launch_user_sort(h, tb, h->sorted[0].p, h->ubucket[0].p, h->urank[0].p, m, (size_t)cap, h->stream, false));
            GORSE_TRY(launch_update_users(h, h->sorted[0].p, h->ubucket[0].p, (size_t)cap, lr, reg, g_exp_mode_exact, nullptr,
                                          h->stream, mode == MODE_STORES));
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        } else {
            GORSE_TRY(launch_update(h, mode == MODE_STORES ? MODE_ATOMIC : mode, tb, tb + cap, tb + 2 * cap, nullptr, 0, m, lr, reg, 0, nullptr, h->stream));
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    }
    return GORSE_OK;
}
