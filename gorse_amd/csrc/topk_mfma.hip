// topk_mfma.hip -- path B of the exact top-k: a bf16 MFMA candidate sweep followed by exact rescoring.
// Reference semantics: common/ann/bruteforce.go:39-83 (keep the k smallest distances, return them ascending),
// distances computed by floats.Dot in AVX512 order (common/floats/src/floats_avx512.c:306-367).
//
// The reference scores all N vectors per query one pair at a time.  Here the N x nq score matrix is produced
// by v_mfma_f32_32x32x16_bf16 tiles and never leaves the CU: every wave keeps the operands of its 32*NCB queries
// in registers for the whole sweep, candidate rows stream through LDS, and the epilogue only compares the 16
// scores a lane holds (all of ONE query: the C layout puts a query per column) against that query's running
// filter threshold.  Scores that pass are appended to a small per-query candidate list in HBM; when a list
// fills, its wave raises the threshold to (K-th best approximate score so far) - margin and compacts the list.
//
// Exactness: |approx - reference| <= delta_q for every pair of one query (delta_q from a forward error bound
// of both summation orders, see topk_mfma_prepare).  Any vector whose reference score is among the K best has an
// approximate score >= (final K-th best approximate score) - 2*delta_q >= every threshold used during the sweep,
// so it is in the list.  The list is then rescored in the reference's own arithmetic order and ranked; a query
// whose K+1 best exact distances are not all distinct (where the reference's result depends on container/heap
// mechanics), whose list overflowed, or that saw a NaN is handed to path A (topk.hip), which replays the
// reference's heaps literally.  Results are therefore identical to path A's, ties included.
#include <algorithm>
#include <cmath>
#include <limits>

#include "rank_keys.hpp"
#include "topk_internal.hpp"
#include "topk_sym.hpp"

using namespace gorse;

namespace gorse {
int g_topk_force_path = 0;
int g_topk_variant = 0;  // probe switches: bit 0 = 64-row tiles, bit 1 = 128-row tiles, bit 2 / 3 = coarse scale test off / on
}

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxRB = 4;      // a tile holds RB 32-row MFMA blocks (RB = 2 or 4: 64 or 128 candidate rows per barrier)
#ifndef GORSE_SWEEP_WAVES
#define GORSE_SWEEP_WAVES 8
#endif
constexpr int kWavesMain = GORSE_SWEEP_WAVES;  // waves per workgroup of the main sweep, two per SIMD (a build-time probe switch)
#ifndef GORSE_SWEEP_NCB
#define GORSE_SWEEP_NCB 2
#endif
#ifndef GORSE_SWEEP_ORDER
#define GORSE_SWEEP_ORDER 0
#endif
#ifndef GORSE_SWEEP_PIPE
#define GORSE_SWEEP_PIPE 1
#endif
#ifndef GORSE_SWEEP_HALVES
#define GORSE_SWEEP_HALVES 1
#endif
#ifndef GORSE_SWEEP_YOUNG_PRIO
#define GORSE_SWEEP_YOUNG_PRIO 0  // A/B switch: 1 = the second-dispatched half of a workgroup's waves runs the tile loop at s_setprio 1
#endif
constexpr int kNcbMain = GORSE_SWEEP_NCB;  // 32-query column blocks per wave for operand depths up to 8 (probe switch)
// the history sweep serves the few queries with ties: small workgroups (2 waves = 64 * NCB queries) spread them over
// many CUs instead of a handful of 8-wave workgroups; deep operands keep more waves (the tile prefetch registers of a
// thread grow as the workgroup shrinks)
// Round 6 (GORSE_HIST_DMA): at the power-of-two depths that have LDS-DMA tiles the history sweep takes the main sweep's form instead --
// eight waves, 512 queries per workgroup, 64-row DMA tiles: the two-wave workgroups staged their tiles through registers and ran
// the matrix pipe at 0.22 PFLOP/s (C4: 11.7 ms for 10,136 queries against 9.5 ms; profiles/r06_zzk_probe_topk_c4_hist_dma_warm.txt).
// The other depths keep the two-wave form.
#ifndef GORSE_HIST_DMA
#define GORSE_HIST_DMA 1
#endif
constexpr bool sweep_hist_dma(int kp) { return GORSE_HIST_DMA && (kp == 2 || kp == 4 || kp == 8); }
constexpr int sweep_waves(bool hist, int kp) { return !hist || sweep_hist_dma(kp) ? kWavesMain : (kp <= 8 ? 2 : (kp <= 12 ? 4 : 8)); }
constexpr int kWaves = kWavesMain;
[[maybe_unused]] constexpr int kThreads = kWaves * 64;
// Tiles reach LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave-instruction, no VGPR round trip, no ds_write pass) when
// the operand depth is a power of two and the sweep is a main sweep: the LDS image of a tile is then lane-linear (unpadded
// rows), and the 16-byte pieces of a row are XOR-swizzled by the row number on the SOURCE address and again on the read, so
// that the 16 lanes of a ds_read_b128 group fall on 16 different slots of the 256-byte bank row.  Other depths (and their
// history sweeps' small workgroups) stage through registers into rows padded by 16 bytes.
constexpr bool sweep_dma(int kp, bool hist, int rb, int wv) {
    return (!hist || sweep_hist_dma(kp)) && (kp & (kp - 1)) == 0 && (32 * rb * kp * 2) % (64 * wv) == 0 && (32 * rb) % 64 == 0;
}
// LDS of a sweep workgroup: NBUF tile buffers (+ their row scales and, without DMA, block bounds), five words per query, the
// tile counters.  DMA: four buffers (tile t is multiplied while t + 1 has landed
// and t + 2 is in flight; the fourth keeps the waves a tile apart); register staging: three where they fit next to the rest
// in 144 KB (the main sweep up to KP = 8 with 128-row tiles), else two; the history sweep keeps two (its small workgroups
// share a CU).
constexpr size_t sweep_row_bytes(int kp, bool dma) { return (size_t)kp * 32 + (dma ? 0 : 16); }
constexpr size_t sweep_tile_bytes(int kp, int rb, bool dma) {
    return (size_t)32 * rb * sweep_row_bytes(kp, dma) + (size_t)32 * rb * 4 + (dma ? 0 : 2 * kMaxRB * 4);
}
constexpr size_t sweep_fixed_bytes(int bq) { return (size_t)5 * bq * 4 + 64; }
// SYM (the symmetric all-pairs sweep, see topk_sweep_kernel): per tile buffer the thresholds of the tile's rows AS QUERIES (their
// value and its raw-score form), per wave a staging area of kStageCap (key, row query | column) entries
constexpr int kStageCap = 128;      // staged foreign candidates per wave
constexpr int kStageFlushAt = 64;   // a wave empties its staging area once it holds this many
constexpr size_t sweep_sym_bytes(int rb, int wv, bool sym) { return sym ? (size_t)2 * 32 * rb * 4 : 0; }  // per tile buffer
constexpr size_t sweep_stage_bytes(int wv, bool sym) { return sym ? (size_t)wv * kStageCap * 8 + 64 : 0; }  // + 4 x kMaxRB block minima
constexpr int sweep_bufs(int kp, int rb, int bq, int wv, bool hist, bool sym = false) {
    if (sweep_dma(kp, hist, rb, wv))
        return 4 * (sweep_tile_bytes(kp, rb, true) + sweep_sym_bytes(rb, wv, sym)) + sweep_fixed_bytes(bq) + sweep_stage_bytes(wv, sym) <= (size_t)156 * 1024 ? 4 : 3;
    return !hist && 3 * sweep_tile_bytes(kp, rb, false) + sweep_fixed_bytes(bq) <= (size_t)144 * 1024 ? 3 : 2;
}
constexpr size_t sweep_lds_bytes(int kp, int rb, int bq, int wv, bool hist, bool sym = false) {
    return sweep_bufs(kp, rb, bq, wv, hist, sym) * (sweep_tile_bytes(kp, rb, sweep_dma(kp, hist, rb, wv)) + sweep_sym_bytes(rb, wv, sym)) +
           sweep_fixed_bytes(bq) + sweep_stage_bytes(wv, sym);
}
constexpr int64_t kMinSweepQueries = 768;  // fewer queries in a call take the scan (see topk_mfma_usable)
constexpr int kCap = 512;      // candidate-list capacity per query
constexpr int kCapF = 512;     // SYM: capacity of a query's FOREIGN list (candidates other workgroups found for it)
constexpr int kEPL = kCap / 64;
constexpr int kCompactAt = kCap - 64;   // compact a list once it holds more than this (a block adds <= 32)
constexpr int kOverflowAt = kCap - 128; // a compaction that keeps more than this cannot make progress
constexpr int kMaxKth = 256;
constexpr int kHistCap = 4096;          // history entries per query of the tie-replay sweep (see topk_tie_replay_kernel)
constexpr int kReplayCap = 8192;        // power of two >= kCap + kHistCap: entries the replay sorts
constexpr int64_t kReplayChunk = 16384; // flagged queries per history sweep (history buffers = 512 MB per row slice)
constexpr int kMaxSlices = 8;           // row slices of a history sweep (see topk_tie_sort_kernel)
constexpr int kMaxSlicesFew = 32;       // ... of one that serves few queries (chunk_finish)
constexpr int64_t kChunkQ = (int64_t)1 << 20;

using gorse::rank::fkey;      // order-preserving float -> uint of an approximate score, and back (rank_keys.hpp)
using gorse::rank::fkey_inv;
__device__ __forceinline__ int lane_rank(uint64_t m) {  // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

constexpr int kDefaultRowsPerTile = 128;  // one barrier per four MFMA row blocks: 8 % faster than 64 at C4 (profiles/r02_d_probe_topk_warm.txt)
inline int topk_rows_per_tile() { return (g_topk_variant & 1) ? 64 : ((g_topk_variant & 2) ? 128 : kDefaultRowsPerTile); }

struct SweepParams {
    const uint16_t *A;     // candidate operands, N x KPAD bf16
    const uint16_t *B;     // query operands, nq x KPAD bf16
    const float *rscale;   // per-row value of the epilogue (cosine scale, Euclidean bias, 1 for -dot; NaN: row not admissible)
    const float *qmargin;  // per-query 2*delta_q
    uint2 *cbuf;           // nq x kCap (key, index)
    int32_t *ccnt;         // nq
    uint8_t *cflag;        // nq: 1 = this path cannot decide the query
    uint2 *hbuf;           // HIST sweeps: nq x kHistCap entries a compaction dropped, in arrival order
    int32_t *hcnt;         // HIST sweeps: nq
    int64_t N, nq;
    int kth;
    int compact_at;        // a sub-list longer than this triggers the compaction of its query (<= kCompactAt / 2)
    int ep;                // EP_SCALE / EP_BIAS / EP_COARSE: what the epilogue does with the per-row values (host side: picks the kernel)
    unsigned long long *prof;  // PROF instantiations: 16 cycle / event counters summed over all waves
    // warm start (see topk_mfma_search): a PILOT sweep walks every tile_stride-th tile with kth = a small j and writes the
    // threshold it ends with to f_out; the main sweep starts from f0 and verifies it (compact_query: a threshold that a later
    // K-th-best bound does not reach flags the query 2 = "sweep again from -inf")
    const float *f0;   // nq initial thresholds or null (-inf)
    float *f_out;      // pilot: nq final thresholds; null otherwise
    int tile_stride;   // 1, or the pilot's sampling stride over the row tiles
    int probe;         // timing probes of the MAIN sweep (results are garbage): 1 = every threshold +inf (no block ever qualifies: the
                       // sweep's floor), 2 = a qualifying block does nothing, 3 = it tests and counts its candidates without storing them
    int nslices;       // HIST: row slices (grid.y); the per-query outputs are then nslices x nq long, slice-major
    const float *fwarm;  // HIST: nslices x nq starting thresholds of the slices (tie_warm_kernel: each a valid lower bound of the K-th
                         // best score among the rows IN FRONT of the slice, so nothing the reference's heap can have accepted is
                         // missed; -inf = cold), or null: every slice starts cold.  Trusted: a compaction never flags them.
    float rs_min, rs_max;  // smallest and largest row value of the index (EP_COARSE bound of the DMA sweeps)
    // SYM sweeps (the queries are the stored rows sym_q0 .. sym_q0 + nq): see topk_sweep_kernel
    int64_t sym_q0;        // row of query 0, a multiple of the tile height
    const float *frow;     // nq thresholds of the rows as queries (the pilot's: frozen for the foreign workgroups)
    const float *frow_raw; // RAWF: their raw-score form (raw_threshold of frow)
    uint2 *fbuf;           // nq x kCapF foreign lists
    int32_t *fcnt;         // nq entries appended to them (may exceed kCapF: the list then overflowed)
    unsigned long long *sym_stats;  // [0] queries without a pilot threshold, [1] warm starts the rescoring could not verify, [2] foreign
                                    // lists that overflowed, [3] (query, block) hits beyond a wave's staging area
    int sym_probe;         // timing probes of the symmetric sweep (results are garbage): 1 = no tile is read along its rows, 2 = the
                           // block and row tests run but a hit does nothing, 3 = hits are staged but never flushed
    // SYM, triangle sharding over GPUs (gorse_topk_tri_*): this launch sweeps the query blocks C with C % sym_world == sym_rank only --
    // their own lists, and what they find for the rows of EVERY earlier block (foreign lists of queries other ranks own included);
    // 0 of 1 = the whole triangle
    int sym_rank, sym_world;
    int sym_slices;  // SYM: row slices per query block (grid.y; 1 = the whole block's tiles in one workgroup)
};

// One wave raises the filter threshold of local query ql and compacts its list.
// HIST: the entries the new threshold drops are appended to the query's history (hb, counter s_hc[ql]) instead of
// being forgotten; list + history then hold every vector the reference's heap can have accepted.
// SYM: the list holds only the rows of the workgroup's OWN part of the sweep (the rest of the query's candidates are in its
// foreign list), so what it proves about the warm start is one-sided: a K-th-best bound that clears the threshold verifies it
// (the own rows are a subset of all rows) and raises it; one that does not proves nothing -- the warm start is then verified over
// both lists by topk_rescore_kernel, which learns from the final threshold (f_out) whether the sweep ever raised it.
template <bool HIST, bool SYM = false>
__device__ __forceinline__ void compact_query(uint2 *qb, int ql, int kth, int *s_cnt, float *s_f, const float *s_mg,
                                              uint8_t *flag, uint2 *hb = nullptr, int *s_hc = nullptr, bool trust = false) {
    const int lane = threadIdx.x & 63;
    // the list is two interleaved sub-lists: even slots belong to the lane holding rows 0-3, 8-11, ... of the query's
    // column (lane < 32), odd slots to its partner (lane >= 32); each appends at its own counter (slot 2 * c + half)
    const int n_lo = s_cnt[2 * ql], n_hi = s_cnt[2 * ql + 1];
    const int n = n_lo + n_hi;
    // The list is re-read by loads the compiler does not see (one asm statement: the wave's appends drained to L2, eight
    // L1-bypassing loads, their wait).  A load hipcc can see inside the tile loop makes its wait-count pass put
    // s_waitcnt vmcnt(0) at the head of every iteration (the load may be pending along the back edge), and that drains the
    // tiles the LDS-DMA has in flight.  All kCap slots are allocated, so the loads need no predicate; slots past the
    // sub-lists' lengths are masked below (key 0 = never counted: trial >= 1).
    static_assert(kEPL == 8, "compact_query loads eight entries per lane");
    unsigned long long ev[kEPL];
    {
        const uint2 *src = qb + lane;
        asm volatile("s_waitcnt vmcnt(0)\n\t"
                     "global_load_dwordx2 %0, %8, off sc1\n\t"
                     "global_load_dwordx2 %1, %8, off offset:512 sc1\n\t"
                     "global_load_dwordx2 %2, %8, off offset:1024 sc1\n\t"
                     "global_load_dwordx2 %3, %8, off offset:1536 sc1\n\t"
                     "global_load_dwordx2 %4, %8, off offset:2048 sc1\n\t"
                     "global_load_dwordx2 %5, %8, off offset:2560 sc1\n\t"
                     "global_load_dwordx2 %6, %8, off offset:3072 sc1\n\t"
                     "global_load_dwordx2 %7, %8, off offset:3584 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(ev[0]), "=&v"(ev[1]), "=&v"(ev[2]), "=&v"(ev[3]), "=&v"(ev[4]), "=&v"(ev[5]), "=&v"(ev[6]), "=&v"(ev[7])
                     : "v"(src)
                     : "memory");
    }
    uint32_t key[kEPL], idx[kEPL];
    bool valid[kEPL];
#pragma unroll
    for (int j = 0; j < kEPL; j++) {
        const int e = j * 64 + lane;
        valid[j] = (e & 1) ? (e >> 1) < n_hi : (e >> 1) < n_lo;
        key[j] = valid[j] ? (uint32_t)ev[j] : 0u;
        idx[j] = (uint32_t)(ev[j] >> 32);
    }
    float newf = -__builtin_inff();
    if (n >= kth) {
        // a lower bound of the K-th largest key: bisection on the 16 leading key bits (sign, exponent, 7 mantissa
        // bits), the rest left at zero.  Any bound <= the true K-th best keeps the exactness argument; the 2^-7
        // relative slack keeps a few more entries and halves the cost of a compaction.
        uint32_t prefix = 0;
        for (int b = 31; b >= 16; --b) {
            const uint32_t trial = prefix | (1u << b);
            int c = 0;
#pragma unroll
            for (int j = 0; j < kEPL; j++) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[j] >= trial));
            if (c >= kth) prefix = trial;
        }
        newf = fkey_inv(prefix) - s_mg[ql];
    }
    // SYM: the own list is a PART of the query's candidates, so its K-th best says nothing against the warm start: a bound
    // below the current threshold (or no bound at all) leaves threshold and list as they are
    if (SYM && newf <= s_f[ql]) {
        newf = s_f[ql];
        // nothing will be dropped: the next attempt waits for 64 more entries (s_hc, unused without HIST, holds the list length that
        // allows it) instead of following every append
        if (lane == 0) s_hc[ql] = n + 64;
    }
    // trust (a history slice started from tie_warm_kernel's bound): the starting threshold is a proven lower bound of the K-th best
    // score of the rows in front of the slice -- the slice's own rows need not prove it again, and it never falls
    if (trust && newf < s_f[ql]) newf = s_f[ql];
    if (newf < s_f[ql]) {
        // Only a warm-started threshold can be above what the list proves (thresholds derived from the list never fall):
        // the K-th best of everything above it is not known to clear it by the margin, so rows may have been dropped
        // wrongly.  The query is swept again from -inf by the host (flag 2).
        if (lane == 0) {
            s_f[ql] = __builtin_inff();
            s_cnt[2 * ql] = 0;  // the list is dropped: nothing of it triggers another compaction
            s_cnt[2 * ql + 1] = 0;
            *flag = 2;
        }
        return;
    }
    int base = 0;
    int hbase = HIST ? s_hc[ql] : 0;
    bool hist_full = false;
#pragma unroll
    for (int j = 0; j < kEPL; j++) {
        const bool keep = valid[j] && (fkey_inv(key[j]) >= newf);
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) qb[base + lane_rank(m)] = make_uint2(key[j], idx[j]);  // survivors packed into slots 0 .. base-1
        base += __builtin_popcountll(m);
        if (HIST) {
            const uint64_t md = __builtin_amdgcn_ballot_w64(valid[j] && !keep);
            const int nd = __builtin_popcountll(md);
            if (hbase + nd > kHistCap)
                hist_full = true;
            else if (valid[j] && !keep)
                hb[hbase + lane_rank(md)] = make_uint2(key[j], idx[j]);
            if (!hist_full) hbase += nd;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        s_cnt[2 * ql] = (base + 1) >> 1;  // slots 0 .. base-1 read as the two interleaved sub-lists again
        s_cnt[2 * ql + 1] = base >> 1;
        s_f[ql] = newf;
        if (HIST) s_hc[ql] = hbase;
        if (base > kOverflowAt || hist_full) {  // ties / a margin too wide for the list: stop collecting, path A decides
            s_f[ql] = __builtin_inff();
            *flag = 1;
        }
    }
}

// What the epilogue does with a block's 32 x 32 raw scores before it compares them with the thresholds:
constexpr int EP_NONE = 0;    // nothing (-dot without a mask; register-staged sweeps only)
constexpr int EP_SCALE = 1;   // every score times its row's value (cosine with unequal norms; masked rows: NaN)
constexpr int EP_BIAS = 2;    // every score plus its row's value (Euclidean: -|x|^2 / 2)
constexpr int EP_COARSE = 3;  // the block's LARGEST raw score times the extreme row value bounds every scaled score of the block
                              // (values positive and nearly equal, or all 1): the rows are scaled only when that bound reaches a threshold

// Maxima of MFMA results are taken by instructions the COMPILER emits (fmaxf nests become v_max3_f32): an inline-asm
// v_max3_f32 reading an accumulator right behind the MFMA that writes it gets no wait states from hipcc (asm statements are
// not padded: cdna_hip_programming.md 5.7) and reads the registers before the matrix pipe has written them -- measured:
// maxima too small, blocks with a candidate skipped, 23,000 of a million queries failing the warm start's verification.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float max2f(float a, float b) { return fmaxf(a, b); }

// PROF (probe only): s_memtime stamps around the phases of the tile loop, summed over the waves into p.prof:
// [0] tile movement (DMA issue / tile store + prefetch issue), [1] MFMA + epilogues (candidate paths included), [2] candidate
// paths alone, [3] waiting for a tile, [4] row blocks examined, [5] row blocks that took the candidate path, [6] whole kernel,
// [7] waves, [8] candidate path: set-up + row scaling, [9] appends, [10] compaction check / compaction, [11] candidate paths
//
// SYM -- the symmetric form of an all-pairs sweep.  When the queries are the stored rows sym_q0 .. sym_q0 + nq themselves, the
// score of (query c, row r) is also the score of (query r, row c): the workgroup of query block C multiplies only the row tiles
// that do NOT lie in a later query block (the rows before the query range, the query blocks 0 .. C, the rows behind the range) and,
// for the tiles of the EARLIER query blocks, reads every 32 x 32 block of scores a second time along its rows: in the MFMA's D
// layout a lane holds ONE column and sixteen rows, so the second test is sixteen compares against the rows' own thresholds
// (brought into LDS with the tile).  What passes is a candidate of the ROW's query among the columns: it is staged in LDS per
// wave and appended -- in batches, by returning atomics on a per-query counter -- to that query's FOREIGN list; foreign
// thresholds are the pilot's and stay frozen (any threshold that was ever valid is a valid lower bound).  Every (query, row) pair
// is covered exactly once: by the query's own workgroup when the row's tile is not one it skips, else by the workgroup of the
// row's block (tests/test_topk_sym_schedule_cpu.py walks the schedule).  Workgroups are issued longest first (block C = grid - 1 -
// blockIdx.x sweeps C + 1 query blocks): half the MFMA work of the square sweep, and no partial last round.
template <int KP, int NCB, int EP, bool HIST, int RB, bool PROF = false, bool SYM = false>
__global__ __launch_bounds__(64 * sweep_waves(HIST, KP), (sweep_waves(HIST, KP) + 3) / 4 < 2 ? 2 : (sweep_waves(HIST, KP) + 3) / 4) void topk_sweep_kernel(SweepParams p) {
    constexpr int kWaves = sweep_waves(HIST, KP);
    constexpr int kThreads = kWaves * 64;
    constexpr int kTR = 32 * RB;
    constexpr bool SCALE = EP != EP_NONE;  // a per-row value travels with the tile
    unsigned long long c_store = 0, c_comp = 0, c_slow = 0, c_bar = 0, n_blk = 0, n_slow = 0, t_begin = 0, ts = 0;
    unsigned long long c_free = 0, c_issue = 0, c_land = 0, c_post = 0;  // the tile top: buffer wait, DMA issue, landing wait, announcement
    unsigned long long c_s1 = 0, c_s2 = 0, c_s3 = 0, n_hits = 0;
    if (PROF) t_begin = __builtin_amdgcn_s_memtime();
    constexpr int KPAD = KP * 16;
    constexpr bool DMA = sweep_dma(KP, HIST, RB, kWaves);
    constexpr bool PIPE = GORSE_SWEEP_PIPE && KP <= 8 && !HIST;  // row blocks software-pipelined: see the row-block loop
    static_assert(!DMA || SCALE, "the DMA sweeps mask the rows past N through the row values (NaN padding)");
    // register staging: +16 B per row, consecutive rows start 4 banks apart, ds_read_b128 conflict-free; DMA: unpadded rows,
    // the pieces of a row swizzled instead (see piece_swizzle)
    constexpr int ROWB = (int)sweep_row_bytes(KP, DMA);
    constexpr int QW = 32 * NCB;
    constexpr int BQ = QW * kWaves;
    constexpr int CHUNKS = kTR * KP * 2;  // 16-byte pieces per tile
    constexpr int CPT = (CHUNKS + kThreads - 1) / kThreads;
    constexpr int NBUF = sweep_bufs(KP, RB, BQ, kWaves, HIST, SYM);
    static_assert(!SYM || (DMA && !HIST && !PROF && BQ % kTR == 0 && EP != EP_NONE), "SYM: a DMA main sweep whose query blocks are whole tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *s_tile = smem;
    float *s_rs = reinterpret_cast<float *>(smem + (size_t)NBUF * kTR * ROWB);
    float *s_fq = s_rs + NBUF * kTR;                // SYM: the thresholds of the tile's rows as queries ...
    float *s_fqr = s_fq + (SYM ? NBUF * kTR : 0);   // ... and their raw-score form (RAWF)
    float *s_bmm = s_fqr + (SYM ? NBUF * kTR : 0);  // register staging only: per buffer and 32-row block the (min, max) row scale
    int *s_cnt = reinterpret_cast<int *>(s_bmm + (DMA ? 0 : NBUF * 2 * kMaxRB));  // 2 per query: the interleaved sub-list lengths
    float *s_f = reinterpret_cast<float *>(s_cnt + 2 * BQ);
    float *s_mg = s_f + BQ;
    int *s_hc = reinterpret_cast<int *>(s_mg + BQ);
    // tile counters: s_sync[b] = waves whose part of a tile has reached buffer b (ever), s_sync[NBUF + b] = waves that
    // have finished reading one
    int *s_sync = s_hc + BQ;
    uint2 *s_stage = reinterpret_cast<uint2 *>(s_sync + 16);  // SYM: kWaves x kStageCap staged foreign candidates
    float *s_fmin = reinterpret_cast<float *>(s_stage + kWaves * kStageCap);  // SYM: per tile buffer and 32-row block the smallest row threshold

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // SYM: longest workgroups first; of a triangle shard the blocks sym_rank, sym_rank + sym_world, ... (the grid holds exactly those)
    const int cblk = SYM ? p.sym_rank + p.sym_world * (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int64_t wgq0 = (int64_t)cblk * BQ;
    for (int t = tid; t < BQ; t += kThreads) {
        const int64_t q = wgq0 + t;
        s_cnt[2 * t] = 0;
        s_cnt[2 * t + 1] = 0;
        s_hc[t] = 0;
        s_f[t] = q < p.nq && p.probe != 1 ? (p.f0 ? p.f0[q] : (HIST && p.fwarm ? p.fwarm[(int64_t)blockIdx.y * p.nq + q] : -__builtin_inff()))
                                            : __builtin_inff();
        s_mg[t] = q < p.nq ? p.qmargin[q] : 0.0f;
    }
    if (tid < 2 * NBUF) s_sync[tid] = 0;
    // query operands: resident in registers for the whole sweep
    bf16x8 bfrag[NCB][KP];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        int64_t q = wgq0 + w * QW + cb * 32 + (lane & 31);
        if (q >= p.nq) q = p.nq - 1;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.B + q * KPAD) + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KP; ks++) {
            uint4 v = src[ks * 2];
            bfrag[cb][ks] = *reinterpret_cast<bf16x8 *>(&v);
        }
    }
    // the operands are consumed here once, so that hipcc waits for their loads NOW: a load still pending when the tile loop is
    // entered costs a wait in every iteration (and that wait drains the LDS-DMA of the tiles in flight)
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int ks = 0; ks < KP; ks++) asm volatile("" : "+v"(bfrag[cb][ks]));

    const int64_t stride_rows = (int64_t)p.tile_stride * kTR;  // the pilot samples every tile_stride-th tile
    // A history sweep may be cut into row slices (blockIdx.y): every slice sweeps its own tiles from a cold threshold and
    // keeps its own lists; topk_tie_sort_kernel joins them (see there for why that is a superset of the unsliced history).
    const int64_t NT_all = (p.N + stride_rows - 1) / stride_rows;
    const int64_t T0 = NT_all * blockIdx.y / gridDim.y, T1 = NT_all * (blockIdx.y + 1) / gridDim.y;
    // SYM (one slice, stride 1): the tiles [skip_lo, skip_lo + skip_n) -- whole tiles of LATER query blocks -- are left to those
    // blocks' workgroups; the tiles [tr_lo, tr_hi) -- the earlier query blocks -- are read along their rows as well
    const SymSchedule sched(SYM ? p.sym_q0 : 0, SYM ? p.nq : 0, kTR, BQ, cblk);  // topk_sym.hpp
    // SYM with row slices (gridDim.y > 1: the long blocks of a triangle shard): slice y takes an even share of the tiles the block
    // multiplies, with its own candidate lists (qslice below), the foreign side unchanged
    const int64_t NTs = SYM ? sched.tiles(NT_all) : 0;
    const int64_t S0 = SYM ? NTs * blockIdx.y / gridDim.y : 0, S1 = SYM ? NTs * (blockIdx.y + 1) / gridDim.y : 0;
    const int64_t NT = SYM ? S1 - S0 : T1 - T0;
    auto tile_index = [&](int64_t tl) -> int64_t {  // the tl-th tile this workgroup multiplies
        if constexpr (SYM) return sched.tile_index(S0 + tl);
        else return T0 + tl;
    };
    const int64_t qslice = (int64_t)blockIdx.y * p.nq;  // this slice's rows of the per-query output arrays

    // ---- register staging (history sweeps, operand depths that are not a power of two) ----
    uint4 pre[DMA ? 1 : CPT];
    float pre_rs = 0.0f;
    auto load_tile = [&](int64_t tl) {
        const int64_t base_row = tile_index(tl) * stride_rows;
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const int ch = tid + c * kThreads;
            if (ch < CHUNKS) {
                const int row = ch / (KP * 2), cc = ch % (KP * 2);
                const int64_t grow = base_row + row;
                pre[DMA ? 0 : c] = grow < p.N ? reinterpret_cast<const uint4 *>(p.A + grow * KPAD)[cc] : make_uint4(0, 0, 0, 0);
            }
        }
        if (SCALE && tid < kTR) pre_rs = base_row + tid < p.N ? p.rscale[base_row + tid] : 0.0f;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const int ch = tid + c * kThreads;
            if (ch < CHUNKS) {
                const int row = ch / (KP * 2), cc = ch % (KP * 2);
                *reinterpret_cast<uint4 *>(s_tile + (size_t)buf * kTR * ROWB + row * ROWB + cc * 16) = pre[DMA ? 0 : c];
            }
        }
        if (SCALE && tid < kTR) {
            s_rs[buf * kTR + tid] = pre_rs;
            // (min, max) of the 32 scales of this row block; rows past N carry 0 and never qualify anyway
            float mn = pre_rs > 0.0f ? pre_rs : __builtin_inff(), mx = pre_rs;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                mn = fminf(mn, __shfl_xor(mn, o, 64));
                mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            }
            if ((tid & 31) == 0) {
                s_bmm[(buf * kMaxRB + (tid >> 5)) * 2 + 0] = mn;
                s_bmm[(buf * kMaxRB + (tid >> 5)) * 2 + 1] = mx;
            }
        }
    };

    // ---- LDS-DMA ----
    // Piece c (16 bytes) of row r lives at piece c ^ piece_swizzle(r) of the row's LDS image: the 16 lanes of a ds_read_b128
    // group read one piece column of 16 different rows, and the XOR spreads them over the 16 slots of the 256-byte bank row.
    constexpr int PPR = KP * 2;  // pieces per row
    auto piece_swizzle = [](int row) -> int {
        if constexpr (PPR >= 16) return row & 15;
        else if constexpr (PPR == 8) return (row >> 1) & 7;
        else if constexpr (PPR == 4) return (row >> 2) & 3;
        else return (row >> 3) & 1;
    };
    // A wave-instruction moves 64 pieces.  The two waves that share a SIMD (w and w + kWaves / 2: a workgroup's waves go round
    // the four SIMDs) take turns: the lower half of the waves moves the even tiles, the upper half the odd ones, so that
    // while one of the two spends its ~150 cycles per DMA instruction the other has the MFMA pipe to itself -- issued by all
    // eight at the top of every tile, both waves of every SIMD stood in that phase together (17 % of the sweep,
    // profiles/r03_l_probe_c4_prof.txt).  DMA instruction j of issuing wave wi covers pieces [(wi * DPW + j) * 64, +64) of
    // the tile; the row values (one dword per row) follow as instructions of 64 rows each, issued by the first kTR / 64 of them.
    constexpr int ISS = !DMA ? 1 : (GORSE_SWEEP_HALVES && kWaves >= 4 ? kWaves / 2 : kWaves);  // waves that move one tile
    constexpr bool HALVES = DMA && ISS < kWaves;
    constexpr int DPW = DMA ? CHUNKS / 64 / ISS : 0;  // tile instructions per issuing wave
    static_assert(!DMA || (CHUNKS % (64 * ISS) == 0 && kTR % 64 == 0 && kTR / 64 <= ISS), "DMA tiling");
    constexpr int RSW = DMA ? kTR / 64 : 0;              // issuing waves that also move a row-value instruction
    const int wu = __builtin_amdgcn_readfirstlane(w);    // provably wave-uniform (LDS destinations, branch conditions)
    const int wi = wu % ISS, grp = wu / ISS;             // place among the waves that move a tile; which tiles (parity)
    constexpr int ISS_OR_ALL = DMA ? ISS : kWaves;       // announcements that complete a tile
    // The DMA is issued from inline assembly: the builtin makes hipcc treat every later LDS access of the kernel (the tile
    // counters, the fragment reads) as dependent on it and put s_waitcnt vmcnt(0) in front -- which drains the very tiles
    // that are meant to stay in flight.  An asm statement is invisible to that bookkeeping (cdna_hip_programming.md 5.7);
    // the waits are dma_wait's, by count.  M0 = LDS destination of lane 0; the statement saves and restores it.
    // Addressing: the tile's first byte is a scalar (it advances by one tile per iteration), the lane's place inside the
    // tile a 32-bit register computed once -- global_load_lds with an SGPR base and a VGPR offset.  The last tile, whose
    // rows may lie past N, clamps its rows instead (their scores are discarded through the NaN padding of the row values).
    const unsigned lds_tiles = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s_tile;
    const unsigned lds_rs = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)s_rs;
    const unsigned lds_fq = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)s_fq;
    const unsigned lds_fqr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)s_fqr;
    constexpr bool RAWF_ = EP == EP_COARSE && DMA;  // (RAWF below)
    constexpr int NXD = SYM ? (RAWF_ ? 2 : 1) : 0;  // SYM: extra row-value DMA instructions per tile of the first RSW issuing waves
    unsigned dsrc[DMA ? DPW : 1];  // byte offset of this lane's piece of instruction j from the tile's first byte
    if constexpr (DMA) {
#pragma unroll
        for (int j = 0; j < DPW; j++) {
            const int piece = (wi * DPW + j) * 64 + lane;
            const int row = piece / PPR, cs = piece % PPR;  // LDS position; its content is source piece cs ^ swizzle
            dsrc[j] = (unsigned)(row * (KPAD * 2) + ((cs ^ piece_swizzle(row)) << 4));
        }
    }
    auto dma_tile = [&](int64_t tl, int buf) {
        if constexpr (DMA) {
            const int64_t base_row = tile_index(tl) * stride_rows;
            const bool inside = base_row + kTR <= p.N;  // wave-uniform
            const unsigned char *tile0 = reinterpret_cast<const unsigned char *>(p.A) + base_row * (KPAD * 2);
#pragma unroll
            for (int j = 0; j < DPW; j++) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_tiles + (unsigned)buf * (kTR * ROWB) + (unsigned)(wi * DPW + j) * 1024u);
                unsigned keep;
                if (inside) {
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(dsrc[j]), "s"(tile0), "s"(dst) : "memory");
                } else {
                    const int piece = (wi * DPW + j) * 64 + lane;
                    const int row = piece / PPR, cs = piece % PPR;
                    int64_t grow = base_row + row;
                    if (grow >= p.N) grow = p.N - 1;
                    const uint16_t *src = p.A + grow * KPAD + ((cs ^ piece_swizzle(row)) << 3);
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                }
            }
            if (wi < RSW) {  // the row values are padded with NaN past N (topk_mfma_prepare): no clamp
                const float *src = p.rscale + base_row + wi * 64 + lane;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_rs + (unsigned)(buf * kTR + wi * 64) * 4u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                if constexpr (SYM) {
                    // the thresholds of the tile's rows as queries (any valid address for a tile outside the query range: its
                    // values are never looked at)
                    int64_t qi = base_row - p.sym_q0 + wi * 64 + lane;
                    qi = qi < 0 ? 0 : (qi >= p.nq ? p.nq - 1 : qi);
                    const float *s1 = p.frow + qi;
                    const unsigned d1 = __builtin_amdgcn_readfirstlane(lds_fq + (unsigned)(buf * kTR + wi * 64) * 4u);
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(s1), "s"(d1) : "memory");
                    if constexpr (RAWF_) {
                        const float *s2 = p.frow_raw + qi;
                        const unsigned d2 = __builtin_amdgcn_readfirstlane(lds_fqr + (unsigned)(buf * kTR + wi * 64) * 4u);
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(s2), "s"(d2) : "memory");
                    }
                }
            }
        }
    };
    // "everything this wave issued before the DMA instructions of its last `keep` (0 or 1) tiles has completed": loads (the DMA
    // among them) retire in order, so once no more than one tile's instructions are outstanding, every older tile of this wave
    // is in LDS -- whatever stores (the candidate appends) are still on their way.
    auto dma_wait = [&](int keep) {
        if constexpr (DMA) {
            if (keep == 0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (wi < RSW)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW + 1 + NXD) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
        }
    };
    auto post = [&](int *ctr) {  // DMA: this wave's part of a tile is in LDS (dma_wait came first)
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // SYM: the waves that moved the rows' thresholds (64 rows each) leave the smallest one of every 32-row block next to them, before
    // they announce the tile: a block whose largest score stays below it needs no row-by-row test (LDS runs a wave's operations in order)
    auto landed = [&](int b) {
        if constexpr (SYM) {
            if (wi < RSW) {
                float v = (RAWF_ ? s_fqr : s_fq)[b * kTR + wi * 64 + lane];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
                if ((lane & 31) == 0) s_fmin[b * kMaxRB + wi * 2 + (lane >> 5)] = v;
            }
        }
    };

    // The waves of a workgroup share every tile but do not march in step: a wave that takes the candidate path
    // used to hold the other seven at the tile's barrier -- 27 % of the sweep (profiles/r02_f_probe_topk_prof.txt).  With NBUF
    // buffers and two counters per buffer a wave only waits for what it needs: tile t complete in its buffer before it
    // multiplies it, the tile that buffer held before read by everyone before it is overwritten.
    // LDS executes a wave's operations in issue order, so a counter increment behind the wave's own LDS traffic needs no
    // fence (a workgroup-scope release would also drain the DMA in flight: s_waitcnt vmcnt(0)); the compiler is kept from
    // moving memory operations across by the asm statements' memory clobbers.
    auto arrive = [&](int *ctr) {
        if constexpr (DMA)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_reads of the tile have returned
        else
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's LDS traffic has completed
        if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_for = [&](int *ctr, int target) {
        // bounded (a few seconds): a protocol error shows as wrong rows in the parity tests, not as a hung device
        for (unsigned spins = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && spins < (1u << 26); spins++)
            __builtin_amdgcn_s_sleep(1);
        if constexpr (DMA)
            asm volatile("" ::: "memory");
        else
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    __syncthreads();  // the per-query words and the counters are initialised
    if constexpr (HALVES) {
        if (grp == 0 && NT > 0) {
            dma_tile(0, 0);
            dma_wait(0);
            landed(0);
            post(&s_sync[0]);
        }
        if (grp == 1 && NT > 1) dma_tile(1, 1);  // announced at the top of tile 0
    } else if constexpr (DMA) {
        if (NT > 0) dma_tile(0, 0);
        if (NT > 1) dma_tile(1, 1);
        if (NT > 0) {
            dma_wait(NT > 1 ? 1 : 0);  // tile 0 has landed (tile 1 may be in flight)
            landed(0);
            post(&s_sync[0]);
        }
    } else {
        if (NT > 0) {
            load_tile(0);
            store_tile(0);
            arrive(&s_sync[0]);
        }
        if (NT > 1) load_tile(1);
    }

    // RAWF (EP_COARSE with the index-wide bounds of the row values): the block and quad tests compare RAW maxima with the
    // threshold divided by the extreme row value, rounded down -- fl(raw * s) >= f with s <= rs_max (f > 0), or s >= rs_min and
    // raw < 0 (f <= 0), implies raw >= fraw -- so that rows are scaled only inside a quad that holds a candidate
    constexpr bool RAWF = EP == EP_COARSE && DMA;
    auto raw_threshold = [&](float f) -> float {
        const float q = f / (f > 0.0f ? p.rs_max : p.rs_min);
        return fabsf(q) == __builtin_inff() ? q : q - fabsf(q) * 2.4e-7f;
    };
    float fth[NCB], fraw[RAWF ? NCB : 1];
    int cnt[NCB];      // length of this lane's sub-list of its query (mirrored in s_cnt around a compaction)
    uint2 *mine[NCB];  // the lane's sub-list: slots 2 * c + (lane >> 5) of its query's list (see compact_query)
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const int ql = w * QW + cb * 32 + (lane & 31);
        fth[cb] = s_f[ql];
        if (RAWF) fraw[RAWF ? cb : 0] = raw_threshold(fth[cb]);
        cnt[cb] = 0;
        mine[cb] = p.cbuf + (qslice + wgq0 + ql) * kCap + (lane >> 5);
    }
    // DMA: the lane's read offsets inside a 32-row block, one per k-step (row & 15 is the same in every block of a tile)
    unsigned aoff[DMA ? KP : 1];
    if constexpr (DMA) {
        const int r = lane & 31;
#pragma unroll
        for (int ks = 0; ks < KP; ks++) aoff[ks] = (unsigned)(r * ROWB + (((ks * 2 + (lane >> 5)) ^ piece_swizzle(r)) << 4));
    }

    // SYM: what the column's own row contributes to a score read along the rows (its scale / bias; NaN for a column that must
    // not emit: a lane past nq, a column whose own tile also holds rows outside the query range -- every workgroup multiplies
    // that tile itself -- and a masked row)
    float csv[NCB];
    int stage_n = 0;  // staged foreign candidates of this wave (wave-uniform)
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        csv[cb] = __builtin_nanf("");
        if constexpr (SYM) {
            const int64_t q = wgq0 + w * QW + cb * 32 + (lane & 31);
            if (SymSchedule::column_emits(p.sym_q0, p.nq, kTR, q)) csv[cb] = p.rscale[p.sym_q0 + q];
            asm volatile("" : "+v"(csv[cb]));  // consumed now: no load pending when the tile loop is entered
        }
    }
    // the staged entries to their queries' foreign lists: one returning atomic per entry, issued from an asm statement together
    // with its wait (as compact_query's loads: a memory operation hipcc can see inside the tile loop costs a vmcnt(0) per tile)
    auto flush_stage = [&]() {
        if constexpr (SYM) {
            uint2 *stg = s_stage + w * kStageCap;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int e0 = 0; e0 < stage_n; e0 += 64) {
                const int e = e0 + lane;
                const bool on = e < stage_n;
                const uint2 ent = stg[on ? e : 0];
                const uint32_t rowq = ent.y & 0xfffffu, col = ent.y >> 20;
                int32_t *ctr = p.fcnt + rowq;
                int slot = 0;
                const int one = 1;
                if (on)
                    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(slot) : "v"(ctr), "v"(one) : "memory");
                if (on && slot < kCapF) p.fbuf[(int64_t)rowq * kCapF + slot] = make_uint2(ent.x, (uint32_t)(p.sym_q0 + wgq0 + col));
            }
            stage_n = 0;
        }
    };
    // the fragments of the row block about to be multiplied (PIPE), read from block-0-relative address `blk` of a tile
    bf16x8 af[PIPE ? KP : 1];
    auto frag = [&](const unsigned char *blk, int ks) -> bf16x8 {
        return DMA ? *reinterpret_cast<const bf16x8 *>(blk + aoff[DMA ? ks : 0])
                   : *reinterpret_cast<const bf16x8 *>(blk + (lane & 31) * ROWB + (lane >> 5) * 16 + ks * 32);
    };
    // MI355X_MICROARCH.md "Two waves per SIMD", item 4: of the two waves that share a SIMD the younger one (w >= kWaves / 2) loses the
    // issue arbitration on every segment; a static priority for that half is the A/B switch GORSE_SWEEP_YOUNG_PRIO
    const bool young = GORSE_SWEEP_YOUNG_PRIO && !HIST && wu >= kWaves / 2;
    auto base_prio = [&]() {
        if (young)
            __builtin_amdgcn_s_setprio(1);
        else
            __builtin_amdgcn_s_setprio(0);
    };
    if (GORSE_SWEEP_YOUNG_PRIO) base_prio();
    int buf = 0, round = 0;  // tile t lives in buffer t % NBUF and is that buffer's (t / NBUF)-th tile
    for (int64_t t = 0; t < NT; t++) {
        if (PROF) ts = __builtin_amdgcn_s_memtime();
        const int nb = buf + 1 == NBUF ? 0 : buf + 1;
        if constexpr (HALVES) {
            // This wave's half moves the tiles of its parity: at such a tile's top it issues tile t + 2 (into the buffer tile
            // t + 2 - NBUF was read from); at the other tiles' tops it waits for what it issued a tile ago -- everything it has
            // in flight: the appends of the last row blocks ride along, a few hundred cycles now and then -- and announces
            // tile t + 1, three row blocks ahead of its first use.
            const int b2 = nb + 1 == NBUF ? 0 : nb + 1;
            if ((int)(t & 1) == grp) {
                if (t + 2 < NT) {
                    if (t + 2 >= NBUF) {
                        unsigned long long tw = 0;
                        if (PROF) tw = __builtin_amdgcn_s_memtime();
                        wait_for(&s_sync[NBUF + b2], kWaves * (int)((t + 2) / NBUF));
                        if (PROF) {
                            const unsigned long long now = __builtin_amdgcn_s_memtime();
                            c_bar += now - tw;
                            c_free += now - tw;
                        }
                    }
                    unsigned long long ti = 0;
                    if (PROF) ti = __builtin_amdgcn_s_memtime();
                    dma_tile(t + 2, b2);
                    if (PROF) c_issue += __builtin_amdgcn_s_memtime() - ti;
                }
            } else if (t + 1 < NT) {
                unsigned long long ti = 0;
                if (PROF) ti = __builtin_amdgcn_s_memtime();
                dma_wait(0);
                if (PROF) {
                    const unsigned long long now = __builtin_amdgcn_s_memtime();
                    c_land += now - ti;
                    ti = now;
                }
                landed(nb);
                post(&s_sync[nb]);
                if (PROF) c_post += __builtin_amdgcn_s_memtime() - ti;
            }
        } else if constexpr (DMA) {
            // tile t + 2 goes into the buffer tile t + 2 - NBUF was read from; tile t + 1 (issued a tile ago) has landed by now
            // and is announced here, one tile ahead of its use, so that a wave may run a tile ahead of the slowest one
            const int b2 = nb + 1 == NBUF ? 0 : nb + 1;
            if (t + 2 < NT) {
                if (t + 2 >= NBUF) {
                    unsigned long long tw = 0;
                    if (PROF) tw = __builtin_amdgcn_s_memtime();
                    wait_for(&s_sync[NBUF + b2], kWaves * (int)((t + 2) / NBUF));
                    if (PROF) {
                        const unsigned long long now = __builtin_amdgcn_s_memtime();
                        c_bar += now - tw;
                        c_free += now - tw;
                    }
                }
                unsigned long long ti = 0;
                if (PROF) ti = __builtin_amdgcn_s_memtime();
                dma_tile(t + 2, b2);
                if (PROF) c_issue += __builtin_amdgcn_s_memtime() - ti;
            }
            if (t + 1 < NT) {
                unsigned long long ti = 0;
                if (PROF) ti = __builtin_amdgcn_s_memtime();
                dma_wait(t + 2 < NT ? 1 : 0);
                if (PROF) {
                    const unsigned long long now = __builtin_amdgcn_s_memtime();
                    c_land += now - ti;
                    ti = now;
                }
                landed(nb);
                post(&s_sync[nb]);
                if (PROF) c_post += __builtin_amdgcn_s_memtime() - ti;
            }
        } else {
            if (t + 1 < NT) {  // rows of tile t + 1 (loaded during tile t - 1) into the buffer tile t + 1 - NBUF was read from
                if (t + 1 >= NBUF) {
                    unsigned long long tw = 0;
                    if (PROF) tw = __builtin_amdgcn_s_memtime();
                    wait_for(&s_sync[NBUF + nb], kWaves * (round + (nb == 0 ? 1 : 0)));  // = kWaves * ((t + 1) / NBUF)
                    if (PROF) c_bar += __builtin_amdgcn_s_memtime() - tw;
                }
                store_tile(nb);
                arrive(&s_sync[nb]);
            }
            if (t + 2 < NT) load_tile(t + 2);
        }
        if (PROF) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            c_store += now - ts;
            ts = now;
        }
        // PIPE: tile t has been waited for inside the last row block of tile t - 1 (its first fragments are in registers)
        if (!PIPE || t == 0) wait_for(&s_sync[buf], ISS_OR_ALL * (round + 1));
        if (PROF) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            c_bar += now - ts;
            ts = now;
        }
        const unsigned char *tb = s_tile + (size_t)buf * kTR * ROWB;
        if (PIPE && t == 0) {
#pragma unroll
            for (int ks = 0; ks < KP; ks++) af[PIPE ? ks : 0] = frag(tb, ks);
        }
        const int64_t base_row = tile_index(t) * stride_rows;
        // SYM: a tile of an earlier query block is read along its rows too (wave-uniform)
        const bool trans = SYM && sched.transposed(tile_index(t)) && p.sym_probe != 1;
        // register staging zero-fills the rows past N; their scores are set to NaN below.  The DMA sweeps need nothing: the
        // row values of those rows are NaN
        const int valid = DMA ? kTR : (int)std::min<int64_t>(kTR, p.N - base_row);
        // not unrolled: four copies of the epilogue and its candidate path cost the C4 instantiation 34 spilled VGPRs
#pragma unroll 1
        for (int rb = 0; rb < RB; rb++) {
            f32x16 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[cb][r] = 0.0f;
            if constexpr (PIPE) {
                // Software pipeline over row blocks (and across tiles): the fragments of block rb are in registers when its
                // MFMAs start; each register quad is refilled with the NEXT block's fragment as soon as the last MFMA reading it
                // is issued, so the LDS round trip runs under the rest of the chain and the epilogue instead of ahead of the
                // first MFMA.  The next block of the last row block is block 0 of tile t + 1, announced a tile ago.
                const unsigned char *nx = tb + (rb + 1) * 32 * ROWB;
                if (rb + 1 == RB) {
                    nx = tb;  // last tile: a re-read nobody uses
                    if (t + 1 < NT) {
                        unsigned long long tw = 0;
                        if (PROF) tw = __builtin_amdgcn_s_memtime();
                        wait_for(&s_sync[nb], ISS_OR_ALL * (round + (nb == 0 ? 1 : 0) + 1));
                        if (PROF) c_bar += __builtin_amdgcn_s_memtime() - tw;
                        nx = s_tile + (size_t)nb * kTR * ROWB;
                    }
                }
#if GORSE_SWEEP_ORDER == 1
#pragma unroll
                for (int cb = 0; cb + 1 < NCB; cb++)
#pragma unroll
                    for (int ks = 0; ks < KP; ks++)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfrag[cb][ks], acc[cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KP; ks++) {
                    acc[NCB - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfrag[NCB - 1][ks], acc[NCB - 1], 0, 0, 0);
                    af[ks] = frag(nx, ks);
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
#pragma unroll
                for (int ks = 0; ks < KP; ks++) {
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfrag[cb][ks], acc[cb], 0, 0, 0);
                    af[ks] = frag(nx, ks);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            } else {
            const unsigned char *rowp = tb + (rb * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
            const unsigned char *blkp = tb + rb * 32 * ROWB;  // DMA: + aoff[ks]
            // candidate fragments: up to 8 k-steps of ds_read_b128 in flight ahead of the MFMAs that use them
            constexpr int PF = KP < 8 ? KP : 8;
#pragma unroll
            for (int k0 = 0; k0 < KP; k0 += PF) {
                bf16x8 a[PF];
#pragma unroll
                for (int j = 0; j < PF; j++)
                    if (k0 + j < KP)
                        a[j] = DMA ? *reinterpret_cast<const bf16x8 *>(blkp + aoff[DMA ? k0 + j : 0])
                                   : *reinterpret_cast<const bf16x8 *>(rowp + (k0 + j) * 32);
                __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead: hipcc otherwise sinks each next to its MFMA
#pragma unroll
                for (int j = 0; j < PF; j++)
                    if (k0 + j < KP) {
#pragma unroll
                        for (int cb = 0; cb < NCB; cb++)
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], bfrag[cb][k0 + j], acc[cb], 0, 0, 0);
                    }
            }
            }
            // C layout: lane holds column (= query) lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5).
            // The epilogue shares the SIMD's issue slots with the sibling wave's MFMAs (about five ordinary instructions fit
            // beside one MFMA: MI355X_MICROARCH.md), so it is counted in instructions: 10 for the maxima of a column block,
            // 4 for the bound, 2 for the vote.
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                if constexpr (SYM) {
                    if (trans) {
                        // the block read along its rows: score (query = row, candidate = this lane's column) against the ROW's threshold.
                        // RAWF: raw scores against the rows' raw thresholds, the column's scale only behind that test.
                        const float4 *t4 = reinterpret_cast<const float4 *>((RAWF ? s_fqr : s_fq) + buf * kTR + rb * 32);
                        const float cv = csv[cb];
                        auto col_score = [&](float raw) -> float { return EP == EP_BIAS ? raw + cv : raw * cv; };
                        // first the block's largest score against the smallest threshold of its 32 rows (landed): most blocks end here
                        const float fmin_blk = s_fmin[buf * kMaxRB + rb];
                        float mq[4];
#pragma unroll
                        for (int g = 0; g < 4; g++) mq[g] = max2f(max3f(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2]), acc[cb][4 * g + 3]);
                        const float mraw = max2f(max3f(mq[0], mq[1], mq[2]), mq[3]);
                        if (__builtin_amdgcn_ballot_w64((RAWF ? mraw : col_score(mraw)) >= fmin_blk) != 0) {
                        // sixteen differences and their maxima by quad (the compiler turns a chain of sixteen compares into a bit mask
                        // built from v_cndmask / shifts, five instructions per score); a NaN score (a column that must not emit)
                        // is skipped by the hardware maximum, -0 from a flushed difference only opens the exact test below
                        float qd[4];
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const float4 th = t4[2 * g + (lane >> 5)];
                            const float a0 = RAWF ? acc[cb][4 * g + 0] : col_score(acc[cb][4 * g + 0]);
                            const float a1 = RAWF ? acc[cb][4 * g + 1] : col_score(acc[cb][4 * g + 1]);
                            const float a2 = RAWF ? acc[cb][4 * g + 2] : col_score(acc[cb][4 * g + 2]);
                            const float a3 = RAWF ? acc[cb][4 * g + 3] : col_score(acc[cb][4 * g + 3]);
                            qd[g] = max2f(max3f(a0 - th.x, a1 - th.y, a2 - th.z), a3 - th.w);
                        }
                        const float dm = max2f(max3f(qd[0], qd[1], qd[2]), qd[3]);
                        if (__builtin_amdgcn_ballot_w64(dm >= 0.0f) != 0 && p.sym_probe != 2) {
                            const float4 *f4 = reinterpret_cast<const float4 *>(s_fq + buf * kTR + rb * 32);
                            const uint32_t colw = (uint32_t)(w * QW + cb * 32 + (lane & 31)) << 20;  // the column inside the workgroup's block
                            uint2 *stg = s_stage + w * kStageCap;
#pragma unroll
                            for (int g = 0; g < 4; g++) {
                                if (__builtin_amdgcn_ballot_w64(qd[g] >= 0.0f) == 0) continue;
                                const float4 fr4 = f4[2 * g + (lane >> 5)];
                                const float frow[4] = {fr4.x, fr4.y, fr4.z, fr4.w};
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int r = 4 * g + e;
                                    const float sc = col_score(acc[cb][r]);  // NaN for a column that must not be emitted (csv)
                                    const bool pass = sc >= frow[e];
                                    const uint64_t m = __builtin_amdgcn_ballot_w64(pass);
                                    if (m != 0) {
                                        const int c = __builtin_popcountll(m);
                                        const uint32_t rowq = (uint32_t)(base_row - p.sym_q0) + (uint32_t)(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
                                        if (stage_n + c <= kStageCap) {
                                            if (pass) stg[stage_n + lane_rank(m)] = make_uint2(fkey(sc), rowq | colw);
                                            stage_n += c;
                                        } else if (pass) {
                                            p.cflag[rowq] = 1;  // more hits in one block than the staging area holds: that query takes the tie path
                                            atomicAdd(p.sym_stats + 3, 1ull);
                                        }
                                    }
                                }
                            }
                            if (stage_n >= kStageFlushAt) {
                                if (p.sym_probe == 3)
                                    stage_n = 0;
                                else
                                    flush_stage();
                            }
                        }
                        }
                    }
                }
                auto scale_rows = [&]() {
                    const float4 *r4 = reinterpret_cast<const float4 *>(s_rs + buf * kTR + rb * 32);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const float4 s = r4[2 * g + (lane >> 5)];
                        if (EP == EP_BIAS) {  // q.x - |x|^2 / 2 = (|q|^2 - |q - x|^2) / 2: same order as the distance
                            acc[cb][4 * g + 0] += s.x;
                            acc[cb][4 * g + 1] += s.y;
                            acc[cb][4 * g + 2] += s.z;
                            acc[cb][4 * g + 3] += s.w;
                        } else {
                            acc[cb][4 * g + 0] *= s.x;
                            acc[cb][4 * g + 1] *= s.y;
                            acc[cb][4 * g + 2] *= s.z;
                            acc[cb][4 * g + 3] *= s.w;
                        }
                    }
                };
                if (EP == EP_SCALE || EP == EP_BIAS) scale_rows();
                if (!DMA && valid < kTR) {  // last tile: rows past N never qualify (NaN fails every >=, the maxima skip it)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (row >= valid) acc[cb][r] = __builtin_nanf("");
                    }
                }
                // maxima of the four register quads (rows 8 g + 4 (lane >> 5) + 0..3), then of the block: the candidate path
                // looks only into the quads whose maximum qualifies
                float gm[4];
#pragma unroll
                for (int g = 0; g < 4; g++) gm[g] = max2f(max3f(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2]), acc[cb][4 * g + 3]);
                float m = max2f(max3f(gm[0], gm[1], gm[2]), gm[3]);
                if (EP == EP_COARSE && !RAWF) {  // an upper bound of every scaled score of the block: the row values are positive
                    const float mn = s_bmm[(buf * kMaxRB + rb) * 2 + 0];
                    const float mx = s_bmm[(buf * kMaxRB + rb) * 2 + 1];
                    m = m * (m >= 0.0f ? mx : mn);
                }
                if (PROF) n_blk++;
                if (__builtin_amdgcn_ballot_w64(m >= (RAWF ? fraw[RAWF ? cb : 0] : fth[cb])) != 0) {
                    unsigned long long tsl = 0;
                    if (PROF) {
                        tsl = __builtin_amdgcn_s_memtime();
                        n_slow++;
                    }
                    __builtin_amdgcn_s_setprio(3);  // the wave on this path is the one its workgroup waits for
                    if (EP == EP_COARSE && !RAWF) {
                        scale_rows();
#pragma unroll
                        for (int g = 0; g < 4; g++) gm[g] = max2f(max3f(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2]), acc[cb][4 * g + 3]);
                    }
                    float4 rs4[RAWF ? 4 : 1];  // RAWF: the row values of the four quads, in flight while the quads are tested
                    if (RAWF) {
                        const float4 *r4 = reinterpret_cast<const float4 *>(s_rs + buf * kTR + rb * 32);
#pragma unroll
                        for (int g = 0; g < 4; g++) rs4[RAWF ? g : 0] = r4[2 * g + (lane >> 5)];
                    }
                    const float f = fth[cb];
                    const float fq = RAWF ? fraw[RAWF ? cb : 0] : f;  // what a quad's maximum is compared with
                    // Every lane appends to ITS OWN sub-list of the query (even / odd slots, see compact_query), so no slot
                    // exchange between the two lanes of a query and no counting pass are needed.  f is +inf for lanes past nq
                    // and for flagged queries; rows past N are NaN.  A block that comes here holds one or two candidates among
                    // its 1024 scores: a quad is opened only if its maximum qualifies (a NaN maximum never does: the hardware
                    // maximum skips NaN operands, and a quad of four NaNs holds no candidate).
                    unsigned long long tq = 0;
                    int cnt_was = 0;
                    if (PROF) {
                        tq = __builtin_amdgcn_s_memtime();
                        c_s1 += tq - tsl;
                        cnt_was = cnt[cb];
                    }
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        if (gm[g] >= fq && p.probe != 2) {
                            if (RAWF) {
                                const float4 sc = rs4[RAWF ? g : 0];
                                acc[cb][4 * g + 0] *= sc.x;
                                acc[cb][4 * g + 1] *= sc.y;
                                acc[cb][4 * g + 2] *= sc.z;
                                acc[cb][4 * g + 3] *= sc.w;
                            }
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int r = 4 * g + e;
                                if (acc[cb][r] >= f) {
                                    const uint32_t row = (uint32_t)(base_row + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
                                    if (p.probe != 3) mine[cb][2 * cnt[cb]] = make_uint2(fkey(acc[cb][r]), row);
                                    cnt[cb]++;
                                }
                            }
                        }
                    }
                    if (PROF) {
                        const unsigned long long now = __builtin_amdgcn_s_memtime();
                        c_s2 += now - tq;
                        tq = now;
                        n_hits += __builtin_amdgcn_ballot_w64(cnt[cb] != cnt_was) != 0;  // blocks that did append something
                    }
                    uint64_t need = __builtin_amdgcn_ballot_w64(cnt[cb] > p.compact_at);
                    if (need) {
                        const int ql = w * QW + cb * 32 + (lane & 31);
                        s_cnt[2 * ql + (lane >> 5)] = cnt[cb];
                        need = (need | (need >> 32)) & 0xffffffffull;  // either sub-list of a query
                        do {
                            const int l = __builtin_ctzll(need);
                            need &= need - 1;
                            const int qlc = w * QW + cb * 32 + l;
                            // (see compact_query; a sub-list holds kCap / 2 entries and a block adds at most 16 to one)
                            if (SYM && s_cnt[2 * qlc] + s_cnt[2 * qlc + 1] < s_hc[qlc] && s_cnt[2 * qlc] <= kCap / 2 - 32 &&
                                s_cnt[2 * qlc + 1] <= kCap / 2 - 32)
                                continue;
                            compact_query<HIST, SYM>(p.cbuf + (qslice + wgq0 + qlc) * kCap, qlc, p.kth, s_cnt, s_f, s_mg,
                                                p.cflag + qslice + wgq0 + qlc,
                                                HIST ? p.hbuf + (qslice + wgq0 + qlc) * kHistCap : nullptr, s_hc, HIST && p.fwarm != nullptr);
                        } while (need);
                        cnt[cb] = s_cnt[2 * ql + (lane >> 5)];
                        fth[cb] = s_f[ql];
                        if (RAWF) fraw[RAWF ? cb : 0] = raw_threshold(fth[cb]);
                    }
                    base_prio();
                    if (PROF) {
                        const unsigned long long now = __builtin_amdgcn_s_memtime();
                        c_s3 += now - tq;
                        c_slow += now - tsl;
                    }
                }
            }
        }
        if (PROF) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            c_comp += now - ts;
            ts = now;
        }
        arrive(&s_sync[NBUF + buf]);  // this wave has read tile t
        buf = nb;
        if (nb == 0) round++;
    }
    if (PROF && lane == 0) {
        atomicAdd(p.prof + 0, c_store);
        atomicAdd(p.prof + 1, c_comp);
        atomicAdd(p.prof + 2, c_slow);
        atomicAdd(p.prof + 3, c_bar);
        atomicAdd(p.prof + 4, n_blk);
        atomicAdd(p.prof + 5, n_slow);
        atomicAdd(p.prof + 6, (unsigned long long)__builtin_amdgcn_s_memtime() - t_begin);
        atomicAdd(p.prof + 7, 1ull);
        atomicAdd(p.prof + 8, c_s1);
        atomicAdd(p.prof + 9, c_s2);
        atomicAdd(p.prof + 10, c_s3);
        atomicAdd(p.prof + 11, n_hits);
        atomicAdd(p.prof + 12, c_free);
        atomicAdd(p.prof + 13, c_issue);
        atomicAdd(p.prof + 14, c_land);
        atomicAdd(p.prof + 15, c_post);
    }
    if (SYM && stage_n > 0) flush_stage();
    // final threshold + compaction of every list this wave owns
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) s_cnt[2 * (w * QW + cb * 32 + (lane & 31)) + (lane >> 5)] = cnt[cb];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int l = 0; l < QW; l++) {
        const int ql = w * QW + l;
        if (wgq0 + ql >= p.nq) break;
        const int64_t qg = qslice + wgq0 + ql;
        if (s_f[ql] == __builtin_inff()) {  // flagged
            if (p.f_out && lane == 0) p.f_out[qg] = -__builtin_inff();
            continue;
        }
        compact_query<HIST, SYM>(p.cbuf + qg * kCap, ql, p.kth, s_cnt, s_f, s_mg, p.cflag + qg,
                            HIST ? p.hbuf + qg * kHistCap : nullptr, s_hc, HIST && p.fwarm != nullptr);
        if (lane == 0) {
            p.ccnt[qg] = s_cnt[2 * ql] + s_cnt[2 * ql + 1];  // packed by the final compaction: slots 0 .. count-1
            if (HIST) p.hcnt[qg] = s_hc[ql];
            if (p.f_out) p.f_out[qg] = s_f[ql] == __builtin_inff() ? -__builtin_inff() : s_f[ql];
        }
    }
}

// A stored vector into the LDS row of a 16-lane group: from the bf16 rows as given when the index is bf16 (half the bytes,
// the same fp32 values once widened), else from the fp32 rows.
__device__ __forceinline__ void gather_row(float *row, const float *X, const uint16_t *Xb, int64_t i, int d, int lane) {
    if (Xb && (d & 7) == 0) {  // 16-byte pieces
        const uint4 *src = reinterpret_cast<const uint4 *>(Xb + i * d);
        for (int e = lane; e < d / 8; e += kGroup) {
            const uint4 v = src[e];
            float4 *dst = reinterpret_cast<float4 *>(row + 8 * e);
            dst[0] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                                 __uint_as_float(v.y & 0xffff0000u));
            dst[1] = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16),
                                 __uint_as_float(v.w & 0xffff0000u));
        }
    } else if (Xb) {
        for (int e = lane; e < d; e += kGroup) row[e] = __uint_as_float((uint32_t)Xb[i * d + e] << 16);
    } else {
        for (int e = lane; e < d; e += kGroup) row[e] = X[i * d + e];
    }
}

// ---- exact rescoring + ranking of one query's candidate list ------------------------------------------
struct RescoreParams {
    const float *X;        // N x d fp32 stored vectors
    const uint16_t *Xb;    // the same vectors as given in bf16 (a bf16 index), or null: the rows are gathered from here -- half
                           // the bytes, the same fp32 values once widened
    const float *norm2;    // N
    const float *Qf;       // nq x d fp32 query vectors, or null (queries are stored vectors)
    const float *qn2;      // nq query norms (cosine)
    const int64_t *qid;    // stored-vector id per query, or null
    int64_t q0;            // used when qid == null and Qf == null: id = q0 + t
    const uint2 *cbuf;
    const int32_t *ccnt;
    uint8_t *cflag;
    int d, metric, k, prune0;
    int64_t expect;        // min(k, number of admissible vectors)
    int32_t *out_idx;
    float *out_dist;
    int32_t *out_cnt;
    // SYM sweeps: the query's foreign list, and what the verification of its warm start needs (see below)
    const uint2 *fbuf;     // nq x kCapF, or null
    const int32_t *fcnt;
    const float *f0;       // the thresholds the main sweep started from
    const float *ffinal;   // the thresholds its workgroups ended with (above f0: raised, hence verified, by the own list)
    unsigned long long *sym_stats;  // (SweepParams)
    const float *qmargin;
    int kth;
    int tri_rank, tri_world, tri_bq;  // triangle sharding: only the queries of the blocks this rank owns (t / tri_bq % tri_world == tri_rank)
    int own_slices;                   // SYM sweeps cut into row slices: own lists / counters / flags / final thresholds per slice ...
    int64_t own_stride;               // ... own_stride queries apart (slice-major)
};
constexpr int kMaxOwnSlices = 4;
constexpr int kCapT = kCap + kCapF;  // candidates the rescoring takes per query: its own list + its foreign list

__global__ __launch_bounds__(kBlock) void topk_rescore_kernel(RescoreParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int d = p.d;
    float *sq = smem_f;                          // d
    float *sx = sq + d;                          // 2 * kGroupsPerBlock * d
    float *s_e = sx + (size_t)2 * kGroupsPerBlock * d;  // kCapT
    int *s_i = reinterpret_cast<int *>(s_e + kCapT);  // kCapT
    int *s_misc = s_i + kCapT;                   // [0] nonpositive in top-k, [1] flag, [2] SYM: candidates that clear the warm start,
                                                 // [3], [4] the histogram selection's bin and remainder, [5] survivors of the pruning
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (kGroup - 1), gib = tid / kGroup;
    const int k = p.k;
    if (p.tri_world > 1 && (int)((t / p.tri_bq) % p.tri_world) != p.tri_rank) return;  // another rank's query
    // SYM with row slices (a triangle shard whose long blocks are cut over several workgroups: topk_sweep_kernel, sym_slices): the
    // query has one own list, counter, flag and final threshold PER SLICE, slice-major like a history sweep's
    const int nsl = p.own_slices > 1 ? p.own_slices : 1;
    {
        uint8_t f = p.cflag[t];
        for (int sl = 1; sl < nsl; sl++) {
            const uint8_t g = p.cflag[(int64_t)sl * p.own_stride + t];
            f = f ? f : g;
        }
        if (f) {  // path A / the tie path fills this row (a flag of a later slice moves to where the host's list is made from)
            if (tid == 0 && !p.cflag[t]) p.cflag[t] = f;
            return;
        }
    }
    int n_own_at[kMaxOwnSlices + 1];
    n_own_at[0] = 0;
#pragma unroll
    for (int sl = 0; sl < kMaxOwnSlices; sl++) n_own_at[sl + 1] = n_own_at[sl] + (sl < nsl ? p.ccnt[(int64_t)sl * p.own_stride + t] : 0);
    const int n_own = n_own_at[kMaxOwnSlices];
    const int n_for = p.fbuf ? p.fcnt[t] : 0;
    if (n_for > kCapF || n_own + (n_for > 0 ? n_for : 0) > kCapT) {  // the foreign list overflowed (or, sliced, the lists together exceed
        if (tid == 0) {                                                // what this kernel ranks): the tie path sweeps this query on its own
            p.cflag[t] = 1;
            atomicAdd(p.sym_stats + 2, 1ull);
        }
        return;
    }
    const int n_all = n_own + n_for;
    const uint2 *fb = p.fbuf ? p.fbuf + t * kCapF : nullptr;
    const int64_t self = p.Qf ? -1 : (p.qid ? p.qid[t] : p.q0 + t);
    const float *qrow = p.Qf ? p.Qf + t * d : p.X + self * d;
    for (int e = tid; e < d; e += kBlock) sq[e] = qrow[e];
    if (tid < 6) s_misc[tid] = 0;
    __syncthreads();
    const VecShape vs(d);
    const float qq = p.metric == GORSE_METRIC_COSINE ? p.qn2[t] : 0.0f;
    const uint2 *cb = p.cbuf + t * kCap;
    // the candidates into registers: kCapT / kBlock per thread
    constexpr int EPT = kCapT / kBlock;
    uint32_t ekey[EPT], eidx[EPT];
    bool ev[EPT];
#pragma unroll
    for (int s = 0; s < EPT; s++) {
        const int c = tid + s * kBlock;
        ev[s] = c < n_all;
        uint2 ent = make_uint2(0u, 0u);
        if (ev[s]) {
            if (c >= n_own) {
                ent = fb[c - n_own];
            } else {
                int sl = 0;
#pragma unroll
                for (int u = 1; u < kMaxOwnSlices; u++) sl += c >= n_own_at[u] ? 1 : 0;
                ent = cb[(int64_t)sl * p.own_stride * kCap + (c - n_own_at[sl])];
            }
        }
        ekey[s] = ent.x;
        eidx[s] = ent.y;
    }
    // SYM: a warm start that the query's own list could not verify is verified here over both lists -- the threshold f0 is valid
    // iff (kth-th best approximate score) - margin >= f0, and with the own threshold still at f0 the two lists hold every row that
    // reaches f0.  (A final own threshold above f0 was derived from -- and verified by -- the own list; f0 = -inf needs no proof.)
    const float mg = p.qmargin[t];
    // (sliced: a slice that raised its threshold above f0 proved f0 over its own rows, a subset of all rows: that verifies it)
    float ffin = p.fbuf ? p.ffinal[t] : 0.0f;
    for (int sl = 1; sl < nsl && p.fbuf; sl++) ffin = fmaxf(ffin, p.ffinal[(int64_t)sl * p.own_stride + t]);
    const bool verify = p.fbuf && p.f0[t] > -__builtin_inff() && !(ffin > p.f0[t]);
    if (verify) {
        const float f0 = p.f0[t];
        int c_ok = 0;
#pragma unroll
        for (int s = 0; s < EPT; s++) c_ok += ev[s] && fkey_inv(ekey[s]) - mg >= f0;
        if (c_ok) atomicAdd(&s_misc[2], c_ok);
    }
    // Pruning by the approximate scores (round 5): only a row whose approximate score reaches (kth-th best approximate score of the
    // lists) - margin can be among the kth best in exact arithmetic -- the sweep's own rule, applied once more to what the lists hold
    // at the end (thresholds frozen for the foreign side, never tightened for a short own part: ~270 entries at C4, ~110 of which
    // pass) -- so the exact distances, whose gathers are this kernel's time, and the sort are those of the survivors.  The kth-th
    // largest key's 16 leading bits (a lower bound, as in compact_query) by two 256-bin histograms in LDS.
    int *s_h = reinterpret_cast<int *>(s_e);  // 256 bins (s_e is written later)
    float thr = -__builtin_inff();
    if (n_all >= p.kth && n_all > p.kth + 8) {  // uniform
        uint32_t prefix = 0;
        int need = p.kth;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            s_h[tid] = 0;  // kBlock == 256 bins
            __syncthreads();
            const int sh = pass == 0 ? 24 : 16;
#pragma unroll
            for (int s = 0; s < EPT; s++)
                if (ev[s] && (pass == 0 || (ekey[s] >> 24) == (prefix >> 24))) atomicAdd(&s_h[(ekey[s] >> sh) & 255u], 1);
            __syncthreads();
            if (tid < 64) {  // bins in DESCENDING order: position 4 tid + j is bin 255 - (4 tid + j)
                int c[4], tot = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    c[j] = s_h[255 - (4 * tid + j)];
                    tot += c[j];
                }
                int incl = tot;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (tid >= o) incl += v;
                }
                int acc = incl - tot;
                if (acc < need && incl >= need) {  // exactly one lane
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (acc < need && acc + c[j] >= need) {
                            s_misc[3] = 255 - (4 * tid + j);
                            s_misc[4] = need - acc;
                        }
                        acc += c[j];
                    }
                }
            }
            __syncthreads();
            prefix |= (uint32_t)s_misc[3] << sh;
            need = s_misc[4];
            __syncthreads();
        }
        thr = fkey_inv(prefix) - mg;
    } else {
        __syncthreads();
    }
    // the survivors' row ids, packed (any order)
#pragma unroll
    for (int s = 0; s < EPT; s++)
        if (ev[s] && fkey_inv(ekey[s]) >= thr) s_i[atomicAdd(&s_misc[5], 1)] = (int)eidx[s];
    __syncthreads();
    const int n = s_misc[5];
    // exact distances, two candidates per 16-lane group and step (their gathers in flight together: the loop is latency-bound)
    for (int c0 = gib; c0 < n; c0 += 2 * kGroupsPerBlock) {
        const int c1 = c0 + kGroupsPerBlock;
        const bool two = c1 < n;
        const int64_t i0 = s_i[c0], i1 = s_i[two ? c1 : c0];
        float *row0 = sx + (size_t)gib * d, *row1 = sx + (size_t)(kGroupsPerBlock + gib) * d;
        gather_row(row0, p.X, p.Xb, i0, d, lane);
        gather_row(row1, p.X, p.Xb, i1, d, lane);
        __builtin_amdgcn_wave_barrier();
        float r[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const float *row = u ? row1 : row0;
            const int64_t i = u ? i1 : i0;
            if (p.metric == GORSE_METRIC_EUCLIDEAN || p.metric == kMetricEuclidBf16) {
                r[u] = euclid_any_lds(p.metric, sq, row, vs, lane);
            } else {
                const float ab = dot512_lds(sq, row, vs, lane);
                if (p.metric == GORSE_METRIC_NEG_DOT)
                    r[u] = -ab;
                else
                    r[u] = 1.0f - ab / (sqrtf(qq) * sqrtf(p.norm2[i]));
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            s_e[c0] = r[0];
            if (two) s_e[c1] = r[1];
        }
    }
    __syncthreads();
    // Order the admissible candidates by exact distance: a bitonic sort of (order-preserving key of the distance, index) in
    // LDS -- n = ~240 at C4, where ranking every candidate against every other one was two thirds of this kernel's time.
    // Equal distances are adjacent afterwards; one inside the top k + 1 hands the query to the tie path, so the order
    // among equals never shows.  The query itself, a NaN distance and the padding up to the power of two sort last.
    uint32_t *s_key = reinterpret_cast<uint32_t *>(s_e);
    int P = 2;
    while (P < n) P <<= 1;
    {
        uint32_t kreg[kCapT / kBlock];
        int ireg[kCapT / kBlock];
#pragma unroll
        for (int s = 0; s < kCapT / kBlock; s++) {
            const int c = tid + s * kBlock;
            kreg[s] = 0xffffffffu;
            ireg[s] = -1;
            if (c < n && s_i[c] != self) {
                const float e = s_e[c];
                if (e != e) {
                    s_misc[1] = 1;
                } else {
                    kreg[s] = gorse::rank::dist_key(e);
                    ireg[s] = s_i[c] | (gorse::rank::dist_is_negative_zero(e) ? (int)0x80000000 : 0);  // bit 31: the distance is -0
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < kCapT / kBlock; s++) {
            const int c = tid + s * kBlock;
            if (c < P) {
                s_key[c] = kreg[s];
                s_i[c] = ireg[s];
            }
        }
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int x = tid; x < P / 2; x += kBlock) {
                const int lo = 2 * x - (x & (stride - 1));  // the x-th pair of this step: (lo, lo + stride)
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t a = s_key[lo], b = s_key[hi];
                if ((a > b) == up) {
                    s_key[lo] = b;
                    s_key[hi] = a;
                    const int ia = s_i[lo];
                    s_i[lo] = s_i[hi];
                    s_i[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    int admissible = 0;  // the keys below the sentinel, counted by everyone (a binary search over the sorted keys)
    {
        int lo = 0, hi = P;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_key[mid] != 0xffffffffu) lo = mid + 1;
            else hi = mid;
        }
        admissible = lo;
    }
    const int top = admissible < k ? admissible : k;
    // ties that reach the top k + 1; the non-positive distances among the top k (prune0: they are the first ones)
    for (int r = tid; r <= k && r < admissible; r += kBlock) {
        const uint32_t a = s_key[r];
        if ((r > 0 && s_key[r - 1] == a) || (r + 1 < admissible && s_key[r + 1] == a)) s_misc[1] = 1;
        if (r < k && p.prune0 && gorse::rank::dist_key_nonpositive(a)) atomicAdd(&s_misc[0], 1);
    }
    __syncthreads();
    if (verify && s_misc[2] < p.kth) {  // the warm start was too high: swept again from -inf with the tie queries
        if (tid == 0) {
            p.cflag[t] = 2;
            atomicAdd(p.sym_stats + 1, 1ull);
        }
        return;
    }
    if (s_misc[1] || top < p.expect) {  // ties, NaN, or a list that cannot hold the answer: path A
        if (tid == 0) p.cflag[t] = 1;
        return;
    }
    const int dropped = s_misc[0];
    for (int r = dropped + tid; r < top; r += kBlock) {
        const uint32_t a = s_key[r];
        const int iv = s_i[r];
        p.out_idx[t * k + r - dropped] = iv & 0x7fffffff;
        p.out_dist[t * k + r - dropped] = gorse::rank::dist_from_key(a, iv < 0);
    }
    const int cnt = top - dropped;
    for (int r = cnt + tid; r < k; r += kBlock) {
        p.out_idx[t * k + r] = -1;
        p.out_dist[t * k + r] = __builtin_inff();
    }
    if (tid == 0) p.out_cnt[t] = cnt;
}


// ---- tie replay: the reference's heap history of one query, rebuilt from list + history ------------------------
// The rescoring kernel hands over queries whose top k+1 exact distances are not all distinct: there the reference's
// answer depends on the ARRANGEMENT of its heap array, i.e. on everything Bruteforce pushed (bruteforce.go:46-53:
// Push, then Pop when the queue holds more than k).  A history sweep (HIST) has recorded, for such a query, every
// vector whose approximate score reached the filter threshold of its time.  A vector that was NOT recorded had
// approx < threshold = (K-th best approx among earlier vectors) - 2 delta, so at least k EARLIER vectors are strictly
// closer in exact arithmetic: the reference pushed it to the root and popped it straight away.  Such a push/pop pair
// (operator T below) depends only on the heap state, not on the vector -- it is the identity unless equal distances
// sit on the path it touches, and then it permutes them with a short period (<= ~log2 k).  So the replay is: exact
// distances for the recorded vectors (reference arithmetic), sort by index, literal container/heap Push/Pop for each
// (goheap.hpp), and T^gap for every gap of unrecorded vectors in between, T^gap evaluated by cycle detection.
// Queries with a NaN, an overflowed history or an undetected cycle are flagged 2 and go to path A.
struct ReplayParams {
    const float *X;
    const uint16_t *Xb;    // bf16 index: the rows as given (see RescoreParams), else null
    const float *norm2;
    const float *Qf;       // by-vector queries: fp32 rows of the chunk (row pos[t]), or null
    const float *qn2;      // query norms of the chunk (row pos[t]), cosine only
    const int64_t *self;   // stored-vector id of the query, or -1 for by-vector queries
    const int32_t *pos;    // row of the query in the chunk's result arrays
    const uint2 *cbuf;
    const int32_t *ccnt;
    const uint2 *hbuf;
    const int32_t *hcnt;
    const float *fslice;   // nslices x nq: the threshold every slice of the history sweep ended with (-inf: none)
    int nslices;           // cbuf / ccnt / hbuf / hcnt / cflag / fslice hold nslices x nq entries, slice-major
    unsigned long long *prof;  // probe (variant bit 4): replay counters, see gorse_hip_test_get_sweep_profile; else null
    int64_t nq;
    uint8_t *cflag;        // in: flags of the history sweep (non-zero: undecidable); out [0, nq): 2 = undecided here
    int64_t N;
    int d, metric, k, prune0;
    int32_t *out_idx;
    float *out_dist;
    int32_t *out_cnt;
    int32_t *sidx;         // per query kReplayCap recorded vector ids, ascending (stage 1 -> stage 2)
    float *sdst;           // their exact distances
    int32_t *scount;       // how many, or -1 when a distance is NaN
};

// stage 1 of the tie path: exact distances (the reference's arithmetic) of one query's recorded vectors, sorted by
// index; one workgroup per query, everything parallel.  Output: sidx / sdst (kReplayCap slots per query), scount.
__global__ __launch_bounds__(kBlock) void topk_tie_sort_kernel(ReplayParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int d = p.d;
    int *s_idx = reinterpret_cast<int *>(smem_f);           // kReplayCap
    float *s_dst = smem_f + kReplayCap;                      // kReplayCap
    float *sq = s_dst + kReplayCap;                          // d
    float *sx = sq + d;                                      // kGroupsPerBlock * d
    int *s_misc = reinterpret_cast<int *>(sx + (size_t)kGroupsPerBlock * d);  // [0] NaN seen, [1] recorded rows kept by the join
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (kGroup - 1), gib = tid / kGroup;
    // Join of the row slices.  Slice s swept its rows from a cold threshold, so its list + history hold every row of the slice
    // whose approximate score reached (K-th best approximate score among the slice's EARLIER rows) - 2 delta -- a superset
    // of what an unsliced sweep records there, and far too many.  A row of slice s is needed only if its score reaches
    // F_s = max over the earlier slices u < s of f_u, f_u = the threshold slice u ended with = (a lower bound of the K-th
    // best approximate score of slice u's rows) - 2 delta: a row below F_s has at least K earlier rows whose approximate
    // scores exceed its own by more than 2 delta, i.e. K earlier rows that are strictly closer in exact arithmetic, which
    // is the replay's condition for treating it as one of the unrecorded rows of a gap (see ReplayParams).
    int *s_n = s_misc + 1;
    bool flagged = false;
    for (int sl = 0; sl < p.nslices; sl++) flagged |= p.cflag[(int64_t)sl * p.nq + t] != 0;
    if (flagged) {  // some slice could not hold this query (overflow): path A
        if (tid == 0) p.cflag[t] = 1;
        return;
    }
    const int64_t self = p.self[t];
    const int64_t row = p.pos[t];
    const float *qrow = p.Qf ? p.Qf + row * d : p.X + self * d;
    for (int e = tid; e < d; e += kBlock) sq[e] = qrow[e];
    if (tid == 0) s_misc[0] = 0, s_n[0] = 0;
    __syncthreads();
    float fprev = -__builtin_inff();
    for (int sl = 0; sl < p.nslices; sl++) {
        const int64_t qs = (int64_t)sl * p.nq + t;
        const int n1 = p.ccnt[qs], n2 = p.hcnt[qs];
        const uint2 *cb = p.cbuf + qs * kCap, *hb = p.hbuf + qs * kHistCap;
        for (int c = tid; c < n1 + n2; c += kBlock) {
            const uint2 ent = c < n1 ? cb[c] : hb[c - n1];
            if (fkey_inv(ent.x) >= fprev) {
                const int at = atomicAdd(s_n, 1);
                if (at < kReplayCap) s_idx[at] = (int)ent.y;
            }
        }
        fprev = fmaxf(fprev, p.fslice[qs]);
    }
    __syncthreads();
    const int n = s_n[0];
    if (n > kReplayCap) {  // more recorded rows than the replay sorts: path A
        if (tid == 0) p.cflag[t] = 1;
        return;
    }
    __syncthreads();
    const VecShape vs(d);
    const float qq = p.metric == GORSE_METRIC_COSINE ? p.qn2[row] : 0.0f;
    for (int c = gib; c < n; c += kGroupsPerBlock) {  // exact distances, as topk_rescore_kernel
        const int64_t i = s_idx[c];
        float *xr = sx + (size_t)gib * d;
        gather_row(xr, p.X, p.Xb, i, d, lane);
        __builtin_amdgcn_wave_barrier();
        float r;
        if (p.metric == GORSE_METRIC_EUCLIDEAN || p.metric == kMetricEuclidBf16) {
            r = euclid_any_lds(p.metric, sq, xr, vs, lane);
        } else {
            const float ab = dot512_lds(sq, xr, vs, lane);
            if (p.metric == GORSE_METRIC_NEG_DOT)
                r = -ab;
            else
                r = 1.0f - ab / (sqrtf(qq) * sqrtf(p.norm2[i]));
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            s_dst[c] = r;
            if (r != r) s_misc[0] = 1;
        }
    }
    int P = 2;
    while (P < n) P <<= 1;
    for (int c = n + tid; c < P; c += kBlock) {
        s_idx[c] = 0x7fffffff;
        s_dst[c] = 0.0f;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)  // bitonic sort by index
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = tid; e < P / 2; e += kBlock) {
                const int a = (e / stride) * 2 * stride + (e % stride), b = a + stride;
                const bool up = (a & size) == 0;
                const int ia = s_idx[a], ib = s_idx[b];
                if ((ia > ib) == up) {
                    s_idx[a] = ib;
                    s_idx[b] = ia;
                    const float da = s_dst[a];
                    s_dst[a] = s_dst[b];
                    s_dst[b] = da;
                }
            }
            __syncthreads();
        }
    for (int c = tid; c < n; c += kBlock) {
        p.sidx[t * kReplayCap + c] = s_idx[c];
        p.sdst[t * kReplayCap + c] = s_dst[c];
    }
    if (tid == 0) p.scount[t] = s_misc[0] ? -1 : n;  // -1: a NaN distance (the reference panics on it): path A
}

// The reference's heap as a wave-uniform register structure: entry j lives in lane j & 63 of register j >> 6
// (capacity 256 = kMaxKth), every index and weight the sift rules look at is a scalar (v_readlane / v_writelane), so a
// Push or Pop costs no LDS or memory round trip.  The sift loops are the hole form of container/heap's swap loops
// (goheap.hpp): the moving element is compared with the same neighbours and ends in the same slot, and every other
// element is moved exactly as the swaps would move it.
template <bool DESC>
struct RegHeap {
    int w[4], v[4];  // weights (float bits) and values
    int n;
    __device__ __forceinline__ RegHeap() : n(0) {
#pragma unroll
        for (int r = 0; r < 4; r++) w[r] = v[r] = 0;
    }
    __device__ __forceinline__ static bool less(float a, float b) { return DESC ? a > b : a < b; }
    __device__ __forceinline__ float getw(int j) const {
        const int l = j & 63;
        int x;
        switch (j >> 6) {
        case 0: x = __builtin_amdgcn_readlane(w[0], l); break;
        case 1: x = __builtin_amdgcn_readlane(w[1], l); break;
        case 2: x = __builtin_amdgcn_readlane(w[2], l); break;
        default: x = __builtin_amdgcn_readlane(w[3], l); break;
        }
        return __int_as_float(x);
    }
    __device__ __forceinline__ int getv(int j) const {
        const int l = j & 63;
        switch (j >> 6) {
        case 0: return __builtin_amdgcn_readlane(v[0], l);
        case 1: return __builtin_amdgcn_readlane(v[1], l);
        case 2: return __builtin_amdgcn_readlane(v[2], l);
        default: return __builtin_amdgcn_readlane(v[3], l);
        }
    }
    __device__ __forceinline__ void set(int j, float wt, int val) {
        const bool me = (int)(threadIdx.x & 63) == (j & 63);
        const int wb = __float_as_int(wt);
        switch (j >> 6) {
        case 0: w[0] = me ? wb : w[0]; v[0] = me ? val : v[0]; break;
        case 1: w[1] = me ? wb : w[1]; v[1] = me ? val : v[1]; break;
        case 2: w[2] = me ? wb : w[2]; v[2] = me ? val : v[2]; break;
        default: w[3] = me ? wb : w[3]; v[3] = me ? val : v[3]; break;
        }
    }
    __device__ __forceinline__ void push(int val, float wt) {  // heap.Push: append, up(n - 1)
        int j = n++;
        while (j > 0) {
            const int i = (j - 1) / 2;
            const float wi = getw(i);
            if (!less(wt, wi)) break;
            set(j, wi, getv(i));
            j = i;
        }
        set(j, wt, val);
    }
    // heap.Pop: swap(0, n - 1), down(0, n - 1), take the last; returns the removed root
    __device__ __forceinline__ void pop(int &val, float &wt) {
        val = getv(0);
        wt = getw(0);
        const int m = --n;  // the last element (index m) moves to the root and sifts down over m elements
        if (m == 0) return;
        const float wl = getw(m);
        const int vl = getv(m);
        int i = 0;
        for (;;) {
            const int j1 = 2 * i + 1;
            if (j1 >= m) break;
            int j = j1;
            float wj = getw(j1);
            if (j1 + 1 < m) {
                const float w2 = getw(j1 + 1);
                if (less(w2, wj)) {
                    j = j1 + 1;
                    wj = w2;
                }
            }
            if (!less(wj, wl)) break;
            set(i, wj, getv(j));
            i = j;
        }
        set(i, wl, vl);
    }
    // every lane: do the first n values equal those of `o`?
    __device__ __forceinline__ bool same_values(const RegHeap &o, int lane) const {
        bool diff = false;
#pragma unroll
        for (int r = 0; r < 4; r++) diff |= (lane + 64 * r < n) && v[r] != o.v[r];
        return __builtin_amdgcn_ballot_w64(diff) == 0;
    }
};

#ifdef GORSE_PROBE  // the round-2 replay (a wave per query): kept for `make probe-lib` only, the library ships the lane kernel
// stage 2 of the tie path: one WAVE per query replays the reference's heap over the sorted entries (see the comment
// above ReplayParams); four queries per workgroup, no LDS.
__global__ __launch_bounds__(256) void topk_tie_replay_kernel(ReplayParams p, int64_t nq, int g_replay_literal) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= nq) return;
    if (p.cflag[t]) return;
    const int n = p.scount[t];
    const int k = p.k;
    const int64_t self = p.self[t];
    const int64_t row = p.pos[t];
    bool undecided = n < 0;
    const float kInf = __builtin_inff();
    const unsigned long long t_begin = p.prof ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long n_push = 0, n_tpow = 0, n_slow_tpow = 0, n_T = 0;
    RegHeap<true> mx;
    auto apply_T = [&]() {  // a push that goes to the root and is popped at once
        int dv;
        float dw;
        mx.push(-1, kInf);
        mx.pop(dv, dw);
    };
    // T leaves the heap array as it is unless equal weights sit where its push and pop look.  With the heap full (n = k),
    // push(+inf) climbs from slot n to the root and shifts the path's elements down one level; pop then sends the element
    // E that came to rest in slot n (the old occupant of slot p1 = parent(n)) back down from the root, and every level
    // restores its old occupant iff (a) the path's child wins the comparison of the two children -- certain when it is the
    // LEFT child (container/heap prefers the left on a tie), and when it is the right child only if the old parent is
    // strictly greater than the left child -- and (b) the old parent is strictly greater than E.  E then stops in p1, whose
    // remaining child cannot beat it.  A dozen scalar comparisons instead of two sift passes and a snapshot compare
    // (tests/test_replay_claim_cpu.py checks the criterion against the literal T).
    auto t_is_identity = [&]() -> bool {
        const int n = mx.n;
        if (n < 1) return false;
        int child = (n - 1) / 2;  // p1
        const float e = mx.getw(child);
        if (!(e < kInf)) return false;
        while (child > 0) {
            const int parent = (child - 1) / 2;
            const float wp = mx.getw(parent);
            if (!(wp > e) || !(wp < kInf)) return false;
            if ((child & 1) == 0 && !(wp > mx.getw(child - 1))) return false;  // right child: the left sibling must lose
            child = parent;
        }
        return true;
    };
    auto t_pow = [&](int64_t gap) {
        n_tpow++;
        if (!(g_replay_literal) && t_is_identity()) return;
        n_slow_tpow++;
        int64_t steps = 0;
        while (steps < gap && steps < 16) {  // fixpoint (the usual case: no equal distances on the path) or pre-period
            const RegHeap<true> snap = mx;
            apply_T();
            steps++;
            n_T++;
            if (mx.same_values(snap, lane)) return;
        }
        if (steps == gap) return;
        const RegHeap<true> snap = mx;  // inside the cycle now (or not: then the period search fails, query flagged)
        int64_t period = 0;
        bool closed = false;
        while (period < 64) {
            apply_T();
            period++;
            steps++;
            n_T++;
            if (steps == gap) return;
            if (mx.same_values(snap, lane)) {
                closed = true;
                break;
            }
        }
        if (!closed) {
            undecided = true;
            return;
        }
        const int64_t rem = (gap - steps) % period;
        for (int64_t r = 0; r < rem; r++) apply_T();
    };
    const int32_t *sidx = p.sidx + t * kReplayCap;
    const float *sdst = p.sdst + t * kReplayCap;
    int64_t prev = -1;
    int64_t pend = 0;  // T applications owed: unrecorded vectors + recorded ones that are strictly worse than the root
    for (int e0 = 0; e0 < n && !undecided; e0 += 64) {
        const int bi = e0 + lane < n ? sidx[e0 + lane] : 0;
        const int bd = e0 + lane < n ? __float_as_int(sdst[e0 + lane]) : 0;
        const int cntb = n - e0 < 64 ? n - e0 : 64;
        for (int j = 0; j < cntb && !undecided; j++) {
            const int64_t i = __builtin_amdgcn_readlane(bi, j);
            const float dd = __int_as_float(__builtin_amdgcn_readlane(bd, j));
            if (i == self) continue;
            if (i == prev) {  // cannot happen: every recorded vector lives in exactly one place
                undecided = true;
                break;
            }
            int64_t gap = i - prev - 1;
            if (self > prev && self < i) gap--;
            if (gap > 0 && mx.n < k) {  // unrecorded vectors before the heap is full would have been accepted
                undecided = true;
                break;
            }
            pend += gap;
            prev = i;
            if (mx.n == k && dd > mx.getw(0)) {  // goes to the root and is popped at once: one more T
                pend++;
                continue;
            }
            if (pend > 0) {
                t_pow(pend);
                pend = 0;
                if (undecided) break;
            }
            n_push++;
            mx.push((int32_t)i, dd);
            if (mx.n > k) {
                int dv;
                float dw;
                mx.pop(dv, dw);
            }
        }
    }
    if (!undecided) {
        int64_t gap = p.N - 1 - prev;
        if (self > prev) gap--;
        if (gap > 0 && mx.n < k) undecided = true;
        pend += gap;
        if (!undecided && pend > 0) t_pow(pend);
    }
    if (p.prof && lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_begin;
        atomicAdd(p.prof + 0, dt);
        atomicMax(p.prof + 1, dt);
        atomicAdd(p.prof + 2, (unsigned long long)(n > 0 ? n : 0));
        atomicAdd(p.prof + 3, 1ull);
        atomicAdd(p.prof + 4, n_push);
        atomicAdd(p.prof + 5, n_tpow);
        atomicAdd(p.prof + 6, n_slow_tpow);
        atomicAdd(p.prof + 7, n_T);
        atomicMax(p.prof + 8, n_T);
        atomicMax(p.prof + 9, (unsigned long long)(n > 0 ? n : 0));
        atomicAdd(p.prof + 10, undecided ? 1ull : 0ull);
    }
    if (undecided) {
        if (lane == 0) p.cflag[t] = 2;
        return;
    }
    RegHeap<false> mn;  // Reverse(): re-push in array order (pq.go:121-128)
    const int hn = mx.n;
    for (int e = 0; e < hn; e++) mn.push(mx.getv(e), mx.getw(e));
    int cnt = 0;
    while (mn.n > 0) {
        int v;
        float w;
        mn.pop(v, w);
        if (!p.prune0 || w > 0) {
            if (lane == 0) {
                p.out_idx[row * k + cnt] = v;
                p.out_dist[row * k + cnt] = w;
            }
            cnt++;
        }
    }
    if (lane == 0) p.out_cnt[row] = cnt;
    for (int e = cnt + lane; e < k; e += 64) {
        p.out_idx[row * k + e] = -1;
        p.out_dist[row * k + e] = kInf;
    }
}
#endif  // GORSE_PROBE


// ---- the same replay with one LANE per query ----------------------------------------------------------------------
// The replay is scalar work: the wave-per-query kernel above issues ~1.8 M cycles of readlane / writelane traffic per
// query (18 ms for the 9,934 tie queries of C4, profiles/r03_r_probe_c4_replay.txt).  Here every lane replays its own
// query and keeps its heap in its own LDS column (slot j of lane q at word j * QPW + q: bank q whatever the slot, so the
// lanes never conflict, and no lane ever reads another's column: no barriers).  The ancestors of a push and the path of
// the identity test are known before the first comparison, so their weights are read together -- one LDS round trip
// instead of one per level; only the pop walks level by level.
constexpr int kLaneDepth = 8;   // ancestors of slot 256
constexpr int kLaneLog = 24;    // slots one T can write (<= 9 by its push, <= 9 by its pop)
template <bool DESC, int stride>
struct LaneHeap {
    int *W, *V;  // this lane's column: slot j at [j * stride]
    int n;
    __device__ __forceinline__ static bool less(float a, float b) { return DESC ? a > b : a < b; }
    __device__ __forceinline__ float getw(int j) const { return __int_as_float(W[j * stride]); }
    __device__ __forceinline__ int getv(int j) const { return V[j * stride]; }
    // log (probe of change): slot and old value of everything written, when lg != nullptr
    int *lg;
    int nlog;
    __device__ __forceinline__ void set(int j, float wt, int val) {
        if (lg) {
            if (nlog < kLaneLog) {
                lg[(2 * nlog) * stride] = j;
                lg[(2 * nlog + 1) * stride] = V[j * stride];
            }
            nlog++;
        }
        W[j * stride] = __float_as_int(wt);
        V[j * stride] = val;
    }
    __device__ __forceinline__ void push(int val, float wt) {  // heap.Push: append, up(n - 1)
        int j = n++;
        int anc[kLaneDepth], av[kLaneDepth];
        float aw[kLaneDepth];
        int a = j;
#pragma unroll
        for (int l = 0; l < kLaneDepth; l++) {
            const bool ok = a > 0;
            a = ok ? (a - 1) / 2 : 0;
            anc[l] = ok ? a : -1;
            aw[l] = getw(a);
            av[l] = getv(a);
        }
#pragma unroll
        for (int l = 0; l < kLaneDepth; l++) {
            if (anc[l] < 0 || !less(wt, aw[l])) break;
            set(j, aw[l], av[l]);
            j = anc[l];
        }
        set(j, wt, val);
    }
    __device__ __forceinline__ void pop(int &val, float &wt) {  // heap.Pop: swap(0, n - 1), down(0, n - 1), take the last
        val = getv(0);
        wt = getw(0);
        const int m = --n;
        if (m == 0) return;
        const float wl = getw(m);
        const int vl = getv(m);
        int i = 0;
        for (;;) {
            const int j1 = 2 * i + 1;
            if (j1 >= m) break;
            const bool two = j1 + 1 < m;
            const int j2 = two ? j1 + 1 : j1;
            const float w1 = getw(j1), w2 = getw(j2);
            const int v1 = getv(j1), v2 = getv(j2);
            int j = j1, vj = v1;
            float wj = w1;
            if (two && less(w2, wj)) {
                j = j2;
                wj = w2;
                vj = v2;
            }
            if (!less(wj, wl)) break;
            set(i, wj, vj);
            i = j;
        }
        set(i, wl, vl);
    }
};

template <int QPW>  // queries (= active lanes) per workgroup: a constant so that a slot's address is a shift
__global__ __launch_bounds__(64) void topk_tie_replay_lane_kernel(ReplayParams p, int64_t nq, int slots) {
    constexpr int qpw = QPW;
    extern __shared__ int smem_i[];
    const int lane = threadIdx.x;
    const int64_t t = (int64_t)blockIdx.x * qpw + lane;
    if (lane >= qpw || t >= nq) return;
    if (p.cflag[t]) return;
    int *W = smem_i + lane, *V = W + slots * qpw, *LG = V + slots * qpw, *SN = LG + 2 * kLaneLog * qpw;  // SN: a snapshot of V
    const int n = p.scount[t];
    const int k = p.k;
    const int64_t self = p.self[t];
    const int64_t row = p.pos[t];
    bool undecided = n < 0;
    const float kInf = __builtin_inff();
    const unsigned long long t_begin = p.prof ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long n_push = 0, n_tpow = 0, n_slow_tpow = 0, n_T = 0;
    LaneHeap<true, QPW> mx;
    mx.W = W, mx.V = V, mx.n = 0, mx.lg = nullptr, mx.nlog = 0;
    // topk_tie_replay_kernel's t_is_identity, the path's weights read before the first comparison
    auto t_is_identity = [&]() -> bool {
        const int hn = mx.n;
        if (hn < 1) return false;
        int child = (hn - 1) / 2;  // p1
        const float e = mx.getw(child);
        float wp[kLaneDepth], ws[kLaneDepth];
        bool has[kLaneDepth], right[kLaneDepth];
        int c = child;
#pragma unroll
        for (int l = 0; l < kLaneDepth; l++) {
            has[l] = c > 0;
            const int parent = has[l] ? (c - 1) / 2 : 0;
            right[l] = has[l] && (c & 1) == 0;
            wp[l] = mx.getw(parent);
            ws[l] = mx.getw(right[l] ? c - 1 : 0);
            c = parent;
        }
        if (!(e < kInf)) return false;
#pragma unroll
        for (int l = 0; l < kLaneDepth; l++) {
            if (!has[l]) break;
            if (!(wp[l] > e) || !(wp[l] < kInf)) return false;
            if (right[l] && !(wp[l] > ws[l])) return false;
        }
        return true;
    };
    auto apply_T = [&]() {
        int dv;
        float dw;
        mx.push(-1, kInf);
        mx.pop(dv, dw);
    };
    // T^gap when T is not the identity: literal applications until nothing changes (the change is read off the log of
    // what T wrote), then topk_tie_replay_kernel's search for the period against a snapshot of the values
    auto t_pow = [&](int64_t gap) {
        n_tpow++;
        if (t_is_identity()) return;
        n_slow_tpow++;
        int64_t steps = 0;
        while (steps < gap && steps < 16) {
            const int before = mx.n;
            mx.lg = LG;
            mx.nlog = 0;
            apply_T();
            mx.lg = nullptr;
            steps++;
            n_T++;
            bool same = mx.nlog <= kLaneLog;
            for (int r = 0; r < mx.nlog && same; r++) {
                const int sl = LG[(2 * r) * qpw];
                if (sl < before && V[sl * qpw] != LG[(2 * r + 1) * qpw]) same = false;
            }
            if (same) return;
        }
        if (steps == gap) return;
        const int hn = mx.n;
        for (int sl = 0; sl < hn; sl++) SN[sl * qpw] = V[sl * qpw];  // inside the cycle now (or not: then the query is flagged)
        int64_t period = 0;
        bool closed = false;
        while (period < 64) {
            apply_T();
            period++;
            steps++;
            n_T++;
            if (steps == gap) return;
            bool same = true;
            for (int sl = 0; sl < hn && same; sl++) same = SN[sl * qpw] == V[sl * qpw];
            if (same) {
                closed = true;
                break;
            }
        }
        if (!closed) {
            undecided = true;
            return;
        }
        const int64_t rem = (gap - steps) % period;
        for (int64_t r = 0; r < rem; r++) apply_T();
    };
    const int32_t *sidx = p.sidx + t * kReplayCap;
    const float *sdst = p.sdst + t * kReplayCap;
    int64_t prev = -1;
    int64_t pend = 0;
    // Lanes run in step, so an entry that only counts (strictly worse than a full heap's root: the reference pushes it to the
    // root and pops it at once) must not cost its lane a turn of the heap work the others do: every lane first walks over
    // such entries (inner loop), then the lanes that stopped at a real push do it together.
    float rootw = kInf;  // W[0] of a non-empty heap
    int e0 = 0;
    int nxt_i = n > 0 ? sidx[0] : 0;  // one entry ahead (16-byte groups two ahead were slower: profiles/r03_v_kernel_stats_default.txt)
    float nxt_d = n > 0 ? sdst[0] : 0.0f;
    while (e0 < n && !undecided) {
        int64_t i = 0;
        float dd = 0.0f;
        bool have = false;
        while (e0 < n) {
            i = nxt_i;
            dd = nxt_d;
            e0++;
            if (e0 < n) {
                nxt_i = sidx[e0];
                nxt_d = sdst[e0];
            }
            if (i == self) continue;
            if (i == prev) {
                undecided = true;
                break;
            }
            int64_t gap = i - prev - 1;
            if (self > prev && self < i) gap--;
            if (gap > 0 && mx.n < k) {
                undecided = true;
                break;
            }
            pend += gap;
            prev = i;
            if (mx.n == k && dd > rootw) {
                pend++;
                continue;
            }
            have = true;
            break;
        }
        if (!have || undecided) break;
        if (pend > 0) {
            t_pow(pend);
            pend = 0;
            if (undecided) break;
        }
        n_push++;
        mx.push((int32_t)i, dd);
        if (mx.n > k) {
            int dv;
            float dw;
            mx.pop(dv, dw);
        }
        rootw = mx.getw(0);
    }
    if (!undecided) {
        int64_t gap = p.N - 1 - prev;
        if (self > prev) gap--;
        if (gap > 0 && mx.n < k) undecided = true;
        pend += gap;
        if (!undecided && pend > 0) t_pow(pend);
    }
    if (p.prof) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_begin;
        atomicAdd(p.prof + 0, dt);
        atomicMax(p.prof + 1, dt);
        atomicAdd(p.prof + 2, (unsigned long long)(n > 0 ? n : 0));
        atomicAdd(p.prof + 3, 1ull);
        atomicAdd(p.prof + 4, n_push);
        atomicAdd(p.prof + 5, n_tpow);
        atomicAdd(p.prof + 6, n_slow_tpow);
        atomicAdd(p.prof + 7, n_T);
        atomicMax(p.prof + 8, n_T);
        atomicMax(p.prof + 9, (unsigned long long)(n > 0 ? n : 0));
        atomicAdd(p.prof + 10, undecided ? 1ull : 0ull);
    }
    if (undecided) {
        p.cflag[t] = 2;
        return;
    }
    // Reverse(): re-push in array order (pq.go:121-128) -- in place: the min-heap of the first e entries lives in the slots
    // the max-heap's first e entries have been read from
    const int hn = mx.n;
    LaneHeap<false, QPW> mn;
    mn.W = W, mn.V = V, mn.n = 0, mn.lg = nullptr, mn.nlog = 0;
    for (int e = 0; e < hn; e++) {
        const int val = V[e * qpw];
        const float wt = __int_as_float(W[e * qpw]);
        mn.push(val, wt);
    }
    int cnt = 0;
    while (mn.n > 0) {
        int v;
        float w;
        mn.pop(v, w);
        if (!p.prune0 || w > 0) {
            p.out_idx[row * k + cnt] = v;
            p.out_dist[row * k + cnt] = w;
            cnt++;
        }
    }
    p.out_cnt[row] = cnt;
    for (int e = cnt; e < k; e++) {
        p.out_idx[row * k + e] = -1;
        p.out_dist[row * k + e] = kInf;
    }
}

// rows pos[t] of a bf16 operand matrix / a float vector -> compact arrays of the flagged queries
__global__ void gather_pos_kernel(const uint16_t *__restrict__ op, const float *__restrict__ margin,
                                  const int32_t *__restrict__ pos, int kpad, uint16_t *__restrict__ op_out,
                                  float *__restrict__ margin_out) {
    const int64_t t = blockIdx.x;
    const int64_t src = pos[t];
    const uint4 *s = reinterpret_cast<const uint4 *>(op + src * kpad);
    uint4 *o = reinterpret_cast<uint4 *>(op_out + t * kpad);
    for (int e = threadIdx.x; e < kpad / 8; e += blockDim.x) o[e] = s[e];
    if (threadIdx.x == 0) margin_out[t] = margin[src];
}

// ---- operand construction ------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t bf16_rne(float x) {
    uint32_t b = __float_as_uint(x);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}
__device__ __forceinline__ float bf16_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// fp32 rows -> [hi | lo | hi] (role 0, candidates) or [hi | hi | lo] (role 1, queries), zero padded to kpad:
// sum over the operand = hi.hi + lo.hi + hi.lo, the three leading terms of the split product.
__global__ void split_f32_kernel(const float *__restrict__ X, int64_t n, int d, int kpad, int role,
                                 uint16_t *__restrict__ out) {
    const int64_t total = n * kpad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / kpad;
        const int e = (int)(t % kpad);
        uint16_t v = 0;
        if (e < 3 * d) {
            const int part = e / d, j = e % d;
            const float x = X[row * d + j];
            const uint16_t hi = bf16_rne(x);
            const bool want_lo = role == 0 ? part == 1 : part == 2;
            v = want_lo ? bf16_rne(x - bf16_f32(hi)) : hi;
        }
        out[t] = v;
    }
}
// bf16 rows (or fp32 rows that hold bf16 values exactly) -> zero padded bf16 operand rows
__global__ void pad_bf16_kernel(const uint16_t *__restrict__ Xb, const float *__restrict__ Xf, int64_t n, int d,
                                int kpad, uint16_t *__restrict__ out) {
    const int64_t total = n * kpad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / kpad;
        const int e = (int)(t % kpad);
        uint16_t v = 0;
        if (e < d) v = Xb ? Xb[row * d + e] : (uint16_t)(__float_as_uint(Xf[row * d + e]) >> 16);
        out[t] = v;
    }
}
__global__ void rscale_kernel(const float *__restrict__ norm2, int64_t n, float *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = 1.0f / sqrtf(norm2[t]);
}
// gather operand rows / norms of stored vectors for an id list
__global__ void gather_queries_kernel(const uint16_t *__restrict__ opB, const float *__restrict__ norm2,
                                      const int64_t *__restrict__ qid, int64_t nq, int kpad,
                                      uint16_t *__restrict__ opQ, float *__restrict__ qn2) {
    const int64_t t = blockIdx.x;
    const int64_t src = qid[t];
    const uint4 *s = reinterpret_cast<const uint4 *>(opB + src * kpad);
    uint4 *o = reinterpret_cast<uint4 *>(opQ + t * kpad);
    for (int e = threadIdx.x; e < kpad / 8; e += blockDim.x) o[e] = s[e];
    if (threadIdx.x == 0) qn2[t] = norm2[src];
}
// Euclidean: score = q.x - |x|^2 / 2 stands for (|q|^2 - d^2) / 2 with d = the reference's floats.Euclidean.  Its error
// against that quantity: the dot product's (coef |q| |x|), the fp32 norm in the bias (d u |x|^2 / 2) and the
// reference's own rounding of the sum of squares and the square root ((d + 8) u (|q| + |x|)^2 / 2); margin = 2 x that.
__global__ void margin_euclid_kernel(const float *__restrict__ qn2, int64_t nq, float coef, float xmax, float du,
                                     float *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nq; t += (int64_t)gridDim.x * blockDim.x) {
        const float qn = sqrtf(qn2[t]);
        const float u = 5.9604645e-8f;  // 2^-24; the + 4u also covers the rounding of the bias addition itself
        const float delta = coef * qn * xmax + 0.5f * (du + 4.0f * u) * xmax * xmax + 0.5f * (du + 8.0f * u) * (qn + xmax) * (qn + xmax);
        out[t] = 2.0f * delta * 1.01f + 1e-30f;
    }
}
__global__ void bias_kernel(const float *__restrict__ norm2, int64_t n, float *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = -0.5f * norm2[t];
}
__global__ void margin_kernel(const float *__restrict__ qn2, int64_t nq, float coef, float other,
                              float *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nq; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = 2.0f * coef * sqrtf(qn2[t]) * other * 1.001f + 1e-30f;
}

const int kSupportedKP[] = {1, 2, 3, 4, 6, 8, 12, 16, 24};

template <int KP, int NCB, int EP, bool HIST, int RB, bool SYM = false>
int32_t launch_sweep_one(gorse_topk *h, const SweepParams &p) {
    constexpr int WV = sweep_waves(HIST, KP);
    constexpr int BQ = 32 * NCB * WV;
    const size_t lds = sweep_lds_bytes(KP, RB, BQ, WV, HIST, SYM);
    unsigned grid = (unsigned)ceil_div(p.nq, BQ);
    if (SYM && p.sym_world > 1) {  // the blocks of this triangle shard
        if ((int64_t)grid <= p.sym_rank) return GORSE_OK;
        grid = (unsigned)ceil_div((int64_t)grid - p.sym_rank, p.sym_world);
    }
    GORSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&topk_sweep_kernel<KP, NCB, EP, HIST, RB, false, SYM>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    topk_sweep_kernel<KP, NCB, EP, HIST, RB, false, SYM>
        <<<dim3(grid, HIST ? (unsigned)std::max(p.nslices, 1) : (SYM ? (unsigned)std::max(p.sym_slices, 1) : 1u)), dim3(WV * 64), lds, h->stream>>>(p);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

#ifdef GORSE_PROBE  // `make probe-lib`: the shipped library carries neither the instrumented twin nor its counters
template <int RB>
int32_t launch_sweep_prof(gorse_topk *h, const SweepParams &p) {  // the instrumented twin of the C4-shaped sweep (probe only)
    constexpr int BQ = 32 * kNcbMain * kWaves;
    const size_t lds = sweep_lds_bytes(8, RB, BQ, kWaves, false);
    GORSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&topk_sweep_kernel<8, kNcbMain, EP_COARSE, false, RB, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    topk_sweep_kernel<8, kNcbMain, EP_COARSE, false, RB, true><<<dim3((unsigned)ceil_div(p.nq, BQ)), dim3(kThreads), lds, h->stream>>>(p);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}
#endif

template <int KP, int NCB, bool HIST, int RB, bool SYM = false>
int32_t launch_sweep_ep(gorse_topk *h, const SweepParams &p) {
    switch (p.ep) {
        case EP_SCALE: return launch_sweep_one<KP, NCB, EP_SCALE, HIST, RB, SYM>(h, p);
        case EP_BIAS: return launch_sweep_one<KP, NCB, EP_BIAS, HIST, RB, SYM>(h, p);
        case EP_COARSE: return launch_sweep_one<KP, NCB, EP_COARSE, HIST, RB, SYM>(h, p);
    }
    return fail(GORSE_ERR_INVALID, "unknown epilogue %d", p.ep);
}

// the operand depths whose main sweep has a symmetric form: 128-row DMA tiles, two column blocks per wave
constexpr bool sweep_sym_kp(int kp) { return kp == 2 || kp == 4 || kp == 8; }

template <int KP, int NCB>
int32_t launch_sweep(gorse_topk *h, const SweepParams &p, bool hist, bool sym = false) {
    if constexpr (sweep_sym_kp(KP) && NCB == kNcbMain && sweep_dma(KP, false, 4, kWavesMain)) {
        if (sym) return launch_sweep_ep<KP, NCB, false, 4, true>(h, p);
    }
    if (sym) return fail(GORSE_ERR_INVALID, "no symmetric sweep for operand depth %d", KP);
    // 128-row tiles (one tile hand-over per four MFMA row blocks) where LDS allows; the history sweep of the few flagged
    // queries keeps the 64-row form
    const bool wide = !hist && KP <= 8 && topk_rows_per_tile() == 128;
#ifdef GORSE_PROBE
    if constexpr (KP == 8 && NCB == kNcbMain) {
        if ((g_topk_variant & 16) && p.ep == EP_COARSE && !hist && p.prof)
            return wide ? launch_sweep_prof<4>(h, p) : launch_sweep_prof<2>(h, p);
    }
#endif
    if (wide) {
        if constexpr (KP <= 8) return launch_sweep_ep<KP, NCB, false, 4>(h, p);
    }
    return hist ? launch_sweep_ep<KP, NCB, true, 2>(h, p) : launch_sweep_ep<KP, NCB, false, 2>(h, p);
}

int32_t dispatch_sweep(gorse_topk *h, const SweepParams &p, bool hist, bool sym = false) {
    switch (h->kp) {
        case 1: return launch_sweep<1, kNcbMain>(h, p, hist);
        case 2: return launch_sweep<2, kNcbMain>(h, p, hist, sym);
        case 3: return launch_sweep<3, kNcbMain>(h, p, hist);
        case 4: return launch_sweep<4, kNcbMain>(h, p, hist, sym);
        case 6: return launch_sweep<6, kNcbMain>(h, p, hist);
        case 8:
            // variant bit 27 (probe): the history sweep of the tie path with ONE column block per wave -- 64 queries per two-wave
            // workgroup instead of 128: twice the workgroups, each with half the acceptances of a cold start
            if (hist && (g_topk_variant & (1 << 27))) return launch_sweep<8, 1>(h, p, true);
            return launch_sweep<8, kNcbMain>(h, p, hist, sym);
        case 12: return launch_sweep<12, 1>(h, p, hist);  // 2 column blocks would spill
        case 16: return launch_sweep<16, 1>(h, p, hist);
        case 24: return launch_sweep<24, 1>(h, p, hist);
    }
    return fail(GORSE_ERR_INVALID, "unsupported operand depth %d", h->kp);
}

// The history sweep's row slices need not start cold (round 6).  A slice that starts at row r_s may drop every row whose approximate
// score stays below (K-th best approximate score among ANY set of rows in front of r_s) - 2 delta: at least K earlier rows are then
// strictly closer in exact arithmetic, and the reference's heap pushed such a row to its root and popped it at once (ReplayParams).
// The main sweep's lists of a flagged query -- its own list and, after a symmetric sweep, its foreign list: ~270 distinct rows with
// their approximate scores, whatever became of the query's thresholds -- are such a set: one wave per flagged query takes the list
// entries in front of every slice and, where there are at least kth of them, leaves a lower bound of their kth-th best score minus
// the query's margin as the slice's starting threshold (the bound as compact_query forms it: the 16 leading key bits).  Slice 0 and
// the slices with fewer than kth list entries in front of them start cold (-inf).  At C4: ~270 entries per query, so from the fourth
// slice of eight on a slice records the few dozen rows that reach its bound instead of the ~710 a cold start accepts.
struct WarmParams {
    const int32_t *pos;  // row of the flagged query in the chunk's arrays
    int64_t m2, N;
    int nsl, tile_rows, kth;
    const uint2 *cbuf;  // the main sweep's own lists (own_slices of them per query, own_stride queries apart) ...
    const int32_t *ccnt;
    int own_slices;
    int64_t own_stride;
    const uint2 *fbuf;  // ... and foreign lists (or null)
    const int32_t *fcnt;
    const float *margin;  // m2: the flagged queries' 2 delta (gather_pos_kernel)
    float *fwarm;         // nsl x m2, slice-major
};
__global__ __launch_bounds__(64) void tie_warm_kernel(WarmParams p) {
    const int64_t t = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t q = p.pos[t];
    constexpr int EPL = kCapT / 64;  // at most kCapT entries are looked at: any subset of the lists is a valid set
    int n_at[kMaxOwnSlices + 2];
    n_at[0] = 0;
#pragma unroll
    for (int sl = 0; sl < kMaxOwnSlices; sl++) {
        int c = sl < p.own_slices ? p.ccnt[(int64_t)sl * p.own_stride + q] : 0;
        c = c < 0 ? 0 : (c > kCap ? kCap : c);
        n_at[sl + 1] = n_at[sl] + c;
    }
    int nf = p.fbuf ? p.fcnt[q] : 0;
    nf = nf < 0 ? 0 : (nf > kCapF ? kCapF : nf);
    const int n_own = n_at[kMaxOwnSlices];
    const int n_all = n_own + nf < kCapT ? n_own + nf : kCapT;
    uint32_t key[EPL];
    int64_t row[EPL];
#pragma unroll
    for (int j = 0; j < EPL; j++) {
        const int c = j * 64 + lane;
        uint2 ent = make_uint2(0u, 0u);
        if (c < n_all) {
            if (c >= n_own) {
                ent = p.fbuf[q * kCapF + (c - n_own)];
            } else {
                int sl = 0;
#pragma unroll
                for (int u = 1; u < kMaxOwnSlices; u++) sl += c >= n_at[u] ? 1 : 0;
                ent = p.cbuf[((int64_t)sl * p.own_stride + q) * kCap + (c - n_at[sl])];
            }
        }
        key[j] = c < n_all ? ent.x : 0u;
        row[j] = c < n_all ? (int64_t)ent.y : p.N;
    }
    const float mg = p.margin[t];
    const int64_t nt_all = (p.N + p.tile_rows - 1) / p.tile_rows;
    for (int s = 0; s < p.nsl; s++) {
        const int64_t r_s = (nt_all * s / p.nsl) * p.tile_rows;  // topk_sweep_kernel: the slice's first tile T0 = NT_all * y / slices
        uint32_t k_in[EPL];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            k_in[j] = row[j] < r_s ? key[j] : 0u;  // key 0: never counted (trial >= 2^16)
            cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(k_in[j] != 0u));
        }
        float f = -__builtin_inff();
        if (s > 0 && cnt >= p.kth) {
            uint32_t prefix = 0;
            for (int b = 31; b >= 16; --b) {
                const uint32_t trial = prefix | (1u << b);
                int c = 0;
#pragma unroll
                for (int j = 0; j < EPL; j++) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(k_in[j] >= trial));
                if (c >= p.kth) prefix = trial;
            }
            f = fkey_inv(prefix) - mg;
        }
        if (lane == 0) p.fwarm[(int64_t)s * p.m2 + t] = f;
    }
}

// SYM: the pilot's thresholds as the main sweep's foreign side uses them.  A query the pilot could not give a threshold (-inf)
// would make every column of every block a foreign candidate: it gets +inf instead -- nothing is collected for it, neither by
// its own workgroup nor by the others -- and the flag that sends it to the tie path's sweep from -inf.  raw = the form the
// RAWF test compares raw scores with (topk_sweep_kernel's raw_threshold), or null.
__global__ void sym_thresholds_kernel(float *__restrict__ f0, float *__restrict__ raw, uint8_t *__restrict__ cflag, int64_t n,
                                      float rs_min, float rs_max, unsigned long long *__restrict__ stats) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float f = f0[t];
        if (f == -__builtin_inff()) {
            f = __builtin_inff();
            f0[t] = f;
            cflag[t] = 2;
        }
        if (raw) {
            const float q = f / (f > 0.0f ? rs_max : rs_min);
            raw[t] = fabsf(q) == __builtin_inff() ? q : q - fabsf(q) * 2.4e-7f;
        }
    }
}

__global__ void pilot_unset_count_kernel(const float *__restrict__ f0, int64_t n, unsigned long long *__restrict__ stats) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        if (f0[t] == -__builtin_inff()) {
            atomicAdd(stats + 0, 1ull);  // the search's total
            atomicAdd(stats + 4, 1ull);  // this chunk's
        }
}

// The flagged queries of a chunk, listed in ascending order by ONE workgroup (a megabyte of flag bytes: two passes of a few
// microseconds): pos[] their positions in the chunk, self[] the stored ids the tie path excludes (-1: a query vector), cnt[0] how
// many, cnt[1] how many of them carry flag 2 (a warm start that could not be verified).  Of a triangle shard only the queries this
// rank owns count.
__global__ __launch_bounds__(1024) void flag_compact_kernel(const uint8_t *__restrict__ cflag, int64_t m, int tri_rank, int tri_world, int bq,
                                                            int64_t q0, const int64_t *__restrict__ qid, int by_vector,
                                                            int32_t *__restrict__ pos, int64_t *__restrict__ self, long long *__restrict__ cnt) {
    __shared__ int part[1024], part2[1024];
    const int tid = threadIdx.x;
    const int64_t per = ((m + 1023) / 1024 + 15) & ~(int64_t)15, a = std::min<int64_t>(m, tid * per), b = std::min<int64_t>(m, a + per);
    auto mine = [&](int64_t t) { return tri_world <= 1 || (int)((t / bq) % tri_world) == tri_rank; };
    // sixteen flags per load (a thread's slice starts on a multiple of 16 and the buffer is 256-byte aligned; one byte per load was
    // 0.85 ms of every C4 pass for a million flags, profiles/r06_zl_timeline_topk_c4.txt), nearly all of them zero
    auto each_set = [&](auto &&f) {
        for (int64_t t0 = a; t0 < b; t0 += 16) {
            if (t0 + 16 <= m) {
                const uint4 w = *reinterpret_cast<const uint4 *>(cflag + t0);
                if (!(w.x | w.y | w.z | w.w)) continue;
            }
            const int64_t t1 = std::min<int64_t>(b, t0 + 16);
            for (int64_t t = t0; t < t1; t++) {
                const uint8_t v = cflag[t];
                if (v && mine(t)) f(t, v);
            }
        }
    };
    int c1 = 0, c2 = 0;
    each_set([&](int64_t, uint8_t v) { c1++, c2 += v == 2; });
    part[tid] = c1;
    part2[tid] = c2;
    __syncthreads();
    if (tid == 0) {
        int run = 0, run2 = 0;
        for (int i = 0; i < 1024; i++) {
            const int v = part[i];
            part[i] = run;
            run += v;
            run2 += part2[i];
        }
        cnt[0] = run;
        cnt[1] = run2;
    }
    __syncthreads();
    int at = part[tid];
    each_set([&](int64_t t, uint8_t) {
        pos[at] = (int32_t)t;
        self[at] = by_vector ? -1 : (qid ? qid[t] : q0 + t);
        at++;
    });
}

__global__ void fill_kernel(float *__restrict__ out, int64_t n, float v) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) out[t] = v;
}

}  // namespace

namespace gorse {

bool topk_mfma_usable(const gorse_topk *h, int64_t nq, int k) {
    if (!h->mfma_ok || g_topk_force_path == 1) return false;
    if (k + 1 > kMaxKth) return false;
    // Below ~800 queries the scan (path A: distances in the reference's order, select_fast_kernel) is faster: 0.12 ms per query at
    // N = 1M, against the ~40-100 ms one or two workgroups need to sweep a million rows on their own (profiles/
    // r02_v / r02_w_probe_query_latency.txt: 64 queries 7.4 ms on the scan, 41 ms on the sweep; 512 queries 96 ms on the sweep).
    return g_topk_force_path >= 2 || nq >= kMinSweepQueries;
}

int32_t topk_mfma_prepare(gorse_topk *h) {
    h->mfma_ok = false;
    if (h->metric != GORSE_METRIC_NEG_DOT && h->metric != GORSE_METRIC_COSINE && h->metric != GORSE_METRIC_EUCLIDEAN)
        return GORSE_OK;
    const int64_t N = h->N;
    const int d = h->d;
    const bool bf = h->dtype == GORSE_DTYPE_BF16;
    const int need = (int)ceil_div(bf ? d : 3 * (int64_t)d, 16);
    int kp = 0;
    for (int c : kSupportedKP)
        if (c >= need) {
            kp = c;
            break;
        }
    if (kp == 0) return GORSE_OK;  // too deep for register-resident query operands: path A
    // the error bound needs finite norms (and non-zero ones for cosine, whose reference distance is then NaN)
    std::vector<float> n2((size_t)N);
    GORSE_HIP_CHECK(hipMemcpyAsync(n2.data(), h->norm2.p, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    float mx = 0.0f;
    for (float v : n2) {
        if (!(v == v) || std::isinf(v)) return GORSE_OK;
        if (h->metric == GORSE_METRIC_COSINE && !(v > 0.0f)) return GORSE_OK;
        mx = std::max(mx, v);
    }
    h->max_norm = std::sqrt(mx) * 1.0001f;
    {   // cosine: row scales within 2 % of each other -> the block-level bound loses nothing (topk_sweep_kernel)
        float mn2 = std::numeric_limits<float>::infinity();
        for (float v : n2) mn2 = std::min(mn2, v);
        h->coarse_ok = h->metric == GORSE_METRIC_COSINE && mn2 > 0.0f && std::sqrt(mx / mn2) <= 1.02f;
        if (h->metric == GORSE_METRIC_COSINE && mn2 > 0.0f) {  // bounds of 1 / sqrtf(norm2[i]) as the device rounds it
            h->rs_min = (1.0f / std::sqrt(mx)) * (1.0f - 1e-6f);
            h->rs_max = (1.0f / std::sqrt(mn2)) * (1.0f + 1e-6f);
        }
    }
    h->kp = kp;
    const int kpad = kp * 16;
    // forward error of two different summation orders of the same K' exact products (the reference's fp32
    // chain, gamma <= K' u, and the MFMA's, gamma <= 2 K' u allowing truncating adders), plus the fp32 roundings
    // of the cosine formula; fp32 inputs add the dropped lo.lo / residual terms of the RNE split (3 * 2^-16).
    const double u = std::ldexp(1.0, -24);
    double coef = bf ? (3.0 * d + 64.0) * u : (9.0 * d + 64.0) * u + 3.2 * std::ldexp(1.0, -16);
    coef += 16.0 * u;
    h->err_coef = (float)(coef * 1.01);
    if (bf && d == kpad) {
        h->opA = h->opB = h->Xb.p;
    } else {
        GORSE_TRY(h->opA_own.alloc((size_t)N * kpad));
        if (bf) {
            pad_bf16_kernel<<<dim3(2048), dim3(256), 0, h->stream>>>(h->Xb.p, nullptr, N, d, kpad, h->opA_own.p);
            h->opA = h->opB = h->opA_own.p;
        } else {
            GORSE_TRY(h->opB_own.alloc((size_t)N * kpad));
            split_f32_kernel<<<dim3(2048), dim3(256), 0, h->stream>>>(h->X.p, N, d, kpad, 0, h->opA_own.p);
            split_f32_kernel<<<dim3(2048), dim3(256), 0, h->stream>>>(h->X.p, N, d, kpad, 1, h->opB_own.p);
            h->opA = h->opA_own.p;
            h->opB = h->opB_own.p;
        }
        GORSE_HIP_CHECK(hipGetLastError());
    }
    // the per-row value of the sweep's epilogue: cosine 1 / |x|, Euclidean -|x|^2 / 2 (added), -dot 1; kTopkRowPad NaN entries
    // behind it -- a tile of the sweep reaches past N, and a NaN value keeps those rows out of every list
    GORSE_TRY(h->rscale.alloc((size_t)N + kTopkRowPad));
    if (h->metric == GORSE_METRIC_EUCLIDEAN)
        bias_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>(h->norm2.p, N, h->rscale.p);
    else if (h->metric == GORSE_METRIC_COSINE)
        rscale_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>(h->norm2.p, N, h->rscale.p);
    else
        fill_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>(h->rscale.p, N, 1.0f);
    fill_kernel<<<dim3(1), dim3(256), 0, h->stream>>>(h->rscale.p + N, kTopkRowPad, __builtin_nanf(""));
    GORSE_HIP_CHECK(hipGetLastError());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->mfma_ok = true;
    return GORSE_OK;
}

}  // namespace gorse

namespace {
constexpr int kSymBQ = 32 * kNcbMain * kWavesMain;  // queries per workgroup (= query block) of the main sweep: the unit of a triangle shard

// What the stages of one chunk of a search share.  topk_mfma_search runs them back to back; the triangle-sharded search
// (gorse_topk_tri_*) keeps the state in the handle between its calls: the pilots over THIS rank's slice of the queries, then -- once
// every rank's thresholds have arrived -- the main sweep over this rank's query blocks, then -- once the foreign lists have been
// exchanged -- rescoring and tie path for the queries this rank owns.
}  // namespace
namespace gorse {
struct TopkChunkState {
    int64_t nq = 0, m = 0, c0 = 0, q_contig_begin = -1, mb = 0;
    const int64_t *qid_host = nullptr;
    const float *qv_dev = nullptr;
    bool by_vector = false, contiguous = false, euclid = false, warm = false, sym = false;
    int k = 0, kth = 0, prune0 = 0, pilot_stride = 16;
    int64_t expect = 0;
    float other = 0.0f;
    const uint16_t *Bop = nullptr;
    const float *qn2 = nullptr, *Qf = nullptr;
    SweepParams sp;
    int tri_rank = 0, tri_world = 1;
    int sym_slices = 1;  // row slices per query block of the symmetric main sweep (a triangle shard with few, long workgroups)
    int tri_stage = 0;  // gorse_topk_tri_*: 0 none, 1 begun (pilots of the own slice done), 2 swept, 3 finished
    int64_t tri_lo = 0, tri_hi = 0;  // the slice of the queries whose pilots this rank ran
};
}  // namespace gorse
namespace {
using ChunkState = gorse::TopkChunkState;

// the per-query arrays of a sweep restricted to the queries [lo, lo + n) (a pilot over one rank's slice)
SweepParams slice_params(const SweepParams &sp, int64_t lo, int64_t n, int kpad) {
    SweepParams p = sp;
    p.B = sp.B + lo * (int64_t)kpad;
    p.qmargin = sp.qmargin + lo;
    p.cbuf = sp.cbuf + lo * kCap;
    p.ccnt = sp.ccnt + lo;
    p.cflag = sp.cflag + lo;
    if (sp.f0) p.f0 = sp.f0 + lo;
    p.nq = n;
    return p;
}

int32_t search_setup(gorse_topk *h, ChunkState &cs, const int64_t *qid_host, int64_t q_contig_begin, const float *qv_dev, int64_t nq,
                     int k, int prune0) {
    const int d = h->d, kpad = h->kp * 16;
    cs = ChunkState();
    cs.nq = nq, cs.qid_host = qid_host, cs.q_contig_begin = q_contig_begin, cs.qv_dev = qv_dev, cs.k = k, cs.prune0 = prune0;
    cs.by_vector = qv_dev != nullptr;
    cs.contiguous = !cs.by_vector && qid_host == nullptr;
    const bool exclude_self = !cs.by_vector;
    cs.kth = k + (exclude_self ? 1 : 0);
    // with a mask the query's own row may or may not be admissible: the larger count only sends a query whose list comes out
    // one short (fewer than k admissible rows in all) to path A
    const int64_t admissible = h->has_mask ? h->n_admissible : h->N - (exclude_self ? 1 : 0);
    cs.expect = std::min<int64_t>(k, admissible);
    cs.euclid = h->metric == GORSE_METRIC_EUCLIDEAN;
    cs.other = h->metric == GORSE_METRIC_COSINE ? 1.0f : h->max_norm;
    const int64_t mb = std::min(nq, kChunkQ);
    cs.mb = mb;
    GORSE_TRY(h->cbuf.ensure((size_t)mb * kCap));
    GORSE_TRY(h->ccnt.ensure((size_t)mb));
    GORSE_TRY(h->cflag.ensure((size_t)mb));
    GORSE_TRY(h->qmargin.ensure((size_t)mb));
    GORSE_TRY(h->res_idx.ensure((size_t)mb * k));
    GORSE_TRY(h->res_dist.ensure((size_t)mb * k));
    GORSE_TRY(h->res_cnt.ensure((size_t)mb));
    if (!cs.contiguous) {
        GORSE_TRY(h->opQ.ensure((size_t)mb * kpad));
        GORSE_TRY(h->qn2.ensure((size_t)mb));
    }
    if (qid_host) GORSE_TRY(h->qid.ensure((size_t)mb));
    (void)d;
    h->n_fallback = 0;
    h->n_tie = 0;
    h->n_resweep = 0;
    return GORSE_OK;
}

// stage A: the chunk's query operands, margins and the main sweep's parameters (cs.sp); decides whether the sweep is warm-started
int32_t chunk_prepare(gorse_topk *h, ChunkState &cs) {
    const uint16_t *Bop;
    const float *qn2;
    const float *Qf = nullptr;
    const int64_t m = cs.m, c0 = cs.c0, q_contig_begin = cs.q_contig_begin;
    const int64_t *qid_host = cs.qid_host;
    const float *qv_dev = cs.qv_dev;
    const int d = h->d, kpad = h->kp * 16, kth = cs.kth;
    const bool contiguous = cs.contiguous, euclid = cs.euclid;
    const float other = cs.other;
    if (contiguous) {
        Bop = h->opB + (q_contig_begin + c0) * kpad;
        qn2 = h->norm2.p + q_contig_begin + c0;
    } else if (qid_host) {
        GORSE_HIP_CHECK(hipMemcpyAsync(h->qid.p, qid_host + c0, (size_t)m * 8, hipMemcpyHostToDevice, h->stream));
        gather_queries_kernel<<<dim3((unsigned)m), dim3(64), 0, h->stream>>>(h->opB, h->norm2.p, h->qid.p, m, kpad,
                                                                            h->opQ.p, h->qn2.p);
        GORSE_HIP_CHECK(hipGetLastError());
        Bop = h->opQ.p;
        qn2 = h->qn2.p;
    } else {
        Qf = qv_dev + c0 * d;
        if (h->dtype == GORSE_DTYPE_BF16)
            pad_bf16_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>(nullptr, Qf, m, d, kpad, h->opQ.p);
        else
            split_f32_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>(Qf, m, d, kpad, 1, h->opQ.p);
        GORSE_HIP_CHECK(hipGetLastError());
        GORSE_TRY(topk_compute_norms(h, Qf, m, h->qn2.p));
        Bop = h->opQ.p;
        qn2 = h->qn2.p;
    }
    if (euclid)
        margin_euclid_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(m, 256), 1024)), dim3(256), 0, h->stream>>>(
            qn2, m, h->err_coef, h->max_norm, (float)(d * 5.9604645e-8), h->qmargin.p);
    else
        margin_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(m, 256), 1024)), dim3(256), 0, h->stream>>>(
            qn2, m, h->err_coef, other, h->qmargin.p);
    GORSE_HIP_CHECK(hipGetLastError());
    GORSE_HIP_CHECK(hipMemsetAsync(h->cflag.p, 0, (size_t)m, h->stream));
    SweepParams &sp = cs.sp;
    sp.A = h->opA;
    sp.B = Bop;
    sp.rscale = h->has_mask ? h->rscale_m.p : h->rscale.p;
    sp.qmargin = h->qmargin.p;
    sp.cbuf = h->cbuf.p;
    sp.ccnt = h->ccnt.p;
    sp.cflag = h->cflag.p;
    sp.hbuf = nullptr;
    sp.hcnt = nullptr;
    // epilogue: Euclidean adds its bias to every score; cosine multiplies every score by the row scale unless the scales
    // lie within 2 % of each other (then the block's largest raw score times the extreme scale bounds the block and
    // rows are scaled only behind that test; variant bit 2 / 3: off / on whatever the norms); -dot has the value 1
    // (NaN for a masked row), which the block test never needs
    const bool cos_coarse = (g_topk_variant & 4) ? false : ((g_topk_variant & 8) ? true : h->coarse_ok);
    sp.ep = euclid ? EP_BIAS : (h->metric == GORSE_METRIC_COSINE ? (cos_coarse ? EP_COARSE : EP_SCALE) : EP_COARSE);
    // a list is compacted (threshold raised to its K-th best) once a sub-list holds 128: 4.7 ms of the C4 pass less than at
    // kCompactAt / 2 = 224 (fewer rows accepted late in the sweep; profiles/r03_p_probe_c4_floor.txt)
    sp.compact_at = (g_topk_variant & 32) ? kCompactAt / 2 : ((g_topk_variant & 64) ? 96 : 128);
    sp.prof = nullptr;
    if (g_topk_variant & 16) {
        GORSE_TRY(h->sweep_prof.ensure(16));
        GORSE_HIP_CHECK(hipMemsetAsync(h->sweep_prof.p, 0, 16 * sizeof(unsigned long long), h->stream));
        sp.prof = h->sweep_prof.p;
    }
    sp.N = h->N;
    sp.nq = m;
    sp.kth = kth;
    sp.f0 = nullptr, sp.f_out = nullptr, sp.tile_stride = 1;
    sp.nslices = 1;
    sp.probe = (g_topk_variant >> 17) & 3;  // variant bits 17-18: timing probes of the main sweep (the call then returns garbage)
    // bounds of the per-row value for the DMA sweeps' block test: the cosine scales; without them (a masked -dot index:
    // the value is 1 or NaN) the bound is the score itself
    sp.rs_min = h->metric == GORSE_METRIC_COSINE ? h->rs_min : 1.0f;
    sp.rs_max = h->metric == GORSE_METRIC_COSINE ? h->rs_max : 1.0f;
    sp.sym_q0 = 0;
    sp.frow = sp.frow_raw = nullptr;
    sp.fbuf = nullptr;
    sp.fcnt = nullptr;
    sp.sym_stats = nullptr;
    sp.sym_rank = 0, sp.sym_world = 1, sp.sym_slices = 1;
    sp.fwarm = nullptr;
    sp.sym_probe = (g_topk_variant >> 25) & 3;  // variant bits 25-26: timing probes of the symmetric sweep (the call returns garbage)
    // Warm start.  A streaming threshold that begins at -inf accepts ~kth * ln(N / kth) * (its lag) rows per query (1800
    // at C4) and every accepted row costs its 32 x 32 block the slow epilogue.  A pilot sweep over every 16th row tile
    // with kth_pilot = j yields the j-th best score of a 1/16 sample: with X ~ Binomial(kth - 1, 1/16) sample members
    // among the true kth - 1 best, that score is below the kth best of ALL rows unless X >= j -- j is chosen for a
    // tail of ~1e-3 -- and about 16 j rows lie above it.  The main sweep starts there and VERIFIES it (compact_query
    // flags a query whose K-th-best bound does not reach its threshold); those few queries join the tie queries' stage.
    // (probe: variant bit 20 = every 8th row tile, bit 21 = every 32nd, bit 22 = without the 1/256 pilot in front)
    cs.pilot_stride = (g_topk_variant & (1 << 20)) ? 8 : ((g_topk_variant & (1 << 21)) ? 32 : 16);
    // probe / test switches: bit 8 = no warm start, bit 9 = warm start whatever N, bit 10 = a pilot kth of 2 (thresholds
    // far too high: most queries fail the verification and are swept again)
    cs.warm = !(g_topk_variant & 256) && kth >= 8 && (h->N >= (int64_t)1 << 17 || (g_topk_variant & 512));
    cs.Bop = Bop, cs.qn2 = qn2, cs.Qf = Qf;
    return GORSE_OK;
}

// stage B: the pilot sweeps (see chunk_prepare's comment on the warm start) over the queries [lo, hi) of the chunk: their thresholds
// land in h->f0[lo .. hi)
int32_t chunk_pilots(gorse_topk *h, ChunkState &cs, int64_t lo, int64_t hi) {
    if (hi <= lo) return GORSE_OK;
    // two pilots: the 1/16 sample's own sweep would start cold (and a cold start accepts ~1500 rows per query however
    // few rows there are: every block of its 1/16 of the tiles would take the slow epilogue), so a 1/256 sample --
    // a subset of the 1/16 sample -- proposes ITS thresholds first, by the same rule
    auto pilot_kth = [](int of, int ratio) {
        const double lam = (double)(of - 1) / ratio;
        return (int)std::ceil(lam + 3.3 * std::sqrt(lam) + 1.5);
    };
    SweepParams p2 = slice_params(cs.sp, lo, hi - lo, h->kp * 16);
    p2.kth = (g_topk_variant & 1024) ? 2 : pilot_kth(cs.kth, cs.pilot_stride);
    p2.tile_stride = cs.pilot_stride;
    p2.f_out = h->f0.p + lo;
    p2.prof = nullptr;  // the instrumented twin profiles the main sweep only
    p2.probe = 0;
    if ((h->N >= (int64_t)1 << 18 || (g_topk_variant & 512)) && !(g_topk_variant & (1 << 22))) {
        SweepParams p1 = p2;
        p1.kth = pilot_kth(p2.kth, 16);
        p1.tile_stride = cs.pilot_stride * 16;
        p1.f_out = h->f1.p + lo;
        GORSE_TRY(dispatch_sweep(h, p1, false));
        GORSE_HIP_CHECK(hipMemsetAsync(h->cflag.p + lo, 0, (size_t)(hi - lo), h->stream));
        p2.f0 = h->f1.p + lo;
    }
    GORSE_TRY(dispatch_sweep(h, p2, false));
    GORSE_HIP_CHECK(hipMemsetAsync(h->cflag.p + lo, 0, (size_t)(hi - lo), h->stream));  // a pilot's flags say nothing about the query
    return GORSE_OK;
}

// stage C: the main sweep -- symmetric where the call allows it (cs.sym says which form ran)
int32_t chunk_main(gorse_topk *h, ChunkState &cs) {
    const int64_t m = cs.m;
    SweepParams &sp = cs.sp;
    if (cs.warm) sp.f0 = h->f0.p;
    // The symmetric form (topk_sweep_kernel, SYM): the queries are a contiguous range of the stored rows that starts on a tile
    // boundary, the sweep is warm-started (a foreign workgroup cannot tighten a threshold: it needs a good one from the start)
    // and there are at least two query blocks.  Variant bit 23 switches it off (the square sweep: tests, ablation).
    const int64_t sym_q0 = cs.q_contig_begin + cs.c0;
    bool sym = cs.warm && cs.contiguous && !(g_topk_variant & (1 << 23)) && sweep_sym_kp(h->kp) && topk_rows_per_tile() == 128 &&
               sym_q0 % 128 == 0 && m >= 2 * 32 * kNcbMain * kWavesMain && sp.probe <= 1 && !sp.prof;
    if (cs.warm) {
        // sym_stats[0..3]: the counters of the whole search (every chunk adds to them); [4]: THIS chunk's queries without a pilot
        // threshold -- cleared per chunk, so that a chunk that was not eligible for the symmetric form leaves nothing behind for the
        // next one's decision (round 5 took a difference against a host copy that only symmetric chunks updated)
        GORSE_TRY(h->sym_stats.ensure(8));
        if (cs.c0 == 0) GORSE_HIP_CHECK(hipMemsetAsync(h->sym_stats.p, 0, 8 * sizeof(unsigned long long), h->stream));
        else GORSE_HIP_CHECK(hipMemsetAsync(h->sym_stats.p + 4, 0, sizeof(unsigned long long), h->stream));
        pilot_unset_count_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(m, 256), 1024)), dim3(256), 0, h->stream>>>(h->f0.p, m, h->sym_stats.p);
        GORSE_HIP_CHECK(hipGetLastError());
    }
    if (sym && cs.tri_world == 1) {
        // A query the pilot left without a threshold costs the symmetric form a sweep of its own (the tie path's), the square form
        // only a cold start: where the pilot fails for many (rows sorted so that the systematic sample misleads it -- the
        // Euclidean case of test_warm_started_sweep_returns_the_same_rows), the square sweep is the better one.  One 8-byte read
        // behind the pilots.
        unsigned long long unset = 0;
        GORSE_HIP_CHECK(hipMemcpyAsync(&unset, h->sym_stats.p + 4, sizeof(unset), hipMemcpyDeviceToHost, h->stream));
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        if ((int64_t)unset * 32 > m && cs.tri_world == 1) sym = false;
    }
    h->last_sym = sym;
    if (sym) {
        GORSE_TRY(h->fbuf.ensure((size_t)cs.mb * kCapF));
        GORSE_TRY(h->fcnt.ensure((size_t)cs.mb));
        const bool rawf = sp.ep == EP_COARSE;
        if (rawf) GORSE_TRY(h->f0raw.ensure((size_t)cs.mb));
        GORSE_HIP_CHECK(hipMemsetAsync(h->fcnt.p, 0, (size_t)m * 4, h->stream));
        sym_thresholds_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(m, 256), 1024)), dim3(256), 0, h->stream>>>(
            h->f0.p, rawf ? h->f0raw.p : nullptr, h->cflag.p, m, sp.rs_min, sp.rs_max, h->sym_stats.p);
        GORSE_HIP_CHECK(hipGetLastError());
        sp.sym_q0 = sym_q0;
        sp.frow = h->f0.p;
        sp.frow_raw = rawf ? h->f0raw.p : nullptr;
        sp.fbuf = h->fbuf.p;
        sp.fcnt = h->fcnt.p;
        sp.sym_stats = h->sym_stats.p;
        sp.f_out = h->f1.p;  // the thresholds the workgroups end with: topk_rescore_kernel's verification reads them
        sp.sym_rank = cs.tri_rank, sp.sym_world = cs.tri_world;
        sp.sym_slices = cs.sym_slices;
        if (cs.sym_slices > 1)  // (slice 0's flags were cleared by chunk_prepare and may carry sym_thresholds_kernel's 2)
            GORSE_HIP_CHECK(hipMemsetAsync(h->cflag.p + cs.mb, 0, (size_t)(cs.sym_slices - 1) * cs.mb, h->stream));
    }
    GORSE_TRY(dispatch_sweep(h, sp, false, sym));
    cs.sym = sym;
    return GORSE_OK;
}

// stage D: exact rescoring of the lists, the tie path for what that leaves undecided, the literal scan for what THAT leaves; the
// chunk's rows end up in h->res_idx / res_dist / res_cnt (a triangle shard: the rows of the queries this rank owns)
int32_t chunk_finish(gorse_topk *h, ChunkState &cs) {
    const int64_t m = cs.m, c0 = cs.c0, q_contig_begin = cs.q_contig_begin;
    const int64_t *qid_host = cs.qid_host;
    const int d = h->d, kpad = h->kp * 16, k = cs.k, kth = cs.kth, prune0 = cs.prune0;
    const int64_t expect = cs.expect;
    const bool by_vector = cs.by_vector, sym = cs.sym;
    const uint16_t *Bop = cs.Bop;
    const float *qn2 = cs.qn2, *Qf = cs.Qf;
    const SweepParams &sp = cs.sp;
    int tok;
    RescoreParams rp;
    rp.X = h->X.p;
    rp.Xb = h->dtype == GORSE_DTYPE_BF16 ? h->Xb.p : nullptr;
    rp.norm2 = h->norm2.p;
    rp.Qf = Qf;
    rp.qn2 = qn2;
    rp.qid = qid_host ? h->qid.p : nullptr;
    rp.q0 = q_contig_begin + c0;
    rp.tri_rank = cs.tri_rank, rp.tri_world = cs.tri_world, rp.tri_bq = kSymBQ;
    rp.own_slices = sym ? cs.sym_slices : 1, rp.own_stride = cs.mb;
    rp.cbuf = h->cbuf.p;
    rp.ccnt = h->ccnt.p;
    rp.cflag = h->cflag.p;
    rp.d = d;
    rp.metric = h->kernel_metric();
    rp.k = k;
    rp.prune0 = prune0;
    rp.expect = expect;
    rp.out_idx = h->res_idx.p;
    rp.out_dist = h->res_dist.p;
    rp.out_cnt = h->res_cnt.p;
    rp.fbuf = sym ? h->fbuf.p : nullptr;
    rp.fcnt = sym ? h->fcnt.p : nullptr;
    rp.f0 = sym ? h->f0.p : nullptr;
    rp.ffinal = sym ? h->f1.p : nullptr;
    rp.sym_stats = sym ? h->sym_stats.p : nullptr;
    rp.qmargin = h->qmargin.p;
    rp.kth = kth;
    const size_t lds = ((size_t)(1 + 2 * kGroupsPerBlock) * d + 2 * kCapT + 8) * 4;
    tok = h->prof.begin(GORSE_PROF_TOPK_SELECT, h->stream);
    topk_rescore_kernel<<<dim3((unsigned)m), dim3(kBlock), lds, h->stream>>>(rp);
    GORSE_HIP_CHECK(hipGetLastError());
    h->prof.end(tok, h->stream);
    // The queries the sweep + rescoring could not decide (ties in the top k + 1, NaN, overflow; of a triangle shard: among the queries
    // this rank owns -- a foreign query's flag, a staging overflow seen here, has travelled to its owner) are listed ON THE DEVICE,
    // in ascending order, with the stored ids the replay excludes: the host reads two words.  (Rounds 1-5 copied the chunk's flag
    // bytes out and walked them: 0.63 ms between the rescoring and the tie path of a C4 pass, profiles/r06_i_timeline_topk_c4_before.txt.)
    GORSE_TRY(h->fl_pos.ensure((size_t)cs.mb));
    GORSE_TRY(h->fl_self.ensure((size_t)cs.mb));
    GORSE_TRY(h->fl_cnt.ensure(2));
    flag_compact_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(h->cflag.p, m, cs.tri_rank, cs.tri_world, kSymBQ,
                                                              by_vector ? (int64_t)-1 : q_contig_begin + c0, qid_host ? h->qid.p : nullptr,
                                                              by_vector ? 1 : 0, h->fl_pos.p, h->fl_self.p, h->fl_cnt.p);
    GORSE_HIP_CHECK(hipGetLastError());
    long long fl[2] = {0, 0};
    GORSE_HIP_CHECK(hipMemcpyAsync(fl, h->fl_cnt.p, sizeof(fl), hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    const int64_t n_flagged = fl[0];
    h->n_resweep += fl[1];
    auto fetch_pos = [&](int64_t at, int64_t n, std::vector<int64_t> &out) -> int32_t {  // flagged positions [at, at + n) to the host
        std::vector<int32_t> tmp((size_t)n);
        GORSE_HIP_CHECK(hipMemcpyAsync(tmp.data(), h->fl_pos.p + at, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        for (int32_t v : tmp) out.push_back(v);
        return GORSE_OK;
    };
    auto stored_id = [&](int64_t t) -> int64_t {
        return by_vector ? -1 : (qid_host ? qid_host[c0 + t] : q_contig_begin + c0 + t);
    };
    // stage 2: history sweep + literal heap replay (topk_tie_sort_kernel, topk_tie_replay_kernel) for the flagged queries
    std::vector<int64_t> rest;
    // (with a mask the replay's gap arithmetic -- which counts the rows between two recorded ones -- would count masked
    // rows too: those queries take the literal scan, which skips masked rows)
    if (n_flagged > 0 && g_topk_force_path != 3 && !h->has_mask) {
        std::vector<uint8_t> f2;
        for (int64_t f0 = 0; f0 < n_flagged; f0 += kReplayChunk) {
            const int64_t m2 = std::min<int64_t>(kReplayChunk, n_flagged - f0);
            const int32_t *pos_dev = h->fl_pos.p + f0;
            const int64_t *self_dev = h->fl_self.p + f0;
            GORSE_TRY(h->rp_op.ensure((size_t)m2 * kpad));
            GORSE_TRY(h->rp_margin.ensure((size_t)m2));
            // Row slices: a history sweep serves few queries (1 % of a chunk), so its workgroups are few, and each would walk
            // all N rows on its own -- 56 ms for 10,000 queries at C4, three quarters of the tie path.  Cut into slices of
            // rows, slices x as many workgroups each walk 1 / slices of the rows; topk_tie_sort_kernel joins the slices.
            // Variant bit 14: eight slices whatever N (lets small test inputs take the path); bit 15: one slice.
            int nsl = (int)std::min<int64_t>(kMaxSlices, std::max<int64_t>(1, h->N / 32768));
            // Few flagged queries -- a rank of a sharded search holds 1 / world of them, and the history sweep's time does not
            // shrink with their number: every workgroup (128 queries) walks its slice of the rows whatever else runs -- take more,
            // shorter slices until the launch has ~512 workgroups (at most kMaxSlicesFew, slices of at least 16384 rows).  A full
            // single-GPU C4 pass (10,136 queries: 80 x 8 workgroups) keeps its eight: sixteen were measured there, 9.2 + 14.0 ms
            // against 11.4 + 11.4 (every slice starts cold, the join and the replay pay for it).
            {
                // (counted in 128-query units whatever the history workgroup holds -- 512 queries in the eight-wave DMA form of round 6:
                // with that form's true count the rule would take 32 slices at C4's 10,136 queries, measured 228.3 ms per pass against
                // 220.2 with eight, 224.3 with sixteen: profiles/r06_zzl_probe_topk_c4_hist_slices.txt)
                const int64_t wgs = ceil_div(m2, (int64_t)64 * kNcbMain);
                while (nsl >= kMaxSlices && nsl < kMaxSlicesFew && wgs * nsl < 512 && h->N / (2 * (int64_t)nsl) >= 16384) nsl *= 2;
            }
            if (g_topk_variant & 16384) nsl = kMaxSlices;
            if (g_topk_variant & 32768) nsl = 1;
            const size_t sm2 = (size_t)nsl * (size_t)m2;
            GORSE_TRY(h->rp_cbuf.ensure(sm2 * kCap));
            GORSE_TRY(h->rp_hbuf.ensure(sm2 * kHistCap));
            GORSE_TRY(h->rp_ccnt.ensure(sm2));
            GORSE_TRY(h->rp_hcnt.ensure(sm2));
            GORSE_TRY(h->rp_flag.ensure(sm2));
            GORSE_TRY(h->rp_fslice.ensure(sm2));
            gather_pos_kernel<<<dim3((unsigned)m2), dim3(64), 0, h->stream>>>(Bop, h->qmargin.p, pos_dev, kpad,
                                                                             h->rp_op.p, h->rp_margin.p);
            GORSE_HIP_CHECK(hipGetLastError());
            GORSE_HIP_CHECK(hipMemsetAsync(h->rp_flag.p, 0, sm2, h->stream));
            GORSE_HIP_CHECK(hipMemsetAsync(h->rp_hcnt.p, 0, sm2 * 4, h->stream));
            GORSE_HIP_CHECK(hipMemsetAsync(h->rp_ccnt.p, 0, sm2 * 4, h->stream));
            SweepParams hp = sp;
            hp.f0 = nullptr;  // the history sweep records what a threshold that starts at -inf would have kept
            hp.fwarm = nullptr;
            // ... or, slice by slice, what a threshold that starts at a bound the main sweep's lists prove would have kept
            // (tie_warm_kernel; variant bit 11: every slice cold, the form of rounds 2-5)
            if (nsl > 1 && !(g_topk_variant & 2048)) {
                GORSE_TRY(h->rp_fwarm.ensure(sm2));
                WarmParams wp;
                wp.pos = pos_dev, wp.m2 = m2, wp.N = h->N, wp.nsl = nsl, wp.tile_rows = 64 /* the history sweep's tiles: launch_sweep */,
                wp.kth = kth;
                wp.cbuf = h->cbuf.p, wp.ccnt = h->ccnt.p;
                wp.own_slices = sym ? std::max(1, cs.sym_slices) : 1, wp.own_stride = cs.mb;
                wp.fbuf = sym ? h->fbuf.p : nullptr, wp.fcnt = sym ? h->fcnt.p : nullptr;
                wp.margin = h->rp_margin.p;
                wp.fwarm = h->rp_fwarm.p;
                tie_warm_kernel<<<dim3((unsigned)m2), dim3(64), 0, h->stream>>>(wp);
                GORSE_HIP_CHECK(hipGetLastError());
                hp.fwarm = h->rp_fwarm.p;
            }
            hp.B = h->rp_op.p;
            hp.qmargin = h->rp_margin.p;
            hp.cbuf = h->rp_cbuf.p;
            hp.ccnt = h->rp_ccnt.p;
            hp.cflag = h->rp_flag.p;
            hp.hbuf = h->rp_hbuf.p;
            hp.hcnt = h->rp_hcnt.p;
            hp.f_out = h->rp_fslice.p;
            hp.nslices = nsl;
            hp.nq = m2;
            tok = h->prof.begin(GORSE_PROF_TOPK_HIST, h->stream);
            GORSE_TRY(dispatch_sweep(h, hp, true));
            h->prof.end(tok, h->stream);
            ReplayParams pp;
            pp.X = h->X.p;
            pp.Xb = h->dtype == GORSE_DTYPE_BF16 ? h->Xb.p : nullptr;
            pp.norm2 = h->norm2.p;
            pp.Qf = Qf;
            pp.qn2 = qn2;
            pp.self = self_dev;
            pp.pos = pos_dev;
            pp.cbuf = h->rp_cbuf.p;
            pp.ccnt = h->rp_ccnt.p;
            pp.hbuf = h->rp_hbuf.p;
            pp.hcnt = h->rp_hcnt.p;
            pp.fslice = h->rp_fslice.p;
            pp.nslices = nsl;
            pp.prof = (g_topk_variant & 16) && h->sweep_prof.n >= 16 ? h->sweep_prof.p : nullptr;  // the sweep's counters are overwritten
            if (pp.prof) GORSE_HIP_CHECK(hipMemsetAsync(pp.prof, 0, 16 * sizeof(unsigned long long), h->stream));
            pp.nq = m2;
            pp.cflag = h->rp_flag.p;
            pp.N = h->N;
            pp.d = d;
            pp.metric = h->kernel_metric();
            pp.k = k;
            pp.prune0 = prune0;
            pp.out_idx = h->res_idx.p;
            pp.out_dist = h->res_dist.p;
            pp.out_cnt = h->res_cnt.p;
            GORSE_TRY(h->rp_sidx.ensure((size_t)m2 * kReplayCap));
            GORSE_TRY(h->rp_sdst.ensure((size_t)m2 * kReplayCap));
            GORSE_TRY(h->rp_scount.ensure((size_t)m2));
            pp.sidx = h->rp_sidx.p;
            pp.sdst = h->rp_sdst.p;
            pp.scount = h->rp_scount.p;
            const size_t rlds = ((size_t)2 * kReplayCap + (size_t)(1 + kGroupsPerBlock) * d + 4) * 4;
            GORSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&topk_tie_sort_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
            tok = h->prof.begin(GORSE_PROF_TOPK_REPLAY, h->stream);
            topk_tie_sort_kernel<<<dim3((unsigned)m2), dim3(kBlock), rlds, h->stream>>>(pp);
            GORSE_HIP_CHECK(hipGetLastError());
            // one lane per query (probe build, variant bit 19 / 16: the round-2 kernel, a wave per query)
#ifdef GORSE_PROBE
            const bool lanes = !(g_topk_variant & (1 << 19)) && !(g_topk_variant & 65536);
#else
            const bool lanes = true;
#endif
            if (lanes) {
                const int slots = k + 1;
                int qpw = 64;
                while (qpw > 16 && (size_t)(3 * slots + 2 * kLaneLog) * qpw * 4 > (size_t)144 * 1024) qpw >>= 1;
                // The lanes of a wave replay their queries in step: a wave takes as long as its lanes' branches laid end to end, and
                // the launch as long as its slowest wave, however few queries there are (1,250 queries of one rank of eight: 11.9 ms,
                // the 10,136 of a whole C4 pass: 11.5).  Fewer queries per wave while the launch has fewer waves than the chip has CUs
                // (variant bit 28: 64 whatever the size).
                while (qpw > 16 && !(g_topk_variant & (1 << 28)) && ceil_div(m2, (int64_t)qpw) < 256) qpw >>= 1;
                const size_t llds = (size_t)(3 * slots + 2 * kLaneLog) * qpw * 4;
                auto launch_lanes = [&](auto tag) -> int32_t {
                    constexpr int Q = decltype(tag)::value;
                    GORSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&topk_tie_replay_lane_kernel<Q>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
                    topk_tie_replay_lane_kernel<Q><<<dim3((unsigned)ceil_div(m2, Q)), dim3(64), llds, h->stream>>>(pp, m2, slots);
                    GORSE_HIP_CHECK(hipGetLastError());
                    return GORSE_OK;
                };
                if (qpw == 64) GORSE_TRY(launch_lanes(std::integral_constant<int, 64>()));
                else if (qpw == 32) GORSE_TRY(launch_lanes(std::integral_constant<int, 32>()));
                else GORSE_TRY(launch_lanes(std::integral_constant<int, 16>()));
            }
#ifdef GORSE_PROBE
            if (!lanes) {
                topk_tie_replay_kernel<<<dim3((unsigned)ceil_div(m2, 4)), dim3(256), 0, h->stream>>>(pp, m2, (g_topk_variant & 65536) ? 1 : 0);
                GORSE_HIP_CHECK(hipGetLastError());
            }
#endif
            h->prof.end(tok, h->stream);
            f2.resize((size_t)m2);
            GORSE_HIP_CHECK(hipMemcpyAsync(f2.data(), h->rp_flag.p, (size_t)m2, hipMemcpyDeviceToHost, h->stream));
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
            int64_t undecided = 0;
            for (int64_t r = 0; r < m2; r++) undecided += f2[r] != 0;
            h->n_tie += m2 - undecided;
            if (undecided > 0) {  // rare: their positions come to the host now
                std::vector<int64_t> at;
                GORSE_TRY(fetch_pos(f0, m2, at));
                for (int64_t r = 0; r < m2; r++)
                    if (f2[r]) rest.push_back(at[(size_t)r]);
            }
        }
    } else if (n_flagged > 0) {
        GORSE_TRY(fetch_pos(0, n_flagged, rest));
    }
    // stage 3: whatever is left replays the reference literally over all N vectors (path A); its rows go into
    // the chunk's device result arrays like everybody else's
    if (!rest.empty()) {
        h->n_fallback += (int64_t)rest.size();
        const int64_t bq = topk_scan_block_queries(h);
        std::vector<int64_t> ids((size_t)std::min<int64_t>(bq, (int64_t)rest.size()));
        std::vector<int32_t> ti((size_t)ids.size() * k), tc(ids.size());
        std::vector<float> td((size_t)ids.size() * k);
        GORSE_TRY(h->qbuf.ensure(ids.size() * (size_t)d));
        GORSE_TRY(h->qnorm.ensure(ids.size()));
        GORSE_TRY(h->qidx.ensure(ids.size()));
        for (size_t f0 = 0; f0 < rest.size(); f0 += (size_t)bq) {
            const int64_t fm = std::min<int64_t>(bq, (int64_t)(rest.size() - f0));
            for (int64_t r = 0; r < fm; r++) {
                const int64_t t = rest[f0 + r];
                ids[r] = stored_id(t);
                const float *src = by_vector ? Qf + t * d : h->X.p + ids[r] * d;
                GORSE_HIP_CHECK(hipMemcpyAsync(h->qbuf.p + r * d, src, (size_t)d * 4, hipMemcpyDeviceToDevice, h->stream));
                GORSE_HIP_CHECK(hipMemcpyAsync(h->qnorm.p + r, qn2 + t, 4, hipMemcpyDeviceToDevice, h->stream));
            }
            const int64_t *excl = nullptr;
            if (!by_vector) {
                GORSE_HIP_CHECK(hipMemcpyAsync(h->qidx.p, ids.data(), (size_t)fm * 8, hipMemcpyHostToDevice, h->stream));
                excl = h->qidx.p;
            }
            GORSE_TRY(topk_scan_block(h, fm, excl, by_vector ? nullptr : ids.data(), k, prune0, ti.data(), td.data(), tc.data(),
                                      /*reroute=*/false));  // these queries come FROM the replay: the literal kernel ends it
            for (int64_t r = 0; r < fm; r++) {
                const int64_t t = rest[f0 + r];
                GORSE_HIP_CHECK(hipMemcpyAsync(h->res_idx.p + t * k, ti.data() + r * k, (size_t)k * 4, hipMemcpyHostToDevice, h->stream));
                GORSE_HIP_CHECK(hipMemcpyAsync(h->res_dist.p + t * k, td.data() + r * k, (size_t)k * 4, hipMemcpyHostToDevice, h->stream));
                GORSE_HIP_CHECK(hipMemcpyAsync(h->res_cnt.p + t, tc.data() + r, 4, hipMemcpyHostToDevice, h->stream));
            }
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    }
    return GORSE_OK;
}

}  // namespace

namespace gorse {

int32_t topk_mfma_search(gorse_topk *h, const int64_t *qid_host, int64_t q_contig_begin, const float *qv_dev,
                         int64_t nq, int k, int prune0, int32_t *idx_out, float *dist_out, int32_t *cnt_out) {
    ChunkState &cs = *h->chunk_state();
    GORSE_TRY(search_setup(h, cs, qid_host, q_contig_begin, qv_dev, nq, k, prune0));
    for (int64_t c0 = 0; c0 < nq; c0 += kChunkQ) {
        const int64_t m = std::min(kChunkQ, nq - c0);
        cs.c0 = c0, cs.m = m;
        GORSE_TRY(chunk_prepare(h, cs));
        int tok = h->prof.begin(GORSE_PROF_TOPK_SWEEP, h->stream);
        if (cs.warm) {
            GORSE_TRY(h->f0.ensure((size_t)cs.mb));
            GORSE_TRY(h->f1.ensure((size_t)cs.mb));
            GORSE_TRY(chunk_pilots(h, cs, 0, m));
            if (g_topk_variant & (1 << 24)) {  // probe: stop behind the pilot, keep its flags and list lengths (results are garbage)
                h->dbg_flags.resize((size_t)m);
                h->dbg_counts.resize((size_t)m);
                GORSE_HIP_CHECK(hipMemcpyAsync(h->dbg_flags.data(), h->cflag.p, (size_t)m, hipMemcpyDeviceToHost, h->stream));
                GORSE_HIP_CHECK(hipMemcpyAsync(h->dbg_counts.data(), h->ccnt.p, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
                GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
                h->prof.end(tok, h->stream);
                return GORSE_OK;
            }
        }
        GORSE_TRY(chunk_main(h, cs));
        h->last_sym = cs.sym;
        if (cs.sp.probe || cs.sp.sym_probe) {  // timing probe: nothing behind the sweep is meaningful
            h->prof.end(tok, h->stream);
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
            continue;
        }
        // The queries whose warm start failed its verification carry flag 2: topk_rescore_kernel leaves flagged queries alone
        // and stage 2 of chunk_finish (history sweep from -inf + literal replay) answers them together with the tie queries -- a sweep
        // of their own would cost one workgroup a full pass over the rows (48 ms for 86 queries, profiles/r02_h_*).
        h->prof.end(tok, h->stream);
        GORSE_TRY(chunk_finish(h, cs));
        if (idx_out)
            GORSE_HIP_CHECK(hipMemcpyAsync(idx_out + c0 * k, h->res_idx.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, h->stream));
        if (dist_out)
            GORSE_HIP_CHECK(hipMemcpyAsync(dist_out + c0 * k, h->res_dist.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, h->stream));
        if (cnt_out)
            GORSE_HIP_CHECK(hipMemcpyAsync(cnt_out + c0, h->res_cnt.p, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    return GORSE_OK;
}

}  // namespace gorse

// ---- triangle-sharded all-pairs search (top-k row) --------------------------------------------
// Query-row sharding gives a rank the saving of the symmetric sweep on its own diagonal square only (1 / world^2 of the matrix).
// Here the TRIANGLE is sharded: rank r sweeps the query blocks C with C % world == r -- each block's own lists, and what it finds
// for the rows of every earlier block as queries, whoever owns them -- so the union over the ranks is exactly the single-rank
// symmetric sweep, every score still serving both of its queries.  The pilots are sharded by contiguous slices of the queries
// (their thresholds are all-gathered: 4 bytes per query); after the sweep a rank holds partial foreign lists of queries the
// OTHER ranks own, which travel to the owners (pack -> all-to-all -> unpack: ~130 entries of 8 bytes per query in all); rescoring
// and tie path then run on the owner.  The exchange itself is the caller's (gorse_amd/dist.py: torch.distributed, or host memory when
// the ranks are emulated on one device); every buffer that crosses the boundary may be host or device memory.
namespace {
__device__ __forceinline__ int64_t tri_query_of(int64_t i, int rank, int world) {  // the i-th query a rank owns
    return ((i / kSymBQ) * world + rank) * (int64_t)kSymBQ + i % kSymBQ;
}
__global__ void tri_pack_counts_kernel(const int32_t *__restrict__ fcnt, const uint8_t *__restrict__ cflag, int64_t n_owned, int dest,
                                       int world, int32_t *__restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_owned; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = tri_query_of(i, dest, world);
        const int32_t c = fcnt[t];
        // -1 = "this part of the list is useless": a staging overflow here (flag 1; flag 2 = no pilot threshold is known to every
        // rank) or more entries than a list holds -- the owner then sends the query down the tie path
        counts[i] = (cflag[t] == 1 || c > kCapF) ? -1 : c;
    }
}
// exclusive prefix of max(counts, 0) by ONE workgroup (n <= a million words: two passes over a few megabytes)
__global__ __launch_bounds__(1024) void tri_scan_kernel(const int32_t *__restrict__ counts, int64_t n, int64_t *__restrict__ offsets) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024, a = tid * per, b = a + per < n ? a + per : n;
    long long s = 0;
    for (int64_t i = a; i < b; i++) s += counts[i] > 0 ? counts[i] : 0;
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; i++) {
            const long long v = part[i];
            part[i] = run;
            run += v;
        }
        offsets[n] = run;
    }
    __syncthreads();
    long long run = part[tid];
    for (int64_t i = a; i < b; i++) {
        offsets[i] = run;
        run += counts[i] > 0 ? counts[i] : 0;
    }
}
__global__ void tri_pack_entries_kernel(const uint2 *__restrict__ fbuf, const int32_t *__restrict__ counts, const int64_t *__restrict__ offsets,
                                        int64_t n_owned, int dest, int world, uint2 *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n_owned; i += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int c = counts[i];
        const uint2 *src = fbuf + tri_query_of(i, dest, world) * kCapF;
        uint2 *dst = out + offsets[i];
        for (int e = lane; e < c; e += 64) dst[e] = src[e];
    }
}
__global__ void tri_unpack_kernel(uint2 *__restrict__ fbuf, int32_t *__restrict__ fcnt, const int32_t *__restrict__ counts,
                                  const int64_t *__restrict__ offsets, const uint2 *__restrict__ entries, int64_t n_owned, int rank, int world) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n_owned; i += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int64_t t = tri_query_of(i, rank, world);
        const int c = counts[i];
        const int base = fcnt[t];  // (every lane reads it before lane 0 writes: a wave runs in step)
        if (c < 0) {
            if (lane == 0) fcnt[t] = kCapF + 1;  // overflowed on the sender's side: topk_rescore_kernel flags the query
            continue;
        }
        const uint2 *src = entries + offsets[i];
        uint2 *dst = fbuf + t * kCapF;
        for (int e = lane; e < c; e += 64)
            if (base + e < kCapF) dst[base + e] = src[e];
        if (lane == 0) fcnt[t] = base + c;  // may pass kCapF: the list overflowed at its owner
    }
}

struct TriGeom {
    int64_t nq, nblk;
    int world;
    int64_t blocks_of(int r) const { return nblk > r ? (nblk - r + world - 1) / world : 0; }
    int64_t owned(int r) const {  // queries rank r owns: whole blocks, but for the globally last one
        const int64_t nb = blocks_of(r);
        if (nb == 0) return 0;
        const int64_t last = r + (nb - 1) * world;  // its last block
        return (nb - 1) * kSymBQ + std::min<int64_t>(kSymBQ, nq - last * kSymBQ);
    }
    void slice(int r, int64_t *lo, int64_t *hi) const {  // the queries whose pilots rank r runs: whole blocks, even split
        *lo = std::min(nq, nblk * r / world * kSymBQ);
        *hi = std::min(nq, nblk * (r + 1) / world * kSymBQ);
    }
};
TriGeom tri_geom(const ChunkState &cs) { return TriGeom{cs.m, ceil_div(cs.m, kSymBQ), cs.tri_world}; }

int32_t tri_state(gorse_topk *h, int stage_min, ChunkState **out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    ChunkState *cs = h->cstate;
    if (!cs || cs->tri_stage < stage_min) return fail(GORSE_ERR_INVALID, "gorse_topk_tri_*: called out of order (stage %d, needs %d)", cs ? cs->tri_stage : 0, stage_min);
    GORSE_TRY(h->use());
    *out = cs;
    return GORSE_OK;
}
}  // namespace

gorse::TopkChunkState *gorse_topk::chunk_state() {
    if (!cstate) cstate = new gorse::TopkChunkState();
    return cstate;
}
namespace gorse {
void topk_mfma_release(gorse_topk *h) {
    delete h->cstate;
    h->cstate = nullptr;
}
}  // namespace gorse

extern "C" int32_t gorse_topk_tri_begin(gorse_topk *h, int64_t q_begin, int64_t q_end, int32_t k, int32_t rank, int32_t world) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (q_begin < 0 || q_end > h->N || q_begin >= q_end || k <= 0) return fail(GORSE_ERR_RANGE, "bad query range");
    if (world < 1 || rank < 0 || rank >= world) return fail(GORSE_ERR_INVALID, "rank %d of %d", rank, world);
    const int64_t nq = q_end - q_begin;
    GORSE_TRY(h->use());
    // what the symmetric sweep needs (chunk_main): a warm-started MFMA search over one chunk that starts on a tile boundary
    const bool ok = topk_mfma_usable(h, nq, k) && nq <= kChunkQ && sweep_sym_kp(h->kp) && topk_rows_per_tile() == 128 && q_begin % 128 == 0 &&
                    nq >= 2 * kSymBQ && k + 1 >= 8 && h->N >= (int64_t)1 << 17 && !(g_topk_variant & ((1 << 23) | 256));
    if (!ok)
        return fail(GORSE_ERR_INVALID, "this search has no symmetric form (operand depth, a range that does not start on a 128-row boundary, fewer than "
                                       "%d queries or more than %lld, an index of fewer than 2^17 rows): shard its query rows instead", 2 * kSymBQ, (long long)kChunkQ);
    ChunkState &cs = *h->chunk_state();
    GORSE_TRY(search_setup(h, cs, nullptr, q_begin, nullptr, nq, k, 0));
    cs.c0 = 0, cs.m = nq, cs.tri_rank = rank, cs.tri_world = world;
    // A launch is as long as its longest workgroup, and the longest block of the triangle sweeps every row with its 512 queries on ONE
    // CU: 40 ms at C4, whatever the number of ranks (a rank of eight holds 244 workgroups for 256 CUs: its main sweep took those 40 ms
    // where its share of the work is 19).  Few, long workgroups are therefore cut by ROWS: slice y of a block takes an even share of
    // its tiles, with own lists of its own (slice-major, like a history sweep's: topk_rescore_kernel takes them together); 1, 2 or 4
    // slices, the fewest that give the launch ~768 workgroups.  (Variant bits 29-30 force 1 / 2 / 4: tests, ablation.)
    {
        const int64_t blocks = tri_geom(cs).blocks_of(rank);
        int S = 1;
        while (S < kMaxOwnSlices && blocks * S < 768) S *= 2;
        const int forced = (g_topk_variant >> 29) & 3;
        if (forced) S = forced == 1 ? 1 : (forced == 2 ? 2 : 4);
        cs.sym_slices = world > 1 || forced ? S : 1;
        if (cs.sym_slices > 1) {
            GORSE_TRY(h->cbuf.ensure((size_t)cs.sym_slices * cs.mb * kCap));
            GORSE_TRY(h->ccnt.ensure((size_t)cs.sym_slices * cs.mb));
            GORSE_TRY(h->cflag.ensure((size_t)cs.sym_slices * cs.mb));
        }
    }
    GORSE_TRY(chunk_prepare(h, cs));
    if (!cs.warm) return fail(GORSE_ERR_INVALID, "the sweep of this search is not warm-started");
    GORSE_TRY(h->f0.ensure((size_t)cs.mb));
    GORSE_TRY(h->f1.ensure((size_t)cs.sym_slices * cs.mb));
    tri_geom(cs).slice(rank, &cs.tri_lo, &cs.tri_hi);
    const int tok = h->prof.begin(GORSE_PROF_TOPK_SWEEP, h->stream);
    GORSE_TRY(chunk_pilots(h, cs, cs.tri_lo, cs.tri_hi));
    h->prof.end(tok, h->stream);
    cs.tri_stage = 1;
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_tri_slice(gorse_topk *h, int32_t rank, int64_t *lo, int64_t *hi, int64_t *owned) {
    ChunkState *cs;
    GORSE_TRY(tri_state(h, 1, &cs));
    if (rank < 0 || rank >= cs->tri_world) return fail(GORSE_ERR_INVALID, "rank %d of %d", rank, cs->tri_world);
    int64_t a, b;
    tri_geom(*cs).slice(rank, &a, &b);
    if (lo) *lo = a;
    if (hi) *hi = b;
    if (owned) *owned = tri_geom(*cs).owned(rank);
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_tri_thresholds_get(gorse_topk *h, int64_t lo, int64_t hi, float *dst) {
    ChunkState *cs;
    GORSE_TRY(tri_state(h, 1, &cs));
    if (lo < 0 || hi > cs->m || lo > hi || !dst) return fail(GORSE_ERR_RANGE, "bad slice");
    GORSE_HIP_CHECK(hipMemcpyAsync(dst, h->f0.p + lo, (size_t)(hi - lo) * 4, hipMemcpyDefault, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}
extern "C" int32_t gorse_topk_tri_thresholds_put(gorse_topk *h, int64_t lo, int64_t hi, const float *src) {
    ChunkState *cs;
    GORSE_TRY(tri_state(h, 1, &cs));
    if (lo < 0 || hi > cs->m || lo > hi || !src) return fail(GORSE_ERR_RANGE, "bad slice");
    GORSE_HIP_CHECK(hipMemcpyAsync(h->f0.p + lo, src, (size_t)(hi - lo) * 4, hipMemcpyDefault, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // the source may be reused
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_tri_sweep(gorse_topk *h) {
    ChunkState *cs;
    GORSE_TRY(tri_state(h, 1, &cs));
    const int tok = h->prof.begin(GORSE_PROF_TOPK_SWEEP, h->stream);
    GORSE_TRY(chunk_main(h, *cs));
    h->prof.end(tok, h->stream);
    if (!cs->sym) return fail(GORSE_ERR_INVALID, "the main sweep did not take the symmetric form");
    cs->tri_stage = 2;
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_tri_pack(gorse_topk *h, int32_t dest, int64_t *n_counts, int64_t *n_entries) {
    ChunkState *cs;
    GORSE_TRY(tri_state(h, 2, &cs));
    if (dest < 0 || dest >= cs->tri_world || dest == cs->tri_rank) return fail(GORSE_ERR_INVALID, "bad destination rank %d", dest);
    const int64_t n = tri_geom(*cs).owned(dest);
    GORSE_TRY(h->tri_counts.ensure((size_t)std::max<int64_t>(n, 1)));
    GORSE_TRY(h->tri_offsets.ensure((size_t)n + 1));
    long long total = 0;
    if (n > 0) {
        tri_pack_counts_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 2048)), dim3(256), 0, h->stream>>>(
            h->fcnt.p, h->cflag.p, n, dest, cs->tri_world, h->tri_counts.p);
        tri_scan_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(h->tri_counts.p, n, reinterpret_cast<int64_t *>(h->tri_offsets.p));
        GORSE_HIP_CHECK(hipGetLastError());
        GORSE_HIP_CHECK(hipMemcpyAsync(&total, h->tri_offsets.p + n, 8, hipMemcpyDeviceToHost, h->stream));
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        GORSE_TRY(h->tri_entries.ensure((size_t)std::max<long long>(total, 1)));
        if (total > 0) {
            tri_pack_entries_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(n, 4), 8192)), dim3(256), 0, h->stream>>>(
                h->fbuf.p, h->tri_counts.p, reinterpret_cast<const int64_t *>(h->tri_offsets.p), n, dest, cs->tri_world, h->tri_entries.p);
            GORSE_HIP_CHECK(hipGetLastError());
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // the message is complete when this returns: another handle's stream may read it
        }
    }
    h->tri_packed_counts = n, h->tri_packed_entries = total;
    if (n_counts) *n_counts = n;
    if (n_entries) *n_entries = total;
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_tri_pack_read(gorse_topk *h, int32_t *counts, uint64_t *entries) {
    ChunkState *cs;
    GORSE_TRY(tri_state(h, 2, &cs));
    if (counts && h->tri_packed_counts > 0)
        GORSE_HIP_CHECK(hipMemcpyAsync(counts, h->tri_counts.p, (size_t)h->tri_packed_counts * 4, hipMemcpyDefault, h->stream));
    if (entries && h->tri_packed_entries > 0)
        GORSE_HIP_CHECK(hipMemcpyAsync(entries, h->tri_entries.p, (size_t)h->tri_packed_entries * 8, hipMemcpyDefault, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_tri_unpack(gorse_topk *h, int32_t src, const int32_t *counts, int64_t n_counts, const uint64_t *entries,
                                         int64_t n_entries) {
    ChunkState *cs;
    GORSE_TRY(tri_state(h, 2, &cs));
    if (src < 0 || src >= cs->tri_world || src == cs->tri_rank) return fail(GORSE_ERR_INVALID, "bad source rank %d", src);
    const int64_t n = tri_geom(*cs).owned(cs->tri_rank);
    if (n_counts != n) return fail(GORSE_ERR_INVALID, "rank %d owns %lld queries, the message holds %lld", cs->tri_rank, (long long)n, (long long)n_counts);
    if (n == 0) return GORSE_OK;
    if (!counts || (n_entries > 0 && !entries) || n_entries < 0) return fail(GORSE_ERR_INVALID, "NULL message");
    GORSE_TRY(h->tri_in_counts.ensure((size_t)n));
    GORSE_TRY(h->tri_offsets.ensure((size_t)n + 1));
    GORSE_TRY(h->tri_in_entries.ensure((size_t)std::max<int64_t>(n_entries, 1)));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->tri_in_counts.p, counts, (size_t)n * 4, hipMemcpyDefault, h->stream));
    if (n_entries > 0) GORSE_HIP_CHECK(hipMemcpyAsync(h->tri_in_entries.p, entries, (size_t)n_entries * 8, hipMemcpyDefault, h->stream));
    tri_scan_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(h->tri_in_counts.p, n, reinterpret_cast<int64_t *>(h->tri_offsets.p));
    long long total = 0;
    GORSE_HIP_CHECK(hipMemcpyAsync(&total, h->tri_offsets.p + n, 8, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (total != n_entries) return fail(GORSE_ERR_INVALID, "the message's counts add up to %lld entries, it holds %lld", total, (long long)n_entries);
    tri_unpack_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(n, 4), 8192)), dim3(256), 0, h->stream>>>(
        h->fbuf.p, h->fcnt.p, h->tri_in_counts.p, reinterpret_cast<const int64_t *>(h->tri_offsets.p), h->tri_in_entries.p, n, cs->tri_rank, cs->tri_world);
    GORSE_HIP_CHECK(hipGetLastError());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // the caller's buffers may be reused
    return GORSE_OK;
}

extern "C" int32_t gorse_topk_tri_finish(gorse_topk *h, int32_t *idx_out, float *dist_out) {
    ChunkState *cs;
    GORSE_TRY(tri_state(h, 2, &cs));
    GORSE_TRY(chunk_finish(h, *cs));
    // the rows of the queries this rank owns into the caller's nq x k arrays (the others are left alone): one strided copy for
    // the whole blocks, one for a partial last block
    const TriGeom g = tri_geom(*cs);
    const int k = cs->k, W = cs->tri_world, r = cs->tri_rank;
    const int64_t nb = g.blocks_of(r);
    if (nb > 0 && (idx_out || dist_out)) {
        const int64_t last = r + (nb - 1) * W;
        const bool partial = (last + 1) * kSymBQ > g.nq;
        const int64_t full = partial ? nb - 1 : nb;
        const size_t width = (size_t)kSymBQ * k * 4, pitch = width * W, first = (size_t)r * kSymBQ * k;
        if (full > 0) {
            if (idx_out) GORSE_HIP_CHECK(hipMemcpy2DAsync(idx_out + first, pitch, h->res_idx.p + first, pitch, width, (size_t)full, hipMemcpyDeviceToHost, h->stream));
            if (dist_out) GORSE_HIP_CHECK(hipMemcpy2DAsync(dist_out + first, pitch, h->res_dist.p + first, pitch, width, (size_t)full, hipMemcpyDeviceToHost, h->stream));
        }
        if (partial) {
            const size_t at = (size_t)last * kSymBQ * k, bytes = (size_t)(g.nq - last * kSymBQ) * k * 4;
            if (idx_out) GORSE_HIP_CHECK(hipMemcpyAsync(idx_out + at, h->res_idx.p + at, bytes, hipMemcpyDeviceToHost, h->stream));
            if (dist_out) GORSE_HIP_CHECK(hipMemcpyAsync(dist_out + at, h->res_dist.p + at, bytes, hipMemcpyDeviceToHost, h->stream));
        }
    }
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    cs->tri_stage = 3;
    return GORSE_OK;
}

// One process, N handles (one per GPU -- or N on one device: the emulation the tests run): the whole triangle-sharded search in one
// call, the two exchanges as device-to-device copies between the handles (hipMemcpyDefault: peer-to-peer over xGMI where the devices
// can reach each other, staged otherwise).  What a Go master that owns all GPUs of a node calls (integration/go/common/ann/
// bruteforce_hip.go SearchAllSharded); one process per GPU drives gorse_topk_tri_* itself and moves the messages over its own transport.
extern "C" int32_t gorse_topk_tri_all_pairs_local(gorse_topk **hs, int32_t n, int64_t q_begin, int64_t q_end, int32_t k, int32_t *idx_out,
                                                  float *dist_out) {
    if (!hs || n < 1) return fail(GORSE_ERR_INVALID, "no handles");
    for (int r = 0; r < n; r++)
        if (!hs[r]) return fail(GORSE_ERR_INVALID, "handle %d is NULL", r);
    for (int r = 0; r < n; r++) GORSE_TRY(gorse_topk_tri_begin(hs[r], q_begin, q_end, k, r, n));  // every rank's pilots are enqueued ...
    for (int r = 0; r < n; r++) {                                                                // ... before anybody waits for them
        GORSE_TRY(hs[r]->use());
        GORSE_HIP_CHECK(hipStreamSynchronize(hs[r]->stream));
    }
    for (int r = 0; r < n; r++)  // all-gather of the thresholds: every rank fetches the other ranks' slices from where they lie
        for (int s = 0; s < n; s++) {
            if (s == r) continue;
            int64_t lo = 0, hi = 0;
            GORSE_TRY(gorse_topk_tri_slice(hs[s], s, &lo, &hi, nullptr));
            if (hi > lo) GORSE_TRY(gorse_topk_tri_thresholds_put(hs[r], lo, hi, hs[s]->f0.p + lo));
        }
    for (int r = 0; r < n; r++) GORSE_TRY(gorse_topk_tri_sweep(hs[r]));  // enqueued on every device, then awaited
    for (int r = 0; r < n; r++) {
        GORSE_TRY(hs[r]->use());
        GORSE_HIP_CHECK(hipStreamSynchronize(hs[r]->stream));
    }
    for (int src = 0; src < n; src++)  // all-to-all of the foreign lists
        for (int dst = 0; dst < n; dst++) {
            if (dst == src) continue;
            int64_t nc = 0, ne = 0;
            GORSE_TRY(gorse_topk_tri_pack(hs[src], dst, &nc, &ne));
            if (nc > 0) GORSE_TRY(gorse_topk_tri_unpack(hs[dst], src, hs[src]->tri_counts.p, nc, reinterpret_cast<const uint64_t *>(hs[src]->tri_entries.p), ne));
        }
    for (int r = 0; r < n; r++) GORSE_TRY(gorse_topk_tri_finish(hs[r], idx_out, dist_out));  // every rank writes the rows it owns
    return GORSE_OK;
}
