/*
 * gorse_hip_test.h -- test and probe hooks of libgorse_hip.so.  NOT part of the drop-in boundary (include/gorse_hip.h is
 * what the reference's cgo files bind): these switches let the parity tests drive each of two equivalent code paths, let
 * small inputs reach paths that large ones take, and let the probes of scripts/ read phase counters.  Results never
 * depend on them; the product runs with every one at its default.
 */
#ifndef GORSE_HIP_TEST_H
#define GORSE_HIP_TEST_H

#include "gorse_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* exp flavour used by the GORSE_BPR_SEQUENTIAL schedule: 0 = device expf (default),
 * 1 = the float32 FreeBSD/math32 scheme restated in oracle/gorse_oracle.c (orc_exp_restated),
 * which makes device and oracle factors comparable bit for bit. */
void gorse_hip_test_set_exact_exp(int32_t mode);
/* schedule switches of the Hogwild update path (bit 5: no hot-row replicas = the round-1 kernel; bit 7: force the user-run schedule (triplets counting-sorted by user, p_u
 * register-resident over a user's samples), bit 25: the user-run schedule sends only the POSITIVE item's update of a hot item
 * through the replicas (round 2), bit 28: force the per-sample schedule, bit 29: the user sort ranks
 * samples in stream order (single thread; makes a run's order deterministic for the parity test), bit 21: the user-run
 * schedule's chunk preparation WITHOUT user bins (round 4's form: a returning atomic per sample on its user's run counter),
 * bit 22: the chunk preparation on the update stream instead of beside it (kernel timelines: every kernel alone).  Used by
 * scripts/gpu_probe_*.py and the tests; 0 (the default) is the only value the product ever runs with. */
void gorse_hip_test_set_variant(int32_t variant);
/* top-k path choice: 0 = automatic (MFMA sweep for >= 768 queries, any of the three metrics, k <= 255), 1 = always the
 * literal scan (path A), 2 = the MFMA sweep whenever its operands exist.  Both paths return identical results;
 * the hook exists so the parity tests can drive each one. */
void gorse_hip_test_set_topk_path(int32_t path);
/* probe switches of the MFMA sweep: bit 0 = 64 candidate rows per LDS tile, bit 1 = 128 (default: what the library
 * ships with), bit 2 / bit 3 = block-level row-scale bound of the cosine sweep off / on (default: on when all norms
 * are within 2 % of each other), bit 4 = the instrumented twin (see below), bit 5 / bit 6 = compact a candidate list
 * when one of its two sub-lists exceeds 224 / 96 entries (default 128), bit 14 = the history
 * sweep of the tie path in eight row slices whatever the index size (default: one slice per 32768 rows, at most eight),
 * bit 15 = in one slice, bit 16 = the tie replay applies every T = "push +inf, pop" literally instead of first testing
 * whether it leaves the heap as it is (and a wave per query), bit 19 = the tie replay with a wave per query for every query
 * (default: a lane per query), bits 17-18 = timing probes of the main sweep (1: no block ever qualifies, 2: a qualifying
 * block does nothing, 3: it counts its candidates without storing them; the search then returns after the sweep, results
 * undefined), bits 20-22 = the warm start's pilot sample: every 8th / every 32nd row tile instead of every 16th, and without the
 * 1/256 pilot in front (scripts/gpu_probe_topk_c4.py pilot).  Results never depend on the others. */
void gorse_hip_test_set_topk_variant(int32_t variant);
/* (round 6) bit 27 = the tie path's history sweep with 64 instead of 128 queries per workgroup, bit 28 = the tie replay with 64 queries
 * per wave whatever the launch's size, bits 29-30 = the row slices per query block of a triangle-sharded sweep (gorse_topk_tri_*):
 * 1 = one, 2 = two, 3 = four (0: by the launch's size).  Results never depend on them. */
/* variant bit 8 (256) switches the warm start of the sweep off (pilot sweep over every 16th row tile -> initial thresholds,
 * verified by the main sweep; csrc/topk_mfma.hip topk_mfma_search), bit 9 (512) switches it on below its size limit of 2^17
 * rows, bit 10 (1024) gives the pilot a kth of 2 so that most warm starts fail their verification.  This returns how many
 * queries of the last search failed it and were swept again from -inf. */
int32_t gorse_hip_test_topk_resweeps(gorse_topk *h, int64_t *n /*out*/);
/* variant bit 23 (1 << 23) switches the symmetric form of an all-pairs sweep off (the square sweep of rounds 1-4; the symmetric
 * form is taken when the queries are a contiguous range of the stored rows that starts on a 128-row boundary, the sweep is
 * warm-started, and the operands are 32 / 64 / 128 bf16 values deep).  This returns whether the last search's main sweep took it. */
int32_t gorse_hip_test_topk_last_symmetric(gorse_topk *h, int32_t *sym /*out*/);
/* counters of the last symmetric search: [0] queries the pilot left without a threshold (they take the tie path's sweep), [1] warm
 * starts the rescoring could not verify over both lists, [2] foreign lists that overflowed, [3] hits beyond a wave's staging area */
int32_t gorse_hip_test_topk_sym_stats(gorse_topk *h, uint64_t *out4 /*host*/);
/* the per-query flags of the last MFMA search's last chunk (the first n queries) as the host read them behind the rescoring: non-zero =
 * the query left sweep + rescoring undecided and went on to the tie path (ties among its k + 1 best exact distances, a warm start
 * that could not be verified, no pilot threshold, an overflowing list) */
int32_t gorse_hip_test_topk_get_flags(gorse_topk *h, uint8_t *flags /*host*/, int64_t n);
/* after a symmetric sweep: the entries other workgroups appended to each query's foreign list (more than 512 = the list overflowed) */
int32_t gorse_hip_test_topk_get_foreign_counts(gorse_topk *h, int32_t *counts /*host*/, int64_t n);
/* variant bit 24: the search stops behind the pilot sweep (results undefined); this returns the pilot's flags and list lengths */
int32_t gorse_hip_test_topk_get_pilot_state(gorse_topk *h, uint8_t *flags /*host*/, int32_t *counts /*host*/, int64_t n);
/* the warm-start thresholds of the last search's last chunk (the first n queries) */
int32_t gorse_hip_test_topk_get_thresholds(gorse_topk *h, float *out /*host*/, int64_t n);
/* variant bit 4 runs an instrumented twin of the C4-shaped sweep (d = 128 bf16, cosine); this returns its sixteen
 * counters summed over all waves: s_memtime ticks in [0] tile store + prefetch issue, [1] MFMA + epilogues, [2] the
 * candidate paths inside [1], [3] barrier wait, then [4] row blocks examined, [5] row blocks with a candidate, [6]
 * kernel ticks, [7] waves, and the candidate path split into [8] count + exchange, [9] appends, [10] compaction
 * check / compaction, with [11] candidate-path blocks that appended something; the tile top split into [12] wait for the
 * buffer, [13] LDS-DMA issue, [14] wait for the previous tile's DMA, [15] its announcement. */
int32_t gorse_hip_test_get_sweep_profile(gorse_topk *h, uint64_t *out16 /*host*/);
/* 1 = every query of the literal scan (path A) goes through the one-thread-per-query heap kernel; 0 (default) = the queries whose
 * k + 1 smallest distances are pairwise distinct are answered by select_fast_kernel (same results). */
void gorse_hip_test_set_scan_literal(int32_t on);
/* probe: 1 = a gorse_mf handle created afterwards runs its update stream at the device's highest stream priority and its
 * sampler / sort stream at the lowest; 0 (default) = both at the same priority (measured no better: r02_ak). */
void gorse_hip_test_set_stream_priorities(int32_t on);
/* probe: n > 1 = the sampler / sort stream of a gorse_mf handle created afterwards is confined to every n-th CU
 * (hipExtStreamCreateWithCUMask); 0 / 1 = the whole chip (default). */
void gorse_hip_test_set_prep_cu_stride(int32_t n);
/* sparse top-k (csrc/sparse*.h*).  Results never depend on any of these.
 * slots: at most this many single-wave workgroups per launch of sparse_tile_kernel (0 = the library's 16 per CU). */
void gorse_hip_test_set_sparse_slots(int64_t max_slots);
/* groups that a query of ordinary length visits one by one with directly indexed accumulators; the groups behind them are taken
 * several at once with hashed accumulators (csrc/sparse_kernels.hpp, super-visits).  -1 (default) = the leading groups that hold
 * more than their even share of the stored entries; 0 = none (every group through the super-visit loop, which falls back to the
 * direct accumulators where one group alone has too many postings); a value >= the number of groups = no super-visits. */
void gorse_hip_test_set_sparse_head(int32_t groups);
/* probe: a hashed super-visit of the sparse list walk takes at most (accumulators >> cap_shift) postings into its table of
 * (accumulators / 2) slots: 2 (default) = half full at most, 3 = a quarter, ...; outside 2..6 = the default. */
void gorse_hip_test_set_sparse_table(int32_t cap_shift);
/* timing probe (results are garbage): low byte 1 = a whole-query item of the UNSYMMETRIC sparse list walk visits only the row groups up to
 * its own row's -- the postings a symmetric walk of the whole-query items would still meet (sparse); 0 = off
 * (default).  Bits 8..: most workgroups of the dense-vector kernel of the heavy queries, in units of 256 (0 = 1024, the default; results
 * unaffected). */
void gorse_hip_test_set_sparse_probe(int32_t probe);
/* The symmetric form of gorse_sparse_all_pairs over all rows (csrc/sparse_kernels.hpp, SymArgs): mode 0 = never (the walk every other
 * call takes), -1 / 1 = when the call is eligible (default), 2 = likewise, but the front (gorse_hip_test_set_sparse_front) does not deliver.  c1 / c2 / c3 > 0 replace the capacities of the three tiers of foreign lists
 * (tests overflow them on purpose: the rows then take the unsymmetric walk in a second launch); 0 = the defaults.  Results never differ. */
void gorse_hip_test_set_sparse_sym(int32_t mode, int32_t c1, int32_t c2, int32_t c3);
/* handles created AFTERWARDS: the FRONT = the longest rows in a row group of their own when they are fewer than a group holds (phantom
 * scratch ids behind them: csrc/sparse_host.hpp, RowOrder).  1 (default) = the rows longer than 1 .. 2 x the split threshold in force at
 * creation (the largest multiple that leaves 300 rows: gorse_sparse_create), > 1 = the rows longer than this, 0 = plain longest-first
 * numbering.  A symmetric all-pairs pass splits exactly the front's rows into per-group work items. */
void gorse_hip_test_set_sparse_front(int32_t front);
/* the last call of the handle: out[0] = it ran in symmetric form, [1] = rows redone after an overflow, [2] = foreign entries ranked,
 * [3] = the longest foreign list */
void gorse_hip_test_sparse_sym_stats(const gorse_sparse *h, int64_t out[4]);
/* rows per group of a handle created AFTERWARDS (the posting lists are cut by row group, csrc/sparse_kernels.hpp): a power of two
 * in 256 .. 16384; 0 = 2048.  A workgroup holds a group's accumulators: 5.5 bytes of LDS per row. */
void gorse_hip_test_set_sparse_tile(int32_t rows);
/* queries with more than `entries` entries are answered by one work item per row group and a merge instead of one item
 * (default 2048; <= 0 = never): lets small test inputs take that path. */
void gorse_hip_test_set_sparse_split(int64_t entries);
/* of those, queries with more than `entries` entries (default 16384; <= 0 = never) do not walk posting lists: every stored row is
 * scored against a dense copy of the query, one row per lane (sparse_rows_kernel). */
void gorse_hip_test_set_sparse_heavy(int64_t entries);
/* how products reach the LDS accumulators: 1 = ds_add_f32 (no return value, no wait), 0 = load / add / store by the same
 * wave, -1 = the library's choice (ds_add_f32 unless a product of a stored and a query value could fall below 2^-100,
 * where partial sums may be subnormal and the LDS adder's handling of those is not relied upon). */
void gorse_hip_test_set_sparse_atomic(int32_t mode);
/* probe: with on != 0 the following calls of the handle record what every work item (a query, or one group of a long query)
 * did; with out != NULL copies up to cap records of the last call as 16 uint64 each: {start, end (100 MHz ticks), query,
 * group + 1 of a long query (0 = whole query), entries, chunks taken 64 lists at once, their rounds, segments walked one list at a time,
 * groups read back densely, groups read back by re-walking, flattened batches, rows two lists of a batch shared,
 * 10 ns ticks in the 64-lists-at-once path / in the batches / in the read-backs / until the end of the eighth group}.  Returns the number of work items of the last call. */
int64_t gorse_hip_test_sparse_trace(gorse_sparse *h, int32_t on, uint64_t *out /*host or NULL*/, int64_t cap);
/* ALS row-solve choice: 0 = automatic (Gram form: on the fp32 MFMA for nFactors <= 64, als_wide_kernel for 65..128; the residual
 * sweep beyond), 1 = always the residual sweep (the reference's own recurrence), 2 = always the MFMA Gram form (nFactors <= 64).
 * Both meet the 1e-4 relative bar; the hook lets the parity tests drive each one.  Probe bits on top of the choice: 4 = the
 * waves of als_row_kernel's workgroup accumulate and solve in lockstep, 8 = als_wide_kernel builds G by fused multiply-adds
 * (round 2) instead of the fp32 MFMA, 16 / 32 = timing probes of als_wide_kernel (no sweep / S not added: results undefined),
 * 64 = the first form of the Gram kernels' gather stage whatever the shape (the product takes it for factor matrices of
 * >= 4 GB or >= 2^24 rows; otherwise 32-bit offsets from a scalar base, see csrc/als.hip gram_load_stage32).
 * Which MFMA form the Gram of a row takes (csrc/als.hip als_gram_mode; default: nFactors 32 / 64 on the bf16 MFMA over three-way
 * split values, 16 / 48 on the fp32 MFMA in 16 x 16 tiles, anything else or without the fast gather stage fp32 32 x 32 tiles):
 * 1024 = no bf16 form (32 / 64 take the 16 x 16 fp32 tiles), 128 = no 16 x 16 tiles either (1024 | 128: everything in 32 x 32
 * fp32 tiles, the form of rounds 1-3); 256 = eight waves per workgroup where twelve are the default (nFactors <= 32);
 * 2048 = als_long_solve_kernel adds a long row's partial Gram matrices itself however many there are (default: rows of more
 * than eight chunks go through als_partial_reduce_kernel first; the two are equal in every bit). */
void gorse_hip_test_set_als_path(int32_t path);
/* thresholds of the Gram-form row plan, for handles created AFTERWARDS: rows longer than long_row feedbacks
 * are cut into chunks of `chunk` entries; <= 0 restores the library's choice (round 5: by the side's size -- the even
 * share of one of ~4096 wave slots, a power of two between 256 (512 from nFactors 64 on) and 4096; chunk = the
 * threshold).  Lets small test inputs exercise the long-row path. */
void gorse_hip_test_set_als_plan(int32_t long_row, int32_t chunk);
/* probe: enable != 0 makes als_row_kernel stamp its phases with s_memtime; out16 (may be NULL) receives, for the last
 * user half-sweep and then the last item half-sweep, ticks summed over the waves in [0] Gram accumulation (gathers +
 * MFMA), [1] M to LDS, [2] the d-step solve, then [3] rows, [4] feedback entries, [5] kernel ticks, [6] waves, [7] 0. */
int32_t gorse_hip_test_als_profile(gorse_mf *h, int32_t enable, uint64_t *out16 /*host or NULL*/);
/* probe: samples per chunk of a gorse_mf handle created AFTERWARDS (0 = the library's choice: 32 x users, clamped to
 * [4M, 128M]); the user-run schedule applies a chunk at a time. */
void gorse_hip_test_set_bpr_chunk(int64_t samples);
/* which item updates of the user-run BPR kernel may take the STORE route (csrc/bpr.hip ST_*): bit 0 = the NEGATIVE item of a
 * sample, when its class is "cold", is updated by one write-through store of fma(t, lr, row) instead of d atomic dwords (the
 * reference's own unlocked write, model.go:478-488; its row is then gathered one sample ahead instead of two), bit 1 = the
 * positive item likewise, bit 2 = the store adds to a row re-read in the same iteration instead of the gathered snapshot;
 * < 0 = the library's default.  Which items are cold is fixed at gorse_mf_create (gorse_hip_test_set_bpr_cold_window). */
void gorse_hip_test_set_bpr_store_mode(int32_t store_mode);
/* `make probe-lib` builds only (a measured dead end, csrc/bpr.hip SEG): n = 2..8 segments per user run of the user-run schedule at
 * nFactors <= 32; 0 / 1 = the plain form.  The shipped library ignores it. */
void gorse_hip_test_set_bpr_user_segments(int32_t segments);
/* probe builds (make probe-lib) only: which software pipeline of the atomics-only user-run kernel runs.  Bits 0..7: 0 = the shipped
 * one, 1..6 = (rows gathered G samples ahead, ids IA ahead) = (2,4) (3,5) (3,6) (4,6) (4,8) (6,9) of the same kernel, 10..14 = the ring
 * kernel without register rotation (ring size / id lead) = 3/1, 4/2, 6/3, 8/4, 6/2; bits 8..19: threads per workgroup of the ring
 * kernel (0 = 256); bits 20..: how many of a wave's four 16-lane groups work (0 = 4).  Ignored by the product build. */
void gorse_hip_test_set_bpr_user_depth(int32_t which);
/* floats.MM (csrc/sgemm.hip): 1 = the NN / TN / TT chains on the vector ALU whatever the shape (default: on the fp32 MFMA from 64 x 64
 * results on; both are the same l-ascending fmaf chain, bit for bit).  The second hook returns the kernel time of the last
 * gorse_hip_sgemm call in milliseconds (hipEvents around the launch, copies excluded): bench.py's `mm` object. */
void gorse_hip_test_set_sgemm_valu(int32_t on);
double gorse_hip_test_sgemm_last_ms(void);
/* 1 = this is a `make probe-lib` build (csrc/Makefile, -DGORSE_PROBE): it also carries the instrumented twin of the top-k sweep, the
 * wave-per-query tie replay, the sparse kernel's trace instantiation and the positive-side / re-reading forms of the BPR store
 * route.  The library `make all` ships (0) answers the switches that would select those with its own nearest form. */
int32_t gorse_hip_test_probe_build(void);
/* test hook: runs the user-run schedule's preparation of ONE chunk (user draws, counting sort of the sample ids by user, item
 * draws by run: csrc/bpr.hip launch_prepare_users) for samples [sample_base, sample_base + n) and returns the run offsets
 * (off[u] .. off[u + 1] = the positions of user u's samples; off[U] .. off[U + 1] = samples whose user draw failed) and the
 * (positive, negative) pair at every position; a pair of -1 = no negative found.  The multiset of triplets equals what
 * gorse_bpr_sample_triplets returns for the same range. */
int32_t gorse_hip_test_bpr_prepare_chunk(gorse_mf *h, int64_t n, uint64_t seed, uint64_t epoch, int64_t sample_base,
                                         int32_t *off /*U + 2, host*/, int32_t *si /*n, host*/, int32_t *sj /*n, host*/);
/* items expected to be touched (as a positive: their share of the feedback; as a negative: 1 / items) less than once per
 * `samples` samples are of class "cold" in handles created AFTERWARDS; 0 = no cold items (every update an atomic). */
void gorse_hip_test_set_bpr_cold_window(int64_t samples);

#ifdef __cplusplus
}
#endif
#endif /* GORSE_HIP_TEST_H */
