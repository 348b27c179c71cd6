/*
 * gorse_hip.h -- C ABI of libgorse_hip.so, the MI355X (gfx950) implementation of
 * Gorse's collaborative-filtering training and exact embedding top-k hot path.
 *
 * This is the drop-in boundary: exactly what a cgo file built with
 * `//go:build cgo && hip` inside the reference's packages model/cf, common/ann and
 * common/floats would bind (see INTEGRATION.md for the Go stubs).  The reference has
 * no FFI for this path today; its precedent is the cgo BLAS binding
 * common/blas/blas_openblas.go:19-26 (pointer to the first element of a flat Go
 * slice, synchronous call, nothing retained) and build-tag twins such as
 * model/ctr/fm.go vs fm_xla.go.  The same conventions hold here:
 *
 *   - plain C, no C++/torch types; every pointer marked "host" is caller-owned host
 *     memory that is only read/written for the duration of the call (never retained,
 *     so Go's cgo pointer rules are met); pointers marked "device" are HIP device
 *     addresses on the handle's device;
 *   - every function returns int32: 0 = ok, <0 = error (gorse_hip_last_error() gives a
 *     thread-local message).  Where the reference panics (slice length mismatch) or
 *     returns an error (ann.Bruteforce.SearchIndex out of range) the matching code is
 *     returned and nothing is modified;
 *   - one handle = one GPU.  Multi-GPU = one handle per GPU, in one process (the Go master: one goroutine
 *     per GPU) or in several; the exchange between the replicas runs INSIDE the library over RCCL
 *     (gorse_comm_*, gorse_mf_item_allreduce, gorse_mf_rows_allgather below).  The gorse_mf_item_delta_*
 *     calls remain for a caller that brings its own collective;
 *   - calls on one handle must be serialised by the caller (the Go side holds a mutex,
 *     as logics/vector_writer.go does); different handles are independent;
 *   - long calls poll a caller-supplied cancel flag between kernel launches, the
 *     equivalent of ctx.Err() checks in common/parallel/parallel.go:36-38.
 */
#ifndef GORSE_HIP_H
#define GORSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GORSE_HIP_ABI_VERSION 1

/* return codes */
#define GORSE_OK 0
#define GORSE_ERR_INVALID (-1)   /* bad argument (a Go panic in the reference, e.g. floats.go:96-99) */
#define GORSE_ERR_HIP (-2)       /* HIP runtime / launch failure */
#define GORSE_ERR_CANCELLED (-3) /* cancel flag observed: ctx.Err() != nil, model/cf/model.go:490-493 */
#define GORSE_ERR_NO_DEVICE (-4) /* no usable gfx950 device */
#define GORSE_ERR_RANGE (-5)     /* "index out of range", common/ann/bruteforce.go:41-43 */
#define GORSE_ERR_NOMEM (-6)

/* BPR update schedules (gorse_bpr_epoch / gorse_bpr_apply_triplets `mode`) */
#define GORSE_BPR_HOGWILD_ATOMIC 0 /* Jobs > 1, NO lost item update: all samples of a chunk in flight.  User rows are  *
                                    * updated exactly (one group owns a user's run), every item row by fp32 atomics     *
                                    * (hot items through replica rows)                                                   */
#define GORSE_BPR_SEQUENTIAL 1     /* dependency-levelled: bit-faithful to the reference with Jobs = 1  */
#define GORSE_BPR_HOGWILD_RACY 2   /* write-through load/fma/store, lost updates like the CPU Hogwild   */
#define GORSE_BPR_HOGWILD_STORES 3 /* the schedule Fit runs with Jobs > 1 (the reference's own Hogwild semantics):       *
                                    * GORSE_BPR_HOGWILD_ATOMIC, except that the NEGATIVE item of a sample, when it is a  *
                                    * COLD item (expected to be touched less than once per cold window of samples, see   *
                                    * gorse_mf_set_bpr_cold_window; default 32768), is updated by the reference's own    *
                                    * unlocked load / fma / store (model.go:478-488): such a write can overwrite a       *
                                    * concurrent update of the same row, as two of the reference's workers can (measured *
                                    * at C3: 4 % of the cold rows' updates, NDCG@10 unchanged).      *
                                    * Where the handle has no cold item, or runs the per-sample schedule, it IS mode 0.  */

/* top-k element types and metrics */
#define GORSE_DTYPE_F32 0
#define GORSE_DTYPE_BF16 1      /* uint16 = upper half of the fp32 word, common/bfloats/bfloats.go:24-30 */
#define GORSE_METRIC_NEG_DOT 0  /* distance = -floats.Dot        (logics/cf.go:32-34)                     */
#define GORSE_METRIC_EUCLIDEAN 1 /* distance = floats.Euclidean  (common/ann/ann_test.go)                  */
#define GORSE_METRIC_COSINE 2   /* distance = 1 - a.b/(|a||b|)   (storage/vectors/database.go:29-33)      */
#define GORSE_METRIC_EUCLIDEAN_BF16 3 /* distance = bfloats.Euclidean (common/bfloats/bfloats.go:49-57; the summation order
                                       * of src/bfloats_avx512.c:26-59: 16 unfused partials added one after the other, fused
                                       * scalar tail); bf16 indexes only                                                    */

typedef struct gorse_mf gorse_mf;     /* one matrix-factorisation model resident on one GPU */
typedef struct gorse_topk gorse_topk; /* one exact nearest-neighbour index resident on one GPU */

/* ---- library ------------------------------------------------------------------- */
int32_t gorse_hip_abi_version(void);
const char *gorse_hip_last_error(void);
int32_t gorse_hip_device_count(int32_t *n);

/* ---- cf.MatrixFactorization state (model/cf/model.go:118-127) ------------------------
 * U users, I items, d = nFactors.  user_indptr[U+1] / user_indices[nnz] are
 * dataset.CFSplit.GetUserFeedback() flattened IN STORED ORDER (dataset/dataset.go:231-240);
 * item_indptr / item_indices are GetItemFeedback() likewise and may be NULL when only BPR
 * is run.  The library keeps its own device copies (plus a row-sorted copy of the user
 * rows for the negative sampler's membership test, model.go:461-468). */
int32_t gorse_mf_create(gorse_mf **h, int32_t device, int64_t U, int64_t I, int32_t d,
                        const int64_t *user_indptr /*host*/, const int32_t *user_indices /*host*/,
                        const int64_t *item_indptr /*host or NULL*/, const int32_t *item_indices /*host or NULL*/);
int32_t gorse_mf_destroy(gorse_mf *h);

/* BaseMatrixFactorization.UserFactor / ItemFactor as flat row-major U*d / I*d float32.
 * Either pointer may be NULL to skip that matrix. (Init draws stay on the Go side:
 * model/cf/model.go:532-540, common/util/random.go:45-60.) */
int32_t gorse_mf_set_factors(gorse_mf *h, const float *P /*host*/, const float *Q /*host*/);
int32_t gorse_mf_get_factors(gorse_mf *h, float *P /*host*/, float *Q /*host*/);

/* internalPredict for n (user, item) index pairs, model/cf/model.go:195-203:
 * out[t] = floats.Dot(UserFactor[u[t]], ItemFactor[items[t]]) in the reference's AVX512
 * operation order, 0 when either index is negative. */
int32_t gorse_mf_score(gorse_mf *h, const int32_t *u /*host*/, const int32_t *items /*host*/, int64_t n,
                       float *out /*host*/);

/* cf.Rank for many users at once, model/cf/evaluator.go:162-169: user users[t] ranks the
 * candidates cand[cand_indptr[t] .. cand_indptr[t+1]) through heap.TopKFilter(topk)
 * (common/heap/filter.go:23-59, Go container/heap tie behaviour).  rank_out is
 * n_users*topk, padded with -1; rank_len[t] = number of valid entries. */
int32_t gorse_mf_rank(gorse_mf *h, int64_t n_users, const int32_t *users /*host*/, const int64_t *cand_indptr /*host*/,
                      const int32_t *cand /*host*/, int32_t topk, int32_t *rank_out /*host*/, int32_t *rank_len /*host*/);

/* dataset.SampleUserNegatives on the device (dataset/dataset.go:242-253 through RandomGenerator.SampleInt32,
 * common/util/random.go:108-132): for EVERY user num_candidates distinct items outside (the user's feedback in the test split
 * given here, union the user's train feedback the handle holds), in draw order -- or ALL remaining items ascending when no
 * more than num_candidates are left (random.go:115-121).  User u draws from its own Philox4x32-10 stream keyed by
 * (seed, "neg", u) through Go's Int31n (the reference's single math/rand stream, seeded 0, is not reproducible: SURVEY.md
 * 8c); the reference passes seed 0.  neg_out (host or NULL) receives U * num_candidates items padded with -1, neg_len (host
 * or NULL) the counts.  The call also leaves, resident on the device, the candidate lists Evaluate ranks
 * (model/cf/evaluator.go:47-53: for each user WITH test feedback, ascending, the test items followed by the negatives):
 * gorse_mf_rank_resident ranks them without any upload -- the Evaluate between two epochs of a Fit. */
int32_t gorse_mf_sample_user_negatives(gorse_mf *h, const int64_t *test_indptr /*host, U+1*/,
                                       const int32_t *test_indices /*host*/, int32_t num_candidates, uint64_t seed,
                                       int32_t *neg_out /*host or NULL*/, int32_t *neg_len /*host or NULL*/);
/* how many users have test feedback / how many candidates their lists hold (sizes of gorse_mf_rank_resident's outputs) */
int32_t gorse_mf_resident_candidates(gorse_mf *h, int64_t *n_users /*out*/, int64_t *n_candidates /*out*/);
/* A number that changes with every gorse_mf_sample_user_negatives on this handle (0 = no lists resident).  A handle may be lent to
 * several models in turn (host/gorse_cf.hpp ResidentDataset): a model remembers the number its own sampling returned and ranks
 * the resident lists only while it still reads the same one. */
int32_t gorse_mf_resident_generation(gorse_mf *h, uint64_t *generation /*out*/);
/* gorse_mf_rank over the resident candidate lists: users_out (host or NULL) n_users ids, rank_out n_users * topk padded with
 * -1, rank_len n_users. */
int32_t gorse_mf_rank_resident(gorse_mf *h, int32_t topk, int32_t *users_out /*host or NULL*/, int32_t *rank_out /*host*/,
                               int32_t *rank_len /*host*/);

/* ---- BPR, model/cf/model.go:446-494 ------------------------------------------------------
 * One call = n_samples SGD steps (the reference does CountFeedback() per epoch).
 * Sampling (model.go:449-468) runs on the device from a counter-based Philox4x32-10 stream
 * keyed by (seed, epoch, sample_base + sample index) and consumed through Go's Int31n
 * algorithm; gorse_bpr_sample_triplets returns exactly the triplets an epoch call with the
 * same (seed, epoch, sample_base) applies, so a caller (or a test) can replay them.
 * loss_out (may be NULL) receives sum log1p(exp(-diff)), the reference's unused `cost`. */
int32_t gorse_bpr_epoch(gorse_mf *h, int64_t n_samples, float lr, float reg, uint64_t seed, uint64_t epoch,
                        int64_t sample_base, int32_t mode, const volatile int32_t *cancel /*host or NULL*/,
                        double *loss_out /*host or NULL*/);
/* Same, but only enqueues the work on the handle's stream (hogwild modes only). */
int32_t gorse_bpr_epoch_enqueue(gorse_mf *h, int64_t n_samples, float lr, float reg, uint64_t seed, uint64_t epoch,
                                int64_t sample_base, int32_t mode);
/* The cold window of THIS handle (GORSE_BPR_HOGWILD_STORES): item i is cold when (its share of the training feedback + 1 / I) x
 * samples < 1, i.e. when fewer than one touch of its row is expected per `samples` samples in flight.  gorse_mf_create starts
 * from 32768; 0 = no item is cold (mode 3 is then mode 0); takes effect from the next epoch call.  Returns the number of cold
 * items through n_cold (may be NULL). */
int32_t gorse_mf_set_bpr_cold_window(gorse_mf *h, int64_t samples, int64_t *n_cold /*out, may be NULL*/);
/* Which form of the GORSE_BPR_HOGWILD_ATOMIC schedule this handle runs: 1 = user runs (the chunk's samples are
 * counting-sorted by user and one 16-lane group applies all samples of a user with p_u in registers; chosen when
 * there are >= 4096 users and nFactors is 8/16/32/64/128), 0 = one group per sample.  Both apply exactly the triplets
 * gorse_bpr_sample_triplets returns, in a different (Hogwild-legal) order; GORSE_BPR_SCHEDULE=users|samples in the
 * environment overrides the choice. */
int32_t gorse_mf_bpr_schedule(gorse_mf *h, int32_t *user_runs /*out*/);
int32_t gorse_bpr_sample_triplets(gorse_mf *h, int64_t n, uint64_t seed, uint64_t epoch, int64_t sample_base,
                                  int32_t *u /*host*/, int32_t *i /*host*/, int32_t *j /*host*/);
/* Apply a host-supplied triplet stream (test hook and replay path). Triplets with a
 * negative index are skipped. */
int32_t gorse_bpr_apply_triplets(gorse_mf *h, const int32_t *u /*host*/, const int32_t *i /*host*/,
                                 const int32_t *j /*host*/, int64_t n, float lr, float reg, int32_t mode);

/* ---- ALS (eALS), model/cf/model.go:641-738 ------------------------------------------------
 * One call = one epoch: S = sum q q^T over items with feedback, user sweep, S = sum p p^T
 * over users with feedback, item sweep.  Needs item_indptr/item_indices.
 * Arithmetic: fp32 in, fp32 out, every sum in fp32.  For nFactors 32, 64 and 65..128 the PRODUCTS of the per-row Gram matrices are
 * formed on the bf16 matrix unit from an exact three-way split of each float (hi + mid + lo, all bf16): six exact partial products
 * per pair, the three smallest (below 2^-23 of the product) dropped -- as close to a float64 evaluation as the fp32 matrix unit is
 * (DESIGN.md section 4).  GORSE_ALS_GRAM=fp32 in the environment keeps every product on the fp32 unit (about 1.2x the epoch time). */
int32_t gorse_als_epoch(gorse_mf *h, float weight, float reg, const volatile int32_t *cancel /*host or NULL*/);
/* Row-sharded ALS (one process per GPU): every process holds the whole dataset and both factor
 * matrices, solves only user rows [u_begin,u_end) and item rows [i_begin,i_end) (rows are independent inside a
 * half-sweep: model.go:659, 707 hand them to parallel.Parallel), and the caller all-gathers the row blocks after
 * each half.  set_ranges rebuilds the row plans (default after create: all rows); half_epoch(side) = S over ALL
 * rows of the other side + the sweep of the owned rows of `side` (0 = users, model.go:645-690; 1 = items,
 * :693-738); gorse_als_epoch = half 0 then half 1 on the current ranges. */
int32_t gorse_als_set_ranges(gorse_mf *h, int64_t u_begin, int64_t u_end, int64_t i_begin, int64_t i_end);
int32_t gorse_als_half_epoch(gorse_mf *h, int32_t side, float weight, float reg);
/* Same, but only enqueues the kernels on the handle's stream (gorse_mf_rows_allgather is ordered behind them on that stream;
 * gorse_mf_synchronize ends the epoch): what a caller driving several handles from one thread uses, so that the devices
 * solve their row ranges at the same time. */
int32_t gorse_als_half_epoch_enqueue(gorse_mf *h, int32_t side, float weight, float reg);
/* factor rows [begin,end) of side 0 (P) / 1 (Q) -> / <- a device buffer owned by the caller */
int32_t gorse_mf_rows_export(gorse_mf *h, int32_t side, int64_t begin, int64_t end, float *dst /*device*/);
int32_t gorse_mf_rows_import(gorse_mf *h, int32_t side, int64_t begin, int64_t end, const float *src /*device*/);

/* ---- multi-GPU exchange (one process per GPU; Q replicated, users sharded) ------------------
 * mark:   Q_sync <- Q
 * export: dst[I*d] <- Q - Q_sync                 (device buffer owned by the caller)
 * import: Q <- Q_sync + src ; Q_sync <- Q        (src = all-reduced sum of every rank's export) */
int32_t gorse_mf_item_sync_mark(gorse_mf *h);
int32_t gorse_mf_item_delta_export(gorse_mf *h, float *dst /*device*/);
int32_t gorse_mf_item_delta_import(gorse_mf *h, const float *src /*device*/);
/* ---- the same exchanges with RCCL behind the boundary ------------------------------------------------
 * the reference trains in ONE process and one goroutine (master/tasks.go:879-1034), so the
 * multi-GPU path has to be callable from there: a gorse_comm is one rank of an RCCL communicator owned by this
 * library (librccl is dlopen'ed on first use).  Two ways to get the ranks:
 *   one process, N GPUs  : gorse_comm_create_local(comms, devices, N)  -- one handle + one communicator per device,
 *                          every exchange call below receives all N pairs (issued as one RCCL group);
 *   one process per GPU  : rank 0 calls gorse_comm_unique_id and ships the 128 bytes to the others (bench.py: a
 *                          torch.distributed broadcast; Go: whatever RPC the deployment has), every rank calls
 *                          gorse_comm_create(id, world, rank, device) and passes its ONE (handle, communicator) pair.
 * The collectives are enqueued on the handles' own streams between the kernels that fill and consume their buffers:
 * an exchange synchronises nothing with the host.
 *   gorse_mf_item_allreduce : Q <- Q_sync + sum over ranks (Q - Q_sync); Q_sync <- Q   (one all-reduce of I*d fp32;
 *                             gorse_mf_item_sync_mark once before the first epoch)
 *   gorse_mf_rows_allgather : after gorse_als_half_epoch(side): rank r's rows [row_splits[r], row_splits[r+1]) of P (side 0)
 *                             or Q (side 1) reach every replica (one broadcast per owner, grouped; (U or I)*d fp32)
 *   gorse_comm_allreduce_f32: n host floats summed over the ranks in place (metric partial sums of a sharded Evaluate);
 *                             one process per GPU only -- a process holding several ranks uses gorse_comm_allreduce_f32_local */
typedef struct gorse_comm gorse_comm;
#define GORSE_COMM_ID_BYTES 128
int32_t gorse_comm_unique_id(uint8_t *id /*host, GORSE_COMM_ID_BYTES*/);
int32_t gorse_comm_create(gorse_comm **c, const uint8_t *id /*host*/, int32_t world, int32_t rank, int32_t device);
int32_t gorse_comm_create_local(gorse_comm **comms /*out: n*/, const int32_t *devices /*n*/, int32_t n);
int32_t gorse_comm_destroy(gorse_comm *c);
int32_t gorse_comm_info(gorse_comm *c, int32_t *world /*out*/, int32_t *rank /*out*/);
int32_t gorse_mf_item_allreduce(gorse_mf *const *handles, gorse_comm *const *comms, int32_t n);
int32_t gorse_mf_rows_allgather(gorse_mf *const *handles, gorse_comm *const *comms, int32_t n, int32_t side,
                                const int64_t *row_splits /*host, world + 1*/);
int32_t gorse_comm_allreduce_f32(gorse_comm *c, float *buf /*host, in place*/, int64_t n);
/* gorse_comm_allreduce_f32 blocks until EVERY rank has called it: a process that owns several ranks
 * (gorse_comm_create_local) calls this instead, once, with all of them: bufs[i] = rank i's n floats. */
int32_t gorse_comm_allreduce_f32_local(gorse_comm *const *comms, int32_t n_comms, float *const *bufs /*host, in place*/, int64_t n);
/* GORSE_OK when RCCL can be opened in this process.  gorse_comm_create is a collective initialisation: every rank checks
 * this first and the ranks agree on the answers (over whatever carried the unique id) before any of them enters it. */
int32_t gorse_comm_available(void);
/* Raw device addresses of the resident factor matrices (row-major U*d, I*d). */
int32_t gorse_mf_device_ptrs(gorse_mf *h, float **P /*out: device*/, float **Q /*out: device*/);

/* ---- stream / measurement ---------------------------------------------------------------------
 * All work of a handle runs on the handle's own HIP stream(s).  Profiling mode brackets every
 * launch of the dominant kernels with hipEvents on the stream they run on and accumulates
 * (count, milliseconds) per kernel class; bench.py reads these for the roofline line. */
int32_t gorse_mf_synchronize(gorse_mf *h);
/* Epoch pacing for a Fit loop that ENQUEUES its epochs between two evaluations (gorse_bpr_epoch_enqueue).  Replaces two things the
 * reference's loop has for free (model/cf/model.go:446-503): it checks ctx per sample (:449), and it logs the epoch's duration as
 * fit_time (:496-503).
 *   gorse_mf_epoch_throttle: returns once at most max_in_flight of the epochs issued so far are unfinished on the device (0 = all
 *     done), looking at *cancel (host, may be NULL) every ~20 us while it waits: GORSE_ERR_CANCELLED as soon as the flag is set (the
 *     caller then drains with gorse_mf_synchronize, at most max_in_flight + 1 epochs).  A Fit that calls it with 2 in front of every
 *     enqueued epoch still has the next epoch's preparation running under the current update kernel, and sees a cancel within
 *     two epochs instead of within Verbose.
 *   gorse_mf_epoch_times: the DEVICE time of the BPR epochs that have finished since the last reset -- per epoch from the moment the
 *     update stream reaches it (= the end of the previous epoch's last update kernel when epochs follow each other: an epoch enqueued
 *     while its predecessor is in flight takes that one's end event as its begin, so the times of back-to-back epochs add up to the
 *     stream's time) to the end of its own last update kernel: hipEvents on the handle's stream, no host clock -- and how many epochs
 *     are still in flight. */
int32_t gorse_mf_epoch_throttle(gorse_mf *h, int32_t max_in_flight, const volatile int32_t *cancel /*host or NULL*/);
int32_t gorse_mf_epoch_times(gorse_mf *h, int64_t *epochs /*out*/, double *total_ms /*out*/, int64_t *in_flight /*out, may be NULL*/,
                             int32_t reset);
int32_t gorse_mf_set_profiling(gorse_mf *h, int32_t on);
#define GORSE_PROF_BPR_UPDATE 0
#define GORSE_PROF_BPR_SAMPLE 1 /* the sampler kernels (user draws; item draws by run) */
#define GORSE_PROF_ALS_SWEEP 2
#define GORSE_PROF_ALS_GRAM 3
#define GORSE_PROF_BPR_SORT 4 /* counting sort of a chunk's sample ids by USER (scan of the run counters + scatter) */
#define GORSE_PROF_COMM 5 /* the RCCL collectives of gorse_mf_item_allreduce / gorse_mf_rows_allgather          */
#define GORSE_PROF_NCLASSES 6
int32_t gorse_mf_get_profile(gorse_mf *h, int32_t kernel_class, int64_t *launches, double *total_ms);
int32_t gorse_mf_reset_profile(gorse_mf *h);

/* ---- exact top-k: ann.Index / ann.Bruteforce, common/ann/ann.go:21-25, bruteforce.go:24-83 ----
 * X is N x d row-major, float32 or bf16 (uint16).  Search results follow the reference
 * exactly: the k smallest distances kept in a max-heap (pq.go), Reverse(), popped ascending;
 * ties resolved as Go's container/heap resolves them; prune0 drops results with distance <= 0.
 * idx_out / dist_out are nq*k, padded with -1 / +inf; count_out[t] = valid entries. */
int32_t gorse_topk_create(gorse_topk **h, int32_t device, int64_t N, int32_t d, int32_t dtype, int32_t metric,
                          const void *X /*host*/);
int32_t gorse_topk_destroy(gorse_topk *h);
/* Bruteforce.SearchIndex for nq stored vectors q[t] (the vector itself is excluded, i != q). */
int32_t gorse_topk_search_index(gorse_topk *h, const int64_t *q /*host*/, int64_t nq, int32_t k, int32_t prune0,
                                int32_t *idx_out /*host*/, float *dist_out /*host*/, int32_t *count_out /*host*/);
/* Bruteforce.SearchVector for nq query vectors (nq x d, same dtype as the index). */
int32_t gorse_topk_search_vector(gorse_topk *h, const void *qv /*host*/, int64_t nq, int32_t k, int32_t prune0,
                                 int32_t *idx_out /*host*/, float *dist_out /*host*/, int32_t *count_out /*host*/);
/* SearchIndex for every stored vector q in [q_begin, q_end): the item-to-item bulk build.
 * Results stay on the device unless host pointers are given (either may be NULL). */
int32_t gorse_topk_all_pairs(gorse_topk *h, int64_t q_begin, int64_t q_end, int32_t k, int32_t *idx_out /*host or NULL*/,
                             float *dist_out /*host or NULL*/);
/* ---- triangle-sharded all-pairs search over W GPUs (top-k row) -----------------------------------------------------
 * gorse_topk_all_pairs over the stored rows [q_begin, q_end) takes the SYMMETRIC form of the sweep: every score of the square block
 * serves both of its queries.  Sharding the QUERY ROWS over W ranks keeps that saving on each rank's diagonal square only; these
 * calls shard the TRIANGLE instead -- rank r sweeps the query blocks (512 queries) C with C % W == r -- so the W ranks together do
 * exactly the single-rank sweep.  One handle per rank, every rank holds the whole index; the calls of one search, in order:
 *   gorse_topk_tri_begin            the search's buffers + the PILOT sweeps of this rank's slice of the queries (slice r of W, whole
 *                                   blocks); GORSE_ERR_INVALID when the search has no symmetric form (shard its query rows then)
 *   gorse_topk_tri_slice            which queries a rank's pilots cover, how many queries it owns (arithmetic: any rank can be asked)
 *   gorse_topk_tri_thresholds_get / _put   the pilot thresholds of a slice out of / into the handle: an ALL-GATHER of 4 bytes per
 *                                   query, every rank must hold all of them before ...
 *   gorse_topk_tri_sweep            ... the main sweep of this rank's blocks: own lists, and foreign lists for EVERY earlier block's rows
 *   gorse_topk_tri_pack(dest) / _pack_read   the foreign lists this rank holds for the queries `dest` owns, as one message: a count per
 *                                   owned query (-1: the sender's part overflowed), then the (key, row) entries end to end
 *   gorse_topk_tri_unpack(src, ...) a message from `src` appended to this rank's own queries' foreign lists: an ALL-TO-ALL of ~130
 *                                   entries x 8 bytes per query in all
 *   gorse_topk_tri_finish           exact rescoring + tie path for the queries this rank owns; their rows are written into the
 *                                   caller's nq x k arrays (the other rows are left alone; NULL = results stay on the device)
 * The exchange itself is the caller's (integration/go/common/ann/bruteforce_hip.go: RCCL; gorse_amd/dist.py: torch.distributed,
 * or host memory when the ranks are emulated on one device): every buffer that crosses this boundary may be HOST OR DEVICE memory.
 * Results are those of gorse_topk_all_pairs, row for row, bit for bit (tests/test_gpu_topk_tri.py). */
int32_t gorse_topk_tri_begin(gorse_topk *h, int64_t q_begin, int64_t q_end, int32_t k, int32_t rank, int32_t world);
int32_t gorse_topk_tri_slice(gorse_topk *h, int32_t rank, int64_t *lo /*out*/, int64_t *hi /*out*/, int64_t *owned /*out*/);
int32_t gorse_topk_tri_thresholds_get(gorse_topk *h, int64_t lo, int64_t hi, float *dst /*host or device*/);
int32_t gorse_topk_tri_thresholds_put(gorse_topk *h, int64_t lo, int64_t hi, const float *src /*host or device*/);
int32_t gorse_topk_tri_sweep(gorse_topk *h);
int32_t gorse_topk_tri_pack(gorse_topk *h, int32_t dest, int64_t *n_counts /*out*/, int64_t *n_entries /*out*/);
int32_t gorse_topk_tri_pack_read(gorse_topk *h, int32_t *counts /*host or device, n_counts*/, uint64_t *entries /*host or device, n_entries*/);
int32_t gorse_topk_tri_unpack(gorse_topk *h, int32_t src, const int32_t *counts, int64_t n_counts, const uint64_t *entries,
                              int64_t n_entries);
int32_t gorse_topk_tri_finish(gorse_topk *h, int32_t *idx_out /*host nq x k or NULL*/, float *dist_out /*host nq x k or NULL*/);
/* ONE process that holds all n handles (one per GPU of a node -- the Go master's mode -- or n on one device: the emulation the tests
 * run): the whole sequence above in one call, both exchanges as device-to-device copies between the handles.  hs[r] is rank r; every
 * handle holds the same index.  idx_out / dist_out: all nq x k rows (every rank writes the rows it owns), or NULL. */
int32_t gorse_topk_tri_all_pairs_local(gorse_topk **hs, int32_t n, int64_t q_begin, int64_t q_end, int32_t k, int32_t *idx_out /*host or NULL*/,
                                       float *dist_out /*host or NULL*/);
int32_t gorse_topk_synchronize(gorse_topk *h);
#define GORSE_PROF_TOPK_SCORE 0   /* path A: dist_kernel (pair-at-a-time scan in the reference's order)      */
#define GORSE_PROF_TOPK_RESCORE 1 /* path A: select_fast_kernel (+ the literal container/heap select_kernel)    */
#define GORSE_PROF_TOPK_SWEEP 2   /* path B: topk_sweep_kernel (bf16 MFMA candidate sweep + threshold filter) */
#define GORSE_PROF_TOPK_SELECT 3  /* path B: topk_rescore_kernel (exact rescoring + ranking of the lists)     */
#define GORSE_PROF_TOPK_HIST 4    /* path B: history sweep of the queries with ties in their top k+1           */
#define GORSE_PROF_TOPK_REPLAY 5  /* path B: topk_tie_sort_kernel + topk_tie_replay_kernel (literal heap replay) */
int32_t gorse_topk_set_profiling(gorse_topk *h, int32_t on);
int32_t gorse_topk_get_profile(gorse_topk *h, int32_t kernel_class, int64_t *launches, double *total_ms);
/* admissible[r] == 0 removes row r from every later search -- as if ann.Bruteforce held the admissible rows only (their ids
 * unchanged): what IsHidden and the categories filter of storage/vectors/xvec.go:386-394 need, evaluated by the caller once
 * per filter instead of an over-fetch loop around the search.  NULL = all rows (the default).  The mask is applied inside the
 * candidate sweep (a masked row scores NaN) and in the literal scan. */
int32_t gorse_topk_set_mask(gorse_topk *h, const uint8_t *admissible /*host, N bytes, or NULL*/);
/* statistics of the last all_pairs / search call: queries that took the exact fallback path */
int32_t gorse_topk_last_stats(gorse_topk *h, int64_t *n_fallback, int64_t *n_tie_resolved);

/* ---- exact sparse top-k: the sparse collections of vectors.Database --------------------------------
 * storage/vectors/database.go:90-97 (Vector.Indices / Values), xvec.go:241-247 (dimension 0 = sparse, distance Dot
 * only, Flat = exact index), filled by the IDF item-to-item / user-to-user writers of logics/vector_writer.go:192-209
 * (ascending ids, value sqrt(idf)) and queried by logics/item_to_item.go:50-88.
 * N stored vectors as CSR: indptr[N+1], indices[nnz] STRICTLY ASCENDING inside a row, values[nnz].
 * Score of (query, row) = sum over the common indices, in ascending index order, of q_value * x_value, each product
 * and each addition rounded to float32 (no fused multiply-add); a row sharing no index scores 0.  Results per query
 * are what QueryVectors returns (xvec.go:379-446): all admissible rows ranked by score, descending (equal scores in
 * ascending row order: the reference's Flat index is a third-party module whose tie order no test pins), cut to k, and
 * the rows with Score == 0 dropped AFTER the cut (xvec.go:419-421; TestSparse, database_test.go:217-224: the disjoint
 * vector is not returned).  With positive values -- all the IDF writers produce -- that is simply the k best rows
 * sharing an index with the query.  idx_out / score_out are nq*k padded with -1 / -inf, count_out[t] = valid entries. */
typedef struct gorse_sparse gorse_sparse;
int32_t gorse_sparse_create(gorse_sparse **h, int32_t device, int64_t N, const int64_t *indptr /*host*/,
                            const uint32_t *indices /*host*/, const float *values /*host*/);
int32_t gorse_sparse_destroy(gorse_sparse *h);
/* admissible[r] == 0 removes row r from every later result (IsHidden / the categories CONTAIN_ALL filter of
 * xvec.go:379-446 evaluated by the caller); NULL = all rows admissible (the default). */
int32_t gorse_sparse_set_mask(gorse_sparse *h, const uint8_t *admissible /*host, N bytes, or NULL*/);
/* nq query vectors as CSR (same ordering rule as the stored rows; indices the index never saw match nothing).
 * exclude[t] (array may be NULL, entries may be -1) is a stored row treated as absent for query t. */
int32_t gorse_sparse_search(gorse_sparse *h, int64_t nq, const int64_t *q_indptr /*host*/,
                            const uint32_t *q_indices /*host*/, const float *q_values /*host*/,
                            const int64_t *exclude /*host or NULL*/, int32_t k, int32_t *idx_out /*host*/,
                            float *score_out /*host*/, int32_t *count_out /*host*/);
/* every stored row q in [q_begin, q_end) as a query (the item-to-item / user-to-user refresh): exclude_self != 0
 * treats row q as absent from its own ranking.  Host pointers may be NULL (results stay on the device).
 * A call over ALL rows with exclude_self and no mask walks every pair of ordinary rows once and delivers the score to both
 * rankings (the same products in the same order: the same bits; csrc/sparse_kernels.hpp, SymArgs) -- results identical to
 * any other way of asking, ~2 KB of device scratch per row.  The handle keeps the work plan of its last all-pairs call. */
int32_t gorse_sparse_all_pairs(gorse_sparse *h, int64_t q_begin, int64_t q_end, int32_t k, int32_t exclude_self,
                               int32_t *idx_out /*host or NULL*/, float *score_out /*host or NULL*/,
                               int32_t *count_out /*host or NULL*/);
int32_t gorse_sparse_synchronize(gorse_sparse *h);
/* measurement: hipEvent pairs around sparse_tile_kernel (+ the merge of split queries) on the handle's stream, and the
 * work of the last call: postings = sum over its queries and their indices of the posting-list lengths (= multiply-adds
 * performed; the kernel reads 8 bytes per posting); hits = (query, row) pairs with a non-zero inner product. */
int32_t gorse_sparse_set_profiling(gorse_sparse *h, int32_t on);
int32_t gorse_sparse_get_profile(gorse_sparse *h, int64_t *launches, double *total_ms);
int32_t gorse_sparse_last_stats(gorse_sparse *h, int64_t *postings, int64_t *hits);

/* ---- floats.MM / blas.SGEMM, common/floats/floats.go:241, mm.go:19-49 ------------------------
 * Row-major C(m x n) = op(A) op(B) with the reference's own semantics: the NN, TN and TT
 * cases ACCUMULATE into C, the NT case overwrites it (mm.go:20-48, floats_avx512.c:443-480). */
int32_t gorse_hip_sgemm(int32_t device, int32_t transA, int32_t transB, int32_t m, int32_t n, int32_t k,
                        const float *a /*host*/, int32_t lda, const float *b /*host*/, int32_t ldb, float *c /*host*/,
                        int32_t ldc);
/* the same product on matrices that already lie in the memory of `device` (a caller that keeps its operands resident pays no
 * PCIe transfer); synchronous: C is complete when the call returns */
int32_t gorse_hip_sgemm_device(int32_t device, int32_t transA, int32_t transB, int32_t m, int32_t n, int32_t k,
                               const float *a /*device*/, int32_t lda, const float *b /*device*/, int32_t ldb, float *c /*device*/,
                               int32_t ldc);

#ifdef __cplusplus
}
#endif
#endif /* GORSE_HIP_H */
