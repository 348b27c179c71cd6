// mf_internal.hpp -- the gorse_mf handle: everything one model keeps resident in HBM.
#pragma once
#include <vector>

#include "cf_device.hpp"

#define GORSE_HOT_DONE_STRIPES 32  // words of the workers' arrival counter (bpr.hip worker_done), GORSE_HOT_DONE_STRIDE words apart
#define GORSE_HOT_DONE_STRIDE 64
#ifndef GORSE_HOT_REPLICAS
#define GORSE_HOT_REPLICAS 8  // replica rows per hot item (bpr.hip kHotReplicas); A/B builds of bpr.hip: make ab AB_FLAGS=-DGORSE_HOT_REPLICAS=16
#endif
// replica rows ALLOCATED (and cleared) per hot item by gorse_mf_create: exactly what the kernels use.  An A/B build of bpr.hip alone with
// more replicas (make ab AB_SRC=bpr AB_FLAGS=-DGORSE_HOT_REPLICAS=16) needs mf.hip to allocate as many: build the library with
// -DGORSE_HOT_REPLICAS_ALLOC=32 for such a session (round 5 always allocated 32: up to 16 MB of allocation + memset per handle at
// nFactors 128, in the very Fit whose handle creation the same round had shortened)
#ifndef GORSE_HOT_REPLICAS_ALLOC
#define GORSE_HOT_REPLICAS_ALLOC GORSE_HOT_REPLICAS
#endif
static_assert(GORSE_HOT_REPLICAS_ALLOC >= GORSE_HOT_REPLICAS, "gorse_mf_create must allocate every replica row the kernels use");

struct gorse_mf {
    int device = 0;
    int64_t U = 0, I = 0, nnz = 0;
    int64_t max_user_row = 0, max_item_row = 0;  // longest feedback rows
    int d = 0;
    bool has_item_csr = false;
    hipStream_t stream = nullptr;   // update kernels, copies
    hipStream_t stream2 = nullptr;  // sampler running ahead of the update kernels
    hipEvent_t ev_sampled[2] = {nullptr, nullptr};
    hipEvent_t ev_consumed[2] = {nullptr, nullptr};
    // epoch pacing (gorse_mf_epoch_throttle / gorse_mf_epoch_times): a ring of (begin, end) event pairs, one per BPR epoch, recorded
    // on the update stream; created with the first epoch.  ep_seq = epochs issued, ep_done = epochs whose events have been read.
    static constexpr int kEpochRing = 16;
    hipEvent_t ev_ep_begin[kEpochRing] = {}, ev_ep_end[kEpochRing] = {};
    bool ep_events = false;
    // An epoch that is enqueued while the one before it is still in flight, with nothing else issued on the handle in between (ep_chain:
    // set by the epoch, cleared by every other entry point through use()), begins where that one ended: its begin IS the previous slot's
    // end event (ep_begin_prev) and no event of its own is recorded -- one barrier packet less in front of its update kernel.
    bool ep_begin_prev[kEpochRing] = {};
    mutable bool ep_chain = false;
    uint64_t ep_seq = 0, ep_done = 0;
    double ep_ms = 0.0;     // device milliseconds of the epochs read so far (since the last reset)
    int64_t ep_timed = 0;   // how many epochs that sum covers
    int64_t ep_untimed = 0; // epochs whose events were overwritten before anybody read them (more than kEpochRing in flight)
    // factors, row-major, row stride d (rows 16-byte aligned whenever d % 4 == 0)
    gorse::DevBuf<float> P, Q, Qsync;
    // dataset.CFSplit user->items (stored order + row-sorted copy) and item->users
    gorse::DevBuf<int64_t> uptr, iptr;
    gorse::DevBuf<int32_t> uidx, uidx_sorted, iidx;
    // BPR triplet chunk buffers (double-buffered: sampler fills one while the other is applied)
    gorse::DevBuf<int32_t> trip[2];
    size_t trip_cap = 0;  // samples per buffer
    // user-run schedule (bpr.hip): the chunk's triplets counting-sorted by user
    gorse::DevBuf<int32_t> sorted[2];  // su | si | sj, each trip_cap long
    gorse::DevBuf<int32_t> ubucket[2]; // U + 2 run offsets per triplet buffer (sorted on stream2)
    gorse::DevBuf<int32_t> urank[2];   // arrival rank of a sample inside its user's run, per triplet buffer
    gorse::DevBuf<int32_t> scan_tmp2;  // per-tile sums of the sort's scan (the sort may run on the sampler stream)
    gorse::DevBuf<int32_t> ubins;      // binned preparation: bin totals, bin starts at kMaxBins (preparations are serial: one copy)
    gorse::DevBuf<int32_t> ubinmat;    // binned preparation: the tile x bin count matrix
    int64_t chunk_seq = 0;             // chunks enqueued so far: buffer = chunk_seq & 1, across calls
    // hot-row replicas of the Hogwild schedule (bpr.hip): popular items' positive updates land here
    gorse::DevBuf<int32_t> hot_slot, hot_items, hot_done;
    gorse::DevBuf<float> hot_rep;
    int n_hot = 0;
    int64_t n_cold = 0;  // items of class "cold" in hot_slot (-2): updates by write-through store (bpr.hip)
    int64_t cold_window = 0;               // what n_cold was computed for (gorse_mf_set_bpr_cold_window)
    std::vector<int32_t> h_item_count;     // training feedbacks per item and
    std::vector<int32_t> h_hot_slot;       // the class of every item (replica slot, -1 warm, -2 cold) as last uploaded: host copies for a re-classification
    gorse::DevBuf<int32_t> order;  // sequential mode: samples sorted by dependency level
    gorse::DevBuf<double> loss;
    gorse::DevBuf<int32_t> fail_count;
    // ALS scratch
    gorse::DevBuf<float> gram, gram_partial, als_scratch;
    // ALS row plan per side (0 = user rows, 1 = item rows), built at create time (als.hip):
    // rows with at most kAlsLongRow feedbacks are solved by one wave each (list sorted by length,
    // longest first); longer rows are cut into chunks whose partial Gram matrices are reduced
    // in chunk order by the long-row solver.
    struct AlsPlan {
        gorse::DevBuf<int32_t> short_rows;               // n_short row ids
        gorse::DevBuf<int32_t> chunk_row, chunk_cnt;     // n_chunks
        gorse::DevBuf<int32_t> chunk_order;              // n_chunks: the chunks longest first
        gorse::DevBuf<int64_t> chunk_beg;                // n_chunks: offset into the side's indices
        gorse::DevBuf<int32_t> long_rows, long_first, long_nch;  // n_long
        gorse::DevBuf<int32_t> long_ident, long_one;             // n_long: 0, 1, 2, ... and 1, 1, 1, ... (the solve after als_partial_reduce_kernel)
        int32_t max_nch = 0;                                     // most chunks of one row
        // rows with feedback (the rows S = sum x x^T runs over, model.go:645-658), cut into Gram chunks
        gorse::DevBuf<int32_t> fb_rows, g_cnt;
        gorse::DevBuf<int64_t> g_beg;
        int64_t n_short = 0, n_chunks = 0, n_long = 0, n_fb_rows = 0, n_gchunks = 0;
    } als_plan[2];
    std::vector<int64_t> h_uptr, h_iptr;      // host copies of the CSR row pointers (row plans are rebuilt from them)
    int64_t als_lo[2] = {0, 0}, als_hi[2] = {0, 0};  // row range of each side this handle solves (gorse_als_set_ranges)
    gorse::DevBuf<unsigned long long> als_prof;  // probe: phase counters of als_row_kernel (gorse_hip_test_als_profile)
    gorse::DevBuf<float> als_zeros;    // 64 zero words: where padding lanes of the gathers read
    gorse::DevBuf<float> als_partial;  // n_chunks x (d*d + d) partial Gram matrices + column sums
    gorse::DevBuf<float> als_reduced;  // n_long x (d*d + d): a long row's partials added up (als_partial_reduce_kernel)
    // Evaluate's resident split (eval.hip): the test rows, every user's sampled negatives, and the candidate CSR of the users
    // with test feedback (user ids ascending, "test items then negatives" per user)
    gorse::DevBuf<int64_t> ev_tptr, ev_upos, ev_cpos, ev_cptr;
    gorse::DevBuf<int32_t> ev_tidx, ev_neg, ev_neglen, ev_has, ev_clen, ev_users, ev_cand;
    int64_t ev_users_n = 0, ev_cand_n = 0;
    bool ev_valid = false;
    uint64_t ev_generation = 0;  // bumped by every gorse_mf_sample_user_negatives (gorse_mf_resident_generation)
    // generic staging
    gorse::DevBuf<char> stage, rank_in;
    gorse::KernelProfile prof{GORSE_PROF_NCLASSES};

    int32_t use() const {
        ep_chain = false;  // (an entry point that issues nothing restores it: gorse_mf_epoch_throttle / _times)
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return gorse::fail(GORSE_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return GORSE_OK;
    }
};

namespace gorse {
// implemented in bpr.hip / als.hip, used across files
int32_t mf_sync_streams(gorse_mf *h);
// epoch pacing: mark the begin / end of one epoch on the update stream; read the finished ones (wait = block for all of them)
int32_t mf_epoch_begin(gorse_mf *h, bool chained);
int32_t mf_epoch_end(gorse_mf *h);
int32_t mf_epoch_harvest(gorse_mf *h, bool wait);
int32_t mf_delta_export_async(gorse_mf *h, float *dst);        // mf.hip: dst <- Q - Q_sync, enqueued on h->stream
int32_t mf_delta_import_async(gorse_mf *h, const float *src);  // mf.hip: Q <- Q_sync + src; Q_sync <- Q
int32_t als_build_plan(gorse_mf *h, int side, const int64_t *ptr, int64_t rows, int64_t lo, int64_t hi);
// mf.hip: Rank + TopKFilter for n_users users over candidate lists ALREADY on the device (nc entries in all); the rank lists go
// to the host arrays (n_users * topk padded with -1, lengths), enqueued on h->stream -- the caller synchronises
int32_t mf_rank_device(gorse_mf *h, int64_t n_users, const int32_t *d_users, const int64_t *d_cand_ptr, const int32_t *d_cand,
                       int64_t nc, int32_t topk, int32_t *rank_out, int32_t *rank_len);
}
