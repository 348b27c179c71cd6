// sparse.hip -- exact top-k over SPARSE vectors on gfx950: the sparse collections of vectors.Database
// (storage/vectors/database.go:90-97 Vector.Indices / Values; xvec.go:241-247: dimension 0 = sparse, distance Dot,
// Flat = exact index) that the IDF item-to-item / user-to-user writers fill (logics/vector_writer.go:192-209) and
// QueryItemToItem / QueryUserToUser read (logics/item_to_item.go:50-88, user_to_user.go:50-88).
// Host side of the gorse_sparse_* entry points of include/gorse_hip.h; the kernels are in sparse_kernels.hpp, the host
// part of the index construction (CSR validation, row order, the list of distinct indices) in sparse_host.hpp.
#include <algorithm>

#include "common.hpp"
#include "sparse_host.hpp"
#include "sparse_kernels.hpp"

using namespace gorse;
using gorse::sparse::TileArgs;

struct gorse_sparse {
    int device = 0;
    int64_t N = 0, nnz = 0, Dc = 0;
    int64_t Np = 0;  // scratch ids: N + the phantom ids behind the front (sparse_host.hpp, RowOrder)
    int64_t front_cut = 0;  // the rows longer than this are the front (0 = no front); the split threshold of a symmetric pass
    int32_t logG = 0, ngroups = 0;  // groups of G = 1 << logG consecutive scratch ids
    hipStream_t stream = nullptr;
    // stored rows as CSR with directory entries instead of raw indices (the queries of all_pairs), the tiled posting lists
    DevBuf<int64_t> r_ptr;
    DevBuf<int32_t> r_cid, orig_of, new_of;
    DevBuf<float> r_val;
    DevBuf<uint32_t> dims, off;
    DevBuf<sparse::Posting> post;
    DevBuf<uint8_t> mask_sid;
    bool has_mask = false;
    int64_t n_admissible = 0;  // rows with mask != 0 (N without a mask)
    float stored_small = 0.0f;  // smallest non-zero |value| stored
    sparse::RowOrder order;    // host copy: masks arrive in the caller's row order
    std::vector<double> group_share;  // per group: share of the stored entries that its rows hold
    std::vector<int64_t> r_ptr_host;
    // staging of one call
    DevBuf<int64_t> q_ptr, q_excl;
    DevBuf<uint32_t> q_idx;
    DevBuf<int32_t> q_cid, out_idx, out_cnt, next, split_t, split_n, part_cnt, heavy_t, heavy_pslot, range_start;
    int32_t n_ranges = 0;  // row ranges of about equal cost for sparse_rows_kernel
    hipStream_t stream2 = nullptr;  // the heavy queries run next to the others
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf<uint2> dense;  // heavy queries of a launch: n x Dc {present, value bits}
    DevBuf<float> q_val, out_score;
    DevBuf<sparse::Work> work;
    DevBuf<unsigned long long> part_keys, stat;
    // the symmetric form of an all-pairs pass (sparse::SymArgs)
    DevBuf<unsigned long long> sym_tp, sym_flist, sym_own;
    DevBuf<uint32_t> sym_neg, sym_fcnt;
    DevBuf<float> sym_fm, sym_fmt;  // the front's score matrix and its transpose (SymArgs)
    DevBuf<int32_t> sym_redo;
    // The work plan of the last all-pairs call (the queries are the stored rows, which never change): the host's sorts and the
    // uploads of a 200,000-row pass were ~1.5 ms of its 36 (r06_y2: kernels 26.5 ms, pass 28.9).  `valid`: the host lists below are
    // the plan of (q_first, nq, split, heavy); `on_device`: work / split_* / heavy_* on the device still hold it.
    struct Plan {
        bool valid = false, on_device = false;
        int64_t q_first = -1, nq = -1, split = 0, heavy = 0;
        std::vector<int32_t> longs, shorts, heavy_t, heavy_pslot, nparts;
        std::vector<sparse::Work> work;
    } plan;
    int64_t last_sym[4] = {0, 0, 0, 0};  // the last call: ran symmetric, rows redone, foreign entries ranked, longest foreign list
    DevBuf<sparse::Trace> trace;  // probe (gorse_hip_test_sparse_trace)
    std::vector<sparse::Trace> trace_host;
    bool trace_on = false;
    KernelProfile prof{1};
    int64_t last_postings = 0, last_hits = 0;
    int32_t use() const {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return fail(GORSE_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return GORSE_OK;
    }
};

namespace {

// probes / test hooks (include/gorse_hip_test.h); results never depend on them
int g_sparse_tile = 0;           // rows per group (power of two, 256 .. 16384); 0 = 2048
int64_t g_sparse_split = 2048;   // queries with more entries than this become one work item per group; <= 0 = never
int64_t g_sparse_heavy = 16384;  // ... and with more than this they are scored row by row against a dense copy (sparse_rows_kernel);
                                 // <= 0 = never
int g_sparse_atomic = -1;        // -1 = ds_add_f32 unless the values call for the load/add/store form, 0 / 1 = force
int64_t g_sparse_max_slots = 0;  // workgroups per launch (0 = 16 per CU)
int g_sparse_sym = -1;            // an eligible all-pairs pass walks its whole-query pairs once (SymArgs): -1 / 1 = yes, 0 = never,
                                  // 2 = yes, but the front (the long rows in a group of their own) does not deliver
int g_sparse_sym_caps[3] = {0, 0, 0};  // foreign list capacities of the three tiers (0 = the defaults): tests overflow them on purpose
int64_t g_sparse_rows_wgs = 1024;  // most workgroups of sparse_rows_kernel (probe: bits 8.. of gorse_hip_test_set_sparse_probe x 256)
int g_sparse_front = 1;           // handles created afterwards give their longest rows a row group of their own (RowOrder): 0 = no, 1 = the rows
                                  // longer than 1 .. 2 x the split threshold (gorse_sparse_create), > 1 = the rows longer than this
int g_sparse_tri_probe = 0;      // timing probe: whole-query items stop at their own group (gorse_hip_test_set_sparse_probe)
int g_sparse_cap_shift = 2;      // postings a super-visit's table takes: accumulators >> this (gorse_hip_test_set_sparse_table)
int g_sparse_head = -1;          // groups a whole-query item visits one by one (the rest in hashed super-visits); -1 = head_groups_of()

// The head: the leading groups (rows come longest first, so the groups' shares of the stored entries fall) that hold more than
// their even share of the entries -- there a (query, group) visit meets hundreds of postings and the directly indexed accumulators
// pay; behind them a visit meets a few dozen and the super-visits of sparse_tile_kernel take several groups at once.
int head_groups_of(const gorse_sparse *h) {
    if (g_sparse_head >= 0) return std::min<int>(g_sparse_head, h->ngroups);
    int n = 0;
    while (n < h->ngroups && h->group_share[(size_t)n] * h->ngroups > 1.0) n++;
    return std::max(1, std::min(n, h->ngroups));
}

int pick_log_group() {
    int l = 11;  // 2048 rows: 8 KB of accumulators + 2 KB of stamps + 1 KB of touched list, 10 waves per CU with KP = 256
                 // (C3-shard item-to-item: 94 ms against 101 with 4096 and 122 with 8192, profiles/r02_i_probe_sparse_c3.txt)
    if (g_sparse_tile > 0) {
        l = 8;
        while ((1 << l) < g_sparse_tile && l < 14) l++;
    }
    return l;
}
template <int KP>
int32_t launch_tiles(const TileArgs &a, unsigned grid, size_t lds, bool atomic, bool sym, hipStream_t s) {  // s: the stream of this launch
    auto kern = atomic ? sparse::sparse_tile_kernel<KP, true, false> : sparse::sparse_tile_kernel<KP, false, false>;
    if (sym) kern = atomic ? sparse::sparse_tile_kernel<KP, true, false, 1> : sparse::sparse_tile_kernel<KP, false, false, 1>;
    if (sym && a.sym.front) kern = atomic ? sparse::sparse_tile_kernel<KP, true, false, 2> : sparse::sparse_tile_kernel<KP, false, false, 2>;
#ifdef GORSE_PROBE  // the trace instantiation exists in `make probe-lib` builds only
    if (a.trace && !sym) kern = atomic ? sparse::sparse_tile_kernel<KP, true, true> : sparse::sparse_tile_kernel<KP, false, true>;
#endif
    GORSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3(grid), dim3(sparse::kBlock), lds, s>>>(a);
    return GORSE_OK;
}

int32_t scan_exclusive(uint32_t *x, int64_t n, DevBuf<uint32_t> &sums, hipStream_t s) {
    const int64_t per = (int64_t)sparse::kScanBlock * sparse::kScanPer;
    const int64_t nb = ceil_div(n, per);
    GORSE_TRY(sums.ensure((size_t)nb));
    sparse::sparse_scan_sums_kernel<<<dim3((unsigned)nb), dim3(sparse::kScanBlock), 0, s>>>(x, n, sums.p);
    sparse::sparse_scan_top_kernel<<<dim3(1), dim3(sparse::kScanBlock), 0, s>>>(sums.p, nb);
    sparse::sparse_scan_apply_kernel<<<dim3((unsigned)nb), dim3(sparse::kScanBlock), 0, s>>>(x, n, sums.p);
    GORSE_HIP_CHECK(hipGetLastError());
    return GORSE_OK;
}

// nq queries = CSR rows q_first .. of device arrays (qp, qc, qv); results into the handle's out_* buffers and, where
// given, the host arrays.  q_len_host[t + 1] - q_len_host[t] = number of entries of query t (host copy of the offsets);
// q_small = smallest non-zero |value| among the queries' entries (0 = none).
int32_t run_queries(gorse_sparse *h, const int64_t *qp, const int32_t *qc, const float *qv, int64_t q_first, int64_t nq,
                    const int64_t *q_len_host, const int64_t *excl_dev, int exclude_self, int k, bool atomic, int32_t *idx_out,
                    float *score_out, int32_t *cnt_out) {
    const int kp = sparse::pick_kp(k);
    if (!kp) return fail(GORSE_ERR_INVALID, "k = %d: must be in 1..1024", k);
    if (nq > INT32_MAX / 2) return fail(GORSE_ERR_INVALID, "too many queries in one call");
    GORSE_TRY(h->out_idx.ensure((size_t)nq * k));
    GORSE_TRY(h->out_score.ensure((size_t)nq * k));
    GORSE_TRY(h->out_cnt.ensure((size_t)nq));
    GORSE_TRY(h->stat.ensure(4));
    GORSE_TRY(h->next.ensure(sparse::kQueueWords));
    // Work items: a long query as one item per group (+ a merge of the partial rankings), the others as one item each.  Long
    // queries first, everything longest first: the launch ends with the cheap items.  The partial rankings of the long queries
    // are bounded (kPartBytes): a call with more long queries than fit takes several launches.
    const int64_t ng = std::max(h->ngroups, h->n_ranges);  // most partial rankings of one query
    gorse_sparse::Plan &plan = h->plan;
    // The split threshold of this call.  A symmetric pass's whole-query items stop at their own row group, so a long row as ONE item is
    // no tail there; one work item per group is what costs (the C3 shard pass: 24.3 ms with the rows longer than 2048 split, 21.9 with
    // those longer than 4096, profiles/r06_zo_probe_gpu_probe_sparse_knobs.txt): such a pass splits the front's rows only.
    const bool sym_call = g_sparse_sym != 0 && (g_sparse_sym > 0 || h->ngroups >= 2) && qp == h->r_ptr.p && q_first == 0 && nq == h->N &&
                          exclude_self && !excl_dev && !h->has_mask && !h->trace_on;
    const int64_t split_used = sym_call && h->front_cut > 0 && h->order.pad > 0 ? h->front_cut : g_sparse_split;
    const bool plan_hit = qp == h->r_ptr.p && plan.valid && plan.q_first == q_first && plan.nq == nq && plan.split == split_used &&
                          plan.heavy == g_sparse_heavy;
    if (!plan_hit) plan.valid = plan.on_device = false;
    std::vector<int32_t> &longs = plan.longs, &shorts = plan.shorts;
    if (!plan_hit) {   // counting sort by length, longest first (stable in t)
        longs.clear(), shorts.clear();
        const int64_t cut = split_used > 0 ? split_used : INT64_MAX;
        std::vector<int32_t> count;
        int64_t longest_short = 0;
        for (int64_t t = 0; t < nq; t++) {
            const int64_t L = q_len_host[t + 1] - q_len_host[t];
            if (L > cut)
                longs.push_back((int32_t)t);
            else
                longest_short = std::max(longest_short, L);
        }
        if (longest_short <= (1 << 22)) {
            count.assign((size_t)longest_short + 2, 0);
            for (int64_t t = 0; t < nq; t++) {
                const int64_t L = q_len_host[t + 1] - q_len_host[t];
                if (L <= cut) count[(size_t)(longest_short - L) + 1]++;
            }
            for (size_t i = 1; i < count.size(); i++) count[i] += count[i - 1];
            shorts.resize((size_t)nq - longs.size());
            for (int64_t t = 0; t < nq; t++) {
                const int64_t L = q_len_host[t + 1] - q_len_host[t];
                if (L <= cut) shorts[(size_t)count[(size_t)(longest_short - L)]++] = (int32_t)t;
            }
        } else {  // never split and very long: a comparison sort
            for (int64_t t = 0; t < nq; t++)
                if (q_len_host[t + 1] - q_len_host[t] <= cut) shorts.push_back((int32_t)t);
            std::stable_sort(shorts.begin(), shorts.end(), [&](int32_t x, int32_t y) {
                return q_len_host[x + 1] - q_len_host[x] > q_len_host[y + 1] - q_len_host[y];
            });
        }
        std::stable_sort(longs.begin(), longs.end(), [&](int32_t x, int32_t y) {
            return q_len_host[x + 1] - q_len_host[x] > q_len_host[y + 1] - q_len_host[y];
        });
    }
    constexpr size_t kPartBytes = (size_t)2 << 30;
    constexpr size_t kDenseBytes = (size_t)1 << 30;
    const size_t per_launch = std::max<size_t>(
        1, std::min({kPartBytes / ((size_t)ng * kp * 8), kDenseBytes / ((size_t)std::max<int64_t>(h->Dc, 1) * sizeof(uint2)),
                     (size_t)32768}));  // 32768: the heavy queries of a launch are a grid dimension
    auto is_heavy = [&](int32_t t) { return g_sparse_heavy > 0 && q_len_host[t + 1] - q_len_host[t] > g_sparse_heavy; };
    GORSE_HIP_CHECK(hipMemsetAsync(h->stat.p, 0, 4 * sizeof(unsigned long long), h->stream));
    TileArgs a;
    // The symmetric form: every stored row is a query and leaves out only itself; the long / heavy rows are the first scratch ids
    // (rows are numbered longest first and a query's length is its row's).
    // (a collection of one row group has nothing to leave out: S-ml100k's 1682 items 0.82 ms unsymmetric, 0.92 symmetric,
    // profiles/r06_zg_probe_gpu_probe_sparse_shapes.txt -- the symmetric form is the default from two groups on)
    bool sym = sym_call && !shorts.empty() && longs.size() < (size_t)h->N;
    for (size_t l = 0; sym && l < longs.size(); l++) sym = h->order.rank_of(h->order.new_of[(size_t)longs[l]]) < (int64_t)longs.size();
    a.sym = sparse::SymArgs{};
    if (sym) {
        sparse::SymArgs &y = a.sym;
        const int64_t N = h->Np;  // (per scratch id)
        // Foreign list capacities.  A row's bound is published by its own item; what arrives before that is appended unfiltered, and
        // that is the rows at the head of the work list while the launch fills (up to a launch's waves deliver to them at once).
        y.first = (int32_t)h->order.sid_of_rank((int64_t)longs.size());
        y.t1 = (int32_t)std::min<int64_t>(N, 16384), y.t2 = (int32_t)std::min<int64_t>(N, 65536);
        const int32_t most = (int32_t)std::min<int64_t>(N, 1 << 20);  // a list never holds more than N - 1 entries
        y.c1 = std::min(most, g_sparse_sym_caps[0] > 0 ? g_sparse_sym_caps[0] : 8192);
        y.c2 = std::min(most, g_sparse_sym_caps[1] > 0 ? g_sparse_sym_caps[1] : 1024);
        y.c3 = std::min(most, g_sparse_sym_caps[2] > 0 ? g_sparse_sym_caps[2] : (N > ((int64_t)1 << 21) ? 64 : 256));
        const size_t cells = (size_t)y.t1 * y.c1 + (size_t)(y.t2 - y.t1) * y.c2 + (size_t)(N - y.t2) * y.c3;
        GORSE_TRY(h->sym_tp.ensure((size_t)N));
        GORSE_TRY(h->sym_neg.ensure((size_t)N));
        GORSE_TRY(h->sym_fcnt.ensure((size_t)N));
        GORSE_TRY(h->sym_flist.ensure(cells));
        GORSE_TRY(h->sym_own.ensure((size_t)N * kp));
        GORSE_TRY(h->sym_redo.ensure((size_t)N + 1));
        GORSE_HIP_CHECK(hipMemsetAsync(h->sym_tp.p, 0, (size_t)N * 8, h->stream));
        GORSE_HIP_CHECK(hipMemsetAsync(h->sym_neg.p, 0, (size_t)N * 4, h->stream));
        GORSE_HIP_CHECK(hipMemsetAsync(h->sym_fcnt.p, 0, (size_t)N * 4, h->stream));
        GORSE_HIP_CHECK(hipMemsetAsync(h->sym_redo.p, 0, 4, h->stream));
        // The front delivers when the long / heavy rows of this call are exactly the rows that got group 0 to themselves at creation.
        y.front = 0;
        constexpr size_t kFrontBytes = (size_t)16 << 30;  // the score matrix and its transpose: beyond this the front keeps its scores
        // (... and fewer front rows than 2.5 k give front_bound too little to bound an item by: the foreign lists would fill)
        if (g_sparse_sym != 2 && h->order.pad > 0 && (int64_t)longs.size() == h->order.n_front && 2 * (int64_t)h->order.n_front >= 5 * (int64_t)k &&
            (size_t)h->order.n_front * (size_t)(N - y.first) * 8 <= kFrontBytes) {
            y.front = h->order.n_front;
            y.fw = (y.front + 63) / 64 * 64;
            y.Ns = N - y.first;
            GORSE_TRY(h->sym_fm.ensure((size_t)y.front * (size_t)y.Ns));
            GORSE_TRY(h->sym_fmt.ensure((size_t)y.Ns * (size_t)y.fw));
            GORSE_HIP_CHECK(hipMemsetAsync(h->sym_fm.p, 0, (size_t)y.front * (size_t)y.Ns * 4, h->stream));
            y.fm = h->sym_fm.p, y.fmt = h->sym_fmt.p;
        }
        y.tl = 0;
        if (!y.front) {   // the heads of the first lists, which their rows' items read while they fill (sym_tighten): one strided clear.  The first
            // four row groups: behind them a row's own walk reaches its first read-back in well under a millisecond.  (With a front
            // that delivers, an item's bound comes from the front's column instead: nothing to clear.)
            y.tl = std::min<int32_t>(y.t1, 8192);
            const size_t look = (size_t)std::min<int32_t>(y.c1, sparse::kSymLook);
            if (look == (size_t)y.c1)
                GORSE_HIP_CHECK(hipMemsetAsync(h->sym_flist.p, 0, (size_t)y.tl * y.c1 * 8, h->stream));
            else
                GORSE_HIP_CHECK(hipMemset2DAsync(h->sym_flist.p, (size_t)y.c1 * 8, 0, look * 8, (size_t)y.tl, h->stream));
        }
        y.tp = h->sym_tp.p, y.neg = h->sym_neg.p, y.fcnt = h->sym_fcnt.p, y.flist = h->sym_flist.p, y.own = h->sym_own.p;
    }
    a.off = h->off.p, a.post = h->post.p, a.ngroups = h->ngroups, a.logG = h->logG, a.part_stride = (int32_t)ng;
    a.head_groups = head_groups_of(h);
    a.cap_shift = g_sparse_cap_shift;
    a.tri_probe = g_sparse_tri_probe;
    a.N = h->N, a.Np = h->Np;
    a.orig_of = h->orig_of.p, a.new_of = h->new_of.p;
    a.q_ptr = qp, a.q_cid = qc, a.q_val = qv, a.q_first = q_first;
    a.exclude = excl_dev, a.exclude_self = exclude_self;
    a.mask_sid = h->has_mask ? h->mask_sid.p : nullptr;
    a.n_admissible = h->has_mask ? h->n_admissible : h->N;
    a.k = k;
    a.out_idx = h->out_idx.p, a.out_score = h->out_score.p, a.out_cnt = h->out_cnt.p;
    a.stat = h->stat.p;
    a.next = h->next.p;
    const int64_t slots = g_sparse_max_slots > 0 ? g_sparse_max_slots : 256 * 16;
    // LDS of a workgroup: ranking buffer + per accumulator 4 B (sum) + 1 B (stamp) + 0.5 B (touched list)
    const size_t lds = (size_t)2 * kp * 8 + ((size_t)11 << h->logG) / 2 + 64;  // + the batch assembly's board
    auto launch_work = [&](size_t n_items, bool symmetric) -> int32_t {  // a.work / a.n_work are set
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)n_items, slots));
        switch (kp) {
            case 128: GORSE_TRY(launch_tiles<128>(a, grid, lds, atomic, symmetric, h->stream)); break;
            case 256: GORSE_TRY(launch_tiles<256>(a, grid, lds, atomic, symmetric, h->stream)); break;
            case 512: GORSE_TRY(launch_tiles<512>(a, grid, lds, atomic, symmetric, h->stream)); break;
            default: GORSE_TRY(launch_tiles<1024>(a, grid, lds, atomic, symmetric, h->stream)); break;
        }
        GORSE_HIP_CHECK(hipGetLastError());
        return GORSE_OK;
    };
    auto launch_merge = [&](size_t n_long) -> int32_t {  // split_t / split_n / part_* of n_long queries -> their result rows
        sparse::MergeArgs m;
        m.split_t = h->split_t.p, m.split_n = h->split_n.p, m.n_split = (int32_t)n_long, m.part_stride = (int32_t)ng;
        m.part_keys = h->part_keys.p, m.part_cnt = h->part_cnt.p;
        m.q_first = q_first, m.N = h->N, m.exclude = excl_dev, m.exclude_self = exclude_self;
        m.mask_sid = a.mask_sid, m.new_of = h->new_of.p, m.n_admissible = a.n_admissible, m.k = k;
        m.out_idx = a.out_idx, m.out_score = a.out_score, m.out_cnt = a.out_cnt;
        const unsigned mg = (unsigned)std::min<size_t>(n_long, 4096);
        switch (kp) {
            case 128: sparse::sparse_merge_kernel<128><<<dim3(mg), dim3(sparse::kBlock), 0, h->stream>>>(m); break;
            case 256: sparse::sparse_merge_kernel<256><<<dim3(mg), dim3(sparse::kBlock), 0, h->stream>>>(m); break;
            case 512: sparse::sparse_merge_kernel<512><<<dim3(mg), dim3(sparse::kBlock), 0, h->stream>>>(m); break;
            default: sparse::sparse_merge_kernel<1024><<<dim3(mg), dim3(sparse::kBlock), 0, h->stream>>>(m); break;
        }
        GORSE_HIP_CHECK(hipGetLastError());
        return GORSE_OK;
    };
    const int tok = h->prof.begin(0, h->stream);
    std::vector<sparse::Work> &work = plan.work;
    std::vector<int32_t> &heavy_t = plan.heavy_t, &heavy_pslot = plan.heavy_pslot, &nparts = plan.nparts;
    const bool one_launch = longs.size() <= per_launch;  // (a plan is kept across calls only then)
    h->trace_host.clear();
    for (size_t l0 = 0; l0 == 0 || l0 < longs.size(); l0 += per_launch) {
        const size_t l1 = std::min(longs.size(), l0 + per_launch);
        if (!(plan_hit && one_launch)) {
            work.clear(), heavy_t.clear(), heavy_pslot.clear();
            // the parts of the long queries by estimated cost (entries of the query x share of the stored entries in the group's
            // rows), dearest first and at a raised wave priority where one part alone is a sizeable piece of the launch: (longest
            // query, most popular rows) ran for 60 of the launch's 68 ms (profiles/r02_o_probe_sparse_trace.txt)
            nparts.assign(l1 - l0, h->ngroups);
            for (size_t l = l0; l < l1; l++) {
                if (is_heavy(longs[l])) {
                    nparts[l - l0] = h->n_ranges;
                    heavy_t.push_back(longs[l]);
                    heavy_pslot.push_back((int32_t)(l - l0));
                    continue;
                }
                for (int32_t g = 0; g < h->ngroups; g++) work.push_back(sparse::Work{longs[l], g, (int32_t)(l - l0), 0});
            }
            auto cost = [&](const sparse::Work &w) { return (double)(q_len_host[w.t + 1] - q_len_host[w.t]) * h->group_share[(size_t)w.part]; };
            std::stable_sort(work.begin(), work.end(), [&](const sparse::Work &x, const sparse::Work &y) { return cost(x) > cost(y); });
            for (sparse::Work &w : work) w.prio = cost(w) >= 4096.0;
            if (l0 == 0)
                for (int32_t t : shorts) work.push_back(sparse::Work{t, -1, 0, 0});
        }
        if (work.empty() && heavy_t.empty()) break;
        const size_t n_long = l1 - l0;
        const bool upload = !(plan_hit && one_launch && plan.on_device);
        GORSE_TRY(h->work.ensure(work.size()));
        if (upload)
            GORSE_HIP_CHECK(hipMemcpyAsync(h->work.p, work.data(), work.size() * sizeof(sparse::Work), hipMemcpyHostToDevice, h->stream));
        if (n_long > 0) {
            GORSE_TRY(h->split_t.ensure(n_long));
            GORSE_TRY(h->split_n.ensure(n_long));
            GORSE_TRY(h->part_keys.ensure(n_long * (size_t)ng * (size_t)kp));
            GORSE_TRY(h->part_cnt.ensure(n_long * (size_t)ng * 2));
            if (upload) {
                GORSE_HIP_CHECK(hipMemcpyAsync(h->split_n.p, nparts.data(), n_long * 4, hipMemcpyHostToDevice, h->stream));
                GORSE_HIP_CHECK(hipMemcpyAsync(h->split_t.p, longs.data() + l0, n_long * 4, hipMemcpyHostToDevice, h->stream));
            }
        }
        GORSE_HIP_CHECK(hipMemsetAsync(h->next.p, 0, sparse::kQueueWords * sizeof(int32_t), h->stream));
        if (!heavy_t.empty()) {  // the heavy queries: dense copies, then every stored row against them -- next to the others
            const size_t nh = heavy_t.size();
            GORSE_TRY(h->heavy_t.ensure(nh));
            GORSE_TRY(h->heavy_pslot.ensure(nh));
            GORSE_TRY(h->dense.ensure(nh * (size_t)h->Dc + 1));
            if (upload) {
                GORSE_HIP_CHECK(hipMemcpyAsync(h->heavy_t.p, heavy_t.data(), nh * 4, hipMemcpyHostToDevice, h->stream));
                GORSE_HIP_CHECK(hipMemcpyAsync(h->heavy_pslot.p, heavy_pslot.data(), nh * 4, hipMemcpyHostToDevice, h->stream));
            }
            GORSE_HIP_CHECK(hipEventRecord(h->ev_fork, h->stream));  // uploads, memsets, the buffers of the previous launch
        }
        // The list walk is enqueued BEFORE the dense-vector kernel's chain (which waits for the fork event only): it then holds its
        // waves when the other kernel arrives.  In the other order -- decided by a race of tens of microseconds until round 6 -- the
        // dense-vector kernel takes its slots first and the pass is 1-3 ms longer (profiles/r06_zz3_timeline_sparse_front.txt: list walk
        // 18.7 ms with the dense-vector kernel 18.0 beside it, against 20.4 with 13.9).
        a.work = h->work.p, a.n_work = (int32_t)work.size();
        a.part_keys = h->part_keys.p, a.part_cnt = h->part_cnt.p;
        a.trace = nullptr;
#ifdef GORSE_PROBE
        if (h->trace_on) {
            GORSE_TRY(h->trace.ensure(work.size()));
            a.trace = h->trace.p;
        }
#endif
        if (!work.empty()) GORSE_TRY(launch_work(work.size(), sym));
        if (!heavy_t.empty()) {
            const size_t nh = heavy_t.size();
            GORSE_HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
            GORSE_HIP_CHECK(hipMemsetAsync(h->dense.p, 0, (nh * (size_t)h->Dc + 1) * sizeof(uint2), h->stream2));
            sparse::sparse_dense_query_kernel<<<dim3(64, (unsigned)nh), dim3(256), 0, h->stream2>>>(qp, qc, qv, q_first, h->heavy_t.p,
                                                                                                    h->Dc, h->dense.p);
            sparse::RowsArgs r;
            r.r_ptr = h->r_ptr.p, r.r_cid = h->r_cid.p, r.r_val = h->r_val.p, r.orig_of = h->orig_of.p;
            r.N = h->N, r.range_start = h->range_start.p, r.n_ranges = h->n_ranges, r.part_stride = (int32_t)ng;
            r.dense = h->dense.p, r.Dc = h->Dc;
            r.heavy_t = h->heavy_t.p, r.heavy_pslot = h->heavy_pslot.p, r.n_heavy = (int32_t)nh;
            r.q_first = q_first, r.exclude = excl_dev, r.exclude_self = exclude_self, r.mask_sid = a.mask_sid;
            r.k = k, r.next = h->next.p + 8;
            r.part_keys = h->part_keys.p, r.part_cnt = h->part_cnt.p, r.stat = h->stat.p;
            r.fm = sym && a.sym.front ? a.sym.fm : nullptr, r.Ns = a.sym.Ns, r.first = a.sym.first, r.new_of = h->new_of.p;
            // at most four of its waves per CU: the kernel is bound by its longest row, not by throughput, and at 170 VGPRs a full
            // grid of it would take the register files from the list walk it runs next to
            const unsigned rgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)nh * h->n_ranges, std::min<int64_t>(slots, g_sparse_rows_wgs)));
            switch (kp) {
                case 128: sparse::sparse_rows_kernel<128><<<dim3(rgrid), dim3(sparse::kBlock), 0, h->stream2>>>(r); break;
                case 256: sparse::sparse_rows_kernel<256><<<dim3(rgrid), dim3(sparse::kBlock), 0, h->stream2>>>(r); break;
                case 512: sparse::sparse_rows_kernel<512><<<dim3(rgrid), dim3(sparse::kBlock), 0, h->stream2>>>(r); break;
                default: sparse::sparse_rows_kernel<1024><<<dim3(rgrid), dim3(sparse::kBlock), 0, h->stream2>>>(r); break;
            }
            GORSE_HIP_CHECK(hipGetLastError());
            GORSE_HIP_CHECK(hipEventRecord(h->ev_join, h->stream2));
        }
        if (!heavy_t.empty()) GORSE_HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        if (n_long > 0) GORSE_TRY(launch_merge(n_long));
        // the host lists (work, heavy_*, nparts, split_t) and the trace buffer are reused by the next iteration
        if ((n_long > 0 && !one_launch) || h->trace_on) GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->trace_on && a.trace) {  // (a.trace stays null outside `make probe-lib` builds: nothing was recorded)
            const size_t at = h->trace_host.size();
            h->trace_host.resize(at + work.size());
            GORSE_HIP_CHECK(hipMemcpy(h->trace_host.data() + at, h->trace.p, work.size() * sizeof(sparse::Trace), hipMemcpyDeviceToHost));
        }
    }
    int32_t n_redo = 0;
    bool redo_overwrote = false;
    if (sym) {  // own keys + foreign lists -> the whole-query rows
        sparse::SymMergeArgs m;
        m.sym = a.sym, m.N = h->N, m.new_of = h->new_of.p, m.orig_of = h->orig_of.p, m.k = k;
        if (a.sym.front) {
            sparse::sparse_front_transpose_kernel<<<dim3((unsigned)ceil_div(a.sym.Ns, (int64_t)64), (unsigned)(a.sym.fw / 64)), dim3(256), 0, h->stream>>>(a.sym);
            GORSE_HIP_CHECK(hipGetLastError());
        }
        m.out_idx = a.out_idx, m.out_score = a.out_score, m.out_cnt = a.out_cnt;
        m.redo = h->sym_redo.p, m.stat = h->stat.p;
        const unsigned mg = (unsigned)std::min<int64_t>(h->N, 256 * 32);
        switch (kp) {
            case 128: sparse::sparse_sym_merge_kernel<128><<<dim3(mg), dim3(sparse::kBlock), 0, h->stream>>>(m); break;
            case 256: sparse::sparse_sym_merge_kernel<256><<<dim3(mg), dim3(sparse::kBlock), 0, h->stream>>>(m); break;
            case 512: sparse::sparse_sym_merge_kernel<512><<<dim3(mg), dim3(sparse::kBlock), 0, h->stream>>>(m); break;
            default: sparse::sparse_sym_merge_kernel<1024><<<dim3(mg), dim3(sparse::kBlock), 0, h->stream>>>(m); break;
        }
        GORSE_HIP_CHECK(hipGetLastError());
        GORSE_HIP_CHECK(hipMemcpyAsync(&n_redo, h->sym_redo.p, 4, hipMemcpyDeviceToHost, h->stream));
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (n_redo > 0) {
            // Rows whose foreign list overflowed take the unsymmetric walk -- as LONG queries do: one work item per row group and a
            // merge.  (As whole-query items, 100 rows of the C3 shard's users were 100 waves for 26 ms behind a 430 ms pass,
            // profiles/r06_zh_timeline_sparse_users.txt; the first k rows behind the front never see k candidates of their own, so
            // nothing bounds what is delivered to them.)  More rows than the partial rankings hold: whole-query items.
            std::vector<int32_t> rows((size_t)n_redo);
            GORSE_HIP_CHECK(hipMemcpy(rows.data(), h->sym_redo.p + 1, (size_t)n_redo * 4, hipMemcpyDeviceToHost));
            std::sort(rows.begin(), rows.end(), [&](int32_t x, int32_t y) {
                const int64_t lx = q_len_host[x + 1] - q_len_host[x], ly = q_len_host[y + 1] - q_len_host[y];
                return lx != ly ? lx > ly : x < y;
            });
            const bool as_parts = (size_t)n_redo <= per_launch && g_sparse_split > 0;
            std::vector<sparse::Work> again;
            if (as_parts) {
                for (size_t j = 0; j < rows.size(); j++)
                    for (int32_t g = 0; g < h->ngroups; g++) again.push_back(sparse::Work{rows[j], g, (int32_t)j, 0});
                auto cost = [&](const sparse::Work &w) { return (double)(q_len_host[w.t + 1] - q_len_host[w.t]) * h->group_share[(size_t)w.part]; };
                std::stable_sort(again.begin(), again.end(), [&](const sparse::Work &x, const sparse::Work &y) { return cost(x) > cost(y); });
                const std::vector<int32_t> n_parts(rows.size(), h->ngroups);
                GORSE_TRY(h->split_t.ensure(rows.size()));
                GORSE_TRY(h->split_n.ensure(rows.size()));
                GORSE_TRY(h->part_keys.ensure(rows.size() * (size_t)ng * (size_t)kp));
                GORSE_TRY(h->part_cnt.ensure(rows.size() * (size_t)ng * 2));
                GORSE_HIP_CHECK(hipMemcpy(h->split_t.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
                GORSE_HIP_CHECK(hipMemcpy(h->split_n.p, n_parts.data(), rows.size() * 4, hipMemcpyHostToDevice));
                a.part_keys = h->part_keys.p, a.part_cnt = h->part_cnt.p;
            } else
                for (int32_t t : rows) again.push_back(sparse::Work{t, -1, 0, 0});
            redo_overwrote = true;  // (the device's work list is no longer the plan's)
            GORSE_TRY(h->work.ensure(again.size()));
            GORSE_HIP_CHECK(hipMemcpy(h->work.p, again.data(), again.size() * sizeof(sparse::Work), hipMemcpyHostToDevice));
            GORSE_HIP_CHECK(hipMemsetAsync(h->next.p, 0, sparse::kQueueWords * sizeof(int32_t), h->stream));
            a.work = h->work.p, a.n_work = (int32_t)again.size();
            a.trace = nullptr;
            GORSE_TRY(launch_work(again.size(), false));
            if (as_parts) GORSE_TRY(launch_merge(rows.size()));
        }
    }
    h->prof.end(tok, h->stream);
    plan.valid = plan.on_device = qp == h->r_ptr.p && one_launch;
    if (redo_overwrote) plan.on_device = false;
    plan.q_first = q_first, plan.nq = nq, plan.split = split_used, plan.heavy = g_sparse_heavy;
    unsigned long long st[4] = {0, 0, 0, 0};
    GORSE_HIP_CHECK(hipMemcpyAsync(st, h->stat.p, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    if (idx_out)
        GORSE_HIP_CHECK(hipMemcpyAsync(idx_out, h->out_idx.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
    if (score_out)
        GORSE_HIP_CHECK(hipMemcpyAsync(score_out, h->out_score.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
    if (cnt_out) GORSE_HIP_CHECK(hipMemcpyAsync(cnt_out, h->out_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // also covers the uploads from the work list and `longs`
    h->last_postings = (int64_t)st[0];
    h->last_hits = (int64_t)st[1];
    h->last_sym[0] = sym, h->last_sym[1] = n_redo, h->last_sym[2] = (int64_t)st[2], h->last_sym[3] = (int64_t)st[3];
    // per-call scratch that a handle should not sit on between calls (a collection keeps its handle until it changes)
    constexpr size_t kKeepBytes = (size_t)256 << 20;
    if (h->part_keys.n * sizeof(unsigned long long) > kKeepBytes) h->part_keys.release();
    if (h->dense.n * sizeof(uint2) > kKeepBytes) h->dense.release();
    return GORSE_OK;
}

// ds_add_f32 is used when no partial sum can be subnormal: every term then is 0 or at least 2^-100 in magnitude, and a
// non-zero difference of such terms is at least 2^-123
bool atomic_ok(float stored_small, float query_small) {
    if (g_sparse_atomic >= 0) return g_sparse_atomic != 0;
    if (stored_small == 0.0f || query_small == 0.0f) return true;  // no non-zero product at all
    return (double)stored_small * (double)query_small >= 0x1p-100;
}

}  // namespace

extern "C" int32_t gorse_sparse_create(gorse_sparse **out, int32_t device, int64_t N, const int64_t *indptr,
                                       const uint32_t *indices, const float *values) {
    if (!out) return fail(GORSE_ERR_INVALID, "handle pointer is NULL");
    *out = nullptr;
    if (N <= 0 || !indptr) return fail(GORSE_ERR_INVALID, "N must be positive and indptr non-NULL");
    if (N > INT32_MAX) return fail(GORSE_ERR_INVALID, "N must fit int32");
    const std::string bad = sparse::validate_csr(N, indptr, indices);
    if (!bad.empty()) return fail(GORSE_ERR_INVALID, "stored vectors: %s", bad.c_str());
    const int64_t base = indptr[0], nnz = indptr[N] - indptr[0];
    if (nnz > 0 && !values) return fail(GORSE_ERR_INVALID, "values is NULL");
    if (nnz >= (int64_t)UINT32_MAX) return fail(GORSE_ERR_INVALID, "more than 2^32 - 2 stored entries");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GORSE_ERR_NO_DEVICE, "no HIP device visible (libgorse_hip needs an MI355X / gfx950)");
    if (device < 0 || device >= ndev) return fail(GORSE_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    gorse_sparse *h = new (std::nothrow) gorse_sparse();
    if (!h) return fail(GORSE_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->N = N;
    h->nnz = nnz;
    h->logG = pick_log_group();
    {   // The front: the rows longer than `front_cut` -- the largest of split x {2, 1.75, 1.5, 1.25, 1} that leaves at least 300 of them
        // (three times the hundred neighbours the reference keeps per item, config.go: front_bound wants a few k scores), or the
        // value of the test hook.
        int64_t cut = 0;
        if (g_sparse_front > 1)
            cut = g_sparse_front;
        else if (g_sparse_front == 1 && g_sparse_split > 0) {
            std::vector<int64_t> lens((size_t)N);
            for (int64_t r = 0; r < N; r++) lens[(size_t)r] = indptr[r + 1] - indptr[r];
            const size_t want = (size_t)std::min<int64_t>(300, N);
            std::nth_element(lens.begin(), lens.begin() + (want - 1), lens.end(), std::greater<int64_t>());
            const int64_t len300 = lens[want - 1];  // at least 300 rows are this long or longer
            cut = g_sparse_split;
            for (int q = 8; q > 4; q--)
                if (g_sparse_split * q / 4 < len300) {
                    cut = g_sparse_split * q / 4;
                    break;
                }
        }
        h->front_cut = cut;
        h->order = sparse::order_rows(N, indptr, cut, (int64_t)1 << h->logG);
    }
    const int64_t Np = h->Np = h->order.Np;
    if (Np > INT32_MAX) {
        delete h;
        return fail(GORSE_ERR_INVALID, "N must fit int32");
    }
    const std::vector<uint32_t> dims = sparse::distinct_indices(indices + base, nnz);
    h->Dc = (int64_t)dims.size();
    h->ngroups = (int32_t)ceil_div(Np, (int64_t)1 << h->logG);
    h->group_share.assign((size_t)h->ngroups, 0.0);
    for (int64_t sid = 0; sid < Np; sid++) {
        const int64_t r = h->order.orig_of[(size_t)sid];
        if (r < 0) continue;
        h->group_share[(size_t)(sid >> h->logG)] += (double)(indptr[r + 1] - indptr[r]) / (double)std::max<int64_t>(nnz, 1);
    }
    int32_t rc = [&]() -> int32_t {
        const int64_t cells = h->Dc * h->ngroups;
        if (cells >= (int64_t)1 << 33)
            return fail(GORSE_ERR_INVALID, "index too large: %lld distinct indices x %d row groups", (long long)h->Dc, h->ngroups);
        GORSE_TRY(h->use());
        GORSE_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        GORSE_HIP_CHECK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        GORSE_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        GORSE_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        {   // Row ranges of about equal cost for the heavy queries (sparse_rows_kernel).  Rows come longest first.  A row of more
            // than 1024 entries is walked by the whole wave, 64 entries per step: cost ~ its length; shorter rows go 64 at a time,
            // one per lane: a block costs ~ 27 x its first (longest) row's length (16 entries per ~5 us step against 64 entries
            // per ~0.7 us step).  A range closes when it reaches 1 / 192 of the total -- single long rows become ranges of their own.
            auto len_of = [&](int64_t sid) { const int64_t r = h->order.orig_of[(size_t)sid]; return r < 0 ? (int64_t)0 : indptr[r + 1] - indptr[r]; };
            constexpr int64_t kLongRow = 1024;  // = sparse_rows_kernel's
            auto cost_at = [&](int64_t sid, int64_t first) -> double {
                const int64_t len = len_of(sid);
                if (len > kLongRow) return (double)len;
                return (sid - first) % 64 == 0 ? 27.0 * (double)(len + 1) : 0.0;
            };
            double total = 0;
            for (int64_t sid = 0; sid < Np; sid++) total += cost_at(sid, 0);
            const double target = std::max(total / 192.0, 1.0);
            std::vector<int32_t> starts{0};
            double cost = 0;
            for (int64_t sid = 0; sid < Np; sid++) {
                cost += cost_at(sid, starts.back());
                if (cost >= target && sid + 1 < Np) {
                    starts.push_back((int32_t)(sid + 1));
                    cost = 0;
                }
            }
            starts.push_back((int32_t)Np);
            h->n_ranges = (int32_t)starts.size() - 1;
            GORSE_TRY(h->range_start.alloc(starts.size()));
            GORSE_HIP_CHECK(hipMemcpyAsync(h->range_start.p, starts.data(), starts.size() * 4, hipMemcpyHostToDevice, h->stream));
            GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
        }
        std::vector<int64_t> &ptr0 = h->r_ptr_host;
        ptr0.resize((size_t)N + 1);
        for (int64_t r = 0; r <= N; r++) ptr0[(size_t)r] = indptr[r] - base;
        GORSE_TRY(h->r_ptr.alloc((size_t)N + 1));
        // + 1 / + 2: the tile kernel's look-ahead loads clamp to element 0 (and 1) of every array instead of branching
        GORSE_TRY(h->r_cid.alloc((size_t)nnz + 1));
        GORSE_TRY(h->r_val.alloc((size_t)nnz + 1));
        GORSE_TRY(h->orig_of.alloc((size_t)Np));
        GORSE_TRY(h->new_of.alloc((size_t)N));
        GORSE_TRY(h->dims.alloc((size_t)h->Dc));
        GORSE_TRY(h->off.alloc((size_t)cells + 2));
        GORSE_TRY(h->post.alloc((size_t)nnz + 1));
        DevBuf<uint32_t> raw, cursor, sums;
        GORSE_TRY(raw.alloc((size_t)nnz));
        GORSE_TRY(cursor.alloc((size_t)cells + 1));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->orig_of.p, h->order.orig_of.data(), (size_t)Np * 4, hipMemcpyHostToDevice, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->new_of.p, h->order.new_of.data(), (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->r_ptr.p, ptr0.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice, h->stream));
        GORSE_HIP_CHECK(hipMemsetAsync(h->off.p, 0, ((size_t)cells + 2) * 4, h->stream));
        if (nnz > 0) {
            GORSE_HIP_CHECK(hipMemcpyAsync(h->dims.p, dims.data(), dims.size() * 4, hipMemcpyHostToDevice, h->stream));
            GORSE_HIP_CHECK(hipMemcpyAsync(raw.p, indices + base, (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
            GORSE_HIP_CHECK(hipMemcpyAsync(h->r_val.p, values + base, (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
            const unsigned eg = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, ceil_div(nnz, 256)));
            sparse::sparse_translate_kernel<<<dim3(eg), dim3(256), 0, h->stream>>>(raw.p, nnz, h->dims.p, h->Dc, h->r_cid.p);
            const unsigned rg = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, ceil_div(N, 4)));
            // counts per (index, group) -> exclusive scan = the directory -> scatter through a cursor copy
            sparse::BuildArgs b;
            b.r_ptr = h->r_ptr.p, b.r_cid = h->r_cid.p, b.r_val = h->r_val.p, b.N = N, b.new_of = h->new_of.p;
            b.stride = h->ngroups, b.shift = h->logG, b.cnt = h->off.p, b.post = h->post.p;
            sparse::sparse_build_kernel<false><<<dim3(rg), dim3(256), 0, h->stream>>>(b);
            GORSE_TRY(scan_exclusive(h->off.p, cells + 1, sums, h->stream));
            GORSE_HIP_CHECK(hipMemcpyAsync(cursor.p, h->off.p, ((size_t)cells + 1) * 4, hipMemcpyDeviceToDevice, h->stream));
            b.cnt = cursor.p;
            sparse::sparse_build_kernel<true><<<dim3(rg), dim3(256), 0, h->stream>>>(b);
            GORSE_HIP_CHECK(hipGetLastError());
        }
        GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));  // the temporaries die with this scope
        return GORSE_OK;
    }();
    if (rc != GORSE_OK) {
        std::string keep = last_error();
        gorse_sparse_destroy(h);
        last_error() = keep;
        return rc;
    }
    h->stored_small = sparse::smallest_magnitude(values ? values + base : nullptr, nnz);
    *out = h;
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_destroy(gorse_sparse *h) {
    if (!h) return GORSE_OK;
    (void)hipSetDevice(h->device);
    if (h->stream2) {
        (void)hipStreamSynchronize(h->stream2);
        (void)hipStreamDestroy(h->stream2);
    }
    if (h->stream) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->stream);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_set_mask(gorse_sparse *h, const uint8_t *admissible) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    if (!admissible) {
        h->has_mask = false;
        return GORSE_OK;
    }
    std::vector<uint8_t> by_sid((size_t)h->Np);
    h->n_admissible = 0;
    for (int64_t s = 0; s < h->Np; s++) {
        const int64_t r = h->order.orig_of[(size_t)s];
        by_sid[(size_t)s] = r >= 0 && admissible[r] != 0;
        h->n_admissible += by_sid[(size_t)s];
    }
    GORSE_TRY(h->mask_sid.ensure((size_t)h->Np));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->mask_sid.p, by_sid.data(), (size_t)h->Np, hipMemcpyHostToDevice, h->stream));
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->has_mask = true;
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_search(gorse_sparse *h, int64_t nq, const int64_t *q_indptr, const uint32_t *q_indices,
                                       const float *q_values, const int64_t *exclude, int32_t k, int32_t *idx_out,
                                       float *score_out, int32_t *count_out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (nq < 0 || k <= 0 || (nq > 0 && !q_indptr)) return fail(GORSE_ERR_INVALID, "bad arguments");
    if (nq == 0) return GORSE_OK;
    const std::string bad = sparse::validate_csr(nq, q_indptr, q_indices);
    if (!bad.empty()) return fail(GORSE_ERR_INVALID, "queries: %s", bad.c_str());
    const int64_t base = q_indptr[0], qnnz = q_indptr[nq] - base;
    if (qnnz > 0 && !q_values) return fail(GORSE_ERR_INVALID, "q_values is NULL");
    if (exclude)
        for (int64_t t = 0; t < nq; t++)
            if (exclude[t] < -1 || exclude[t] >= h->N)
                return fail(GORSE_ERR_RANGE, "exclude[%lld] = %lld out of range", (long long)t, (long long)exclude[t]);
    GORSE_TRY(h->use());
    std::vector<int64_t> ptr0((size_t)nq + 1);
    for (int64_t t = 0; t <= nq; t++) ptr0[(size_t)t] = q_indptr[t] - base;
    GORSE_TRY(h->q_ptr.ensure((size_t)nq + 1));
    GORSE_TRY(h->q_idx.ensure((size_t)qnnz));
    GORSE_TRY(h->q_cid.ensure((size_t)qnnz + 1));
    GORSE_TRY(h->q_val.ensure((size_t)qnnz + 1));
    GORSE_HIP_CHECK(hipMemcpyAsync(h->q_ptr.p, ptr0.data(), ((size_t)nq + 1) * 8, hipMemcpyHostToDevice, h->stream));
    if (qnnz > 0) {
        GORSE_HIP_CHECK(hipMemcpyAsync(h->q_idx.p, q_indices + base, (size_t)qnnz * 4, hipMemcpyHostToDevice, h->stream));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->q_val.p, q_values + base, (size_t)qnnz * 4, hipMemcpyHostToDevice, h->stream));
        const unsigned eg = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, ceil_div(qnnz, 256)));
        sparse::sparse_translate_kernel<<<dim3(eg), dim3(256), 0, h->stream>>>(h->q_idx.p, qnnz, h->dims.p, h->Dc, h->q_cid.p);
        GORSE_HIP_CHECK(hipGetLastError());
    }
    if (exclude) {
        GORSE_TRY(h->q_excl.ensure((size_t)nq));
        GORSE_HIP_CHECK(hipMemcpyAsync(h->q_excl.p, exclude, (size_t)nq * 8, hipMemcpyHostToDevice, h->stream));
    }
    const bool atomic = atomic_ok(h->stored_small, sparse::smallest_magnitude(q_values ? q_values + base : nullptr, qnnz));
    // run_queries ends with a stream synchronisation, which also covers the uploads from ptr0
    return run_queries(h, h->q_ptr.p, h->q_cid.p, h->q_val.p, 0, nq, ptr0.data(), exclude ? h->q_excl.p : nullptr, 0, k, atomic,
                       idx_out, score_out, count_out);
}

extern "C" int32_t gorse_sparse_all_pairs(gorse_sparse *h, int64_t q_begin, int64_t q_end, int32_t k, int32_t exclude_self,
                                          int32_t *idx_out, float *score_out, int32_t *count_out) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (q_begin < 0 || q_end > h->N || q_begin > q_end || k <= 0) return fail(GORSE_ERR_RANGE, "bad query range");
    if (q_begin == q_end) return GORSE_OK;
    GORSE_TRY(h->use());
    return run_queries(h, h->r_ptr.p, h->r_cid.p, h->r_val.p, q_begin, q_end - q_begin, h->r_ptr_host.data() + q_begin, nullptr,
                       exclude_self != 0, k, atomic_ok(h->stored_small, h->stored_small), idx_out, score_out, count_out);
}

extern "C" int32_t gorse_sparse_synchronize(gorse_sparse *h) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_set_profiling(gorse_sparse *h, int32_t on) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    h->prof.on = on != 0;
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_get_profile(gorse_sparse *h, int64_t *launches, double *total_ms) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    GORSE_TRY(h->use());
    GORSE_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->prof.resolve();
    if (launches) *launches = h->prof.launches[0];
    if (total_ms) *total_ms = h->prof.ms[0];
    return GORSE_OK;
}

extern "C" int32_t gorse_sparse_last_stats(gorse_sparse *h, int64_t *postings, int64_t *hits) {
    if (!h) return fail(GORSE_ERR_INVALID, "handle is NULL");
    if (postings) *postings = h->last_postings;
    if (hits) *hits = h->last_hits;
    return GORSE_OK;
}

extern "C" void gorse_hip_test_set_sparse_slots(int64_t max_slots) { g_sparse_max_slots = max_slots; }
extern "C" void gorse_hip_test_set_sparse_head(int32_t groups) { g_sparse_head = groups; }
extern "C" void gorse_hip_test_set_sparse_probe(int32_t probe) {
    g_sparse_tri_probe = probe & 0xff;
    g_sparse_rows_wgs = (probe >> 8) > 0 ? (int64_t)(probe >> 8) * 256 : 1024;
}
extern "C" void gorse_hip_test_set_sparse_front(int32_t front) { g_sparse_front = front; }
extern "C" void gorse_hip_test_set_sparse_sym(int32_t mode, int32_t c1, int32_t c2, int32_t c3) {
    g_sparse_sym = mode;
    g_sparse_sym_caps[0] = c1, g_sparse_sym_caps[1] = c2, g_sparse_sym_caps[2] = c3;
}
extern "C" void gorse_hip_test_sparse_sym_stats(const gorse_sparse *h, int64_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = h ? h->last_sym[i] : 0;
}
extern "C" void gorse_hip_test_set_sparse_table(int32_t cap_shift) { g_sparse_cap_shift = cap_shift >= 2 && cap_shift <= 6 ? cap_shift : 2; }
// probe: per-work-item records of the NEXT calls of this handle (on != 0), or the records of the last call: up to cap rows of
// 16 uint64 {t0, t1 (100 MHz ticks), query, group + 1 of a long query (0 = the whole query), entries, chunks taken 64 lists at once, their rounds, segments walked
// one list at a time, groups read back densely, groups read back by re-walking, flattened batches, rows shared inside a batch}; returns the number of work items
extern "C" int64_t gorse_hip_test_sparse_trace(gorse_sparse *h, int32_t on, uint64_t *out, int64_t cap) {
    if (!h) return -1;
    h->trace_on = on != 0;
    if (!out) return (int64_t)h->trace_host.size();
    const int64_t n = std::min<int64_t>(cap, (int64_t)h->trace_host.size());
    for (int64_t i = 0; i < n; i++) {
        const sparse::Trace &t = h->trace_host[(size_t)i];
        uint64_t *o = out + i * 16;
        o[10] = t.batches, o[11] = t.shared_rows, o[12] = t.ticks_once, o[13] = t.ticks_flat, o[14] = t.ticks_back, o[15] = t.ticks_head;
        o[0] = t.t0, o[1] = t.t1, o[2] = (uint64_t)t.t, o[3] = (uint64_t)(t.part + 1), o[4] = t.entries;
        o[5] = t.fast_chunks, o[6] = t.rounds, o[7] = t.slow_segments, o[8] = t.dense_groups, o[9] = t.sparse_groups;
    }
    return (int64_t)h->trace_host.size();
}
extern "C" void gorse_hip_test_set_sparse_tile(int32_t rows) { g_sparse_tile = rows; }
extern "C" void gorse_hip_test_set_sparse_split(int64_t entries) { g_sparse_split = entries; }
extern "C" void gorse_hip_test_set_sparse_heavy(int64_t entries) { g_sparse_heavy = entries; }
extern "C" void gorse_hip_test_set_sparse_atomic(int32_t mode) { g_sparse_atomic = mode; }

