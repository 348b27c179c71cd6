// sparse_kernels.hpp -- device code of the exact sparse top-k (sparse.hip): the sparse collections of
// vectors.Database (storage/vectors/database.go:90-97, xvec.go:241-247: Dot over Indices / Values, exact Flat index)
// that the IDF item-to-item / user-to-user writers fill (logics/vector_writer.go:192-209).
//
// Round 2 design (round 1's kernel kept its accumulators in global memory and walked one posting list per memory round
// trip: a 16,384-entry query was a 125 ms tail on one wave and the row-streaming path of the 64 longest queries took 11.6 s
// of the 12.6 s pass, profiles/r02_a_kernel_stats_i2i.txt):
//
// * The stored rows are numbered longest first (scratch ids).  ONE WAVE answers one work item with its accumulators in LDS:
//   it walks the query's indices in ascending order and streams every index's postings (8 bytes each) into the accumulators
//   with ds_add_f32.  A row occurs at most once per posting list, so the lanes of one instruction never collide, and the LDS
//   executes one wave's instructions in issue order: every accumulator receives its products in ascending index order -- the
//   float32 merge-order sparse dot of the oracle, bit for bit.
// * The posting lists are sorted by row GROUP (G consecutive scratch ids, G accumulators in LDS): segment (c, g) holds the
//   postings of index c that fall into group g.  A query of ordinary length is ONE work item that visits the groups one after
//   the other; a LONG query (more than 2048 entries) is one work item PER GROUP, and a merge kernel joins the partial
//   rankings.  A HEAVY query (more than 16384 entries) does not walk posting lists at all: it is scattered into a dense vector
//   over the indices and every stored row is scored against that, one row per lane (sparse_rows_kernel) -- a heavy query reaches
//   most rows anyway, and its (query, most popular rows) part kept one wave busy for 65 of a launch's 68 ms, every one of its
//   100,000 lists having postings there (profiles/r02_p / r02_r_probe_sparse_trace.txt).  (As one item the longest query of the C3 shard took 186 ms on its own; round 2's first answer, a second
//   arrangement by row stripes with 2.5 x the LDS per item, left the chip at 4 waves per CU for 40 % of the pass and spent
//   its time on the rows that the lists of a stripe share -- profiles/r02_f / r02_n_probe_sparse_trace.txt.)
// * 64 lists at once: where every list of a chunk contributes at most 8 postings, every lane gathers the postings of ITS
//   list, the lanes stamp their rows in a byte-per-row tag array and read the stamps back -- a foreign stamp means two lists
//   share a row and the order of their products matters; rounds of "everybody below the lowest loser, then the loser" keep
//   that order (see apply_at_once).  Longer segments go one list at a time with their postings over the lanes.
// * Nothing waits for memory: the (index, value) pairs, directory entries and postings of the next three visits are in flight
//   while one is applied.
// * A group is read back either by a scan of its accumulators (dense) or from the list of accumulators that were +0
//   before an add (sparse); both leave the accumulators zero.
//
// Round 6: an all-pairs pass over all rows takes the SYMMETRIC form -- a pair of rows is walked once and the score delivered to the
// other row's ranking; the long rows form a front that delivers to everybody (SymArgs below).
//
// Ranking: 64-bit keys (order-preserving score bits, ~row) are distinct, so "the k largest keys, descending" is one
// well-defined answer whatever order the lanes append in.  Keys above the running threshold go to an LDS buffer of 2*KP
// entries; when it overflows, the k-th largest key is found by bisection on the key bits (the keys sit in registers, one
// ballot per key and bit), the keys below it are dropped and it becomes the threshold.  One sort at the end of the item.
// The reference ranks ALL admissible documents (one sharing no index scores 0), cuts to topK and THEN drops Score == 0
// (xvec.go:419-421), so zero-score documents use up slots: only the non-zero rows are ranked here, and the number of
// results follows from the counts of positive / negative rows and the number of admissible rows (see `written`).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include "rank_keys.hpp"

namespace gorse {
namespace sparse {

// LDS pointers that carry their address space: a volatile access through a generic pointer compiles to FLAT instructions
// (the address-space inference leaves volatile accesses alone), and a FLAT access waits for EVERY outstanding global load
// (s_waitcnt vmcnt(0)) -- which turned the whole look-ahead of the visit pipeline into dead code until round 2's session m.
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) float lds_f32;

constexpr int kBlock = 64;  // one wavefront per workgroup
constexpr int kGather = 8;  // longest segment the 64-lists-at-once path takes

struct Posting {
    int32_t loc;  // the row's scratch id (its accumulator in a directly indexed group: loc & (G - 1))
    float val;
};
typedef int32_t __attribute__((may_alias)) lds_key;  // hash keys of the super-visits live in the accumulators' LDS words

struct Work {
    int32_t t;      // query of the call
    int32_t part;   // -1 = the whole query (all groups), else group `part` of a long query
    int32_t pslot;  // long queries: which block of partial rankings
    int32_t prio;   // != 0: an item that alone is a sizeable part of the launch (a long query in a group of popular rows); its wave
                    // runs at a raised priority
};

// probe (gorse_hip_test_sparse_trace): what one work item did
struct Trace {
    unsigned long long t0, t1;  // s_memrealtime (100 MHz) at its start / end
    int32_t t, part;
    uint32_t entries, fast_chunks, rounds, slow_segments, dense_groups, sparse_groups;
    uint32_t batches, shared_rows;  // flattened path: batches of 64 postings, rows of a batch that two lists shared
    uint32_t ticks_once, ticks_flat, ticks_back, ticks_head;  // 10 ns ticks inside apply_at_once / apply_flattened (or one list at
                                                              // a time) / the read-backs; from the start to the end of view 7
};

// The counters exist in the probe's instantiation of the kernel only (TRACE): in the shipped one they would hold sixteen scalar
// registers of a kernel that already spills them.
template <bool TRACE>
struct Tracer {
    Trace r{};
    __device__ inline void add(uint32_t Trace::*field, uint32_t by = 1) {
        if constexpr (TRACE) r.*field += by;
    }
    __device__ inline unsigned long long now() const {
        if constexpr (TRACE) return __builtin_amdgcn_s_memrealtime();
        return 0;
    }
};

// The SYMMETRIC form of an all-pairs pass (round 6; every stored row is a query, nothing masked or excluded but the row itself).
// score(q, r) and score(r, q) are the same products added in the same order, so a pair of whole-query rows is walked ONCE, by the
// shorter one: rows are numbered longest first, a whole-query item q visits the groups up to its own and keeps the rows r with
// sid(r) < sid(q) for its own ranking -- and DELIVERS the score to r's ranking when r is a whole-query row too (sid(r) >= first;
// the long / heavy queries in front of them walk everything themselves, as before):
//   * tp[sid] = { low word: a lower bound of the ord of r's k-th key, published by r's own item whenever its threshold moves and
//     exactly at its end; high word: positive scores delivered to r, counted while the deliverer saw fewer than k -- written()
//     needs min(pos, k) only -- plus r's own count at its end }, neg[sid]: negative scores delivered, always counted;
//   * a delivered score whose ord is >= the published bound is appended to r's FOREIGN LIST (fcnt[sid] hands out the slots; the
//     work list runs longest first, so r's bound is usually final before the shorter rows reach it -- on the C3 shard nothing a
//     final bound lets through exists at all, what is appended arrived while r was still in flight with no bound yet; the lists
//     of the first rows are therefore the long ones: three tiers of capacity);
//   * r's item leaves its own sorted keys in own[t * KP ..) instead of a result; sparse_sym_merge_kernel ranks own + foreign
//     (keys are distinct: one total order) and writes the row; a row whose foreign list overflowed goes on the redo list and takes
//     the unsymmetric walk in a second launch.
constexpr int kSymLook = 1024;  // entries at the head of the foreign list of a row in front of SymArgs::tl that the row's own item may read for a bound (sym_tighten)
struct SymArgs {
    int32_t first;             // the first scratch id that receives (= number of long / heavy rows)
    int32_t t1, t2;            // foreign list capacities: c1 for sid < t1, c2 for sid < t2, c3 behind
    int32_t tl;                // <= t1: the rows in front of it tighten their bound from their foreign list's head (sym_tighten)
    int32_t c1, c2, c3;
    unsigned long long *tp;    // N
    uint32_t *neg, *fcnt;      // N each
    unsigned long long *flist; // see sym_list_at
    unsigned long long *own;   // N x KP, by query
    // The FRONT (the long / heavy rows in a row group of their own, scratch ids [0, front); first = the group's size): the rows behind
    // it leave group 0 out of their walk, and the front's own kernels -- which score every row anyway -- leave score(front row, r) in
    // fm[slot][sid(r) - first], slot = the front row's scratch id.  No bound filters these: the front IS most rows' best candidates.
    // A row's item starts with a bound from the column as far as it is written (the front's items come first in the work list);
    // sparse_front_transpose_kernel turns the matrix and the merge ranks own + foreign + front.  0 = no front.
    int32_t front;
    int32_t fw;                // slots per row of fmt: front rounded up to 64
    int64_t Ns;                // rows behind the front: Np - first
    float *fm;                 // front x Ns
    float *fmt;                // Ns x fw
};
// A stale value only costs work (a weaker bound lets more through, a lower count adds once more): both words only ever grow.
#ifndef GORSE_SPARSE_SYM_PLAIN_LOAD
#define GORSE_SPARSE_SYM_PLAIN_LOAD 0
#endif
__device__ inline unsigned long long sym_load_tp(const unsigned long long *p) {
#if GORSE_SPARSE_SYM_PLAIN_LOAD
    return *reinterpret_cast<const volatile unsigned long long *>(p);
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ inline int64_t sym_list_at(const SymArgs &y, int32_t sid, uint32_t &cap) {
    if (sid < y.t1) {
        cap = (uint32_t)y.c1;
        return (int64_t)sid * y.c1;
    }
    if (sid < y.t2) {
        cap = (uint32_t)y.c2;
        return (int64_t)y.t1 * y.c1 + (int64_t)(sid - y.t1) * y.c2;
    }
    cap = (uint32_t)y.c3;
    return (int64_t)y.t1 * y.c1 + (int64_t)(y.t2 - y.t1) * y.c2 + (int64_t)(sid - y.t2) * y.c3;
}

struct TileArgs {
    // segment (c, g) = post[off[c * ngroups + g], off[c * ngroups + g + 1]), rows g * G + loc
    const uint32_t *off;
    const Posting *post;
    int32_t ngroups, logG;
    int32_t tri_probe;    // timing probe (results are garbage): a whole-query item visits only the groups up to its own row's -- what a
                          // symmetric walk of the whole-query items would leave of the list walk (gorse_hip_test_set_sparse_probe)
    int32_t cap_shift;    // the hashed table of a super-visit takes at most (accumulators >> cap_shift) postings (2: half full at most)
    int32_t head_groups;  // a whole-query item visits the groups [0, head_groups) one by one with directly indexed accumulators and
                          // the rest in SUPER-VISITS of several groups with hashed accumulators (see sparse_tile_kernel); = ngroups: never
    int32_t part_stride;  // partial rankings per block of part_keys / part_cnt
    int64_t N, Np;                    // stored rows; scratch ids (N + the phantom ids behind the front: sparse_host.hpp)
    const int32_t *orig_of, *new_of;  // scratch id <-> caller's row (orig_of = -1: a phantom id)
    // queries: CSR rows q_first .. of (q_ptr, q_cid, q_val); q_cid = directory entry of the index or -1 (never stored)
    const int64_t *q_ptr;
    const int32_t *q_cid;
    const float *q_val;
    int64_t q_first;
    const int64_t *exclude;  // per query: a stored row left out of its result (-1 = none); may be null
    int exclude_self;        // all pairs: query t is stored row q_first + t, left out of its own result
    const uint8_t *mask_sid;  // admissible[scratch id] or null
    int64_t n_admissible;     // number of admissible rows (N without a mask)
    const Work *work;
    int32_t n_work;
    int32_t *next;     // work counters: kQueueStripes of them from word kQueueBase on, kQueueStride words apart (sparse_tile_kernel)
    int k;
    int32_t *out_idx;   // nq x k, padded with -1
    float *out_score;   // nq x k, padded with -inf
    int32_t *out_cnt;   // nq
    unsigned long long *part_keys;  // per (pslot, group): the best KP keys of the part in any order, padded with 0
    int32_t *part_cnt;              // per (pslot, group): rows scoring above / below zero
    unsigned long long *stat;       // [0] += postings walked, [1] += rows with a non-zero score
    Trace *trace;                   // probe: one record per work item, or null
    SymArgs sym;                    // the SYM instantiation's whole-query items (see SymArgs)
};

// the key encodings and the count of returned results: rank_keys.hpp (shared with the host library's CPU test hook)
using rank::kZeroOrd;
using rank::key_row;
using rank::key_score;
using rank::make_key;
using rank::score_ord;
using rank::written;

// bitonic sort of b[0..CAP) into descending order by the 64 lanes of the workgroup; ends with a barrier
template <int CAP>
__device__ inline void sort_desc(unsigned long long *b, int tid) {
    for (int size = 2; size <= CAP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < CAP; i += kBlock) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const unsigned long long x = b[i], y = b[j];
                    if (desc ? x < y : x > y) {
                        b[i] = y;
                        b[j] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__device__ inline int lanes_below(unsigned long long m, int lane) { return __popcll(m & (((unsigned long long)1 << lane) - 1)); }
__device__ inline uint32_t lane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ inline float lane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline long long wave_sum(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// The buffer is full (CAP keys, all distinct): keep the k largest -- and, while they fit into KP slots, the keys that share
// the k-th key's score -- and make the smallest kept key minus one the new threshold.  The keys sit in registers (CAP / 64 per
// lane); the k-th largest is found bit by bit: a key bit is set in the answer when at least k keys are >= the trial.
template <int KP>
__device__ inline void cut_to_k(unsigned long long *s_buf, int k, int &bcnt, unsigned long long &thr, int lane) {
    constexpr int CAP = 2 * KP, PER = CAP / kBlock;
    unsigned long long key[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) key[j] = s_buf[j * kBlock + lane];
    auto count_ge = [&](unsigned long long t) {
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) c += __popcll(__ballot(key[j] >= t));
        return c;
    };
    unsigned long long kth = 0;
    for (int b = 63; b >= 32; --b) {  // the score bits
        const unsigned long long trial = kth | ((unsigned long long)1 << b);
        if (count_ge(trial) >= k) kth = trial;
    }
    if (count_ge(kth) > KP)  // more equal scores than the buffer keeps: the row bits decide
        for (int b = 31; b >= 0; --b) {
            const unsigned long long trial = kth | ((unsigned long long)1 << b);
            if (count_ge(trial) >= k) kth = trial;
        }
    __syncthreads();  // every lane holds its keys: the buffer can be rewritten
    int base = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const bool keep = key[j] >= kth;
        const unsigned long long m = __ballot(keep);
        if (keep) s_buf[base + lanes_below(m, lane)] = key[j];
        base += __popcll(m);
    }
    __syncthreads();
    bcnt = base;
    thr = kth ? kth - 1 : 0;
}

// The wave's ranking buffer: lanes with want hand their key in; bcnt (slots in use) and thr (keys <= thr cannot be among the
// k best) are wave-uniform.
template <int KP>
__device__ inline void push(unsigned long long *s_buf, int k, int &bcnt, unsigned long long &thr, unsigned long long key, bool want,
                            int lane) {
    constexpr int CAP = 2 * KP;
    unsigned long long m = __ballot(want);
    while (m) {
        const int slot = bcnt + lanes_below(m, lane);
        if (want && slot < CAP) {
            s_buf[slot] = key;
            want = false;
        }
        const int total = bcnt + __popcll(m);
        if (total <= CAP) {
            bcnt = total;
            break;
        }
        __syncthreads();  // slots 0 .. CAP-1 are all written
        cut_to_k<KP>(s_buf, k, bcnt, thr, lane);
        want = want && key > thr;
        m = __ballot(want);
    }
}

// sorts what the buffer holds (padding with 0) -- positive scores first, then the negative ones; all lanes together.
// Round 6: only the smallest power of two that holds the keys is sorted (the rest is padding) -- every work item ends here, and most
// hold far fewer keys than the buffer takes (a 256-entry sort is 36 passes of 4 steps, a 64-entry one 21 of 1).
template <int KP>
__device__ inline void finish(unsigned long long *s_buf, int bcnt, int lane) {
    constexpr int CAP = 2 * KP;
    __syncthreads();
    for (int i = bcnt + lane; i < CAP; i += kBlock) s_buf[i] = 0;
    __syncthreads();
    if (bcnt <= 0) return;
    if (bcnt <= kBlock)
        sort_desc<kBlock>(s_buf, lane);
    else if (bcnt <= CAP / 2)
        sort_desc<(CAP / 2 > kBlock ? CAP / 2 : kBlock)>(s_buf, lane);
    else
        sort_desc<CAP>(s_buf, lane);
}

// result row t from the sorted buffer: cnt entries, the rest padded
__device__ inline void write_result(const unsigned long long *s_buf, int cnt, int k, int64_t t, int32_t *out_idx, float *out_score,
                                    int32_t *out_cnt, int lane) {
    for (int i = lane; i < k; i += kBlock) {
        const unsigned long long key = i < cnt ? s_buf[i] : 0;
        out_idx[t * k + i] = i < cnt ? key_row(key) : -1;
        out_score[t * k + i] = i < cnt ? key_score(key) : __uint_as_float(0xff800000u);
    }
    if (lane == 0) out_cnt[t] = cnt;
}

// Result row t from the UNSORTED buffer (bcnt distinct keys, cnt of them returned).  Up to 256 keys: no sort -- a key's place in the
// row is the number of keys above it; every lane counts that for its (at most four) keys against the whole buffer, one broadcast
// read per key, no barrier.  (A 128-key bitonic sort is 28 passes of reads, writes and a barrier, with every wave of a CU on the
// LDS at once: the symmetric pass's merge went 1.02 -> 0.68 ms.)  More keys: the sort.  All lanes together.
template <int KP>
__device__ inline void write_ranked(unsigned long long *s_buf, int bcnt, int cnt, int k, int64_t t, int32_t *out_idx, float *out_score,
                                    int32_t *out_cnt, int lane) {
    if (bcnt > 256) {
        finish<KP>(s_buf, bcnt, lane);
        write_result(s_buf, cnt, k, t, out_idx, out_score, out_cnt, lane);
        return;
    }
    __syncthreads();
    unsigned long long mine[4];
    int above[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) mine[j] = j * kBlock + lane < bcnt ? s_buf[j * kBlock + lane] : ~0ull;
    for (int i = 0; i < bcnt; i++) {
        const unsigned long long other = s_buf[i];
#pragma unroll
        for (int j = 0; j < 4; j++) above[j] += other > mine[j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j * kBlock + lane < bcnt && above[j] < cnt) {
            out_idx[t * k + above[j]] = key_row(mine[j]);
            out_score[t * k + above[j]] = key_score(mine[j]);
        }
    for (int i = (cnt < bcnt ? cnt : bcnt) + lane; i < k; i += kBlock) {
        out_idx[t * k + i] = -1;
        out_score[t * k + i] = __uint_as_float(0xff800000u);
    }
    if (lane == 0) out_cnt[t] = cnt;
}

// ATOMIC: ds_add_rtn_f32 / ds_add_f32.  !ATOMIC: load / add / store by the same wave, for inputs whose partial sums may be
// subnormal (the LDS adder's handling of those is not relied upon); same order, same bits, slower.
// acc_add_old returns the accumulator's value BEFORE the add (exactly +0 = the row had not been reached, or its sum is back
// at zero).
template <bool ATOMIC>
__device__ inline float acc_add_old(float *acc, int32_t i, float term) {
    if (ATOMIC) return __hip_atomic_fetch_add(&acc[i], term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    volatile lds_f32 *a = (volatile lds_f32 *)acc;
    const float old = a[i];
    a[i] = __fadd_rn(old, term);
    return old;
}
template <bool ATOMIC>
__device__ inline void acc_add(float *acc, int32_t i, float term) {
    if (ATOMIC) {
        __hip_atomic_fetch_add(&acc[i], term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        volatile lds_f32 *a = (volatile lds_f32 *)acc;
        a[i] = __fadd_rn(a[i], term);
    }
}
__device__ inline float acc_take(float *acc, int32_t i) {
    return __hip_atomic_exchange(&acc[i], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// What a lane holds of one VISIT = (group, chunk of 64 of the query's indices): its index's directory entry and value,
// the segment [s, e) of that index's posting list, and the first kGather postings of it.  The three parts are loaded one visit
// apart (see the pipeline in sparse_tile_kernel), so that no load is waited for.
// Loaded values are never touched in the stage that loads them (a select on a loaded value is a wait for it): `in` says whether
// the lane has an entry at all, cid / qv / s / e are raw until the next stage folds `in` into them.
struct Visit {
    bool in;  // the lane has an entry of the query in this visit (stage 1), and the index is stored (after stage 2)
    int32_t cid;
    float qv;
    uint32_t s, e;  // valid where in (after stage 3's fold: s = e = 0 elsewhere)
    Posting P[kGather];  // P[j] valid for j < e - s
};

// the group's state while it accumulates
struct GroupState {
    uint32_t walked;  // postings applied
    int tcnt;         // entries of the touched list
};

// appends the accumulators of the lanes with `add` to the touched list (entries past its capacity are counted, not stored:
// the read-back then scans the accumulators instead)
__device__ inline void touch(uint16_t *touched, int tcap, GroupState &gs, bool add, int i, int lane) {
    const unsigned long long m = __ballot(add);
    if (!m) return;
    const int at = gs.tcnt + lanes_below(m, lane);
    if (add && at < tcap) touched[at] = (uint16_t)i;
    gs.tcnt += __popcll(m);
}

// 64 lists at once (every segment of the visit has at most kGather postings, already in v.P).
// Rounds: every pending lane stamps the rows of its list with its lane number and reads the stamps back; a lane that finds a
// foreign stamp shares a row with another pending list (it "lost" that row).  Let lim be the lowest loser: every pending lane
// below it won all its rows, and whoever else holds one of those rows is a loser ABOVE lim, i.e. a later list -- so the lanes
// below lim are applied together (their rows are distinct among themselves), then lane lim alone (all earlier lists are in),
// and the lanes above lim go round again.  Without sharing: one round.
// Rows whose accumulator was +0 before the add go on the touched list (a sum that returns to zero and is reached again is
// listed twice; the read-back takes it once).
// DEPTH = the longest list of the chunk, rounded up to 2, 4 or kGather: the loops below run over DEPTH postings per lane, and a
// chunk of the tail groups (a few lists with one or two postings each) pays for 2, not for 8.
template <bool ATOMIC, bool TRACE, int DEPTH>
__device__ inline void apply_at_once(const Visit &v, float *acc, volatile lds_u8 *tag, uint16_t *touched, int tcap, int lane,
                                     GroupState &gs, Tracer<TRACE> &tr, int lm) {
    const uint32_t len = v.e - v.s;
    bool pending = len > 0;
    tr.add(&Trace::fast_chunks);
    for (;;) {
        tr.add(&Trace::rounds);
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
            if (pending && (uint32_t)j < len) tag[(v.P[j].loc & lm)] = (uint8_t)lane;
        bool lost = false;
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            uint32_t stamp = (uint32_t)lane;
            if (pending && (uint32_t)j < len) stamp = tag[(v.P[j].loc & lm)];
            lost = lost | (stamp != (uint32_t)lane);
        }
        const unsigned long long ml = __ballot(lost);
        const int lim = ml ? __ffsll((long long)ml) - 1 : 64;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {  // the lanes below lim together, then lane lim alone
            if (pass == 1 && !ml) break;
            const bool go = pending && (pass == 0 ? lane < lim : lane == lim);
            float old[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; j++) {
                old[j] = 1.0f;
                if (go && (uint32_t)j < len) old[j] = acc_add_old<ATOMIC>(acc, (v.P[j].loc & lm), __fmul_rn(v.qv, v.P[j].val));
            }
#pragma unroll
            for (int j = 0; j < DEPTH; j++) {
                const unsigned long long m = __ballot(go && (uint32_t)j < len);
                if (!m) break;
                gs.walked += (uint32_t)__popcll(m);
                touch(touched, tcap, gs, __float_as_uint(old[j]) == 0, (v.P[j].loc & lm), lane);
            }
        }
        if (!ml) break;
        pending = pending && lane > lim;
        if (!__ballot(pending)) break;
    }
}

// Flattened batches (visits with a segment longer than kGather).  The segments of the visit are laid end to end in list order
// and cut into batches of 64 postings, one per lane: a batch holds the tail of one list, whole lists, and the head of another.
// Within one list the rows are distinct; two lists of a batch may share a row, and then the earlier list's product has to go
// first.  A batch that mixes lists stamps its rows (as apply_at_once does); where a lane reads a foreign stamp, the lanes that
// share that row are found by a ballot and ranked by lane number = list order, and the batch is applied in rank order: round r
// adds the r-th sharer of every row, all of them to different accumulators.  The LDS executes the rounds in issue order, so
// every accumulator still receives its products in ascending index order.  (One list at a time the C3-shard item-to-item
// pass spent 0.27 us per SEGMENT of 4 postings on average -- 3.2e8 of them, profiles/r02_k_probe_sparse_trace.txt.)
// Assembly is a wave-uniform walk over the lists; kFlatAhead batches are in flight while one is applied.
// HASH (the super-visits of the tail groups): the rows of several groups share the workgroup's accumulator words as an open-addressed
// table -- word h of the first half holds a row's scratch id + 1 (0 = free), word S + h its sum -- so the accumulator of a posting
// is found by probing (multiplicative hash, linear probing, ds_cmpst) instead of loc & lm; everything after that (stamps, rank
// rounds, ds_add_f32 in issue order) is the same, and `touched` lists the slots this visit claimed.  The caller guarantees
// at most S / 2 postings per super-visit, so the table never fills.
template <bool ATOMIC, bool TRACE, bool HASH = false>
__device__ inline void apply_flattened(const Posting *__restrict__ post, const Visit &v, float *acc, volatile lds_u8 *tag,
                                       uint16_t *touched, int tcap, int lane, GroupState &gs, Tracer<TRACE> &tr, int lm,
                                       volatile lds_u8 *board) {
    struct Batch {
        Posting P;
        float q;
        int n;       // postings of the batch (wave-uniform); 0: none left
        bool mixed;  // more than one list
    };
    // The segments laid end to end: lane l's list covers the positions [start_l, incl_l) of the concatenation (an inclusive scan
    // of the lengths over the lanes, DPP only); batch b is the positions [64 b, 64 b + 64).  A position finds its list without a walk
    // over the lists and without a dependent chain: every list that STARTS inside the batch's window writes its lane number to
    // board[start - p0] (64 bytes of LDS), every position reads its byte back, and a prefix maximum over the lanes (DPP) carries the
    // last start forward; positions in front of the first start belong to the list that continues from the batch before (the first
    // lane whose prefix exceeds p0: one ballot).  Two LDS round trips per batch (board, then the owner's segment start / value by
    // ds_bpermute).  Until round 4 the batch was assembled by a wave-uniform walk over the lists -- three v_readlane and a dozen scalar
    // operations per list, ~330 cycles for a list that contributes two postings (profiles/r04_g_probe_sparse_trace.txt: 2.4e8 such
    // segments); a binary search over the prefixes (six dependent ds_bpermute) was no faster (profiles/r04_j_*_bsearch.txt).
    const uint32_t len = v.e - v.s;
    auto dpp_scan = [&](uint32_t x, auto op) {  // inclusive scan over the 64 lanes; 0 is the identity of op
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));  // row_shr:1
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));  // row_shr:2
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));  // row_shr:4
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));  // row_shr:8
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
        return x;
    };
    const uint32_t incl = dpp_scan(len, [](uint32_t a, uint32_t b) { return a + b; });
    const uint32_t total = lane_u32(incl, kBlock - 1);
    const uint32_t start = incl - len;
    uint32_t next_p = 0;  // first position of the next batch to assemble
    auto bperm = [&](uint32_t val, uint32_t from) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)val); };
    auto fill = [&](Batch &b) {
        const uint32_t p0 = next_p;
        b.n = p0 < total ? (int)min(total - p0, (uint32_t)kBlock) : 0;
        next_p += kBlock;
        if (p0 >= total) {  // nothing left: the slot's one load all the same (see below), none of the assembly
            b.mixed = false;
            b.q = 0.0f;
            b.P = post[0];
            return;
        }
        const uint32_t rel = start - p0;
        if (len > 0 && rel < (uint32_t)kBlock) board[rel] = (uint8_t)(lane + 1);
        uint32_t f = board[lane];
        board[lane] = 0;
        f = dpp_scan(f, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
        const unsigned long long over = __ballot(incl > p0);  // the list that continues into this window (or starts at p0)
        const uint32_t carry = over ? (uint32_t)(__ffsll((long long)over) - 1) : 0u;
        const uint32_t owner = f ? f - 1 : carry;
        const uint32_t o_start = bperm(start, owner), o_s = bperm(v.s, owner);
        b.q = __uint_as_float(bperm(__float_as_uint(v.qv), owner));
        const bool have = lane < b.n;
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)owner);
        b.mixed = __ballot(have && owner != first) != 0;
        if (TRACE) tr.add(&Trace::slow_segments, b.n > 0 ? (uint32_t)(__builtin_amdgcn_readlane((int)owner, b.n - 1) - (int)first + 1) : 0);
        b.P = post[have ? o_s + (p0 + (uint32_t)lane - o_start) : 0];  // every fill issues exactly one load (lanes past n read
                                                                       // posting 0 and are not applied): the wait for a batch is then
                                                                       // "all but the kFlatAhead - 1 younger loads", not "all loads"
    };
    auto apply = [&](const Batch &b) {
        const bool have = lane < b.n;
        int32_t row = b.P.loc & lm;
        bool claimed = false;
        float *sums = acc;
        if constexpr (HASH) {  // lm = S - 1
            const int S = lm + 1;
            lds_key *hkey = reinterpret_cast<lds_key *>(acc);
            sums = acc + S;
            if (have) {
                const int32_t key = b.P.loc + 1;
                uint32_t hs = ((uint32_t)b.P.loc * 2654435761u) >> (__builtin_clz((unsigned)S) + 1);
                for (;;) {
                    const int32_t old = atomicCAS(&hkey[hs], 0, key);
                    claimed = old == 0;
                    if (old == 0 || old == key) break;
                    hs = (hs + 1) & (uint32_t)lm;
                }
                row = (int32_t)hs;
            }
        }
        const float term = __fmul_rn(b.q, b.P.val);
        unsigned long long lost = 0;
        if (b.mixed) {
            if (have) tag[row] = (uint8_t)lane;
            uint32_t stamp = (uint32_t)lane;
            if (have) stamp = tag[row];
            lost = __ballot(stamp != (uint32_t)lane);
        }
        if (!lost) {
            if (have) acc_add<ATOMIC>(sums, row, term);
        } else {
            int rank = 0, last = 0;
            while (lost) {
                const int l = __ffsll((long long)lost) - 1;
                const int32_t r = __builtin_amdgcn_readlane(row, l);
                const unsigned long long same = __ballot(have && row == r);
                if (have && row == r) rank = lanes_below(same, lane);
                lost &= ~same;
                const int sharers = (int)__popcll(same) - 1;
                last = sharers > last ? sharers : last;
                tr.add(&Trace::shared_rows);
            }
            for (int rd = 0; rd <= last; rd++)
                if (have && rank == rd) acc_add<ATOMIC>(sums, row, term);
        }
        touch(touched, tcap, gs, HASH ? claimed : have, row, lane);
        gs.walked += (uint32_t)b.n;
        tr.add(&Trace::batches);
    };
    // kFlatAhead batches in flight; a slot is refilled right after it was applied (its own load has been waited for, so the
    // refill never waits for a register that may still be a load's destination)
    if (total <= (uint32_t)(2 * kBlock)) {  // one or two batches (most calls of the tail windows): no ring
        Batch b0, b1;
        fill(b0);
        fill(b1);
        apply(b0);
        if (b1.n > 0) apply(b1);
        return;
    }
    constexpr int kFlatAhead = 6;
    Batch ring[kFlatAhead];
#pragma unroll
    for (int i = 0; i < kFlatAhead; i++) fill(ring[i]);
    bool more = ring[0].n > 0;
    while (more) {
#pragma unroll
        for (int slot = 0; slot < kFlatAhead; slot++) {
            if (ring[slot].n == 0) {
                more = false;
                break;
            }
            apply(ring[slot]);
            fill(ring[slot]);
        }
    }
}

// Measured (profiles/r06_d_ab_sparse_queue.txt, both builds in one session, twice): ONE counter 36.64-36.69 ms per C3-shard pass, 32 stripes
// 38.66-38.71 ms.  The head of the queue on one memory line is not what the pass waits for -- 3e5 items in 36 ms are 8 M returning
// atomics per second against the 88 M a line serves, and a wave meets the counter ~70 times per launch -- while the stripes cost the
// dearest-first order its meaning at the END of the launch: a wave drains its home stripe and then walks the others one by one,
// so the last, cheap items of 32 lists are found late.  The single counter stays; the switch is kept for the record.
#ifndef GORSE_SPARSE_QUEUE_STRIPES
#define GORSE_SPARSE_QUEUE_STRIPES 1  // make ab AB=q32 AB_SRC=sparse AB_FLAGS=-DGORSE_SPARSE_QUEUE_STRIPES=32
#endif
constexpr int kQueueStripes = GORSE_SPARSE_QUEUE_STRIPES;  // a power of two
constexpr int kQueueStride = 64;   // words between two stripes' counters: 256 bytes
constexpr int kQueueBase = 64;     // words 0 .. 15 of the buffer: the heavy-query kernel's eight queues (RowsArgs::next = next + 8)
constexpr int kQueueWords = kQueueBase + kQueueStripes * kQueueStride;

// the k-th largest of the wave's 64 R values (0 = fewer than k of them are non-zero): bit by bit, one ballot per value and bit
template <int R>
__device__ inline uint32_t kth_largest_ord(const uint32_t (&o)[R], int k) {
    uint32_t kth = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t trial = kth | (1u << b);
        int cge = 0;
#pragma unroll
        for (int j = 0; j < R; j++) cge += __popcll(__ballot(o[j] >= trial));
        if (cge >= k) kth = trial;
    }
    return kth;
}

// the k-th largest ord among the first 256 scores of a column of the front's matrix (SymArgs::fm), 0 = fewer than k are there.
// Not inlined: the list walk's registers are a budget (three of its waves + one of sparse_rows_kernel per SIMD).
__device__ __attribute__((noinline)) uint32_t front_bound(const float *col, int64_t Ns, int front, int k, int lane) {
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int slot = j * kBlock + lane;
        const float x = slot < front ? __hip_atomic_load(&col[(size_t)slot * Ns], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        o[j] = (__float_as_uint(x) << 1) != 0 ? score_ord(x) : 0u;
    }
    return kth_largest_ord<4>(o, k);
}

// SYMMODE: 0 = the unsymmetric walk, 1 = the symmetric form (SymArgs), 2 = the symmetric form with a delivering front.  Two
// instantiations rather than one with a switch: the bound of an item comes from its foreign list (sym_tighten) in one and from the
// front's column (front_bound) in the other, and either alone keeps the kernel at 112 registers.
template <int KP, bool ATOMIC, bool TRACE, int SYMMODE = 0>
__global__ __launch_bounds__(kBlock) void sparse_tile_kernel(TileArgs a) {
    constexpr bool SYM = SYMMODE != 0, FRONT = SYMMODE == 2;
    constexpr int CAP = 2 * KP;
    extern __shared__ __align__(16) unsigned char s_mem[];
    const int NL = 1 << a.logG;  // accumulators in LDS
    unsigned long long *s_buf = reinterpret_cast<unsigned long long *>(s_mem);
    float *acc = reinterpret_cast<float *>(s_mem + (size_t)CAP * 8);
    volatile lds_u8 *tag = (volatile lds_u8 *)(s_mem + (size_t)CAP * 8 + (size_t)NL * 4);
    uint16_t *touched = reinterpret_cast<uint16_t *>(s_mem + (size_t)CAP * 8 + (size_t)NL * 5);  // NL / 4 entries
    volatile lds_u8 *board = (volatile lds_u8 *)(s_mem + (size_t)CAP * 8 + (size_t)NL * 5 + (size_t)NL / 2);  // 64 bytes (apply_flattened)
    const int lane = threadIdx.x;
    board[lane] = 0;
    for (int i = lane; i < NL; i += kBlock) acc[i] = 0.0f;
    __syncthreads();
    // The work list is taken through kQueueStripes counters (1 in the shipped build: see above), each on a 256-byte line of its own:
    // stripe s hands out the items s, s + kQueueStripes, ... (the list is sorted dearest first, so every stripe is too).  A wave starts
    // on its workgroup's stripe and moves on when a stripe runs dry; a dry stripe stays dry, so kQueueStripes dry ones in a row end it.
    int qstripe = (int)(blockIdx.x & (kQueueStripes - 1)), qdry = 0;
    for (;;) {
        int w = 0;
        if (lane == 0) w = atomicAdd(a.next + kQueueBase + qstripe * kQueueStride, 1);
        w = __builtin_amdgcn_readfirstlane(w) * kQueueStripes + qstripe;
        if (w >= a.n_work) {
            if (++qdry == kQueueStripes) break;
            qstripe = (qstripe + 1) & (kQueueStripes - 1);
            continue;
        }
        const Work wk = a.work[w];
        if (wk.prio)
            __builtin_amdgcn_s_setprio(3);
        else
            __builtin_amdgcn_s_setprio(0);
        const int64_t t = wk.t, qr = a.q_first + t;
        const int64_t qs = a.q_ptr[qr];
        const int L = (int)(a.q_ptr[qr + 1] - qs);  // a query's indices are distinct uint32 and the host caps L below 2^31
        const int64_t ex = a.exclude ? a.exclude[t] : (a.exclude_self ? qr : (int64_t)-1);
        const bool ex_in = ex >= 0 && ex < a.N;
        const int32_t ex_sid = ex_in ? a.new_of[ex] : -1;
        int bcnt = 0;
        unsigned long long thr = 0;
        int my_pos = 0, my_neg = 0, my_hit = 0;  // per lane: at most N / 64 + 1 each
        unsigned long long walked_q = 0;
        Tracer<TRACE> tr;
        if constexpr (TRACE) tr.r.t0 = tr.now();
        // the views of this item: every group, or the one group of this part of a long query
        const bool whole = wk.part < 0;
        const uint32_t *off = a.off;
        const Posting *post = a.post;
        const int dir_stride = a.ngroups;
        // SYM: a whole-query item takes the rows in front of its own (and the timing probe stops there too)
        const bool symw = SYM && whole;
        const int32_t sid_q = SYM || (whole && a.tri_probe && qr < a.N) ? a.new_of[qr] : 0;
        const int glim = symw || (whole && a.tri_probe && qr < a.N) ? (sid_q >> a.logG) + 1 : a.ngroups;
        const int gfirst = FRONT && symw ? 1 : 0;  // SYM with a front: group 0 is the front's, which delivers (SymArgs)
        uint32_t pub = 0;  // SYM: the bound this item has published for its row
        // (a part of a front row: where its scores go, indexed by the other row's scratch id)
        float *fm_row = FRONT && !whole ? a.sym.fm + ((size_t)sid_q * a.sym.Ns - a.sym.first) : nullptr;
        if constexpr (FRONT) {
            // the front's column of this row, as far as it is written (a score is stored once, whole; what is missing reads as the
            // cleared 0): the k-th largest ord among the scores of the 64 R longest rows opens the item's threshold and is published
            if (symw) {
                const uint32_t kth = front_bound(a.sym.fm + (sid_q - a.sym.first), a.sym.Ns, a.sym.front, a.k, lane);
                if (kth > 0) {
                    pub = kth;
                    thr = ((unsigned long long)kth << 32) - 1;
                    if (lane == 0) __hip_atomic_store(reinterpret_cast<uint32_t *>(&a.sym.tp[sid_q]), pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // (the item's first loads stay behind this: the block's registers are free again)
        // a whole-query item: the head groups [gfirst, ghead) here, the others in super-visits below
        const int ghead = a.head_groups < gfirst ? gfirst : (a.head_groups < glim ? a.head_groups : glim);
        const int nviews = whole ? ghead - gfirst : 1;
        const int nacc = NL;
        const int lm = NL - 1;
        const int tcap = nacc >> 2;
        const int nch = (int)(((int64_t)L + kBlock - 1) / kBlock);  // chunks of 64 indices
        const int64_t V = (int64_t)nch * nviews;                    // visits, view-major
        // The pipeline: while visit v is applied, the postings of v + 1, the directory entries of v + 2 and v + 3 and the (index,
        // value) pairs of v + 4 and v + 5 are in flight (round 5: one visit deeper each -- a visit of a tail group applies a few dozen
        // postings in less time than a load takes, and the registers are there since the tail left this loop).  (A query of one chunk re-reads its 64 pairs at every visit: two cached loads, and no
        // branch in the loop that would make the compiler wait for everything in flight.)
        // stage 1 and stage 2 are each called for v = 0, 1, 2, ... in turn: they keep their own (view, chunk) counters
        int c1 = 0, c2 = 0;
        int g2 = whole ? gfirst : wk.part;  // group of the visit stage 2 is at
        // Every stage issues the SAME loads on every path (clamped addresses, results masked afterwards; the host pads each
        // array by one element): a load that a branch may skip makes the compiler's wait for any OLDER load "wait for all".
        auto stage1 = [&](int64_t v, Visit &x) {
            const int at = c1 * kBlock + lane;
            x.in = v < V && at < L;
            x.cid = a.q_cid[qs + (x.in ? at : 0)];
            x.qv = a.q_val[qs + (x.in ? at : 0)];
            if (++c1 == nch) c1 = 0;
        };
        auto stage2 = [&](int64_t v, Visit &x) {  // folds stage 1: in &= the index is stored
            x.in = x.in && x.cid >= 0;
            const uint32_t *o = off + (x.in ? (size_t)x.cid * dir_stride + g2 : (size_t)0);
            x.s = o[0], x.e = o[1];
            if (++c2 == nch) {
                c2 = 0;
                g2++;
            }
        };
        auto stage3 = [&](Visit &x) {  // folds stage 2: s = e = 0 where the lane has nothing
            if (!x.in) x.s = 0, x.e = 0;
            const uint32_t len = x.e - x.s;
            const bool once = !__ballot(len > (uint32_t)kGather);  // a visit with a longer segment takes the flattened path, which loads
                                                                   // its postings itself: eight gathers of 64 lines each for nothing
#pragma unroll
            for (int j = 0; j < kGather; j++)  // (what the loop top waits for is everything in flight anyway: skipping costs nothing)
                if (once && __ballot((uint32_t)j < len)) x.P[j] = post[x.s + ((uint32_t)j < len ? j : 0)];
        };
        Visit v0, v1, v2, v3, v4, v5;
        stage1(0, v0);
        stage1(1, v1);
        stage1(2, v2);
        stage1(3, v3);
        stage1(4, v4);
        stage2(0, v0);
        stage2(1, v1);
        stage2(2, v2);
        stage3(v0);
        GroupState gs{0, 0};
        // one candidate per lane: sid = the row's scratch id, x = its sum; og = orig_of[sid] where the caller has loaded it
        auto consider = [&](bool have, int32_t sid, float x, int32_t og, bool og_loaded) {
            have = have && (__float_as_uint(x) << 1) != 0;  // a zero score is dropped by the reference's wrapper
            if (!__ballot(have)) return;
            my_hit += have;
            if constexpr (FRONT) {  // a part of a front row: the score is the other row's candidate too
                if (fm_row && have && sid >= a.sym.first) fm_row[sid] = x;
            }
            if (symw)
                have = have && sid < sid_q;
            else
                have = have && sid != ex_sid && (!a.mask_sid || a.mask_sid[have ? sid : 0]);
            const uint32_t ord = score_ord(x);
            my_pos += have && ord > kZeroOrd;
            my_neg += have && ord < kZeroOrd;
            if constexpr (SYM) {
                const bool dl = symw && have && sid >= a.sym.first;
                if (__ballot(dl)) {  // the other row's side of the pair
                    const int32_t rs = dl ? sid : a.sym.first;
                    const unsigned long long tp = sym_load_tp(&a.sym.tp[rs]);
                    // a bound above zero says the row holds k positive scores already: its counts no longer matter (written())
                    const bool counts = dl && (uint32_t)tp <= kZeroOrd;
                    if (counts && ord > kZeroOrd && (uint32_t)(tp >> 32) < (uint32_t)a.k)
                        __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(&a.sym.tp[rs]) + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (counts && ord < kZeroOrd) __hip_atomic_fetch_add(&a.sym.neg[rs], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (dl && ord >= (uint32_t)tp) {
                        const uint32_t slot = __hip_atomic_fetch_add(&a.sym.fcnt[rs], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        uint32_t cap;
                        const int64_t at = sym_list_at(a.sym, rs, cap);
                        if (slot < cap)
                            __hip_atomic_store(&a.sym.flist[at + slot], make_key(ord, (int32_t)qr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            const bool cand = have && ord >= (uint32_t)(thr >> 32);
            if (!__ballot(cand)) return;
            if (!og_loaded) og = cand ? a.orig_of[sid] : 0;
            const unsigned long long key = cand ? make_key(ord, og) : 0;
            push<KP>(s_buf, a.k, bcnt, thr, key, cand && key > thr, lane);
            if constexpr (SYM) {
                if (symw && (uint32_t)(thr >> 32) > pub) {  // the ranking's threshold moved: a better bound for whoever delivers here
                    pub = (uint32_t)(thr >> 32);
                    if (lane == 0)
                        __hip_atomic_store(reinterpret_cast<uint32_t *>(&a.sym.tp[sid_q]), pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };
        // reads group gg's directly indexed accumulators back (a scan, or the touched list) and leaves them zero
        auto read_back = [&](int gg) {
            const unsigned long long c0 = tr.now();
            if (gs.walked > 0) {
                // og = orig_of[sid], loaded one step ahead of its use (the read-back used to wait for this gather inside every step
                // that had a candidate: 43 us per group of 2048 accumulators, profiles/r02_k_probe_sparse_trace.txt)
                auto sid_of = [&](int32_t i) { return (int32_t)((gg << a.logG) + i); };  // N fits int32
                auto orig_at = [&](bool in, int32_t i) {
                    const int32_t sid = sid_of(i);
                    return in && sid < a.Np ? a.orig_of[sid] : 0;
                };
                if ((int64_t)gs.walked * 4 >= nacc || gs.tcnt > tcap) {
                    tr.add(&Trace::dense_groups);
                    constexpr int kStep = 4;
                    int32_t og_next[kStep];
#pragma unroll
                    for (int j = 0; j < kStep; j++) og_next[j] = orig_at(j * kBlock + lane < nacc, j * kBlock + lane);
                    for (int i0 = 0; i0 < nacc; i0 += kStep * kBlock) {
                        int32_t og[kStep];
                        float x[kStep];
#pragma unroll
                        for (int j = 0; j < kStep; j++) {
                            const int i = i0 + j * kBlock + lane, in = i0 + (kStep + j) * kBlock + lane;
                            og[j] = og_next[j];
                            og_next[j] = orig_at(in < nacc, in);
                            x[j] = i < nacc ? acc[i] : 0.0f;
                            if (__float_as_uint(x[j]) != 0) acc[i] = 0.0f;
                        }
#pragma unroll
                        for (int j = 0; j < kStep; j++)
                            consider(i0 + j * kBlock + lane < nacc, sid_of(i0 + j * kBlock + lane), x[j], og[j], true);
                    }
                } else {
                    tr.add(&Trace::sparse_groups);
                    for (int i0 = 0; i0 < gs.tcnt; i0 += kBlock) {
                        const bool have = i0 + lane < gs.tcnt;
                        const int i = have ? (int)touched[i0 + lane] : 0;
                        const float x = have ? acc_take(acc, i) : 0.0f;  // an accumulator listed twice: the first taker gets it
                        consider(have, sid_of(i), x, 0, false);  // few candidates once the threshold has risen: the gather is the exception
                    }
                }
                walked_q += gs.walked;
            }
            if constexpr (TRACE) {
                const unsigned long long c1 = tr.now();
                tr.r.ticks_back += (uint32_t)(c1 - c0);
                if (whole && gg == 7) tr.r.ticks_head = (uint32_t)(c1 - tr.r.t0);
            }
            gs = GroupState{0, 0};
        };
        // SYM: a bound from what has been DELIVERED to this row so far.  A row of the first groups publishes nothing of its own until
        // its first read-back, which for the longest whole-query rows is milliseconds away (2000 lists over the rows that hold half of
        // all entries) -- and half the launch delivers to it meanwhile, unfiltered (20,000 entries in one list, r06_x).  Between two
        // visits the item looks at its foreign list: any k delivered scores bound the k-th key from below, so the k-th largest ord of
        // the entries that arrived since the last look is published (if higher).  Only the zeroed head of the list is read (kSymLook
        // entries, cleared by the host before the pass for the rows in front of SymArgs::tl): a slot handed out but not yet written
        // reads as 0.
        uint32_t seen = 0;
        auto sym_tighten = [&]() {
            if (sid_q >= a.sym.tl || seen >= (uint32_t)kSymLook) return;
            uint32_t fc = 0;
            if (lane == 0) fc = __hip_atomic_load(&a.sym.fcnt[sid_q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fc = (uint32_t)__builtin_amdgcn_readfirstlane((int)fc);
            uint32_t cap;
            const int64_t at = sym_list_at(a.sym, sid_q, cap);
            if (fc > cap) fc = cap;
            if (fc > (uint32_t)kSymLook) fc = (uint32_t)kSymLook;
            // (a window of R entries per lane: 4 keeps the kernel at the 3 waves per SIMD + one wave of sparse_rows_kernel beside them)
            constexpr int R = KP <= 256 ? 4 : 16;
            constexpr uint32_t kWindow = (uint32_t)(R * kBlock);
            const uint32_t need = (uint32_t)a.k + ((uint32_t)a.k >> 1) > 256u ? (uint32_t)a.k + ((uint32_t)a.k >> 1) : 256u;
            if (fc < seen + (need < kWindow ? need : kWindow)) return;
            const uint32_t n = fc - seen < kWindow ? fc - seen : kWindow;
            uint32_t o[R];
#pragma unroll
            for (int j = 0; j < R; j++) {
                const uint32_t i = (uint32_t)(j * kBlock + lane);
                const unsigned long long key = i < n ? __hip_atomic_load(&a.sym.flist[at + seen + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                o[j] = (uint32_t)(key >> 32);
            }
            seen += n;
            const uint32_t kth = kth_largest_ord<R>(o, a.k);
            if (kth > pub) {
                pub = kth;
                if (lane == 0) __hip_atomic_store(reinterpret_cast<uint32_t *>(&a.sym.tp[sid_q]), pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        int c = 0;  // chunk of visit v inside its view
        int g = whole ? gfirst : wk.part;
        for (int64_t v = 0; v < V; v++) {
            // consumers first: each stage needs what the stage before it loaded during the PREVIOUS visit, so whatever the compiler
            // waits for here has had a whole visit to arrive
            stage3(v1);
            stage2(v + 3, v3);
            stage1(v + 5, v5);
            {   // visit v
                const uint32_t len = v0.e - v0.s;
                if (__ballot(len > 0)) {
                    const unsigned long long c0 = tr.now();
                    const bool once = !__ballot(len > (uint32_t)kGather);
                    if (once) {
                        if (!__ballot(len > 2u))
                            apply_at_once<ATOMIC, TRACE, 2>(v0, acc, tag, touched, tcap, lane, gs, tr, lm);
                        else if (!__ballot(len > 4u))
                            apply_at_once<ATOMIC, TRACE, 4>(v0, acc, tag, touched, tcap, lane, gs, tr, lm);
                        else
                            apply_at_once<ATOMIC, TRACE, kGather>(v0, acc, tag, touched, tcap, lane, gs, tr, lm);
                    } else
                        apply_flattened<ATOMIC, TRACE>(post, v0, acc, tag, touched, tcap, lane, gs, tr, lm, board);
                    tr.add(once ? &Trace::ticks_once : &Trace::ticks_flat, (uint32_t)(tr.now() - c0));
                }
            }
            if (++c == nch) {  // the view is complete: read it back
                read_back(g);
                c = 0;
                g++;
            }
            if constexpr (SYM && !FRONT) {
                if (symw) sym_tighten();
            }
            v0 = v1, v1 = v2, v2 = v3, v3 = v4, v4 = v5;
        }
        // ---- the tail groups of a whole-query item: SUPER-VISITS ------------------------------------------------------------
        // Behind the head groups a (query, group) visit finds a few dozen postings spread over as many lists, and its fixed cost
        // (three pipeline stages, stamps, read-back: ~460 instructions) was 70 % of the C3-shard pass (profiles/r02_af_probe_sparse_trace.txt).
        // Here several consecutive groups are taken at once -- a WINDOW: the directory gives every list's postings in [gb, ge) as ONE
        // contiguous segment (off is a prefix array over (list, group)); the windows are cut so that at most S / 2 postings -- hence at
        // most that many distinct rows -- are met, and they are accumulated through apply_flattened<HASH> in an open-addressed table
        // that lives in the accumulator words.  Chunk after chunk, list after list: every row still receives its products in
        // ascending index order.  A single group with more postings than that takes the direct accumulators (apply_flattened).
        // Round 5: the windows are PLANNED, 63 groups at a time, before any of them is visited -- lane i sums the directory entries
        // off[c][g0 + i] over the lists of the query (one coalesced 256-byte read per list, eight in flight), the differences of
        // neighbouring lanes are the postings the query meets per group, and a scalar greedy pass cuts the windows (lane j of `plan`
        // = window j).  Until then every window was sized by its own count pass (two dependent loads per chunk, then three more per
        // chunk to apply it: ~5 exposed round trips per (window, chunk)); with the plan known the visits run through the same kind of
        // software pipeline as the head groups: index / value pairs two visits ahead, directory entries one visit ahead.
        if (whole && ghead < glim) {
            const int S = nacc >> 1;
            const uint32_t cap_t = (uint32_t)(nacc >> a.cap_shift);
            lds_key *hkey = reinterpret_cast<lds_key *>(acc);
            float *hval = acc + S;
            for (int g0 = ghead; g0 < glim; g0 += kBlock - 1) {
                const int gcount = glim - g0 < kBlock - 1 ? glim - g0 : kBlock - 1;  // groups g0 .. g0 + gcount - 1
                // -- the plan
                const int gi = g0 + (lane < gcount ? lane : gcount);
                uint32_t psum = 0;  // sum over the lists of off[c][gi], modulo 2^32 (the differences are what is used)
                for (int ch = 0; ch < nch; ch++) {
                    const int at = ch * kBlock + lane;
                    int32_t cid = a.q_cid[qs + (at < L ? at : 0)];
                    if (at >= L) cid = -1;
                    const int nl = L - ch * kBlock < kBlock ? L - ch * kBlock : kBlock;
                    for (int j0 = 0; j0 < nl; j0 += 8) {
                        uint32_t tv[8];
                        int32_t cj[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            cj[u] = __builtin_amdgcn_readlane(cid, j0 + u);
                            tv[u] = off[(size_t)(cj[u] < 0 ? 0 : cj[u]) * dir_stride + gi];
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) psum += cj[u] < 0 ? 0u : tv[u];
                    }
                }
                const uint32_t pnext = (uint32_t)__builtin_amdgcn_ds_bpermute((lane < kBlock - 1 ? lane + 1 : lane) << 2, (int)psum);
                const uint32_t tot = lane < gcount ? pnext - psum : 0u;  // postings of the query in group g0 + lane
                uint32_t plan = 0;  // lane j: window j = begin | end << 8 (relative to g0) | direct << 16
                int nwin = 0;
                {
                    int wb = 0;
                    uint32_t sum = 0;
                    auto emit = [&](int b, int e, uint32_t direct) {
                        if (lane == nwin) plan = (uint32_t)b | ((uint32_t)e << 8) | (direct << 16);
                        nwin++;
                    };
                    for (int i = 0; i < gcount; i++) {
                        const uint32_t ti = lane_u32(tot, i);
                        if (ti > cap_t) {  // more than the table takes: the group on its own, directly indexed accumulators
                            if (sum > 0) emit(wb, i, 0);
                            emit(i, i + 1, 1);
                            wb = i + 1, sum = 0;
                        } else if (sum + ti > cap_t) {
                            emit(wb, i, 0);
                            wb = i, sum = ti;
                        } else
                            sum += ti;
                    }
                    if (sum > 0) emit(wb, gcount, 0);
                }
                // -- the visits: window-major, chunk after chunk
                const int64_t Vb = (int64_t)nwin * nch;
                int t1 = 0, t2 = 0, j2 = 0;
                auto tstage1 = [&](int64_t v, Visit &x) {
                    const int at = t1 * kBlock + lane;
                    x.in = v < Vb && at < L;
                    x.cid = a.q_cid[qs + (x.in ? at : 0)];
                    x.qv = a.q_val[qs + (x.in ? at : 0)];
                    if (++t1 == nch) t1 = 0;
                };
                auto tstage2 = [&](Visit &x) {  // folds stage 1; the window of the visit from the plan
                    x.in = x.in && x.cid >= 0;
                    const uint32_t w = lane_u32(plan, j2 < kBlock ? j2 : kBlock - 1);
                    const size_t base = x.in ? (size_t)x.cid * dir_stride + g0 : (size_t)0;
                    x.s = off[base + (x.in ? (w & 255u) : 0u)];
                    x.e = off[base + (x.in ? ((w >> 8) & 255u) : 0u)];
                    if (++t2 == nch) {
                        t2 = 0;
                        j2++;
                    }
                };
                Visit x0, x1, x2;
                tstage1(0, x0);
                tstage1(1, x1);
                tstage2(x0);
                int tc = 0, jw = 0;
                for (int64_t v = 0; v < Vb; v++) {
                    tstage2(x1);
                    tstage1(v + 2, x2);
                    if (!x0.in) x0.s = 0, x0.e = 0;
                    const uint32_t w = lane_u32(plan, jw);
                    const bool direct = (w >> 16) != 0;
                    const int gg = g0 + (int)(w & 255u);
                    if (__ballot(x0.e > x0.s)) {
                        const unsigned long long c0 = tr.now();
                        if (direct)
                            apply_flattened<ATOMIC, TRACE>(post, x0, acc, tag, touched, tcap, lane, gs, tr, lm, board);
                        else
                            apply_flattened<ATOMIC, TRACE, true>(post, x0, acc, tag, touched, tcap, lane, gs, tr, S - 1, board);
                        tr.add(&Trace::ticks_flat, (uint32_t)(tr.now() - c0));
                    }
                    if (++tc == nch) {  // the window is complete: read it back
                        tc = 0;
                        jw++;
                        if (direct)
                            read_back(gg);
                        else {
                            const unsigned long long c1 = tr.now();
                            for (int i0 = 0; i0 < gs.tcnt; i0 += kBlock) {  // the slots this super-visit claimed, each once
                                const bool have = i0 + lane < gs.tcnt;
                                const int slot = have ? (int)touched[i0 + lane] : 0;
                                int32_t sid = 0;
                                float xsum = 0.0f;
                                if (have) {
                                    sid = hkey[slot] - 1;
                                    xsum = hval[slot];
                                    hkey[slot] = 0;
                                    hval[slot] = 0.0f;
                                }
                                consider(have, sid, xsum, 0, false);
                            }
                            tr.add(&Trace::ticks_back, (uint32_t)(tr.now() - c1));
                            tr.add(&Trace::sparse_groups);
                            walked_q += gs.walked;
                            gs = GroupState{0, 0};
                        }
                    }
                    x0 = x1, x1 = x2;
                }
            }
        }
        const long long pos = wave_sum((long long)my_pos), neg = wave_sum((long long)my_neg), hit = wave_sum((long long)my_hit);
        {   // No sort here.  Keys that go to a merge are taken in any order, a result row is written by counting (write_ranked): what
            // is needed is the best KP of the buffer -- a cut where it holds more, and for a symmetric item wherever it holds k: the
            // cut's threshold is the exact bound its own walk gives.
            __syncthreads();
            for (int i = bcnt + lane; i < CAP; i += kBlock) s_buf[i] = 0;
            __syncthreads();
            if (bcnt > KP || (symw && bcnt >= a.k)) cut_to_k<KP>(s_buf, a.k, bcnt, thr, lane);
        }
        if (symw) {  // the row's own half: its keys, its counts, and the exact bound of its own walk
            for (int i = lane; i < KP; i += kBlock) a.sym.own[(size_t)t * KP + i] = i < bcnt ? s_buf[i] : 0;
            if (lane == 0) {
                const uint32_t kth = bcnt >= a.k ? (uint32_t)((thr + 1) >> 32) : 0u;
                if (kth > pub)
                    __hip_atomic_store(reinterpret_cast<uint32_t *>(&a.sym.tp[sid_q]), kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pos) __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(&a.sym.tp[sid_q]) + 1, (uint32_t)pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (neg) __hip_atomic_fetch_add(&a.sym.neg[sid_q], (uint32_t)neg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (whole) {
            const bool ex_counts = ex_in && (!a.mask_sid || a.mask_sid[ex_sid]);
            const int cnt = written(pos, neg, a.n_admissible - (ex_counts ? 1 : 0), a.k);
            write_ranked<KP>(s_buf, bcnt, cnt, a.k, t, a.out_idx, a.out_score, a.out_cnt, lane);
        } else {
            const size_t part = (size_t)wk.pslot * a.part_stride + wk.part;
            for (int i = lane; i < KP; i += kBlock) a.part_keys[part * KP + i] = i < bcnt ? s_buf[i] : 0;
            if (lane == 0) {
                a.part_cnt[part * 2] = (int32_t)pos;
                a.part_cnt[part * 2 + 1] = (int32_t)neg;
            }
        }
        if (lane == 0 && a.stat) {
            atomicAdd(&a.stat[0], walked_q);
            atomicAdd(&a.stat[1], (unsigned long long)hit);
        }
        if constexpr (TRACE) {
            if (lane == 0) {
                tr.r.t1 = tr.now();
                tr.r.t = wk.t, tr.r.part = wk.part, tr.r.entries = (uint32_t)L;
                a.trace[w] = tr.r;
            }
        }
        __syncthreads();  // s_buf is reused by the next work item
    }
}

// ---- heavy queries: every stored row against the dense query ------------------------------------------------------------
// The query's (index, value) pairs are scattered into dense[c] = {1, value bits} over the directory entries c (zero = the
// query does not hold the index).  One lane scores one stored row: its entries in storage order = ascending index, each
// looked up in the dense vector, the products of the matches added one after the other -- the oracle's merge-order sum, with
// no accumulator shared between lanes.  A work item = (heavy query, row range): the range's rows 64 at a time, longest rows
// first, so the lanes of a step have rows of similar length; it leaves a partial ranking like a part of a long query.  The
// ranges are cut by cost; a long row is walked by the whole wave (64 consecutive entries per step, added in entry order) and the
// longest rows are ranges of their own -- as row groups of 2048 the first group was 50 ms on one wave, and with one lane per row
// the longest row alone 36 ms (profiles/r02_s_probe_sparse_c3.txt, r02_u_kernel_stats.txt).
struct RowsArgs {
    const int64_t *r_ptr;  // stored rows in the caller's order: r_ptr[N + 1], r_cid / r_val
    const int32_t *r_cid;
    const float *r_val;
    const int32_t *orig_of;  // scratch id -> caller's row
    int64_t N;
    const int32_t *range_start;  // n_ranges + 1 scratch ids (the last = N): ranges of about equal cost
    int32_t n_ranges;
    int32_t part_stride;  // partial rankings per block of part_keys / part_cnt
    const uint2 *dense;  // n_heavy x Dc
    int64_t Dc;
    const int32_t *heavy_t;      // query of the call
    const int32_t *heavy_pslot;  // its block of partial rankings
    int32_t n_heavy;
    int64_t q_first;
    const int64_t *exclude;
    int exclude_self;
    const uint8_t *mask_sid;
    int k;
    int32_t *next;  // 8 work counters, one per queue
    unsigned long long *part_keys;
    int32_t *part_cnt;
    unsigned long long *stat;
    // the symmetric pass's front (SymArgs): non-zero scores of the rows from `first` on go to fm[slot of the query][sid - first]; null = off
    float *fm;
    int64_t Ns;
    int32_t first;
    const int32_t *new_of;
};

__global__ void sparse_dense_query_kernel(const int64_t *q_ptr, const int32_t *q_cid, const float *q_val, int64_t q_first,
                                          const int32_t *heavy_t, int64_t Dc, uint2 *dense) {
    const int h = blockIdx.y;
    const int64_t qs = q_ptr[q_first + heavy_t[h]], qe = q_ptr[q_first + heavy_t[h] + 1];
    for (int64_t e = qs + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < qe; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = q_cid[e];
        if (c >= 0) dense[(size_t)h * Dc + c] = make_uint2(1u, __float_as_uint(q_val[e]));
    }
}

template <int KP>
__global__ __launch_bounds__(kBlock) void sparse_rows_kernel(RowsArgs a) {
    constexpr int CAP = 2 * KP;
    constexpr int kDeep = 16;  // entries of a row in flight per lane
    constexpr int64_t kLongRow = 1024;  // rows with more entries are walked by the whole wave, one row at a time
    constexpr int kDenseStep = 16;      // ... and a step of 64 entries with more matches than this adds all 64 lanes (see below)
    __shared__ unsigned long long s_buf[CAP];
    const int lane = threadIdx.x;
    // Eight queues, one per XCD (workgroup b has been observed on XCD b % 8; for speed only): queue x holds the heavy queries
    // x, x + 8, ..., query after query, so that the waves of one XCD look up ONE dense vector (1 MB at the C3 shard) at a time
    // and its L2 keeps it.  With one queue over all queries the launch's working set was all 64 vectors and the kernel ran at
    // the rate of the L2 misses: 42 ms whatever the cut of the rows (profiles/r02_ah_kernel_stats_i2i.txt).  A workgroup whose
    // queue is empty helps the next one.
    for (int turn = 0; turn < 8; turn++) {
        const int x = ((int)blockIdx.x + turn) & 7;
        const int n_mine = (a.n_heavy - x + 7) / 8;  // queries x, x + 8, ...
        const int n_items = n_mine * a.n_ranges;
    for (;;) {
        int w = 0;
        if (lane == 0) w = atomicAdd(a.next + x, 1);
        w = __builtin_amdgcn_readfirstlane(w);
        if (w >= n_items) break;
        const int g = w % a.n_ranges, h = x + 8 * (w / a.n_ranges);
        const int64_t t = a.heavy_t[h];
        const int64_t ex = a.exclude ? a.exclude[t] : (a.exclude_self ? a.q_first + t : (int64_t)-1);
        const uint2 *dense = a.dense + (size_t)h * a.Dc;
        int bcnt = 0;
        unsigned long long thr = 0;
        int my_pos = 0, my_neg = 0, my_hit = 0;
        unsigned long long matched = 0;
        const int64_t s0 = a.range_start[g], s1 = a.range_start[g + 1];
        for (int64_t sb = s0; sb < s1; sb += kBlock) {
            const int64_t sid = sb + lane;
            const bool in = sid < s1;
            const int32_t row_or_none = a.orig_of[in ? sid : s0];  // (-1: a phantom id behind the front, an empty row)
            const int32_t row = row_or_none < 0 ? 0 : row_or_none;
            int64_t e = a.r_ptr[row];
            const int64_t end = in && row_or_none >= 0 ? a.r_ptr[row + 1] : e;
            float acc = 0.0f;
            // Long rows first, one at a time with the whole wave: 64 consecutive entries per step (coalesced), their products
            // added in lane order = entry order.  One lane walking a 116,000-entry row 16 entries at a time was the kernel's
            // critical path (38-47 ms, profiles/r02_u_kernel_stats.txt); rows come longest first, so the long ones are the
            // first lanes of a block.
            for (unsigned long long ml = __ballot(end - e > kLongRow); ml; ml &= ml - 1) {
                const int l = __ffsll((long long)ml) - 1;
                const int64_t e0 = ((int64_t)__builtin_amdgcn_readlane((int)(e >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)e, l);
                const int64_t e1 = ((int64_t)__builtin_amdgcn_readlane((int)(end >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)end, l);
                float sum = 0.0f;  // the same in every lane
                auto fetch = [&](int64_t base, float &prod) {  // this lane's entry of the step: its product, and whether it counts
                    const int64_t at = base + lane;
                    const bool have = at < e1;
                    const int32_t c = a.r_cid[have ? at : 0];
                    const float v = a.r_val[have ? at : 0];
                    const uint2 q = dense[have ? c : 0];
                    prod = __fmul_rn(__uint_as_float(q.y), v);
                    return have && q.x != 0;
                };
                float p_next;
                bool hit_next = fetch(e0, p_next);
                for (int64_t base = e0; base < e1; base += kBlock) {
                    const float p_cur = p_next;
                    const bool hit = hit_next;
                    hit_next = fetch(base + kBlock, p_next);  // in flight while this step is added up
                    unsigned long long mh = __ballot(hit);
                    matched += lane == 0 ? (unsigned long long)__popcll(mh) : 0;
                    if (__popcll(mh) > kDenseStep) {
                        // Many matches in the step: all 64 lanes in lane order with -0 where there is no match -- x + (-0) is x for
                        // every x, so the sum is the matches' sum in entry order, bit for bit -- two instructions per lane (a read of
                        // a fixed lane, an add) instead of the eight of the loop below per match (find the bit, read that lane, add,
                        // clear the bit, branch): a wave alone on its SIMD issues one instruction every four cycles, and this loop
                        // was the kernel's largest share of them (r06_zy2_pmc_SQ_i2i.txt: 40 % of its cycles issuing).
                        const int pz = hit ? __float_as_int(p_cur) : (int)0x80000000;
#pragma unroll
                        for (int b = 0; b < kBlock; b++) sum = __fadd_rn(sum, __int_as_float(__builtin_amdgcn_readlane(pz, b)));
                    } else
                        for (; mh; mh &= mh - 1) {
                            const int b = __ffsll((long long)mh) - 1;
                            sum = __fadd_rn(sum, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_cur), b)));
                        }
                }
                if (lane == l) {
                    acc = sum;
                    e = end;  // done: the per-lane walk below skips it
                }
            }
            // three blocks of kDeep entries in flight per lane: the entries of block b + 2 are loaded while the lookups of block
            // b + 1 are under way and block b is added up
            int32_t c1[kDeep], c2[kDeep];
            float v0[kDeep], v1[kDeep], v2[kDeep];
            uint2 q0[kDeep], q1[kDeep];
            auto load_entries = [&](int64_t from, int32_t (&c)[kDeep], float (&v)[kDeep]) {
#pragma unroll
                for (int j = 0; j < kDeep; j++) {
                    const int64_t at = from + j < end ? from + j : 0;  // element 0 exists (the arrays are padded)
                    c[j] = a.r_cid[at], v[j] = a.r_val[at];
                }
            };
            auto look_up = [&](int64_t from, const int32_t (&c)[kDeep], uint2 (&q)[kDeep]) {
#pragma unroll
                for (int j = 0; j < kDeep; j++) q[j] = dense[from + j < end ? c[j] : 0];
            };
            load_entries(e, c1, v0);
            look_up(e, c1, q0);
            load_entries(e + kDeep, c1, v1);
            while (__ballot(e < end)) {
                look_up(e + kDeep, c1, q1);
                load_entries(e + 2 * kDeep, c2, v2);
#pragma unroll
                for (int j = 0; j < kDeep; j++) {
                    const bool hit = e + j < end && q0[j].x != 0;
                    const float sum = __fadd_rn(acc, __fmul_rn(__uint_as_float(q0[j].y), v0[j]));
                    acc = hit ? sum : acc;
                    matched += hit;
                }
#pragma unroll
                for (int j = 0; j < kDeep; j++) q0[j] = q1[j], v0[j] = v1[j], v1[j] = v2[j], c1[j] = c2[j];
                e += kDeep;
            }
            // rank (the same rules as the read-back of sparse_tile_kernel)
            bool have = in && (__float_as_uint(acc) << 1) != 0;
            if (a.fm && have && sid >= a.first) a.fm[(size_t)a.new_of[a.q_first + t] * a.Ns + (sid - a.first)] = acc;
            if (__ballot(have)) {
                my_hit += have;
                have = have && (int64_t)row != ex && (!a.mask_sid || a.mask_sid[have ? sid : s0]);
                const uint32_t ord = score_ord(acc);
                my_pos += have && ord > kZeroOrd;
                my_neg += have && ord < kZeroOrd;
                const bool cand = have && ord >= (uint32_t)(thr >> 32);
                if (__ballot(cand)) {
                    const unsigned long long key = cand ? make_key(ord, row) : 0;
                    push<KP>(s_buf, a.k, bcnt, thr, key, cand && key > thr, lane);
                }
            }
        }
        const long long pos = wave_sum((long long)my_pos), neg = wave_sum((long long)my_neg), hit = wave_sum((long long)my_hit);
        const long long walked = wave_sum((long long)matched);
        finish<KP>(s_buf, bcnt, lane);
        const size_t part = (size_t)a.heavy_pslot[h] * a.part_stride + g;
        for (int i = lane; i < KP; i += kBlock) a.part_keys[part * KP + i] = bcnt > 0 ? s_buf[i] : 0;
        if (lane == 0) {
            a.part_cnt[part * 2] = (int32_t)pos;
            a.part_cnt[part * 2 + 1] = (int32_t)neg;
            if (a.stat) {
                atomicAdd(&a.stat[0], (unsigned long long)walked);
                atomicAdd(&a.stat[1], (unsigned long long)hit);
            }
        }
        __syncthreads();  // s_buf is reused by the next work item
    }
    }
}

// the partial rankings of a long query (one per group) -> its result row
struct MergeArgs {
    const int32_t *split_t;  // query of every block of partial rankings
    const int32_t *split_n;  // rankings in the block
    int32_t n_split, part_stride;
    const unsigned long long *part_keys;
    const int32_t *part_cnt;
    int64_t q_first, N;
    const int64_t *exclude;
    int exclude_self;
    const uint8_t *mask_sid;
    const int32_t *new_of;
    int64_t n_admissible;
    int k;
    int32_t *out_idx;
    float *out_score;
    int32_t *out_cnt;
};

template <int KP>
__global__ __launch_bounds__(kBlock) void sparse_merge_kernel(MergeArgs a) {
    constexpr int CAP = 2 * KP;
    __shared__ unsigned long long s_buf[CAP];
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < a.n_split; b += gridDim.x) {
        const int64_t t = a.split_t[b];
        int bcnt = 0;
        unsigned long long thr = 0;
        long long pos = 0, neg = 0;
        const int nparts = a.split_n[b];
        for (int s = 0; s < nparts; s++) {
            const size_t part = (size_t)b * a.part_stride + s;
            pos += a.part_cnt[part * 2];
            neg += a.part_cnt[part * 2 + 1];
            for (int i = 0; i < KP; i += kBlock) {
                const unsigned long long key = a.part_keys[part * KP + i + lane];
                push<KP>(s_buf, a.k, bcnt, thr, key, key > thr, lane);
            }
        }
        finish<KP>(s_buf, bcnt, lane);
        const int64_t ex = a.exclude ? a.exclude[t] : (a.exclude_self ? a.q_first + t : (int64_t)-1);
        const bool ex_in = ex >= 0 && ex < a.N;
        const bool ex_counts = ex_in && (!a.mask_sid || a.mask_sid[a.new_of[ex_in ? ex : 0]]);
        const int cnt = written(pos, neg, a.n_admissible - (ex_counts ? 1 : 0), a.k);
        write_result(s_buf, cnt, a.k, t, a.out_idx, a.out_score, a.out_cnt, lane);
        __syncthreads();
    }
}

// SYM: own keys + foreign list -> the result row of every whole-query row (see SymArgs)
struct SymMergeArgs {
    SymArgs sym;
    int64_t N;
    const int32_t *new_of, *orig_of;
    int k;
    int32_t *out_idx;
    float *out_score;
    int32_t *out_cnt;
    int32_t *redo;                // [0] = rows on the list, [1 ..] = the rows (queries) whose foreign list overflowed
    unsigned long long *stat;     // [2] += foreign entries ranked, [3] = max over the rows (atomicMax)
};

// fm (front x Ns) -> fmt (Ns x fw): 64 x 64 tiles through LDS, slots past the front read as 0
__global__ __launch_bounds__(256) void sparse_front_transpose_kernel(SymArgs y) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int s0 = (int)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) tile[j][tx] = s0 + j < y.front && r0 + tx < y.Ns ? y.fm[(size_t)(s0 + j) * y.Ns + r0 + tx] : 0.0f;
    __syncthreads();
    for (int j = ty; j < 64; j += 4)
        if (r0 + j < y.Ns) y.fmt[(size_t)(r0 + j) * y.fw + s0 + tx] = tile[tx][j];
}

template <int KP>
__global__ __launch_bounds__(kBlock) void sparse_sym_merge_kernel(SymMergeArgs a) {
    constexpr int CAP = 2 * KP;
    __shared__ unsigned long long s_buf[CAP];
    const int lane = threadIdx.x;
    for (int64_t t = blockIdx.x; t < a.N; t += gridDim.x) {
        const int32_t sid = a.new_of[t];
        if (sid < a.sym.first) continue;  // a long / heavy row: its own kernels wrote it
        uint32_t cap;
        const int64_t at = sym_list_at(a.sym, sid, cap);
        const uint32_t fc = a.sym.fcnt[sid];
        if (lane == 0 && fc) {
            atomicAdd(&a.stat[2], (unsigned long long)fc);
            atomicMax(&a.stat[3], (unsigned long long)fc);
        }
        if (fc > cap) {
            if (lane == 0) a.redo[1 + atomicAdd(&a.redo[0], 1)] = (int32_t)t;
            continue;
        }
        const unsigned long long tp = a.sym.tp[sid];
        long long pos = (long long)(tp >> 32), neg = (long long)a.sym.neg[sid];
        const float *frow = a.sym.front ? a.sym.fmt + (size_t)(sid - a.sym.first) * a.sym.fw : nullptr;
        const unsigned long long *own = a.sym.own + (size_t)t * KP;
        int bcnt = 0;
        unsigned long long thr = 0;
        for (int i = 0; i < KP; i += kBlock) {
            const unsigned long long key = own[i + lane];
            push<KP>(s_buf, a.k, bcnt, thr, key, key > thr, lane);
        }
        for (uint32_t i = 0; i < fc; i += kBlock) {
            const unsigned long long key = i + lane < fc ? a.sym.flist[at + i + lane] : 0;
            push<KP>(s_buf, a.k, bcnt, thr, key, key > thr, lane);
        }
        if (frow) {  // the front's scores: all of them count, and what the row's published bound lets through is ranked (the bound
                     // holds for every candidate of the row)
            const uint32_t bound = (uint32_t)tp;
            int fp = 0, fn = 0;
            for (int i = 0; i < a.sym.fw; i += kBlock) {
                const float x = frow[i + lane];
                fp += x > 0.0f, fn += x < 0.0f;
                const uint32_t ord = score_ord(x);
                const bool cand = (__float_as_uint(x) << 1) != 0 && ord >= bound && ord >= (uint32_t)(thr >> 32);
                if (!__ballot(cand)) continue;
                const unsigned long long key = cand ? make_key(ord, a.orig_of[i + lane]) : 0;
                push<KP>(s_buf, a.k, bcnt, thr, key, cand && key > thr, lane);
            }
            pos += wave_sum((long long)fp), neg += wave_sum((long long)fn);
        }
        if (bcnt > KP) {  // (at least k keys: cut to them first)
            __syncthreads();
            for (int i = bcnt + lane; i < CAP; i += kBlock) s_buf[i] = 0;
            __syncthreads();
            cut_to_k<KP>(s_buf, a.k, bcnt, thr, lane);
        }
        const int cnt = written(pos, neg, a.N - 1, a.k);
        write_ranked<KP>(s_buf, bcnt, cnt, a.k, t, a.out_idx, a.out_score, a.out_cnt, lane);
        __syncthreads();
    }
}

// ---- index construction on the device -------------------------------------------------------------------------------------
// dims = the sorted distinct indices of the stored rows (built by the host while it validates the CSR); the directory entry
// of an index is its position there.
__device__ inline int32_t find_dim(const uint32_t *dims, int64_t Dc, uint32_t idx) {
    int64_t lo = 0, hi = Dc;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (dims[mid] < idx)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < Dc && dims[lo] == idx ? (int32_t)lo : -1;
}

__global__ void sparse_translate_kernel(const uint32_t *idx, int64_t n, const uint32_t *dims, int64_t Dc, int32_t *cid) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        cid[e] = find_dim(dims, Dc, idx[e]);
}

struct BuildArgs {
    const int64_t *r_ptr;  // N + 1
    const int32_t *r_cid;  // directory entry of every stored entry
    const float *r_val;
    int64_t N;
    const int32_t *new_of;  // scratch id of every row
    int32_t stride;         // directory entries per index = groups
    int32_t shift;          // log2 G (group = sid >> shift, loc = sid mod G)
    uint32_t *cnt;  // Dc * stride (+ 1): entries per (index, group); the cursor of the scatter pass afterwards
    Posting *post;
};

// one wave per stored row; SCATTER = false counts the entries of every (index, group), true places them
template <bool SCATTER>
__global__ __launch_bounds__(256) void sparse_build_kernel(BuildArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < a.N; r += nwaves) {
        const int64_t sid = a.new_of[r];
        const int32_t bucket = (int32_t)(sid >> a.shift);
        const int32_t loc = (int32_t)sid;  // the posting carries the scratch id: a super-visit spans several groups
        for (int64_t e = a.r_ptr[r] + lane; e < a.r_ptr[r + 1]; e += 64) {
            uint32_t *c = a.cnt + (size_t)a.r_cid[e] * a.stride + bucket;
            if (SCATTER)
                a.post[atomicAdd(c, 1u)] = Posting{loc, a.r_val[e]};
            else
                atomicAdd(c, 1u);
        }
    }
}

// exclusive scan of n uint32 in place, three launches: chunk sums, one workgroup over the sums, chunks again
constexpr int kScanBlock = 1024;
constexpr int kScanPer = 8;  // elements per thread
__device__ inline uint32_t block_exclusive(uint32_t v, uint32_t *s_part, int tid, uint32_t *total) {
    s_part[tid] = v;
    __syncthreads();
    for (int o = 1; o < kScanBlock; o <<= 1) {
        const uint32_t x = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += x;
        __syncthreads();
    }
    const uint32_t incl = s_part[tid];
    if (total) *total = s_part[kScanBlock - 1];
    __syncthreads();
    return incl - v;
}
__global__ __launch_bounds__(kScanBlock) void sparse_scan_sums_kernel(const uint32_t *x, int64_t n, uint32_t *sums) {
    __shared__ uint32_t s_part[kScanBlock];
    const int tid = threadIdx.x;
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + tid) * kScanPer;
    uint32_t v = 0;
    for (int j = 0; j < kScanPer; j++)
        if (base + j < n) v += x[base + j];
    uint32_t total;
    block_exclusive(v, s_part, tid, &total);
    if (tid == 0) sums[blockIdx.x] = total;
}
// one workgroup: exclusive scan of the nb chunk sums in place
__global__ __launch_bounds__(kScanBlock) void sparse_scan_top_kernel(uint32_t *sums, int64_t nb) {
    __shared__ uint32_t s_part[kScanBlock];
    const int tid = threadIdx.x;
    const int64_t per = (nb + kScanBlock - 1) / kScanBlock;
    const int64_t lo = (int64_t)tid * per, hi = lo + per < nb ? lo + per : nb;
    uint32_t v = 0;
    for (int64_t i = lo; i < hi; i++) v += sums[i];
    uint32_t run = block_exclusive(v, s_part, tid, nullptr);
    for (int64_t i = lo; i < hi; i++) {
        const uint32_t x = sums[i];
        sums[i] = run;
        run += x;
    }
}
__global__ __launch_bounds__(kScanBlock) void sparse_scan_apply_kernel(uint32_t *x, int64_t n, const uint32_t *sums) {
    __shared__ uint32_t s_part[kScanBlock];
    const int tid = threadIdx.x;
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + tid) * kScanPer;
    uint32_t loc[kScanPer];
    uint32_t v = 0;
    for (int j = 0; j < kScanPer; j++) {
        loc[j] = base + j < n ? x[base + j] : 0;
        v += loc[j];
    }
    uint32_t run = sums[blockIdx.x] + block_exclusive(v, s_part, tid, nullptr);
    for (int j = 0; j < kScanPer; j++)
        if (base + j < n) {
            x[base + j] = run;
            run += loc[j];
        }
}

}  // namespace sparse
}  // namespace gorse
