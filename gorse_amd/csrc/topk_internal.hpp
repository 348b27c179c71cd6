// topk_internal.hpp -- the gorse_topk handle shared by topk.hip (path A: exact VALU scan + literal
// heaps) and topk_mfma.hip (path B: bf16 MFMA candidate sweep + exact rescoring).
#pragma once
#include "cf_device.hpp"

#define GORSE_PROF_TOPK_NCLASSES 6

namespace gorse {
struct TopkChunkState;  // topk_mfma.hip: what the stages of one chunk of an MFMA search share
}

struct gorse_topk {
    int device = 0;
    int64_t N = 0;
    int d = 0, dtype = 0, metric = 0;
    hipStream_t stream = nullptr;
    gorse::DevBuf<float> X;       // N x d fp32 (bf16 inputs are expanded by <<16, bfloats.go:32-38)
    gorse::DevBuf<uint16_t> Xb;   // N x d bf16 as given
    gorse::DevBuf<float> norm2;   // floats.Dot(x, x) per stored vector
    gorse::DevBuf<float> qbuf, qnorm, dist;
    gorse::DevBuf<int64_t> qidx;
    gorse::DevBuf<int32_t> out_idx, out_cnt, heap_v, scan_literal;
    gorse::DevBuf<float> out_dist, heap_w;
    // ---- path B (topk_mfma.hip) ----
    bool mfma_ok = false;          // operands built and the data admits a rigorous error bound
    int kp = 0;                    // k-steps of 16 of the MFMA operands (KPAD = 16 * kp)
    float err_coef = 0.0f;         // |approx - reference| <= err_coef * |q| * |x| (dot of the operands)
    float max_norm = 0.0f;         // max_i sqrt(norm2[i])
    bool coarse_ok = false;        // cosine with nearly equal norms: block-level scale bound in the sweep
    float rs_min = 0.0f, rs_max = 0.0f;  // cosine: bounds of the row scales 1 / sqrt(norm2[i]) (the DMA sweeps' block bound)
    gorse::DevBuf<uint16_t> opA_own, opB_own;  // bf16 operand matrices N x KPAD (candidate / query roles)
    const uint16_t *opA = nullptr, *opB = nullptr;  // may alias Xb
    gorse::DevBuf<float> rscale;   // the sweep's per-row value: cosine 1 / sqrt(norm2[i]), Euclidean -norm2[i] / 2, -dot 1; N + kTopkRowPad
                                   // entries, the padding NaN (what the sweep's tiles read for the rows past N)
    gorse::DevBuf<uint16_t> opQ;   // query operands of the current chunk when they are not rows of opB
    gorse::DevBuf<float> qn2, qmargin, qf32;
    gorse::DevBuf<int64_t> qid;
    gorse::DevBuf<uint2> cbuf;     // per query: kCap (approx key, candidate index) entries
    gorse::DevBuf<int32_t> ccnt;
    gorse::DevBuf<uint8_t> cflag;
    gorse::DevBuf<int32_t> res_idx, res_cnt;
    gorse::DevBuf<float> res_dist;
    // tie replay (history sweep of the flagged queries + topk_tie_sort_kernel + topk_tie_replay_kernel)
    gorse::DevBuf<int32_t> rp_pos, rp_ccnt, rp_hcnt, fl_pos;  // fl_*: the chunk's flagged queries, listed on the device (flag_compact_kernel)
    gorse::DevBuf<int64_t> fl_self;
    gorse::DevBuf<long long> fl_cnt;
    gorse::DevBuf<int64_t> rp_self;
    gorse::DevBuf<uint16_t> rp_op;
    gorse::DevBuf<float> rp_margin, rp_fslice;  // rp_fslice: final threshold of every (row slice, query) of a history sweep
    gorse::DevBuf<float> rp_fwarm;              // starting threshold of every (row slice, query) of a history sweep (tie_warm_kernel)
    gorse::DevBuf<uint2> rp_cbuf, rp_hbuf;
    gorse::DevBuf<uint8_t> rp_flag;
    gorse::DevBuf<int32_t> rp_sidx, rp_scount;
    gorse::DevBuf<float> rp_sdst;
    gorse::DevBuf<unsigned long long> sweep_prof;  // probe: phase counters of the instrumented sweep
    gorse::KernelProfile prof{GORSE_PROF_TOPK_NCLASSES};
    // gorse_topk_set_mask: rows with mask == 0 take no part in any search (as if they had never been added)
    gorse::DevBuf<uint8_t> mask;
    gorse::DevBuf<float> rscale_m;  // the sweep's per-row value (cosine scale / Euclidean bias / 1 for -dot) with NaN for masked rows
    bool has_mask = false;
    int64_t n_admissible = 0;
    bool bf16_order = false;  // GORSE_METRIC_EUCLIDEAN_BF16: metric is kept as GORSE_METRIC_EUCLIDEAN, the distance kernels get id 3
    int kernel_metric() const { return bf16_order ? 3 : metric; }
    int64_t n_fallback = 0, n_tie = 0, n_resweep = 0;  // of the last search: path A rows, tie replays, warm starts swept again
    gorse::DevBuf<float> f0, f1;                       // warm-start thresholds of a chunk (topk_mfma_search): pilot, pre-pilot
    // the symmetric all-pairs sweep (topk_mfma.hip, SYM): per query a foreign candidate list and its counter, the raw-score form
    // of the thresholds; last_sym: did the last search's (last chunk's) main sweep take the symmetric form
    gorse::DevBuf<uint2> fbuf;
    gorse::DevBuf<int32_t> fcnt;
    gorse::DevBuf<float> f0raw;
    gorse::DevBuf<unsigned long long> sym_stats;  // counters of the last symmetric search (gorse_hip_test_topk_sym_stats)
    bool last_sym = false;
    // the state of the search in progress (topk_mfma.hip): its stages run back to back in topk_mfma_search, call by call in the
    // triangle-sharded search (gorse_topk_tri_*), whose message buffers follow
    gorse::TopkChunkState *cstate = nullptr;
    gorse::TopkChunkState *chunk_state();
    gorse::DevBuf<int32_t> tri_counts, tri_in_counts;
    gorse::DevBuf<long long> tri_offsets;
    gorse::DevBuf<uint2> tri_entries, tri_in_entries;
    int64_t tri_packed_counts = 0, tri_packed_entries = 0;
    std::vector<uint8_t> dbg_flags;   // probe (variant bit 24): the pilot's flags and list lengths of the last chunk
    std::vector<int32_t> dbg_counts;
    int32_t use() const {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return gorse::fail(GORSE_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return GORSE_OK;
    }
};

namespace gorse {
constexpr int64_t kTopkRowPad = 256;  // NaN entries behind the last row value: one tile of the sweep reaches at most 127 rows past N
// floats.Euclidean (squared part) in AVX512 order for rows staged in LDS: floats_avx512.c:374-441
__device__ __forceinline__ float euclid512_lds(const float *a, const float *b, const VecShape &vs, int lane) {
    float acc = 0.0f;
    for (int c = 0; c < vs.nfull; c++) {
        float v = a[16 * c + lane] - b[16 * c + lane];
        v = v * v;
        acc = c == 0 ? v : v + acc;
    }
    float sum = group_tree16(acc);
    if (vs.has8) {
        int e = vs.nfull * 16 + (lane & 7);
        float v = a[e] - b[e];
        sum += group_tree8(v * v);
    }
    for (int e = vs.tail0; e < vs.d; e++) {
        float v = a[e] - b[e];
        sum = fmaf(v, v, sum);
    }
    return sqrtf(sum);
}

// bfloats.Euclidean (squared part) in the order of the reference's AVX512BW kernel (common/bfloats/src/bfloats_avx512.c:26-59 as
// shipped in bfloats_avx512.s, restated by the oracle's orc_bf16_euclidean): rows are the bf16 values expanded by << 16;
// sixteen partial sums with UNFUSED multiply + add, the partials added SEQUENTIALLY (a vaddss chain, not the shuffle tree of
// floats.Euclidean), the n % 16 tail one element at a time with a fused multiply-add.
__device__ __forceinline__ float euclid_bf16_lds(const float *a, const float *b, int d, int lane) {
    const int nfull = d / 16;
    float acc = 0.0f;
    for (int c = 0; c < nfull; c++) {
        const float v = a[16 * c + lane] - b[16 * c + lane];
        acc = acc + v * v;  // -ffp-contract=off: two roundings, as vmulps + vaddps
    }
    float sum = 0.0f;
    const int base = threadIdx.x & ~15;  // first lane of this 16-lane group inside its wave
#pragma unroll
    for (int l = 0; l < 16; l++) sum = sum + __shfl(acc, (base & 63) + l, 64);
    for (int e = 16 * nfull; e < d; e++) {
        const float v = a[e] - b[e];
        sum = fmaf(v, v, sum);
    }
    return sqrtf(sum);
}
// the distance of `metric` (GORSE_METRIC_EUCLIDEAN or the kernels' own id 3 = bfloats order) for rows staged in LDS
constexpr int kMetricEuclidBf16 = 3;
__device__ __forceinline__ float euclid_any_lds(int metric, const float *a, const float *b, const VecShape &vs, int lane) {
    return metric == kMetricEuclidBf16 ? euclid_bf16_lds(a, b, vs.d, lane) : euclid512_lds(a, b, vs, lane);
}

// topk.hip
int32_t topk_compute_norms(gorse_topk *h, const float *V, int64_t n, float *out);
// path A on queries whose fp32 rows sit in h->qbuf (nq x d) [+ h->qnorm]; qidx_dev = exclusion ids or null (qidx_host: the same
// ids on the host).  Results land in h->out_idx / out_dist / out_cnt and, where given, in the host arrays.  Queries whose answer
// depends on the heap's history (ties among the k + 1 smallest distances, NaN) go to the MFMA path's tie replay where that is
// usable and `reroute` allows it, else to the literal heap kernel.
int32_t topk_scan_block(gorse_topk *h, int64_t nq, const int64_t *qidx_dev, const int64_t *qidx_host, int k, int prune0,
                        int32_t *idx_out, float *dist_out, int32_t *cnt_out, bool reroute = true);
int64_t topk_scan_block_queries(const gorse_topk *h);
// topk_mfma.hip
int32_t topk_mfma_prepare(gorse_topk *h);  // at create: operands, scales, error bound
// Queries: stored vectors qid_host[0..nq) (exclude_self) or, when qid_host is NULL and qv_dev is given, nq fp32
// query vectors already on the device.  Host outputs may be NULL.
int32_t topk_mfma_search(gorse_topk *h, const int64_t *qid_host, int64_t q_contig_begin, const float *qv_dev,
                         int64_t nq, int k, int prune0, int32_t *idx_out, float *dist_out, int32_t *cnt_out);
bool topk_mfma_usable(const gorse_topk *h, int64_t nq, int k);
void topk_mfma_release(gorse_topk *h);  // at destroy: the search state
extern int g_topk_variant;     // probe switches of the sweep (tile rows, coarse scale test)
extern int g_topk_force_path;  // 0 auto, 1 path A only, 2 path B whenever it is usable, 3 = 2 without the tie replay
}  // namespace gorse
