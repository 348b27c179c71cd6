// sparse_emu.cpp -- TEST INFRASTRUCTURE ONLY: gorse_amd/csrc/sparse_kernels.hpp compiled for the CPU through
// tests/emu/hip_emu.hpp, behind one C function that tests/test_sparse_kernel_emu_cpu.py compares with the oracle.
// The index construction in front of the kernel is the product's own host code (sparse_host.hpp).
#include "hip_emu.hpp"

#include "../../gorse_amd/csrc/sparse_host.hpp"
#include "../../gorse_amd/csrc/sparse_kernels.hpp"

#include <algorithm>

using namespace gorse::sparse;

static int g_device_build = 0;
static int g_hot = 0;  // 512: the query kernel keeps the cells of the 512 longest rows in (emulated) LDS
extern "C" __attribute__((visibility("default"))) void emu_sparse_set_hot(int rows) { g_hot = rows; }
static int64_t g_heavy_dims = 0;  // > 0: queries with more entries go through the row-streaming kernels, as in sparse.hip
extern "C" __attribute__((visibility("default"))) void emu_sparse_set_heavy(int64_t dims) { g_heavy_dims = dims; }
// 1: emu_sparse_search also runs the postings build kernels, checks them against the host build and lets the query kernel
// walk the lists they produced
extern "C" __attribute__((visibility("default"))) void emu_sparse_set_device_build(int on) { g_device_build = on; }

// queries: q_ptr == NULL -> the stored rows q_first .. q_first + nq (all pairs), else the given CSR rows 0 .. nq.
// `rounds` launches are made over the same scratch (serial bases advance like in the library); the outputs hold the
// results of the last one.  Returns 0, or -1 for an invalid input (message not kept: the tests feed valid data).
// Built with -fvisibility=hidden: the kernel template and the index builder have the same mangled names as the host stubs
// inside libgorse_hip.so, which the test process may have loaded with RTLD_GLOBAL; only this entry point is exported.
extern "C" __attribute__((visibility("default"))) int emu_sparse_search(int64_t N, const int64_t *indptr, const uint32_t *indices, const float *values, int64_t nq,
                                 const int64_t *q_ptr, const uint32_t *q_idx, const float *q_val, int64_t q_first,
                                 const int64_t *exclude, int exclude_self, const uint8_t *mask, int k, int grid, int block,
                                 int rounds, uint32_t serial_base, int32_t *out_idx, float *out_score, int32_t *out_cnt,
                                 unsigned long long *stat2) {
    if (!validate_csr(N, indptr, indices).empty()) return -1;
    if (q_ptr && !validate_csr(nq, q_ptr, q_idx).empty()) return -1;
    const RowOrder order = order_rows(N, indptr);
    Postings post;
    if (!build_postings(N, indptr, indices, values, post, order.new_of.data()).empty()) return -1;
    if (g_device_build) {  // the same postings from the three build kernels (entries of a list in any order)
        const int64_t nnz = indptr[N] - indptr[0], D = post.D;
        std::vector<int64_t> rp((size_t)N + 1);
        for (int64_t r = 0; r <= N; r++) rp[(size_t)r] = indptr[r] - indptr[0];
        std::vector<unsigned long long> pp((size_t)D + 1, 0), cursor((size_t)(D > 0 ? D : 1), 0);
        std::vector<int32_t> prow((size_t)(nnz > 0 ? nnz : 1), -1);
        std::vector<float> pval((size_t)(nnz > 0 ? nnz : 1), 0.0f);
        BuildArgs b;
        b.r_ptr = rp.data(), b.r_idx = indices + indptr[0], b.r_val = values + indptr[0];
        b.N = N, b.nnz = nnz, b.D = D;
        b.p_ptr = pp.data(), b.cursor = cursor.data(), b.p_row = prow.data(), b.p_val = pval.data();
        b.new_of = order.new_of.data();
        emu::launch(3, (unsigned)block, [&] { sparse_count_kernel(b); });
        emu::launch(1, (unsigned)(block < kScanBlock ? block : kScanBlock), [&] { sparse_scan_kernel(b); });
        emu::launch(2, (unsigned)block, [&] { sparse_scatter_kernel(b); });
        // same directory, every list the same SET of (row, value) as the host build
        for (int64_t t = 0; t <= D; t++)
            if ((int64_t)pp[(size_t)t] != post.ptr[(size_t)t]) return -2;
        for (int64_t t = 0; t < D; t++) {
            if ((int64_t)cursor[(size_t)t] != post.ptr[(size_t)t + 1]) return -3;
            std::vector<std::pair<int32_t, float>> x, y;
            for (int64_t e = post.ptr[(size_t)t]; e < post.ptr[(size_t)t + 1]; e++) {
                x.emplace_back(post.row[(size_t)e], post.val[(size_t)e]);
                y.emplace_back(prow[(size_t)e], pval[(size_t)e]);
            }
            std::sort(x.begin(), x.end());
            std::sort(y.begin(), y.end());
            if (x != y) return -4;
        }
        post.row.assign(prow.begin(), prow.begin() + nnz);  // the query kernel below walks the device-built lists
        post.val.assign(pval.begin(), pval.begin() + nnz);
    }
    const int kp = pick_kp(k);
    if (!kp || grid < 1 || block < 1) return -1;
    std::vector<Cell> cell((size_t)grid * N, 0);
    std::vector<int32_t> touched((size_t)grid * N);
    QueryArgs a;
    a.p_ptr = post.ptr.data(), a.p_row = post.row.data(), a.p_val = post.val.data(), a.D = post.D;
    a.q_ptr = q_ptr ? q_ptr : indptr, a.q_idx = q_ptr ? q_idx : indices, a.q_val = q_ptr ? q_val : values;
    a.q_first = q_ptr ? 0 : q_first, a.nq = nq;
    a.exclude = exclude, a.exclude_self = exclude_self, a.mask = mask, a.N = N;
    a.heavy_dims = g_heavy_dims > 0 ? g_heavy_dims : INT64_MAX;
    a.n_admissible = N;
    if (mask) {
        a.n_admissible = 0;
        for (int64_t r = 0; r < N; r++) a.n_admissible += mask[r] != 0;
    }
    a.cell = cell.data(), a.touched = touched.data();
    a.orig_of = order.orig_of.data();
    a.k = k, a.out_idx = out_idx, a.out_score = out_score, a.out_cnt = out_cnt, a.stat = stat2;
    const uint32_t per_launch = (uint32_t)((nq + grid - 1) / grid);
    for (int r = 0; r < rounds; r++) {
        if (stat2) stat2[0] = stat2[1] = 0;
        a.serial_base = serial_base + (uint32_t)r * per_launch;
        auto body = [&] {
            if (g_hot == 512) {
                switch (kp) {
                    case 256: sparse_query_kernel<256, 512>(a); break;
                    case 512: sparse_query_kernel<512, 512>(a); break;
                    default: sparse_query_kernel<1024, 512>(a); break;
                }
                return;
            }
            switch (kp) {
                case 256: sparse_query_kernel<256, 0>(a); break;
                case 512: sparse_query_kernel<512, 0>(a); break;
                default: sparse_query_kernel<1024, 0>(a); break;
            }
        };
        emu::launch((unsigned)grid, (unsigned)block, body);
        // the queries the kernel skipped, kHeavyBatch per pass over the stored rows (sparse.hip's run_queries)
        std::vector<int64_t> rp((size_t)N + 1);
        for (int64_t r2 = 0; r2 <= N; r2++) rp[(size_t)r2] = indptr[r2] - indptr[0];
        std::vector<int64_t> heavy;
        for (int64_t t = 0; t < nq; t++)
            if (a.q_ptr[a.q_first + t + 1] - a.q_ptr[a.q_first + t] > a.heavy_dims) heavy.push_back(t);
        std::vector<float> hscore((size_t)kHeavyBatch * N);
        std::vector<uint8_t> hcommon((size_t)kHeavyBatch * N);
        HeavyArgs ha;
        ha.r_ptr = rp.data(), ha.r_idx = indices + indptr[0], ha.r_val = values + indptr[0], ha.N = N;
        ha.q_ptr = a.q_ptr, ha.q_idx = a.q_idx, ha.q_val = a.q_val, ha.q_first = a.q_first;
        ha.score = hscore.data(), ha.common = hcommon.data();
        ha.exclude = exclude, ha.exclude_self = exclude_self, ha.mask = mask, ha.n_admissible = a.n_admissible;
        ha.k = k, ha.out_idx = out_idx, ha.out_score = out_score, ha.out_cnt = out_cnt, ha.stat = stat2;
        for (size_t at = 0; at < heavy.size(); at += kHeavyBatch) {
            ha.nb = (int)(heavy.size() - at < (size_t)kHeavyBatch ? heavy.size() - at : (size_t)kHeavyBatch);
            for (int b = 0; b < kHeavyBatch; b++) ha.hq[b] = b < ha.nb ? heavy[at + b] : 0;
            emu::launch(2, (unsigned)block, [&] { sparse_heavy_score_kernel(ha); });
            emu::launch((unsigned)ha.nb, (unsigned)block, [&] { sparse_heavy_rank_kernel(ha); });
        }
    }
    return 0;
}
