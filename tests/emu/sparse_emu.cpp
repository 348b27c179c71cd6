// sparse_emu.cpp -- TEST INFRASTRUCTURE ONLY: gorse_amd/csrc/sparse_kernels.hpp compiled for the CPU through
// tests/emu/hip_emu.hpp, behind one C function that tests/test_sparse_kernel_emu_cpu.py compares with the oracle.
// The index construction in front of the kernel is the product's own host code (sparse_host.hpp).
#include "hip_emu.hpp"

#include "../../gorse_amd/csrc/sparse_host.hpp"
#include "../../gorse_amd/csrc/sparse_kernels.hpp"

using namespace gorse::sparse;

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
    Postings post;
    if (!build_postings(N, indptr, indices, values, post).empty()) return -1;
    const int kp = pick_kp(k);
    if (!kp || grid < 1 || block < 1) return -1;
    std::vector<Cell> cell((size_t)grid * N, 0);
    std::vector<int32_t> touched((size_t)grid * N);
    QueryArgs a;
    a.p_ptr = post.ptr.data(), a.p_row = post.row.data(), a.p_val = post.val.data(), a.D = post.D;
    a.q_ptr = q_ptr ? q_ptr : indptr, a.q_idx = q_ptr ? q_idx : indices, a.q_val = q_ptr ? q_val : values;
    a.q_first = q_ptr ? 0 : q_first, a.nq = nq;
    a.exclude = exclude, a.exclude_self = exclude_self, a.mask = mask, a.N = N;
    a.n_admissible = N;
    if (mask) {
        a.n_admissible = 0;
        for (int64_t r = 0; r < N; r++) a.n_admissible += mask[r] != 0;
    }
    a.cell = cell.data(), a.touched = touched.data();
    a.k = k, a.out_idx = out_idx, a.out_score = out_score, a.out_cnt = out_cnt, a.stat = stat2;
    const uint32_t per_launch = (uint32_t)((nq + grid - 1) / grid);
    for (int r = 0; r < rounds; r++) {
        if (stat2) stat2[0] = stat2[1] = 0;
        a.serial_base = serial_base + (uint32_t)r * per_launch;
        auto body = [&] {
            switch (kp) {
                case 64: sparse_query_kernel<64>(a); break;
                case 128: sparse_query_kernel<128>(a); break;
                case 256: sparse_query_kernel<256>(a); break;
                case 512: sparse_query_kernel<512>(a); break;
                default: sparse_query_kernel<1024>(a); break;
            }
        };
        emu::launch((unsigned)grid, (unsigned)block, body);
    }
    return 0;
}
