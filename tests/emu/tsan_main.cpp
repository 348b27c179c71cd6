// tsan_main.cpp -- TEST INFRASTRUCTURE ONLY: the emulated sparse kernel under ThreadSanitizer.  Every plain load/store of
// the kernel that two work-items could reach without a barrier (or an atomic) in between is reported as a data race, i.e.
// this checks the barrier placement of sparse_kernels.hpp.  Built and run by tests/test_sparse_kernel_emu_cpu.py:
//   g++ -O1 -g -std=c++17 -fsanitize=thread -ffp-contract=off -pthread tsan_main.cpp sparse_emu.cpp -o tsan_sparse
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

extern "C" int emu_sparse_search(int64_t N, const int64_t *indptr, const uint32_t *indices, const float *values, int64_t nq,
                                 const int64_t *q_ptr, const uint32_t *q_idx, const float *q_val, int64_t q_first,
                                 const int64_t *exclude, int exclude_self, const uint8_t *mask, int k, int grid, int block,
                                 int rounds, uint32_t serial_base, int32_t *out_idx, float *out_score, int32_t *out_cnt,
                                 unsigned long long *stat2);
extern "C" void emu_sparse_set_device_build(int on);
extern "C" void emu_sparse_set_heavy(int64_t dims);

int main() {
    std::mt19937 rng(5);
    const int64_t N = 700, D = 24;
    std::vector<int64_t> ptr{0};
    std::vector<uint32_t> idx;
    std::vector<float> val;
    for (int64_t r = 0; r < N; r++) {
        for (uint32_t t = 0; t < D; t++)
            if (rng() % 4 == 0) {
                idx.push_back(t);
                val.push_back((float)(rng() % 1000) / 500.0f - 1.0f);
            }
        ptr.push_back((int64_t)idx.size());
    }
    int rc = 0;
    emu_sparse_set_device_build(1);  // the postings build kernels run under the sanitizer too
    emu_sparse_set_heavy(7);         // and so do the row-streaming kernels: rows with more than 7 of the 24 indices
    for (int k : {5, 70}) {  // KP = 64 (many overflows with ~690 hits per query) and KP = 128
        const int64_t nq = 24;
        std::vector<int32_t> oi((size_t)nq * k), oc((size_t)nq);
        std::vector<float> os((size_t)nq * k);
        unsigned long long stat[2];
        rc |= emu_sparse_search(N, ptr.data(), idx.data(), val.data(), nq, nullptr, nullptr, nullptr, 100, nullptr, 1, nullptr, k,
                                3, 8, 2, 0, oi.data(), os.data(), oc.data(), stat);
    }
    std::printf("tsan run done rc=%d\n", rc);
    return rc;
}
