// probe_atomics4.hip -- RETURNING integer atomics on FEW addresses (work-queue heads, run counters, bin cursors): what one address
// serves per second, against the number of addresses and of waves that ask.  Every wave's lane 0 does `iters` fetch-adds on
// counter[(wave + it) % naddr * stride] and waits for each result before the next (as a work loop does); the rate is
// (waves x iters) / time.  Variants: returning / fire-and-forget; one lane per wave / all 64 lanes on 64 different addresses.
// build: hipcc --offload-arch=gfx950 -O3 scripts/probe_atomics4.hip -o gpurun_bin/probe_atomics4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool RET>
__global__ void k_queue(int* ctr, int naddr, int stride, int iters, int* sink) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int acc = 0;
    if (lane == 0) {
        for (int it = 0; it < iters; it++) {
            int* p = ctr + (size_t)((wave + it) % naddr) * stride;
            if (RET) {
                int v = __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += v;
                asm volatile("" : "+v"(acc));  // the next request is not issued before this one has returned
            } else {
                __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (acc == 0x7fffffff) *sink = acc;
}

// every lane its own sample: address = hash % naddr (the run counters of bpr_sample_user_kernel)
__global__ void k_counters(int* ctr, int naddr, int iters, int* sink) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    int acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t x = tid * 2654435761u + it * 0x9e3779b9u;
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        acc += __hip_atomic_fetch_add(ctr + x % naddr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0x7fffffff) *sink = acc;
}

int main() {
    int *ctr, *sink;
    const size_t words = (size_t)1 << 24;
    CK(hipMalloc(&ctr, words * 4));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](auto launch) {
        hipMemset(ctr, 0, words * 4);
        launch();  // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    printf("# work-queue pattern: lane 0 of every wave, each request waits for the one before (returning) or not (fire-and-forget)\n");
    for (int waves : {256, 1024, 4096}) {
        for (int naddr : {1, 2, 8, 64, 1024}) {
            for (int stride : {1, 64}) {
                if (naddr == 1 && stride != 1) continue;
                const int iters = 200;
                float ms = time([&] { k_queue<true><<<waves / 4, 256>>>(ctr, naddr, stride, iters, sink); });
                float ms2 = time([&] { k_queue<false><<<waves / 4, 256>>>(ctr, naddr, stride, iters, sink); });
                printf("waves %5d addresses %5d stride %3d words: returning %8.3f ms = %7.1f M/s (%.3f us per request of an address), fire-and-forget %8.3f ms = %7.1f M/s\n",
                       waves, naddr, stride, ms, waves * (double)iters / ms / 1e3, ms * 1e3 / (waves * (double)iters / naddr), ms2,
                       waves * (double)iters / ms2 / 1e3);
            }
        }
    }
    printf("# counters pattern: every lane a returning add on a random one of N counters (1M threads x 1)\n");
    for (int naddr : {1, 64, 1024, 6040, 65536, 1000000}) {
        float ms = time([&] { k_counters<<<4096, 256>>>(ctr, naddr, 1, sink); });
        printf("counters %8d: %8.3f ms for 1048576 adds = %7.1f M/s; per counter %.1f adds, %.3f us each\n", naddr, ms, 1048576.0 / ms / 1e3,
               1048576.0 / naddr, ms * 1e3 / (1048576.0 / naddr));
    }
    return 0;
}
