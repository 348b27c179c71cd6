// probe_atomics.hip -- how do fp32 atomics / write-through stores to hot and cold rows behave on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/probe_atomics.hip -o gpurun_bin/probe_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// OP: 0 atomic add (no return), 1 sc1 store, 2 plain store, 3 plain load (sum kept), 4 sc1 load
// PAT: 0 = every wave hits row 0 (64 lanes x 4B contiguous);  1 = each 16-lane group hits a random row's 64B piece
//      2 = whole wave hits one random row (256B);  3 = 16-lane groups: group 0 hits hot row 0, others random
template <int OP, int PAT>
__global__ void k(float* buf, uint32_t nrows, int iters, float* sink) {
    const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t r;
        int off;
        if (PAT == 0) { r = 0; off = lane; }
        else if (PAT == 1) { r = hash(wave * 4 + grp + it * 0x9e3779b9u) % nrows; off = (it & 3) * 16 + gl; }
        else if (PAT == 2) { r = hash(wave + it * 0x9e3779b9u) % nrows; off = lane; }
        else { r = grp == 0 ? 0 : hash(wave * 4 + grp + it * 0x9e3779b9u) % nrows; off = (it & 3) * 16 + gl; }
        float* p = buf + (size_t)r * 64 + off;
        if (OP == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (OP == 1) __hip_atomic_store(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (OP == 2) *p = 1.0f;
        else if (OP == 3) acc += *(volatile float*)p;
        else acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 12345.f) *sink = acc;
}

template <int OP, int PAT>
int run(const char* name, float* buf, uint32_t nrows, float* sink) {
    const int blocks = 256 * 8, threads = 256, iters = 64;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<OP, PAT><<<blocks, threads>>>(buf, nrows, 4, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<OP, PAT><<<blocks, threads>>>(buf, nrows, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double winstr = (double)blocks * threads / 64 * iters;
    printf("%-44s %8.3f ms  %7.2f ns/wave-instr  %8.1f G lane-ops/s  %7.1f GB/s\n", name, ms, ms * 1e6 / winstr,
           winstr * 64 / (ms * 1e-3) / 1e9, winstr * 256 / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    uint32_t nrows = 1u << 20;  // 1M rows x 256 B = 256 MB
    float *buf, *sink;
    CK(hipMalloc(&buf, (size_t)nrows * 256)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, (size_t)nrows * 256));
    for (uint32_t rows : {1u << 20, 1u << 12}) {
        printf("---- rows = %u (%.1f MB)\n", rows, rows * 256.0 / 1e6);
        run<0, 0>("atomic  all waves -> row 0 (256B)", buf, rows, sink);
        run<0, 3>("atomic  1 of 4 groups -> hot row, 3 random", buf, rows, sink);
        run<0, 1>("atomic  16-lane groups -> random 64B pieces", buf, rows, sink);
        run<0, 2>("atomic  wave -> random row (256B)", buf, rows, sink);
        run<1, 0>("sc1 st  all waves -> row 0", buf, rows, sink);
        run<1, 1>("sc1 st  16-lane groups -> random 64B pieces", buf, rows, sink);
        run<1, 2>("sc1 st  wave -> random row", buf, rows, sink);
        run<2, 1>("plain st 16-lane groups -> random 64B pieces", buf, rows, sink);
        run<2, 2>("plain st wave -> random row", buf, rows, sink);
        run<3, 0>("plain ld all waves -> row 0", buf, rows, sink);
        run<3, 1>("plain ld 16-lane groups -> random 64B pieces", buf, rows, sink);
        run<3, 2>("plain ld wave -> random row", buf, rows, sink);
        run<4, 1>("sc1 ld  16-lane groups -> random 64B pieces", buf, rows, sink);
        run<4, 2>("sc1 ld  wave -> random row", buf, rows, sink);
    }
    return 0;
}
