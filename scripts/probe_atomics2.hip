// probe_atomics2.hip -- is the memory-side atomic unit of gfx950 bound per operation or per byte?  fp32 / u32 / u64 / f64 adds of 16-lane groups
// onto random 64-byte (128-byte for the 8-byte types) pieces and whole-wave rows.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/probe_atomics2.hip -o gpurun_bin/probe_atomics2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// OP: 0 f32 add, 1 u32 add, 2 u64 add, 3 f64 add, 4 f32 add with return
template <int OP, int PAT>
__global__ void k(char* buf, uint32_t nrows, int iters, float* sink) {
    const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    constexpr int W = (OP == 2 || OP == 3) ? 8 : 4;
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t r;
        int off;
        if (PAT == 1) { r = hash(wave * 4 + grp + it * 0x9e3779b9u) % nrows; off = ((it & 1) * 16 + gl); }
        else { r = hash(wave + it * 0x9e3779b9u) % nrows; off = lane; }
        char* p = buf + (size_t)r * 512 + (size_t)off * W;
        if (OP == 0) __hip_atomic_fetch_add((float*)p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (OP == 1) __hip_atomic_fetch_add((uint32_t*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (OP == 2) __hip_atomic_fetch_add((unsigned long long*)p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (OP == 3) __hip_atomic_fetch_add((double*)p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else acc += __hip_atomic_fetch_add((float*)p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 12345.f) *sink = acc;
}

template <int OP, int PAT>
int run(const char* name, char* buf, uint32_t nrows, float* sink) {
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
    double ops = (double)blocks * threads * iters;
    const int W = (OP == 2 || OP == 3) ? 8 : 4;
    printf("%-52s %8.3f ms  %8.1f G lane-ops/s  %7.1f GB/s\n", name, ms, ops / (ms * 1e-3) / 1e9, ops * W / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    uint32_t nrows = 1u << 19;  // 512K rows x 512 B = 256 MB
    char* buf; float* sink;
    CK(hipMalloc(&buf, (size_t)nrows * 512)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, (size_t)nrows * 512));
    for (uint32_t rows : {1u << 19, 1u << 12}) {
        printf("---- rows = %u (%.1f MB)\n", rows, rows * 512.0 / 1e6);
        run<0, 1>("f32 add  16-lane groups -> random 64B pieces", buf, rows, sink);
        run<1, 1>("u32 add  16-lane groups -> random 64B pieces", buf, rows, sink);
        run<2, 1>("u64 add  16-lane groups -> random 128B pieces", buf, rows, sink);
        run<3, 1>("f64 add  16-lane groups -> random 128B pieces", buf, rows, sink);
        run<4, 1>("f32 add returning, 16-lane groups", buf, rows, sink);
        run<0, 2>("f32 add  wave -> random row (256B)", buf, rows, sink);
        run<2, 2>("u64 add  wave -> random row (512B)", buf, rows, sink);
        run<3, 2>("f64 add  wave -> random row (512B)", buf, rows, sink);
    }
    return 0;
}
