// probe_atomics3.hip -- the atomic unit under BPR's access mix: a 16-lane group adds a whole 256-byte row (d = 64: four 64-byte
// pieces) at a random row of an N-row table, (a) nothing else, (b) after LOADING that row (agent-scope loads, as the gathers of
// csrc/bpr.hip) two iterations earlier, (c) the same with plain cached loads, (d) rows drawn only from the eighth of the table that
// belongs to the XCD the wave runs on (are atomics cheaper when a line is only ever touched from one XCD?), (e) a zipf-like draw.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/probe_atomics3.hip -o gpurun_bin/probe_atomics3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// LOADS: 0 none, 1 agent-scope (sc1) loads, 2 plain loads;  PART: rows of the wave's own XCD only;  ZIPF: skewed draw
template <int LOADS, bool PART, bool ZIPF>
__global__ void k(float* buf, uint32_t nrows, int iters, float* sink, float* snap = nullptr, int refreshers = 0, int* done = nullptr) {
    if ((int)blockIdx.x < refreshers) {  // copy the table into the snapshot, pass after pass, until the workers are done
        const int64_t n4 = (int64_t)nrows * 16, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)refreshers * blockDim.x;
        int passes = 0;
        for (;; passes++) {
            for (int64_t e = tid; e < n4; e += nt) {
                float4 v;
                v.x = __hip_atomic_load(buf + 4 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v.y = __hip_atomic_load(buf + 4 * e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v.z = __hip_atomic_load(buf + 4 * e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v.w = __hip_atomic_load(buf + 4 * e + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(snap + 4 * e, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(snap + 4 * e + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(snap + 4 * e + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(snap + 4 * e + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)gridDim.x - refreshers) break;
        }
        if (tid == 0) sink[1] = (float)passes;
        return;
    }
    const float* src = snap ? snap : buf;
    const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
    const uint32_t wave = ((blockIdx.x - refreshers) * blockDim.x + threadIdx.x) >> 6;
    uint32_t xcc = 0;
    if (PART) {
        uint32_t v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        xcc = v & 7;
    }
    auto draw = [&](int it) {
        uint32_t h = hash((wave * 4 + grp) * 2654435761u + it * 0x9e3779b9u);
        uint32_t r;
        if (ZIPF) {  // r ~ nrows^u: log-uniform (density 1/r)
            float u = (h >> 8) * (1.0f / 16777216.0f);
            r = (uint32_t)__expf(u * __logf((float)nrows));
            if (r >= nrows) r = nrows - 1;
            r = hash(r) % nrows;  // scatter the popular rows over the table
        } else r = h % nrows;
        if (PART) r = (r % (nrows / 8)) + xcc * (nrows / 8);
        return r;
    };
    float acc = 0;
    uint32_t r0 = draw(0), r1 = draw(1);
    float a0[4], a1[4], a2[4];
    for (int c = 0; c < 4; c++) { a0[c] = 0; a1[c] = 0; a2[c] = 0; }
    for (int it = 0; it < iters; it++) {
        const uint32_t r2 = draw(it + 2);
        if (LOADS) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float* p = src + (size_t)r2 * 64 + 16 * c + gl;
                a2[c] = LOADS == 1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *(const volatile float*)p;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            acc += a0[c];
            __hip_atomic_fetch_add(buf + (size_t)r0 * 64 + 16 * c + gl, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) { a0[c] = a1[c]; a1[c] = a2[c]; }
        r0 = r1; r1 = r2;
    }
    if (acc == 12345.f) *sink = acc;
    if (done) {
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int LOADS, bool PART, bool ZIPF>
int run(const char* name, float* buf, uint32_t nrows, float* sink, int blocks, float* snap = nullptr, int refreshers = 0, int* done = nullptr) {
    const int threads = 256, iters = 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    if (done) CK(hipMemset(done, 0, 4));
    k<LOADS, PART, ZIPF><<<blocks + refreshers, threads>>>(buf, nrows, 8, sink, snap, refreshers, done);
    CK(hipDeviceSynchronize());
    if (done) CK(hipMemset(done, 0, 4));
    CK(hipEventRecord(a));
    k<LOADS, PART, ZIPF><<<blocks + refreshers, threads>>>(buf, nrows, iters, sink, snap, refreshers, done);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)blocks * threads * iters * 4;
    float passes = 0;
    if (refreshers) CK(hipMemcpy(&passes, sink + 1, 4, hipMemcpyDeviceToHost));
    printf("  %-64s %8.3f ms  %8.1f G atomic dwords/s", name, ms, ops / (ms * 1e-3) / 1e9);
    if (refreshers) printf("  (%d refresher workgroups: %.0f passes, one per %.2f us)", refreshers, passes, ms * 1e3 / (passes > 0 ? passes : 1));
    printf("\n");
    return 0;
}

int main() {
    const uint32_t maxrows = 1u << 20;
    float *buf, *sink, *snap;
    int* done;
    CK(hipMalloc(&buf, (size_t)maxrows * 256)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&snap, (size_t)32768 * 256)); CK(hipMalloc(&done, 4));
    CK(hipMemset(snap, 0, (size_t)32768 * 256));
    CK(hipMemset(buf, 0, (size_t)maxrows * 256));
    for (int blocks : {378, 2048}) {
        for (uint32_t rows : {3704u, 29632u, 1u << 20}) {
            printf("---- %d workgroups of 256, table of %u rows x 256 B (%.1f MB)\n", blocks, rows, rows * 256.0 / 1e6);
            run<0, false, false>("atomics only, uniform rows", buf, rows, sink, blocks);
            run<1, false, false>("+ agent-scope loads of the row two iterations ahead", buf, rows, sink, blocks);
            run<2, false, false>("+ plain loads of the row two iterations ahead", buf, rows, sink, blocks);
            run<0, true, false>("atomics only, rows of the wave's own XCD", buf, rows, sink, blocks);
            run<1, true, false>("+ agent-scope loads, rows of the wave's own XCD", buf, rows, sink, blocks);
            run<0, false, true>("atomics only, log-uniform (zipf 1) rows", buf, rows, sink, blocks);
            run<1, false, true>("+ agent-scope loads, log-uniform rows", buf, rows, sink, blocks);
            if (rows <= 32768) {
                run<1, false, false>("+ agent-scope loads from a SEPARATE table (no refresher)", buf, rows, sink, blocks, snap);
                for (int rf : {16, 64})
                    run<1, false, false>("+ agent-scope loads from a snapshot being refreshed", buf, rows, sink, blocks, snap, rf, done);
            }
        }
    }
    return 0;
}
