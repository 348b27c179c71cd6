// hip_emu.hpp -- TEST INFRASTRUCTURE ONLY: runs the body of a HIP kernel on the CPU, one OS thread per work-item, one
// workgroup at a time, so that the control flow of a kernel (barrier placement, slot hand-out, loop termination, index
// arithmetic) can be checked against the oracle in the CPU test-suite of a container that has no GPU.  It supports
// exactly the subset gorse_amd/csrc/sparse_kernels.hpp is written in: threadIdx / blockIdx / blockDim / gridDim (.x),
// static __shared__ variables, __syncthreads, __syncthreads_or, atomicAdd on int and unsigned long long, the float bit
// casts and the round-to-nearest float intrinsics.  Nothing of the product includes this file; it says nothing about
// performance and nothing about gfx950 code generation -- the GPU parity tests (tests/test_gpu_vectors_sparse.py) do.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace emu {
struct Dim {
    unsigned x = 1, y = 1, z = 1;
};
inline thread_local Dim t_idx, b_idx;
inline Dim b_dim, g_dim;
// sense-reversing spin barrier (a pthread barrier sleeps in the kernel at every one of the kernel's many barriers)
struct Barrier {
    unsigned n = 1;
    unsigned waiting = 0;
    unsigned phase = 0;
    void init(unsigned count) { n = count, waiting = 0, phase = 0; }
    void wait() {
        const unsigned my = __atomic_load_n(&phase, __ATOMIC_ACQUIRE);
        if (__atomic_add_fetch(&waiting, 1, __ATOMIC_ACQ_REL) == n) {
            __atomic_store_n(&waiting, 0, __ATOMIC_RELAXED);
            __atomic_store_n(&phase, my + 1, __ATOMIC_RELEASE);
        } else {
            for (unsigned spins = 0; __atomic_load_n(&phase, __ATOMIC_ACQUIRE) == my; spins++)
                if (spins > 64) std::this_thread::yield();
        }
    }
};
inline Barrier bar;
inline int or_slot[3];
inline thread_local unsigned or_call;

// grid x block launch: `block` persistent threads walk the workgroups in order
inline void launch(unsigned grid, unsigned block, const std::function<void()> &body) {
    b_dim.x = block;
    g_dim.x = grid;
    or_slot[0] = or_slot[1] = or_slot[2] = 0;
    bar.init(block);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < block; t++)
        th.emplace_back([&, t] {
            t_idx.x = t;
            or_call = 0;
            for (unsigned b = 0; b < grid; b++) {
                b_idx.x = b;
                body();
                bar.wait();  // static __shared__ storage is reused by the next workgroup
            }
        });
    for (auto &x : th) x.join();
}
}  // namespace emu

#define __global__
#define __device__
#define __shared__ static
#define __launch_bounds__(n)
#define threadIdx emu::t_idx
#define blockIdx emu::b_idx
#define blockDim emu::b_dim
#define gridDim emu::g_dim

inline void __syncthreads() { emu::bar.wait(); }
// three rotating accumulators: call n uses slot n % 3 and clears slot (n + 1) % 3 before its barrier; slot n % 3 is
// cleared again during call n + 2, after every thread has passed the barrier of call n + 1 and therefore read it
inline int __syncthreads_or(int pred) {
    const unsigned n = emu::or_call++;
    __atomic_store_n(&emu::or_slot[(n + 1) % 3], 0, __ATOMIC_RELAXED);
    if (pred) __atomic_store_n(&emu::or_slot[n % 3], 1, __ATOMIC_RELAXED);
    emu::bar.wait();
    return __atomic_load_n(&emu::or_slot[n % 3], __ATOMIC_RELAXED);
}
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline uint32_t __float_as_uint(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float __uint_as_float(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// built with -ffp-contract=off: one rounding per operation, like the device intrinsics
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
