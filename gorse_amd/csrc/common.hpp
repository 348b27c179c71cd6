// common.hpp -- shared plumbing of libgorse_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gorse_hip.h"
#include "../../include/gorse_hip_test.h"

namespace gorse {

// ---- error reporting ---------------------------------------------------------------
inline std::string &last_error() {
    static thread_local std::string e;
    return e;
}
inline int32_t fail(int32_t code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}
#define GORSE_HIP_CHECK(expr)                                                                          \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return gorse::fail(GORSE_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,            \
                               hipGetErrorString(_e));                                                 \
    } while (0)
#define GORSE_TRY(expr)              \
    do {                             \
        int32_t _r = (expr);         \
        if (_r != GORSE_OK) return _r; \
    } while (0)

// ---- device buffer -----------------------------------------------------------------
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int32_t alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(GORSE_ERR_NOMEM, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
        }
        n = count;
        return GORSE_OK;
    }
    int32_t ensure(size_t count) { return count <= n ? GORSE_OK : alloc(count); }
};

// ---- per-kernel-class event profiling --------------------------------------------------
// Each profiled launch is bracketed by an event pair recorded on the stream the kernel
// runs on; pairs are resolved lazily (hipEventElapsedTime) when the profile is read.
struct KernelProfile {
    struct Pair {
        hipEvent_t a, b;
        int cls;
    };
    bool on = false;
    std::vector<Pair> pending;
    std::vector<hipEvent_t> pool;
    std::vector<int64_t> launches;
    std::vector<double> ms;
    explicit KernelProfile(int nclasses) : launches(nclasses, 0), ms(nclasses, 0.0) {}
    ~KernelProfile() {
        for (auto &p : pending) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    // usage: auto t = prof.begin(cls, stream); launch; prof.end(t, stream);
    int begin(int cls, hipStream_t s) {
        if (!on) return -1;
        Pair p{get(), get(), cls};
        (void)hipEventRecord(p.a, s);
        pending.push_back(p);
        return (int)pending.size() - 1;
    }
    void end(int tok, hipStream_t s) {
        if (tok >= 0) (void)hipEventRecord(pending[tok].b, s);
    }
    void resolve() {  // caller has synchronised the streams
        for (auto &p : pending) {
            float t = 0;
            if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
                ms[p.cls] += t;
                launches[p.cls] += 1;
            }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
    void reset() {
        resolve();
        for (auto &x : launches) x = 0;
        for (auto &x : ms) x = 0;
    }
};

// ---- Philox4x32-10 (Salmon et al., SC'11) + Go math/rand Int31n -------------------------
struct Philox {
    uint32_t c0, c1, c2, c3, k0, k1;
    uint32_t buf[4];
    int pos;
    __host__ __device__ inline void init(uint64_t seed, uint64_t epoch, uint64_t sample) {
        c0 = (uint32_t)sample;
        c1 = (uint32_t)(sample >> 32);
        c2 = 0;  // block counter
        c3 = (uint32_t)epoch;
        k0 = (uint32_t)seed;
        k1 = (uint32_t)(seed >> 32);
        pos = 4;
    }
    __host__ __device__ inline void block() {
        uint32_t a0 = c0, a1 = c1, a2 = c2, a3 = c3, x0 = k0, x1 = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            uint64_t p0 = (uint64_t)0xD2511F53u * a0;
            uint64_t p1 = (uint64_t)0xCD9E8D57u * a2;
            uint32_t n0 = (uint32_t)(p1 >> 32) ^ a1 ^ x0;
            uint32_t n1 = (uint32_t)p1;
            uint32_t n2 = (uint32_t)(p0 >> 32) ^ a3 ^ x1;
            uint32_t n3 = (uint32_t)p0;
            a0 = n0;
            a1 = n1;
            a2 = n2;
            a3 = n3;
            x0 += 0x9E3779B9u;
            x1 += 0xBB67AE85u;
        }
        buf[0] = a0;
        buf[1] = a1;
        buf[2] = a2;
        buf[3] = a3;
        c2++;
        pos = 0;
    }
    __host__ __device__ inline uint32_t int31() {
        if (pos == 4) block();
        // static indexing keeps buf in registers on the device
        uint32_t v = pos == 0 ? buf[0] : pos == 1 ? buf[1] : pos == 2 ? buf[2] : buf[3];
        pos++;
        return v >> 1;
    }
    // Go math/rand (*Rand).Int31n
    __host__ __device__ inline int32_t int31n(int32_t n) {
        if ((n & (n - 1)) == 0) return (int32_t)(int31() & (uint32_t)(n - 1));
        uint32_t mx = (uint32_t)((1u << 31) - 1 - (1u << 31) % (uint32_t)n);
        uint32_t v = int31();
        while (v > mx) v = int31();
        return (int32_t)(v % (uint32_t)n);
    }
};

constexpr int kMaxDraws = 4096;  // per-sample retry cap (the reference spins forever)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Host-side one-off passes over a dataset (row sorts, validation, counting) split over the host's cores: a Fit of the
// 10M x 1M set hands over 1e9 feedback entries, and a single thread spends most of a minute on them.  fn(t, begin, end)
// works on rows [begin, end); `weight` (may be null) is the CSR row pointer the cut points are balanced by.
template <typename F>
inline void parallel_rows(int64_t rows, const int64_t *weight, F fn) {
    const int64_t work = weight ? weight[rows] - weight[0] : rows;
    int nt = (int)std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), 64);
    // a thread per 64K entries at least (starting one costs ~0.1 ms): S-ml1m's million entries took 20 ms of a 45 ms Fit on one thread
    nt = (int)std::min<int64_t>(nt, work >> 16);
    if (nt < 2 || rows < 2 * nt) nt = 1;
    if (nt == 1) {
        fn(0, (int64_t)0, rows);
        return;
    }
    std::vector<int64_t> cut((size_t)nt + 1, rows);
    cut[0] = 0;
    for (int t = 1; t < nt; t++) {
        if (weight)
            cut[t] = std::lower_bound(weight, weight + rows + 1, weight[0] + work * t / nt) - weight;
        else
            cut[t] = rows * t / nt;
        cut[t] = std::min(std::max(cut[t], cut[t - 1]), rows);
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t] { fn(t, cut[t], cut[t + 1]); });
    for (auto &x : th) x.join();
}

}  // namespace gorse
