// fake_hip.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for the HIP runtime entry points libgorse_hip.so imports, so that the
// library's own HOST code for the sparse top-k (csrc/sparse.hip: validation, index build, uploads, scratch and stamp
// management, launches, downloads, statistics) runs in a container without a GPU.  Built as libfakehip.so and LD_PRELOADed
// into a child pytest process by tests/test_sparse_fake_runtime_cpu.py, it makes "device" memory plain host memory, streams
// synchronous, and hipLaunchKernel run the CPU emulation (hip_emu.hpp) of the kernels of csrc/sparse_kernels.hpp, found by
// the names the library's own module constructor registers.  Every other kernel is refused with hipErrorInvalidDeviceFunction.
// Nothing of the product knows about this file; it proves nothing about gfx950 code generation or speed.
#include "hip_emu.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

#include "../../gorse_amd/csrc/sparse_kernels.hpp"

using namespace gorse::sparse;

namespace {
struct Dim3 {
    uint32_t x, y, z;
};
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
const hipError_t kSuccess = 0, kInvalidValue = 1, kInvalidDeviceFunction = 98;

std::map<const void *, std::string> &kernels() {
    static std::map<const void *, std::string> k;
    return k;
}
hipError_t g_last = 0;
Dim3 g_cfg_grid, g_cfg_block;
size_t g_cfg_shmem;
hipStream_t g_cfg_stream;
int g_fatbin_token;

int template_int(const std::string &name, int which = 0) {  // "...kernelILi128ELi512EEE...": 0 -> 128, 1 -> 512
    size_t at = name.find("ILi");
    for (int w = 0; w < which && at != std::string::npos; w++) at = name.find("ELi", at + 1);
    return at == std::string::npos ? 0 : std::atoi(name.c_str() + at + 3);
}
}  // namespace

#define API extern "C" __attribute__((visibility("default")))

API hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return kSuccess;
}
API hipError_t hipSetDevice(int d) { return d == 0 ? kSuccess : kInvalidValue; }
API hipError_t hipGetLastError() {
    const hipError_t e = g_last;
    g_last = kSuccess;
    return e;
}
API const char *hipGetErrorString(hipError_t e) {
    return e == kSuccess ? "no error" : (e == kInvalidDeviceFunction ? "fake HIP runtime: this kernel is not emulated" : "fake HIP runtime: error");
}
API hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = std::malloc(1);
    return kSuccess;
}
API hipError_t hipStreamDestroy(hipStream_t s) {
    std::free(s);
    return kSuccess;
}
API hipError_t hipStreamSynchronize(hipStream_t) { return kSuccess; }
API hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return kSuccess; }
API hipError_t hipMalloc(void **p, size_t n) {
    *p = std::malloc(n ? n : 1);
    return *p ? kSuccess : 2;  // hipErrorOutOfMemory
}
API hipError_t hipFree(void *p) {
    std::free(p);
    return kSuccess;
}
API hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, int, hipStream_t) {
    std::memcpy(dst, src, n);
    return kSuccess;
}
API hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) {
    std::memset(dst, v, n);
    return kSuccess;
}
API hipError_t hipEventCreate(hipEvent_t *e) {
    *e = std::malloc(1);
    return kSuccess;
}
API hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
API hipError_t hipEventDestroy(hipEvent_t e) {
    std::free(e);
    return kSuccess;
}
API hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return kSuccess; }
API hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
    *ms = 1.0f;
    return kSuccess;
}
API hipError_t hipFuncSetAttribute(const void *, int, int) { return kSuccess; }

// what the module constructor hipcc emits into libgorse_hip.so calls at load time
API void **__hipRegisterFatBinary(const void *) { return reinterpret_cast<void **>(&g_fatbin_token); }
API void __hipUnregisterFatBinary(void **) {}
API void __hipRegisterFunction(void **, const void *hostFunction, char *, const char *deviceName, unsigned, void *, void *, void *, void *,
                               int *) {
    kernels()[hostFunction] = deviceName;
}
API void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
API hipError_t __hipPushCallConfiguration(Dim3 grid, Dim3 block, size_t shmem, hipStream_t stream) {
    g_cfg_grid = grid, g_cfg_block = block, g_cfg_shmem = shmem, g_cfg_stream = stream;
    return kSuccess;
}
API hipError_t __hipPopCallConfiguration(Dim3 *grid, Dim3 *block, size_t *shmem, hipStream_t *stream) {
    *grid = g_cfg_grid, *block = g_cfg_block, *shmem = g_cfg_shmem, *stream = g_cfg_stream;
    return kSuccess;
}

API hipError_t hipLaunchKernel(const void *func, Dim3 grid, Dim3 block, void **args, size_t, hipStream_t) {
    auto it = kernels().find(func);
    const std::string name = it == kernels().end() ? "" : it->second;
    if (grid.y != 1 || grid.z != 1 || block.y != 1 || block.z != 1) return g_last = kInvalidValue;
    if (name.find("sparse_query_kernel") != std::string::npos) {
        const QueryArgs a = *static_cast<const QueryArgs *>(args[0]);
        const int kp = template_int(name, 0), hot = template_int(name, 1);
        // the kernel is written for any workgroup size up to KP / 4: eight OS threads per workgroup instead of the real 64
        // lanes keep this container's cores from being oversubscribed (the 64-lane form runs in test_sparse_kernel_emu_cpu.py)
        emu::launch(grid.x, block.x < 8 ? block.x : 8, [&] {
            switch (kp * 10000 + hot) {
#define Q(KP, HOT) \
    case KP * 10000 + HOT: sparse_query_kernel<KP, HOT>(a); break;
                Q(256, 0) Q(512, 0) Q(1024, 0)
                Q(256, 512) Q(512, 512) Q(1024, 512)
                Q(256, 1024) Q(512, 1024) Q(1024, 1024)
#undef Q
                default: std::fprintf(stderr, "fake HIP runtime: unknown instantiation %s\n", name.c_str()); std::abort();
            }
        });
        return kSuccess;
    }
    if (name.find("sparse_count_kernel") != std::string::npos || name.find("sparse_scan_kernel") != std::string::npos ||
        name.find("sparse_scatter_kernel") != std::string::npos) {
        const BuildArgs b = *static_cast<const BuildArgs *>(args[0]);
        // the real grids (thousands of 256-thread workgroups) would take minutes here: same kernels, fewer work-items
        const bool scan = name.find("sparse_scan_kernel") != std::string::npos;
        const unsigned g = scan ? 1 : (grid.x < 3 ? grid.x : 3), t = block.x < 16 ? block.x : 16;
        emu::launch(g, t, [&] {
            if (name.find("sparse_count_kernel") != std::string::npos)
                sparse_count_kernel(b);
            else if (scan)
                sparse_scan_kernel(b);
            else
                sparse_scatter_kernel(b);
        });
        return kSuccess;
    }
    if (name.find("sparse_heavy_score_kernel") != std::string::npos) {
        const HeavyArgs h = *static_cast<const HeavyArgs *>(args[0]);
        emu::launch(grid.x < 3 ? grid.x : 3, block.x < 16 ? block.x : 16, [&] { sparse_heavy_score_kernel(h); });
        return kSuccess;
    }
    if (name.find("sparse_heavy_rank_kernel") != std::string::npos) {
        const HeavyArgs h = *static_cast<const HeavyArgs *>(args[0]);
        // the real block is 1024 lanes: 8 OS threads run the same code (any block size up to KP = 1024 is legal)
        emu::launch(grid.x, block.x < 8 ? block.x : 8, [&] { sparse_heavy_rank_kernel(h); });
        return kSuccess;
    }
    return g_last = kInvalidDeviceFunction;
}
