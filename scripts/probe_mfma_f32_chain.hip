// probe_mfma_f32_chain.hip -- is v_mfma_f32_32x32x2_f32 the l-ascending fmaf chain, bit for bit?  One wave computes a 32 x 32 tile over K steps
// with the MFMA (two l per instruction) and the same tile with fmaf, element by element, from the same C; the two are compared as bits.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/probe_mfma_f32_chain.hip -o gpurun_bin/probe_mfma_f32_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_mfma(const float* A, const float* B, const float* C, float* D, int K) {  // A 32 x K, B K x 32, row-major
    const int l = threadIdx.x;
    f16v acc;
    for (int r = 0; r < 16; r++) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = A[(l & 31) * K + k0 + (l >> 5)], b = B[(k0 + (l >> 5)) * 32 + (l & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
__global__ void k_chain(const float* A, const float* B, const float* C, float* D, int K) {
    const int i = blockIdx.x, j = threadIdx.x;
    float acc = C[i * 32 + j];
    for (int l = 0; l < K; l++) acc = fmaf(A[i * K + l], B[l * 32 + j], acc);
    D[i * 32 + j] = acc;
}
int main() {
    const int K = 64;
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    int bad_total = 0;
    for (int trial = 0; trial < 6; trial++) {
        std::vector<float> A(32 * K), B(K * 32), C(32 * 32), D1(32 * 32), D2(32 * 32);
        for (auto* v : {&A, &B, &C})
            for (auto& x : *v) {
                float m = (float)(rnd() % 2000001) / 1000000.0f - 1.0f;
                int e = trial < 2 ? 0 : (int)(rnd() % (trial < 4 ? 12 : 60)) - (trial < 4 ? 6 : 30);  // wider and wider exponent ranges
                x = ldexpf(m, e);
                if (trial == 5 && rnd() % 7 == 0) x = 0.0f * (rnd() & 1 ? -1.0f : 1.0f);
            }
        float *dA, *dB, *dC, *dD1, *dD2;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMalloc(&dD1, D1.size() * 4)); CK(hipMalloc(&dD2, D2.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
        k_mfma<<<1, 64>>>(dA, dB, dC, dD1, K);
        k_chain<<<32, 32>>>(dA, dB, dC, dD2, K);
        CK(hipMemcpy(D1.data(), dD1, D1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(D2.data(), dD2, D2.size() * 4, hipMemcpyDeviceToHost));
        // host fmaf chain as a third opinion
        int bad = 0, bad_host = 0;
        double worst = 0;
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                float acc = C[i * 32 + j];
                for (int l = 0; l < K; l++) acc = fmaf(A[i * K + l], B[l * 32 + j], acc);
                uint32_t x, y, z;
                memcpy(&x, &D1[i * 32 + j], 4); memcpy(&y, &D2[i * 32 + j], 4); memcpy(&z, &acc, 4);
                bad += x != y;
                bad_host += y != z;
                if (x != y) worst = fmax(worst, fabs((double)D1[i * 32 + j] - D2[i * 32 + j]) / fmax(fabs((double)D2[i * 32 + j]), 1e-300));
            }
        printf("trial %d: MFMA vs device fmaf chain: %d of 1024 elements differ (worst relative %.3g); device chain vs host chain: %d differ\n", trial, bad, worst, bad_host);
        bad_total += bad;
    }
    printf("%s\n", bad_total == 0 ? "v_mfma_f32_32x32x2_f32 IS the l-ascending fmaf chain on these inputs" : "NOT bit-identical");
    return 0;
}
