// rank_keys.hpp -- the integer encodings the ranking kernels sort and compare by, usable on host and device: the kernels
// include it (topk_mfma.hip, sparse_kernels.hpp), and so does the host library's test hook (gorse_amd/host/gorse_host_capi.cpp:
// gh_test_rank_key), so that their order / round-trip properties are checked without a GPU (tests/test_rank_keys_cpu.py).
#pragma once
#include <cstdint>
#include <cstring>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GORSE_RK_HD __host__ __device__ inline
#else
#define GORSE_RK_HD inline
#endif

namespace gorse {
namespace rank {

GORSE_RK_HD uint32_t f2u(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return u;
}
GORSE_RK_HD float u2f(uint32_t u) {
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// dense sweep: order-preserving float -> uint of an approximate score, every bit pattern distinct (-0 < +0), and back
GORSE_RK_HD uint32_t fkey(float x) {
    const uint32_t b = f2u(x);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
GORSE_RK_HD float fkey_inv(uint32_t k) { return u2f((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

// dense rescoring: key of an exact distance for the bitonic sort -- ascending, -0 and +0 equal (the reference compares
// floats), the sign of a zero kept apart as a flag so that the distance's bits come back; 0xffffffff sorts last (the query
// itself, padding); NaN never gets a key (the query is flagged)
constexpr uint32_t kDistKeyLast = 0xffffffffu;
GORSE_RK_HD uint32_t dist_key(float e) {
    const uint32_t u = f2u(e);
    return e == 0.0f ? 0x80000000u : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
}
GORSE_RK_HD bool dist_is_negative_zero(float e) { return f2u(e) == 0x80000000u; }
GORSE_RK_HD float dist_from_key(uint32_t key, bool negative_zero) {
    return u2f(negative_zero ? 0x80000000u : ((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key));
}
GORSE_RK_HD bool dist_key_nonpositive(uint32_t key) { return key <= 0x80000000u; }  // !(e > 0) for a keyed distance

// sparse top-k: larger score <=> larger ord, -0 counts as +0; a 64-bit key ranks by score, then by ASCENDING row
constexpr uint32_t kZeroOrd = 0x80000000u;  // ordered bits of +0
GORSE_RK_HD uint32_t score_ord(float score) {
    uint32_t u = f2u(score);
    if ((u << 1) == 0) u = 0;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
GORSE_RK_HD unsigned long long make_key(uint32_t ord, int32_t row) {
    return ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)row);
}
GORSE_RK_HD float key_score(unsigned long long key) {
    uint32_t u = (uint32_t)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return u2f(u);
}
GORSE_RK_HD int32_t key_row(unsigned long long key) { return (int32_t)(0xFFFFFFFFu - (uint32_t)key); }
// results the reference returns (xvec.go:379-446): it ranks every admissible document, cuts to k, drops Score == 0.
// pos / neg = admissible rows scoring above / below zero, adm = admissible rows; the rest score zero.
GORSE_RK_HD int written(long long pos, long long neg, long long adm, int k) {
    if (pos >= k) return k;
    const long long zeros = adm - pos - neg;
    long long n = pos;
    if (pos + zeros < k) n += neg < k - pos - zeros ? neg : k - pos - zeros;
    return (int)n;
}

}  // namespace rank
}  // namespace gorse
