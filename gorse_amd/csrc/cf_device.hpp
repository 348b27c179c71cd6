// cf_device.hpp -- device primitives shared by the CF kernels.
//
// Work layout: one SAMPLE (or one (user,item) pair) per 16-lane group = one DPP "row" of
// a wave64, i.e. 4 samples per wavefront.  Lane l of a group owns vector elements
// l, l+16, l+32, ... -- the same lane<->element map as the 16 fp32 lanes of the reference's
// AVX512 kernels (common/floats/src/floats_avx512.c), so the per-lane FMA chain followed by
// the rotate-add butterfly below reproduces `_mm512_dot` bit for bit, while every global
// access of a group is one contiguous 64-byte segment.
#pragma once
#include "common.hpp"

namespace gorse {

constexpr int kGroup = 16;         // lanes per sample
constexpr int kBlock = 256;        // threads per workgroup (4 waves, 16 groups)
constexpr int kGroupsPerBlock = kBlock / kGroup;

// add the value held by lane (l - n) mod 16 of the same 16-lane row (DPP row_ror:n)
template <int N>
__device__ __forceinline__ float row_ror_add(float v) {
    int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false);
    return v + __int_as_float(moved);
}

// 16 -> 1: the reduction tree of floats_avx512.c:328-340 ((l,l+8), (l,l+4), (l,l+2), (0,1)).
// After the first step lanes l and l^8 hold the same bits, so rotating by 4, 2, 1 pairs the
// same partial sums the AVX shuffles pair; every lane ends with the identical total.
__device__ __forceinline__ float group_tree16(float v) {
    v = row_ror_add<8>(v);
    v = row_ror_add<4>(v);
    v = row_ror_add<2>(v);
    v = row_ror_add<1>(v);
    return v;
}
// 8 -> 1 (floats_avx512.c:350-358); the 8 products must be replicated in lanes 8..15.
__device__ __forceinline__ float group_tree8(float v) {
    v = row_ror_add<4>(v);
    v = row_ror_add<2>(v);
    v = row_ror_add<1>(v);
    return v;
}

// The same 8 -> 1 tree inside each aligned group of EIGHT lanes (no mirror: the two halves of a 16-lane row hold different data): the
// partner of lane l is l ^ 4, then l ^ 2, then l ^ 1 -- the pairs the rotations above add, and since an addition's operands commute bit
// for bit every lane of the eight ends with group_tree8's total.  l ^ 4 by two DPP moves (quad_perm [3,2,1,0]: l ^ 3, then
// row_half_mirror: 7 - l = l ^ 7), l ^ 2 and l ^ 1 by one quad_perm each.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float group_tree8_halves(float v) {
    v = v + dpp_move<0x141>(dpp_move<0x1B>(v));
    v = v + dpp_move<0x4E>(v);
    v = v + dpp_move<0xB1>(v);
    return v;
}

// Vector-length bookkeeping of the AVX512 kernels: full 16-chunks, optional 8-tail, scalar tail.
struct VecShape {
    int d, nfull, has8, tail0;  // tail0 = first scalar-tail element
    __host__ __device__ explicit VecShape(int d_) : d(d_) {
        nfull = d / 16;
        has8 = (d % 16) >= 8;
        tail0 = nfull * 16 + (has8 ? 8 : 0);
    }
    // element e belongs to the unfused (mul, add) 8-lane tail of MulConstAdd?
    __host__ __device__ bool unfused(int e) const { return has8 && e >= nfull * 16 && e < tail0; }
};

// floats.Dot in AVX512 order for rows staged in LDS (generic d).  a, b: LDS pointers.
__device__ __forceinline__ float dot512_lds(const float *a, const float *b, const VecShape &vs, int lane) {
    float acc = 0.0f;
    if (vs.nfull > 0) acc = a[lane] * b[lane];
    for (int c = 1; c < vs.nfull; c++) acc = fmaf(a[16 * c + lane], b[16 * c + lane], acc);
    float sum = group_tree16(acc);
    if (vs.has8) {
        int e = vs.nfull * 16 + (lane & 7);
        sum += group_tree8(a[e] * b[e]);
    }
    for (int e = vs.tail0; e < vs.d; e++) sum = fmaf(a[e], b[e], sum);
    return sum;
}

// floats.Dot in AVX512 order for d == 16*NC held in registers.
template <int NC>
__device__ __forceinline__ float dot512_regs(const float (&a)[NC], const float (&b)[NC]) {
    float acc = a[0] * b[0];
#pragma unroll
    for (int c = 1; c < NC; c++) acc = fmaf(a[c], b[c], acc);
    return group_tree16(acc);
}

// math32.Exp restated (see oracle/gorse_oracle.c orc_exp_restated; chewxy/math32 exp.go states
// the FreeBSD e_exp.c scheme in float32).  Written without contraction (the library is built
// with -ffp-contract=off) so host oracle and device agree bit for bit.
__device__ __forceinline__ float exp_restated(float x) {
    const float Ln2Hi = 6.9313812256e-01f, Ln2Lo = 9.0580006145e-06f, Log2e = 1.4426950216e+00f;
    const float P1 = 1.66666666666666019037e-01f, P2 = -2.77777777770155933842e-03f,
                P3 = 6.61375632143793436117e-05f, P4 = -1.65339022054652515390e-06f,
                P5 = 4.13813679705723846039e-08f;
    if (x != x) return x;
    if (x > 88.72283905206835f) return __int_as_float(0x7f800000);
    if (x < -103.97207708f) return 0.0f;
    if (-3.725290298e-09f < x && x < 3.725290298e-09f) return 1.0f + x;
    int k = 0;
    if (x < 0)
        k = (int)(Log2e * x - 0.5f);
    else if (x > 0)
        k = (int)(Log2e * x + 0.5f);
    float hi = x - (float)k * Ln2Hi;
    float lo = (float)k * Ln2Lo;
    float r = hi - lo;
    float t = r * r;
    float c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    float y = 1.0f - ((lo - (r * c) / (2.0f - c)) - hi);
    return ldexpf(y, k);
}

}  // namespace gorse
