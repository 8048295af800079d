// bf16x3.h -- the three-limb bf16 representation of an fp32 value (internal; shared by the opt-in kernels of the interval
// scorer: scorer_mfma.hip forward, scorer_bwd_gemm.hip backward).
//
// x = hi + mid + lo EXACTLY (round-to-nearest splits: 8 + 8 + 8 significant bits and the signs of the remainders).  A product
// of two values is the sum of nine limb products, each exact in fp32 (8 x 8 bits); six of them are accumulated by
// v_mfma_f32_32x32x16_bf16, smallest first: hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi.  What is dropped (mid*lo, lo*mid,
// lo*lo) is below 2^-23 |x y| per term -- the size of the rounding of one fp32 fmaf of the exact-fp32 kernels.
//
// COMPILER HAZARD (amdclang 22, ROCm 7.2): `__builtin_bit_cast(T, v[i])` of an ELEMENT of an ext_vector_type value reads element 0
// for every i (seen on the result of __builtin_amdgcn_raw_buffer_load_b128 and on an f32x16 accumulator: stores of acc[t][r] all
// wrote acc[t][0]).  Bit-cast the whole vector and index the result, or use __float_as_uint / __uint_as_float on the element.
#pragma once
#include <hip/hip_runtime.h>

namespace semicrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Limbs3 { bf16x8 h, m, l; };

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    unsigned r;                                          // {bf16(a) in bits 15:0, bf16(b) in bits 31:16}, round to nearest even
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));      // (written out: from `(__bf16)x` the compiler converts
    return r;                                            // the first element a second time, alone, for the shift below)
}
#ifndef SEMICRF_SPLIT_SCALAR_SUB
#define SEMICRF_SPLIT_SCALAR_SUB 1   // (0: `a - b`, i.e. v_pk_add_f32 -- measured 6-11 % slower kernels: backward 1.69 -> 1.58 ms, projection 0.238 -> 0.212)
#endif
__device__ __forceinline__ f32x2 sub2(const f32x2 a, const f32x2 b)
{
#if SEMICRF_SPLIT_SCALAR_SUB
    // two v_sub_f32 instead of the v_pk_add_f32 the compiler makes of `a - b` (packed fp32 next to matrix instructions is slow)
    f32x2 r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.x));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.y));
    return r;
#else
    return a - b;
#endif
}
// plain fp32 arithmetic of the staging lanes in the same spirit (the compiler would pair neighbouring operations into v_pk_*)
#if SEMICRF_SPLIT_SCALAR_SUB
__device__ __forceinline__ float fmac1(float a, float b, float c) { asm("v_fmac_f32 %0, %1, %2" : "+v"(c) : "v"(a), "v"(b)); return c; }
__device__ __forceinline__ float add1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
__device__ __forceinline__ float fmac1(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ float add1(float a, float b) { return a + b; }
#endif
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l)
{
    const unsigned hu = cvt_pk_bf16(a, b);
    const f32x2 x = {a, b};
    const f32x2 hf = {__builtin_bit_cast(float, hu << 16), __builtin_bit_cast(float, hu & 0xffff0000u)};
    const f32x2 r1 = sub2(x, hf);                                                // exact
    const unsigned mu = cvt_pk_bf16(r1.x, r1.y);
    const f32x2 mf = {__builtin_bit_cast(float, mu << 16), __builtin_bit_cast(float, mu & 0xffff0000u)};
    const f32x2 r2 = sub2(r1, mf);                                               // exact, and fits 8 bits
    h = hu; m = mu; l = cvt_pk_bf16(r2.x, r2.y);
}
// Eight values at once, level by level: the four pairs' chains (convert, widen, subtract, convert, ...) are independent, and
// written pair after pair the compiler keeps them in that order (every instruction waits for the one in front of it: ~12 cycles
// per instruction in the backward kernel's cycle stamps).
__device__ __forceinline__ Limbs3 split8(const v4f a, const v4f b)
{
    const f32x2 x[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
    unsigned h[4], m[4], l[4];
    f32x2 r1[4], r2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = cvt_pk_bf16(x[i].x, x[i].y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 hf = {__builtin_bit_cast(float, h[i] << 16), __builtin_bit_cast(float, h[i] & 0xffff0000u)};
        r1[i] = sub2(x[i], hf);                                                  // exact
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = cvt_pk_bf16(r1[i].x, r1[i].y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 mf = {__builtin_bit_cast(float, m[i] << 16), __builtin_bit_cast(float, m[i] & 0xffff0000u)};
        r2[i] = sub2(r1[i], mf);                                                 // exact, and fits 8 bits
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) l[i] = cvt_pk_bf16(r2[i].x, r2[i].y);
    Limbs3 r;
    r.h = __builtin_bit_cast(bf16x8, (u32x4){h[0], h[1], h[2], h[3]});
    r.m = __builtin_bit_cast(bf16x8, (u32x4){m[0], m[1], m[2], m[3]});
    r.l = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
    return r;
}
// The same pair after pair (fewer values alive at once: the forward's tile kernel, which sits at the register limit, keeps this form)
__device__ __forceinline__ Limbs3 split8_pairs(const v4f a, const v4f b)
{
    unsigned h[4], m[4], l[4];
    split_pair(a.x, a.y, h[0], m[0], l[0]);
    split_pair(a.z, a.w, h[1], m[1], l[1]);
    split_pair(b.x, b.y, h[2], m[2], l[2]);
    split_pair(b.z, b.w, h[3], m[3], l[3]);
    Limbs3 r;
    r.h = __builtin_bit_cast(bf16x8, (u32x4){h[0], h[1], h[2], h[3]});
    r.m = __builtin_bit_cast(bf16x8, (u32x4){m[0], m[1], m[2], m[3]});
    r.l = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
    return r;
}
__device__ __forceinline__ f32x16 mma6(const Limbs3& A, const Limbs3& B, f32x16 acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l, B.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.h, acc, 0, 0, 0);
    return acc;
}

}  // namespace semicrf
