// Shared device helpers for libmmx_hip.so (gfx950 / CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// MMX_LAB = 1 (csrc/Makefile, target lab): the measurement build libmmx_hip_lab.so of the SAME sources - shader-clock stamps at
// the phase boundaries of the decode projections and of the fused estimator tail, switched on through the two mmx_lab_* entry
// points of include/mmx_hip_lab.h.  The product library is built with MMX_LAB = 0: no stamp is compiled in, no kernel reads
// a mutable global, no entry point carries a measurement argument.
#ifndef MMX_LAB
#define MMX_LAB 0
#endif
constexpr bool LAB = MMX_LAB != 0;

#define MMX_OK 0
#define MMX_EARG (-1)
#define MMX_CHECK_ARG(c) do { if (!(c)) return MMX_EARG; } while (0)
#define MMX_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return -(int)e_ - 1000; } while (0)

// entry points without a matrix product: the split builds (MMX_X2 / MMX_X3 = 2 / 3) store activations as fp32
#define MMX_ACT_DTYPE(d) ((d) >= 2 ? 0 : (d))

// More than 64 KB of dynamic LDS is an opt-in per function AND per device.  MMX_LDS_OPT_IN(fn, bytes) raises the limit
// of `fn` to 160 KB the first time the CURRENT device launches it (one flag per device, per call site = per template
// instantiation; racing first calls write the same value).  The first launch of any shape is an eager one (the engines
// record a graph only from the second call on), so the attribute write never happens inside a stream capture.
#define MMX_LDS_OPT_IN(fn, bytes)                                                                                   \
    do {                                                                                                            \
        if ((bytes) > 64 * 1024) {                                                                                  \
            static bool done_[64] = {};                                                                             \
            int dev_ = 0;                                                                                           \
            if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return MMX_EARG;                       \
            if (!done_[dev_]) {                                                                                     \
                hipError_t e_ = hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                if (e_ != hipSuccess) return -1000 - (int)e_;                                                       \
                done_[dev_] = true;                                                                                 \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)

typedef unsigned short bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even; NaN stays NaN (MI355X_MICROARCH.md "Correctness boundaries")
// hardware conversion: a plain cast to __bf16 lowers to v_cvt_pk_bf16_f32 on gfx950 (RNE, NaN preserved)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(unsigned, v);
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
    static __device__ __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};

// Cross-lane adds on the VALU (DPP) instead of the LDS pipe: hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS-pipe round
// trip of ~100 cycles), which sits on the dependent chain of every short reduction.  CTRL: quad_perm [1,0,3,2] = 0xB1 (xor 1),
// quad_perm [2,3,0,1] = 0x4E (xor 2), row_half_mirror = 0x141 (lane i <-> 7 - i inside 8 lanes: the other quad once both quads
// are reduced), row_mirror = 0x140 (lane i <-> 15 - i inside 16 lanes: the other half once both halves are reduced).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over aligned groups of 8 / 16 consecutive lanes, result in every lane of the group
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v = group8_sum(v);
    v += dpp_f32<0x140>(v);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// activations used by the fused epilogues (codes are part of the C ABI, include/mmx_hip.h)
enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_MISH = 4, ACT_TANH = 5 };

// compile-time activation: a runtime `switch` inside the per-element epilogue is if-converted by the compiler
// into evaluating EVERY activation (erf, tanh, log1p, exp ...) and selecting — ~190 instructions per output
// element, which made every GEMM epilogue and row-norm compute bound.  Kernels dispatch once, outside the loops.
// erf for the bf16 build: Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 (one rcp + one exp + 5 fma instead of the
// ~60-instruction libm path that cost 8 us on every FF1 epilogue); the fp32 parity build keeps erff.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}

// GELU through the same erf (A-S 7.1.26), every operation written out (explicit FMAs, no contraction left to the compiler) and
// in a two-element form whose multiplies / adds / FMAs are packed instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two
// elements per issue slot).  The fused estimator kernels evaluate it in two different register layouts (row layout of the 16- / 32-row
// tiles, the MFMA's own layout in the 64-row split tile); spelled out like this both give the same bits per element, whatever
// the vectoriser does with the surrounding loop.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t v) {
#pragma clang fp contract(off)
    const f32x2_t x = v * 0.70710678118654752f;
    const f32x2_t ax = __builtin_elementwise_abs(x);
    const f32x2_t d = __builtin_elementwise_fma(ax, (f32x2_t)(0.3275911f), (f32x2_t)(1.0f));
    f32x2_t t;
    t.x = __builtin_amdgcn_rcpf(d.x);
    t.y = __builtin_amdgcn_rcpf(d.y);
    f32x2_t p = (f32x2_t)(1.061405429f);
    p = __builtin_elementwise_fma(p, t, (f32x2_t)(-1.453152027f));
    p = __builtin_elementwise_fma(p, t, (f32x2_t)(1.421413741f));
    p = __builtin_elementwise_fma(p, t, (f32x2_t)(-0.284496736f));
    p = __builtin_elementwise_fma(p, t, (f32x2_t)(0.254829592f));
    const f32x2_t m = (ax * ax) * -1.4426950408889634f;    // exp(-x^2) = 2^(-x^2 log2 e)
    f32x2_t e;
    e.x = __builtin_amdgcn_exp2f(m.x);
    e.y = __builtin_amdgcn_exp2f(m.y);
    const f32x2_t r = __builtin_elementwise_fma(-(p * t), e, (f32x2_t)(1.0f));
    f32x2_t er;
    er.x = __builtin_copysignf(r.x, x.x);
    er.y = __builtin_copysignf(r.y, x.y);
    return (v * 0.5f) * (er + 1.0f);
}

// erf for the fused bf16 epilogues whose result is rounded to bf16: odd polynomial of degree 17 on |x| <= 3 (clamped
// beyond: 1 - erf(3) = 2.2e-5), |error| <= 2.4e-5 - no transcendental (v_rcp / v_exp issue at quarter rate; the FF1
// epilogue of the estimator's fused tail kernel is VALU bound: tools/tail_lab.py --stamps), all FMAs pack into
// v_pk_fma_f32.  GELU from it: |error| <= 5.2e-5 absolute.
__device__ __forceinline__ float erf_poly(float x) {
    const float u = __builtin_fminf(__builtin_fmaxf(x, -3.0f), 3.0f);
    const float t = u * u;
    float p = 4.074212256e-08f;
    p = p * t - 1.944823225e-06f;
    p = p * t + 4.106053166e-05f;
    p = p * t - 5.110369530e-04f;
    p = p * t + 4.235427827e-03f;
    p = p * t - 2.510286123e-02f;
    p = p * t + 1.110793352e-01f;
    p = p * t - 3.753148615e-01f;
    p = p * t + 1.128268480e+00f;
    return u * p;
}

// ACT_GELU_POLY: GELU through erf_poly (internal to the fused bf16 kernels, not an ABI activation code)
constexpr int ACT_GELU_POLY = 100;

template <int ACT, bool PRECISE>
__device__ __forceinline__ float act_c(float v, float slope) {
    if constexpr (ACT == ACT_LRELU) return v > 0.f ? v : v * slope;
    else if constexpr (ACT == ACT_GELU_POLY) return 0.5f * v * (1.f + erf_poly(v * 0.70710678118654752f));
    else if constexpr (ACT == ACT_GELU) return 0.5f * v * (1.f + (PRECISE ? erff(v * 0.70710678118654752f) : erf_fast(v * 0.70710678118654752f)));
    else if constexpr (ACT == ACT_SILU) return v / (1.f + (PRECISE ? expf(-v) : __expf(-v)));
    else if constexpr (ACT == ACT_MISH) {
        // x * tanh(softplus(x)) with tanh(log(1+e)) = (e^2 + 2e) / (e^2 + 2e + 2), e = exp(x)  (exact identity);
        // torch's softplus threshold: x > 20 -> x
        if (v > 20.f) return v;
        const float e = PRECISE ? expf(v) : __expf(v);
        const float n = e * (e + 2.f);
        return v * (n / (n + 2.f));
    } else if constexpr (ACT == ACT_TANH) return tanhf(v);
    else return v;
}

template <bool PRECISE>
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case ACT_LRELU: return v > 0.f ? v : v * slope;
        case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        case ACT_SILU: return v / (1.f + (PRECISE ? expf(-v) : __expf(-v)));
        case ACT_MISH: {   // x * tanh(softplus(x)), torch softplus threshold 20
            float sp = v > 20.f ? v : log1pf(PRECISE ? expf(v) : __expf(v));
            return v * tanhf(sp);
        }
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// sin for the parity builds (fp32 / split): libm's sinf is ~60 instructions with its large-argument path, and Snake takes a
// sine of every element of every DAC layer.  Cody-Waite reduction by pi in two fused steps (n * PI_HI is exact inside the
// FMA) and an odd degree-11 minimax polynomial on [-pi/2, pi/2]: |error| <= 1.3e-7 for |x| <= 4e4 (one fp32 rounding of
// a value <= 1; checked against float64 over N(0, s), s = 1 .. 1e4) - the accuracy of a correctly rounded sinf, 11 VALU
// instructions.  The Snake arguments alpha * x of the decoder are O(1 .. 100).
__device__ __forceinline__ float sin_cw(float x) {
    const float n = __builtin_rintf(x * 0.318309886183790672f);
    float r = __builtin_fmaf(-n, 3.14159274101257324f, x);           // PI_HI = float(pi)
    r = __builtin_fmaf(-n, -8.74227765734758577e-8f, r);             // PI_LO = pi - PI_HI
    const float t = r * r;
    float p = -2.3846693509e-08f;
    p = p * t + 2.7522619348e-06f;
    p = p * t - 1.9840804453e-04f;
    p = p * t + 8.3333300427e-03f;
    p = p * t - 1.6666667163e-01f;
    const float s = __builtin_fmaf(r * t, p, r);
    return (static_cast<int>(n) & 1) ? -s : s;
}

// dac-vae/layers.py:22  snake(x,a) = x + (a + 1e-9)^-1 * sin(a x)^2   (exact operation order)
template <bool PRECISE>
__device__ __forceinline__ float snake_apply(float x, float alpha) {
    float s = PRECISE ? sin_cw(alpha * x) : __sinf(alpha * x);
    return x + (1.0f / (alpha + 1e-9f)) * (s * s);
}
