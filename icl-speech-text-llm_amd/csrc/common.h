// common.h — shared device/host helpers for libicl_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/icl_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- error plumbing (host) ------------------------------------------------------------------
void icl_set_error(const char* fmt, ...);
#define ICL_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      icl_set_error(__VA_ARGS__);           \
      return ICL_EINVAL;                    \
    }                                       \
  } while (0)
#define ICL_CHECK_LAUNCH(name)                                                      \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      icl_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));         \
      return ICL_ELAUNCH;                                                           \
    }                                                                               \
  } while (0)

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned int, v);
}
// RoPE on eight (i, i + D/2) pairs of one head (HF rotate_half pairing): lo/hi hold 8 bf16 each, c0|c1 and s0|s1 the eight
// f32 cos/sin values of those columns.  Products and sums are rounded separately (no FMA contraction) — the arithmetic of
// the f32 oracle — and the function is shared by the rope kernel and the fused QKV-GEMM epilogue so that both round alike.
__device__ __forceinline__ void rope_rot8(u32x4 lo, u32x4 hi, f32x4 c0, f32x4 c1, f32x4 s0, f32x4 s1, u32x4& olo,
                                          u32x4& ohi) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a0 = __uint_as_float(lo[j] << 16), a1 = __uint_as_float(lo[j] & 0xffff0000u);
    const float b0 = __uint_as_float(hi[j] << 16), b1 = __uint_as_float(hi[j] & 0xffff0000u);
    const float cA = j < 2 ? c0[2 * j] : c1[2 * j - 4], cB = j < 2 ? c0[2 * j + 1] : c1[2 * j - 3];
    const float sA = j < 2 ? s0[2 * j] : s1[2 * j - 4], sB = j < 2 ? s0[2 * j + 1] : s1[2 * j - 3];
    olo[j] = pack_bf16x2(__fsub_rn(__fmul_rn(a0, cA), __fmul_rn(b0, sA)), __fsub_rn(__fmul_rn(a1, cB), __fmul_rn(b1, sB)));
    ohi[j] = pack_bf16x2(__fadd_rn(__fmul_rn(b0, cA), __fmul_rn(a0, sA)), __fadd_rn(__fmul_rn(b1, cB), __fmul_rn(a1, sB)));
  }
}
// exact-erf GELU for the GEMM epilogues (128 values per thread, VALU-bound next to a K = 1280 main loop):
//     gelu(x) = x Phi(x) = relu(x) - |x| Q(|x|),   Q(t) = 1 - Phi(t) = erfc(t / sqrt 2) / 2 = 2^P(t)
// with P = log2 Q as ONE degree-7 polynomial (the -t^2/2 log2 e of the Gaussian tail folded into its quadratic term; minimax fit of
// log2(erfcx(t / sqrt 2) / 2) on [0, 7]: |dP| <= 1.4e-5, i.e. Q to 1e-5 RELATIVE everywhere; t is clamped to 7, where |x| Q < 1e-11).
// |gelu - exact| <= 1.5e-6 absolute and <= 1.3e-5 relative (tools/ checked over [-20, 20] in f32 arithmetic): two orders below a
// bf16 rounding of the result.  Per value pair: |x|, clamp, seven packed FMAs, two v_exp_f32, relu, one packed FMA — the
// Abramowitz-Stegun 7.1.26 form it replaces (rounds 1-3) needed a v_rcp_f32 per value on top (76 vs 108 issue cycles per pair) and
// carried a mistyped leading coefficient (0.53060 for 0.53070: 2.6e-5 absolute error, still far inside the tolerance, found when
// this form was checked against it).  Explicit FMAs only (the same contraction at every call site: all tile shapes must
// round alike), written on pairs so that hipcc emits v_pk_fma_f32 (two values per issue slot).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 ax = __builtin_elementwise_abs(x);
  const f32x2 t = __builtin_elementwise_min(ax, f32x2{7.0f, 7.0f});
  f32x2 p = {-1.5882353683e-06f, -1.5882353683e-06f};
  p = __builtin_elementwise_fma(p, t, f32x2{5.6153733911e-05f, 5.6153733911e-05f});
  p = __builtin_elementwise_fma(p, t, f32x2{-8.8321799332e-04f, -8.8321799332e-04f});
  p = __builtin_elementwise_fma(p, t, f32x2{8.3032251061e-03f, 8.3032251061e-03f});
  p = __builtin_elementwise_fma(p, t, f32x2{-5.3501311510e-02f, -5.3501311510e-02f});
  p = __builtin_elementwise_fma(p, t, f32x2{-4.5896438201e-01f, -4.5896438201e-01f});
  p = __builtin_elementwise_fma(p, t, f32x2{-1.1510356065e+00f, -1.1510356065e+00f});
  p = __builtin_elementwise_fma(p, t, f32x2{-1.0000140186e+00f, -1.0000140186e+00f});
  const f32x2 q = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
  const f32x2 relu = __builtin_elementwise_max(x, f32x2{0.0f, 0.0f});
  return __builtin_elementwise_fma(-ax, q, relu);
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
  const f32x2 a = gelu_erf2(f32x2{v[0], v[1]}), b = gelu_erf2(f32x2{v[2], v[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2(f32x2{x, x})[0]; }
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // no IEEE division sequence

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_reduce_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
