// Shared device/host helpers for the MaskBit gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// 16-bit storage type of activations and packed weights.  IEEE fp16 by default: the sampler feeds its
// own output back for 64 steps and classifier-free guidance multiplies logit errors by up to ~8x,
// so the 8-bit mantissa of bf16 costs ~1e-2 token mismatch against the fp32 reference where fp16's
// 11 bits give ~1e-3 at the same MFMA rate (measured; DESIGN.md "Precision").  Build with
// -DMB_HALF_BF16=1 to get the bf16 variant for A/B runs.  Accumulation is always fp32.
#ifndef MB_HALF_BF16
#define MB_HALF_BF16 0
#endif
#if MB_HALF_BF16
typedef __bf16 h16;
#define MB_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MB_H16_MAX 3.3e38f
#else
typedef _Float16 h16;
#define MB_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MB_H16_MAX 65504.0f
#endif
typedef __attribute__((ext_vector_type(2))) h16 h16x2;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(8))) h16 h16x8;

// fp32 -> storage half with saturation instead of +-inf (fp16 only; a no-op clamp for bf16)
__device__ __forceinline__ h16 to_h(float x) { return (h16)__builtin_amdgcn_fmed3f(x, -MB_H16_MAX, MB_H16_MAX); }
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MB_WAVE 64

// Global -> LDS DMA of 16 B per lane: LDS destination is wave-uniform base + lane*16.
#define MB_GLDS16_AUX(gptr, ldsptr, aux)                                                        \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),       \
                                   (void __attribute__((address_space(3)))*)(ldsptr), 16, 0, aux)
#define MB_GLDS16(gptr, ldsptr) MB_GLDS16_AUX(gptr, ldsptr, 0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Exact-erf GELU on a pair of values with packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32: two lanes-worth per issue).
// gelu(x) = x*Phi(x) = max(x,0) - |x| * erfc(|x|/sqrt2)/2, and erfc(a) = t*P(t)*exp(-a^2) with t = 1/(1 + 0.3275911 a)
// (Abramowitz-Stegun 7.1.26, |err| < 1.5e-7); the 1/2 and the 1/sqrt2 are folded into the constants. Writing the
// negative branch as -|x|*erfc/2 also avoids the 1-(1-e) cancellation of the textbook 0.5*x*(1+erf) form.
// (Tried: A-S 7.1.28, erfc(a) = (1 + a1 a + .. + a6 a^6)^-16 -- one quarter-rate instruction per element instead of two.  0.75 % off the FFN-up
// GEMM, 2-7 positions of token parity lost in every mode (its error, <= 3e-7 nominal, is no longer smooth after the ^16): not kept.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Reductions over the four 16-lane rows of a wave (lanes 16 / 32 apart) on the VALU: v_permlane16_swap / v_permlane32_swap of a value with
// itself leave (own, partner's) in the two results, in either order -- max and sum are symmetric, so no select.  (__shfl_xor goes through the
// LDS crossbar: ds_bpermute + its address + a lgkmcnt wait, in the middle of the attention softmax's dependent chain.)
__device__ __forceinline__ float rows_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// Maximum over the 16 lanes of a lane row on the VALU's DPP path (quad swaps, then the half-row and row mirrors: max is symmetric) -- four
// v_max with a DPP operand instead of four ds_bpermute round trips.
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, false)));    // quad_perm [1, 0, 3, 2]
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, false)));    // quad_perm [2, 3, 0, 1]
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, false)));   // row_half_mirror
  return fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, false))); // row_mirror
}
__device__ __forceinline__ float rows_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
#ifdef MB_NO_GELU                                           /* experiment (timing only): what the GELU arithmetic of the FFN-up epilogue costs */
  return x;
#endif
  const float a0 = fabsf(x.x), a1 = fabsf(x.y);
  const f32x2 t = {__builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, a0, 1.0f)),
                   __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, a1, 1.0f))};
  f32x2 p = __builtin_elementwise_fma(t, (f32x2)(0.5f * 1.061405429f), (f32x2)(-0.5f * 1.453152027f));
  p = __builtin_elementwise_fma(p, t, (f32x2)(0.5f * 1.421413741f));
  p = __builtin_elementwise_fma(p, t, (f32x2)(-0.5f * 0.284496736f));
  p = __builtin_elementwise_fma(p, t, (f32x2)(0.5f * 0.254829592f));
  p = p * t;
  const f32x2 w = (x * x) * (f32x2)(-0.5f * 1.4426950408889634f);          // -a^2 * log2(e)
  const f32x2 e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const f32x2 h = p * e;                                                    // erfc(|x|/sqrt2)/2
  return (f32x2){fmaf(-a0, h.x, fmaxf(x.x, 0.0f)), fmaf(-a1, h.y, fmaxf(x.y, 0.0f))};
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2((f32x2){x, x}).x; }

// MX-fp4 (e2m1) operands of the mini-tile correction passes (gemm_ht.hip).  A block of lo values (one LayerNorm row, or 64 columns of an
// attention / GELU output row) shares one power-of-two scale chosen from the block's largest |lo|: v = lo * 2^s with 4 <= max|v| < 8,
// i.e. s = 129 - biased_exponent(max|lo|); values in [5, 8) saturate to 6 (measured residual 1.7-2.5 % of the lo variance on Gaussian,
// GELU and heavy-tailed rows, DESIGN.md "Precision").  The MFMA's E8M0 scale byte that undoes 2^s is 127 - s = biased_exponent - 2.
// e2m1 codes 0..7 = {0, .5, 1, 1.5, 2, 3, 4, 6}, bit 3 = sign; element 2j is the low nibble of byte j.
__device__ __forceinline__ uint32_t fp4_scale_byte(float maxlo) {            // 0 for an all-zero block (every product is then 0 anyway)
  const int e = (__float_as_int(maxlo) >> 23) & 0xff;
  return (uint32_t)(e >= 2 ? e - 2 : 0);
}
__device__ __forceinline__ float fp4_scale_mul(float maxlo) {                // 2^s as a float (0 for an all-zero / denormal block: codes become 0)
  const int e = (__float_as_int(maxlo) >> 23) & 0xff;
  return e >= 2 ? __int_as_float((256 - e) << 23) : 0.0f;                    // 2^(129 - e): biased exponent 256 - e in [2, 254]
}
// The same for operands whose LARGEST elements must not be clipped (activation values, weight rounding errors: a saturated outlier is a
// systematic error on exactly the products that dominate the big outputs): 3 < max|v| <= 6, i.e. one binade lower when the mantissa exceeds 1.5.
__device__ __forceinline__ int fp4_nosat_exp(float amax) {                  // biased exponent E with amax * 2^(129 - E) in (3, 6]
  const int b = __float_as_int(amax);
  return ((b >> 23) & 0xff) + (((b & 0x7fffff) > 0x400000) ? 1 : 0);
}
__device__ __forceinline__ uint32_t fp4_scale_byte_nosat(float amax) { const int e = fp4_nosat_exp(amax); return (uint32_t)(e >= 2 ? e - 2 : 0); }
__device__ __forceinline__ float fp4_scale_mul_nosat(float amax) {
  const int e = fp4_nosat_exp(amax);
  return (e >= 2 && e <= 254) ? __int_as_float((256 - e) << 23) : 0.0f;
}
__device__ __forceinline__ uint32_t fp4_code(float v) {                      // v already scaled; round to nearest e2m1, saturating at 6
  const float a = fminf(fabsf(v), 6.0f);
  int k = ((__float_as_int(a) >> 23) & 0xff) - 127;                          // floor(log2 a) for a >= 1
  k = max(0, min(k, 2));
  const float r = rintf(a * __int_as_float((128 - k) << 23));                // a / 2^(k-1): grid step 0.5 below 2, 1 below 4, 2 above
  return (uint32_t)((int)r + 2 * k) | (v < 0.0f ? 8u : 0u);
}
// 4 consecutive columns -> 16 bits, on the hardware converter: v_cvt_scalef32_pk_fp4_f32 DIVIDES its two inputs by the scale operand, rounds to
// nearest even on the e2m1 grid and saturates at 6 (probed: tools/micro/mfma_f4_probe.hip) -- bit for bit what fp4_code does, in one
// instruction per pair (the software form costs ~20 VALU operations per value, which the GELU epilogue and the LayerNorm kernels felt).
__device__ __forceinline__ uint32_t fp4_pack4(float v0, float v1, float v2, float v3, float mul) {
  const float inv = mul > 0.f ? __builtin_amdgcn_rcpf(mul) : 1.0f;          // powers of two: exact
  const float z = mul > 0.f ? 1.0f : 0.0f;                                  // all-zero block (mul = 0): codes 0
  uint32_t w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, v0 * z, v1 * z, inv, 0);
  return __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v2 * z, v3 * z, inv, 1) & 0xffffu;
}

// LayerNorm affine of one element, written with explicit roundings: the LayerNorm kernel and the GEMM epilogue that
// re-derives the normalised residual from (y, mean, rstd) must produce the same bits.
__device__ __forceinline__ float ln_affine(float v, float mean, float rstd, float g, float b) {
  return __fmaf_rn(__fmul_rn(__fsub_rn(v, mean), rstd), g, b);
}

// XCD-aware bijective remap: hardware places block b on XCD b%8; give every XCD a contiguous
// chunk of the logical tile list so neighbouring tiles (sharing an operand panel) share an L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
