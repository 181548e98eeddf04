// h16 MFMA GEMM for the LFQBert trunk: out[M,N] = A[M,K] . W[N,K]^T + bias, fused epilogues.
//
// Replaces the nn.Linear / packed in_proj calls of modeling/bert.py:27-33 (FFN), :84,137 (MHA
// projections), :411-417 (head).  Both operands are K-contiguous ("TN"), which is exactly the MFMA
// fragment order, so tiles go HBM -> LDS by 16-byte LDS-DMA with no register round trip.
//
// Tile: 128(M) x 128(N) x 64(K), 256 threads = 4 waves as 2x2, each wave 64x64 = 4x4 MFMA
// 16x16x32 tiles.  The MFMA A operand is the WEIGHT tile and the B operand the ACTIVATION tile, so
// that a lane ends up holding 4 consecutive output features of one token row
// (D[n=(lane>>4)*4+r][m=lane&15]) -> 8/16-byte row-major stores and float4 bias/residual loads.
// LDS rows are 128 B (64 h16); the 16-byte slot index is XOR-swizzled with (row>>1)&7 -- applied
// on the per-lane SOURCE address (the DMA destination is lane-linear) and again on the fragment
// read -- which makes every ds_read_b128 lane group hit 16 distinct slots.
// Double-buffered: the DMA for K-tile t+1 is in flight while tile t is multiplied.
#include <cstdio>
#include <cstdlib>

#include "mb_kernels.h"

namespace mb {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;      // 16 KiB per operand tile

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // [buf][X|W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (a.N + BN - 1) / BN;
  const int tiles_m = (a.M + BM - 1) / BM;
  const int L = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (L / tiles_n) * BM, n0 = (L % tiles_n) * BN;
  const int K = a.K, KA = a.kw ? a.kw : a.K, nka = KA / BK;
  const int KW = a.kw ? a.kw : a.K, nkw = KW / BK;   // split activations: W has kw columns and is swept twice, A2 takes over from A

  // ---- staging: wave w owns rows [32w, 32w+32) of both tiles; 4 DMA instructions of 8 rows each
  const h16* xsrc[4];
  const h16* wsrc[4];
  const ptrdiff_t a2 = a.A2 ? a.A2 - a.A : 0;       // second sweep of A: the lo halves (split activations)
  const ptrdiff_t w2 = a.W2 ? a.W2 - a.W : 0;       // three sweeps (GemmArgs.W2): the third pairs A with the weights' lo halves
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wave * 32 + j * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    const int mr = min(m0 + row, a.M - 1), nr = min(n0 + row, a.N - 1);
    xsrc[j] = a.A + (size_t)mr * KA + slot * 8;
    wsrc[j] = a.W + (size_t)nr * KW + slot * 8;
  }
  auto stage = [&](int t, int buf) {
    const int ta = t < nka ? t : t - nka;
    char* xb = smem + buf * 2 * TILE_BYTES + wave * 32 * 128;
    char* wb = xb + TILE_BYTES;
    if (a.W2) {                                    // sweeps (A, W), (A2, W), (A, W2)
      const int seg = t / nkw, tt = t - seg * nkw;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        MB_GLDS16(xsrc[j] + (seg == 1 ? a2 : 0) + tt * BK, xb + j * 8 * 128);
        MB_GLDS16(wsrc[j] + (seg == 2 ? w2 : 0) + tt * BK, wb + j * 8 * 128);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      MB_GLDS16(xsrc[j] + (t < nka ? 0 : a2) + ta * BK, xb + j * 8 * 128);
      MB_GLDS16(wsrc[j] + (t < nkw ? t : t - nkw) * BK, wb + j * 8 * 128);
    }
  };

  // ---- fragment read offsets (independent of the 16-row tile index: (row>>1)&7 == (lane&15)>>1)
  int foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
    foff[kk] = (lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ ((lane & 15) >> 1)) * 16);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  stage(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // tile t landed for every wave; buf (t+1)&1 is free
    if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
    const char* xb = smem + (t & 1) * 2 * TILE_BYTES + wm * 64 * 128;
    const char* wb = smem + (t & 1) * 2 * TILE_BYTES + TILE_BYTES + wn * 64 * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h16x8 wf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *(const h16x8*)(wb + i * 16 * 128 + foff[kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *(const h16x8*)(xb + j * 16 * 128 + foff[kk]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = MB_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
    }
  }

  // ---- epilogue: lane holds out[m][n..n+3], m = ..+(lane&15), n = ..+(lane>>4)*4
  const float sc = a.scale ? *a.scale : 1.0f;
  bool satb = false;                                 // fp16 epilogues: this lane stores a value outside fp16's range, or a NaN (GemmArgs.sat)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + (lane & 15);
    if (m >= a.M) continue;
    size_t orow = (size_t)m;
    const float* brow = a.bias;
    if (EPI == EPI_LOGITS_F32) {
      const int sq = m / a.period, pos = m - sq * a.period;
      if (pos == a.period - 1) continue;           // class-token row is not a prediction (bert.py:503)
      orow = (size_t)sq * (a.period - 1) + pos;
      if (a.bias_per_pos) brow = a.bias + (size_t)pos * a.N;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      const float4 b = *(const float4*)(brow + n);
      float v0 = fmaf(acc[i][j][0], sc, b.x), v1 = fmaf(acc[i][j][1], sc, b.y), v2 = fmaf(acc[i][j][2], sc, b.z), v3 = fmaf(acc[i][j][3], sc, b.w);
      if (EPI == EPI_RES_F32) {
        const float4 r = *(const float4*)(a.residual + (size_t)m * a.N + n);
        if (a.ln_stats) {                            // residual = LayerNorm(previous y), re-derived (GemmArgs)
          const float2 st = *(const float2*)(a.ln_stats + 2 * (size_t)m);
          const float4 g = *(const float4*)(a.ln_g + n), be = *(const float4*)(a.ln_b + n);
          v0 += ln_affine(r.x, st.x, st.y, g.x, be.x); v1 += ln_affine(r.y, st.x, st.y, g.y, be.y);
          v2 += ln_affine(r.z, st.x, st.y, g.z, be.z); v3 += ln_affine(r.w, st.x, st.y, g.w, be.w);
        } else { v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w; }
      }
      if (EPI == EPI_GELU_H16 || EPI == EPI_GELU_F32) {
        const f32x2 g01 = gelu_erf2((f32x2){v0, v1}), g23 = gelu_erf2((f32x2){v2, v3});
        v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y;
      }
      if (EPI == EPI_H16 || EPI == EPI_GELU_H16) {
        satb |= !(fabsf(v0) <= MB_H16_MAX) | !(fabsf(v1) <= MB_H16_MAX) | !(fabsf(v2) <= MB_H16_MAX) | !(fabsf(v3) <= MB_H16_MAX);   // (fmaxf would drop a NaN)
        h16x4 o = {to_h(v0), to_h(v1), to_h(v2), to_h(v3)};
        *(h16x4*)(a.out_h16 + orow * a.N + n) = o;
      } else {
        *(float4*)(a.out_f32 + orow * a.N + n) = make_float4(v0, v1, v2, v3);
      }
    }
  }
  if ((EPI == EPI_H16 || EPI == EPI_GELU_H16) && a.sat && satb) atomicAdd(a.sat, 1u);
}

int gemm_tn(hipStream_t s, GemmEpi epi, const GemmArgs& a, int variant) {
  // variant: 0 auto; -1 this 128x128 kernel; 6 / 8 the half-tile kernel with that MT; 257 its sequence-aligned tiles.
  // Returns 0, or -1 when the request needs the half-tile kernel (pair tiles, mini-tile passes) and the shape is outside it
  // (the caller reports it; nothing is launched).
  if (a.W2 && (!a.A2 || !a.kw || a.K != 3 * a.kw || !a.scale)) return -1;
  if (variant >= 0 && !a.W2 && gemm_ht_supported(epi, a)) {
    if (variant % 1000 == 257 && a.M % (a.seq_rows ? a.seq_rows : 257)) variant = variant - variant % 1000;
    gemm_ht(s, epi, a, variant);
    return 0;
  }
  if (a.pair_rows || a.nlo) return -1;   // (mini-tile passes exist in the half-tile kernel only: never dropped silently)
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  dim3 grid(tiles), block(256);
  switch (epi) {
    case EPI_H16: hipLaunchKernelGGL(gemm_tn_kernel<EPI_H16>, grid, block, 0, s, a); break;
    case EPI_GELU_H16: hipLaunchKernelGGL(gemm_tn_kernel<EPI_GELU_H16>, grid, block, 0, s, a); break;
    case EPI_RES_F32: hipLaunchKernelGGL(gemm_tn_kernel<EPI_RES_F32>, grid, block, 0, s, a); break;
    case EPI_GELU_F32: hipLaunchKernelGGL(gemm_tn_kernel<EPI_GELU_F32>, grid, block, 0, s, a); break;
    case EPI_LOGITS_F32: hipLaunchKernelGGL(gemm_tn_kernel<EPI_LOGITS_F32>, grid, block, 0, s, a); break;
  }
  return 0;
}

__global__ void cast_kernel(const float* __restrict__ src, h16* __restrict__ dst, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    const float4 v = *(const float4*)(src + i);
    *(h16x4*)(dst + i) = h16x4{to_h(v.x), to_h(v.y), to_h(v.z), to_h(v.w)};
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t j = n & ~(size_t)3; j < n; ++j) dst[j] = to_h(src[j]);
}
void cast_f32_to_h16(hipStream_t s, const float* src, h16* dst, size_t n) {
  const int blocks = (int)min((size_t)2048, (n / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(cast_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n);
}


// ---- e2m1 copies of a weight for the mini-tile passes (gemm_ht.hip): one workgroup per weight row, one power-of-two scale per (row, 128 columns) =
// per (row, mini-tile): thread t holds columns [4 t, 4 t + 4) of a 1024-column sweep, so a 128-column block is the 32 lanes of a half wave.
__device__ __forceinline__ float half_wave_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// VALUES: e2m1(fp16(W) * 2^r), r = the best of three candidates around 2.2 / rms of the block (measured optimum for Gaussian blocks: quantisation error
// ~1.5 % of the block's energy) by summed squared error -- the largest elements may saturate when that serves the rest.
__global__ __launch_bounds__(256) void w4_kernel(const float* __restrict__ W, uint8_t* __restrict__ out, int N, int K, uint8_t* __restrict__ scale_out) {
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* w = W + (size_t)n * K;
  const float F4V[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  for (int k = tid * 4; k < K; k += 1024) {
    const float4 v4 = *(const float4*)(w + k);
    const float v[4] = {(float)(h16)v4.x, (float)(h16)v4.y, (float)(h16)v4.z, (float)(h16)v4.w};   // the fp16 engine multiplies by fp16(W): quantise THAT value
    const float rms = sqrtf(half_wave_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) / 128.0f);
    int r0 = 0;
    if (rms > 0.f && rms < 3.0e38f) r0 = (int)rintf(log2f(2.2f / rms));
    r0 = max(-100, min(100, r0));
    float err[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float m = ldexpf(1.0f, r0 - 1 + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float q = F4V[fp4_code(v[e] * m) & 7];
        const float d = fabsf(v[e]) * m - q;
        err[c] += d * d / (m * m);
      }
      err[c] = half_wave_sum(err[c]);
    }
    const int best = (err[1] <= err[0] && err[1] <= err[2]) ? 1 : (err[0] <= err[2] ? 0 : 2);
    const int r = r0 - 1 + best;
    if ((tid & 31) == 0) scale_out[w4_scale_index(n, k >> 7, K)] = (uint8_t)max(0, min(254, 127 - r));
    *(uint16_t*)(out + w4_packed_offset(n, k, K)) = (uint16_t)fp4_pack4(v[0], v[1], v[2], v[3], ldexpf(1.0f, r));
  }
}

// fp16 ROUNDING ERRORS: e2m1((W - fp16(W)) * 2^r), r from the block's largest |error| without saturation (a clipped error is a systematic one on exactly
// the products that dominate).  Per (row, 128 columns) since round 6: an error is +-ulp(W) / 2, so on heavy-tailed weights one large weight per row used to
// set the scale of all its columns (9 % of the error energy left against 3 % on Gaussian rows, profiles/r05_parity.md; emulated gain of the block scales
// on the sampled logits' top-2 gap: -10 % by themselves, -30 % next to the activation-lo sets, tests/diag/error_budget.py EB_STUDY=r6).
__global__ __launch_bounds__(256) void w4lo_kernel(const float* __restrict__ W, uint8_t* __restrict__ out, int N, int K, uint8_t* __restrict__ scale_out) {
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* w = W + (size_t)n * K;
  for (int k = tid * 4; k < K; k += 1024) {
    const float4 v = *(const float4*)(w + k);
    const float e0 = v.x - (float)(h16)v.x, e1 = v.y - (float)(h16)v.y, e2 = v.z - (float)(h16)v.z, e3 = v.w - (float)(h16)v.w;
    const float mx = half_wave_max(fmaxf(fmaxf(fabsf(e0), fabsf(e1)), fmaxf(fabsf(e2), fabsf(e3))));
    if ((tid & 31) == 0) scale_out[w4_scale_index(n, k >> 7, K)] = (uint8_t)fp4_scale_byte_nosat(mx);
    *(uint16_t*)(out + w4_packed_offset(n, k, K)) = (uint16_t)fp4_pack4(e0, e1, e2, e3, fp4_scale_mul_nosat(mx));
  }
}

// ---- split-weight repack ----------------------------------------------------------------------
__global__ void absmax_kernel(const float* __restrict__ src, size_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(src[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns
}
__global__ void split_kernel(const float* __restrict__ src, h16* __restrict__ dst, int N, int K, const unsigned* __restrict__ amax,
                             float* __restrict__ scale_out, h16* __restrict__ lo_plane) {
  const float mx = __uint_as_float(*amax);
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);                   // mx = f * 2^e, f in [0.5, 1)
  const int S = mx > 0.f ? 15 - e : 0;                                   // mx * 2^S in [2^14, 2^15): below the fp16 maximum
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = ldexpf(1.0f, -S);
  const size_t n = (size_t)N * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float w = ldexpf(src[i], S);
    const h16 hi = to_h(w);
    dst[i] = hi; lo_plane[i] = to_h(w - (float)hi);                               // two [N,K] planes
  }
}
void split_f32_to_h16_planes(hipStream_t s, const float* src, h16* hi, h16* lo, int N, int K, float* scale_out, unsigned* tmp) {
  if (!hi || !lo) abort();                              // both planes are written unconditionally (callers own both buffers)
  const size_t n = (size_t)N * K;
  const int blocks = (int)min((size_t)2048, (n + 255) / 256);
  (void)hipMemsetAsync(tmp, 0, sizeof(unsigned), s);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, src, n, tmp);
  hipLaunchKernelGGL(split_kernel, dim3(blocks), dim3(256), 0, s, src, hi, N, K, tmp, scale_out, lo);
}
void w4_from_f32(hipStream_t s, const float* src, uint8_t* dst4, int N, int K, uint8_t* scale_out) {
  (void)hipMemsetAsync(dst4, 0, (size_t)N * K / 2, s);
  hipLaunchKernelGGL(w4_kernel, dim3(N), dim3(256), 0, s, src, dst4, N, K, scale_out);
}
void w4lo_from_f32(hipStream_t s, const float* src, uint8_t* dst4, int N, int K, uint8_t* scale_out) {
  (void)hipMemsetAsync(dst4, 0, (size_t)N * K / 2, s);
  hipLaunchKernelGGL(w4lo_kernel, dim3(N), dim3(256), 0, s, src, dst4, N, K, scale_out);
}
}  // namespace mb
