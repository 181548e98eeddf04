// Row-wise kernels of the LFQBert trunk: one 64-lane wavefront per token row, no LDS.
//   embed_ln      : bit tokens -> {-1,0,+1} vector -> input_proj + class token + pos_emb -> LayerNorm
//                   (modeling/bert.py:440-454 preprocess_tokens, :482-496)
//   layernorm_rows: the post-norm LayerNorm(eps=1e-12) after each residual add (bert.py:69-70,137-139)
// Both write the fp32 residual stream and its h16 copy (the next GEMM's A operand) in one pass.
#include "mb_kernels.h"

namespace mb {

constexpr int MAXS = 32;  // scalar fallback path: d <= 64 * MAXS = 2048
// NV > 0: vector path, d == 256*NV, row lives in NV float4 registers per lane (fully unrolled).
// NV == 0: generic scalar path for odd widths (tiny test configs).

// e2m1 copy of a row held in registers (lane: columns q * 256 + lane * 4 .. + 3) for the trunk GEMMs' mini-tile passes (GemmArgs.lo): one
// power-of-two scale per 64 columns -- 16 lanes of one q -- chosen from the block's largest element without saturation (3 < max <= 6), its E8M0
// byte stored in the lane-ordered scale array.  A massive-activation channel then costs the resolution of its own 64-column block, not of the row.
template <int NV>
__device__ __forceinline__ void fp4_row_store(const float4* v, int lane, uint8_t* x4row, uint8_t* scales, int nseq, int seq, int tok, int groups) {
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    float am = fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w)));
    am = row16_max(am);
    if ((lane & 15) == 0) scales[fp4_scale_index(q * 4 + (lane >> 4), nseq, seq, tok, groups)] = (uint8_t)fp4_scale_byte_nosat(am);
    *(uint16_t*)(x4row + q * 128 + lane * 2) = (uint16_t)fp4_pack4(v[q].x, v[q].y, v[q].z, v[q].w, fp4_scale_mul_nosat(am));
  }
}
// values (f4.x4) and / or fp16 lo halves (f4.xl4) of row `row` (token row % 257 of sequence row / 257; class-token rows take no part)
template <int NV>
__device__ __forceinline__ void fp4_rows_out(const Fp4Rows& f4, float4* v, int lane, int row, int d) {
  const int seq = row / f4.seq_rows, tok = row - seq * f4.seq_rows, groups = (f4.seq_rows - 1) >> 6;
  if (tok == f4.seq_rows - 1) return;
  if (f4.x4) fp4_row_store<NV>(v, lane, f4.x4 + (size_t)row * 2 * d, f4.x4s, f4.nseq, seq, tok, groups);
  if (f4.xl4) {
#pragma unroll
    for (int q = 0; q < NV; ++q)
      v[q] = make_float4(v[q].x - (float)to_h(v[q].x), v[q].y - (float)to_h(v[q].y), v[q].z - (float)to_h(v[q].z), v[q].w - (float)to_h(v[q].w));
    fp4_row_store<NV>(v, lane, f4.xl4 + (size_t)row * 2 * d, f4.xl4s, f4.nseq, seq, tok, groups);
  }
}

// Two-pass mean / variance of a row held in registers, then affine + store.
template <int NV>
__device__ __forceinline__ void ln_finish(float4* v, const float* sc, int ns, int d, int lane,
                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                          float eps, float* xo, h16* xb, float* st = nullptr, h16* xl = nullptr,
                                          const Fp4Rows* f4 = nullptr, int row = 0) {
  // f4 (optional): e2m1 copies of the row for the mini-tile passes (vector path only)
  // xl (optional): the fp16 lo halves o - fp16(o) of the same row, for the split-activation GEMMs
  constexpr bool VEC = NV > 0;
  constexpr int nv = NV;
  float s = 0.f;
  if (VEC) {
#pragma unroll
    for (int q = 0; q < nv; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
  }
  else     { for (int q = 0; q < ns; ++q) s += sc[q]; }
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
  if (VEC) {
#pragma unroll
    for (int q = 0; q < nv; ++q) {
      const float a = v[q].x - mean, b = v[q].y - mean, c = v[q].z - mean, e = v[q].w - mean;
      ss += (a * a + b * b) + (c * c + e * e);
    }
  } else {
    for (int q = 0; q < ns; ++q) { const int c = lane + q * 64; if (c < d) { const float a = sc[q] - mean; ss += a * a; } }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
  if (st && lane == 0) *(float2*)st = make_float2(mean, rstd);       // for the GEMM epilogue that re-derives these rows
  if (VEC) {
#pragma unroll
    for (int q = 0; q < nv; ++q) {
      const int c = q * 256 + lane * 4;
      const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
      float4 o;
      o.x = ln_affine(v[q].x, mean, rstd, g.x, b.x); o.y = ln_affine(v[q].y, mean, rstd, g.y, b.y);
      o.z = ln_affine(v[q].z, mean, rstd, g.z, b.z); o.w = ln_affine(v[q].w, mean, rstd, g.w, b.w);
      if (xo) *(float4*)(xo + c) = o;
      const h16x4 hi = {to_h(o.x), to_h(o.y), to_h(o.z), to_h(o.w)};
      if (xb) *(h16x4*)(xb + c) = hi;
      if (xl) *(h16x4*)(xl + c) = h16x4{to_h(o.x - (float)hi[0]), to_h(o.y - (float)hi[1]), to_h(o.z - (float)hi[2]), to_h(o.w - (float)hi[3])};
      if (f4) v[q] = o;                                               // keep the normalised row for the e2m1 copies
    }
    if constexpr (VEC) { if (f4) fp4_rows_out<NV>(*f4, v, lane, row, d); }
  } else {
    for (int q = 0; q < ns; ++q) {
      const int c = lane + q * 64;
      if (c < d) {
        const float o = ln_affine(sc[q], mean, rstd, gamma[c], beta[c]);
        if (xo) xo[c] = o;
        if (xb) xb[c] = to_h(o);
        if (xl) xl[c] = to_h(o - (float)to_h(o));
      }
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, float* x_f32,
                                                      h16* x_h16, float* stats, int M, int d, h16* x_lo, Fp4Rows f4) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* yr = y + (size_t)row * d;
  constexpr bool VEC = NV > 0;
  float4 v[VEC ? NV : 1]; float sc[VEC ? 1 : MAXS];
  const int ns = (d + 63) / 64;
  if (VEC) {
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = *(const float4*)(yr + q * 256 + lane * 4);
  }
  else     { for (int q = 0; q < ns; ++q) { const int c = lane + q * 64; sc[q] = c < d ? yr[c] : 0.f; } }
  ln_finish<NV>(v, sc, ns, d, lane, gamma, beta, eps, x_f32 ? x_f32 + (size_t)row * d : nullptr,
                 x_h16 ? x_h16 + (size_t)row * d : nullptr, stats ? stats + (size_t)row * 2 : nullptr,
                 x_lo ? x_lo + (size_t)row * d : nullptr,
                 (f4.x4 || f4.xl4) ? &f4 : nullptr, row);
}

// ---- "CFG pair" LayerNorm (differential classifier-free guidance, DESIGN.md "Precision"): one wave normalises row r of the conditional
// stream and its unconditional twin r + P together and writes  x_h16[r] = fp16(x_c),  x_h16[r + P] = fp16(x_u - x_c)  -- the operands of
// the pair GEMM (gemm_ht.hip): the rounding error of x_c is then common to both streams and cancels in (c - u).  Row statistics of both rows
// go to `stats` (the residual GEMMs re-derive LayerNorm(y) from them).  f4 (optional, mini-tile passes): e2m1 of the conditional row's values
// and / or lo halves with per-64-column scales (fp4_rows_out; the difference rows take no part in those passes).
template <int NV>
__device__ __forceinline__ float2 ln_normalize(float4* v, int d, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int lane) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const float a = v[q].x - mean, b = v[q].y - mean, c = v[q].z - mean, e = v[q].w - mean;
    ss += (a * a + b * b) + (c * c + e * e);
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c = q * 256 + lane * 4;
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    v[q] = make_float4(ln_affine(v[q].x, mean, rstd, g.x, b.x), ln_affine(v[q].y, mean, rstd, g.y, b.y),
                       ln_affine(v[q].z, mean, rstd, g.z, b.z), ln_affine(v[q].w, mean, rstd, g.w, b.w));
  }
  return make_float2(mean, rstd);
}

// store the pair operands of one row pair held in registers (xc, xu = the two fp32 rows)
template <int NV>
__device__ __forceinline__ void pair_store(float4* xc, const float4* xu, int lane, h16* oc, h16* ou, const Fp4Rows& f4, int row, int d) {
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c = q * 256 + lane * 4;
    *(h16x4*)(oc + c) = h16x4{to_h(xc[q].x), to_h(xc[q].y), to_h(xc[q].z), to_h(xc[q].w)};
    *(h16x4*)(ou + c) = h16x4{to_h(xu[q].x - xc[q].x), to_h(xu[q].y - xc[q].y), to_h(xu[q].z - xc[q].z), to_h(xu[q].w - xc[q].w)};
  }
  if (f4.x4 || f4.xl4) fp4_rows_out<NV>(f4, xc, lane, row, d);      // (last use of xc: the lo halves overwrite it)
}

template <int NV>
__global__ __launch_bounds__(256) void ln_pair_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      h16* __restrict__ x_h16, float* __restrict__ stats, int P, int d, Fp4Rows f4) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P) return;
  float4 vc[NV], vu[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) { vc[q] = *(const float4*)(y + (size_t)row * d + q * 256 + lane * 4); vu[q] = *(const float4*)(y + (size_t)(row + P) * d + q * 256 + lane * 4); }
  const float2 sc = ln_normalize<NV>(vc, d, gamma, beta, eps, lane), su = ln_normalize<NV>(vu, d, gamma, beta, eps, lane);
  if (stats && lane == 0) { *(float2*)(stats + 2 * (size_t)row) = sc; *(float2*)(stats + 2 * (size_t)(row + P)) = su; }
  pair_store<NV>(vc, vu, lane, x_h16 + (size_t)row * d, x_h16 + (size_t)(row + P) * d, f4, row, d);
}

// pair operands from fp32 rows that already exist (the embedding LayerNorm's output, the first residual): x32 [2P, d] -> x_h16 as above
template <int NV>
__global__ __launch_bounds__(256) void pairify_kernel(const float* __restrict__ x32, h16* __restrict__ x_h16, int P, int d, Fp4Rows f4) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P) return;
  float4 vc[NV], vu[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) { vc[q] = *(const float4*)(x32 + (size_t)row * d + q * 256 + lane * 4); vu[q] = *(const float4*)(x32 + (size_t)(row + P) * d + q * 256 + lane * 4); }
  pair_store<NV>(vc, vu, lane, x_h16 + (size_t)row * d, x_h16 + (size_t)(row + P) * d, f4, row, d);
}

int layernorm_pair(hipStream_t s, const float* y, const float* gamma, const float* beta, float eps, h16* x_h16, float* stats, int P, int d,
                   const Fp4Rows& f4) {
  dim3 grid((P + 3) / 4), block(256);
  if (d == 1024) hipLaunchKernelGGL(ln_pair_kernel<4>, grid, block, 0, s, y, gamma, beta, eps, x_h16, stats, P, d, f4);
  else if (d == 768) hipLaunchKernelGGL(ln_pair_kernel<3>, grid, block, 0, s, y, gamma, beta, eps, x_h16, stats, P, d, f4);
  else return -1;
  return 0;
}
int pairify_rows(hipStream_t s, const float* x32, h16* x_h16, int P, int d, const Fp4Rows& f4) {
  dim3 grid((P + 3) / 4), block(256);
  if (d == 1024) hipLaunchKernelGGL(pairify_kernel<4>, grid, block, 0, s, x32, x_h16, P, d, f4);
  else if (d == 768) hipLaunchKernelGGL(pairify_kernel<3>, grid, block, 0, s, x32, x_h16, P, d, f4);
  else return -1;
  return 0;
}

void layernorm_rows(hipStream_t s, const float* y, const float* gamma, const float* beta, float eps,
                    float* x_f32, h16* x_h16, float* stats, int M, int d, h16* x_lo, const Fp4Rows& f4) {
  dim3 grid((M + 3) / 4), block(256);      // f4 (e2m1 values / lo halves): vector path only (d = 768 / 1024; mb_gen_create restricts the modes to those)
  if (d == 1024) hipLaunchKernelGGL(ln_rows_kernel<4>, grid, block, 0, s, y, gamma, beta, eps, x_f32, x_h16, stats, M, d, x_lo, f4);
  else if (d == 768) hipLaunchKernelGGL(ln_rows_kernel<3>, grid, block, 0, s, y, gamma, beta, eps, x_f32, x_h16, stats, M, d, x_lo, f4);
  else hipLaunchKernelGGL(ln_rows_kernel<0>, grid, block, 0, s, y, gamma, beta, eps, x_f32, x_h16, stats, M, d, x_lo, Fp4Rows{});
}

// One wave per (sequence, row).  Rows 0..seq-1 are image tokens, row seq is the class token (LAST,
// bert.py:489).  A token contributes sum_j sign_j * W[:, g*gbits + j] where sign is +-1 from bit j
// of group g's index (LSB first) and 0 when the group carries the mask token (index == 2^gbits).
// w_in is held TRANSPOSED ([K][d], repacked at load) so that the K loads of a lane are independent
// float4 reads of 4 neighbouring features; the sign vector is two wave-uniform bit masks.
constexpr int MAXBITS = 24;

template <int NV>
__global__ __launch_bounds__(256) void embed_ln_kernel(EmbedArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int N = a.seq + 1;
  if (row >= a.nb * N) return;
  const int sq = row / N, t = row - sq * N;
  const int d = a.d, K = a.m * a.gbits;
  uint32_t plus = 0, live = 0;       // bit j of `plus` set -> +1; bit j of `live` clear -> 0
  const float* cls = nullptr;
  const float* trow[8] = {};         // Bert: the embedding-table row of each group's token (index 2^gbits = mask token)
  if (t < a.seq) {
    for (int g = 0; g < a.m; ++g) {
      const int64_t idx = a.tokens[((size_t)sq * a.seq + t) * a.m + g];
      if (a.tables && g < 8) {
        const int64_t nrow = ((int64_t)1 << a.gbits) + 1;
        const int64_t ix = idx < 0 ? 0 : (idx >= nrow ? nrow - 1 : idx);            // never read outside the table (host validates)
        trow[g] = a.tables + ((size_t)g * nrow + (size_t)ix) * d;
      }
      const uint32_t gm = (1u << a.gbits) - 1u;
      if (idx != ((int64_t)1 << a.gbits)) { live |= gm << (g * a.gbits); plus |= ((uint32_t)idx & gm) << (g * a.gbits); }
    }
  } else {
    int64_t lab = a.labels[sq];
    if (a.drop && a.drop[sq]) lab = a.nclass;                       // bert.py:482-484 (not in place)
    lab = lab < 0 ? 0 : (lab > a.nclass ? a.nclass : lab);          // never read outside class_emb (host validates)
    cls = a.class_emb + (size_t)lab * d;
  }
  const float* pos = a.pos + (size_t)t * d;
  constexpr bool VEC = NV > 0;
  float4 v[VEC ? NV : 1]; float sc[VEC ? 1 : MAXS];
  const int ns = (d + 63) / 64;
  if (VEC) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int c = q * 256 + lane * 4;
      float4 e = cls ? *(const float4*)(cls + c) : (a.tables ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(a.b_in + c));
      if (!cls && a.tables) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
          if (g < a.m) { const float4 w = *(const float4*)(trow[g] + c); e.x += w.x; e.y += w.y; e.z += w.z; e.w += w.w; }
      } else if (!cls) {
#pragma unroll
        for (int j = 0; j < MAXBITS; ++j) {
          if (j < K) {
            const float4 w = *(const float4*)(a.w_in + (size_t)j * d + c);
            const float sj = ((live >> j) & 1u) ? (((plus >> j) & 1u) ? 1.f : -1.f) : 0.f;
            e.x = fmaf(sj, w.x, e.x); e.y = fmaf(sj, w.y, e.y); e.z = fmaf(sj, w.z, e.z); e.w = fmaf(sj, w.w, e.w);
          }
        }
      }
      const float4 p = *(const float4*)(pos + c);
      v[q] = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    }
  } else {
    for (int q = 0; q < ns; ++q) {
      const int c = lane + q * 64;
      float e = 0.f;
      if (c < d) {
        if (cls) e = cls[c];
        else if (a.tables) {
          for (int g = 0; g < a.m && g < 8; ++g) e += trow[g][c];
        } else {
          e = a.b_in[c];
          for (int j = 0; j < K; ++j) {
            const float sj = ((live >> j) & 1u) ? (((plus >> j) & 1u) ? 1.f : -1.f) : 0.f;
            e = fmaf(sj, a.w_in[(size_t)j * d + c], e);
          }
        }
        e += pos[c];
      }
      sc[q] = e;
    }
  }
  ln_finish<NV>(v, sc, ns, d, lane, a.gamma, a.beta, 1e-12f, a.x_f32 + (size_t)row * d, a.x_h16 + (size_t)row * d, nullptr,
                 a.x_lo ? a.x_lo + (size_t)row * d : nullptr,
                 (a.f4.x4 || a.f4.xl4) ? &a.f4 : nullptr, row);
}

// The same for the guided (CFG pair) forward: sequences [0, nb) conditional, their label-dropped twins nb sequences further down.  A twin has the
// same tokens as its conditional sequence, so every image-token row is embedded and normalised ONCE and written twice (difference operand 0); only the
// class-token row differs (class_emb[nclass] instead of class_emb[label]).  Writes what embed_ln over [cond | twins] + pairify_rows wrote, bit for
// bit: x_f32 (both rows), x_h16 = fp16(x_c) | fp16(x_u - x_c), and optionally the e2m1 values + block scales of the conditional rows.
template <int NV>
__global__ __launch_bounds__(256) void embed_pair_kernel(EmbedArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int N = a.seq + 1, P = a.nb * N;
  if (row >= P) return;
  const int sq = row / N, t = row - sq * N;
  const int d = a.d, K = a.m * a.gbits;
  uint32_t plus = 0, live = 0;
  const float* cls = nullptr;
  if (t < a.seq) {
    for (int g = 0; g < a.m; ++g) {
      const int64_t idx = a.tokens[((size_t)sq * a.seq + t) * a.m + g];
      const uint32_t gm = (1u << a.gbits) - 1u;
      if (idx != ((int64_t)1 << a.gbits)) { live |= gm << (g * a.gbits); plus |= ((uint32_t)idx & gm) << (g * a.gbits); }
    }
  } else {
    int64_t lab = a.labels[sq];
    lab = lab < 0 ? 0 : (lab > a.nclass ? a.nclass : lab);
    cls = a.class_emb + (size_t)lab * d;
  }
  const float* pos = a.pos + (size_t)t * d;
  float4 vc[NV], vu[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c = q * 256 + lane * 4;
    float4 e = cls ? *(const float4*)(cls + c) : *(const float4*)(a.b_in + c);
    if (!cls) {
#pragma unroll
      for (int j = 0; j < MAXBITS; ++j) {
        if (j < K) {
          const float4 w = *(const float4*)(a.w_in + (size_t)j * d + c);
          const float sj = ((live >> j) & 1u) ? (((plus >> j) & 1u) ? 1.f : -1.f) : 0.f;
          e.x = fmaf(sj, w.x, e.x); e.y = fmaf(sj, w.y, e.y); e.z = fmaf(sj, w.z, e.z); e.w = fmaf(sj, w.w, e.w);
        }
      }
    }
    const float4 p = *(const float4*)(pos + c);
    vc[q] = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    if (cls) {                                            // the twin's class token: the "dropped label" embedding (bert.py:482-484)
      const float4 eu = *(const float4*)(a.class_emb + (size_t)a.nclass * d + c);
      vu[q] = make_float4(eu.x + p.x, eu.y + p.y, eu.z + p.z, eu.w + p.w);
    }
  }
  ln_normalize<NV>(vc, d, a.gamma, a.beta, 1e-12f, lane);
  if (cls) ln_normalize<NV>(vu, d, a.gamma, a.beta, 1e-12f, lane);
  else {
#pragma unroll
    for (int q = 0; q < NV; ++q) vu[q] = vc[q];
  }
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c = q * 256 + lane * 4;
    *(float4*)(a.x_f32 + (size_t)row * d + c) = vc[q];
    *(float4*)(a.x_f32 + (size_t)(row + P) * d + c) = vu[q];
  }
  pair_store<NV>(vc, vu, lane, a.x_h16 + (size_t)row * d, a.x_h16 + (size_t)(row + P) * d, a.f4, row, d);
}

int embed_pair(hipStream_t s, const EmbedArgs& a) {
  if (a.tables || a.m * a.gbits > MAXBITS) return -1;     // (the embedding-table Bert takes the two-kernel path)
  const int rows = a.nb * (a.seq + 1);
  dim3 grid((rows + 3) / 4), block(256);
  if (a.d == 1024) hipLaunchKernelGGL(embed_pair_kernel<4>, grid, block, 0, s, a);
  else if (a.d == 768) hipLaunchKernelGGL(embed_pair_kernel<3>, grid, block, 0, s, a);
  else return -1;
  return 0;
}

void embed_ln(hipStream_t s, const EmbedArgs& a) {
  const int rows = a.nb * (a.seq + 1);
  dim3 grid((rows + 3) / 4), block(256);
  if (a.d == 1024) hipLaunchKernelGGL(embed_ln_kernel<4>, grid, block, 0, s, a);
  else if (a.d == 768) hipLaunchKernelGGL(embed_ln_kernel<3>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(embed_ln_kernel<0>, grid, block, 0, s, a);
}

__global__ void transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * cols) { const int r = i / cols, c = i - r * cols; dst[(size_t)c * rows + r] = src[i]; }
}
void transpose_f32(hipStream_t s, const float* src, float* dst, int rows, int cols) {
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, s, src, dst, rows, cols);
}

}  // namespace mb
