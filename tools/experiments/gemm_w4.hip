// "Four-wave" fp16 MFMA GEMM for the trunk projections -- EXPERIMENTAL sibling of gemm_ht.hip (mb_gemm variant 4; the engine
// does not use it).
//
//   out[M,N] = A[M,K] . W[N,K]^T + bias (+ epilogue); tile 256 x 256 x 64, 256 threads = 4 waves as 2(M) x 2(N),
//   ONE wave per SIMD, each owning a 128 x 128 output block = 8 x 8 MFMA 16x16x32 tiles = 256 accumulators, held in AGPRs
//   by inline-asm MFMAs (left to itself the register allocator rotated them through VGPR copies: 72 v_accvgpr moves per
//   K-tile), both fragment sets of a K-tile double-buffered in VGPRs, memory instructions hand-interleaved with the MFMA
//   groups, two workgroup barriers per K-tile instead of gemm_ht's eight, 128 KiB of LDS fragment reads instead of 240 KiB.
//
// Measured (tools/w4_check.py, M = 32768; in-kernel clock probe): without the operand DMA the K loop takes 2596 clocks per
// K-tile at 2.03 GHz = 1.26 us (gemm_ht: 1.35 us) -- 83 % matrix-pipe duty from one wave per SIMD.  With the DMA it takes
// 1.74-1.89 us against gemm_ht's 1.61 us: the L2 -> LDS path needs ~1.25 us per 64 KiB K-tile, and two whole-K-tile
// buffers give a refill at most one K-tile of lead (a buffer is free only after every wave has read its second fragment
// half), whereas gemm_ht recycles half-tiles and keeps 1.5-2 K-tiles in flight.  Results are bit-identical to gemm_ht.
// Kept as the starting point for a version with finer-grained buffer recycling (QKV 798 vs 1097 TFLOP/s today).
#include <algorithm>

#include "mb_kernels.h"

namespace mb {

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int T_BYTES = 256 * 128;             // one operand tile: 256 rows x 64 halfs (128-byte rows)
  constexpr int PAR_BYTES = 2 * T_BYTES;         // [A | W]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int K = a.K, nk = K / 64;
  const int KA = a.ka ? a.ka : K, nka = KA / 64;
  const int ntiles = tiles_m * tiles_n;

  // tile of this workgroup: XCD-contiguous chunks, 8 x tiles_n super-rows inside (as gemm_ht)
  const int L = xcd_remap(blockIdx.x, ntiles);
  const int sr = L / (8 * tiles_n);
  const int rows_sr = min(8, tiles_m - sr * 8);
  const int rem = L - sr * 8 * tiles_n;
  const int tn = rem / rows_sr, tm = sr * 8 + (rem - tn * rows_sr);
  const int m0 = tm * 256, n0 = tn * 256;

  // DMA plan: wave w stages rows [64w, 64w+64) of both tiles, 8 instructions of 8 rows each
  uint32_t offA[8], offB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = wave * 64 + j * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    offA[j] = (uint32_t)min(m0 + row, a.M - 1) * (uint32_t)KA + slot * 8;
    offB[j] = (uint32_t)min(n0 + row, a.N - 1) * (uint32_t)K + slot * 8;
  }
  auto dma = [&](int t, int par) {
    char* base = smem + par * PAR_BYTES + wave * 64 * 128;
    const int ta = t < nka ? t : t - nka;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      MB_GLDS16(a.A + offA[j] + ta * 64, base + j * 8 * 128);
      MB_GLDS16(a.W + offB[j] + t * 64, base + T_BYTES + j * 8 * 128);
    }
  };

  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = l15 * 128 + (((ks * 4 + g) ^ (l15 >> 1)) * 16);
  const int xbase = wm * 128 * 128, wbase = T_BYTES + wn * 128 * 128;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  h16x8 xa[2][8], wb[2][8];     // [register set][m-tile / n-tile]
  // fragment q of a (K-tile, ks) half: q < 8 activation m-tile q, else weight n-tile q-8
  auto load_frag = [&](int set, const char* par, int ks, int q) {
    if (q < 8) xa[set][q] = *(const h16x8*)(par + xbase + q * 16 * 128 + foff[ks]);
    else wb[set][q - 8] = *(const h16x8*)(par + wbase + (q - 8) * 16 * 128 + foff[ks]);
  };
  // MFMA group q of a half: 4 MFMAs (n-tile q/2, m-tiles 4*(q%2) .. +3)
  auto mma4 = [&](int set, int q) {
    const int i = q >> 1, j0 = (q & 1) * 4;
#pragma unroll
    for (int j = j0; j < j0 + 4; ++j)   // in-place accumulate in AGPRs, spelled out: left to itself the register allocator
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wb[set][i]), "v"(xa[set][j]));   // rotates the 256 accumulators through VGPR copies
  };
  auto dma1 = [&](int t, int par, int j) {             // instruction pair j (A rows + W rows) of this wave's share
    char* base = smem + par * PAR_BYTES + wave * 64 * 128;
    const int ta = t < nka ? t : t - nka;
    MB_GLDS16(a.A + offA[j] + ta * 64, base + j * 8 * 128);
    MB_GLDS16(a.W + offB[j] + t * 64, base + T_BYTES + j * 8 * 128);
  };
#define MB_FENCE() __builtin_amdgcn_sched_barrier(0)

  dma(0, 0);
  if (nk > 1) {
    dma(1, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // K-tile 0 landed; K-tile 1 may still fly
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int q = 0; q < 16; ++q) load_frag(0, smem, 0, q);

  // One K-tile = two halves of 64 MFMAs (16 groups of 4) on alternating fragment sets, hand-interleaved with the memory
  // instructions so that none of them is issued while the matrix pipe is idle:
  //   half 1 (set 0):  groups 0-7  each followed by two ds_reads of set 1 (second half of this K-tile);
  //                    after group 11: lgkmcnt(0) + barrier A  -> this parity is free for everyone;
  //                    groups 12-15 each followed by two DMA instruction pairs of K-tile t+2 (lead 1.1 K-tiles);
  //   half 2 (set 1):  after group 1: vmcnt(16) + barrier B -> K-tile t+1 visible;
  //                    groups 2-9 each followed by two ds_reads of set 0 (first half of K-tile t+1).
  for (int t = 0; t < nk; ++t) {
    const char* par = smem + (t & 1) * PAR_BYTES;
    const char* nxt = smem + ((t + 1) & 1) * PAR_BYTES;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      mma4(0, q);
      if (q < 8) { load_frag(1, par, 1, 2 * q); load_frag(1, par, 1, 2 * q + 1); }
      if (q == 11) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // (A)
      }
      if (q >= 12 && more2) { dma1(t + 2, t & 1, 2 * (q - 12)); dma1(t + 2, t & 1, 2 * (q - 12) + 1); }
      MB_FENCE();
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      mma4(1, q);
      if (q == 1 && more1) {
        if (more2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // K-tile t+1 landed (t+2's 16 may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // (B)
      }
      if (q >= 2 && q < 10 && more1) { load_frag(0, nxt, 0, 2 * (q - 2)); load_frag(0, nxt, 0, 2 * (q - 2) + 1); }
      MB_FENCE();
    }
  }
#undef MB_FENCE
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs retire before the accumulators are read back (asm MFMAs are outside the hazard recogniser)

  // ---- epilogue: lane holds out[m = ..+l15][n = ..+g*4 .. +3] for 8 x 8 (m-tile, n-tile) pairs
  const float osc = a.scale ? *a.scale : 1.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int m = m0 + wm * 128 + j * 16 + l15;
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + wn * 128 + i * 16 + g * 4;
      const float4 b = *(const float4*)(a.bias + n);
      float v0 = fmaf(acc[i][j][0], osc, b.x), v1 = fmaf(acc[i][j][1], osc, b.y), v2 = fmaf(acc[i][j][2], osc, b.z), v3 = fmaf(acc[i][j][3], osc, b.w);
      if (EPI == EPI_RES_F32) {
        const float4 r = *(const float4*)(a.residual + (size_t)m * a.N + n);
        if (a.ln_stats) {
          const float2 st = *(const float2*)(a.ln_stats + 2 * (size_t)m);
          const float4 gm = *(const float4*)(a.ln_g + n), be = *(const float4*)(a.ln_b + n);
          v0 += ln_affine(r.x, st.x, st.y, gm.x, be.x); v1 += ln_affine(r.y, st.x, st.y, gm.y, be.y);
          v2 += ln_affine(r.z, st.x, st.y, gm.z, be.z); v3 += ln_affine(r.w, st.x, st.y, gm.w, be.w);
        } else { v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w; }
      }
      if (EPI == EPI_GELU_H16 || EPI == EPI_GELU_F32) {
        const f32x2 g01 = gelu_erf2((f32x2){v0, v1}), g23 = gelu_erf2((f32x2){v2, v3});
        v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y;
      }
      if (EPI == EPI_H16 || EPI == EPI_GELU_H16) *(h16x4*)(a.out_h16 + (size_t)m * a.N + n) = h16x4{to_h(v0), to_h(v1), to_h(v2), to_h(v3)};
      else *(float4*)(a.out_f32 + (size_t)m * a.N + n) = make_float4(v0, v1, v2, v3);
    }
  }
}

template <int EPI>
static void launch_w4(hipStream_t s, const GemmArgs& a) {
  constexpr int LDS = 2 * 2 * 256 * 128;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    configured = true;
  }
  const int tiles_m = (a.M + 255) / 256, tiles_n = a.N / 256;
  hipLaunchKernelGGL((gemm_w4_kernel<EPI>), dim3(tiles_m * tiles_n), dim3(256), LDS, s, a, tiles_m, tiles_n);
}

void gemm_w4(hipStream_t s, GemmEpi epi, const GemmArgs& a) {
  switch (epi) {
    case EPI_H16: launch_w4<EPI_H16>(s, a); break;
    case EPI_GELU_H16: launch_w4<EPI_GELU_H16>(s, a); break;
    case EPI_RES_F32: launch_w4<EPI_RES_F32>(s, a); break;
    case EPI_GELU_F32: launch_w4<EPI_GELU_F32>(s, a); break;
    default: break;
  }
}

}  // namespace mb
