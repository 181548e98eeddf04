// "Half-tile" fp16 MFMA GEMM for the trunk projections.
//
//   out[M,N] = A[M,K] . W[N,K]^T + bias (+ epilogue); tile (32*MT) x 256 x 64, MT in {6, 8}.
//
// Why this shape: measured on MI355X, a CU pulls at most ~22 B/clk through the L2 -> LDS-DMA path when
// every request is a full 128-byte line (and only ~15 B/clk with 64-byte rows), while its four SIMDs
// retire one 16x16x32 MFMA every ~20 clk each.  A 256x256x64 step needs 64 KiB for 2560 MFMA-clk =
// 25 B/clk; anything smaller is DMA-bound (the 128x128 kernel needs 50 B/clk; a 192x256x32 ring with
// 64-byte rows needed 30 B/clk at half the request efficiency and was slower).  So: K-tiles of 64 halfs (full lines), 256-wide
// tiles, and the whole 160 KiB LDS spent on ONE workgroup.
//
// LDS: two K-tile parities x four half-tiles {A0, A1, B0, B1}.  A-half a holds, for each of the two
// wave rows, that wave's a-th half of its M range; B-half b the b-th 32 columns of each of the four
// wave columns.  A K-tile is consumed in four phases (a,b) = (0,0),(0,1),(1,1),(1,0); every phase is
//     [L] ds_read the half-tile fragments that changed, issue the DMA of ONE half-tile (2 x 1 KiB
//         per wave), on phase 3 wait for the DMA counter (vmcnt(4): never drained in the loop)
//     [M] 4*MH MFMAs (one output quadrant x K=64)
// separated by s_barrier, with the two wave groups (waves 0-3 / 4-7) one barrier apart so that each
// SIMD always has one wave in [M] while its partner is in [L].  A half-tile buffer is re-filled two
// phases after its last reader ([L] of phase q reads, DMA issued in [L] of phase q+2), with the
// half-tile of K-tile t+1 or t+2 -- so data is in flight for 3..6 phases (>= 2000 clk) before use.
// Visibility: each wave waits for its own DMA share before the barrier that precedes the first
// reader's phase ("GEMM hazards").
#include "mb_kernels.h"

namespace mb {

// SEQ = true: "sequence-aligned" tiles.  The trunk's M is nb*257 (256 image tokens + the class token per
// sequence) and 257 is prime, so every ordinary tiling leaves a nearly empty CU round.  With SEQ a tile covers exactly
// one sequence: 256 rows through the regular MT = 8 machinery plus the class-token row as a 17th, one-row m-tile whose
// 4 n-tiles are split between the two wave rows (wm = 0 takes the B0 half in phase 0, wm = 1 the B1 half in phase 1:
// +4 MFMAs per wave per K-tile).  The extra row lives in a 1 KiB "X" buffer per parity, re-filled by one extra DMA
// instruction of wave 7 in phase 3.  tiles = nb * N/256: whole CU rounds for nb = 128.
template <int MT, int EPI, int XP = 0, bool SEQ = false>   // XP (experiments): 1 = DMA only, 2 = no DMA in the main loop
__global__ __launch_bounds__(512, 2) void gemm_ht_kernel(GemmArgs a, int tiles_m, int tiles_n) {
  static_assert(!SEQ || MT == 8, "sequence-aligned tiles use the 256-row machinery");
  constexpr int AUX = 0;   // DMA cache policy: default beats nt (-14 %) and sc1 (-6 %) here, sc0 is equal (measured)
  constexpr int BM = 32 * MT, MH = MT / 2;
  constexpr int AH_ROWS = BM / 2;
  constexpr int AH_BYTES = AH_ROWS * 128, BH_BYTES = 128 * 128;
  constexpr int X_BYTES = SEQ ? 1024 : 0;
  constexpr int PAR_BYTES = 2 * AH_BYTES + 2 * BH_BYTES + X_BYTES;
  constexpr int TILE_ROWS = SEQ ? 257 : BM;
  constexpr int A_INSTR = AH_ROWS / 8;                 // 1 KiB DMA instructions per A half-tile (12 or 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;

  const int L = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int sr = L / (8 * tiles_n);
  const int rows_sr = min(8, tiles_m - sr * 8);
  const int rem = L - sr * 8 * tiles_n;
  const int tn = rem / rows_sr, tm = sr * 8 + (rem - tn * rows_sr);
  const int m0 = tm * TILE_ROWS, n0 = tn * 256;
  const int K = a.K;

  // ---- DMA plan of this wave: 2 instructions per half-tile; lane -> (row 8j + lane>>3, slot lane&7)
  uint32_t offA[2][2], offB[2][2];      // element offsets into A / W
  int dstA[2], dstB[2];                 // byte offset of the instruction inside its half-tile buffer
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ja = min(wave + 8 * j, A_INSTR - 1);     // surplus slot re-loads the last chunk (uniform vmcnt)
    const int hra = ja * 8 + (lane >> 3);
    const int wms = hra / (8 * MT), r = hra - wms * (8 * MT);
    const int slot_a = (lane & 7) ^ ((hra >> 1) & 7);
    dstA[j] = ja * 1024;
    const int jb = wave + 8 * j;
    const int hrb = jb * 8 + (lane >> 3);
    const int wns = hrb >> 5, c = hrb & 31;
    const int slot_b = (lane & 7) ^ ((hrb >> 1) & 7);
    dstB[j] = jb * 1024;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int gm = min(m0 + wms * (16 * MT) + h * (8 * MT) + r, a.M - 1);
      offA[h][j] = (uint32_t)gm * (uint32_t)K + slot_a * 8;
      const int gn = min(n0 + wns * 64 + h * 32 + c, a.N - 1);
      offB[h][j] = (uint32_t)gn * (uint32_t)K + slot_b * 8;
    }
  }
  // X: the class-token row of this sequence, 8 identical source rows (only LDS row 0 is ever consumed)
  const uint32_t offX = (uint32_t)min(m0 + 256, a.M - 1) * (uint32_t)K + ((lane & 7) ^ (((lane >> 3) >> 1) & 7)) * 8;
  auto dma_x = [&](int t) {
    if (SEQ && wave == 7) MB_GLDS16_AUX(a.A + offX + t * 64, smem + (t & 1) * PAR_BYTES + 2 * AH_BYTES + 2 * BH_BYTES, AUX);
  };
  auto dma_a = [&](int t, int h) {
    char* buf = smem + (t & 1) * PAR_BYTES + h * AH_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) MB_GLDS16_AUX(a.A + offA[h][j] + t * 64, buf + dstA[j], AUX);
  };
  auto dma_b = [&](int t, int h) {
    char* buf = smem + (t & 1) * PAR_BYTES + 2 * AH_BYTES + h * BH_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) MB_GLDS16_AUX(a.W + offB[h][j] + t * 64, buf + dstB[j], AUX);
  };

  // ---- fragment read offsets inside a half-tile (rows 128 B, slot swizzled with (row>>1)&7)
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = l15 * 128 + (((ks * 4 + g) ^ (l15 >> 1)) * 16);
  const int xbase = wm * (8 * MT) * 128;              // this wave's rows inside an A half-tile
  const int wbase = 2 * AH_BYTES + wn * 32 * 128;     // this wave's rows inside a B half-tile (from the parity base)

  f32x4 acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  h16x8 xa[MH][2], wb[2][2];
  f32x4 acce[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // SEQ: class row x this wave row's 2 n-tiles
  int xoffe[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) xoffe[ks] = 2 * AH_BYTES + 2 * BH_BYTES + (l15 & 7) * 128 + (((ks * 4 + g) ^ ((l15 & 7) >> 1)) * 16);

  const int nk = K / 64;
  // ---- prologue: all of K-tile 0, plus the two half-tiles of K-tile 1 that no phase of tile 0 stages
  dma_x(0); dma_a(0, 0); dma_b(0, 0); dma_b(0, 1); dma_a(0, 1);
  if (nk > 1) {
    dma_a(1, 0); dma_b(1, 1); dma_x(1);
    if (SEQ && wave == 7) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one barrier behind

#define MB_LOAD_A(H)                                                                            \
  if (XP != 1) _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      xa[i][ks] = *(const h16x8*)(par + (H) * AH_BYTES + xbase + i * 16 * 128 + foff[ks]);
#define MB_LOAD_B(H)                                                                            \
  if (XP != 1) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      wb[i][ks] = *(const h16x8*)(par + wbase + (H) * BH_BYTES + i * 16 * 128 + foff[ks]);
#define MB_SYNC_L()                                     \
  __builtin_amdgcn_s_barrier();                         \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
  __builtin_amdgcn_sched_barrier(0);                    \
  __builtin_amdgcn_s_setprio(1);
#define MB_MMA(AH, BH)                                                                              \
  if (XP != 1) _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int n = 0; n < 2; ++n)       \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
          acc[(BH) * 2 + n][(AH) * MH + i] = MB_MFMA_16x16x32(wb[n][ks], xa[i][ks], acc[(BH) * 2 + n][(AH) * MH + i]); \
  __builtin_amdgcn_s_setprio(0);                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nk; ++t) {
    const char* par = smem + (t & 1) * PAR_BYTES;
    const bool n1 = XP != 2 && t + 1 < nk, n2 = XP != 2 && t + 2 < nk;
    // ---- phase 0: quadrant (A0, B0); refill A1 of the other parity with K-tile t+1
    MB_LOAD_A(0) MB_LOAD_B(0)
    h16x8 xe[2];
    if (SEQ && wm == 0) { xe[0] = *(const h16x8*)(par + xoffe[0]); xe[1] = *(const h16x8*)(par + xoffe[1]); }
    if (n1) dma_a(t + 1, 1);
    MB_SYNC_L()
    if (SEQ && wm == 0) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acce[n] = MB_MFMA_16x16x32(wb[n][ks], xe[ks], acce[n]);
    }
    MB_MMA(0, 0)
    // ---- phase 1: (A0, B1); refill B0 of the other parity with K-tile t+1
    MB_LOAD_B(1)
    if (SEQ && wm == 1) { xe[0] = *(const h16x8*)(par + xoffe[0]); xe[1] = *(const h16x8*)(par + xoffe[1]); }
    if (n1) dma_b(t + 1, 0);
    MB_SYNC_L()
    if (SEQ && wm == 1) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acce[n] = MB_MFMA_16x16x32(wb[n][ks], xe[ks], acce[n]);
    }
    MB_MMA(0, 1)
    // ---- phase 2: (A1, B1); refill A0 of this parity with K-tile t+2
    MB_LOAD_A(1)
    if (n2) dma_a(t + 2, 0);
    MB_SYNC_L() MB_MMA(1, 1)
    // ---- phase 3: (A1, B0); refill B1 of this parity with K-tile t+2; all of K-tile t+1 must have landed
    MB_LOAD_B(0)
    if (n2) {
      dma_b(t + 2, 1); dma_x(t + 2);
      if (SEQ && wave == 7) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MB_SYNC_L() MB_MMA(1, 0)
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();          // balance the barrier count of the two groups
#undef MB_LOAD_A
#undef MB_LOAD_B
#undef MB_SYNC_L
#undef MB_MMA

  // ---- epilogue: acc[nt][mt] holds out[m][n..n+3]; wave rows: half h = mt / MH, tile i = mt % MH
  auto emit = [&](int m, int n, const f32x4& v) {
    const float4 b = *(const float4*)(a.bias + n);
    float v0 = v[0] + b.x, v1 = v[1] + b.y, v2 = v[2] + b.z, v3 = v[3] + b.w;
    if (EPI == EPI_RES_F32) {
      const float4 r = *(const float4*)(a.residual + (size_t)m * a.N + n);
      v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
    }
    if (EPI == EPI_GELU_H16 || EPI == EPI_GELU_F32) {
      v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
    }
    if (EPI == EPI_H16 || EPI == EPI_GELU_H16) {
      *(h16x4*)(a.out_h16 + (size_t)m * a.N + n) = h16x4{to_h(v0), to_h(v1), to_h(v2), to_h(v3)};
    } else {
      *(float4*)(a.out_f32 + (size_t)m * a.N + n) = make_float4(v0, v1, v2, v3);
    }
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * (16 * MT) + (mt / MH) * (8 * MT) + (mt % MH) * 16 + l15;
    if (m >= (SEQ ? m0 + 256 : a.M)) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) emit(m, n0 + wn * 64 + (nt >> 1) * 32 + (nt & 1) * 16 + g * 4, acc[nt][mt]);
  }
  if (SEQ && l15 == 0) {                       // class-token row: wave row wm owns the B-half wm of its 64 columns
#pragma unroll
    for (int n = 0; n < 2; ++n) emit(m0 + 256, n0 + wn * 64 + wm * 32 + n * 16 + g * 4, acce[n]);
  }
}

template <int MT, int EPI, int XP = 0, bool SEQ = false>
static void launch_ht(hipStream_t s, const GemmArgs& a) {
  constexpr int BM = 32 * MT;
  constexpr int LDS = 2 * (BM * 128 + 2 * 128 * 128 + (SEQ ? 1024 : 0));
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)gemm_ht_kernel<MT, EPI, XP, SEQ>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    configured = true;
  }
  const int tiles_m = SEQ ? a.M / 257 : (a.M + BM - 1) / BM, tiles_n = a.N / 256;
  hipLaunchKernelGGL((gemm_ht_kernel<MT, EPI, XP, SEQ>), dim3(tiles_m * tiles_n), dim3(512), LDS, s, a, tiles_m, tiles_n);
}

bool gemm_ht_supported(GemmEpi epi, const GemmArgs& a) {
  return epi != EPI_LOGITS_F32 && a.N % 256 == 0 && a.K % 64 == 0 && a.M >= 512 &&
         (uint64_t)a.M * a.K < (1ull << 32) && (uint64_t)a.N * a.K < (1ull << 32);
}

void gemm_ht(hipStream_t s, GemmEpi epi, const GemmArgs& a, int mt) {
  if (mt == 16) { launch_ht<6, EPI_RES_F32, 1>(s, a); return; }   // ablations (see tools/xp_gemm.py)
  if (mt == 26) { launch_ht<6, EPI_RES_F32, 2>(s, a); return; }
  if (mt == 18) { launch_ht<8, EPI_RES_F32, 1>(s, a); return; }
  if (mt == 28) { launch_ht<8, EPI_RES_F32, 2>(s, a); return; }
  if (mt != 6 && mt != 8 && mt != 257) {
    // pick the tile height that wastes fewer CU-rounds (one workgroup per CU)
    static int num_cu = 0;
    if (!num_cu) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev);
      if (num_cu <= 0) num_cu = 256;
    }
    auto cost = [&](int m) {
      const long tiles = (long)((a.M + 32 * m - 1) / (32 * m)) * (a.N / 256);
      return (double)((tiles + num_cu - 1) / num_cu) * 32 * m * (m == 8 ? 1.0 : 1.04);
    };
    mt = cost(8) <= cost(6) ? 8 : 6;
    if (a.M % 257 == 0) {          // one tile per sequence: costs 272/256 of the MFMA work but leaves no ragged round
      const long tiles = (long)(a.M / 257) * (a.N / 256);
      if ((double)((tiles + num_cu - 1) / num_cu) * 272 < cost(mt)) mt = 257;
    }
  }
#define MB_HT_CASE(E)                                                      \
  case E:                                                                  \
    if (mt == 257) launch_ht<8, E, 0, true>(s, a);                         \
    else if (mt == 8) launch_ht<8, E>(s, a);                               \
    else launch_ht<6, E>(s, a);                                            \
    break;
  switch (epi) {
    MB_HT_CASE(EPI_H16) MB_HT_CASE(EPI_GELU_H16) MB_HT_CASE(EPI_RES_F32) MB_HT_CASE(EPI_GELU_F32)
    default: break;
  }
#undef MB_HT_CASE
}

}  // namespace mb
