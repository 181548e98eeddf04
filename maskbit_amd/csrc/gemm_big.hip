// Large-tile fp16 MFMA GEMM for the four trunk projections (M = nb*257 rows, N in {1024,3072,4096}).
//
//   out[M,N] = A[M,K] . W[N,K]^T + bias (+ epilogue), same contract as gemm.hip.
//
// Tile (32*MT) x 256 x 32, 512 threads = 8 waves as 2(M) x 4(N); a wave owns (16*MT) x 64 outputs
// = MT x 4 MFMA 16x16x32 tiles (MT in 4..8 -> BM in {128,160,192,224,256}; the host picks the BM
// that wastes the fewest CU-rounds for the given M, N -- M = 128*257 makes power-of-two tilings
// leave a 0.4%-full extra round).  One workgroup per CU, 4-stage LDS ring of (BM+256) x 32 halfs
// (<= 128 KiB), filled by 16-byte LDS-DMA two K-tiles ahead with COUNTED vmcnt waits (the DMA
// queue is never drained in the main loop).
//
// Schedule: every K-tile is two barrier-separated phases per wave,
//     [L]  ds_read the tile's fragments, issue the DMA for tile t+2, wait for OWN DMA of tile t+1
//     [M]  4*MT MFMAs
// and the two wave groups (waves 0-3 / 4-7 = one wave of each group per SIMD) run one barrier
// apart (group 1 executes one extra s_barrier up front, group 0 one at the end), so that on every
// SIMD one wave is in [M] while the other is in [L]: the matrix pipe sees back-to-back MFMA phases
// and LDS/DMA latency is hidden behind the partner's math.  Hazards: a tile is read in [L](t) only
// after every wave waited for its own share of it before an earlier barrier (RAW), and the ring slot
// re-filled in [L](t) held tile t-2, whose readers finished two barriers ago (WAR) -- see DESIGN.md.
// LDS rows are 64 B; the 16-byte slot is XOR-swizzled with (row>>2)&3 on the DMA source address and
// on the fragment read, which makes ds_read_b128 conflict-free.
#include "mb_kernels.h"

namespace mb {

constexpr int GB_BN = 256, GB_BK = 32;
// GB_STAGES (template) = LDS ring depth; the DMA runs GB_STAGES-2 K-tiles ahead.

// 64-byte LDS rows: 16-byte slot s of row r lives at physical slot s ^ swz64(r).  ds_read_b128 is
// serviced in the lane groups {0-3,12-15,20-27},{4-11,16-19,28-31} (+32); with h(q) = (-q)&3,
// q = (r>>2)&3, the 16 lanes of every group hit 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz64(int row) { return (0 - (row >> 2)) & 3; }

__device__ long long g_gemm_dbg[8 * 2 * 8 * 16];   // [iter 0..7][wave group][stamp 0..7] x 16 blocks (XP == 4 only)

template <int MT, int EPI, int GB_STAGES, int XP = 0>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int GB_AHEAD = GB_STAGES - 2;
  constexpr int BM = 32 * MT;
  constexpr int ROWS = BM + GB_BN;                 // rows of one stage: [X tile | W tile]
  constexpr int STAGE_BYTES = ROWS * 64;
  constexpr int NCH = ROWS / 16;                   // 16-row DMA chunks per stage
  constexpr int CPW = (NCH + 7) / 8;               // chunks (= DMA instructions) per wave per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // stagger group
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;

  // ---- tile order: XCD-contiguous chunks of the list; inside, super-rows of 8 m-tiles (L2 patch 8 x 4)
  const int L = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int sr = L / (8 * tiles_n);
  const int rows_sr = min(8, tiles_m - sr * 8);
  const int rem = L - sr * 8 * tiles_n;
  const int tn = rem / rows_sr, tm = sr * 8 + (rem - tn * rows_sr);
  const int m0 = tm * BM, n0 = tn * GB_BN;
  const int K = a.K;

  // ---- DMA sources: chunk c covers stage rows [16c, 16c+16); lane -> row 16c + (lane>>2), slot lane&3
  const h16* src[CPW];
  int dst[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int c = min(wave + 8 * j, NCH - 1);      // surplus slots re-load the last chunk (keeps vmcnt uniform)
    const int row = c * 16 + (lane >> 2);
    const int slot = (lane & 3) ^ swz64(row);
    if (c < BM / 16) src[j] = a.A + (size_t)min(m0 + row, a.M - 1) * K + slot * 8;
    else src[j] = a.W + (size_t)min(n0 + row - BM, a.N - 1) * K + slot * 8;
    dst[j] = c * 16 * 64;
  }
  auto stage = [&](int t) {
    char* sb = smem + (t % GB_STAGES) * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < CPW; ++j) MB_GLDS16(src[j] + t * GB_BK, sb + dst[j]);
  };

  // ---- fragment offsets inside a stage (swizzle term depends on lane only: rows are 16-aligned per tile)
  const int foff = l15 * 64 + ((g ^ swz64(l15)) * 16);
  const int xoff = wm * (16 * MT) * 64 + foff;
  const int woff = (BM + wn * 64) * 64 + foff;

  f32x4 acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / GB_BK;
#pragma unroll
  for (int p = 0; p < GB_AHEAD; ++p)
    if (p < nk) stage(p);
  // tile 0 must be complete; up to GB_AHEAD-1 later tiles may stay in flight
  if (nk >= GB_AHEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((GB_AHEAD - 1) * CPW) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // tile 0 complete for every wave
  if (XP != 1 && grp == 1) __builtin_amdgcn_s_barrier();      // stagger: group 1 runs one barrier behind

  int slot_rd = 0, slot_wr = GB_AHEAD % GB_STAGES;
  for (int t = 0; t < nk; ++t) {
    // ---------------- [L] -----------------
    const bool rec = XP == 4 && blockIdx.x < 16 && (wave & 3) == 0 && lane == 0 && t >= 40 && t < 48;
    long long* dbg = g_gemm_dbg + (((size_t)blockIdx.x * 8 + (t - 40)) * 2 + grp) * 8;
    if (rec) dbg[0] = clock64();
    const char* sb = smem + slot_rd * STAGE_BYTES;
    h16x8 wf[4], xf[MT];
    if (XP != 7) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *(const h16x8*)(sb + woff + i * 16 * 64);
#pragma unroll
      for (int j = 0; j < MT; ++j) xf[j] = *(const h16x8*)(sb + xoff + j * 16 * 64);
    }
    if (rec) dbg[1] = clock64();
    if (XP != 2 && t + GB_AHEAD < nk) {
      char* wbuf = smem + slot_wr * STAGE_BYTES;
#pragma unroll
      for (int j = 0; j < CPW; ++j) MB_GLDS16(src[j] + (t + GB_AHEAD) * GB_BK, wbuf + dst[j]);
      if (rec) dbg[2] = clock64();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((GB_AHEAD - 1) * CPW) : "memory");   // own share of tile t+1 has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (rec) dbg[3] = clock64();
    slot_rd = slot_rd + 1 == GB_STAGES ? 0 : slot_rd + 1;
    slot_wr = slot_wr + 1 == GB_STAGES ? 0 : slot_wr + 1;
    __builtin_amdgcn_s_barrier();
    if (rec) dbg[4] = clock64();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (rec) dbg[5] = clock64();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- [M] -----------------
    if (XP != 5 && XP != 6) __builtin_amdgcn_s_setprio(1);
    if (XP == 6) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (XP == 3) { if (j == 0) acc[i][j] = MB_MFMA_16x16x32(wf[i], xf[(i + t) % MT], acc[i][j]); }
        else if (XP != 7) acc[i][j] = MB_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
      }
    if (XP != 5 && XP != 6) __builtin_amdgcn_s_setprio(0);
    if (XP == 6) __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
    if (rec) dbg[6] = clock64();
    __builtin_amdgcn_s_barrier();
    if (rec) dbg[7] = clock64();
  }
  if (XP != 1 && grp == 0) __builtin_amdgcn_s_barrier();      // balance the barrier count of the two groups

  // ---- epilogue: lane holds out[m = ..+l15][n = ..+g*4 .. +3]
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + wm * (16 * MT) + j * 16 + l15;
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + g * 4;
      const float4 b = *(const float4*)(a.bias + n);
      float v0 = acc[i][j][0] + b.x, v1 = acc[i][j][1] + b.y, v2 = acc[i][j][2] + b.z, v3 = acc[i][j][3] + b.w;
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
    }
  }
}

// Pick the M tile that minimises wasted CU-rounds: cost = rounds(tiles) * BM (N tiling is fixed).
static int pick_mt(int M, int N, int num_cu) {
  int best = 8;
  double best_cost = 1e30;
  for (int mt = 8; mt >= 4; --mt) {
    const int bm = 32 * mt;
    const long tiles = (long)((M + bm - 1) / bm) * (N / GB_BN);
    const long rounds = (tiles + num_cu - 1) / num_cu;
    const double cost = (double)rounds * bm * (1.0 + 0.02 * (8 - mt));   // mild preference for larger tiles
    if (cost < best_cost) { best_cost = cost; best = mt; }
  }
  return best;
}

template <int MT, int EPI, int GB_STAGES, int XP = 0>
static void launch_big(hipStream_t s, const GemmArgs& a) {
  constexpr int BM = 32 * MT;
  constexpr int LDS = GB_STAGES * (BM + GB_BN) * 64;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<MT, EPI, GB_STAGES, XP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    configured = true;
  }
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / GB_BN;
  hipLaunchKernelGGL((gemm_big_kernel<MT, EPI, GB_STAGES, XP>), dim3(tiles_m * tiles_n), dim3(512), LDS, s, a, tiles_m, tiles_n);
}

template <int EPI>
static void dispatch_mt(hipStream_t s, const GemmArgs& a, int mt) {
  switch (mt) {
    case 4: launch_big<4, EPI, 5>(s, a); break;
    case 5: launch_big<5, EPI, 5>(s, a); break;
    case 6: launch_big<6, EPI, 5>(s, a); break;
    case 7: launch_big<7, EPI, 5>(s, a); break;
    case 8: launch_big<8, EPI, 5>(s, a); break;
    case 26: launch_big<6, EPI, 5, 1>(s, a); break;  // experiments: 26 no stagger, 36 no DMA, 46 1/MT of the MFMAs
    case 36: launch_big<6, EPI, 5, 2>(s, a); break;
    case 56: launch_big<6, EPI, 5, 4>(s, a); break;  // timestamped
    case 66: launch_big<6, EPI, 5, 5>(s, a); break;  // no setprio
    case 86: launch_big<6, EPI, 5, 7>(s, a); break;  // DMA only
    case 88: launch_big<8, EPI, 5, 7>(s, a); break;
    case 76: launch_big<6, EPI, 5, 6>(s, a); break;  // prio on [L]
    case 46: launch_big<6, EPI, 5, 3>(s, a); break;
    case 14: launch_big<4, EPI, 4>(s, a); break;   // 1x: 4-stage ring (A/B experiments)
    case 15: launch_big<5, EPI, 4>(s, a); break;
    case 16: launch_big<6, EPI, 4>(s, a); break;
    case 17: launch_big<7, EPI, 4>(s, a); break;
    default: launch_big<8, EPI, 4>(s, a); break;
  }
}

int gemm_debug_read(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_dbg), sizeof(long long) * n);
}

bool gemm_big_supported(GemmEpi epi, const GemmArgs& a) {
  return epi != EPI_LOGITS_F32 && a.N % GB_BN == 0 && a.K % GB_BK == 0 && a.K >= 2 * GB_BK && a.M >= 512;
}

void gemm_big(hipStream_t s, GemmEpi epi, const GemmArgs& a, int force_mt) {
  static int num_cu = 0;
  if (!num_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (num_cu <= 0) num_cu = 256;
  }
  const int mt = force_mt ? force_mt : pick_mt(a.M, a.N, num_cu);
  switch (epi) {
    case EPI_H16: dispatch_mt<EPI_H16>(s, a, mt); break;
    case EPI_GELU_H16: dispatch_mt<EPI_GELU_H16>(s, a, mt); break;
    case EPI_RES_F32: dispatch_mt<EPI_RES_F32>(s, a, mt); break;
    case EPI_GELU_F32: dispatch_mt<EPI_GELU_F32>(s, a, mt); break;
    default: break;
  }
}

}  // namespace mb
