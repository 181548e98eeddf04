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
//
// Persistent: a workgroup walks tiles b, b+G, b+2G, ... (G = grid = #CUs).  When a tile's K loop ends, the
// DMA of the NEXT tile's first two K-tiles is issued first, then the epilogue arithmetic runs in place on the
// accumulators (bias, residual, GELU) while that DMA is in flight, then `vmcnt(0)` + barrier, and only then
// the output stores are issued -- they drain behind the next tile's first two K-tiles (the first counted
// wait that covers them is in K-tile 1).  Output bursts of all CUs are otherwise fully exposed: measured
// 9 / 17 / 34 us per tile (fp16 / GELU / fp32+residual epilogue) next to a 26 us K = 1024 main loop.
// (Tried and rejected, twice: starting half of the CUs 4 - 24 us late to de-synchronise those bursts -- slower by the delay itself, in every
// epilogue form: the epilogue is bound per CU, like the main loop, not by aggregate HBM bandwidth.  Also rejected: 16 residual loads in flight per
// lane instead of 4 in the fp32+residual epilogue -- same span per tile (tools/ht_trace.py), 23 - 60 spilled VGPRs; and the bias vector parked in
// LDS instead of fetched per tile -- 0.2 us of a 32 us tile: "pass 1" is bound by the issue of the next tile's 128 KiB prologue DMA; and, for
// GELU tiles, the second wave group doing its arithmetic first and issuing its share of that DMA afterwards -- 1 % slower.
// The fp32 + residual output pass is bound per CU AND chip-wide at once: with 256 / 128 / 64 workgroups running it takes 22.5 / 20.4 / 16.8 us per
// tile (512 KiB read-modify-written in 64-byte row segments), and 256 x 512 KiB in 22.5 us is 5.9 TB/s, the HBM rate.  Landing the residual rows in
// the free LDS by LDS-DMA instead of registers -- 16 KiB in flight per wave, the next tile's prologue deferred -- left a tile at 18.6 us on 64
// workgroups and 108 us per launch on 256: not a latency problem.)
#include <algorithm>
#include <cstdlib>

#include "mb_kernels.h"

namespace mb {

// 16-byte slot swizzle of a 128-byte LDS row (DMA source address and fragment read): slot ^ MB_SWZ(row)
#ifdef MB_SWZ_ROW
#define MB_SWZ(r) ((r) & 7)
#else
#define MB_SWZ(r) (((r) >> 1) & 7)
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
// A fragment pair (two 16-byte LDS reads of one lane: k-steps 0 and 1 of a K-tile) lives in ONE 8-VGPR tuple; the f16 MFMAs take its halves
typedef h16 h16x16 __attribute__((ext_vector_type(16)));

// SEQ = true: "sequence-aligned" tiles.  The trunk's M is nb*257 (256 image tokens + the class token per
// sequence) and 257 is prime, so every ordinary tiling leaves a nearly empty CU round.  With SEQ a tile covers exactly
// one sequence (GemmArgs.seq_rows = 1025, the 512 x 512 models: a quarter of one, the last quarter with the class token): 256 rows through the regular MT = 8 machinery plus the class-token row as a 17th, one-row m-tile whose
// 4 n-tiles are split between the two wave rows (wm = 0 takes the B0 half in phase 0, wm = 1 the B1 half in phase 1:
// +4 MFMAs per wave per K-tile).  The extra row lives in a 1 KiB "X" buffer per parity, re-filled by one extra DMA
// instruction of wave 7 in phase 3.  tiles = nb * N/256: whole CU rounds for nb = 128.
// PAIR = true (with SEQ): "CFG pair" tiles for the differential form of classifier-free guidance (DESIGN.md "Precision").  The M rows are
// a.pair_rows conditional rows followed by a.pair_rows unconditional rows (whole 257-token sequences); in A the unconditional rows hold
// the DIFFERENCE operand fp16(x_u - x_c).  A tile covers 128 tokens of one sequence pair: the A0 half-tile = their conditional rows, the A1
// half-tile = their difference rows, so one lane ends up with acc_c (m-tiles 0..3) and acc_delta (m-tiles 4..7) of the SAME tokens and the
// epilogue emits out_c = f(acc_c), out_u = f(acc_c + acc_delta): the rounding error of the conditional operand is common to both
// streams and cancels in (c - u), which is what the guidance scale multiplies.  The class-token m-tile carries two rows (lane rows 0 / 1 =
// class row of c / its difference row); both tiles of a sequence pair compute it, the second one stores it.
// Timeline instrumentation (tools/ht_trace.py builds its own copy with -DMB_HT_TRACE; never in the product library): wave 0 of every workgroup
// stamps the 100 MHz wall clock at the phase boundaries of its first 8 tiles -> trace[workgroup][tile][8].
#ifdef MB_HT_TRACE
__device__ long long* g_ht_trace = nullptr;
#define MB_TRACE(k) do { if (g_ht_trace && tid == 0 && trace_it < 8) g_ht_trace[((size_t)blockIdx.x * 8 + trace_it) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define MB_TRACE(k) do { } while (0)
#endif

// XP = 6 (round 4): MX-fp4 correction passes as MINI-TILES between the fp16 K-tiles (GemmArgs.lo / nlo).  A mini-tile = 128 token rows (pair
// tiles: the conditional rows; plain tiles: one 128-row half of the sequence) x the tile's 256 weight rows x 128 K-elements = 8 + 16 KiB in a
// buffer behind the two K-tile parities.  Its DMA is issued in phases 1 / 2 of a fp16 K-tile (one + two instructions per wave, older than the
// half-tiles the K-tile's own counted wait leaves in flight, so that wait covers it), and it is multiplied in a FIFTH phase after phase 3 of the
// K-tile that follows its landing: 8 fragment reads, 16 scaled MFMAs per wave, same two-barrier rhythm and wave-group stagger as the other phases.
// What the old lo K-tiles (XP = 5) paid -- a K-tile skeleton of eight barriers and a one-K-tile DMA lead for a quarter of a tile's arithmetic,
// 1.9-2.6 us each -- shrinks to the fifth phase itself; the staging hides under the fp16 K-tiles, whose L2 -> LDS path has the slack.
// NS = 2 / 4 (round 4; plain sequence tiles with mini-tiles, fp32 + residual epilogue): HALF- / QUARTER-COLUMN tiles for small batches.  A tile keeps
// its 256 rows but only 64 / NS of every wave's 64 columns (column block `hb` of the 256-column tile, staged in the B0 slot; phases (A0, B1) and
// (A1, B1) multiply nothing), so an N = 1024 GEMM over 16 sequences runs 128 / 256 tiles instead of 64 on 256 CUs.  Every output element sees the
// same K-tiles and mini-tiles in the same order as in a full tile: bit-identical results, so the choice may depend on the batch size.
template <int MT, int EPI, int XP = 0, bool SEQ = false, bool PAIR = false, int NS = 1>   // XP: 0 = fp16 K-tiles only, 6 = + MX-fp4 mini-tiles (4 / 5, the e4m3 / MX-fp4 lo K-TILES of rounds 1-3, were removed in round 5)
__global__ __launch_bounds__(512, 2) void gemm_ht_kernel(GemmArgs a, int tiles_m, int tiles_n) {
  constexpr bool HN = NS > 1, QN = NS == 4;
  constexpr int NTW = 4 / NS;                          // n-tiles of a wave
  static_assert(NS == 1 || NS == 2 || NS == 4, "column split");
  static_assert(!HN || (SEQ && !PAIR && XP == 6 && EPI == EPI_RES_F32), "half-column tiles: plain sequence tiles with mini-tiles, residual epilogue");
  static_assert(!SEQ || MT == 8, "sequence-aligned tiles use the 256-row machinery");
  static_assert(XP == 0 || XP == 6, "fp16 K-tiles, optionally with mini-tiles");
  static_assert(!PAIR || SEQ, "pair tiles are sequence-aligned");
  constexpr bool MINI = XP == 6;
  static_assert(!MINI || SEQ, "mini-tile passes are written for sequence-aligned tiles");
  constexpr int MINI_A = 128 * 64, MINI_B = 256 * 64;   // bytes: 128 token rows / 256 weight rows x 128 e2m1 values
  // Every instance walks the tile list persistently (round 1 kept the fp32+residual epilogue at one tile per workgroup: it spilled
  // VGPRs inside the K loop then; with the present epilogue it does not: 244-248 VGPRs, no scratch).
  constexpr bool PERSIST = true;
  constexpr int AUX = 0;   // DMA cache policy: default beats nt (-14 %) and sc1 (-6 %) here, sc0 is equal (measured)
  constexpr int BM = 32 * MT, MH = MT / 2;
  constexpr int AH_ROWS = BM / 2;
  constexpr int AH_BYTES = AH_ROWS * 128, BH_BYTES = 128 * 128;
  constexpr int X_BYTES = SEQ ? 1024 : 0;
  constexpr int PAR_BYTES = 2 * AH_BYTES + 2 * BH_BYTES + X_BYTES;
  constexpr int MINI_OFF = 2 * PAR_BYTES;
  constexpr int A_INSTR = AH_ROWS / 8;                 // 1 KiB DMA instructions per A half-tile (12 or 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;
  const int K = a.K, nk = K / 64;
  const int KA = a.kw ? a.kw : K, nka = KA / 64;       // split activations: A and A2 have kw = K/2 columns each
  const int KW = a.kw ? a.kw : K, nkw = KW / 64;       // split activations: W has K/2 columns and is swept twice, A2 (lo halves) takes over from A
  // split activations (A2 / kw): K-tiles >= nka read the lo halves A2 against the same W columns
  const h16* const Alo = a.A2 ? a.A2 : a.A;
  const h16* const Wlo = a.W;
  // sequence tiles: SQ rows per sequence (class token last), TPS tiles per sequence (pair) -- a power of two: (SQ - 1) / 128 pair tiles (2 or 8) of 2 groups
  // of 64 tokens, (SQ - 1) / 256 plain tiles (1 or 4) of 4 groups; the LAST tile of a sequence computes and stores the class-token row(s)
  const int SQ = a.seq_rows ? a.seq_rows : 257;
  const int tps_sh = 31 - __builtin_clz((unsigned)((SQ - 1) >> (PAIR ? 7 : 8)));
  const int TPS = 1 << tps_sh;
  const int ntiles = tiles_m * tiles_n;

  // ---- per-tile DMA plan of this wave: 2 instructions per half-tile; lane -> (row 8j + lane>>3, slot lane&7)
  struct Plan {
    uint32_t offA[2][2], offB[2][2], offX;   // element offsets into A / W
    int m0, n0;
    int cls;                                 // SEQ: the (conditional) class-token row of this tile's sequence (pair); its last tile stores it
    int q;                                   // SEQ: which 256-token (pair: 128-token) part of the sequence this tile covers
    int seq;                                 // SEQ: the (conditional) sequence of this tile
    int hb;                                  // NS > 1: which 64 / NS-column block of every wave's 64 columns this tile computes
  };
  int dstA[2], dstB[2];                       // byte offset of the instruction inside its half-tile buffer
#pragma unroll
  for (int j = 0; j < 2; ++j) { dstA[j] = min(wave + 8 * j, A_INSTR - 1) * 1024; dstB[j] = (QN ? (wave >> 1) * 4 + (wave & 1) : wave + 8 * j) * 1024; }
  auto make_plan = [&](int vb, Plan& p) {
    int lane_o = lane;                                 // opaque copy: keeps the plan's lane arithmetic from being hoisted
    asm volatile("" : "+v"(lane_o));                   // out of the tile loop and held in VGPRs across the K loop
    const int L = xcd_remap(vb, ntiles);              // XCD-contiguous chunks; inside, 8 x tiles_n super-rows
    const int sr = L / (8 * tiles_n);
    const int rows_sr = min(8, tiles_m - sr * 8);
    const int rem = L - sr * 8 * tiles_n;
    const int tn = rem / rows_sr, tm = sr * 8 + (rem - tn * rows_sr);
    p.m0 = SEQ ? (tm >> tps_sh) * SQ + (tm & (TPS - 1)) * (PAIR ? 128 : 256) : tm * BM; p.n0 = (tn / NS) * 256;
    p.hb = tn % NS;
    p.cls = SEQ ? (tm >> tps_sh) * SQ + SQ - 1 : 0; p.q = SEQ ? tm & (TPS - 1) : 0;
    p.seq = SEQ ? tm >> tps_sh : tm;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ja = min(wave + 8 * j, A_INSTR - 1);   // surplus slot re-loads the last chunk (uniform vmcnt)
      const int hra = ja * 8 + (lane_o >> 3);
      const int wms = hra / (8 * MT), r = hra - wms * (8 * MT);
      const int slot_a = (lane_o & 7) ^ MB_SWZ(hra);
      const int hrb = (QN ? (wave >> 1) * 4 + (wave & 1) : wave + 8 * j) * 8 + (lane_o >> 3);   // (quarter-column tiles: 16 of a wave column's 32 slot rows, one instruction per wave)
      const int wns = hrb >> 5, c = hrb & 31;
      const int slot_b = (lane_o & 7) ^ MB_SWZ(hrb);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gm = PAIR ? p.m0 + h * a.pair_rows + hra : min(p.m0 + wms * (16 * MT) + h * (8 * MT) + r, a.M - 1);
        p.offA[h][j] = (uint32_t)gm * (uint32_t)KA + slot_a * 8;
        const int gn = min(p.n0 + wns * 64 + (HN ? p.hb * (64 / NS) : h * 32) + c, a.N - 1);
        p.offB[h][j] = (uint32_t)gn * (uint32_t)KW + slot_b * 8;
      }
    }
    // X: the class-token row of this sequence, 8 identical source rows (only LDS row 0 is ever consumed)
    p.offX = (uint32_t)(PAIR ? p.cls + ((lane_o >> 3) == 1 ? a.pair_rows : 0) : min(p.cls, a.M - 1)) * (uint32_t)KA + ((lane_o & 7) ^ MB_SWZ(lane_o >> 3)) * 8;
  };
  // trace builds, modes 3 / 4: the K loop's DMA is dropped / re-reads K-tiles 0 and 1 (always L2 hits) -- what the loop costs without (slow) memory
#if defined(MB_HT_TRACE) && MB_HT_TRACE == 3
#define MB_TRACE_DMA(t) if ((t) >= 2) return
#elif defined(MB_HT_TRACE) && MB_HT_TRACE == 4
#define MB_TRACE_DMA(t) t &= 1
#else
#define MB_TRACE_DMA(t)
#endif
  auto dma_x = [&](const Plan& p, int t) {
    MB_TRACE_DMA(t);
    if (SEQ && wave == 7) MB_GLDS16_AUX((t < nka ? a.A : Alo) + p.offX + (t < nka ? t : t - nka) * 64, smem + (t & 1) * PAR_BYTES + 2 * AH_BYTES + 2 * BH_BYTES, AUX);
  };
  auto dma_a = [&](const Plan& p, int t, int h) {
    MB_TRACE_DMA(t);
    char* buf = smem + (t & 1) * PAR_BYTES + h * AH_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // (a "compressed" plan -- one per-lane offset + uniform row steps instead of this table of four, 6-11 VGPRs less -- measured 3.5 % slower in the
      // pair mini-tile kernels and equal in the plain ones, round 5: the address arithmetic lands in the [L] phases)
      MB_GLDS16_AUX((t < nka ? a.A : Alo) + p.offA[h][j] + (t < nka ? t : t - nka) * 64, buf + dstA[j], AUX);
    }
  };
  auto dma_b = [&](const Plan& p, int t, int h) {
    if (HN && h == 1) return;                          // half-column tiles stage one B half (in the B0 slot)
    MB_TRACE_DMA(t);
    char* buf = smem + (t & 1) * PAR_BYTES + 2 * AH_BYTES + h * BH_BYTES;
#pragma unroll
    for (int j = 0; j < (QN ? 1 : 2); ++j) {
      MB_GLDS16_AUX((t < nkw ? a.W : Wlo) + p.offB[h][j] + (t < nkw ? t : t - nkw) * 64, buf + dstB[j], AUX);
    }
  };
  // ---- mini-tiles (XP = 6).  nmk per operand set / row half; mini j: set or half ps = j / nmk, K-elements [128 jj, 128 jj + 128), jj = j % nmk.
  // One DMA instruction covers 16 rows x 64 B (lane -> row lane >> 2, 16-byte chunk lane & 3, swizzled with (row >> 1) & 3 = (lane >> 3) & 3: the
  // one of seven candidate layouts of 64-byte rows whose ds_read_b128 fragment reads run at the LDS rate -- tools/micro/lds_b128.hip; the first
  // choice, (row >> 2) & 3, read at half of it: SQ_LDS_BANK_CONFLICT 4.2 M cycles per FFN-up launch); the per-lane part of the address is the same
  // for both operands.
  // (KM: the K extent of the operands = of the corrections; with split activations -- A2 / kw, plain tiles -- the fp16 sweep is twice as long)
  const int KM = a.kw ? a.kw : K;
  const int nmk = KM / 128;
  const int nseq = PAIR ? a.pair_rows / SQ : a.M / SQ;
  const int grp_bytes = PAIR ? TPS * 128 : TPS * 256;    // scale bytes per (64-column block, sequence): 64 per 64-token group
  const bool mini_every = MINI && (PAIR ? a.nlo == 2 : !a.kw);   // one mini-tile per fp16 K-tile (else one per two)
  auto mini_lane = [&]() -> uint32_t {
    int lo_ = lane; asm volatile("" : "+v"(lo_));
    return (uint32_t)((lo_ >> 2) * 2 * KM + (((lo_ & 3) ^ ((lo_ >> 3) & 3)) * 16));
  };
  auto mini_dma_a = [&](const Plan& p, int j) {          // 1 instruction per wave
    const int ps = j >= nmk ? 1 : 0, jj = j - ps * nmk;
    const uint8_t* base = a.lo[PAIR ? ps : 0].A4;
    const uint32_t row0 = PAIR ? p.m0 + wave * 16 : p.m0 + (wave >> 2) * 128 + ps * 64 + (wave & 3) * 16;
    MB_GLDS16_AUX(base + (size_t)row0 * 2 * KM + jj * 64 + mini_lane(), smem + MINI_OFF + wave * 1024, AUX);
  };
  auto mini_dma_b = [&](const Plan& p, int j) {          // 2 instructions per wave, 1 KiB of contiguous memory each (w4_packed_offset)
    const int ps = j >= nmk ? 1 : 0, jj = j - ps * nmk;
    const uint8_t* base = a.lo[PAIR ? ps : 0].W4;
    int lo_ = lane; asm volatile("" : "+v"(lo_));
    if constexpr (QN) {                                  // 64 weight rows: waves 0..3 stage the 16 rows of block hb of wave column w
      if (wave < 4) MB_GLDS16_AUX(base + ((size_t)((p.n0 >> 4) + wave * 4 + p.hb) * nmk + jj) * 1024 + lo_ * 16, smem + MINI_OFF + MINI_A + wave * 1024, AUX);
      return;
    }
    if constexpr (HN) {                                  // 128 weight rows: wave w stages rows [16 (w & 1), +16) of half hb of wave column w >> 1
      MB_GLDS16_AUX(base + ((size_t)((p.n0 >> 4) + (wave >> 1) * 4 + p.hb * 2 + (wave & 1)) * nmk + jj) * 1024 + lo_ * 16, smem + MINI_OFF + MINI_A + wave * 1024, AUX);
      return;
    }
#pragma unroll
    for (int jx = 0; jx < 2; ++jx)
      MB_GLDS16_AUX(base + ((size_t)((p.n0 >> 4) + wave + 8 * jx) * nmk + jj) * 1024 + lo_ * 16, smem + MINI_OFF + MINI_A + (wave + 8 * jx) * 1024, AUX);
  };
  // the token operand's scale dword of this lane for mini j: its four m-tiles' bytes of block 2 jj + (lane >= 32) (GemmArgs.lo: lane order)
  auto mini_scale_off = [&](const Plan& p, int j) -> uint32_t {
    const int ps = j >= nmk ? 1 : 0, jj = j - ps * nmk;
    const int gq = PAIR ? p.q * 2 + wm : p.q * 4 + wm * 2 + ps;
    int lo_ = lane; asm volatile("" : "+v"(lo_));
    if constexpr (!PAIR) return ((((uint32_t)(2 * jj + (lo_ >> 5)) * nseq + p.seq) << tps_sh) << 8) + gq * 64 + (lo_ & 15) * 4;   // (grp_bytes = 256 TPS)
    return ((uint32_t)(2 * jj + (lo_ >> 5)) * nseq + p.seq) * grp_bytes + gq * 64 + (lo_ & 15) * 4;
  };
  // the weight operand's scale dword of this lane for mini j: the bytes of its four n-tiles for K-elements [128 jj, 128 jj + 128) (w4_scale_index; one scale
  // per (weight row, mini-tile) since round 6 -- rounds 2-5 held one dword per operand set for the whole K loop)
  auto mini_wscale_off = [&](const Plan& p, int j) -> uint32_t {
    const int ps = j >= nmk ? 1 : 0, jj = j - ps * nmk;
    int lo_ = lane; asm volatile("" : "+v"(lo_));
    return (uint32_t)((((p.n0 >> 6) + wn) * nmk + jj) * 64 + (lo_ & 15) * 4);
  };
  int mws = 0, mxs = 0;                                 // MINI: the current mini's weight / token scale dwords
  auto mini_scale_issue = [&](const Plan& p, int j) {    // inline asm: counted by the K loop's own vmcnt wait, which the registers are tied through
    const int ps = PAIR ? (j >= nmk ? 1 : 0) : 0;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(mxs) : "v"(mini_scale_off(p, j)), "s"(a.lo[ps].a_scale) : "memory");
    asm volatile("global_load_dword %0, %1, %2" : "=v"(mws) : "v"(mini_wscale_off(p, j)), "s"(a.lo[ps].w_scale) : "memory");
  };
  // all of K-tiles 0 and 1 of a tile (both LDS parities must be free)
  // (nk >= 2 is a precondition of this kernel: gemm_ht_supported)
  auto prologue_rest = [&](const Plan& p) {            // 16 instructions per wave
    dma_a(p, 0, 0); dma_b(p, 0, 0); dma_b(p, 0, 1); dma_a(p, 0, 1);
    dma_a(p, 1, 0); dma_b(p, 1, 1); dma_a(p, 1, 1); dma_b(p, 1, 0);
  };
  auto prologue = [&](const Plan& p) { dma_x(p, 0); dma_x(p, 1); prologue_rest(p); if constexpr (MINI) { mini_dma_a(p, 0); mini_dma_b(p, 0); } };

  // ---- fragment read offsets inside a half-tile (rows 128 B, slot swizzled with (row>>1)&7)
  int foff[2], xoffe[2];
  // PAIR: lane row 1 reads X row 1 (the class row's difference operand): foff of lane row 1 = 128 + slot offset = its X-row offset as well
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    foff[ks] = l15 * 128 + (((ks * 4 + g) ^ MB_SWZ(l15)) * 16);
    // class row: only lane row 0 feeds a result that is kept, so the other 15 lane rows read their usual (conflict-free)
    // A-fragment addresses instead of the X buffer -- 16 lanes on the 8 X rows was a 2-way bank conflict on every read
    xoffe[ks] = (l15 == 0 || (PAIR && l15 == 1)) ? 2 * AH_BYTES + 2 * BH_BYTES + foff[ks] : foff[ks];
  }
  const int xbase = wm * (8 * MT) * 128;              // this wave's rows inside an A half-tile
  const int wbase = 2 * AH_BYTES + wn * 32 * 128;     // this wave's rows inside a B half-tile (from the parity base)

#define MB_LOAD_A(H)                                                                            \
  _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      xa[i] = frag_set(xa[i], *(const h16x8*)(par + (H) * AH_BYTES + xbase + i * 16 * 128 + fo[ks]), ks);
#define MB_LOAD_B(H)                                                                            \
  _Pragma("unroll") for (int i = 0; i < (QN ? 1 : 2); ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      wb[H][i] = frag_set(wb[H][i], *(const h16x8*)(par + wbase + (H) * BH_BYTES + i * 16 * 128 + fo[ks]), ks);
#define MB_SYNC_L()                                     \
  __builtin_amdgcn_s_barrier();                         \
  __builtin_amdgcn_sched_barrier(0);                    \
  __builtin_amdgcn_s_setprio(1);
  // one 16x16 output tile x one K-tile: two f16 MFMAs of K = 32
  auto frag_set = [](h16x16 f, h16x8 v, int ks) -> h16x16 {
    return ks == 0 ? __builtin_shufflevector(__builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7), f, 0, 1, 2, 3, 4, 5, 6, 7, 24, 25, 26, 27, 28, 29, 30, 31)
                   : __builtin_shufflevector(f, __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 16, 17, 18, 19, 20, 21, 22, 23);
  };
  auto mma_tile = [&](f32x4 c, const h16x16& w, const h16x16& x) -> f32x4 {
    c = MB_MFMA_16x16x32(__builtin_shufflevector(w, w, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(x, x, 0, 1, 2, 3, 4, 5, 6, 7), c);
    return MB_MFMA_16x16x32(__builtin_shufflevector(w, w, 8, 9, 10, 11, 12, 13, 14, 15), __builtin_shufflevector(x, x, 8, 9, 10, 11, 12, 13, 14, 15), c);
  };
#define MB_MMA_END                                                                                  \
  __builtin_amdgcn_s_setprio(0);                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_barrier();
#define MB_MMA(AH, BH) MB_MMA_DO(AH, BH) MB_MMA_END
  // experiment (round 6, -DMB_MERGE_PHASES): FOUR barrier crossings per K-tile instead of eight -- the barrier pairs between phases 0 / 1 and between phases
  // 2 / 3 left out, so a wave group's slots are [L0] [M0 L1 M1] [L2] [M2 L3 M3]; the partner group stays one barrier behind
#ifdef MB_MERGE_PHASES
#define MB_MID_END __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0);
#define MB_MID_SYNC() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1);
#else
#define MB_MID_END MB_MMA_END
#define MB_MID_SYNC() MB_SYNC_L()
#endif
  // one mini-tile: 4 m-tiles of accumulator half AH x 4 n-tiles, one scaled MFMA of K = 128 each; op_sel picks the n-tile's / m-tile's scale byte
#define MB_MINI_ONE(AH, N, I)                                                                       \
  acc[N][(AH) * MH + (I)] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                        \
      __builtin_shufflevector(mwb[N], mwb[N], 0, 1, 2, 3, -1, -1, -1, -1), __builtin_shufflevector(mxa[I], mxa[I], 0, 1, 2, 3, -1, -1, -1, -1), \
      acc[N][(AH) * MH + (I)], 4, 4, N, mwsel, I, mxs);
#define MB_MINI_MMA_QN(AH)                                                                          \
  MB_MINI_ONE(AH, 0, 0) MB_MINI_ONE(AH, 0, 1) MB_MINI_ONE(AH, 0, 2) MB_MINI_ONE(AH, 0, 3)           \
  _Pragma("unroll") for (int i = 0; i < MH; ++i) asm volatile("" : "+v"(acc[0][(AH) * MH + i]));
#define MB_MINI_MMA_HN(AH)                                                                          \
  MB_MINI_ONE(AH, 0, 0) MB_MINI_ONE(AH, 1, 0) MB_MINI_ONE(AH, 0, 1) MB_MINI_ONE(AH, 1, 1)           \
  MB_MINI_ONE(AH, 0, 2) MB_MINI_ONE(AH, 1, 2) MB_MINI_ONE(AH, 0, 3) MB_MINI_ONE(AH, 1, 3)           \
  _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int n = 0; n < 2; ++n)      \
    asm volatile("" : "+v"(acc[n][(AH) * MH + i]));
#define MB_MINI_MMA(AH)                                                                             \
  MB_MINI_ONE(AH, 0, 0) MB_MINI_ONE(AH, 1, 0) MB_MINI_ONE(AH, 2, 0) MB_MINI_ONE(AH, 3, 0)           \
  MB_MINI_ONE(AH, 0, 1) MB_MINI_ONE(AH, 1, 1) MB_MINI_ONE(AH, 2, 1) MB_MINI_ONE(AH, 3, 1)           \
  MB_MINI_ONE(AH, 0, 2) MB_MINI_ONE(AH, 1, 2) MB_MINI_ONE(AH, 2, 2) MB_MINI_ONE(AH, 3, 2)           \
  MB_MINI_ONE(AH, 0, 3) MB_MINI_ONE(AH, 1, 3) MB_MINI_ONE(AH, 2, 3) MB_MINI_ONE(AH, 3, 3)           \
  _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int n = 0; n < 4; ++n)      \
    asm volatile("" : "+v"(acc[n][(AH) * MH + i]));
#define MB_MMA_DO(AH, BH)                                                                           \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < MH; ++i) _Pragma("unroll") for (int n = 0; n < (QN ? 1 : 2); ++n)  \
      acc[(BH) * 2 + n][(AH) * MH + i] = mma_tile(acc[(BH) * 2 + n][(AH) * MH + i], wb[BH][n], xa[i]); \
  }

  Plan cur;
  int vb = blockIdx.x;
  make_plan(vb, cur);
  auto scales_load = [&](const Plan& p) {             // plain loads; scales_pack() after a vmcnt(0) that the loaded registers are tied through
    int lo_ = lane; asm volatile("" : "+v"(lo_));
    (void)lo_;
    mws = *(const int*)(a.lo[0].w_scale + mini_wscale_off(p, 0));          // (mini 0 of a tile comes with the prologue)
    mxs = *(const int*)(a.lo[0].a_scale + mini_scale_off(p, 0));
  };
  auto scales_pack = [&]() {
    if constexpr (MINI) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(mws), "+v"(mxs) :: "memory"); return; }
  };
  if constexpr (MINI) scales_load(cur);
  prologue(cur);
  if constexpr (MINI) scales_pack();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // K-tiles 0 and 1 of the first tile are in LDS for everyone

  [[maybe_unused]] int trace_it = 0;
  while (true) {
    MB_TRACE(0);
    f32x4 acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acce[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // SEQ: class row x this wave row's 2 n-tiles
    // every tile of a sequence (pair) stages the class row(s), only the last one stores them -- the others skip their MFMAs
#ifdef MB_NO_CLS                                            /* experiment (timing only): what the class-token rows' MFMAs and fragment reads cost */
    const bool cls_on = false;
#else
    const bool cls_on = __builtin_amdgcn_readfirstlane(cur.q) == TPS - 1;
#endif
    h16x16 xa[MH], wb[2][2];      // wb[0] (the B0 fragments) is kept from phase 0 to phase 3: every operand fragment is read once per K-tile

    if (grp == 1) __builtin_amdgcn_s_barrier();        // stagger: group 1 runs one barrier behind
  // trace builds only (tools/ht_trace.py, HT_DEFS): leave parts of the mini-tile machinery out -- results are garbage, timing only
#ifdef MB_MINI_NO_PHASE
#define MB_MINI_PHASE_ON false
#else
#define MB_MINI_PHASE_ON true
#endif
#ifdef MB_MINI_NO_DMA
#define MB_MINI_X_DMA(...)
#else
#define MB_MINI_X_DMA(...) __VA_ARGS__
#endif
#ifdef MB_MINI_NO_READ
#define MB_MINI_X_READ(...) _Pragma("unroll") for (int n = 0; n < 4; ++n) { mwb[n] = i32x4{lo_, lo_, lo_, lo_}; mxa[n] = i32x4{lo_, lo_, lo_, lo_}; }
#else
#define MB_MINI_X_READ(...) __VA_ARGS__
#endif
#ifdef MB_MINI_NO_MMA
#define MB_MINI_X_MMA(...) asm volatile("" :: "v"(mwb[0]), "v"(mwb[1]), "v"(mwb[2]), "v"(mwb[3]), "v"(mxa[0]), "v"(mxa[1]), "v"(mxa[2]), "v"(mxa[3]), "v"(mwsel), "v"(mxs));
#else
#define MB_MINI_X_MMA(...) __VA_ARGS__
#endif
#define MB_KTILE MB_KTILE_(0)
#define MB_KTILE_(MAH)              /* MAH: plain tiles with mini-tiles: the accumulator half this K loop's mini-tiles update (compile time) */ \
    {                                                                                                    \
      const char* par = smem + (t & 1) * PAR_BYTES; \
      int fo[2], xo[2]; \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { fo[ks] = foff[ks]; xo[ks] = xoffe[ks]; } \
      /* DMA issue is placed where the read phase is short (a global_load_lds blocks the issuing wave until the address unit takes */ \
      /* it): none in phase 0 (12 fragment reads), A1(t+1) in phase 1, A0(t+2) in phase 2, B1, X and B0 of K-tile t+2 in phase 3 */ \
      /* (no reads).  Every half-tile is re-filled >= 2 phases after its last reader; K-tile 1 came with the prologue. */ \
      const bool n1 = t >= 1 && t + 1 < nk, n2 = t + 2 < nk; \
      /* ---- phase 0: quadrant (A0, B0) [+ class row x B0 for wave row 0] */ \
      MB_LOAD_B(0) MB_LOAD_A(0)                          /* B first: the first MFMAs need both B fragments and only xa[0] */ \
      h16x16 xe; \
      if (SEQ && wm == 0 && cls_on) { xe = frag_set(xe, *(const h16x8*)(par + xo[0]), 0); xe = frag_set(xe, *(const h16x8*)(par + xo[1]), 1); } \
      MB_SYNC_L() \
      if (SEQ && wm == 0 && cls_on) { _Pragma("unroll") for (int n = 0; n < (QN ? 1 : 2); ++n) acce[n] = mma_tile(acce[n], wb[0][n], xe); } \
      MB_MMA_DO(0, 0) MB_MID_END \
      /* ---- phase 1: (A0, B1) [+ class row x B1 for wave row 1]; refill A1 of the other parity with K-tile t+1 */ \
      if constexpr (!HN) { MB_LOAD_B(1) } \
      if (!HN && SEQ && wm == 1 && cls_on) { xe = frag_set(xe, *(const h16x8*)(par + xo[0]), 0); xe = frag_set(xe, *(const h16x8*)(par + xo[1]), 1); } \
      /* MINI: the next mini-tile's scale dword and A part go out first (older than everything the phase-3 wait leaves in flight) */ \
      const bool mini_issue = MINI && t >= 1 && (mini_every || !(t & 1)); \
      const int mini_j = mini_every ? t : (t >> 1); \
      if (MINI && mini_issue) { mini_scale_issue(cur, mini_j); MB_MINI_X_DMA(mini_dma_a(cur, mini_j);) } \
      if (n1) dma_a(cur, t + 1, 1); \
      MB_MID_SYNC() \
      if (!HN && SEQ && wm == 1 && cls_on) { _Pragma("unroll") for (int n = 0; n < 2; ++n) acce[n] = mma_tile(acce[n], wb[1][n], xe); } \
      if constexpr (!HN) { MB_MMA_DO(0, 1) } MB_MMA_END \
      /* ---- phase 2: (A1, B1); refill A0 of this parity with K-tile t+2 */ \
      MB_LOAD_A(1) \
      if (MINI && mini_issue) { MB_MINI_X_DMA(mini_dma_b(cur, mini_j);) } \
      if (n2) dma_a(cur, t + 2, 0); \
      MB_SYNC_L() if (!HN) { MB_MMA_DO(1, 1) } MB_MID_END \
      /* ---- phase 3: (A1, B0), B0 still in registers: no LDS reads; refill B1, X and B0 of this parity with K-tile t+2; K-tile t+1 must have landed. */ \
      /* In K-tile 0 nothing is waited for: K-tile 1 arrived with the prologue, and the previous tile's output */ \
      /* stores stay in flight until the wait of K-tile 1. */ \
      if (n2) { dma_b(cur, t + 2, 1); dma_x(cur, t + 2); dma_b(cur, t + 2, 0); } \
      if (t >= 1) { \
        if (MINI) {                               /* the mini-tile's scale dword (requested in phase 1) is older than those: tied through, */ \
          /* ONE asm statement for the three cases: with one statement per case the compiler copied the still-in-flight register to its */ \
          /* loop-carried home AHEAD of two of the waits (v_mov ... ; s_waitcnt) -- a stale scale whenever the load was slow (round 2) */ \
          const int wsel = __builtin_amdgcn_readfirstlane(n2 ? (wave == 7 ? 2 : 1) : 0); \
          if constexpr (QN)                              /* (quarter-column tiles: A0, (X,) B0 of K-tile t+2 stay in flight: 3 / 4 instructions) */ \
            asm volatile("s_cmp_eq_u32 %[w], 0\n\ts_cbranch_scc1 1f\n\ts_cmp_eq_u32 %[w], 1\n\ts_cbranch_scc1 2f\n\ts_waitcnt vmcnt(4)\n\ts_branch 3f\n" \
                         "1:\n\ts_waitcnt vmcnt(0)\n\ts_branch 3f\n2:\n\ts_waitcnt vmcnt(3)\n3:" \
                         : "+v"(mxs), "+v"(mws) : [w] "s"(wsel) : "memory", "scc"); \
          else if constexpr (HN)                         /* (half-column tiles: 4 / 5 instructions) */ \
            asm volatile("s_cmp_eq_u32 %[w], 0\n\ts_cbranch_scc1 1f\n\ts_cmp_eq_u32 %[w], 1\n\ts_cbranch_scc1 2f\n\ts_waitcnt vmcnt(5)\n\ts_branch 3f\n" \
                         "1:\n\ts_waitcnt vmcnt(0)\n\ts_branch 3f\n2:\n\ts_waitcnt vmcnt(4)\n3:" \
                         : "+v"(mxs), "+v"(mws) : [w] "s"(wsel) : "memory", "scc"); \
          else \
          asm volatile("s_cmp_eq_u32 %[w], 0\n\ts_cbranch_scc1 1f\n\ts_cmp_eq_u32 %[w], 1\n\ts_cbranch_scc1 2f\n\ts_waitcnt vmcnt(7)\n\ts_branch 3f\n" \
                       "1:\n\ts_waitcnt vmcnt(0)\n\ts_branch 3f\n2:\n\ts_waitcnt vmcnt(6)\n3:" \
                       : "+v"(mxs), "+v"(mws) : [w] "s"(wsel) : "memory", "scc"); \
        } else if (n2) {                                 /* A0, B1, (X,) B0 of K-tile t+2 may stay in flight */ \
          if (SEQ && wave == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); \
          else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
      } \
      MB_MID_SYNC() MB_MMA_DO(1, 0) MB_MMA_END \
      /* ---- phase 4 (MINI): the mini-tile that landed under this (or the previous) K-tile's wait x the accumulators of its 128 token rows */ \
      if constexpr (MINI) { \
        if (MB_MINI_PHASE_ON && (mini_every || (t & 1))) { \
          const int mj = mini_every ? t : (t >> 1); \
          int lo_ = lane; \
          asm volatile("" : "+v"(lo_)); \
          const int mfo = (lo_ & 15) * 64 + (((lo_ >> 4) ^ ((lo_ >> 1) & 3)) * 16); \
          const char* mbuf = smem + MINI_OFF; \
          i32x4 mxa[4], mwb[4]; \
          MB_MINI_X_READ( \
          _Pragma("unroll") for (int n = 0; n < NTW; ++n) mwb[n] = *(const i32x4*)(mbuf + MINI_A + (wn * (64 / NS) + n * 16) * 64 + mfo); \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) mxa[i] = *(const i32x4*)(mbuf + (wm * 64 + i * 16) * 64 + mfo); ) \
          const int mwsel = HN ? (int)((uint32_t)mws >> ((32 / NS) * cur.hb)) : mws;   /* (column-split tiles: the bytes of this tile's n-tiles) */ \
          MB_SYNC_L() \
          MB_MINI_X_MMA(if constexpr (QN) { MB_MINI_MMA_QN(MAH) } else if constexpr (HN) { MB_MINI_MMA_HN(MAH) } else { MB_MINI_MMA(MAH) }) \
          /* (v_mfma_scale results must not be read by a VALU copy too early, see the class-row blocks above) */ \
          asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); \
          MB_MMA_END \
        } \
      } \
    }
    {
      int t = 0;
      if constexpr (MINI && !PAIR) {        // plain tiles: mini-tile t belongs to K-tile t; the first K / 128 update rows 0..127 of the tile, the others rows 128..255
        for (; t < nk / 2; ++t) MB_KTILE_(0)
        for (; t < nk; ++t) MB_KTILE_(1)
      } else {
        for (; t < nk; ++t) MB_KTILE
      }
    }
#undef MB_KTILE
#undef MB_KTILE_
#undef MB_MINI_X_DMA
#undef MB_MINI_X_READ
#undef MB_MINI_X_MMA
    if (grp == 0) __builtin_amdgcn_s_barrier();        // balance the barrier count of the two groups
    MB_TRACE(1);

    const float* __restrict__ resp = a.residual;
    float* __restrict__ out32 = a.out_f32;
    h16* __restrict__ out16 = a.out_h16;
    const int m0 = cur.m0, n0 = cur.n0, clsrow = cur.cls, tq = cur.q;
    constexpr int NROWS = SEQ ? MT + 1 : MT;
    int l15e = l15, ge = g;                              // opaque copies (see make_plan): no per-row address tables
    asm volatile("" : "+v"(l15e), "+v"(ge));             // carried in VGPRs through the main loop
    auto row_of = [&](int r) {
      if (PAIR) return r < MT ? m0 + (r / MH) * a.pair_rows + wm * 64 + (r % MH) * 16 + l15e : clsrow + (l15e == 1 ? a.pair_rows : 0);
      return r < MT ? m0 + wm * (16 * MT) + (r / MH) * (8 * MT) + (r % MH) * 16 + l15e : clsrow;
    };
    auto col_of = [&](int r, int nt) {
      if (HN) return n0 + wn * 64 + cur.hb * (64 / NS) + (nt & (NTW - 1)) * 16 + ge * 4;      // (this tile's block of the wave's columns)
      return r < MT ? n0 + wn * 64 + (nt >> 1) * 32 + (nt & 1) * 16 + ge * 4 : n0 + wn * 64 + wm * 32 + nt * 16 + ge * 4;
    };
    auto row_ok = [&](int r) {
      if (PAIR) return r < MT ? true : (l15e < 2 && tq == TPS - 1);
      return r < MT ? row_of(r) < (SEQ ? m0 + 256 : a.M) : (l15e == 0 && (!HN || wm == 0) && tq == TPS - 1);
    };
    // ---- next tile: start its first two K-tiles NOW (all LDS is free), so they fly during the epilogue math.
    // The bias of THIS tile is fetched in between by inline-asm loads the compiler does not track: vmcnt retires in
    // order, so one counted wait for "everything but the 16 DMA instructions issued after them" releases the bias
    // while that DMA is still flying (the class-row DMA of wave 7 goes first to keep the count uniform).
    const int nvb = vb + gridDim.x;
    const bool has_next = PERSIST && nvb < ntiles;
    Plan nxt;
    if (has_next) { make_plan(nvb, nxt); dma_x(nxt, 0); dma_x(nxt, 1); }
    f32x4 bias4[4], bcls[2];           // bcls: class-token row (indexing bias4 by wave id would put the array in scratch)
    if constexpr (MINI) {
      // The mini-tile kernels run at the 256-VGPR limit, where the allocator may move registers around: an asm load whose result it does
      // not track could be copied while still in flight.  Plain loads here (the compiler waits for them; the overlap with the next
      // tile's prologue DMA is given up in this mode: <= 0.6 us per tile, profiles/r04_gemm_minitiles.md section 9).
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bias4[nt] = *(const f32x4*)(a.bias + col_of(0, nt));
      if (SEQ) { bcls[0] = *(const f32x4*)(a.bias + col_of(MT, 0)); bcls[1] = *(const f32x4*)(a.bias + col_of(MT, 1)); }
      else { bcls[0] = bias4[0]; bcls[1] = bias4[1]; }
      if (has_next) scales_load(nxt);                                        // this tile's K loops are over: the scale registers are free
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias4[0]), "+v"(bias4[1]), "+v"(bias4[2]), "+v"(bias4[3]), "+v"(bcls[0]), "+v"(bcls[1]) :: "memory");
      if (has_next) scales_pack();
      if (has_next) { prologue_rest(nxt); if constexpr (MINI) { mini_dma_a(nxt, 0); mini_dma_b(nxt, 0); } }
    } else {
#define MB_LDG16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) MB_LDG16(bias4[nt], a.bias + col_of(0, nt));
    if (SEQ) { MB_LDG16(bcls[0], a.bias + col_of(MT, 0)); MB_LDG16(bcls[1], a.bias + col_of(MT, 1)); }
    else { bcls[0] = bias4[0]; bcls[1] = bias4[1]; }
#undef MB_LDG16
    if (has_next) prologue_rest(nxt);
    {
      // ONE asm statement with the registers tied through it (separate per-case waits made the compiler copy the
      // still-in-flight registers ahead of the wait): no next tile -> nothing was issued after the bias -> vmcnt(0).
      const int hn = __builtin_amdgcn_readfirstlane(has_next ? 1 : 0);
      asm volatile("s_cmp_lg_u32 %[hn], 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(0)\n1:\n\ts_waitcnt vmcnt(16)"
                   : "+v"(bias4[0]), "+v"(bias4[1]), "+v"(bias4[2]), "+v"(bias4[3]), "+v"(bcls[0]), "+v"(bcls[1])
                   : [hn] "s"(hn) : "memory", "scc");
    }
    }

    // ---- epilogue: acc[nt][mt] holds out[m][n..n+3] (m = ..+l15, n = ..+g*4); wave rows: half h = mt / MH, tile i = mt % MH.
    // Row r = MT is the class-token row of a sequence-aligned tile (valid in lanes l15 == 0 only).
    // Pass 1 (arithmetic, in place): + bias (+ residual, fetched one m-tile ahead of its use) (+ GELU).
    // Pass 2 (after the DMA wait): stores; fp16 results of two neighbouring n-tiles are exchanged between lane
    // rows g and g^1 with v_permlane16_swap so that a lane stores 8 consecutive columns (16 B; the tail is issue-bound).
    if constexpr (PAIR) {
      // pair tiles: m-tiles 0..MH-1 hold acc_c, MH..MT-1 acc_delta of the same tokens -> (v_c, v_u = v_c + delta) or, with GELU, (h_c, h_u - h_c)
      const float osc = a.scale ? *a.scale : 1.0f;
      auto pair2 = [&](f32x4& c, f32x4& dl, const f32x4& b) {
        f32x4 vc = __builtin_elementwise_fma(c, (f32x4)(osc), b);
        f32x4 vu = __builtin_elementwise_fma(dl, (f32x4)(osc), vc);
        if (EPI == EPI_GELU_H16 || EPI == EPI_GELU_F32) {
          const f32x2 c0 = gelu_erf2((f32x2){vc[0], vc[1]}), c1 = gelu_erf2((f32x2){vc[2], vc[3]});
          const f32x2 u0 = gelu_erf2((f32x2){vu[0], vu[1]}), u1 = gelu_erf2((f32x2){vu[2], vu[3]});
          vc = f32x4{c0.x, c0.y, c1.x, c1.y};
          vu = f32x4{u0.x - c0.x, u0.y - c0.y, u1.x - c1.x, u1.y - c1.y};
        }
        c = vc; dl = vu;
        asm volatile("" : "+v"(c), "+v"(dl));
      };
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) pair2(acc[nt][i], acc[nt][i + MH], bias4[nt]);
      // class rows: lane row 0 = acc_c, lane row 1 = acc_delta of the same 4 features (the lane above in the same lane group)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        f32x4 cup;                                             // the conditional accumulator, as seen by lane row 1
#pragma unroll
        for (int e = 0; e < 4; ++e) cup[e] = __shfl_up(acce[n][e], 1);
        f32x4 c = l15e == 1 ? cup : acce[n], dl = l15e == 1 ? acce[n] : f32x4{0.f, 0.f, 0.f, 0.f};
        pair2(c, dl, bcls[n]);
        acce[n] = l15e == 1 ? dl : c;
      }
    } else {
      const float osc = a.scale ? *a.scale : 1.0f;          // (pre-scaled weights: undo their power of two)
#pragma unroll
      for (int r = 0; r < NROWS; ++r) {
        const int nn = r < MT ? NTW : (QN ? 1 : 2);
#pragma unroll
        for (int nt = 0; nt < nn; ++nt) {
          f32x4& c = r < MT ? acc[nt][r < MT ? r : 0] : acce[nt & 1];
          c = __builtin_elementwise_fma(c, (f32x4)(osc), r < MT ? bias4[nt] : bcls[nt & 1]);
          if (EPI == EPI_GELU_H16 || EPI == EPI_GELU_F32) {
            const f32x2 lo = gelu_erf2((f32x2){c[0], c[1]}), hi = gelu_erf2((f32x2){c[2], c[3]});
            c[0] = lo.x; c[1] = lo.y; c[2] = hi.x; c[3] = hi.y;
          }
          asm volatile("" : "+v"(c));   // pin: the arithmetic stays ABOVE the DMA wait below (it is what hides that latency)
        }
      }
    }
    MB_TRACE(2);
    // The next tile's first K-tiles must be in LDS before anyone reads them.  Every wave waits for ITS share of that DMA here, while nothing of this
    // tile is stored yet (so the wait does not cover store acknowledgements), then issues its stores, and the workgroup barrier comes AFTER the stores
    // (round 3; it used to stand here): with the GELU epilogue the two waves of a SIMD finish their arithmetic 3.8 us apart, and the leading
    // group's stores now go out under the trailing group's arithmetic instead of after it.
    if (has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MB_TRACE(3);
    if constexpr (SEQ && EPI == EPI_GELU_H16) {
      if (a.out4) {
        // e2m1 copy of the (conditional) GELU outputs for the next GEMM's weight-correction mini-tiles: this wave's 64 columns of a row are one scale
        // block (4 n-tiles x 4 lane groups x 4 columns), and the four m-tiles of a lane in one 64-row group are one dword of the lane-ordered scale
        // array (GemmArgs.lo); class-token rows are skipped (they take no part in those passes)
        const uint32_t blk = (uint32_t)(n0 >> 6) + wn;
        const uint32_t nsq = (uint32_t)(PAIR ? a.pair_rows : a.M) / (uint32_t)SQ;
        const uint32_t ngrp = (uint32_t)TPS * (PAIR ? 2 : 4);
        // a lane's four n-tiles are 2 bytes each (4 columns): as 2-byte stores they cost 13.6 us of a 372 us FFN-up launch (A/B, MB_NO_OUT4_STORE).
        // Two lane exchanges give every lane 8 CONSECUTIVE bytes of the row's 32-byte block instead: the odd / even lane-row swap of the fp16
        // stores (n-tiles 2pr <-> 2pr + 1: a lane then holds 8 columns of one n-tile), then the lane halves (g <-> g ^ 2: the neighbouring 8 columns)
        auto emit4 = [&](uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint8_t* dst, uint32_t row, [[maybe_unused]] float mul) {
          const auto sw = __builtin_amdgcn_permlane16_swap((p0 & 0xffffu) | (p2 << 16), (p1 & 0xffffu) | (p3 << 16), false, false);
          const uint32_t q0 = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);      // 32-column group 0: this lane's 4 columns | its neighbour's
          const uint32_t q1 = __builtin_amdgcn_perm(sw[1], sw[0], 0x07060302u);      // 32-column group 1
          const auto sx = __builtin_amdgcn_permlane32_swap(q0, q1, false, false);    // lanes < 32: group 0 of lanes g, g + 2; lanes >= 32: group 1 of g - 2, g
#ifdef MB_NO_OUT4_STORE                                     /* experiment (timing only): what the stores of the e2m1 copy cost */
          if (mul == 12345.0f)
#endif
          *(uint2*)(dst + (size_t)row * 2 * a.N + ((n0 + wn * 64) >> 1) + (ge >> 1) * 16 + (ge & 1) * 8) = make_uint2(sx[0], sx[1]);
        };
#pragma unroll
        for (int hh = 0; hh < (PAIR ? 1 : 2); ++hh) {
          uint32_t sc4 = 0, sl4 = 0;
#pragma unroll
          for (int ii = 0; ii < MH; ++ii) {
            const int i = hh * MH + ii;
            float am = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
              for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(acc[nt][i][e]));
            am = rows_max(am);
            const float mul = fp4_scale_mul_nosat(am);
            const uint32_t row = (uint32_t)row_of(i);
            sc4 |= fp4_scale_byte_nosat(am) << (8 * ii);
            emit4(fp4_pack4(acc[0][i][0], acc[0][i][1], acc[0][i][2], acc[0][i][3], mul), fp4_pack4(acc[1][i][0], acc[1][i][1], acc[1][i][2], acc[1][i][3], mul),
                  fp4_pack4(acc[2][i][0], acc[2][i][1], acc[2][i][2], acc[2][i][3], mul), fp4_pack4(acc[3][i][0], acc[3][i][1], acc[3][i][2], acc[3][i][3], mul),
                  a.out4, row, mul);
            if (a.out4l) {                                   // (precision 4) the fp16 lo halves of the same values: what the fp16 store below rounds away
              f32x4 l[4];
              float aml = 0.f;
#pragma unroll
              for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) { l[nt][e] = acc[nt][i][e] - (float)to_h(acc[nt][i][e]); aml = fmaxf(aml, fabsf(l[nt][e])); }
              aml = rows_max(aml);
              const float mull = fp4_scale_mul_nosat(aml);
              sl4 |= fp4_scale_byte_nosat(aml) << (8 * ii);
              emit4(fp4_pack4(l[0][0], l[0][1], l[0][2], l[0][3], mull), fp4_pack4(l[1][0], l[1][1], l[1][2], l[1][3], mull),
                    fp4_pack4(l[2][0], l[2][1], l[2][2], l[2][3], mull), fp4_pack4(l[3][0], l[3][1], l[3][2], l[3][3], mull), a.out4l, row, mull);
            }
          }
          const uint32_t gq = PAIR ? (uint32_t)tq * 2 + wm : (uint32_t)tq * 4 + wm * 2 + hh;
          if (ge == 0) ((uint32_t*)a.out4_scale)[((blk * nsq + (uint32_t)cur.seq) * ngrp + gq) * 16 + l15e] = sc4;
          if (ge == 0 && a.out4l) ((uint32_t*)a.out4l_scale)[((blk * nsq + (uint32_t)cur.seq) * ngrp + gq) * 16 + l15e] = sl4;
        }
      }
    }
    if (EPI == EPI_RES_F32) {
      // fp32 + residual: two sweeps over the rows, each covering ONE full 128-byte line per row (n-tiles 2p, 2p+1), the
      // residual (and, for the LayerNorm form, the row's {mean, rstd}) fetched one row ahead of its use.  With ln_stats the
      // buffer holds the pre-LayerNorm rows and the normalised residual is re-derived here (GemmArgs); in place is fine:
      // every element is read and written by the same lane.
      const bool lnres = a.ln_stats != nullptr;
      // Order of the memory operations (round 3).  Loads and stores share the in-order vmcnt counter, so the old "fetch row r + 1, store row r" rhythm
      // made every wait for a residual row also wait for the ACKNOWLEDGEMENT of the store two rows back (tools/micro/store_path.hip: 19 us per tile
      // interleaved, 15 / 12 / 9 us with 4 / 8 / 16 loads batched ahead of their stores).  Now: batches of two rows, the next batch's loads issued
      // BEFORE the current batch's stores: 22.8 -> 19.5 us per tile.
      constexpr int RB = (PAIR && MINI) ? 1 : 2;        // rows per batch: 4 float4 per lane -- what fits next to 136 accumulator registers without spilling (2 in the fp4 pair kernel)
      // (Measured with larger / growing batches -- 4 rows, or 2 + 2 + 4 rows then a whole sweep into the registers the first sweep vacated: 16-42 spilled
      // VGPRs in the pair kernels and no gain in the kernels that did not spill: with every CU in its epilogue at once the pass runs at the chip's
      // ~6.5-6.9 TB/s of mixed read + write traffic, profiles/r03_power_and_streams.md section 6.)
      constexpr int NB = MT / RB;                       // batches per sweep
      static_assert(MT % RB == 0, "whole batches");
      constexpr int NQ = QN ? 1 : 2;                      // n-tiles per sweep (quarter-column tiles: one, 64-byte row segments)
      float4 rv[RB][2];
      float2 sv[RB] = {};
      auto fetch = [&](int k) {                         // batch k: sweep p = k / NB (n-tiles 2p, 2p + 1: one 128-byte line per row), rows (k % NB) * RB ..
        const int p = k / NB, r0 = (k % NB) * RB;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const uint32_t rowc = (uint32_t)min(row_of(r0 + j), a.M - 1);                // clamped: always a legal address
#pragma unroll
          for (int q = 0; q < NQ; ++q) rv[j][q] = *(const float4*)((const char*)resp + (size_t)((rowc * (uint32_t)a.N + (uint32_t)col_of(r0 + j, 2 * p + q)) * 4u));
          if (lnres) sv[j] = *(const float2*)(a.ln_stats + 2 * (size_t)rowc);
        }
      };
      auto add_one = [&](f32x4& c, const float4& rr, const float2& st, const f32x4& gm, const f32x4& bt) {
        if (lnres) {
          c[0] += ln_affine(rr.x, st.x, st.y, gm[0], bt[0]); c[1] += ln_affine(rr.y, st.x, st.y, gm[1], bt[1]);
          c[2] += ln_affine(rr.z, st.x, st.y, gm[2], bt[2]); c[3] += ln_affine(rr.w, st.x, st.y, gm[3], bt[3]);
        } else { c[0] += rr.x; c[1] += rr.y; c[2] += rr.z; c[3] += rr.w; }
      };
      f32x4 gm[2] = {}, bt[2] = {};
      auto affine = [&](int p) {
        if (lnres) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) { gm[q] = *(const f32x4*)(a.ln_g + col_of(0, 2 * p + q)); bt[q] = *(const f32x4*)(a.ln_b + col_of(0, 2 * p + q)); }
        }
      };
      auto add = [&](int k) {
        const int p = k / NB, r0 = (k % NB) * RB;
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
          for (int q = 0; q < NQ; ++q) add_one(acc[2 * p + q][r0 + j], rv[j][q], sv[j], gm[q], bt[q]);
      };
      auto store_one = [&](const f32x4& c, int r, int nt) {
        *(float4*)((char*)out32 + (size_t)(((uint32_t)row_of(r) * (uint32_t)a.N + (uint32_t)col_of(r, nt)) * 4u)) = make_float4(c[0], c[1], c[2], c[3]);
      };
      auto store = [&](int k) {
        const int p = k / NB, r0 = (k % NB) * RB;
#pragma unroll
        for (int j = 0; j < RB; ++j)
          if (row_ok(r0 + j)) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) store_one(acc[2 * p + q][r0 + j], r0 + j, 2 * p + q);
          }
      };
      // L(0) | add(0) L(1) | S(0) add(1) L(2) | S(1) add(2) L(3) | ... : the wait for L(k + 1) leaves S(k) in flight
      affine(0);
      fetch(0);
      constexpr int NSWEEP = HN ? 1 : 2;                  // (half-column tiles: one sweep, n-tiles 0 and 1)
#pragma unroll
      for (int k = 0; k < NSWEEP * NB; ++k) {
        asm volatile("" ::: "memory");
        add(k);                                           // (waits for L(k); in place on the accumulators: the load registers are free again)
        if (k + 1 < NSWEEP * NB) { if ((k + 1) % NB == 0) affine((k + 1) / NB); fetch(k + 1); }
        asm volatile("" ::: "memory");
        store(k);
      }
      if (SEQ) {                       // class-token row: this wave row's two n-tiles, lanes l15 == 0 (pair tiles: l15 == 1 = the twin's)
        float4 rr[2]; float2 st = {};
        const uint32_t rowc = (uint32_t)min(row_of(MT), a.M - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          rr[q] = *(const float4*)((const char*)resp + (size_t)((rowc * (uint32_t)a.N + (uint32_t)col_of(MT, q)) * 4u));
          if (lnres) { gm[q] = *(const f32x4*)(a.ln_g + col_of(MT, q)); bt[q] = *(const f32x4*)(a.ln_b + col_of(MT, q)); }
        }
        if (lnres) st = *(const float2*)(a.ln_stats + 2 * (size_t)rowc);
#pragma unroll
        for (int q = 0; q < NQ; ++q) add_one(acce[q], rr[q], st, gm[q], bt[q]);
        if (row_ok(MT)) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) store_one(acce[q], MT, q);
        }
      }
    } else {
      float satm = 0.f;                                  // fp16 epilogues: largest |value| this lane stores (saturation census, GemmArgs.sat);
      bool satnan = false;                               // ... and whether one of them is a NaN (fmaxf returns the other operand)
#pragma unroll
      for (int r = 0; r < NROWS; ++r) {
        const bool ok = row_ok(r);
        const int nn = r < MT ? 4 : 2;
        if (EPI == EPI_H16 || EPI == EPI_GELU_H16) {
#pragma unroll
          for (int pr = 0; pr < nn / 2; ++pr) {
            const f32x4 ca = r < MT ? acc[2 * pr][r < MT ? r : 0] : acce[0];
            const f32x4 cb = r < MT ? acc[2 * pr + 1][r < MT ? r : 0] : acce[1];
            if (a.sat && ok) {
              satm = fmaxf(fmaxf(satm, fmaxf(fmaxf(fabsf(ca[0]), fabsf(ca[1])), fmaxf(fabsf(ca[2]), fabsf(ca[3])))),
                           fmaxf(fmaxf(fabsf(cb[0]), fabsf(cb[1])), fmaxf(fabsf(cb[2]), fabsf(cb[3]))));
              const float sm = (ca[0] + ca[1]) + (ca[2] + ca[3]) + (cb[0] + cb[1]) + (cb[2] + cb[3]);   // NaN if any of the eight is (or +inf next to -inf)
              satnan |= sm != sm;
            }
            const h16x2 ta0 = {to_h(ca[0]), to_h(ca[1])}, ta1 = {to_h(ca[2]), to_h(ca[3])};
            const h16x2 tb0 = {to_h(cb[0]), to_h(cb[1])}, tb1 = {to_h(cb[2]), to_h(cb[3])};
            // swap: lane rows with odd g of the first operand <-> even g of the second (16-lane rows)
            const auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, ta0), __builtin_bit_cast(uint32_t, tb0), false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, ta1), __builtin_bit_cast(uint32_t, tb1), false, false);
            // even g: 8 columns of n-tile 2pr starting at (g/2)*8;  odd g: the same 8 columns of n-tile 2pr+1
            const int n = col_of(r, 2 * pr + (ge & 1)) - (ge & 1) * 4;
            const size_t ob = (size_t)(((uint32_t)row_of(r) * (uint32_t)a.N + (uint32_t)n) * 2u);
            if (ok) *(uint4*)((char*)out16 + ob) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
        } else if (ok) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            if (nt < nn) {
              const f32x4 c = r < MT ? acc[nt][r < MT ? r : 0] : acce[nt & 1];
              *(float4*)((char*)out32 + (size_t)(((uint32_t)row_of(r) * (uint32_t)a.N + (uint32_t)col_of(r, nt)) * 4u)) = make_float4(c[0], c[1], c[2], c[3]);
            }
        }
      }
      if ((EPI == EPI_H16 || EPI == EPI_GELU_H16) && a.sat && (satnan || !(satm <= MB_H16_MAX))) atomicAdd(a.sat, 1u);   // (a NaN counts too)
    }
    if (has_next) __builtin_amdgcn_s_barrier();        // everyone's share of the next tile's K-tiles 0 and 1 has landed (each wave waited above)
    MB_TRACE(4);
#ifdef MB_HT_TRACE
#if MB_HT_TRACE >= 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (mode 2 only: when the stores are acknowledged; changes the overlap with the next tile)
    MB_TRACE(5);
#endif
    ++trace_it;
#endif
    if (!has_next) break;
    cur = nxt;
    vb = nvb;
  }
#undef MB_LOAD_A
#undef MB_LOAD_B
#undef MB_SYNC_L
#undef MB_MMA
#undef MB_MMA_DO
#undef MB_MMA_END
#undef MB_MINI_ONE
#undef MB_MINI_MMA
#undef MB_MINI_MMA_HN
#undef MB_MINI_MMA_QN
}

static const int g_col_split = getenv("MASKBIT_AMD_COL_SPLIT") ? atoi(getenv("MASKBIT_AMD_COL_SPLIT")) : 4;   // A/B switch (experiments): largest column split of small-batch tiles (1 = whole tiles only)
static int g_cu_override = 0;   // mb_set_cu_count: the CUs a persistent grid is sized for (a stream created with a CU mask sees fewer than the device has)
void set_cu_count(int n) { g_cu_override = n > 0 ? n : 0; }
static int num_cu_cached() {
  static int num_cu = 0;
  if (g_cu_override) return g_cu_override;
  if (!num_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (num_cu <= 0) num_cu = 256;
#ifdef MB_HT_TRACE
    if (getenv("MASKBIT_AMD_HT_GRID")) num_cu = atoi(getenv("MASKBIT_AMD_HT_GRID"));   // trace builds: fewer persistent workgroups than CUs
#endif
  }
  return num_cu;
}

template <int MT, int EPI, int XP = 0, bool SEQ = false, bool PAIR = false, int NS = 1>
static void launch_ht(hipStream_t s, const GemmArgs& a, bool persistent = true) {
  constexpr int BM = 32 * MT;
  constexpr int LDS = 2 * (BM * 128 + 2 * 128 * 128 + (SEQ ? 1024 : 0)) + (XP == 6 ? 128 * 64 + 256 * 64 : 0);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)gemm_ht_kernel<MT, EPI, XP, SEQ, PAIR, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    configured = true;
  }
  const int sq_rows = a.seq_rows ? a.seq_rows : 257;
  const int tiles_m = PAIR ? (a.pair_rows / sq_rows) * ((sq_rows - 1) / 128) : (SEQ ? (a.M / sq_rows) * ((sq_rows - 1) / 256) : (a.M + BM - 1) / BM), tiles_n = (a.N / 256) * NS;
  static const bool res_persist = !getenv("MASKBIT_AMD_RES_PERSIST") || atoi(getenv("MASKBIT_AMD_RES_PERSIST")) != 0;   // A/B switch (experiments)
  if (EPI == EPI_RES_F32 && !res_persist) persistent = false;
  const int grid = persistent ? std::min(tiles_m * tiles_n, num_cu_cached()) : tiles_m * tiles_n;   // persistent: one workgroup per CU walks the tile list
  hipLaunchKernelGGL((gemm_ht_kernel<MT, EPI, XP, SEQ, PAIR, NS>), dim3(grid), dim3(512), LDS, s, a, tiles_m, tiles_n);
}

bool gemm_ht_supported(GemmEpi epi, const GemmArgs& a) {
  const int sqr = a.seq_rows ? a.seq_rows : 257;
  const int tok = a.pair_rows ? 128 : 256;                                                                  // tokens of a sequence tile
  if (a.seq_rows && ((sqr - 1) % tok || ((sqr - 1) / tok & ((sqr - 1) / tok - 1)) || (!a.pair_rows && a.M % sqr))) return false;   // sequence tiles: whole tiles, a power of two per sequence
  if (a.pair_rows && (a.pair_rows % sqr || a.M != 2 * a.pair_rows || a.A2 || epi == EPI_GELU_F32)) return false;
  // (plain tiles may combine the mini-tiles with split activations: A2 / kw, K = 2 kw -- hi + lo LayerNorm outputs AND the weight correction)
  if (a.nlo && (a.nlo > 2 || (a.pair_rows ? a.pair_rows % sqr : a.M % sqr) || (a.kw ? a.kw : a.K) % 128 || (!a.pair_rows && a.nlo != 1) ||
                ((a.A2 || a.kw) && (a.pair_rows || !a.A2 || a.K != 2 * a.kw)) ||
                !a.lo[0].A4 || !a.lo[0].W4 || !a.lo[0].a_scale || !a.lo[0].w_scale ||
                (a.nlo == 2 && (!a.lo[1].A4 || !a.lo[1].W4 || !a.lo[1].a_scale || !a.lo[1].w_scale)) || (uint64_t)a.M * a.K * 2 >= (1ull << 32) ||
                epi == EPI_GELU_F32)) return false;
  return epi != EPI_LOGITS_F32 && a.N % 256 == 0 && a.K % 64 == 0 && a.K >= 128 && (a.M >= 512 || a.nlo) &&
         (uint64_t)a.M * a.K < (1ull << 32) && (uint64_t)a.N * a.K < (1ull << 32) &&
         (uint64_t)a.M * a.N * 4 < (1ull << 32);      // 32-bit element / byte offsets inside the kernel
}

void gemm_ht(hipStream_t s, GemmEpi epi, const GemmArgs& a, int mt) {
  bool persistent = true;
  const int sqr = a.seq_rows ? a.seq_rows : 257;
  const long seq_tiles = (long)(a.M / sqr) * ((sqr - 1) / 256);          // plain sequence tiles of this GEMM's rows
  if (mt >= 1000) { persistent = false; mt -= 1000; }   // A/B: one tile per workgroup
  if (mt != 6 && mt != 8 && mt != 257) {
    // pick the tile height that wastes fewer CU-rounds (one workgroup per CU)
    const int num_cu = num_cu_cached();
    auto cost = [&](int m) {
      const long tiles = (long)((a.M + 32 * m - 1) / (32 * m)) * (a.N / 256);
      return (double)((tiles + num_cu - 1) / num_cu) * 32 * m * (m == 8 ? 1.0 : 1.04);
    };
    mt = cost(8) <= cost(6) ? 8 : 6;
    if (a.M % sqr == 0) {          // sequence tiles: cost 272/256 of the MFMA work (257-token sequences) but leave no ragged round
      const long tiles = (long)(a.M / sqr) * ((sqr - 1) / 256) * (a.N / 256);
      if ((double)((tiles + num_cu - 1) / num_cu) * 272 < cost(mt)) mt = 257;
    }
  }
  if (a.nlo) {                                                             // MX-fp4 mini-tile passes (XP = 6; pair or plain sequence tiles)
    switch (epi) {
      case EPI_H16: if (a.pair_rows) launch_ht<8, EPI_H16, 6, true, true>(s, a, persistent); else launch_ht<8, EPI_H16, 6, true>(s, a, persistent); break;
      case EPI_GELU_H16: if (a.pair_rows) launch_ht<8, EPI_GELU_H16, 6, true, true>(s, a, persistent); else launch_ht<8, EPI_GELU_H16, 6, true>(s, a, persistent); break;
      case EPI_RES_F32:
        if (a.pair_rows) launch_ht<8, EPI_RES_F32, 6, true, true>(s, a, persistent);
        // few sequences: quarter- / half-column tiles when whole tiles would leave three quarters / half of the CUs idle (bit-identical results: see the kernel)
        else if (seq_tiles * (a.N / 256) * 4 <= num_cu_cached() && g_col_split >= 4) launch_ht<8, EPI_RES_F32, 6, true, false, 4>(s, a, persistent);
        else if (seq_tiles * (a.N / 256) * 2 <= num_cu_cached() && g_col_split >= 2) launch_ht<8, EPI_RES_F32, 6, true, false, 2>(s, a, persistent);
        else launch_ht<8, EPI_RES_F32, 6, true>(s, a, persistent);
        break;
      default: break;
    }
    return;
  }
  if (a.pair_rows) {                                                       // CFG pair tiles (sequence-aligned, fp16 K-tiles only)
    switch (epi) {
      case EPI_H16: launch_ht<8, EPI_H16, 0, true, true>(s, a, persistent); break;
      case EPI_GELU_H16: launch_ht<8, EPI_GELU_H16, 0, true, true>(s, a, persistent); break;
      case EPI_RES_F32: launch_ht<8, EPI_RES_F32, 0, true, true>(s, a, persistent); break;
      default: break;
    }
    return;
  }
#define MB_HT_CASE(E)                                                      \
  case E:                                                                  \
    if (mt == 257) launch_ht<8, E, 0, true>(s, a, persistent);             \
    else if (mt == 8) launch_ht<8, E>(s, a, persistent);                   \
    else launch_ht<6, E>(s, a, persistent);                                \
    break;
  switch (epi) {
    MB_HT_CASE(EPI_H16) MB_HT_CASE(EPI_GELU_H16) MB_HT_CASE(EPI_RES_F32) MB_HT_CASE(EPI_GELU_F32)
    default: break;
  }
#undef MB_HT_CASE
}

}  // namespace mb

#ifdef MB_HT_TRACE
extern "C" int mb_debug_ht_trace(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(mb::g_ht_trace), &p, sizeof(p)); }
#endif
