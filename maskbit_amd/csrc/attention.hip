// Bidirectional multi-head self-attention for one (sequence, head) per workgroup.
//
// Replaces nn.MultiheadAttention's fast path inside BertAttention (modeling/bert.py:84,137):
// softmax(Q K^T / sqrt(dh)) V over N = seq+1 = 257 tokens, heads = contiguous dh-wide slices of the
// packed in_proj output qkv[row, 0:d | d:2d | 2d:3d].
//
// N = 257 fits on chip, so there is no online softmax: K and V for the whole head are copied
// row-major into LDS by 16-byte LDS-DMA (72 KiB at dh = 64 -> 2 WGs/CU) while each wave fetches the Q
// fragments of all its 16-query tiles into registers.  Scores are computed TRANSPOSED, S^T = K Q^T
// (mfma A = K tile, B = Q tile): a lane holds one query column (q = lane&15) and 4 keys per 16-key
// tile, so row max / row sum are in-lane reductions plus two xor-shuffles and the probabilities are
// already in MFMA B-operand order for O^T = V^T P^T.  The V^T A-operand comes from the hardware
// transpose read ds_read_b64_tr_b16 straight out of the row-major V image: per 16-lane group, lane i
// points at V[key0 + i/4][d0 + 4*(i%4) ..+3] and lane c receives V[key0 + 0..3][d0 + c] (probed on
// gfx950: tools/micro/tr_read.hip).  The 32-key k-block takes keys {tile 2j: g*4..g*4+3, tile 2j+1:
// g*4..g*4+3} for lane group g on both operands, so P never moves between lanes.  The output lane
// owns O[q = lane&15][4 consecutive dh] -> 8-byte row-major stores.
// LDS rows are 2*dh bytes; 16-byte slots are XOR-swizzled (K: conflict-free ds_read_b128; V: with an
// even mask, so the two slots of a 32-byte tr-read segment stay adjacent) on the DMA source address
// and on the reads.  The tr reads are inline asm (no builtin), software-pipelined one k-block ahead
// with counted lgkmcnt waits.
#include <cstdlib>
#include <type_traits>

#include "mb_kernels.h"

namespace mb {

#ifndef MB_ATT_SDEPTH
#define MB_ATT_SDEPTH 3          // key tiles whose K fragments may be in flight ahead of their score MFMAs
#endif
constexpr int ATT_NKT = 18;              // key tiles of 16 -> up to 288 keys
constexpr int ATT_NP = ATT_NKT * 16;     // padded key count
constexpr int ATT_NW = 4;                // waves per workgroup (6 waves x 3 tiles measured slower: 104 vs 89 us; so did two query
                                         // tiles per pass sharing each K / V fragment, and de-synchronising the two workgroups of a CU)
constexpr int ATT_MAXQT = (ATT_NKT + ATT_NW - 1) / ATT_NW;   // q-tiles per wave

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// Timeline instrumentation (tools/att_trace.py builds its own copy with -DMB_ATT_TRACE; never in the product library): wave 0 of every workgroup
// stamps the 100 MHz wall clock -> trace[workgroup][32]: 0 start, 1 loads issued, 2 K / V / Q landed, then per query tile 3+4i .. 6+4i = after the
// score MFMAs, the softmax, the PV MFMAs, the stores.
#ifdef MB_ATT_TRACE
__device__ long long* g_att_trace = nullptr;
#define MB_ATRACE(k) do { if (g_att_trace && tid == 0 && (k) < 32) g_att_trace[((size_t)blockIdx.x + ((AUX == 4 && pass == 1) ? gridDim.x : 0)) * 32 + (k)] = wall_clock64(); } while (0)
#else
#define MB_ATRACE(k) do { } while (0)
#endif

// e2m1 of a lane's output tile o[4][4] (row q = lane & 15; values dh = 16 nt + 4 g + r, g = lane >> 4) * inv, scaled by `mul`, exchanged between the row's
// four lane groups so that the lane ends up with bytes [8 (g & 1) + 16 (g >> 1), +8) of the row's 32-byte (64-value) block: dh tile nt's 8 bytes live in
// lane group g = nt after the two swaps (v_permlane16_swap: odd / even lane rows; v_permlane32_swap: lane halves).  Every lane of the wave must call it.
__device__ __forceinline__ uint2 fp4_row8(const f32x4 (&o)[4], float inv, float mul) {
  uint32_t p[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) p[nt] = fp4_pack4(o[nt][0] * inv, o[nt][1] * inv, o[nt][2] * inv, o[nt][3] * inv, mul);
  const auto sw = __builtin_amdgcn_permlane16_swap((p[0] & 0xffffu) | (p[2] << 16), (p[1] & 0xffffu) | (p[3] << 16), false, false);
  const uint32_t q0 = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);      // 32-value group 0: this lane's 4 values | its neighbour's
  const uint32_t q1 = __builtin_amdgcn_perm(sw[1], sw[0], 0x07060302u);      // 32-value group 1
  const auto sx = __builtin_amdgcn_permlane32_swap(q0, q1, false, false);
  return make_uint2(sx[0], sx[1]);
}

// the same 8 bytes for the fp16 LO HALVES v - fp16(v) of the normalised outputs v = o * inv (precision 4: the out-proj GEMM's activation-lo pass); lo holds
// them on entry
__device__ __forceinline__ uint2 fp4_row8_vals(const f32x4 (&lo)[4], float mul) {
  uint32_t p[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) p[nt] = fp4_pack4(lo[nt][0], lo[nt][1], lo[nt][2], lo[nt][3], mul);
  const auto sw = __builtin_amdgcn_permlane16_swap((p[0] & 0xffffu) | (p[2] << 16), (p[1] & 0xffffu) | (p[3] << 16), false, false);
  const uint32_t q0 = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);
  const uint32_t q1 = __builtin_amdgcn_perm(sw[1], sw[0], 0x07060302u);
  const auto sx = __builtin_amdgcn_permlane32_swap(q0, q1, false, false);
  return make_uint2(sx[0], sx[1]);
}
// e2m1 copy of the lo halves of one (row, head) block held as o[nt][r] * inv: scale byte + 8 bytes of this lane (all 64 lanes take part)
__device__ __forceinline__ uint2 fp4_lo_block(const f32x4 (&o)[4], float inv, uint32_t& sbyte) {
  f32x4 lo[4];
  float am = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float v = o[nt][r] * inv; lo[nt][r] = v - (float)to_h(v); am = fmaxf(am, fabsf(lo[nt][r])); }
  am = rows_max(am);
  sbyte = fp4_scale_byte_nosat(am);
  return fp4_row8_vals(lo, fp4_scale_mul_nosat(am));
}

// AUX = 4 (CFG pair attention, mb_kernels.h attention_pair): one workgroup handles the conditional sequence of a (pair, head) and then its
// unconditional twin, the conditional output tiles parked in registers in between, and stores fp16(o_u - o_c) for the twin: no extra traffic, one
// launch, two workgroups per CU as before.  (Rounds 2-3 also had a two-launch form through fp32 rows in memory -- 119 against 88 us -- removed in
// round 5; tried and dropped earlier: both streams side by side in one 8-wave workgroup with an LDS mailbox, no faster.)  AUX = 5: the plain forward
// with the e2m1 copy of its outputs.
template <int DH, int AUX = 0>
__global__ __launch_bounds__(64 * ATT_NW, 2) void attention_kernel(const h16* __restrict__ qkv, h16* __restrict__ out,
                                                          int N, int d, int heads, float scale_log2e, int sq_off = 0,
                                                          uint8_t* __restrict__ out4 = nullptr, uint8_t* __restrict__ out4s = nullptr, int out4_nseq = 0,
                                                          uint8_t* __restrict__ out4l = nullptr, uint8_t* __restrict__ out4ls = nullptr) {
  // out4 / out4s (AUX 4: conditional sequences; AUX 5: the plain forward, every sequence; optional): e2m1 of the output VALUES (row stride 2d
  // bytes) with one E8M0 scale byte per (row, head) in the lane-ordered layout of the GEMM's mini-tile passes (GemmArgs.lo; out4_nseq sequences),
  // DH = 64, N = 257 -- the token operand of the out-proj GEMM's weight-correction pass
  constexpr int ROW = DH * 2;            // bytes per K / V row
  constexpr int SL = DH / 8;             // 16-byte slots per row (8 or 4)
  constexpr int KS = DH / 32;            // k-steps of the QK^T MFMA
  constexpr int NT = DH / 16;            // output dh tiles
  constexpr int RPI = 64 / SL;           // rows covered by one 1 KiB DMA instruction (8 or 16)
  constexpr int NINST = ATT_NP / RPI;    // DMA instructions per operand (36 or 18)
  __shared__ __attribute__((aligned(16))) char smem[2 * ATT_NP * ROW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Ks = smem;
  char* Vs = Ks + ATT_NP * ROW;
  const int sq0 = blockIdx.x / heads, h = blockIdx.x - sq0 * heads;
  const size_t rs = (size_t)3 * d;                                   // qkv row stride (elements)
  // AUX = 4: the workgroup handles the conditional sequence and then its unconditional twin (two passes over the same code); the conditional
  // output tiles stay in registers (oc) for the subtraction -- one launch, and still two 72 KiB workgroups per CU
  constexpr int NPASS = AUX == 4 ? 2 : 1;
  // (the tiles of a wave's first ATT_MAXQT - 1 rounds; the <= 2 waves that own a tile in the last round -- N = 257: the class-token tile -- park it in
  // a 4 KiB LDS slab each instead: with it in registers the kernel spilled 4 VGPRs, and a kernel with scratch costs more per launch than it saved)
  f32x4 oc[AUX == 4 ? ATT_MAXQT - 1 : 1][DH / 16];
  constexpr int LAST_WAVES = ATT_NKT - ATT_NW * (ATT_MAXQT - 1);       // waves with a tile in the last round
  __shared__ __attribute__((aligned(16))) float oc_last[AUX == 4 ? LAST_WAVES * 64 * DH / 4 : 4];
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
  const int sq = sq0 + (AUX == 4 ? pass * sq_off : sq_off);
  const h16* base = qkv + (size_t)sq * N * rs + h * DH;
  MB_ATRACE(0);
  if (AUX == 4 && pass > 0) __syncthreads();                         // every wave is done with the first pass's K / V image

  auto kswz = [](int row) { return SL == 8 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); };
  auto vswz = [](int row) { return SL == 8 ? (((row >> 1) & 3) << 1) : (((row >> 1) & 1) << 1); };

  // ---- stage K and V by LDS-DMA: instruction j covers rows [j*RPI, (j+1)*RPI); rows >= N re-read row N-1
  // (tried: K first and V waited for only after the first tile's scores and softmax -- no change, 88.9 vs 88.0 us)
  for (int j = wave; j < NINST; j += ATT_NW) {
    const int row = j * RPI + lane / SL, p = lane % SL;
    const h16* src = base + (size_t)min(row, N - 1) * rs;
    MB_GLDS16(src + d + (p ^ kswz(row)) * 8, Ks + j * 1024);
    MB_GLDS16(src + 2 * d + (p ^ vswz(row)) * 8, Vs + j * 1024);
  }
  // ---- Q fragments of every q-tile this wave owns (in flight together with the DMA).  N = 257 makes 17 query tiles for 4 waves: wave w owns
  // tiles w, w+4, .., w+12, and the tiles of the last round (16, 17) go to waves 0, 1.  (Rotating that extra tile over the waves with the pass and
  // the workgroup index, so that it lands on different SIMDs, changed nothing: 26.32 ms per forward either way.)
  const int wlast = wave;
  auto qt_of = [&](int i) { return i < ATT_MAXQT - 1 ? wave + ATT_NW * i : ATT_NW * (ATT_MAXQT - 1) + wlast; };
  const int l15 = lane & 15, g = lane >> 4;
  const int nqt = (N + 15) / 16;
  h16x8 qf[ATT_MAXQT][KS];
  auto q_fetch = [&](int i) {
    const int qrow = min(qt_of(i) * 16 + l15, N - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[i][ks] = *(const h16x8*)(base + (size_t)qrow * rs + (ks * 4 + g) * 8);
  };
  // (AUX 4 runs at the VGPR limit: it fetches the next tile's Q fragments one tile ahead instead of all up front)
#pragma unroll
  for (int i = 0; i < (AUX == 4 ? 1 : ATT_MAXQT); ++i) q_fetch(i);
  MB_ATRACE(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MB_ATRACE(2);

  // per-lane constant parts of the fragment addresses
  const int koff = l15 * ROW;                                       // K fragment: row kt*16 + l15
  const int vrow = g * 4 + (l15 >> 2), vchunk = (l15 & 3) * 8;      // V tr-read: row key0 + g*4 + i/4, 8-byte chunk i%4
  // address of the tr-read for key tile `kt` and dh tile `nt` = per-lane base of the dh tile + kt * 16 rows: the slot
  // swizzle depends on the row only through (row >> 1) & 3, which a multiple of 16 rows does not change, so the key tile
  // is an immediate offset of the instruction and the K loop carries no address arithmetic
  unsigned vb_nt[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    vb_nt[nt] = (unsigned)(uintptr_t)Vs + vrow * ROW + ((((nt * 32 + vchunk) >> 4) ^ vswz(vrow)) << 4) + (vchunk & 8);

#pragma unroll
  for (int i = 0; i < ATT_MAXQT; ++i) {
    const int qt = qt_of(i);
    if (qt >= nqt) break;
    // ---- S^T tiles: s[kt][r] = S[q = l15][key = kt*16 + g*4 + r]
    f32x4 s[ATT_NKT];
#ifndef MB_ATT_NOSPIPE
    // groups of GK key tiles; the K fragments of group n+1 are requested before the MFMAs of group n, and inside a group all first k-steps go
    // before the second ones (no MFMA waits for the one issued just before it)
    constexpr int GK = MB_ATT_SDEPTH, NG = ATT_NKT / GK;
    static_assert(ATT_NKT % GK == 0, "whole groups");
    h16x8 kfr[2][GK][KS];
    auto k_fetch = [&](h16x8 (&dst)[GK][KS], int grp) {
#pragma unroll
      for (int j = 0; j < GK; ++j) {
        const int kt = grp * GK + j, row = kt * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst[j][ks] = *(const h16x8*)(Ks + kt * 16 * ROW + koff + (((ks * 4 + g) ^ kswz(row)) * 16));
      }
    };
    k_fetch(kfr[0], 0);
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
      if (grp + 1 < NG) k_fetch(kfr[(grp + 1) & 1], grp + 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < GK; ++j) {
          const int kt = grp * GK + j;
          if (ks == 0) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (kt == ATT_NKT - 1 && kt * 16 >= N) continue;    // the padding tile holds no key at N = 257
          s[kt] = MB_MFMA_16x16x32(kfr[grp & 1][j][ks], qf[i][ks], s[kt]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int kt = 0; kt < ATT_NKT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = kt * 16 + l15;
      if (kt == ATT_NKT - 1 && kt * 16 >= N) continue;      // the padding tile holds no key at N = 257
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const h16x8 kf = *(const h16x8*)(Ks + kt * 16 * ROW + koff + (((ks * 4 + g) ^ kswz(row)) * 16));
        s[kt] = MB_MFMA_16x16x32(kf, qf[i][ks], s[kt]);
      }
      if (kt % MB_ATT_SDEPTH == MB_ATT_SDEPTH - 1) __builtin_amdgcn_sched_barrier(0);   // bound the fragment prefetch depth (VGPR budget)
    }
#endif
    if (AUX == 4 && i + 1 < ATT_MAXQT) q_fetch(i + 1);
    // (round 3, measured and removed: pulling the twin sequence's K / V rows towards the L2 during the first pass -- one dword per row as LDS-DMA into the
    // padding key tile -- made the launch SLOWER, 90-91 us against 85-88: the second pass's staging is not waiting for HBM.)
    MB_ATRACE(3 + 4 * i);
    // ---- softmax over keys (fp32); only the last two key tiles can hold keys >= N
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < ATT_NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kt >= ATT_NKT - 2) s[kt][r] = (kt * 16 + g * 4 + r < N) ? s[kt][r] : -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = rows_max(mx);
    const float mxs = mx * scale_log2e;
    f32x2 sum2 = {0.f, 0.f};                                  // packed fp32 (v_pk_fma_f32 / v_pk_add_f32): two keys per VALU issue
#pragma unroll
    for (int kt = 0; kt < ATT_NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2 arg = __builtin_elementwise_fma((f32x2){s[kt][r], s[kt][r + 1]}, (f32x2)(scale_log2e), (f32x2)(-mxs));
        const f32x2 p = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};   // exp((s - max)/sqrt(dh)); arg <= 0: bare v_exp_f32
        s[kt][r] = p.x; s[kt][r + 1] = p.y;
        sum2 += p;
      }
    float sum = sum2.x + sum2.y;
    sum = rows_sum(sum);
    float inv = 1.0f / sum;
    asm volatile("" : "+v"(inv));          // every cross-lane op of the softmax has retired before the asm LDS reads start
    MB_ATRACE(4 + 4 * i);
    // ---- O^T = V^T P^T ; V^T fragments by transpose reads, one k-block ahead
    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Two named register sets alternate (never copied: an asm load's destination must not be touched by
    // compiler-generated moves before the counted wait that covers it).
    uint2 va[NT][2], vb[NT][2];
    auto issue = [&](uint2 (&dst)[NT][2], auto kbc) {         // kbc: std::integral_constant -- the key offset is an immediate
      constexpr int kb = decltype(kbc)::value;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned ad = vb_nt[nt];                         // (an ordinary use: asm operands alone do not capture in a generic lambda)
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst[nt][0]) : "v"(ad), "n"(2 * kb * 16 * (DH * 2)) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst[nt][1]) : "v"(ad), "n"((2 * kb + 1) * 16 * (DH * 2)) : "memory");
      }
    };
    auto consume = [&](uint2 (&cur)[NT][2], int kb, bool more_in_flight) {
      // `more_in_flight`: the 2*NT reads of the next k-block were issued after `cur`'s and may stay outstanding
      if constexpr (NT == 4) {
        if (more_in_flight)
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[1][0]), "+v"(cur[1][1]),
                       "+v"(cur[2][0]), "+v"(cur[2][1]), "+v"(cur[3][0]), "+v"(cur[3][1])::"memory");
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[1][0]), "+v"(cur[1][1]),
                       "+v"(cur[2][0]), "+v"(cur[2][1]), "+v"(cur[3][0]), "+v"(cur[3][1])::"memory");
      } else {
        if (more_in_flight)
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[1][0]), "+v"(cur[1][1])::"memory");
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[1][0]), "+v"(cur[1][1])::"memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 p0 = s[2 * kb], p1 = s[2 * kb + 1];
      const h16x8 pf = {(h16)p0[0], (h16)p0[1], (h16)p0[2], (h16)p0[3],      // probabilities are in [0, 1]: no saturation clamp
                         (h16)p1[0], (h16)p1[1], (h16)p1[2], (h16)p1[3]};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x4 raw = {cur[nt][0].x, cur[nt][0].y, cur[nt][1].x, cur[nt][1].y};
        o[nt] = MB_MFMA_16x16x32(__builtin_bit_cast(h16x8, raw), pf, o[nt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int NKB = ATT_NKT / 2;
    issue(va, std::integral_constant<int, 0>{});
    auto step = [&](auto kbc) {
      constexpr int kb = decltype(kbc)::value;
      if constexpr (kb < NKB) {
        if constexpr (kb + 1 < NKB) issue(vb, std::integral_constant<int, kb + 1>{});
        consume(va, kb, kb + 1 < NKB);
        if constexpr (kb + 1 < NKB) {
          if constexpr (kb + 2 < NKB) issue(va, std::integral_constant<int, kb + 2>{});
          consume(vb, kb + 1, kb + 2 < NKB);
        }
      }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 8>{});
    static_assert(NKB <= 10, "add steps");
    MB_ATRACE(5 + 4 * i);
    // ---- o[nt][r] = O[q = l15][dh = nt*16 + g*4 + r]
    const int q = qt * 16 + l15;
    if constexpr ((AUX == 4 || AUX == 5) && DH == 64) {
      if (out4 && (AUX != 4 || pass == 0)) {                          // block (row, head) = this lane's 16 values x its 4 lane groups
        float am = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) am = fmaxf(am, fabsf(o[nt][r]));
        am = rows_max(am) * inv;                                        // (== the maximum of the rounded products: inv > 0, rounding is monotonic)
        const float mul = fp4_scale_mul_nosat(am);
        // a lane's four dh tiles are 2 bytes each (4 values): as 2-byte stores the copy cost 7-9 us of the launch (tools/att_f4_ab.py).  Two lane
        // exchanges -- odd / even lane rows (dh tiles 2j <-> 2j + 1), then the lane halves -- give every lane 8 CONSECUTIVE bytes of the row's 32-byte
        // block (the GELU epilogue's trick, gemm_ht.hip); all 64 lanes take part (the row guard is uniform over a row's four lane groups)
        const uint2 pk8 = fp4_row8(o, inv, mul);
        uint32_t lob = 0;
        uint2 lo8 = make_uint2(0u, 0u);
        if (out4l) lo8 = fp4_lo_block(o, inv, lob);                   // (uniform branch: every lane of the wave takes part in the exchanges)
        if (q < 256) {                                                // (class-token rows take no part in the mini-tile passes)
          const size_t row = (size_t)sq * N + q;
          if (g == 0) out4s[fp4_scale_index(h, out4_nseq, sq, q)] = (uint8_t)fp4_scale_byte_nosat(am);
          *(uint2*)(out4 + row * 2 * d + (h * DH) / 2 + (g >> 1) * 16 + (g & 1) * 8) = pk8;
          if (out4l) {
            if (g == 0) out4ls[fp4_scale_index(h, out4_nseq, sq, q)] = (uint8_t)lob;
            *(uint2*)(out4l + row * 2 * d + (h * DH) / 2 + (g >> 1) * 16 + (g & 1) * 8) = lo8;
          }
        }
      }
    }
    if (q < N) {
      const size_t ooff = ((size_t)sq * N + q) * d + h * DH;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 v = {o[nt][0] * inv, o[nt][1] * inv, o[nt][2] * inv, o[nt][3] * inv};
        if constexpr (AUX == 4) {
          if (i < ATT_MAXQT - 1) { f32x4& r = oc[i < ATT_MAXQT - 1 ? i : 0][nt]; if (pass == 0) r = v; else v = v - r; }
          else { f32x4* slot = (f32x4*)oc_last + ((qt - ATT_NW * (ATT_MAXQT - 1)) * NT + nt) * 64 + lane; if (pass == 0) *slot = v; else v = v - *slot; }
        }
        const h16x4 hi = {to_h(v[0]), to_h(v[1]), to_h(v[2]), to_h(v[3])};
        *(h16x4*)(out + ooff + nt * 16 + g * 4) = hi;
      }
    }
    MB_ATRACE(6 + 4 * i);
  }
  }                                      // pass
}

// ---- long sequences (N > 288, e.g. the 512x512 models' 1025 tokens): K/V streamed in 128-key blocks, online softmax ------------
// One workgroup per (sequence, head, 64-query chunk); wave w owns query tile 4*chunk + w.  Same transposed formulation and the
// same fragment reads as attention_kernel; the running row maximum m and denominator l live in the lane that owns the query
// column, and the output accumulators are rescaled by exp2((m_old - m_new) * c) when a block raises the maximum
// (Milakov-Gimelshein / FlashAttention-style streaming softmax).  K/V blocks are double-buffered through LDS by LDS-DMA.
constexpr int ATTL_KB = 128;             // keys per block
// PAIRM (round 5: the differential guided forward of the 1024 + 1-token models): the workgroup runs its 64 queries against the conditional sequence,
// keeps the normalised fp32 output tile (16 VGPRs), then against the label-dropped twin `sq_off` sequences further down and stores
// fp16(o_u - o_c) there -- the pair form of attention_kernel<DH, 4>.  out4 / out4s (optional, DH = 64): e2m1 of the conditional outputs + one scale
// byte per (row, head), lane-ordered with (N - 1) / 64 token groups per sequence, for the out-proj pair GEMM's weight-correction mini-tiles.
template <int DH, bool PAIRM = false>
__global__ __launch_bounds__(256, 2) void attention_long_kernel(const h16* __restrict__ qkv, h16* __restrict__ out,
                                                               int N, int d, int heads, int nchunk, float scale_log2e, int sq_off = 0,
                                                               uint8_t* __restrict__ out4 = nullptr, uint8_t* __restrict__ out4s = nullptr, int out4_nseq = 0,
                                                               uint8_t* __restrict__ out4l = nullptr, uint8_t* __restrict__ out4ls = nullptr) {
  constexpr int ROW = DH * 2, SL = DH / 8, KS = DH / 32, NT = DH / 16, RPI = 64 / SL;
  constexpr int BLK_BYTES = ATTL_KB * ROW;                 // one operand block
  constexpr int NINST = ATTL_KB / RPI;                     // DMA instructions per operand block (16 or 8)
  constexpr int NKT = ATTL_KB / 16;                        // key tiles per block
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * BLK_BYTES];   // [buffer][K | V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int chunk = bid % nchunk; bid /= nchunk;
  const int sq0 = bid / heads, h = bid - sq0 * heads;
  const size_t rs = (size_t)3 * d;
  auto kswz = [](int row) { return SL == 8 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); };
  auto vswz = [](int row) { return SL == 8 ? (((row >> 1) & 3) << 1) : (((row >> 1) & 1) << 1); };
  const int nblk = (N + ATTL_KB - 1) / ATTL_KB;
  f32x4 oc[NT];                                            // PAIRM: the conditional pass's output tile
#pragma unroll 1
  for (int pass = 0; pass < (PAIRM ? 2 : 1); ++pass) {
  const int sq = sq0 + pass * sq_off;
  const h16* base = qkv + (size_t)sq * N * rs + h * DH;
  if (PAIRM && pass > 0) __syncthreads();                  // (the last block's buffer is free: the loop below ends with a barrier; kept for clarity)
  auto stage = [&](int b, int buf) {                       // rows >= N re-read row N-1 (masked below)
    char* Kb = smem + buf * 2 * BLK_BYTES;
    char* Vb = Kb + BLK_BYTES;
    for (int j = wave; j < NINST; j += 4) {
      const int r = j * RPI + lane / SL, p = lane % SL;    // row inside the block: the swizzle uses the block-local row
      const h16* src = base + (size_t)min(b * ATTL_KB + r, N - 1) * rs;
      MB_GLDS16(src + d + (p ^ kswz(r)) * 8, Kb + j * 1024);
      MB_GLDS16(src + 2 * d + (p ^ vswz(r)) * 8, Vb + j * 1024);
    }
  };
  const int l15 = lane & 15, g = lane >> 4;
  const int qt = chunk * 4 + wave;                         // this wave's query tile (may lie beyond N: computed, never stored)
  h16x8 qf[KS];
  {
    const int qrow = min(qt * 16 + l15, N - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const h16x8*)(base + (size_t)qrow * rs + (ks * 4 + g) * 8);
  }
  const int koff = l15 * ROW;
  const int vrow = g * 4 + (l15 >> 2), vchunk = (l15 & 3) * 8;
  unsigned vb_nt[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) vb_nt[nt] = vrow * ROW + ((((nt * 32 + vchunk) >> 4) ^ vswz(vrow)) << 4) + (vchunk & 8);

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  for (int b = 0; b < nblk; ++b) {
    if (b + 1 < nblk) stage(b + 1, (b + 1) & 1);
    // block b has landed once at most the DMA instructions of block b+1 (NINST/4 pairs per wave) are outstanding
    if (b + 1 < nblk) {
      if constexpr (NINST == 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* Kb = smem + (b & 1) * 2 * BLK_BYTES;
    const unsigned Vb = (unsigned)(uintptr_t)(Kb + BLK_BYTES);
    // ---- S^T tiles of this block
    f32x4 sblk[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      sblk[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = kt * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const h16x8 kf = *(const h16x8*)(Kb + kt * 16 * ROW + koff + (((ks * 4 + g) ^ kswz(row)) * 16));
        sblk[kt] = MB_MFMA_16x16x32(kf, qf[ks], sblk[kt]);
      }
    }
    // ---- online softmax: block maximum, rescale, probabilities
    const int key0 = b * ATTL_KB;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (key0 + kt * 16 + g * 4 + r >= N) sblk[kt][r] = -INFINITY;
        mx = fmaxf(mx, sblk[kt][r]);
      }
    mx = rows_max(mx);
    const float m_new = fmaxf(m_run, mx);                   // finite: every block holds at least one real key
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);   // exp2(-inf) = 0 on the first block
    const float mxs = m_new * scale_log2e;
    float sum = 0.f;
    h16x4 pk[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(sblk[kt][r], scale_log2e, -mxs)); sum += p[r]; }
      pk[kt] = h16x4{(h16)p[0], (h16)p[1], (h16)p[2], (h16)p[3]};
    }
    sum = rows_sum(sum);
    l_run = l_run * alpha + sum;
    m_run = m_new;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
    asm volatile("" : "+v"(l_run));                           // the cross-lane ops above have retired before the asm LDS reads
    // ---- O^T += V^T P^T over the block's 32-key k-blocks
#pragma unroll
    for (int kb = 0; kb < NKT / 2; ++kb) {
      uint2 v[NT][2];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned ad = Vb + vb_nt[nt];
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[nt][0]) : "v"(ad + (2 * kb) * 16 * ROW) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[nt][1]) : "v"(ad + (2 * kb + 1) * 16 * ROW) : "memory");
      }
      if constexpr (NT == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]),
                     "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[3][0]), "+v"(v[3][1])::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1])::"memory");
      const h16x8 pf = __builtin_shufflevector(pk[2 * kb], pk[2 * kb + 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x4 raw = {v[nt][0].x, v[nt][0].y, v[nt][1].x, v[nt][1].y};
        o[nt] = MB_MFMA_16x16x32(__builtin_bit_cast(h16x8, raw), pf, o[nt]);
      }
    }
    __syncthreads();                                          // everyone is done with this buffer before it is refilled
  }
  const int q = qt * 16 + l15;
  const float inv = 1.0f / l_run;
  if constexpr (DH == 64) {
    if (out4s && pass == 0) {                              // (plain form: every sequence; pair form: the conditional ones)                              // block (row, head) = this lane's 16 values x its 4 lane groups (cf. attention_kernel)
      float am = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) am = fmaxf(am, fabsf(o[nt][r]));
      am = rows_max(am) * inv;
      const float mul = fp4_scale_mul_nosat(am);
      const uint2 pk8 = fp4_row8(o, inv, mul);             // (8 consecutive bytes per lane: see attention_kernel)
      uint32_t lob = 0;
      uint2 lo8 = make_uint2(0u, 0u);
      if (out4l) lo8 = fp4_lo_block(o, inv, lob);
      if (q < N - 1) {                                     // (class-token rows take no part in the mini-tile passes)
        const size_t row = (size_t)sq * N + q;
        if (g == 0) out4s[fp4_scale_index(h, out4_nseq, sq, q, (N - 1) >> 6)] = (uint8_t)fp4_scale_byte_nosat(am);
        *(uint2*)(out4 + row * 2 * d + (h * DH) / 2 + (g >> 1) * 16 + (g & 1) * 8) = pk8;
        if (out4l) {
          if (g == 0) out4ls[fp4_scale_index(h, out4_nseq, sq, q, (N - 1) >> 6)] = (uint8_t)lob;
          *(uint2*)(out4l + row * 2 * d + (h * DH) / 2 + (g >> 1) * 16 + (g & 1) * 8) = lo8;
        }
      }
    }
  }
  if (q < N || PAIRM) {
    const size_t ooff = ((size_t)sq * N + min(q, N - 1)) * d + h * DH;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 v = {o[nt][0] * inv, o[nt][1] * inv, o[nt][2] * inv, o[nt][3] * inv};
      if constexpr (PAIRM) { if (pass == 0) oc[nt] = v; else v = v - oc[nt]; }
      if (q >= N) continue;
      const h16x4 hi = {to_h(v[0]), to_h(v[1]), to_h(v[2]), to_h(v[3])};
      *(h16x4*)(out + ooff + nt * 16 + g * 4) = hi;
    }
  }
  }                                      // pass
}

// ---- return_attn=True (bert.py:119,137: need_weights with the default average_attn_weights): the head-averaged softmax
// weights [nb, N, N] fp32 of one layer, from the same fp16 qkv rows the fused kernel consumes.  A diagnostic path (attention
// maps for visualisation), not on the sampling loop: plain fp32 FMAs, one wave per query row, scores kept in LDS.
template <int DH>
__global__ __launch_bounds__(256) void attention_probs_kernel(const h16* __restrict__ qkv, float* __restrict__ out, int N, int d, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows_per_seq = (N + 3) / 4;
  const int seq = blockIdx.x / rows_per_seq, r = (blockIdx.x % rows_per_seq) * 4 + wave;
  if (r >= N) return;
  float* sc = (float*)smem + (size_t)wave * 2 * N;          // scores of the current head
  float* av = sc + N;                                        // running mean over the heads
  for (int j = lane; j < N; j += 64) av[j] = 0.f;
  const float scale = 1.0f / sqrtf((float)DH), invh = 1.0f / (float)heads;
  const h16* base = qkv + (size_t)seq * N * 3 * d;
  for (int h = 0; h < heads; ++h) {
    float q[DH];
    const h16* qp = base + (size_t)r * 3 * d + h * DH;
#pragma unroll
    for (int i = 0; i < DH / 8; ++i) {
      const h16x8 v = *(const h16x8*)(qp + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) q[i * 8 + e] = (float)v[e] * scale;
    }
    float mx = -3.0e38f;
    for (int j = lane; j < N; j += 64) {
      const h16* kp = base + (size_t)j * 3 * d + d + h * DH;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < DH / 8; ++i) {
        const h16x8 v = *(const h16x8*)(kp + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(q[i * 8 + e], (float)v[e], a);
      }
      sc[j] = a;
      mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float w = invh / sum;
    for (int j = lane; j < N; j += 64) av[j] = fmaf(sc[j], w, av[j]);
  }
  float* orow = out + ((size_t)seq * N + r) * N;
  for (int j = lane; j < N; j += 64) orow[j] = av[j];
}

int attention_probs(hipStream_t s, const h16* qkv, float* out, int nb, int N, int d, int heads) {
  const int dh = d / heads;
  const size_t lds = (size_t)4 * 2 * N * sizeof(float);
  if ((dh != 32 && dh != 64) || lds > 160 * 1024) return -1;
  dim3 grid(nb * ((N + 3) / 4)), block(256);
  if (dh == 64) {
    (void)hipFuncSetAttribute((const void*)attention_probs_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attention_probs_kernel<64>, grid, block, lds, s, qkv, out, N, d, heads);
  } else {
    (void)hipFuncSetAttribute((const void*)attention_probs_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attention_probs_kernel<32>, grid, block, lds, s, qkv, out, N, d, heads);
  }
  return 0;
}

void attention(hipStream_t s, const h16* qkv, h16* out, int nb, int N, int d, int heads, uint8_t* out4, uint8_t* out4s) {
  const int dh = d / heads;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
  if (N > ATT_NP) {                                           // longer than one head's K/V fits in LDS: streaming kernel
    const int nchunk = (N + 63) / 64;
    dim3 grid(nb * heads * nchunk), block(256);
    if (dh == 64) hipLaunchKernelGGL(attention_long_kernel<64>, grid, block, 0, s, qkv, out, N, d, heads, nchunk, scale_log2e, 0, (N - 1) % 64 ? nullptr : out4, (N - 1) % 64 ? nullptr : out4s, nb);
    else hipLaunchKernelGGL(attention_long_kernel<32>, grid, block, 0, s, qkv, out, N, d, heads, nchunk, scale_log2e);
    return;
  }
  dim3 grid(nb * heads), block(64 * ATT_NW);
  if (out4 && dh == 64 && N == 257)        // plain forward with the weight-correction mini-tiles: also the e2m1 copy of the outputs
    hipLaunchKernelGGL((attention_kernel<64, 5>), grid, block, 0, s, qkv, out, N, d, heads, scale_log2e, 0, out4, out4s, nb);
  else if (dh == 64) hipLaunchKernelGGL(attention_kernel<64>, grid, block, 0, s, qkv, out, N, d, heads, scale_log2e);
  else hipLaunchKernelGGL(attention_kernel<32>, grid, block, 0, s, qkv, out, N, d, heads, scale_log2e);
}

int attention_pair(hipStream_t s, const h16* qkv, h16* out, int P, int N, int d, int heads, uint8_t* out4, uint8_t* out4s, uint8_t* out4l, uint8_t* out4ls) {
  const int dh = d / heads;
  if (dh != 64 && dh != 32) return -1;
  if ((out4l || out4ls) && (!out4 || !out4s || !out4l || !out4ls)) return -1;   // the lo copy rides with the value copy
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
  // experiment switch (tools/att_stream_ab.py; never set by the product): the streaming kernel for 257-token sequences too -- 64 queries per workgroup,
  // 144 VGPRs, K / V in 128-key blocks: more workgroups and waves in flight per CU against a head's K / V re-read by every 64-query chunk
  static const bool force_stream = getenv("MASKBIT_AMD_ATT_STREAM") && atoi(getenv("MASKBIT_AMD_ATT_STREAM")) != 0;
  if (N > ATT_NP || (force_stream && (N - 1) % 64 == 0)) {     // the 1024 + 1-token models: streaming kernel, both streams of a pair in one workgroup
    if (out4 && (dh != 64 || (N - 1) % 64)) return -1;
    const int nchunk = ((N + 15) / 16 + 3) / 4;
    dim3 grid(P * heads * nchunk), block(256);
    if (dh == 64) hipLaunchKernelGGL((attention_long_kernel<64, true>), grid, block, 0, s, qkv, out, N, d, heads, nchunk, scale_log2e, P, out4, out4s, P, out4l, out4ls);
    else hipLaunchKernelGGL((attention_long_kernel<32, true>), grid, block, 0, s, qkv, out, N, d, heads, nchunk, scale_log2e, P);
    return 0;
  }
  dim3 grid(P * heads), block(64 * ATT_NW);
  if ((out4 || out4s) && (dh != 64 || N != 257)) return -1;   // the fp4 output exists for head dimension 64 and 257-token sequences
  // one workgroup per (pair, head): conditional pass, then the twin; conditional outputs parked in registers
  if (dh == 64) hipLaunchKernelGGL((attention_kernel<64, 4>), grid, block, 0, s, qkv, out, N, d, heads, scale_log2e, P, out4, out4s, P, out4l, out4ls);
  else hipLaunchKernelGGL((attention_kernel<32, 4>), grid, block, 0, s, qkv, out, N, d, heads, scale_log2e, P);
  return 0;
}

}  // namespace mb

#ifdef MB_ATT_TRACE
extern "C" int mb_debug_att_trace(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(mb::g_att_trace), &p, sizeof(p)); }
#endif
#if defined(MB_ATT_TRACE) || defined(MB_ATT_VARIANT)      // experimental builds of this file alone (tools/att_trace.py)
extern "C" int mb_debug_attention_pair(const void* qkv, void* out, int P, int N, int d, int heads, void* stream) {
  return mb::attention_pair((hipStream_t)stream, (const h16*)qkv, (h16*)out, P, N, d, heads, nullptr, nullptr);
}
#endif
