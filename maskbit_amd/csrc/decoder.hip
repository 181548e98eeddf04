// conv-VQGAN decoder (ConvDecoder.forward, modeling/modules/autoencoder.py:399-423) as NHWC h16
// implicit-GEMM convolutions on MFMA.
//
// Layout: every activation is [B, H, W, C] h16 (channels contiguous = the MFMA k order), weights
// are repacked once to [tap][Cout_pad][Cin_pad] h16.  One workgroup computes an 8x16-pixel output
// tile for 128 (or 16) output channels; for each 64-channel input chunk the (8+2)x(16+2) halo tile
// is staged ONCE into LDS and re-used by all 9 taps (a tap is just a shifted LDS row index), while
// the per-tap weight tiles stream in by LDS-DMA, double-buffered, exactly like the GEMM's W tile.
// Fusions:  GroupNorm-apply + SiLU happen on the way into LDS (per-(image,channel) scale/shift
// from the stats pass), nearest-2x upsampling is an index shift (>>1) of the source pixel, bias /
// residual add live in the epilogue, and the last conv writes fp32 NCHW and/or clamp*255 uint8 NHWC.
// GroupNorm statistics (32 groups, eps 1e-6, autoencoder.py:39-43) are a deterministic two-level
// reduction (no float atomics), so outputs are bit-stable run to run.  Level one lives in the epilogue of the conv that PRODUCES the tensor
// (one (sum, sumsq) pair per 8x16-pixel tile and group, written next to the tile), level two in gn_finalize_kernel before the consuming conv;
// only tensors that no conv of this file produced (the encoder's average-pooled ones) still take the separate sweep (gn_partial_kernel).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "mb_decoder.h"
#include "mb_kernels.h"

namespace mb {

constexpr int TH8 = 8, TW = 16;          // output pixel tile: TH x 16 pixels, TH = 8 (4 waves) or 16 (8 waves: twice the pixels per weight tile, four waves per SIMD)
constexpr int CK = 64;                   // input-channel chunk = one 128-byte LDS row

struct ConvArgs {
  const h16* in;        // [B, Hin, Win, Cin] (Hin = H/2 when UP)
  const float2* gn;      // [B, Cin] (scale, shift) or null
  const h16* w;         // [taps][Cout_pad][Cin]
  const float* bias;     // [Cout_pad] or null
  const h16* residual;  // [B, H, W, Cout] or null
  h16* out;             // [B, H, W, Cout]
  float* img_nchw;       // final conv only
  uint8_t* img_u8;       // final conv only
  int B, H, W, Cin, Cout, Cout_pad;
  unsigned* sat;         // counts output groups of 4 whose value left the fp16 range and was clamped (mb_dec_saturation_count)
  float* gn_part;        // or null: GroupNorm partial statistics of the OUTPUT, [B][pixel tiles per image][32 groups][sum, sumsq] -- the consumer's
                         // GroupNorm then needs no sweep over the tensor (its fp16-stored values are what is summed, as that sweep did)
#ifdef MB_CONV_TRACE
  long long* trace;      // tools/dec_trace.py: [workgroup][16] wall-clock stamps of one selected launch
#endif
};

// Timeline instrumentation (tools/dec_trace.py builds its own copy with -DMB_CONV_TRACE; never in the product library): thread 0 of every workgroup
// stamps the 100 MHz wall clock: 0 start; per input-channel chunk c < 3: 1+4c halo free (barrier passed), 2+4c halo staged by this wave, 3+4c weights +
// halo visible (barrier passed), 4+4c the chunk's last tap done; 15 end of the epilogue.
#ifdef MB_CONV_TRACE
#define MB_CTRACE(k) do { if (a.trace && tid == 0 && (k) < 16) a.trace[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define MB_CTRACE(k) do { } while (0)
#endif

__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// TH = 16 (round 3): the 8 x 16 tile ran two 4-wave workgroups per CU (LDS-bound) = two waves per SIMD, each tap step (32 MFMAs per wave) behind
// a barrier and the LDS-DMA of its weight tile: per step 3 500 clocks for 512 clocks of matrix work.  A 16 x 16 tile on 8 waves keeps two
// workgroups per CU (74 KiB each) but FOUR waves per SIMD at the same 125 VGPRs, stages each weight tile for twice the pixels and shrinks the
// halo overhead from 1.41 to 1.27 pixels read per pixel written.
template <int NI, int WN, int KS, bool UP, bool FINAL, int TH = 8>
__global__ __launch_bounds__(32 * TH, TH / 4) void conv_kernel(ConvArgs a) {
  constexpr int NTHR = 32 * TH, NWAVE = NTHR / 64;
  constexpr int WM = NWAVE / WN, MJ = TH / WM, BN = WN * NI * 16;
  // KS = 3: symmetric pad 1.  KS = 2 (the stride-2 Conv2dSame of the encoder, run on a space-to-depth input): no pad before,
  // one zero row/column after (autoencoder.py:18,31-36: TF "SAME" puts the odd pixel at the bottom/right).
  constexpr int PAD = (KS - 1) / 2, HW_ = TW + KS - 1, HALO = (TH + KS - 1) * HW_, NTAP = KS * KS;
  constexpr int WT_BYTES = BN * 128;
  constexpr int HALO_BYTES = (HALO + 7) / 8 * 1024;   // whole 8-pixel DMA groups
  __shared__ __attribute__((aligned(16))) char smem[HALO_BYTES + 2 * WT_BYTES];
  char* halo = smem;
  char* wt = smem + HALO_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;

  const int ntn = a.Cout_pad / BN, ntx = a.W / TW, nty = a.H / TH;
  int bid = blockIdx.x;
  const int tn = bid % ntn; bid /= ntn;
  const int tx = bid % ntx; bid /= ntx;
  const int ty = bid % nty; const int b = bid / nty;
  const int n0 = tn * BN, y0 = ty * TH, x0 = tx * TW;
  const int Cin = a.Cin;
  const int Hin = UP ? a.H / 2 : a.H, Win = UP ? a.W / 2 : a.W;

  // ---- weight-tile DMA: BN rows of 128 B; a wave instruction covers 8 rows
  constexpr int WROWS_PER_WAVE = BN / NWAVE;        // 32 / 16 (BN = 128 on 4 / 8 waves) or 4 (BN = 16)
  constexpr int WINST = (WROWS_PER_WAVE + 7) / 8;   // 4 / 2 or 1
  const h16* wsrc[WINST];
#pragma unroll
  for (int j = 0; j < WINST; ++j) {
    int row = wave * WROWS_PER_WAVE + j * 8 + (lane >> 3);
    if (WROWS_PER_WAVE < 8) row = min(row, BN - 1);
    wsrc[j] = a.w + (size_t)(n0 + row) * Cin + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
  auto stage_w = [&](int t, int buf) {
    const int chunk = t / NTAP, tap = t - chunk * NTAP;
    const size_t off = (size_t)tap * a.Cout_pad * Cin + chunk * CK;
    if (WROWS_PER_WAVE >= 8) {
#pragma unroll
      for (int j = 0; j < WINST; ++j)
        MB_GLDS16(wsrc[j] + off, wt + buf * WT_BYTES + (wave * WROWS_PER_WAVE + j * 8) * 128);
    } else if (wave < 2) {                           // BN = 16: two 8-row instructions in total
      const int row = wave * 8 + (lane >> 3);
      const h16* src = a.w + (size_t)(n0 + row) * Cin + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
      MB_GLDS16(src + off, wt + buf * WT_BYTES + wave * 8 * 128);
    }
  };

  // ---- halo staging (GN-apply + SiLU + zero padding).  The raw rows come in by LDS-DMA, 8 pixels x 128 B per wave instruction, from clamped
  // coordinates; every lane then normalises the 16 bytes IT brought in, in place (no barrier in between: a lane re-reads only its own slot after
  // its own vmcnt wait).  All of a wave's 5-6 instructions are in flight together and hold no registers.  (Round 3, before: a
  // load -> SiLU -> ds_write loop through registers, which the compiler left rolled -- six trips to memory per thread one after the other,
  // ~1.5 us each under load, twice per tile of a 128-channel conv whose workgroup lived 47 us; unrolled with the loads batched it spilled.)
  // Wave w takes pixel groups j = w, w + NWAVE, ..: j keeps its parity, so the 16-byte slot swizzle ((pixel >> 1) & 7 = (4j + lane/16) & 7)
  // maps a lane to ONE logical channel slot for all its groups and the GroupNorm scale / shift of its 8 channels stay in registers.
  constexpr int NGRP = (HALO + 7) / 8, NGW = (NGRP + NWAVE - 1) / NWAVE;
  const int myslot = (lane & 7) ^ (lane >> 4) ^ ((wave & 1) << 2);
  auto stage_halo = [&](int chunk) {
    const int c0 = chunk * CK + myslot * 8;
#pragma unroll
    for (int jj = 0; jj < NGW; ++jj) {
      const int j = wave + jj * NWAVE;
      if (j < NGRP) {
        const int hp = min(j * 8 + (lane >> 3), HALO - 1);
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int Y = min(max(y0 - PAD + hy, 0), a.H - 1), X = min(max(x0 - PAD + hx, 0), a.W - 1);
        const int sy = UP ? (Y >> 1) : Y, sx = UP ? (X >> 1) : X;
        MB_GLDS16(a.in + (((size_t)b * Hin + sy) * Win + sx) * Cin + c0, halo + j * 1024);
      }
    }
    float sc[8], sh[8];
    if (a.gn) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float2 v = a.gn[(size_t)b * Cin + c0 + e]; sc[e] = v.x; sh[e] = v.y; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int jj = 0; jj < NGW; ++jj) {
      const int j = wave + jj * NWAVE;
      const int hp = j * 8 + (lane >> 3);
      if (j < NGRP && hp < HALO) {
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int Y = y0 - PAD + hy, X = x0 - PAD + hx;
        h16x8* slot = (h16x8*)(halo + j * 1024 + lane * 16);
        if (Y >= 0 && Y < a.H && X >= 0 && X < a.W) {
          if (a.gn) {
            h16x8 v = *slot;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = to_h(silu(fmaf((float)v[e], sc[e], sh[e])));
            *slot = v;
          }
        } else {
          *slot = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
    }
  };

  int wfoff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) wfoff[kk] = l15 * 128 + (((kk * 4 + g) ^ (l15 >> 1)) * 16);

  f32x4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int T = (Cin / CK) * NTAP;
  MB_CTRACE(0);
  stage_w(0, 0);
  for (int t = 0; t < T; ++t) {
    const int chunk = t / NTAP, tap = t - chunk * NTAP;
    if (tap == 0) {
      __syncthreads();                      // all waves are done with the previous chunk's halo
      MB_CTRACE(1 + 4 * chunk);
      stage_halo(chunk);
      MB_CTRACE(2 + 4 * chunk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                        // weight tile t landed, halo visible
    if (tap == 0) MB_CTRACE(3 + 4 * chunk);
    if (t + 1 < T) stage_w(t + 1, (t + 1) & 1);
    const int dy = tap / KS, dx = tap - dy * KS;
    const char* wb = wt + (t & 1) * WT_BYTES + wn * NI * 16 * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h16x8 wf[NI], xf[MJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *(const h16x8*)(wb + i * 16 * 128 + wfoff[kk]);
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const int hp = (wm * MJ + j + dy) * HW_ + l15 + dx;
        xf[j] = *(const h16x8*)(halo + hp * 128 + (((kk * 4 + g) ^ ((hp >> 1) & 7)) * 16));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
          acc[i][j] = MB_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
    }
    if (tap == NTAP - 1) MB_CTRACE(4 + 4 * chunk);
  }

  // ---- epilogue: lane holds out[pixel (y = wm*MJ+j, x = l15)][cout = ..+g*4 .. +3]
  // The bias of a lane's channels is fetched once; the residual values of pixel row j + 1 are requested before row j is stored, and the
  // saturation count is one atomic per lane at the end.  (Round 3, before: bias and residual loaded inside the (row, channel tile) loop behind
  // run-time branches -- the compiler waited vmcnt(0) after each of the 32 loads, i.e. also for the previous store: 14.5 us of a 50 us workgroup.)
  float gs[NI], gq[NI];                     // GroupNorm partials of this lane's 4 channels of n-tile i over its MJ pixels
  float4 bv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    gs[i] = 0.f; gq[i] = 0.f;
    bv[i] = a.bias ? *(const float4*)(a.bias + n0 + wn * NI * 16 + i * 16 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (FINAL) {
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int Y = y0 + wm * MJ + j, X = x0 + l15;
      const size_t pix = ((size_t)b * a.H + Y) * a.W + X;
      const float v[4] = {acc[0][j][0] + bv[0].x, acc[0][j][1] + bv[0].y, acc[0][j][2] + bv[0].z, acc[0][j][3] + bv[0].w};
      if (g == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r < a.Cout) {
            if (a.img_nchw) a.img_nchw[(((size_t)b * a.Cout + r) * a.H + Y) * a.W + X] = v[r];
            if (a.img_u8) a.img_u8[pix * a.Cout + r] = (uint8_t)(fminf(fmaxf(v[r], 0.f), 1.f) * 255.0f);
          }
        }
      }
    }
  } else {
    const size_t pix0 = ((size_t)b * a.H + y0 + wm * MJ) * a.W + x0 + l15;        // pixel row j: + j * W
    const int nl = n0 + wn * NI * 16 + g * 4;                                       // channel of n-tile i: + i * 16
    unsigned nsat = 0;
    // straight-line per variant (residual or not; every channel of the tile stored or not): with the run-time tests inside the loop the
    // compiler's wait-count pass fell back to vmcnt(0) in front of every store
    auto body = [&](auto res_c, auto full_c) {
      constexpr bool RES = decltype(res_c)::value, FULL = decltype(full_c)::value;
      h16x4 rv[2][NI];
      auto fetch = [&](int j, h16x4 (&r)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          r[i] = h16x4{0, 0, 0, 0};
          if (RES && (FULL || nl + i * 16 < a.Cout)) r[i] = *(const h16x4*)(a.residual + (pix0 + (size_t)j * a.W) * a.Cout + nl + i * 16);
        }
      };
      fetch(0, rv[0]);
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        if (j + 1 < MJ) fetch(j + 1, rv[(j + 1) & 1]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          if (FULL || nl + i * 16 < a.Cout) {
            float v[4] = {acc[i][j][0] + bv[i].x, acc[i][j][1] + bv[i].y, acc[i][j][2] + bv[i].z, acc[i][j][3] + bv[i].w};
            if (RES) {
              const h16x4 r = rv[j & 1][i];
              v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
            }
            // activations are stored as fp16: values beyond +-65504 are clamped by to_h -- counted, so that a checkpoint whose decoder needs a
            // wider residual stream is noticed instead of silently clipped (random-init weights stay far inside the range)
            nsat += fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) > MB_H16_MAX ? 1u : 0u;
            const h16x4 hv = {to_h(v[0]), to_h(v[1]), to_h(v[2]), to_h(v[3])};
            *(h16x4*)(a.out + (pix0 + (size_t)j * a.W) * a.Cout + nl + i * 16) = hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float f = (float)hv[r]; gs[i] += f; gq[i] = fmaf(f, f, gq[i]); }
          }
        }
      }
    };
    const bool full = n0 + BN <= a.Cout;
    if (a.residual) { if (full) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
    else { if (full) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
    if (nsat) atomicAdd(a.sat, nsat);
  }
  if constexpr (!FINAL) {
    if (a.gn_part) {                        // uniform; requires Cout % 128 == 0 and 4 | 8 | 16 channels per group (launch_conv)
      // 16 pixel columns (lanes of a lane group), then the lane groups that share a GroupNorm group, then the WM wave rows through LDS
      const int cpg = a.Cout >> 5;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { gs[i] += __shfl_xor(gs[i], o); gq[i] += __shfl_xor(gq[i], o); }
        if (cpg >= 8) { gs[i] += __shfl_xor(gs[i], 16); gq[i] += __shfl_xor(gq[i], 16); }
        if (cpg >= 16) { gs[i] += __shfl_xor(gs[i], 32); gq[i] += __shfl_xor(gq[i], 32); }
      }
      __syncthreads();                      // everyone is done with the halo / weight tiles: smem is free
      float* red = (float*)smem;            // [wave][i][g][2]
      if (l15 == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) { red[((wave * NI + i) * 4 + g) * 2] = gs[i]; red[((wave * NI + i) * 4 + g) * 2 + 1] = gq[i]; }
      }
      __syncthreads();
      // one thread per GroupNorm group of this workgroup's BN channels: channel c0 = first channel of the group inside the tile
      const int ngrp = BN / cpg;
      if (tid < ngrp) {
        const int c0 = tid * cpg, wn_ = c0 / (NI * 16), i_ = (c0 % (NI * 16)) / 16, g_ = (c0 % 16) / 4;
        float ts = 0.f, tq = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {      // fixed order over the wave rows
          const int w = m * WN + wn_;
          ts += red[((w * NI + i_) * 4 + g_) * 2]; tq += red[((w * NI + i_) * 4 + g_) * 2 + 1];
        }
        const int ntile = nty * ntx, tile = ty * ntx + tx;
        float* o = a.gn_part + (((size_t)b * ntile + tile) * 32 + (n0 / cpg + tid)) * 2;
        o[0] = ts; o[1] = tq;
      }
    }
  }
  MB_CTRACE(15);
}

// ---- GroupNorm statistics: partial (sum, sumsq) per (image, pixel chunk, group) -----------------
__global__ __launch_bounds__(256) void gn_partial_kernel(const h16* __restrict__ x, float* __restrict__ part, int HW,
                                                         int C, int nchunk) {
  __shared__ float red[2][256 * 8];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int nslot = C / 8, npl = 256 / nslot;       // 8-channel slots per pixel, pixel lanes
  const int slot = tid % nslot, pl = tid / nslot;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  for (int p = p0 + pl; p < p1; p += npl) {
    const h16x8 v = *(const h16x8*)(x + ((size_t)b * HW + p) * C + slot * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] = fmaf(f, f, q[e]); }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][pl * C + slot * 8 + e] = s[e]; red[1][pl * C + slot * 8 + e] = q[e]; }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {               // fixed-order sum over pixel lanes
    float ts = 0.f, tq = 0.f;
    for (int l = 0; l < npl; ++l) { ts += red[0][l * C + c]; tq += red[1][l * C + c]; }
    red[0][c] = ts; red[1][c] = tq;                  // row 0 of the scratch is only read by thread c here
  }
  __syncthreads();
  if (tid < 32) {
    const int cpg = C / 32;
    float ts = 0.f, tq = 0.f;
    for (int e = 0; e < cpg; ++e) { ts += red[0][tid * cpg + e]; tq += red[1][tid * cpg + e]; }
    float* o = part + (((size_t)b * nchunk + chunk) * 32 + tid) * 2;
    o[0] = ts; o[1] = tq;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float2* __restrict__ out, int HW, int C, int nchunk) {
  // 256 threads = 32 groups x 8 chunk lanes: lane l sums chunks l, l+8, ... in order, then the 8 lanes are summed in order (fixed association:
  // bit-stable run to run whatever produced the partials)
  __shared__ float red[2][8][32];
  __shared__ float2 ms[32];
  const int b = blockIdx.x, grp = threadIdx.x & 31, l = threadIdx.x >> 5;
  const int cpg = C / 32;
  float ts = 0.f, tq = 0.f;
  for (int k = l; k < nchunk; k += 8) {
    const float* o = part + (((size_t)b * nchunk + k) * 32 + grp) * 2;
    ts += o[0]; tq += o[1];
  }
  red[0][l][grp] = ts; red[1][l][grp] = tq;
  __syncthreads();
  if (threadIdx.x < 32) {
    ts = 0.f; tq = 0.f;
    for (int k = 0; k < 8; ++k) { ts += red[0][k][grp]; tq += red[1][k][grp]; }
    const float n = (float)HW * (float)cpg;
    const float mean = ts / n;
    const float var = fmaxf(tq / n - mean * mean, 0.f);
    ms[grp] = make_float2(mean, rsqrtf(var + 1e-6f));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float2 m = ms[c / cpg];
    const float sc = m.y * gamma[c];
    out[(size_t)b * C + c] = make_float2(sc, beta[c] - m.x * sc);
  }
}

// ---- tokens -> +-1 latent, NHWC padded to 64 channels (lookup_free.py:96-111, conv_vqgan.py:107-110)
__global__ void latent_kernel(const int64_t* __restrict__ tokens, h16* __restrict__ z, size_t npix, int K) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * CK; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i / CK; const int c = (int)(i - p * CK);
    float v = 0.f;
    if (c < K) v = ((tokens[p] >> c) & 1) ? 1.f : -1.f;
    z[i] = to_h(v);
  }
}

// ---- OIHW fp32 -> [tap][Cout_pad][Cin_pad] h16 ---------------------------------------------------
__global__ void repack_conv_kernel(const float* __restrict__ w, h16* __restrict__ out, int Cout, int Cin, int ks,
                                   int Cout_pad, int Cin_pad) {
  const size_t total = (size_t)ks * ks * Cout_pad * Cin_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin_pad); size_t r = i / Cin_pad;
    const int co = (int)(r % Cout_pad); const int tap = (int)(r / Cout_pad);
    float v = 0.f;
    if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * ks * ks + tap];
    out[i] = to_h(v);
  }
}

// ---- encoder half (ConvEncoder, autoencoder.py:230-286; LFQ sign/pack, lookup_free.py:57-62,113-127) -------------
// image fp32 NCHW -> fp16 NHWC padded to 64 channels
__global__ void pack_image_kernel(const float* __restrict__ img, h16* __restrict__ out, int B, int C, int H, int W) {
  const size_t npix = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * 8; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i >> 3; const int slot = (int)(i & 7);
    const size_t b = p / ((size_t)H * W), yx = p - b * (size_t)H * W;
    h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (slot == 0)
      for (int c = 0; c < C && c < 8; ++c) v[c] = to_h(img[(b * C + c) * (size_t)H * W + yx]);
    *(h16x8*)(out + p * CK + slot * 8) = v;
  }
}
// space-to-depth: x[B,H,W,C] -> y[B,H/2,W/2,4C], channel (py*2+px)*C + c <- pixel (2Y+py, 2X+px)
__global__ void s2d_kernel(const h16* __restrict__ x, h16* __restrict__ y, int B, int H, int W, int C) {
  const int c8 = C / 8;
  const size_t total = (size_t)B * H * W * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int sl = (int)(i % c8); size_t p = i / c8;
    const int X = (int)(p % W); p /= W; const int Y = (int)(p % H); const size_t b = p / H;
    const h16x8 v = *(const h16x8*)(x + ((b * H + Y) * W + X) * C + sl * 8);
    *(h16x8*)(y + (((b * (H / 2) + (Y >> 1)) * (W / 2) + (X >> 1)) * 4 + ((Y & 1) * 2 + (X & 1))) * C + sl * 8) = v;
  }
}
// F.avg_pool2d(kernel 2, stride 2) (autoencoder.py:182): x[B,H,W,C] -> y[B,H/2,W/2,C], fp32 mean of the four fp16 inputs
__global__ void avgpool2_kernel(const h16* __restrict__ x, h16* __restrict__ y, int B, int H, int W, int C) {
  const int c8 = C / 8, Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)B * Ho * Wo * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int sl = (int)(i % c8); size_t p = i / c8;
    const int X = (int)(p % Wo); p /= Wo; const int Y = (int)(p % Ho); const size_t b = p / Ho;
    const h16* src = x + ((b * H + 2 * Y) * W + 2 * X) * C + sl * 8;
    const h16x8 v00 = *(const h16x8*)src, v01 = *(const h16x8*)(src + C), v10 = *(const h16x8*)(src + (size_t)W * C),
                v11 = *(const h16x8*)(src + (size_t)W * C + C);
    h16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = to_h(((float)v00[c] + (float)v01[c] + (float)v10[c] + (float)v11[c]) * 0.25f);
    *(h16x8*)(y + ((b * Ho + Y) * Wo + X) * C + sl * 8) = o;
  }
}
// OIHW 3x3 stride-2 weights -> [tap (by,bx)][Cout_pad][4*Cin] for the 2x2 conv on the space-to-depth input:
// tap (by,bx), channel (py*2+px)*Cin + ci  <-  w[co][ci][2by+py][2bx+px] (zero where that index is 3)
__global__ void repack_down_kernel(const float* __restrict__ w, h16* __restrict__ out, int Cout, int Cin, int Cout_pad) {
  const size_t total = (size_t)4 * Cout_pad * 4 * Cin;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % (4 * Cin)); size_t r = i / (4 * Cin);
    const int co = (int)(r % Cout_pad); const int tap = (int)(r / Cout_pad);
    const int ci = ch % Cin, pp = ch / Cin, py = pp >> 1, px = pp & 1, by = tap >> 1, bx = tap & 1;
    const int dy = 2 * by + py, dx = 2 * bx + px;
    float v = 0.f;
    if (co < Cout && dy < 3 && dx < 3) v = w[(((size_t)co * Cin + ci) * 3 + dy) * 3 + dx];
    out[i] = to_h(v);
  }
}
// z[B*h*w, Kp] (fp16 NHWC) -> indices (bit j = z_j > 0, LSB first), optional +-1 latent and raw z as fp32 NCHW
__global__ void lfq_kernel(const h16* __restrict__ z, int64_t* __restrict__ idx, float* __restrict__ zq, float* __restrict__ zraw,
                           int B, int HW, int K, int Kp) {
  const size_t npix = (size_t)B * HW;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    const size_t b = p / HW, yx = p - b * HW;
    int64_t code = 0;
    for (int j = 0; j < K; ++j) {
      const float v = (float)z[p * Kp + j];
      const bool pos = v > 0.0f;
      code |= (int64_t)pos << j;
      if (zq) zq[(b * K + j) * HW + yx] = pos ? 1.0f : -1.0f;
      if (zraw) zraw[(b * K + j) * HW + yx] = v;
    }
    idx[p] = code;
  }
}

// ================================================================================================
struct Conv {
  std::string name; int cin = 0, cout = 0, ks = 3; bool has_bias = false, up = false;
  bool down = false;       // stride-2 3x3 Conv2dSame, executed as a 2x2 conv on the space-to-depth input (cin_pad = 4*cin)
  int cout_w = 0;          // output channels in the checkpoint (cout may be rounded up for 8-byte stores)
  int cin_pad = 0, cout_pad = 0;
  h16* w = nullptr; float* b = nullptr;
  unsigned* sat = nullptr; // the engine's saturation counter
};
struct Norm { std::string name; int c = 0; float *g = nullptr, *b = nullptr; };
struct ResBlock { Norm n1, n2; Conv c1, c2, sc; bool has_sc = false; };
struct Stage { std::vector<ResBlock> blocks; Conv up; bool has_up = false; };   // `up`: upsample_conv (decoder) / down_conv (encoder)

}  // namespace mb

struct mb_dec {
  mb_dec_cfg c{};
  int max_batch = 0, out_res = 0;
  mb::Conv conv_in, conv_out;
  mb::Norm norm_out;
  std::vector<mb::ResBlock> mid;
  std::vector<mb::Stage> up;
  // encoder half (built when cfg.build_encoder): conv_in, down stages, mid, norm_out, conv_out
  bool has_enc = false;
  mb::Conv e_conv_in, e_conv_out;
  mb::Norm e_norm_out;
  std::vector<mb::ResBlock> e_mid;
  std::vector<mb::Stage> e_down;
  h16* buf[3] = {nullptr, nullptr, nullptr};
  h16* z = nullptr;
  unsigned* sat = nullptr;  // device counter: fp16 clamps in the conv epilogues since the last read
  float* gn_part = nullptr;
  float2* gn_ss = nullptr;
  const void* gn_of = nullptr;   // the tensor whose per-tile GroupNorm partials the last conv left in gn_part (null: none) ...
  int gn_ntile = 0;              // ... and the number of pixel tiles per image they cover
  std::vector<void*> owned;
};

namespace mb {

namespace {
constexpr int GN_MAXCHUNK = 64;

template <typename T>
bool dalloc(mb_dec* d, T** p, size_t n, std::string& err) {
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) { err = std::string("hipMalloc failed: ") + hipGetErrorString(e); return false; }
  d->owned.push_back((void*)*p);
  return true;
}

bool init_conv(mb_dec* d, Conv& c, const std::string& name, int cin, int cout, int ks, bool bias, bool up, bool final_,
               std::string& err) {
  c.name = name; c.cin = cin; c.cout = cout; c.cout_w = cout; c.ks = ks; c.has_bias = bias; c.up = up; c.sat = d->sat;
  c.cin_pad = (cin + CK - 1) / CK * CK;
  c.cout_pad = final_ ? 16 : (cout + 127) / 128 * 128;
  if (!dalloc(d, &c.w, (size_t)ks * ks * c.cout_pad * c.cin_pad, err)) return false;
  if (bias) {
    if (!dalloc(d, &c.b, (size_t)c.cout_pad, err)) return false;
    (void)hipMemset(c.b, 0, c.cout_pad * sizeof(float));
  }
  return true;
}
bool init_norm(mb_dec* d, Norm& n, const std::string& name, int c, std::string& err) {
  n.name = name; n.c = c;
  return dalloc(d, &n.g, (size_t)c, err) && dalloc(d, &n.b, (size_t)c, err);
}
bool init_block(mb_dec* d, ResBlock& rb, const std::string& p, int cin, int cout, std::string& err) {
  rb.has_sc = cin != cout;
  bool ok = init_norm(d, rb.n1, p + ".norm1", cin, err) && init_conv(d, rb.c1, p + ".conv1", cin, cout, 3, false, false, false, err) &&
            init_norm(d, rb.n2, p + ".norm2", cout, err) && init_conv(d, rb.c2, p + ".conv2", cout, cout, 3, false, false, false, err);
  if (ok && rb.has_sc) ok = init_conv(d, rb.sc, p + ".nin_shortcut", cout, cout, 1, false, false, false, err);
  return ok;
}

#ifdef MB_CONV_TRACE
static long long* g_conv_trace = nullptr;
static int g_conv_trace_sel = -1, g_conv_trace_n = 0;
#endif
void launch_conv(hipStream_t s, mb_dec* d, const Conv& c, const h16* in, const float2* gn, const h16* residual, h16* out,
                 float* img, uint8_t* u8, int B, int H, int W, bool final_, bool stats = true) {
  // GroupNorm partials of the output ride in the epilogue when a GroupNorm will read it (stats) and its groups are whole lane groups of a tile
  const int cpg = c.cout / 32;
  const bool part = !final_ && stats && c.cout % 128 == 0 && (cpg == 4 || cpg == 8 || cpg == 16);
  ConvArgs a{in, gn, c.w, c.has_bias ? c.b : nullptr, residual, out, img, u8, B, H, W, c.cin_pad, c.cout, c.cout_pad, c.sat, part ? d->gn_part : nullptr};
#ifdef MB_CONV_TRACE
  a.trace = (g_conv_trace && g_conv_trace_n++ == g_conv_trace_sel) ? g_conv_trace : nullptr;
  if (a.trace) printf("conv launch %d: %s  %dx%d  %d -> %d  ks %d%s\n", g_conv_trace_sel, c.name.c_str(), H, W, c.cin_pad, c.cout, c.ks, c.up ? " up" : "");
#endif
  d->gn_of = part ? (const void*)out : nullptr;
  const int bn = final_ ? 16 : 128;
  // 16-row tiles (8 waves) for the 3x3 convolutions from 32 x 32 maps on; 8-row tiles below (a 16 x 16 map would be one tile per image).  The choice
  // must not depend on the batch: the GroupNorm partial sums are per tile, and results are bit-identical across batch sizes.
  static const int th_force = getenv("MASKBIT_AMD_CONV_TH") ? atoi(getenv("MASKBIT_AMD_CONV_TH")) : 0;   // A/B switch (experiments; read once)
  bool th16 = !final_ && c.ks == 3 && H % 16 == 0 && H >= 32;
  if (th_force == 8) th16 = false;
  const int th = th16 ? 16 : TH8;
  d->gn_ntile = (H / th) * (W / TW);
  dim3 grid((unsigned)((size_t)B * (H / th) * (W / TW) * (c.cout_pad / bn))), block(32 * th);
  if (final_) hipLaunchKernelGGL((conv_kernel<1, 1, 3, false, true>), grid, block, 0, s, a);
  else if (c.ks == 1) hipLaunchKernelGGL((conv_kernel<4, 2, 1, false, false>), grid, block, 0, s, a);
  else if (c.ks == 2) hipLaunchKernelGGL((conv_kernel<4, 2, 2, false, false>), grid, block, 0, s, a);
  else if (c.up && th16) hipLaunchKernelGGL((conv_kernel<4, 2, 3, true, false, 16>), grid, block, 0, s, a);
  else if (c.up) hipLaunchKernelGGL((conv_kernel<4, 2, 3, true, false>), grid, block, 0, s, a);
  else if (th16) hipLaunchKernelGGL((conv_kernel<4, 2, 3, false, false, 16>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((conv_kernel<4, 2, 3, false, false>), grid, block, 0, s, a);
}

void launch_gn(hipStream_t s, mb_dec* d, const Norm& n, const h16* x, int B, int HW) {
  int nchunk = d->gn_ntile;
  if (d->gn_of != (const void*)x) {                   // not the tensor the last conv summed (average-pooled tensors of the encoder): sweep it
    nchunk = HW / 256; if (nchunk < 1) nchunk = 1; if (nchunk > GN_MAXCHUNK) nchunk = GN_MAXCHUNK;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, d->gn_part, HW, n.c, nchunk);
  }
  d->gn_of = nullptr;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, d->gn_part, n.g, n.b, d->gn_ss, HW, n.c, nchunk);
}

// x (buffer index xi) -> returns the buffer index holding the block output
int run_block(hipStream_t s, mb_dec* d, const ResBlock& rb, int xi, int B, int H, int W) {
  const int t1 = (xi + 1) % 3, t2 = (xi + 2) % 3;
  launch_gn(s, d, rb.n1, d->buf[xi], B, H * W);
  launch_conv(s, d, rb.c1, d->buf[xi], d->gn_ss, nullptr, d->buf[t1], nullptr, nullptr, B, H, W, false);
  launch_gn(s, d, rb.n2, d->buf[t1], B, H * W);
  if (!rb.has_sc) {
    launch_conv(s, d, rb.c2, d->buf[t1], d->gn_ss, d->buf[xi], d->buf[t2], nullptr, nullptr, B, H, W, false);
    return t2;
  }
  // shortcut quirk (autoencoder.py:72-73,93-96): out = h + nin_shortcut(h); the block input is dropped
  launch_conv(s, d, rb.c2, d->buf[t1], d->gn_ss, nullptr, d->buf[t2], nullptr, nullptr, B, H, W, false, false);   // read by the shortcut conv, not by a GroupNorm
  launch_conv(s, d, rb.sc, d->buf[t2], nullptr, d->buf[t2], d->buf[t1], nullptr, nullptr, B, H, W, false);
  return t1;
}

bool find_conv(Conv& c, const std::string& n, Conv** hit, bool* is_bias) {
  if (n == c.name + ".weight") { *hit = &c; *is_bias = false; return true; }
  if (n == c.name + ".bias" && c.has_bias) { *hit = &c; *is_bias = true; return true; }
  return false;
}
bool find_norm(Norm& nm, const std::string& n, float** dst) {
  if (n == nm.name + ".weight") { *dst = nm.g; return true; }
  if (n == nm.name + ".bias") { *dst = nm.b; return true; }
  return false;
}
}  // namespace

mb_dec* dec_create(const mb_dec_cfg& cfg, int max_batch, std::string& err) {
  const int R = cfg.num_resolutions;
  if (R < 1 || R > 7) { err = "num_resolutions out of range"; return nullptr; }
  if (cfg.hidden_channels % 64) { err = "hidden_channels must be a multiple of 64 for the HIP decoder"; return nullptr; }
  if (cfg.token_size > CK || cfg.token_size < 1) { err = "token_size must be in [1, 64]"; return nullptr; }
  if (cfg.latent_size % 16) { err = "latent_size must be a multiple of 16"; return nullptr; }
  if (cfg.num_channels > 4) { err = "num_channels > 4 unsupported"; return nullptr; }
  mb_dec* d = new mb_dec();
  d->c = cfg; d->max_batch = max_batch; d->out_res = cfg.latent_size << (R - 1);
  if (!dalloc(d, &d->sat, 1, err)) { dec_destroy(d); return nullptr; }
  (void)hipMemset(d->sat, 0, sizeof(unsigned));
  const int hc = cfg.hidden_channels;
  std::vector<int> mult(cfg.channel_mult, cfg.channel_mult + R);
  mult.push_back(cfg.channel_mult[R - 1]);
  const int top = hc * cfg.channel_mult[R - 1];
  bool ok = init_conv(d, d->conv_in, "decoder.conv_in", cfg.token_size, top, 3, true, false, false, err);
  d->mid.resize(cfg.num_res_blocks);
  for (int r = 0; ok && r < cfg.num_res_blocks; ++r)
    ok = init_block(d, d->mid[r], "decoder.mid.res_blocks." + std::to_string(r), top, top, err);
  d->up.resize(R);
  int last = top;
  size_t max_elems = 0;
  int res = cfg.latent_size;
  max_elems = std::max((size_t)res * res * top, (size_t)d->out_res * d->out_res * (size_t)std::max(CK, hc));
  for (int s = 0; ok && s < R; ++s) {                    // up.0 = coarsest level (autoencoder.py:384-392)
    const int lvl = R - 1 - s;
    const int cin = hc * mult[lvl + 1], cout = hc * mult[lvl];
    Stage& st = d->up[s];
    st.blocks.resize(cfg.num_res_blocks);
    int c = cin;
    for (int r = 0; ok && r < cfg.num_res_blocks; ++r) {
      ok = init_block(d, st.blocks[r], "decoder.up." + std::to_string(s) + ".res_blocks." + std::to_string(r), c, cout, err);
      c = cout;
    }
    max_elems = std::max(max_elems, (size_t)res * res * std::max(cin, cout));
    st.has_up = lvl > 0;
    if (ok && st.has_up) {
      ok = init_conv(d, st.up, "decoder.up." + std::to_string(s) + ".upsample_conv", cout, cout, 3, true, true, false, err);
      res *= 2;
      max_elems = std::max(max_elems, (size_t)res * res * cout);
    }
    last = cout;
  }
  ok = ok && init_norm(d, d->norm_out, "decoder.norm_out", last, err) &&
       init_conv(d, d->conv_out, "decoder.conv_out", last, cfg.num_channels, 3, true, false, true, err);
  if (ok && cfg.build_encoder) {                          // ConvEncoder (autoencoder.py:230-262): mirrors the decoder top-down
    const int enrb = cfg.enc_res_blocks > 0 ? cfg.enc_res_blocks : cfg.num_res_blocks;
    std::vector<int> imult{1};
    imult.insert(imult.end(), cfg.channel_mult, cfg.channel_mult + R);
    ok = ok && init_conv(d, d->e_conv_in, "encoder.conv_in", cfg.num_channels, hc, 3, false, false, false, err);
    d->e_down.resize(R);
    for (int s = 0; ok && s < R; ++s) {
      const int cin = hc * imult[s], cout = hc * imult[s + 1];
      Stage& st = d->e_down[s];
      st.blocks.resize(enrb);
      int c = cin;
      for (int r = 0; ok && r < enrb; ++r) {
        ok = init_block(d, st.blocks[r], "encoder.down." + std::to_string(s) + ".res_blocks." + std::to_string(r), c, cout, err);
        c = cout;
      }
      st.has_up = s < R - 1;                              // a downsampling step follows: down_conv, or avg_pool2d when !sample_with_conv
      st.up.cin = cout;
      if (ok && st.has_up && cfg.sample_with_conv) {      // DownsamplingStage.down_conv: 3x3, stride 2, bias (autoencoder.py:165)
        Conv& dc = st.up;
        dc.name = "encoder.down." + std::to_string(s) + ".down_conv";
        dc.cin = cout; dc.cout = cout; dc.cout_w = cout; dc.ks = 2; dc.has_bias = true; dc.down = true;
        dc.cin_pad = 4 * cout; dc.cout_pad = (cout + 127) / 128 * 128;
        ok = dalloc(d, &dc.w, (size_t)4 * dc.cout_pad * dc.cin_pad, err) && dalloc(d, &dc.b, (size_t)dc.cout_pad, err);
        if (ok) (void)hipMemset(dc.b, 0, dc.cout_pad * sizeof(float));
      }
    }
    d->e_mid.resize(enrb);
    for (int r = 0; ok && r < enrb; ++r)
      ok = init_block(d, d->e_mid[r], "encoder.mid.res_blocks." + std::to_string(r), top, top, err);
    const int k4 = (cfg.token_size + 3) / 4 * 4;           // stored channel count of z (8-byte stores)
    ok = ok && init_norm(d, d->e_norm_out, "encoder.norm_out", top, err) &&
         init_conv(d, d->e_conv_out, "encoder.conv_out", top, k4, 1, true, false, false, err);
    d->e_conv_out.cout_w = cfg.token_size;
    d->has_enc = ok;
  }
  for (int i = 0; ok && i < 3; ++i) ok = dalloc(d, &d->buf[i], (size_t)max_batch * max_elems, err);
  ok = ok && dalloc(d, &d->z, (size_t)max_batch * cfg.latent_size * cfg.latent_size * CK, err) &&
       dalloc(d, &d->gn_part, (size_t)max_batch * std::max(GN_MAXCHUNK, (d->out_res / TH8) * (d->out_res / TW)) * 64, err) &&
       dalloc(d, &d->gn_ss, (size_t)max_batch * 4096, err);
  if (!ok) { dec_destroy(d); return nullptr; }
  return d;
}

void dec_destroy(mb_dec* d) {
  if (!d) return;
  for (void* p : d->owned) (void)hipFree(p);
  delete d;
}

int dec_load(mb_dec* d, const char* name, const float* data, const int64_t* shape, int ndim, hipStream_t s, std::string& err) {
  const std::string n(name);
  if (n.rfind("quantize.", 0) == 0) return 0;                                    // derived buffers
  if (n.rfind("encoder.", 0) == 0 && !d->has_enc) return 0;                       // encode half not built in this engine
  size_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
  std::vector<Conv*> convs{&d->conv_in, &d->conv_out};
  std::vector<Norm*> norms{&d->norm_out};
  auto add_block = [&](ResBlock& rb) {
    convs.push_back(&rb.c1); convs.push_back(&rb.c2); if (rb.has_sc) convs.push_back(&rb.sc);
    norms.push_back(&rb.n1); norms.push_back(&rb.n2);
  };
  for (auto& rb : d->mid) add_block(rb);
  for (auto& st : d->up) { for (auto& rb : st.blocks) add_block(rb); if (st.has_up) convs.push_back(&st.up); }
  if (d->has_enc) {
    convs.push_back(&d->e_conv_in); convs.push_back(&d->e_conv_out); norms.push_back(&d->e_norm_out);
    for (auto& rb : d->e_mid) add_block(rb);
    for (auto& st : d->e_down) { for (auto& rb : st.blocks) add_block(rb); if (st.has_up && d->c.sample_with_conv) convs.push_back(&st.up); }
  }
  for (Conv* c : convs) {
    Conv* hit = nullptr; bool is_bias = false;
    if (!find_conv(*c, n, &hit, &is_bias)) continue;
    if (is_bias) {
      if (numel != (size_t)c->cout_w) { err = n + ": wrong bias size"; return -4; }
      if (hipMemcpyAsync(c->b, data, numel * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) { err = "copy failed"; return -10; }
    } else if (c->down) {
      if (numel != (size_t)c->cout_w * c->cin * 9) { err = n + ": wrong weight size"; return -4; }
      hipLaunchKernelGGL(repack_down_kernel, dim3(512), dim3(256), 0, s, data, c->w, c->cout_w, c->cin, c->cout_pad);
    } else {
      if (numel != (size_t)c->cout_w * c->cin * c->ks * c->ks) { err = n + ": wrong weight size"; return -4; }
      hipLaunchKernelGGL(repack_conv_kernel, dim3(512), dim3(256), 0, s, data, c->w, c->cout_w, c->cin, c->ks, c->cout_pad, c->cin_pad);
    }
    return 0;
  }
  for (Norm* nm : norms) {
    float* dst = nullptr;
    if (!find_norm(*nm, n, &dst)) continue;
    if (numel != (size_t)nm->c) { err = n + ": wrong norm size"; return -4; }
    if (hipMemcpyAsync(dst, data, numel * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) { err = "copy failed"; return -10; }
    return 0;
  }
  err = "unknown checkpoint entry '" + n + "'";
  return -2;
}

int dec_decode(mb_dec* d, const int64_t* tokens, float* img_nchw, uint8_t* img_nhwc_u8, int B, hipStream_t s, std::string& err) {
  if (B <= 0 || B > d->max_batch) { err = "batch outside [1, max_batch]"; return -1; }
  const mb_dec_cfg& c = d->c;
  int res = c.latent_size;
  const size_t npix = (size_t)B * res * res;
  hipLaunchKernelGGL(latent_kernel, dim3((unsigned)std::min<size_t>(2048, (npix * CK + 255) / 256)), dim3(256), 0, s,
                     tokens, d->z, npix, c.token_size);
  launch_conv(s, d, d->conv_in, d->z, nullptr, nullptr, d->buf[0], nullptr, nullptr, B, res, res, false);
  int xi = 0;
  for (auto& rb : d->mid) xi = run_block(s, d, rb, xi, B, res, res);
  for (auto& st : d->up) {
    for (auto& rb : st.blocks) xi = run_block(s, d, rb, xi, B, res, res);
    if (st.has_up) {
      res *= 2;
      const int t = (xi + 1) % 3;
      launch_conv(s, d, st.up, d->buf[xi], nullptr, nullptr, d->buf[t], nullptr, nullptr, B, res, res, false);
      xi = t;
    }
  }
  launch_gn(s, d, d->norm_out, d->buf[xi], B, res * res);
  launch_conv(s, d, d->conv_out, d->buf[xi], d->gn_ss, nullptr, nullptr, img_nchw, img_nhwc_u8, B, res, res, true);
  return 0;
}

// ConvVQModel.encode (conv_vqgan.py:70-83): image [B,C,H,W] fp32 -> code indices [B, h*w] (+ optional +-1 latent / raw z, fp32 NCHW)
int dec_saturation_count(mb_dec* d, unsigned* count, bool reset, hipStream_t s) {
  if (hipMemcpyAsync(count, d->sat, sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess) return -10;
  if (reset && hipMemsetAsync(d->sat, 0, sizeof(unsigned), s) != hipSuccess) return -10;
  return hipStreamSynchronize(s) == hipSuccess ? 0 : -10;
}

int enc_encode(mb_dec* d, const float* img, int64_t* indices, float* zq, float* zraw, int B, hipStream_t s, std::string& err) {
  if (!d->has_enc) { err = "this engine was created without the encoder half (mb_dec_cfg.build_encoder)"; return -1; }
  if (B <= 0 || B > d->max_batch) { err = "batch outside [1, max_batch]"; return -1; }
  const mb_dec_cfg& c = d->c;
  int res = d->out_res;
  const size_t npix = (size_t)B * res * res;
  hipLaunchKernelGGL(pack_image_kernel, dim3((unsigned)std::min<size_t>(4096, (npix * 8 + 255) / 256)), dim3(256), 0, s,
                     img, d->buf[2], B, c.num_channels, res, res);
  launch_conv(s, d, d->e_conv_in, d->buf[2], nullptr, nullptr, d->buf[0], nullptr, nullptr, B, res, res, false);
  int xi = 0;
  for (auto& st : d->e_down) {
    for (auto& rb : st.blocks) xi = run_block(s, d, rb, xi, B, res, res);
    if (st.has_up) {
      const int t = (xi + 1) % 3, t2 = (xi + 2) % 3;
      const size_t n8 = (size_t)B * res * res * (st.up.cin / 8);
      if (!c.sample_with_conv) {                           // F.avg_pool2d(2, 2) (autoencoder.py:182)
        hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)std::min<size_t>(4096, (n8 / 4 + 255) / 256)), dim3(256), 0, s, d->buf[xi], d->buf[t], B, res,
                           res, st.up.cin);
        d->gn_of = nullptr;                             // (no conv wrote this tensor: its GroupNorm takes the separate sweep)
        res /= 2;
        xi = t;
        continue;
      }
      hipLaunchKernelGGL(s2d_kernel, dim3((unsigned)std::min<size_t>(4096, (n8 + 255) / 256)), dim3(256), 0, s, d->buf[xi], d->buf[t], B, res, res,
                         st.up.cin);
      res /= 2;
      launch_conv(s, d, st.up, d->buf[t], nullptr, nullptr, d->buf[t2], nullptr, nullptr, B, res, res, false);
      xi = t2;
    }
  }
  for (auto& rb : d->e_mid) xi = run_block(s, d, rb, xi, B, res, res);
  launch_gn(s, d, d->e_norm_out, d->buf[xi], B, res * res);
  const int t = (xi + 1) % 3;
  launch_conv(s, d, d->e_conv_out, d->buf[xi], d->gn_ss, nullptr, d->buf[t], nullptr, nullptr, B, res, res, false);
  const size_t np = (size_t)B * res * res;
  hipLaunchKernelGGL(lfq_kernel, dim3((unsigned)std::min<size_t>(1024, (np + 255) / 256)), dim3(256), 0, s, d->buf[t], indices, zq, zraw, B, res * res,
                     c.token_size, d->e_conv_out.cout);
  return 0;
}

}  // namespace mb

#ifdef MB_CONV_TRACE
extern "C" int mb_debug_conv_trace(long long* p, int sel) { mb::g_conv_trace = p; mb::g_conv_trace_sel = sel; mb::g_conv_trace_n = 0; return 0; }
#endif
