// Bidirectional multi-head self-attention for one (sequence, head) per workgroup.
//
// Replaces nn.MultiheadAttention's fast path inside BertAttention (modeling/bert.py:84,137):
// softmax(Q K^T / sqrt(dh)) V over N = seq+1 = 257 tokens, heads = contiguous dh-wide slices of the
// packed in_proj output qkv[row, 0:d | d:2d | 2d:3d].
//
// N = 257 fits on chip, so there is no online softmax: K (row-major, XOR-swizzled 16-B slots) and
// V^T (transposed while staging) for the whole head sit in LDS (~73 KiB at dh = 64 -> 2 WGs/CU);
// each wave owns 16-query tiles.  Scores are computed TRANSPOSED, S^T = K Q^T
// (mfma A = K tile, B = Q tile), so a lane holds one query column (q = lane&15) and 4 keys per
// 16-key tile: the row max / row sum are in-lane reductions plus two xor-shuffles, and the
// probabilities are already in MFMA B-operand order for O^T = V^T P^T (k index = key, with the
// 32-key k-block taking keys {tile 2j: g*4..g*4+3, tile 2j+1: g*4..g*4+3} for lane group g -- the
// V^T fragment is read with the same permutation, so P never moves between lanes).  The output
// lane owns O[q = lane&15][4 consecutive dh] -> 8-byte row-major stores.
#include "mb_kernels.h"

namespace mb {

constexpr int ATT_NKT = 18;              // key tiles of 16 -> up to 288 keys
constexpr int ATT_NP = ATT_NKT * 16;     // padded key count
constexpr int ATT_KP = 296;              // V^T row pitch in elements: (KP/2) % 64 == 20 -> conflict-free b64 reads

template <int DH>
__global__ __launch_bounds__(256, 2) void attention_kernel(const h16* __restrict__ qkv, h16* __restrict__ out,
                                                          int N, int d, int heads, float scale) {
  constexpr int KROW = DH * 2;           // bytes per K row
  constexpr int SL = DH / 8;             // 16-byte slots per K row
  constexpr int KS = DH / 32;            // k-steps of the QK^T MFMA
  constexpr int NT = DH / 16;            // output dh tiles
  __shared__ __attribute__((aligned(16))) char smem[ATT_NP * KROW + DH * ATT_KP * 2];
  char* Ks = smem;
  uint32_t* Vt32 = (uint32_t*)(smem + ATT_NP * KROW);
  const char* Vt = smem + ATT_NP * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sq = blockIdx.x / heads, h = blockIdx.x - sq * heads;
  const size_t rs = (size_t)3 * d;                                   // qkv row stride (elements)
  const h16* base = qkv + (size_t)sq * N * rs + h * DH;

  auto kswz = [](int row, int s) { return SL == 8 ? (s ^ ((row >> 1) & 7)) : (s ^ ((row >> 2) & 3)); };

  // ---- stage K: [key][dh] row-major, zero-filled beyond N
  for (int c = tid; c < ATT_NP * SL; c += 256) {
    const int row = c / SL, p = c - row * SL;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N) v = *(const uint4*)(base + (size_t)row * rs + d + kswz(row, p) * 8);
    *(uint4*)(Ks + row * KROW + p * 16) = v;
  }
  // ---- stage V^T: Vt[dh][key]; each item = (key pair j, 8-wide dh slice) -> 8 packed 32-bit writes
  constexpr int NPAIR = ATT_NP / 2;
  for (int it = tid; it < NPAIR * SL; it += 256) {
    const int sl = it / NPAIR, j = it - sl * NPAIR;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (2 * j < N) r0 = *(const uint4*)(base + (size_t)(2 * j) * rs + 2 * d + sl * 8);
    if (2 * j + 1 < N) r1 = *(const uint4*)(base + (size_t)(2 * j + 1) * rs + 2 * d + sl * 8);
    const uint32_t a[4] = {r0.x, r0.y, r0.z, r0.w}, b[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = sl * 8 + 2 * e;
      Vt32[(c * ATT_KP) / 2 + j] = (a[e] & 0xffffu) | (b[e] << 16);
      Vt32[((c + 1) * ATT_KP) / 2 + j] = (a[e] >> 16) | (b[e] & 0xffff0000u);
    }
  }
  __syncthreads();

  const int l15 = lane & 15, g = lane >> 4;
  const int nqt = (N + 15) / 16;
  for (int qt = wave; qt < nqt; qt += 4) {
    // ---- Q fragments straight from global (each element is used once per workgroup)
    const int qrow = min(qt * 16 + l15, N - 1);
    h16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const h16x8*)(base + (size_t)qrow * rs + (ks * 4 + g) * 8);

    // ---- S^T tiles: s[kt][r] = S[q = l15][key = kt*16 + g*4 + r]
    f32x4 s[ATT_NKT];
#pragma unroll
    for (int kt = 0; kt < ATT_NKT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = kt * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const h16x8 kf = *(const h16x8*)(Ks + row * KROW + kswz(row, ks * 4 + g) * 16);
        s[kt] = MB_MFMA_16x16x32(kf, qf[ks], s[kt]);
      }
      if (kt % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // bound the fragment prefetch depth (VGPR budget)
    }
    // ---- softmax over keys (fp32)
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < ATT_NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + g * 4 + r;
        s[kt][r] = key < N ? s[kt][r] : -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < ATT_NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf((s[kt][r] - mx) * scale);
        s[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;

    // ---- O^T = V^T P^T
    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < ATT_NKT / 2; ++kb) {
      const f32x4 p0 = s[2 * kb], p1 = s[2 * kb + 1];
      const h16x8 pf = {to_h(p0[0]), to_h(p0[1]), to_h(p0[2]), to_h(p0[3]),
                         to_h(p1[0]), to_h(p1[1]), to_h(p1[2]), to_h(p1[3])};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const char* vr = Vt + (size_t)(nt * 16 + l15) * (ATT_KP * 2) + (kb * 32 + g * 4) * 2;
        const h16x4 v0 = *(const h16x4*)vr;
        const h16x4 v1 = *(const h16x4*)(vr + 32);
        const h16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[nt] = MB_MFMA_16x16x32(vf, pf, o[nt]);
      }
      if (kb % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- o[nt][r] = O[q = l15][dh = nt*16 + g*4 + r]
    const int q = qt * 16 + l15;
    if (q < N) {
      h16* orow = out + ((size_t)sq * N + q) * d + h * DH;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *(h16x4*)(orow + nt * 16 + g * 4) =
            h16x4{to_h(o[nt][0] * inv), to_h(o[nt][1] * inv), to_h(o[nt][2] * inv), to_h(o[nt][3] * inv)};
    }
  }
}

void attention(hipStream_t s, const h16* qkv, h16* out, int nb, int N, int d, int heads) {
  const int dh = d / heads;
  const float scale = 1.0f / sqrtf((float)dh);
  dim3 grid(nb * heads), block(256);
  if (dh == 64) hipLaunchKernelGGL(attention_kernel<64>, grid, block, 0, s, qkv, out, N, d, heads, scale);
  else hipLaunchKernelGGL(attention_kernel<32>, grid, block, 0, s, qkv, out, N, d, heads, scale);
}

}  // namespace mb
