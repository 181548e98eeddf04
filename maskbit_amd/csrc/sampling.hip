// MaskBit sampling step: everything sample() does between the transformer forward and the
// next forward (modeling/modules/sampling.py:90-131):
//   CFG combine -> softmax -> categorical draw as argmax(p / Exp(1)) (= torch.multinomial(n=1)) ->
//   keep already-decoded tokens -> confidence log p[pred] + scaled Gumbel noise (+inf for decoded
//   positions) -> k-th smallest confidence per image -> re-mask everything <= threshold.
// One 64-lane wave per (position, group) row: the row's C logits live one-per-lane (C/64 per lane
// when C > 64), so max / sum / argmax are wavefront reductions.  The k-th order statistic over the
// n*m confidences is found by rank counting out of LDS (exactly torch.sort(...)[k-1], ties and
// infinities included).  All arithmetic is fp32, as in the reference.
#include "mb_kernels.h"

namespace mb {

// Two launches (round 3; one workgroup per image did both parts, i.e. 64 of 256 CUs ran 32 latency-bound rows per wave: 100 us per step):
//   sample_rows_kernel   -- one wave per (image, position, group) row, 16 rows per 4-wave workgroup over the whole chip: guidance, softmax, draw,
//                           confidence.  Result packed into the int64 slot of tokens_out: low dword = the confidence's float bits, high dword = pred.
//   sample_thresh_kernel -- one workgroup per image: unpack into LDS, k-th smallest confidence by rank counting, re-mask, write tokens (+ pred).
// The arithmetic per row and the order statistic are unchanged (bit-exact with the oracle: tests/test_hip_parity.py).
template <int CPL>   // logits per lane: C <= 64*CPL
__global__ __launch_bounds__(256) void sample_rows_kernel(StepArgs a, const int64_t* __restrict__ tokens_in) {
  const int C = a.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t mask_tok = (int64_t)C;
  const size_t nrows = (size_t)a.B * a.P;
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    const size_t row = (size_t)blockIdx.x * 16 + wave * 4 + it;
    if (row >= nrows) return;
    const float* lc = a.logits_c + row * C;
    const float* lu = a.logits_u ? a.logits_u + row * C : nullptr;
    const float* qn = a.exp_noise + row * C;
    float l[CPL];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      float v = -INFINITY;
      if (c < C) {
        v = lc[c];
        if (lu) v = __fadd_rn(v, __fmul_rn(a.scale, __fsub_rn(v, lu[c])));   // sampling.py:98-99, no FMA contraction
        v = v / a.temperature;                          // :105
      }
      l[i] = v;
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { l[i] = (lane + 64 * i < C) ? expf(l[i] - mx) : 0.f; sum += l[i]; }
    sum = wave_sum(sum);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { l[i] = l[i] / sum; psum += l[i]; }   // probabilities
    psum = wave_sum(psum);                                               // Categorical re-normalises
    float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      if (c < C) {
        const float ratio = (l[i] / psum) / qn[c];
        if (ratio > best) { best = ratio; bi = c; }      // strict '>' keeps the lowest index in-lane
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    const int64_t tin = tokens_in[row];
    const bool masked = tin == mask_tok;
    const int pred = masked ? bi : (int)tin;                             // :111
    // p[pred]: owned by lane pred%64, slot pred/64
    float pv = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) if (pred == lane + 64 * i) pv = l[i];
    pv = __shfl(pv, pred & 63);
    if (lane == 0) {
      const float conf = (masked ? logf(pv) : INFINITY) + a.conf_noise[row];    // :113-118
      a.tokens[row] = (int64_t)(((uint64_t)(uint32_t)pred << 32) | (uint64_t)__float_as_uint(conf));
    }
  }
}

__global__ __launch_bounds__(1024) void sample_thresh_kernel(StepArgs a, const int64_t* __restrict__ tokens_in) {
  extern __shared__ float sm[];
  const int P = a.P;
  float* conf_s = sm;                    // [P]
  int* pred_s = (int*)(sm + P);          // [P]
  int* cnt_s = pred_s + P;               // [16]: per-wave partial counts
  float* thr_s = (float*)(cnt_s + 16);   // [1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int b = blockIdx.x;
  const int64_t mask_tok = (int64_t)a.C;
  // num_masked of SAMPLE 0 (sampling.py:109 reads index [0] for the whole batch)
  int mycnt = 0;
  for (int p = tid; p < P; p += blockDim.x) mycnt += tokens_in[p] == mask_tok;
  mycnt = (int)wave_sum((float)mycnt);
  if (lane == 0) cnt_s[wave] = mycnt;
  if (tid == 0) *thr_s = -INFINITY;
  for (int p = tid; p < P; p += blockDim.x) {        // everything of this image is in LDS before any of its slots is overwritten
    const uint64_t v = (uint64_t)a.tokens[(size_t)b * P + p];
    conf_s[p] = __uint_as_float((uint32_t)v);
    pred_s[p] = (int)(uint32_t)(v >> 32);
  }
  __syncthreads();
  int nm = 0;
  for (int w = 0; w < nw; ++w) nm += cnt_s[w];
  // k = clamp(floor(ratio*P), 1, num_masked-1), threshold = sorted[k-1] (python index, may wrap)
  int k = min(max(a.k_mask_len, 1), nm - 1);
  int idx = k - 1;
  if (idx < 0) idx += P;
  for (int p = tid; p < P; p += blockDim.x) {
    const float x = conf_s[p];
    int lt = 0, le = 0;
    for (int j = 0; j < P; ++j) { const float y = conf_s[j]; lt += y < x; le += y <= x; }
    if (lt <= idx && idx < le) *thr_s = x;
  }
  __syncthreads();
  const float thr = *thr_s;
  for (int p = tid; p < P; p += blockDim.x) {
    const size_t row = (size_t)b * P + p;
    const int pr = pred_s[p];
    a.tokens[row] = conf_s[p] <= thr ? mask_tok : (int64_t)pr;           // :128-129
    if (a.pred) a.pred[row] = (int64_t)pr;
  }
}

int sample_step(hipStream_t s, const StepArgs& a, const int64_t* tokens_in) {
  if (a.C > 4096 || a.P > 8192) return -1;
  const size_t nrows = (size_t)a.B * a.P;
  dim3 grid((unsigned)((nrows + 15) / 16)), block(256);
  if (a.C <= 64) hipLaunchKernelGGL(sample_rows_kernel<1>, grid, block, 0, s, a, tokens_in);
  else if (a.C <= 128) hipLaunchKernelGGL(sample_rows_kernel<2>, grid, block, 0, s, a, tokens_in);
  else if (a.C <= 256) hipLaunchKernelGGL(sample_rows_kernel<4>, grid, block, 0, s, a, tokens_in);
  else if (a.C <= 512) hipLaunchKernelGGL(sample_rows_kernel<8>, grid, block, 0, s, a, tokens_in);
  else if (a.C <= 1024) hipLaunchKernelGGL(sample_rows_kernel<16>, grid, block, 0, s, a, tokens_in);   // single-group codebooks (codebook_splits = 1)
  else if (a.C <= 2048) hipLaunchKernelGGL(sample_rows_kernel<32>, grid, block, 0, s, a, tokens_in);
  else hipLaunchKernelGGL(sample_rows_kernel<64>, grid, block, 0, s, a, tokens_in);
  const size_t shm = (size_t)a.P * 8 + 16 * 4 + 16;
  hipLaunchKernelGGL(sample_thresh_kernel, dim3(a.B), dim3(a.P >= 1024 ? 1024 : 512), shm, s, a, tokens_in);
  return 0;
}

__global__ void fill_i64_kernel(int64_t* dst, int64_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
void fill_i64(hipStream_t s, int64_t* dst, int64_t value, size_t n) {
  hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)min((size_t)1024, (n + 255) / 256)), dim3(256), 0, s, dst, value, n);
}

// combine_factorized_tokens (factorization.py:7-24): sum_g tok[..., g] << (g * K/m), kept integral.
__global__ void combine_groups_kernel(const int64_t* __restrict__ tok, int64_t* __restrict__ codes, size_t rows, int m, int gbits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    int64_t v = 0;
    for (int g = 0; g < m; ++g) v += tok[i * m + g] << (g * gbits);
    codes[i] = v;
  }
}
void combine_groups(hipStream_t s, const int64_t* tokens, int64_t* codes, size_t rows, int m, int gbits) {
  hipLaunchKernelGGL(combine_groups_kernel, dim3((unsigned)min((size_t)1024, (rows + 255) / 256)), dim3(256), 0, s,
                     tokens, codes, rows, m, gbits);
}

}  // namespace mb
