// Internal launcher declarations shared between the kernel translation units and engine.hip.
#pragma once
#include "mb_common.h"

namespace mb {

// ---- GEMM  out[M,N] = A[M,K] * W[N,K]^T + bias (+ epilogue) ------------------------------------
enum GemmEpi {
  EPI_H16 = 0,         // h16 out = acc + bias                      (QKV projection)
  EPI_GELU_H16 = 1,    // h16 out = gelu_erf(acc + bias)            (FFN up projection)
  EPI_RES_F32 = 2,      // f32 out  = acc + bias + residual           (attention out-proj, FFN down)
  EPI_GELU_F32 = 3,     // f32 out  = gelu_erf(acc + bias)            (last_layer.0)
  EPI_LOGITS_F32 = 4,   // f32 out, rows with (m % period)==period-1 dropped, rest compacted (head)
};
struct GemmArgs {
  const h16* A;        // [M,K] row-major
  const h16* W;        // [N,K] row-major (torch Linear layout)
  const float* bias;    // [N]
  const float* residual;// [M,N] fp32 or null
  float* out_f32;
  h16* out_h16;
  int M, N, K;
  int period;           // EPI_LOGITS_F32 only: tokens per sequence incl. the class row
  // out = acc * (*scale) + bias for weights stored pre-scaled by a power of two (the head GEMMs' hi / lo planes, W2 below); null: 1
  const float* scale = nullptr;
  // EPI_RES_F32 with ln_stats != null: `residual` holds the PRE-LayerNorm rows y; the residual that is added is
  // LayerNorm(y) = (y - mean) * rstd * ln_g + ln_b re-derived from ln_stats[m] = {mean, rstd} (what layernorm_rows
  // wrote) -- bit-identical to the fp32 rows that kernel would have stored.  out_f32 may alias residual (in place).
  const float* ln_stats = nullptr;
  const float* ln_g = nullptr;
  const float* ln_b = nullptr;
  // EPI_LOGITS_F32: bias is [period - 1][N] (one row per position: Bert's per-position output bias, bert.py:262,332) when set
  int bias_per_pos = 0;
  // Split-activation ("fp16 hi+lo") GEMMs: A2 = the lo halves x - fp16(x) of the activations whose fp16 hi halves are A, both
  // [M, kw]; W has kw = K/2 columns and is swept twice (K-tiles below K/2 pair A with W, the others A2 with W), so both
  // products land in the same fp32 accumulator.  A2 = null / kw = 0: plain GEMM.
  const h16* A2 = nullptr;
  int kw = 0;
  // ... with W2 != null as well (128x128 kernel only; the two head GEMMs): THREE sweeps of kw columns, K = 3 kw -- (A, W), (A2, W), (A, W2) with
  // W = fp16(w * 2^S), W2 = fp16(w * 2^S - W) [N, kw] each and *scale = 2^-S: hi + lo inputs against hi + lo weights (the lo x lo term, 2^-22, is dropped)
  const h16* W2 = nullptr;
  // "CFG pair" GEMM (sequence-aligned half-tile kernel only): the M = 2 * pair_rows rows are pair_rows conditional rows followed by their
  // unconditional twins (whole sequences); in A the unconditional rows hold the difference operand fp16(x_u - x_c).  Output
  // rows: out_c = f(A_c . W), out_u = f(A_c . W + A_delta . W) -- with the GELU epilogue the unconditional rows receive gelu(u) - gelu(c),
  // i.e. the next GEMM's difference operand.  See gemm_ht.hip and DESIGN.md "Precision".
  int pair_rows = 0;
  // sequence tiles: rows per sequence incl. the class token (0 = 257).  A pair tile is 128 tokens of one sequence pair, a plain tile 256 tokens of one
  // sequence, so any (seq_rows - 1) / 128 (/ 256) that is a power of two is served: 257 (256 x 256 images) and 1 025 (the 512 x 512 models,
  // scripts/eval_maskbit.py:125,139-144); the LAST tile of a sequence (pair) stores the class row(s).
  int seq_rows = 0;
  // GELU epilogue of sequence-aligned tiles (optional): out4 / out4_scale receive e2m1 of the (conditional) OUTPUT values (row stride 2N bytes) and
  // their lane-ordered scale bytes -- the token operand of the next GEMM's weight-correction pass; class-token rows are left untouched.
  uint8_t* out4 = nullptr;
  uint8_t* out4_scale = nullptr;
  // ... and (optional, with out4; precision 4) out4l / out4l_scale: the same for the fp16 LO HALVES v - fp16(v) of those outputs, the token operand of the
  // next GEMM's activation-lo pass
  uint8_t* out4l = nullptr;
  uint8_t* out4l_scale = nullptr;
  // MX-fp4 MINI-TILE correction passes (sequence-aligned half-tile kernel, K % 128 == 0): mini-tiles of 128 token rows x 256 weight rows x 128
  // K-elements (24 KiB of LDS behind the two K-tile parities), staged while the fp16 K-tiles run and multiplied in a fifth phase between them
  // (16 v_mfma_scale_f32_16x16x128_f8f6f4 per wave).  nlo = number of operand sets:
  //   pair tiles : lo[0] (, lo[1]) on the CONDITIONAL rows: K / 128 mini-tiles per set (one per two fp16 K-tiles with one set, one per K-tile with two);
  //   plain tiles: lo[0] on both 128-row halves of the 256-token sequence tile (nlo = 1): 2 K / 128 mini-tiles, one per fp16 K-tile.
  // Operands: A4 = e2m1 token operand, two values per byte, row stride 2 K bytes (first K / 2 used); W4 = e2m1 weight operand, mini-tile-packed
  // (w4_packed_offset; N K / 2 bytes); w_scale = the weights' E8M0 bytes, ONE PER (weight row, 128 K-elements) = per (row, mini-tile), in the kernel's
  // lane order (w4_scale_index: a lane's dword of mini-tile j holds the bytes of its four n-tiles; N K / 128 bytes -- round 6; rounds 2-5 had one byte
  // per weight row: on heavy-tailed weights a row's largest rounding error then took the resolution of all its other columns); a_scale = the token operand's E8M0 bytes per (row, 64 K-elements) in LANE ORDER (fp4_scale_index: one dword per lane holds the
  // scales of its four m-tiles).  Class-token rows take no part in these passes.
  struct LoSet { const uint8_t* A4; const uint8_t* a_scale; const uint8_t* W4; const uint8_t* w_scale; };
  LoSet lo[2] = {};
  int nlo = 0;
  // fp16 epilogues (optional): lanes that stored a value outside fp16's range (or a NaN) add 1 to *sat (mb_gen_saturation_count)
  unsigned* sat = nullptr;
};
// Byte offset of element (row n, K-element k) of an e2m1 WEIGHT operand of the mini-tile passes.  The operand is stored MINI-TILE-PACKED:
// [N / 16][K / 128] chunks of 1 KiB = 16 rows x 64 B, the 16-byte pieces of a row swizzled with (row >> 1) & 3 -- the LDS image of one DMA
// instruction, so that every instruction of a weight mini-tile reads 1 KiB of CONTIGUOUS memory (8 full 128-byte lines) instead of 16 half lines
// of a row-major operand (measured: the L2 -> LDS path pays per line, not per byte -- profiles/r04_gemm_minitiles.md).  N % 16 == 0, K % 128 == 0.
__host__ __device__ inline size_t w4_packed_offset(int n, int k, int K) {
  const int r = n & 15, c = (k & 127) >> 5;
  return ((size_t)(n >> 4) * (K >> 7) + (k >> 7)) * 1024 + r * 64 + ((c ^ ((r >> 1) & 3)) << 4) + ((k & 31) >> 1);
}
// byte index of the E8M0 scale of (weight row n, K-elements [128 j, 128 j + 128)) in the lane-ordered weight scale arrays of the mini-tile passes
// (GemmArgs.lo w_scale; K % 128 == 0, N % 64 == 0): dword ((n >> 6) * (K / 128) + j) * 16 + (n & 15) -- what lane (n & 15) of the wave column (n >> 6)
// loads for mini-tile j --, byte (n >> 4) & 3 = its n-tile
__host__ __device__ inline size_t w4_scale_index(int n, int j, int K) {
  return (((size_t)(n >> 6) * (K >> 7) + j) * 16 + (n & 15)) * 4 + ((n >> 4) & 3);
}
// byte index of the scale of (token row r of sequence seq, 64-column block blk) in the lane-ordered scale arrays of the mini-tile passes
// (groups = 64-token groups per sequence: 4 for the 256-token models, 16 for the 1024-token ones of 512 x 512 images)
__host__ __device__ inline size_t fp4_scale_index(int blk, int nseq, int seq, int r, int groups = 4) {
  return (((size_t)blk * nseq + seq) * groups + (r >> 6)) * 64 + (r & 15) * 4 + ((r >> 4) & 3);
}
int gemm_tn(hipStream_t s, GemmEpi epi, const GemmArgs& a, int variant = 0);   // 0, or -1: lo-pass request outside the half-tile kernel's shapes
bool gemm_ht_supported(GemmEpi epi, const GemmArgs& a);
void gemm_ht(hipStream_t s, GemmEpi epi, const GemmArgs& a, int mt);
void set_cu_count(int n);   // persistent grids are sized for n CUs (0 = the device's count): for launches on CU-masked streams

// e2m1 copy of a weight for the mini-tile passes: e2m1(fp16(W[n][k]) * 2^r) in the mini-tile-packed layout (w4_packed_offset; N K / 2 bytes), r per
// (row, 128 columns) chosen to minimise the block's quantisation error; scale_out [N K / 128] in the kernel's lane order (w4_scale_index).  N % 64 == 0, K % 128 == 0.
void w4_from_f32(hipStream_t s, const float* src, uint8_t* dst4, int N, int K, uint8_t* scale_out);
// the same layout for the weight's fp16 ROUNDING ERROR: e2m1((W - fp16(W)) * 2^r), r from the (row, 128-column) block's largest |error| (no saturation)
void w4lo_from_f32(hipStream_t s, const float* src, uint8_t* dst4, int N, int K, uint8_t* scale_out);

// e2m1 copies written by the row-wise producers for the trunk GEMMs' mini-tile passes (GemmArgs.lo): the rows' VALUES (x4, with the
// lane-ordered per-(row, 64 columns) scale bytes x4s) and / or their fp16 LO HALVES x - fp16(x) (xl4 / xl4s); rows are tokens of nseq sequences of
// 257 (row stride 2d bytes, first d / 2 used; class-token rows are skipped).  hidden = 768 / 1024 only.
struct Fp4Rows { uint8_t* x4 = nullptr; uint8_t* x4s = nullptr; uint8_t* xl4 = nullptr; uint8_t* xl4s = nullptr; int nseq = 0; int seq_rows = 257; };   // seq_rows: tokens per sequence incl. the class token

// ---- LayerNorm over rows of y[M,d] -> x_f32 (optional), x_h16 (optional), stats[M][2] = {mean, rstd} (optional) ---
void layernorm_rows(hipStream_t s, const float* y, const float* gamma, const float* beta, float eps,
                    float* x_f32, h16* x_h16, float* stats, int M, int d, h16* x_lo = nullptr,
                    const Fp4Rows& f4 = Fp4Rows{});   // x_lo: fp16(x - fp16(x)), optional

// "CFG pair" forms (hidden = 768 / 1024 only; -1 otherwise): rows r < P are conditional, r + P their unconditional twins.  Writes
// x_h16[r] = fp16(x_c), x_h16[r + P] = fp16(x_u - x_c), both rows' {mean, rstd}; optional f4: e2m1 copies of the CONDITIONAL rows.
int layernorm_pair(hipStream_t s, const float* y, const float* gamma, const float* beta, float eps, h16* x_h16, float* stats, int P, int d,
                   const Fp4Rows& f4 = Fp4Rows{});
int pairify_rows(hipStream_t s, const float* x32, h16* x_h16, int P, int d, const Fp4Rows& f4 = Fp4Rows{});

// ---- bit-token embed + class token + pos-emb + first LayerNorm (bert.py:440-454, 482-496) -------
struct EmbedArgs {
  const int64_t* tokens;   // [nb, seq, m]
  const int64_t* labels;   // [nb]
  const uint8_t* drop;     // [nb] or null
  const float* w_in;       // [K, d]  (input_proj.weight transposed at load)
  const float* b_in;       // [d]
  const float* class_emb;  // [nclass+1, d]
  const float* pos;        // [seq+1, d]
  const float* gamma; const float* beta;
  float* x_f32; h16* x_h16;   // [nb*(seq+1), d]
  int nb, seq, m, gbits, d, nclass;
  // Bert (modeling/bert.py:313-315): per-group embedding tables [m][2^gbits + 1][d] summed instead of the bit projection
  const float* tables = nullptr;
  h16* x_lo = nullptr;     // optional: lo halves of x_h16 (split-activation GEMMs)
  Fp4Rows f4;              // optional: e2m1 copies for the mini-tile passes (embed_pair: of the conditional rows)
};
void embed_ln(hipStream_t s, const EmbedArgs& a);
// guided forward: a.nb = B conditional sequences (tokens [B, seq, m], labels [B]); rows of the label-dropped twins are written B * (seq + 1) rows further
// down; x_h16 receives the pair operands (fp16(x_c) | fp16(x_u - x_c)), a.f4 (optional) the e2m1 copies of the conditional rows.  -1: shape not
// served (embedding tables, widths other than 768 / 1024) -> embed_ln over [cond | twins] + pairify_rows
int embed_pair(hipStream_t s, const EmbedArgs& a);
void transpose_f32(hipStream_t s, const float* src /*[rows,cols]*/, float* dst /*[cols,rows]*/, int rows, int cols);

// ---- multi-head self-attention over packed qkv [nb*N, 3d] -> out [nb*N, d] -----------------------
void attention(hipStream_t s, const h16* qkv, h16* out, int nb, int N, int d, int heads,
               uint8_t* out4 = nullptr, uint8_t* out4s = nullptr);   // out4 / out4s (N = 257 or (N - 1) % 64 == 0 beyond the one-block kernel's length; head width 64): e2m1 of the outputs + lane-ordered scale
                                                                      // bytes for the out-proj GEMM's mini-tile pass
// "CFG pair" attention: sequences [0, P) are conditional, [P, 2P) their unconditional twins; one workgroup runs a (pair, head) -- conditional pass,
// then the twin with the conditional output tiles kept in registers -- and writes out[r + P*N] = fp16(att_u - att_c), the difference operand of the
// out-proj pair GEMM.  N <= 288: K / V of a head in LDS; longer sequences (the 1024 + 1-token models): the streaming kernel in pair form.
// out4 / out4s (head dimension 64 only): e2m1 of the conditional output values, row stride 2d bytes, + lane-ordered E8M0 scale bytes (GemmArgs.lo)
// out4l / out4ls (optional, with out4; precision 4): the same for the fp16 lo halves o_c - fp16(o_c) of the conditional outputs (the out-proj GEMM's activation-lo pass)
int attention_pair(hipStream_t s, const h16* qkv, h16* out, int P, int N, int d, int heads, uint8_t* out4 = nullptr, uint8_t* out4s = nullptr,
                   uint8_t* out4l = nullptr, uint8_t* out4ls = nullptr);
// head-averaged attention weights [nb, N, N] fp32 of one layer (return_attn=True); -1 if the shape is not supported
int attention_probs(hipStream_t s, const h16* qkv, float* out, int nb, int N, int d, int heads);

// ---- fused sampling step (sampling.py:90-131) ----------------------------------------------------
struct StepArgs {
  const float* logits_c; const float* logits_u;   // [B, n*m, C]
  float scale, temperature;
  const float* exp_noise;    // [B*n*m, C]
  const float* conf_noise;   // [B, n*m]
  int k_mask_len;
  int64_t* tokens;           // [B, n*m] out (must not alias tokens_in)
  int64_t* pred;             // [B, n*m] out (may be null)
  int B, P /* n*m */, C;
};
int sample_step(hipStream_t s, const StepArgs& a, const int64_t* tokens_in);

// ---- small integer helpers of the loop (sampling.py:65, factorization.py:7-24) -------------------
void fill_i64(hipStream_t s, int64_t* dst, int64_t value, size_t n);
void combine_groups(hipStream_t s, const int64_t* tokens /*[rows,m]*/, int64_t* codes /*[rows]*/, size_t rows, int m, int gbits);

// ---- fp32 -> h16 repack ------------------------------------------------------------------------
void cast_f32_to_h16(hipStream_t s, const float* src, h16* dst, size_t n);
// W[N,K] fp32 -> two planes hi[N,K] = fp16(W*2^S), lo[N,K] = fp16(W*2^S - hi) (GemmArgs.W / W2), S chosen from max|W| (tmp: one device uint32 of
// scratch); *scale_out = 2^-S.  Stream-ordered, no host synchronisation.
void split_f32_to_h16_planes(hipStream_t s, const float* src, h16* hi, h16* lo, int N, int K, float* scale_out, unsigned* tmp);

}  // namespace mb
