/* libmaskbit_hip.so -- DIAGNOSTIC entry points: single kernels of the engine on caller buffers, for the unit tests (tests/test_hip_gemm.py,
 * test_hip_pair.py) and the tools under tools/.  Nothing here is part of the reference's surface and no host binding of the product needs it:
 * include/maskbit_hip.h is the ABI.  Same conventions (device pointers, stream-ordered, 0 / negative return, mb_last_error()).
 */
#ifndef MASKBIT_HIP_DIAG_H
#define MASKBIT_HIP_DIAG_H

#include "maskbit_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One GEMM of the trunk family, out[M,N] = A[M,K] . W[N,K]^T + bias with epilogue `epi` (0 fp16 out, 1 gelu->fp16, 2 +residual->fp32, 3 gelu->fp32,
 * 4 logits fp32 with every `period`-th row dropped); A, W, out_h16 are fp16 device buffers.  variant: 0 auto, -1 the 128x128 kernel, 6 / 8 the
 * half-tile kernel with 192 / 256-row tiles, 257 its sequence-aligned tiles (M % 257 == 0). */
int mb_gemm(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32,
            void* out_h16, int M, int N, int K, int period, int variant, mb_stream stream);
/* mb_gemm with the LayerNorm-residual epilogue: ln_stats != NULL: the residual that is added is LayerNorm(residual rows) re-derived from
 * {mean, rstd}[M] (epi 2 only). */
int mb_gemm_ex(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16,
               int M, int N, int K, const float* ln_stats, const float* ln_g, const float* ln_b, int period, int variant, mb_stream stream);
/* LayerNorm over rows (modeling/bert.py:69-70,137-139): any of x_f32 / x_h16 / x_lo (fp16 lo halves) / stats ({mean, rstd} per row) may be NULL. */
int mb_layernorm(const float* y, const float* gamma, const float* beta, float eps, float* x_f32, void* x_h16, void* x_lo, float* stats,
                 int M, int d, mb_stream stream);
/* Split-activation GEMM (the plain forward's LayerNorm outputs as fp16 hi + lo pairs): out = (A_hi + A_lo) . W^T + bias, both [M, kw]. */
int mb_gemm_act_split(int epi, const void* A_hi, const void* A_lo, const void* W, const float* bias, const float* residual,
                      float* out_f32, void* out_h16, int M, int N, int kw, int variant, mb_stream stream);
/* "CFG pair" GEMM (mb_gen_cfg.precision >= 1): rows [0, pair_rows) of A / out are conditional, [pair_rows, 2 pair_rows) their unconditional twins whose
 * A rows hold the difference operand; out_c = f(A_c.W), out_u = f(A_c.W + A_delta.W) (GELU epilogue: the u rows receive gelu(u) - gelu(c)).
 * mb_gemm_mini: a sequence-aligned GEMM (rows % 257 == 0; pair != 0: a pair GEMM over `rows` conditional rows) with nlo MX-fp4 mini-tile passes:
 * lo = nlo x {A4, a_scale, W4, w_scale} device pointers (operand layouts: mb_kernels.h GemmArgs.lo; w_scale = N K / 128 bytes as mb_w4_from_f32 writes them; the token scales in lane order,
 * index (((blk * nseq + seq) * 4 + (r >> 6)) * 64 + (r & 15) * 4 + ((r >> 4) & 3) for token r of sequence seq).  out4 / out4_scale (GELU epilogue,
 * optional): e2m1 of the (conditional) outputs + their lane-ordered scales. */
int mb_gemm_mini(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16, void* out4,
                 void* out4_scale, int rows, int pair, int N, int K, int nlo, const void* const* lo, mb_stream stream);
/* ... the pair form for sequences of seq_rows rows incl. the class token (0 = 257; 1 025 = the 512 x 512 models: eight 128-token pair tiles per sequence
 * pair, (seq_rows - 1) / 64 token groups in the lane-ordered scale arrays).  out4l / out4l_scale (GELU epilogue, optional, with out4; precision 4): the same e2m1
 * copy for the fp16 LO HALVES v - fp16(v) of the (conditional) outputs. */
int mb_gemm_mini_seq(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16, void* out4,
                     void* out4_scale, void* out4l, void* out4l_scale, int rows, int pair, int seq_rows, int N, int K, int nlo, const void* const* lo, mb_stream stream);
/* e2m1 operands of the mini-tile passes: of the fp16 weight values (scales minimising the quantisation error) and of the weight's fp16 rounding error
 * W - fp16(W); dst4 = N K / 2 bytes mini-tile-packed (mb_kernels.h w4_packed_offset), scale_out = N K / 128 bytes -- one E8M0 byte per (weight row, 128
 * K-elements) -- in the kernel's lane order (w4_scale_index: byte ((((n >> 6) * (K / 128) + j) * 16 + (n & 15)) * 4 + ((n >> 4) & 3)); N % 64 == 0, K % 128 == 0. */
int mb_w4_from_f32(const float* W, int N, int K, void* dst4, void* scale_out, mb_stream stream);
int mb_w4lo_from_f32(const float* W, int N, int K, void* dst4, void* scale_out, mb_stream stream);
/* LayerNorm that also writes the e2m1 copies of its output rows: values (x4 / x4_scale) and / or fp16 lo halves (xl4 / xl4_scale); M % 257 == 0,
 * d = 768 / 1024; class-token rows (row % 257 == 256) are skipped. */
/* ... the plain forward's default for QKV / FFN-up (precision >= 2): plain sequence tiles over hi + lo activation halves
 * (A_hi, A_lo [rows, kw]; the fp16 sweep runs twice over W [N, kw]) AND one mini-tile operand set over the kw columns; epi 0 / 1. */
int mb_gemm_mini_split(int epi, const void* A_hi, const void* A_lo, const void* W, const float* bias, void* out_h16, void* out4, void* out4_scale,
                       int rows, int N, int kw, const void* const* lo, mb_stream stream);
int mb_layernorm_f4(const float* y, const float* gamma, const float* beta, float eps, float* x_f32, void* x_h16, void* x4, void* x4_scale, void* xl4,
                    void* xl4_scale, int M, int d, mb_stream stream);
/* Attention of a CFG pair batch (the generator's guided forward, bert.py:84,137 on both streams): qkv [2 pairs N, 3d] fp16 packed in_proj rows, the
 * conditional sequences first, their unconditional twins `pairs` sequences later.  out rows of conditional sequences = softmax(QK^T/sqrt(dh))V in
 * fp16; rows of unconditional sequences = fp16(o_u - o_c), the difference operand of the out-proj pair GEMM (the conditional output tiles stay in
 * registers in between).  N <= 288: one head's K / V in LDS; longer sequences: the streaming kernel. */
int mb_attention_pair(const void* qkv, void* out_h16, int pairs, int N, int d, int heads, mb_stream stream);
/* The same launch as the engine issues it at precision >= 2: + out4 [2 pairs N, 2 d] (first d / 2 bytes of a row used) = e2m1 of the CONDITIONAL outputs, two
 * values per byte, and out4_scale = one E8M0 byte per (row, head) in the lane order of the mini-tile passes (mb_kernels.h fp4_scale_index with (N - 1) / 64
 * token groups per sequence, `pairs` sequences): the token operand of the out-projection's weight-correction pass.  Head width 64, N = 257 or
 * (N - 1) % 64 == 0 beyond 288. */
/* out4l / out4l_scale (optional, both or neither; precision 4): the same for the fp16 lo halves o_c - fp16(o_c) of the conditional outputs. */
int mb_attention_pair_f4(const void* qkv, void* out_h16, void* out4, void* out4_scale, void* out4l, void* out4l_scale, int pairs, int N, int d, int heads, mb_stream stream);
/* Study knob of the weight-correction passes (precision >= 2): they run in trunk layers >= from_layer (default 0 = every layer; depth = none; guided
 * forward) and on the GEMMs of gemm_mask (1 QKV, 2 out-proj, 4 FFN-up, 8 FFN-down; default 15; both forwards).  No subset keeps the default's
 * parity margin (profiles/r04_gemm_minitiles.md section 4): the product never calls this. */
int mb_gen_set_wcorr(mb_gen* g, int from_layer, int gemm_mask);
/* The same kind of study knob for the ACTIVATION-LO sets of precision >= 3: trunk layers >= from_layer, GEMMs of gemm_mask (same bits).  A handle is created
 * with the operands of its precision's own coverage (3: FFN-up, layers >= depth / 2; 4: every GEMM of every layer); the knob can only narrow that. */
int mb_gen_set_alo(mb_gen* g, int from_layer, int gemm_mask);
/* Persistent kernels launch one workgroup per CU.  On a stream created with a CU mask (hipExtStreamCreateWithCUMask) fewer CUs serve the launch:
 * n = the CUs the following launches should size their grids for, 0 = the device's count (default).  Process-wide, not thread-safe. */
int mb_set_cu_count(int n);

#ifdef __cplusplus
}
#endif
#endif /* MASKBIT_HIP_DIAG_H */
