/* libmaskbit_hip.so -- C ABI of the MI355X-native MaskBit sampling engine.
 *
 * The reference (markweberdev/maskbit) has no FFI / plugin layer: its hot path sits behind a
 * Python call surface.  This header is therefore the *new* boundary beneath the Python classes
 * that mirror that surface (maskbit_amd.LFQBert / ConvVQModel / sample); every entry point cites
 * the reference code it replaces (paths relative to the reference root).  Plain pointers and
 * sizes only -- no torch types.  All pointers are DEVICE pointers unless noted; `stream` is a
 * hipStream_t passed as void*.  Calls are stream-ordered, never synchronise the device and never
 * touch the default stream.  Return value: 0 on success, negative on error (message through
 * mb_last_error(), thread-local).  A handle is bound to the device that was current at create
 * time and is not thread-safe (the reference is single-threaded Python on one device,
 * scripts/eval_maskbit.py:65).
 */
#ifndef MASKBIT_HIP_H
#define MASKBIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MB_ABI_VERSION 5

typedef struct mb_gen mb_gen; /* generator engine  (modeling/bert.py LFQBert)            */
typedef struct mb_dec mb_dec; /* tokenizer decoder (modeling/conv_vqgan.py ConvVQModel)  */
typedef void* mb_stream;      /* hipStream_t                                              */

/* LFQBert constructor arguments (modeling/bert.py:345-358). */
typedef struct {
  int bits;    /* K = log2(codebook_size)          */
  int splits;  /* m = codebook_splits              */
  int hidden;  /* hidden_dim (multiple of 64)      */
  int heads;   /* heads; hidden/heads in {32,64}   */
  int depth;   /* transformer layers               */
  int mlp;     /* mlp_dim (multiple of 64)         */
  int seq;     /* (img_size/input_stride)^2 = 256  */
  int nclass;  /* 1000; row nclass = "dropped"     */
  /* Weight precision of the trunk/head GEMMs (no counterpart in the reference, which is fp32):
   * 0 = one fp16 value per weight; 1 = "fp16x2": hi + lo fp16 halves of the power-of-two pre-scaled
   * weight, both multiplied on the MFMA and summed in the fp32 accumulator (weight rounding error
   * 2^-22 instead of 2^-11; twice the GEMM work).  See DESIGN.md, "Precision". */
  int weight_split;
  /* Generator variants of modeling/bert.py: prenorm = use_prenorm (LayerNorm before each sub-layer, raw residual,
   * norm_after_transformer; bert.py:49-59,106-123,498-499); embed_tables = the `Bert` class (bert.py:184-340): per-group
   * nn.Embedding(C+1, hidden) inputs summed, output head tied to those tables plus a per-position bias [seq, C].
   * Checkpoint keys then are tok_emb_list.{g}.weight and bias.{g} instead of input_proj.* / prediction_layer.*. */
  int prenorm;
  int embed_tables;
  /* Activation precision of the GEMMs that consume a LayerNorm output (QKV projection and FFN up-projection): 0 = one fp16
   * value per element; 1 = fp16 hi + lo pairs (x = hi + lo, |x - hi - lo| <= 2^-22 |x|): the LayerNorm kernels store both
   * halves and those two GEMMs sweep the weight twice (hi.W + lo.W in the same fp32 accumulator; twice their work).  With
   * classifier-free guidance this rounding point decides the token parity: see DESIGN.md, "Precision".  2 = additionally the
   * attention output and the FFN hidden are hi + lo pairs (written by the attention kernel and the FFN-up epilogue), so all four
   * trunk GEMMs of a layer do twice their work.  3 = as 2, but the lo halves are stored as e4m3(lo * 2^12) and multiplied with an e4m3
   * copy of the weights on v_mfma_scale_f32_16x16x128_f8f6f4 (whose E8M0 scales undo the powers of two): the correction pass costs half
   * a sweep; needs hidden and mlp to be multiples of 256.  4 = as 3, but the lo halves of the LayerNorm outputs are MX-fp4 (e2m1, one
   * power-of-two scale per row) against an e2m1 copy of the QKV / FFN-up weights (per-row scales): that correction pass costs a quarter
   * sweep (hidden 768 / 1024).  2, 3 and 4 meet the <= 1e-3 token mismatch against the fp32 reference.  Not combined with weight_split. */
  int act_split;
  /* Classifier-free guidance in differential form (mb_gen_forward_cfg / mb_sample; DESIGN.md "Precision"): 0 = off (the guided forward is
   * the plain forward over [cond | uncond]); 1 = the unconditional stream's GEMM operands are carried as fp16(x_u - x_c) next to fp16(x_c),
   * so the operand rounding of x_c is common to both streams and cancels in (c - u), the term the guidance scale multiplies -- at no
   * extra GEMM work (act_split then only concerns the plain forward; weight_split = 1 composes: the pair GEMMs sweep twice); 2 = additionally an
   * MX-fp4 correction pass for the fp16 rounding of the WEIGHTS of all four trunk GEMMs in the guided forward (e2m1 of the conditional operand
   * values with per-(row, 64 columns) scales against e2m1(W - fp16(W)) with per-row scales; a quarter sweep over the conditional half of every
   * tile): measured token mismatch of the fp16x2-weight mode (5.8e-4 / 7.1e-4 against 4.7e-4 / 6.6e-4 on the 12-bit / 14-bit runs) for +17 % time;
   * not combined with act_split = 4 or weight_split.
   * Pair forwards need seq = 256, hidden 768 / 1024, mlp % 256 == 0, post-norm; otherwise the engine falls back to the plain forward. */
  int cfg_pair;
} mb_gen_cfg;

/* ConvDecoder configuration (modeling/modules/autoencoder.py:358-397, configs/tokenizer yaml files). */
typedef struct {
  int token_size;      /* K bits per token = conv_in input channels */
  int hidden_channels; /* 128                                       */
  int num_resolutions; /* 5                                         */
  int num_res_blocks;  /* 2                                         */
  int num_channels;    /* 3                                         */
  int channel_mult[8]; /* [1,1,2,2,4]                               */
  int latent_size;     /* token grid side: 16 (=> 256x256 output)   */
  int build_encoder;   /* 1: also build ConvEncoder (autoencoder.py:230-286) for mb_enc_encode */
  int sample_with_conv;/* encoder downsampling by stride-2 conv (every shipped config); 0 (avg-pool) is not built */
  int enc_res_blocks;  /* num_res_blocks of the encoder (num_res_blocks above is the decoder's); 0 = same */
} mb_dec_cfg;

/* Per-step plan of modeling.modules.sample (modeling/modules/sampling.py:81-124), evaluated on the
 * host exactly as the reference does (float32 torch scalars) and handed over as HOST arrays. */
typedef struct {
  int num_steps;
  int use_guidance;            /* guidance_scale != 0 => 2B sequences per step (sampling.py:83-88) */
  const float* scale;          /* [num_steps] guidance_scale * a_i (sampling.py:91-98)            */
  const float* temperature;    /* [num_steps] softmax temperature (sampling.py:103-105)           */
  const int* mask_len;         /* [num_steps] floor(mask_ratio * n*m) (sampling.py:120-123)       */
  /* Step chunk of this call: step_end = 0 -> the whole run; otherwise steps [step_begin, step_end) -- the first chunk (step_begin 0) starts from the
   * all-masked state, later chunks continue from the state the engine kept, the last one (step_end = num_steps) combines and decodes.  The noise and
   * step_tokens pointers of a call hold the steps of ITS chunk (chunk-relative), the arrays above the whole run.  A chunk is accepted only as the
   * exact continuation of the run in progress on that handle (same B, num_steps, use_guidance; step_begin = the previous chunk's step_end): a
   * generator handle is NOT re-entrant while a chunked run is in progress -- two interleaved runs need two handles.
   * Steps whose scale[i] is exactly 0 run the conditional forward alone (c + 0 (c - u) == c: the unconditional forward cannot change the result). */
  int step_begin, step_end;
} mb_sample_plan;

int mb_abi_version(void);
const char* mb_last_error(void);

/* ---- generator: LFQBert.forward, modeling/bert.py:456-508 --------------------------------- */
int mb_gen_create(const mb_gen_cfg* cfg, int max_seqs, mb_gen** out);
void mb_gen_destroy(mb_gen* g);
/* One call per checkpoint entry (key names of SURVEY.md 8b / BaseModel.load_pretrained,
 * modeling/modules/base_model.py:87-141).  `data` is a device fp32 tensor in the checkpoint's
 * own layout; GEMM weights are repacked to fp16 here (plus the 8- / 4-bit lo-pass copies of the strict mode).  Unknown names return -2. */
int mb_gen_load(mb_gen* g, const char* name, const float* data, const int64_t* shape, int ndim, mb_stream stream);
/* cfg_pair == 2 only: the weight-correction pass of the guided forward runs in trunk layers >= `layer` (default 0 = every layer; depth = none).  The
 * second half of the trunk alone (layer = depth / 2) costs half as much and measured 6.3e-4 instead of 5.0e-4 / 8.4e-4 (all / no layers) on the
 * 12-bit 64-step run of the reference, but 1.26e-3 instead of 6.5e-4 on the 14-bit 256-step one (profiles/r03_parity.md). */
int mb_gen_set_wcorr_from(mb_gen* g, int layer);
/* tokens int64 [nb,seq,m] (value C = masked), labels int64 [nb], drop uint8 [nb] (1 => label
 * replaced by nclass, bert.py:482-484; may be NULL) -> logits fp32 [nb,seq,m,C]. */
int mb_gen_forward(mb_gen* g, const int64_t* tokens, const int64_t* labels, const uint8_t* drop,
                   float* logits, int nb, mb_stream stream);
/* The guided forward of sample() (sampling.py:83-88): tokens int64 [B,seq,m], labels int64 [B] -> logits fp32 [2B,seq,m,C], rows [0,B) the
 * conditional and [B,2B) the label-dropped forward of the same tokens.  With cfg.cfg_pair the two streams run in differential form
 * (see mb_gen_cfg.cfg_pair).  `scale` is reserved (ignored: the precision of the forward does not depend on the guidance scale); pass the
 * step's scale or any number. */
int mb_gen_forward_cfg(mb_gen* g, const int64_t* tokens, const int64_t* labels, float* logits, int B, float scale, mb_stream stream);
/* The same forward with `return_attn=True` (bert.py:461, 505-508; nn.MultiheadAttention need_weights with head averaging,
 * bert.py:119,137): additionally attn fp32 [depth, nb, seq+1, seq+1], layer l's softmax weights averaged over the heads
 * (class token = last row / column).  Visualisation path, not used by sample(). */
int mb_gen_forward_attn(mb_gen* g, const int64_t* tokens, const int64_t* labels, const uint8_t* drop,
                        float* logits, float* attn, int nb, mb_stream stream);

/* ---- one sampling step after the forward: sampling.py:90-131 ------------------------------ *
 * logits_u NULL => no guidance.  `scale` = guidance_scale * a_i (sampling.py:91-98), `temperature`
 * the softmax temperature of this step.  exp_noise [B*n*m, C] is the Exp(1) draw of
 * torch.multinomial; conf_noise [B,n,m] is gumbel*randomize_temperature*(1-progress).
 * k_mask_len = floor(ratio * n*m) (masking.py:41-65, sampling.py:120-123); the clamp to
 * [1, num_masked(sample 0) - 1] happens on the device.  tokens_in / tokens_out [B,n,m] must not
 * alias; pred_out (may be NULL) receives the step's predicted tokens (l_full_tokens entry). */
int mb_sample_step(const float* logits_c, const float* logits_u, float scale, float temperature,
                   const float* exp_noise, const float* conf_noise, int k_mask_len,
                   const int64_t* tokens_in, int64_t* tokens_out, int64_t* pred_out,
                   int B, int n, int m, int C, mb_stream stream);

/* ---- decoder: ConvVQModel.decode_tokens, modeling/conv_vqgan.py:98-112 --------------------- */
int mb_dec_create(const mb_dec_cfg* cfg, int max_batch, mb_dec** out);
void mb_dec_destroy(mb_dec* d);
int mb_dec_load(mb_dec* d, const char* name, const float* data, const int64_t* shape, int ndim, mb_stream stream);
/* tokens int64 [B, n] (K-bit codes) -> img_nchw fp32 [B,3,H,W] unclamped (may be NULL) and/or
 * img_nhwc_u8 uint8 [B,H,W,3] = trunc(clamp(x,0,1)*255) (scripts/eval_maskbit.py:134-135; may be NULL). */
int mb_dec_decode(mb_dec* d, const int64_t* tokens, float* img_nchw, uint8_t* img_nhwc_u8, int B, mb_stream stream);
/* The decoder / encoder keep activations in fp16 (saturating stores).  Number of 4-channel output groups that were clamped at +-65504 since the last
 * reset, over all conv layers (0 for every configuration tested; a trained checkpoint that needs more range shows up here instead of being
 * clipped silently).  Synchronises `stream`. */
int mb_dec_saturation_count(mb_dec* d, unsigned* count, int reset, mb_stream stream);
/* ---- encoder half: ConvVQModel.encode, modeling/conv_vqgan.py:70-83 (ConvEncoder autoencoder.py:264-286 +
 * LookupFreeQuantizer sign/pack lookup_free.py:57-62,113-127).  img fp32 [B,C,H,W] -> indices int64 [B, h*w];
 * zq (+-1 latent, fp32 [B,K,h,w]) and zraw (pre-sign encoder output) may be NULL.  Needs build_encoder = 1. */
int mb_enc_encode(mb_dec* d, const float* img_nchw, int64_t* indices, float* zq, float* zraw, int B, mb_stream stream);

/* ---- whole loop: modeling.modules.sample, sampling.py:55-136 ------------------------------- *
 * Runs num_steps x (forward [+CFG], step) then combine (factorization.py:7-24) + decode.
 * exp_noise [steps, B*n*m, C] and conf_noise [steps, B, n, m] (= gumbel * randomize_temperature *
 * (1-progress)) are drawn by the caller with the reference's RNG protocol.  step_tokens int64
 * [steps,B,n,m] may be NULL.  tokens_out int64 [B,n] receives the combined K-bit codes (may be
 * NULL).  d may be NULL (then both image pointers must be NULL). */
int mb_sample(mb_gen* g, mb_dec* d, const mb_sample_plan* plan, const int64_t* labels, int B,
              const float* exp_noise, const float* conf_noise, int64_t* step_tokens, int64_t* tokens_out,
              float* img_nchw, uint8_t* img_nhwc_u8, mb_stream stream);

/* ---- introspection used by bench.py (not part of the reference surface) -------------------- */
/* Name + accumulated device time (HIP events on the launch stream) of the engine's kernels. */
/* One GEMM of the trunk family, out[M,N] = A[M,K] . W[N,K]^T + bias with epilogue `epi` (0 fp16 out,
 * 1 gelu->fp16, 2 +residual->fp32, 3 gelu->fp32, 4 logits fp32 with every `period`-th row dropped);
 * A, W, out_h16 are fp16 device buffers.  variant: 0 auto, -1 the 128x128 kernel, 6 / 8 the half-tile kernel with
 * 192 / 256-row tiles, 257 its sequence-aligned tiles (M % 257 == 0). */
/* Split-weight diagnostics: repack W[N,K] fp32 -> dst[N,2K] fp16 (hi | lo) + *scale_out, and the GEMM over such a weight
 * (mb_gemm_ex with K = 2*ka, A is [M,ka]); `tmp` is 4 bytes of device scratch. */
int mb_split_weights(const float* W, int N, int K, void* dst_h16, float* scale_out, void* tmp, mb_stream stream);
int mb_gemm_ex(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16,
               int M, int N, int K, int ka /*0, or K/2: split weights*/, const float* scale /*split weights*/,
               const float* ln_stats /*or NULL: residual = LayerNorm(residual rows) from {mean,rstd}[M]*/, const float* ln_g,
               const float* ln_b, int period, int variant, mb_stream stream);
/* LayerNorm over rows (modeling/bert.py:69-70,137-139): any of x_f32 / x_h16 / stats ({mean, rstd} per row) may be NULL. */
int mb_layernorm(const float* y, const float* gamma, const float* beta, float eps, float* x_f32, void* x_h16, void* x_lo, float* stats,
                 int M, int d, mb_stream stream);
/* A GEMM with split activations: out = (A_hi + A_lo) . W^T + bias through the engine's kernels (A_hi, A_lo fp16 [M, kw], W fp16
 * [N, kw]; x_lo as written by mb_layernorm).  Diagnostic / test entry for mb_gen_cfg.act_split. */
/* The same with an e4m3 lo pass: A8 = e4m3(lo * 2^12) and W8 = e4m3(W * 2^(*w8_exp)), both with the row stride of their fp16 siblings
 * (2*kw bytes, first kw used); kw % 128 == 0.  Diagnostic / test entry for mb_gen_cfg.act_split == 3. */
int mb_gemm_f8lo(int epi, const void* A_hi, const void* A8, const void* W, const void* W8, const int* w8_exp, const float* bias,
                 const float* residual, float* out_f32, void* out_h16, int M, int N, int kw, int variant, mb_stream stream);
/* The same with an MX-fp4 lo pass (mb_gen_cfg.act_split == 4): A4 = e2m1(lo * 2^s_m) two values per byte with one E8M0 scale byte per row
 * (a_scale[m], as mb_layernorm_f4 writes them), W4 / w_scale from mb_w4_from_f32 (per-row scales in the kernel's lane order); both 4-bit
 * operands with the row stride of their fp16 siblings (2*kw bytes, first kw/2 used); kw % 256 == 0, N % 64 == 0. */
int mb_w4_from_f32(const float* W, int N, int K, void* dst4, void* scale_out, mb_stream stream);
/* the same layout for the weight's fp16 rounding error W - fp16(W) (operand of the weight-correction pass, mb_gen_cfg.cfg_pair == 2) */
int mb_w4lo_from_f32(const float* W, int N, int K, void* dst4, void* scale_out, mb_stream stream);
int mb_layernorm_f4(const float* y, const float* gamma, const float* beta, float eps, float* x_f32, void* x_h16, void* x4, void* x4_scale,
                    int M, int d, mb_stream stream);
int mb_gemm_f4lo(int epi, const void* A_hi, const void* A4, const void* a_scale, const void* W, const void* W4, const void* w_scale,
                 const float* bias, const float* residual, float* out_f32, void* out_h16, int M, int N, int kw, int variant, mb_stream stream);
/* A "CFG pair" GEMM (mb_gen_cfg.cfg_pair): rows [0, pair_rows) of A / out are conditional, [pair_rows, 2 pair_rows) their unconditional twins whose
 * A rows hold the difference operand; out_c = f(A_c.W), out_u = f(A_c.W + A_delta.W) (GELU epilogue: the u rows receive gelu(u) - gelu(c)).
 * pair_rows % 257 == 0.  A4 / a_scale / W4 / w_scale (all four or none): an MX-fp4 correction pass over kw/256 extra K-tiles. */
int mb_gemm_pair(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16,
                 int pair_rows, int N, int kw, const void* A4, const void* a_scale, const void* W4, const void* w_scale, mb_stream stream);
/* Attention of a CFG pair batch (the generator's guided forward, bert.py:84,137 on both streams): qkv [2 pairs N, 3d] fp16 packed in_proj rows, the
 * conditional sequences first, their unconditional twins `pairs` sequences later.  out rows of conditional sequences = softmax(QK^T/sqrt(dh))V in
 * fp16; rows of unconditional sequences = fp16(o_u - o_c), the difference operand of the out-proj pair GEMM.  aux [pairs N, d] fp32: scratch of the
 * two-launch form (MASKBIT_AMD_ATT_PAIR=2); the default form keeps the conditional rows in registers and leaves it untouched. */
int mb_attention_pair(const void* qkv, void* out_h16, float* aux, int pairs, int N, int d, int heads, mb_stream stream);
int mb_gemm_act_split(int epi, const void* A_hi, const void* A_lo, const void* W, const float* bias, const float* residual,
                      float* out_f32, void* out_h16, int M, int N, int kw, int variant, mb_stream stream);
int mb_gemm(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32,
            void* out_h16, int M, int N, int K, int period, int variant, mb_stream stream);
/* Persistent kernels launch one workgroup per CU.  On a stream created with a CU mask (hipExtStreamCreateWithCUMask) fewer CUs serve the launch:
 * n = the CUs the following launches should size their grids for, 0 = the device's count (default).  Process-wide, not thread-safe. */
int mb_set_cu_count(int n);
int mb_prof_enable(int on); /* 0 off; n >= 1: HIP-event timing of every kernel of every n-th generator forward (forwards n/2, n/2 + n, ..) and of all other calls */
int mb_prof_read(char* buf, int buflen); /* host buffer; writes "name calls total_ms\n" lines */

#ifdef __cplusplus
}
#endif
#endif /* MASKBIT_HIP_H */
