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

#define MB_ABI_VERSION 8
enum { MB_PREC_FP16 = 0, MB_PREC_DIFF = 1, MB_PREC_WCORR = 2, MB_PREC_ALO = 3, MB_PREC_ALO_ALL = 4 };

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
  /* Generator variants of modeling/bert.py: prenorm = use_prenorm (LayerNorm before each sub-layer, raw residual,
   * norm_after_transformer; bert.py:49-59,106-123,498-499); embed_tables = the `Bert` class (bert.py:184-340): per-group
   * nn.Embedding(C+1, hidden) inputs summed, output head tied to those tables plus a per-position bias [seq, C].
   * Checkpoint keys then are tok_emb_list.{g}.weight and bias.{g} instead of input_proj.* / prediction_layer.*. */
  int prenorm;
  int embed_tables;
  /* Precision mode of the engine -- ONE knob (no counterpart in the reference, which is fp32; DESIGN.md "Precision").  fp16 operands, fp32 accumulation
   * everywhere; the two head GEMMs always multiply hi + lo inputs by hi + lo weights.
   *   MB_PREC_FP16  0  single fp16 operands, guided forward = plain forward over [cond | uncond] (1.4e-3 token mismatch on configs[2]: a baseline).
   *   MB_PREC_DIFF  1  classifier-free guidance in DIFFERENTIAL form: the unconditional stream's GEMM operands are fp16(x_u - x_c), so the rounding of
   *                    x_c cancels in (c - u); plain forwards carry the LayerNorm outputs as fp16 hi + lo pairs (~1.0e-3: AT the bound).
   *   MB_PREC_WCORR 2  + MX-fp4 mini-tile correction of the fp16 rounding of all four trunk WEIGHTS (gemm_ht.hip XP = 6), guided and plain (5.5e-4).
   *   MB_PREC_ALO   3  + the same kind of pass for the fp16 rounding of the ACTIVATIONS of the guided forward's conditional stream (e2m1 of their lo halves
   *                    against e2m1 of the fp16 weight): attention outputs in out-proj and LayerNorm outputs in FFN-up, every layer (what the 7-bit-per-group
   *                    codebooks need: 3.8e-4 over four 14-bit / 256-step runs, every run <= 5.5e-4; without it one run is at 1.0e-3; rounds 4-5: FFN-up only).
   *   MB_PREC_ALO_ALL 4 + the FFN hiddens in FFN-down (the QKV set buys nothing in any measured configuration), and the zero-scale steps of a guided run
   *                    through the guided forward as well: what heavy-tailed ("trained-like") weights with massive-activation channels need (round 6).
   * Modes 1-4 need seq in {256, 1024}, hidden in {768, 1024}, mlp % 256 == 0 (2-4 also hidden / heads = 64); other shapes run mode 0 with hi + lo
   * LayerNorm outputs.  The host's default is 2, 3 from 7 bits per group on, 4 for heavy-tailed checkpoints (LFQBert.resolved_precision). */
  int precision;
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
  int sample_with_conv;/* encoder downsampling by stride-2 conv (every shipped config); 0 = 2x2 average pooling */
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
   * Steps whose scale[i] is exactly 0 run the conditional forward alone (c + 0 (c - u) == c: the unconditional forward cannot change the result; at
   * precision 4 they run the guided forward, whose conditional half is the more precise one). */
  int step_begin, step_end;
} mb_sample_plan;

int mb_abi_version(void);
const char* mb_last_error(void);

/* ---- generator: LFQBert.forward, modeling/bert.py:456-508 --------------------------------- */
int mb_gen_create(const mb_gen_cfg* cfg, int max_seqs, mb_gen** out);
void mb_gen_destroy(mb_gen* g);
/* One call per checkpoint entry (key names of SURVEY.md 8b / BaseModel.load_pretrained,
 * modeling/modules/base_model.py:87-141).  `data` is a device fp32 tensor in the checkpoint's
 * own layout; GEMM weights are repacked to fp16 here (plus the e2m1 operands of the correction passes; the two head weights as fp16
 * hi + lo planes).  Unknown names return -2. */
int mb_gen_load(mb_gen* g, const char* name, const float* data, const int64_t* shape, int ndim, mb_stream stream);
/* tokens int64 [nb,seq,m] (value C = masked), labels int64 [nb], drop uint8 [nb] (1 => label
 * replaced by nclass, bert.py:482-484; may be NULL) -> logits fp32 [nb,seq,m,C]. */
int mb_gen_forward(mb_gen* g, const int64_t* tokens, const int64_t* labels, const uint8_t* drop,
                   float* logits, int nb, mb_stream stream);
/* The guided forward of sample() (sampling.py:83-88): tokens int64 [B,seq,m], labels int64 [B] -> logits fp32 [2B,seq,m,C], rows [0,B) the
 * conditional and [B,2B) the label-dropped forward of the same tokens.  With cfg.precision >= 1 the two streams run in differential form
 * (see mb_gen_cfg.precision); the precision of the forward does not depend on the guidance scale the caller combines the two halves with. */
int mb_gen_forward_cfg(mb_gen* g, const int64_t* tokens, const int64_t* labels, float* logits, int B, mb_stream stream);
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

/* ---- measurement hooks used by bench.py (not part of the reference surface) ----------------- */
int mb_prof_enable(int on); /* 0 off; n >= 1: HIP-event timing of every kernel of every n-th generator forward (forwards n/2, n/2 + n, ..) and of all other calls */
int mb_prof_read(char* buf, int buflen); /* host buffer; writes "name calls total_ms\n" lines */
/* fp16 activation stores of the trunk saturate at +-65504 instead of producing infinities.  Number of QKV / FFN-up / attention / LayerNorm output
 * groups of this handle's forwards that were clamped since the last reset (0 for every configuration tested, synthetic heavy-tailed weights
 * included; a checkpoint that needs more range shows up here instead of being clipped silently).  Synchronises `stream`. */
int mb_gen_saturation_count(mb_gen* g, unsigned* count, int reset, mb_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MASKBIT_HIP_H */
