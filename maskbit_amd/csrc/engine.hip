// C ABI of libmaskbit_hip.so (include/maskbit_hip.h): handle objects, checkpoint ingest with h16
// repack, the generator forward schedule, the fused sampling step and the whole sampling loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/maskbit_hip_diag.h"
#include "mb_decoder.h"
#include "mb_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(-10, "%s failed: %s", #expr, hipGetErrorString(e_));     \
  } while (0)

// ---- optional per-kernel device timing with HIP events on the launch stream ------------------
struct Prof {
  bool on = false;
  // Generator forwards are sampled: the kernels of every `stride`-th forward are timed -- counted separately for GUIDED forwards (mb_gen_forward_cfg
  // and the guided steps of mb_sample: kernel names as they are) and PLAIN ones (mb_gen_forward, the unguided / zero-scale steps: names + ".plain"),
  // so that a plain forward never lands in a guided kernel's average whatever the step plan and the chunking (round-3 advice).
  int stride = 1, tick[2] = {0, 0};
  bool fwd_live = true, fwd_plain = false;
  void begin_forward(bool plain) { fwd_plain = plain; fwd_live = (tick[plain]++ % stride) == stride / 2; }   // (the middle of every stride)
  struct Rec { hipEvent_t a, b; int kind; };
  std::vector<Rec> recs;
  std::vector<std::string> names;
  std::map<std::string, int> index;
  std::map<int, std::pair<long, double>> acc;   // kind -> (calls, ms)
  int kind(const char* n0, bool in_forward) {
    const std::string n = (in_forward && fwd_plain) ? std::string(n0) + ".plain" : std::string(n0);
    auto it = index.find(n);
    if (it != index.end()) return it->second;
    names.push_back(n);
    return index[n] = (int)names.size() - 1;
  }
  void drain() {
    for (auto& r : recs) {
      float ms = 0.f;
      if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
        acc[r.kind].first += 1; acc[r.kind].second += ms;
      }
      (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    recs.clear();
  }
} g_prof;

struct ProfScope {
  hipStream_t s; bool live; hipEvent_t a, b; int kind;
  ProfScope(const char* name, hipStream_t st, bool in_forward = false) : s(st), live(g_prof.on && (!in_forward || g_prof.fwd_live)) {
    if (!live) return;
    kind = g_prof.kind(name, in_forward);
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
  }
  ~ProfScope() {
    if (!live) return;
    (void)hipEventRecord(b, s);
    g_prof.recs.push_back({a, b, kind});
  }
};

template <typename T>
int dev_alloc(T** p, size_t n) {
  HIP_TRY(hipMalloc((void**)p, n * sizeof(T)));
  return 0;
}

}  // namespace

// ================================================================================================
// generator
// ================================================================================================
struct mb_gen {
  mb_gen_cfg c{};
  int max_seqs = 0, chunk_seqs = 0, N = 0, C = 0, gbits = 0, device = 0;   // chunk_seqs: sequences per forward pass (workspace size)
  struct Layer {
    h16 *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
  };
  std::vector<Layer> layers;
  float *w_in = nullptr, *b_in = nullptr, *class_emb = nullptr, *pos = nullptr, *ln0g = nullptr, *ln0b = nullptr;
  h16 *wl = nullptr, *wp = nullptr;
  float *bl = nullptr, *lnhg = nullptr, *lnhb = nullptr, *bp = nullptr;
  float *lnag = nullptr, *lnab = nullptr;              // norm_after_transformer (pre-norm variant)
  float *tables = nullptr, *bias_pos = nullptr;       // Bert: embedding tables [m][C+1][d]; output bias [seq][m*C]
  unsigned* split_tmp = nullptr;                        // scratch of the head weights' hi / lo split
  // workspace
  float *y_f32 = nullptr, *ln_stats = nullptr;       // fp32 residual stream (pre-LayerNorm rows) and {mean, rstd} per row
  h16 *x_h16 = nullptr, *x_lo = nullptr, *qkv = nullptr, *att = nullptr, *h = nullptr;   // x_lo: lo halves of x_h16 (head GEMMs; precision >= 1: QKV / FFN-up of plain forwards)
  // precision >= 1: differential CFG forward (pair_ok = the shape allows it).  precision >= 2 (mini_ok = the shape allows it): every trunk GEMM carries the
  // MX-fp4 weight-correction mini-tiles (gemm_ht.hip, XP = 6) -- in the guided forward on the conditional rows, in the plain forward (257-token sequences)
  // on every row: x4 / att4 / h4 hold e2m1 of the LayerNorm outputs, attention outputs and FFN hiddens (values; x4s / att4s / h4s their lane-ordered
  // block scales), w4lo / w4los e2m1 of the weights' fp16 rounding errors.  precision 3 additionally corrects the fp16 rounding of the LayerNorm OUTPUTS
  // in the guided forward's FFN-up GEMM: xl4 / xl4s = e2m1 of their lo halves, w4 / w4s = e2m1 of the (fp16) weight net.0.
  bool pair_ok = false, mini_ok = false;
  uint8_t *x4 = nullptr, *x4s = nullptr, *xl4 = nullptr, *xl4s = nullptr;
  std::vector<uint8_t*> w4, w4s;                                                         // [4 * layer + {qkv, o, 1, 2}]: precision 3: net.0 of the late layers only; precision 4: all
  float* logits_tmp = nullptr;                          // guided forwards over more pairs than one pass holds
  // The two head GEMMs run hi + lo inputs against hi + lo WEIGHTS in every mode (GemmArgs.W2: three sweeps): their rounding reaches the logits
  // un-averaged -- fp16 head weights alone were a quarter of the sampled-logit error variance left after the trunk's weight correction
  // (tests/diag/error_budget.py) -- and the two GEMMs are 0.4 % of a forward.  wl / wp = fp16(w 2^S), wl_lo / wp_lo = the remainders, head_scale = 2^-S.
  // (Bert's tied head, embed_tables: wp holds the tables' first C rows per group as single fp16, no lo plane.)
  h16 *wl_lo = nullptr, *wp_lo = nullptr;
  float* head_scale = nullptr;
  unsigned* sat = nullptr;                              // lanes of the QKV / FFN-up epilogues that clamped a fp16 store (mb_gen_saturation_count)
  std::vector<uint8_t*> w4lo, w4los;                                                     // [4 * layer + {qkv, o, 1, 2}]
  uint8_t *att4 = nullptr, *att4s = nullptr, *h4 = nullptr, *h4s = nullptr;              // e2m1 of the conditional attention outputs / FFN hiddens + block scales
  uint8_t *attl4 = nullptr, *attl4s = nullptr, *hl4 = nullptr, *hl4s = nullptr;          // precision 4: e2m1 of their fp16 LO HALVES (activation-lo sets of out-proj / FFN-down)
  // loop state for mb_sample
  // the run mb_sample is in the middle of (step chunks): samples, total steps, guidance flag, the step the next chunk must begin with (-1: no run)
  int loop_B = 0, loop_steps = 0, loop_guided = 0, loop_next = -1;
  int wcorr_from = 0;                                   // precision >= 2: first trunk layer that carries the correction passes (mb_gen_set_wcorr)
  int wcorr_mask = 15;                                  // ... and which GEMMs of a layer: 1 QKV, 2 out-proj, 4 FFN-up, 8 FFN-down
  // precision >= 3: which GEMMs (same bits) of which layers (>= alo_from) carry the activation-lo set.  Coverage measured on the reference's own runs in round 6
  // (profiles/r06_coverage.md: four 14-bit / 256-step runs, four 12-bit runs, three trained-like runs; mismatches / guided-forward time of 64 pairs):
  //   none 595 / 263 / 271 at 29.9 ms;  FFN-up of layers >= depth / 2 (round 5's precision 3) 506 / 222 / 276 at 30.5;  out-proj + FFN-up 384 / 186 / 219 at 31.6;
  //   out-proj + FFN-up + FFN-down 241 / 115 / 203 at 33.4;  all four 228 / 116 / 200 at 34.3 -- the QKV set buys nothing (as round 5 found for precision 3).
  // precision 3 = out-proj + FFN-up of every layer; precision 4 = + FFN-down.  mb_gen_set_alo (study knob) selects within what the handle was created with
  // (precision 4 builds the QKV operands too, for such studies).
  int alo_mask = 0, alo_from = 0, alo_mask_built = 0, alo_from_built = 0;
  const int64_t* cfg_labels_ready = nullptr;            // gen_forward_cfg: lab_cfg / drop_cfg already hold [labels | labels] / [0 | 1] for this many pairs
  int cfg_ready_B = 0;
  int64_t *tok_a = nullptr, *tok_b = nullptr, *tok_cfg = nullptr, *lab_cfg = nullptr, *pred = nullptr, *codes = nullptr;
  uint8_t* drop_cfg = nullptr;
  float* logits = nullptr;
  std::vector<void*> owned;
  int loaded = 0;
};

namespace {

template <typename T>
int galloc(mb_gen* g, T** p, size_t n) {
  int rc = dev_alloc(p, n);
  if (rc == 0) g->owned.push_back((void*)*p);
  return rc;
}

// The head (bert.py:411-417, 500-503): last_layer.0 + GELU, LayerNorm, prediction layer, on the hi + lo rows the trunk's last LayerNorm left in
// x_h16 / x_lo -- hi + lo inputs against hi + lo weights in every mode (mb_gen::wl_lo)
int head_gemms(mb_gen* g, float* logits, int M, hipStream_t s) {
  using namespace mb;
  const mb_gen_cfg& c = g->c;
  const int d = c.hidden;
  int rc = 0;
  { ProfScope p("gemm_head", s, true);
    GemmArgs ga{g->x_h16, g->wl, g->bl, nullptr, g->y_f32, nullptr, M, d, 3 * d, 0, g->head_scale};
    ga.A2 = g->x_lo; ga.kw = d; ga.W2 = g->wl_lo;
    rc |= gemm_tn(s, EPI_GELU_F32, ga); }
  { ProfScope p("layernorm", s, true);
    layernorm_rows(s, g->y_f32, g->lnhg, g->lnhb, 1e-12f, nullptr, g->x_h16, nullptr, M, d, g->x_lo); }
  { ProfScope p("gemm_head", s, true);
    GemmArgs ga{g->x_h16, g->wp, c.embed_tables ? g->bias_pos : g->bp, nullptr, logits, nullptr, M, c.splits * g->C, (g->wp_lo ? 3 : 2) * d, g->N,
                g->wp_lo ? g->head_scale + 1 : nullptr};
    ga.A2 = g->x_lo; ga.kw = d; ga.W2 = g->wp_lo;
    ga.bias_per_pos = c.embed_tables;
    rc |= gemm_tn(s, EPI_LOGITS_F32, ga); }
  return rc;
}

int gen_forward_impl(mb_gen* g, const int64_t* tokens, const int64_t* labels, const uint8_t* drop, float* logits,
                     int nb, hipStream_t s, float* attn = nullptr) {
  using namespace mb;
  const mb_gen_cfg& c = g->c;
  const int d = c.hidden, f = c.mlp, N = g->N, M = nb * N;
  // return_attn: layer l's head-averaged attention weights go to attn[l][nb][N][N], computed from the same qkv rows
  auto attn_maps = [&](int l) {
    return attn ? attention_probs(s, g->qkv, attn + (size_t)l * nb * N * N, nb, N, d, c.heads) : 0;
  };
  int attn_rc = 0, gemm_rc = 0;
  // The plain forward (mb_gen_forward, sampling without guidance, the zero-scale steps of a guided run), by precision:
  //   >= 1: the LayerNorm output enters FFN-UP as an fp16 hi + lo pair (x_h16 + x_lo: that GEMM sweeps its weight twice, K = 2d).  Rounds 3-4 did the
  //         same in QKV; over configs[1]'s three reference runs + the trained-like one (348 160 positions) the QKV sweep buys nothing -- 153 mismatches
  //         with both, 158 with FFN-up alone, 192 with QKV alone, 195 with neither (profiles/raw/r05/xlo_mask.log) -- and costs 19-69 us per layer;
  //         by layer range (FFN-up sweep; configs[1]'s three runs, 261 120 positions; raw/r05/xlo_layers.log): [0, 24) 144, [6, 24) 140, [12, 24) 147,
  //         [18, 24) 145, [0, 12) 164, none 182 -- the late layers carry all of it, so the sweep runs in layers >= depth / 2 (~80 us per layer saved
  //         in the first half at 64 sequences; the same rule as precision 3's activation-lo set in the guided forward);
  //   >= 2: all four trunk GEMMs also carry the MX-fp4 weight-correction mini-tiles on every row -- the fp16 rounding of the
  //         WEIGHTS is 80 % of the sampled-logit error variance here (tests/diag/error_budget.py: rms 0.0082 single fp16, 0.0073 with hi + lo
  //         activation pairs, 0.0045 with the weight correction alone); both together: 5.3e-4 over configs[1]'s three reference runs.
  const bool xlo = c.precision >= 1;
  const bool wm = g->mini_ok && c.precision >= 2;   // (sequence tiles: 256 + 1 rows, or four tiles per 1024 + 1-row sequence)
  // Both coverage rules above (FFN-up only, late layers only) were measured WITH the weight correction on, on shapes the mini-tiles serve.  Shapes they
  // do not serve (other widths, heads of 32, sequences other than 256 / 1024 tokens) have no weight correction to carry the margin: there the hi + lo
  // LayerNorm outputs enter QKV and FFN-up of EVERY layer, as in rounds 3-4 (round-5 advice: the narrowed rule was a silent regression for them).
  const bool xlo_all = xlo && !wm;
  auto xlo_layer = [&](int l) { return xlo && (xlo_all || 2 * l + 1 >= c.depth); };   // (with the correction: the second half of the layers, see above)
  h16* const xlo_ffn = xlo ? g->x_lo : nullptr;          // the LayerNorm in front of FFN-up writes lo halves (the one in front of QKV only without the correction; the last one feeds the head)
  h16* const xlo_qkv = xlo_all ? g->x_lo : nullptr;
  // the LayerNorms write the MX-fp4 copy (+ scale bytes) only when a GEMM of THIS forward reads it (the buffers also exist for the pair forward)
  Fp4Rows f4x;
  if (wm && (g->wcorr_mask & 5)) { f4x.x4 = g->x4; f4x.x4s = g->x4s; f4x.nseq = nb; f4x.seq_rows = N; }
  const bool wo4 = wm && (g->wcorr_mask & 2);
  auto lo_set = [&](GemmArgs& ga, const uint8_t* a4, const uint8_t* a4s, int widx) {
    if (!wm || !((g->wcorr_mask >> (widx & 3)) & 1)) return;
    ga.nlo = 1; ga.lo[0] = {a4, a4s, g->w4lo[widx], g->w4los[widx]};
  };
  // QKV / FFN-up: consume the LayerNorm output
  auto xgemm = [&](GemmEpi epi, const h16* W, const float* bias, h16* out, int Nout, int widx) {
    GemmArgs ga{g->x_h16, W, bias, nullptr, nullptr, out, M, Nout, d, 0};
    if (wm) ga.seq_rows = N;
    lo_set(ga, g->x4, g->x4s, widx);
    if (wm && epi == EPI_GELU_H16 && (g->wcorr_mask & 8)) { ga.out4 = g->h4; ga.out4_scale = g->h4s; }
    if (((widx & 3) == 2 || xlo_all) && xlo_layer(widx >> 2)) { ga.K = 2 * d; ga.A2 = g->x_lo; ga.kw = d; }   // FFN-up only with the correction (see above)
    ga.sat = g->sat;
    gemm_rc |= gemm_tn(s, epi, ga, wm ? 257 : 0);
  };
  // out-proj / FFN-down: + residual (prev: the LayerNorm whose output is the residual, re-derived from the row statistics; null: the buffer's own rows)
  auto rgemm = [&](const h16* A, const h16* W, const float* bias, int K, int widx, const uint8_t* a4, const uint8_t* a4s, const float* ln_g, const float* ln_b) {
    GemmArgs ga{A, W, bias, g->y_f32, g->y_f32, nullptr, M, d, K, 0};
    if (ln_g) { ga.ln_stats = g->ln_stats; ga.ln_g = ln_g; ga.ln_b = ln_b; }
    if (wm) ga.seq_rows = N;
    lo_set(ga, a4, a4s, widx);
    gemm_rc |= gemm_tn(s, EPI_RES_F32, ga, wm ? 257 : 0);
  };
  {
    ProfScope p("embed_ln", s, true);
    EmbedArgs e{tokens, labels, drop, g->w_in, g->b_in, g->class_emb, g->pos, g->ln0g, g->ln0b,
                g->y_f32, g->x_h16, nb, c.seq, c.splits, g->gbits, d, c.nclass, g->tables};
    e.x_lo = c.depth ? (c.prenorm ? nullptr : xlo_qkv) : g->x_lo;
    e.f4 = f4x;
    embed_ln(s, e);
  }
  for (int l = 0; l < c.depth; ++l) {
    const mb_gen::Layer& L = g->layers[l];
    // post-norm (every shipped config): the fp32 residual stream lives in ONE buffer, y_f32, holding pre-LayerNorm rows; a LayerNorm writes only the
    // fp16 GEMM operand and {mean, rstd}, the next residual GEMM re-derives the normalised rows in its epilogue and updates y_f32 in place (layer 0's
    // first residual is the embedding LayerNorm output, stored as is by embed_ln).  use_prenorm (bert.py:49-59, 106-123): x = x + Attn(LN(x));
    // x = x + FFN(LN(x)): the buffer holds x itself, every LayerNorm only produces the GEMM operand, the residual GEMMs add the buffer's own rows.
    if (c.prenorm) { ProfScope p("layernorm", s, true); layernorm_rows(s, g->y_f32, L.ln1g, L.ln1b, 1e-12f, nullptr, g->x_h16, nullptr, M, d, xlo_qkv, f4x); }
    { ProfScope p("gemm_qkv", s, true); xgemm(EPI_H16, L.wqkv, L.bqkv, g->qkv, 3 * d, 4 * l); }
    { ProfScope p("attention", s, true); attention(s, g->qkv, g->att, nb, N, d, c.heads, wo4 ? g->att4 : nullptr, wo4 ? g->att4s : nullptr); }
    attn_rc |= attn_maps(l);
    { ProfScope p("gemm_attn_out", s, true);
      const bool re = !c.prenorm && l > 0;
      rgemm(g->att, L.wo, L.bo, d, 4 * l + 1, g->att4, g->att4s, re ? g->layers[l - 1].ln2g : nullptr, re ? g->layers[l - 1].ln2b : nullptr); }
    { ProfScope p("layernorm", s, true);
      layernorm_rows(s, g->y_f32, c.prenorm ? L.ln2g : L.ln1g, c.prenorm ? L.ln2b : L.ln1b, 1e-12f, nullptr, g->x_h16, c.prenorm ? nullptr : g->ln_stats, M, d, xlo_layer(l) ? xlo_ffn : nullptr, f4x); }
    { ProfScope p("gemm_ffn_up", s, true); xgemm(EPI_GELU_H16, L.w1, L.b1, g->h, f, 4 * l + 2); }
    { ProfScope p("gemm_ffn_down", s, true);
      rgemm(g->h, L.w2, L.b2, f, 4 * l + 3, g->h4, g->h4s, c.prenorm ? nullptr : L.ln1g, c.prenorm ? nullptr : L.ln1b); }
    if (!c.prenorm) { ProfScope p("layernorm", s, true);
      const bool last = l + 1 == c.depth;                  // the last one feeds the head: plain hi + lo rows
      layernorm_rows(s, g->y_f32, L.ln2g, L.ln2b, 1e-12f, nullptr, g->x_h16, g->ln_stats, M, d, last ? g->x_lo : xlo_qkv, last ? Fp4Rows{} : f4x); }
  }
  if (c.prenorm) { ProfScope p("layernorm", s, true); layernorm_rows(s, g->y_f32, g->lnag, g->lnab, 1e-12f, nullptr, g->x_h16, nullptr, M, d, g->x_lo); }   // norm_after_transformer
  gemm_rc |= head_gemms(g, logits, M, s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  if (attn_rc) return fail(-3, "attention maps: head dim %d / %d tokens not supported", d / c.heads, N);
  if (gemm_rc) return fail(-3, "a trunk GEMM of this forward (%d sequences x %d tokens, hidden %d, mlp %d; precision %d%s) is outside the half-tile "
                               "kernel's shapes: its correction mini-tiles cannot run", nb, N, d, f, c.precision, wm ? ", weight-correction mini-tiles" : "");
  return 0;
}

// Differential CFG forward (mb_gen_cfg.precision >= 1): nb = 2 * B sequences laid out [B conditional | B label-dropped twins] in every buffer.
// Wherever a fp16 GEMM operand is produced (embedding LayerNorm, the LayerNorms, attention output, GELU output), the conditional rows hold
// fp16(x_c) and the unconditional rows the DIFFERENCE fp16(x_u - x_c); the pair GEMM (gemm_ht.hip, PAIR) adds the two products for the
// unconditional outputs.  The fp32 residual stream, qkv and the logits hold ordinary values for both streams.  wmode: MX-fp4 correction
// pass for the fp16 rounding of the QKV / FFN-up weights (conditional rows; the unconditional outputs inherit it through acc_c).
int gen_forward_pair_impl(mb_gen* g, const int64_t* tokens, const int64_t* labels, const uint8_t* drop, float* logits, int B, bool wmode, hipStream_t s) {
  using namespace mb;
  const mb_gen_cfg& c = g->c;
  const int d = c.hidden, f = c.mlp, N = g->N, nb = 2 * B, M = nb * N, P = B * N;
  int rc = 0;
  // measured (profiles/r03_parity.md): the correction pass in layers >= depth / 2 alone buys about 60 % of the gain on the 12-bit runs for half of
  // the cost and next to nothing on the 14-bit one (and per GEMM type no subset is a cheaper "precise": section 5 there) -- a study knob
  // (mb_gen_set_wcorr, include/maskbit_hip_diag.h), not the default
  const int wfrom = g->wcorr_from;
  // The activation-lo sets of precision >= 3.  Rounds 4-5 knew the LayerNorm outputs' set only: over FOUR 14-bit / 256-step reference runs (1 002 744
  // positions; profiles/r05_coverage.md) QKV + FFN-up in all layers 496 mismatches, FFN-up alone 491-493, QKV alone 555, neither 625 -- the QKV set buys
  // nothing --, and round 5 ran it in FFN-up of the layers >= depth / 2 (531).
  // Round 6 (DESIGN.md "Precision"): the activation-lo set beyond the LayerNorm outputs -- the lo halves of the attention outputs (out-proj) and of the FFN
  // hiddens (FFN-down), each as e2m1 with per-(row, 64 columns) scales against e2m1 of the fp16 weight, in every layer: precision 3 = out-proj + FFN-up,
  // precision 4 (what heavy-tailed checkpoints need) = + FFN-down.  With exact weights the fp16 rounding of those three operands alone costs 7e-4 of
  // token mismatch on the early steps of a trained-like run (a third each); emulated on that run (tests/diag/error_budget.py EB_STUDY=r6): rms error
  // of the sampled logits' top-2 gap 0.0062 -> 0.0031 with the three sets (and the per-(row, 128 columns) weight-error scales).
  // (which GEMMs of which layers: mb_gen::alo_mask / alo_from -- bit 0 QKV, 1 out-proj, 2 FFN-up, 3 FFN-down)
  auto alo_on = [&](int gemm, int l) { return wmode && c.precision >= 3 && l >= wfrom && l >= g->alo_from && ((g->alo_mask >> gemm) & 1) && ((g->wcorr_mask >> gemm) & 1); };
  auto f4_for = [&](int consumer_layer, bool feeds_ffn = false) {   // what the producer of layer `consumer_layer`'s LayerNorm operand also writes
    Fp4Rows f;
    if (wmode && consumer_layer >= wfrom && (g->wcorr_mask & 5)) {
      f.x4 = g->x4; f.x4s = g->x4s; f.nseq = B; f.seq_rows = N;
      if (alo_on(feeds_ffn ? 2 : 0, consumer_layer)) { f.xl4 = g->xl4; f.xl4s = g->xl4s; }   // (the lo halves' e2m1 copy: only the LayerNorm in front of a GEMM that carries the set)
    }
    return f;
  };
  // lo: 0 = fp16 only, 1 = weight-correction mini-tiles (a4 / a4s = e2m1 of the conditional operand values), 2 = + the activation-lo set (al4 / al4s = e2m1
  // of the operand's lo halves, against e2m1 of the fp16 weight)
  auto pgemm = [&](GemmEpi epi, const h16* A, const h16* W, const float* bias, h16* out16, float* res, int Nout, int K, int widx, int lo,
                   const uint8_t* a4 = nullptr, const uint8_t* a4s = nullptr, const uint8_t* al4 = nullptr, const uint8_t* al4s = nullptr) {
    GemmArgs ga{A, W, bias, res, res, out16, M, Nout, K, 0};
    ga.pair_rows = P;
    ga.seq_rows = N;
    if (epi != EPI_RES_F32) ga.sat = g->sat;
    if (lo && !((g->wcorr_mask >> (widx & 3)) & 1)) lo = 0;
    if (lo) {
      ga.nlo = lo; ga.lo[0] = {a4, a4s, g->w4lo[widx], g->w4los[widx]};
      if (lo == 2) ga.lo[1] = {al4, al4s, g->w4[widx], g->w4s[widx]};
    }
    return ga;
  };
  {
    ProfScope p("embed_ln", s, true);
    // (tokens / labels / drop are laid out [B conditional | B twins]; the twins repeat the conditional tokens and labels with the drop flag set)
    EmbedArgs e{tokens, labels, nullptr, g->w_in, g->b_in, g->class_emb, g->pos, g->ln0g, g->ln0b,
                g->y_f32, g->x_h16, B, c.seq, c.splits, g->gbits, d, c.nclass, g->tables};
    e.f4 = f4_for(0);
    if (embed_pair(s, e)) {         // shapes the fused kernel does not serve: the two-kernel path
      e.drop = drop; e.nb = nb; e.f4 = Fp4Rows{};
      embed_ln(s, e);
      rc |= pairify_rows(s, g->y_f32, g->x_h16, P, d, f4_for(0));        // y_f32 holds the embedding LayerNorm's fp32 rows here
    }
  }
  // pre-norm: the first sub-layer normalises the embedding rows again (LayerNorm 1 of layer 0); post-norm: the embedding's own pair operands feed QKV
  if (c.prenorm && c.depth > 0) {
    ProfScope p("layernorm", s, true);
    rc |= layernorm_pair(s, g->y_f32, g->layers[0].ln1g, g->layers[0].ln1b, 1e-12f, g->x_h16, nullptr, P, d, f4_for(0));
  }
  for (int l = 0; l < c.depth; ++l) {
    const mb_gen::Layer& L = g->layers[l];
    const bool wl = wmode && l >= wfrom;
    const int xlo_mode = wl ? (alo_on(2, l) ? 2 : 1) : 0;
    { ProfScope p("gemm_qkv", s, true);
      GemmArgs ga = pgemm(EPI_H16, g->x_h16, L.wqkv, L.bqkv, g->qkv, nullptr, 3 * d, d, 4 * l, wl ? (alo_on(0, l) ? 2 : 1) : 0, g->x4, g->x4s, g->xl4, g->xl4s);
      rc |= gemm_tn(s, EPI_H16, ga, 257); }
    const bool wo4 = wl && (g->wcorr_mask & 2), wh4 = wl && (g->wcorr_mask & 8);
    const bool lo_o = wo4 && alo_on(1, l), lo_h = wh4 && alo_on(3, l);      // the producers also write the lo halves' e2m1 copies for a consumer that carries the set
    { ProfScope p("attention", s, true); rc |= attention_pair(s, g->qkv, g->att, B, N, d, c.heads, wo4 ? g->att4 : nullptr, wo4 ? g->att4s : nullptr,
                                                              lo_o ? g->attl4 : nullptr, lo_o ? g->attl4s : nullptr); }
    { ProfScope p("gemm_attn_out", s, true);
      GemmArgs ga = pgemm(EPI_RES_F32, g->att, L.wo, L.bo, nullptr, g->y_f32, d, d, 4 * l + 1, wl ? (alo_on(1, l) ? 2 : 1) : 0, g->att4, g->att4s, g->attl4, g->attl4s);
      if (l > 0 && !c.prenorm) { ga.ln_stats = g->ln_stats; ga.ln_g = g->layers[l - 1].ln2g; ga.ln_b = g->layers[l - 1].ln2b; }
      rc |= gemm_tn(s, EPI_RES_F32, ga, 257); }
    // post-norm: LayerNorm 1 follows the attention block; pre-norm: LayerNorm 2 precedes the FFN (same place in the launch order, other parameters;
    // the stream buffer then holds the raw residual and no GEMM re-derives a LayerNorm from the statistics)
    { ProfScope p("layernorm", s, true);
      rc |= layernorm_pair(s, g->y_f32, c.prenorm ? L.ln2g : L.ln1g, c.prenorm ? L.ln2b : L.ln1b, 1e-12f, g->x_h16, c.prenorm ? nullptr : g->ln_stats, P, d, f4_for(l, true)); }
    { ProfScope p("gemm_ffn_up", s, true);
      GemmArgs ga = pgemm(EPI_GELU_H16, g->x_h16, L.w1, L.b1, g->h, nullptr, f, d, 4 * l + 2, xlo_mode, g->x4, g->x4s, g->xl4, g->xl4s);
      if (wh4) { ga.out4 = g->h4; ga.out4_scale = g->h4s; }
      if (lo_h) { ga.out4l = g->hl4; ga.out4l_scale = g->hl4s; }
      rc |= gemm_tn(s, EPI_GELU_H16, ga, 257); }
    { ProfScope p("gemm_ffn_down", s, true);
      GemmArgs ga = pgemm(EPI_RES_F32, g->h, L.w2, L.b2, nullptr, g->y_f32, d, f, 4 * l + 3, wl ? (alo_on(3, l) ? 2 : 1) : 0, g->h4, g->h4s, g->hl4, g->hl4s);
      if (!c.prenorm) { ga.ln_stats = g->ln_stats; ga.ln_g = L.ln1g; ga.ln_b = L.ln1b; }
      rc |= gemm_tn(s, EPI_RES_F32, ga, 257); }
    { ProfScope p("layernorm", s, true);
      if (c.prenorm) {            // the next layer's LayerNorm 1 (bert.py:49-59, 106-123), or norm_after_transformer in front of the head
        if (l + 1 == c.depth) layernorm_rows(s, g->y_f32, g->lnag, g->lnab, 1e-12f, nullptr, g->x_h16, nullptr, M, d, g->x_lo);
        else rc |= layernorm_pair(s, g->y_f32, g->layers[l + 1].ln1g, g->layers[l + 1].ln1b, 1e-12f, g->x_h16, nullptr, P, d, f4_for(l + 1));
      }
      else if (l + 1 == c.depth) layernorm_rows(s, g->y_f32, L.ln2g, L.ln2b, 1e-12f, nullptr, g->x_h16, g->ln_stats, M, d, g->x_lo);   // feeds the head: plain hi (+ lo) rows
      else rc |= layernorm_pair(s, g->y_f32, L.ln2g, L.ln2b, 1e-12f, g->x_h16, g->ln_stats, P, d, f4_for(l + 1)); }
  }
  rc |= head_gemms(g, logits, M, s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  if (rc) return fail(-3, "differential CFG forward: a kernel refused the shape (%d pairs x %d tokens, hidden %d, mlp %d)", B, N, d, f);
  return 0;
}

// The kernels index with 32-bit element / byte offsets (rows * mlp * 4 < 2^32): forwards over more sequences than that allows run as
// independent chunks (sequences never interact), which also bounds the workspace of very large batches.
int gen_forward(mb_gen* g, const int64_t* tokens, const int64_t* labels, const uint8_t* drop, float* logits, int nb, hipStream_t s,
                float* attn = nullptr) {
  const int chunk = g->chunk_seqs;
  g_prof.begin_forward(true);
  if (nb <= chunk) return gen_forward_impl(g, tokens, labels, drop, logits, nb, s, attn);
  if (attn) return fail(-3, "attention maps are limited to %d sequences per call", chunk);
  const size_t P = (size_t)g->c.seq * g->c.splits;
  for (int b0 = 0; b0 < nb; b0 += chunk) {
    const int nc = nb - b0 < chunk ? nb - b0 : chunk;
    int rc = gen_forward_impl(g, tokens + (size_t)b0 * P, labels + b0, drop ? drop + b0 : nullptr, logits + (size_t)b0 * P * g->C, nc, s);
    if (rc) return rc;
  }
  return 0;
}

// Guided forward (sampling.py:83-88) over B samples: logits rows [0, B) conditional, [B, 2B) label-dropped.
int gen_forward_cfg(mb_gen* g, const int64_t* tokens, const int64_t* labels, float* logits, int B, hipStream_t s) {
  const size_t P = (size_t)g->c.seq * g->c.splits;
  const bool pair = g->pair_ok && g->c.precision >= 1;
  const bool wmode = pair && g->c.precision >= 2;       // weight-rounding correction pass (every step: weight rounding costs parity late in the run too)
  const int chunk = g->chunk_seqs / 2;                  // pairs per pass
  if (chunk < 1) return fail(-1, "engine holds %d sequences: too few for a guided forward", g->chunk_seqs);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nc = B - b0 < chunk ? B - b0 : chunk;
    HIP_TRY(hipMemcpyAsync(g->tok_cfg, tokens + (size_t)b0 * P, nc * P * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    // (the fused pair embedding reads the conditional tokens only: the twins' copy is needed by the two-kernel path and the plain fallback)
    if (!(pair && !g->c.embed_tables && g->c.bits <= 24))
      HIP_TRY(hipMemcpyAsync(g->tok_cfg + nc * P, tokens + (size_t)b0 * P, nc * P * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    if (!(nc == B && g->cfg_labels_ready == labels && g->cfg_ready_B == B)) {   // (mb_sample marks them ready for the steps of one call)
      HIP_TRY(hipMemcpyAsync(g->lab_cfg, labels + b0, nc * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
      HIP_TRY(hipMemcpyAsync(g->lab_cfg + nc, labels + b0, nc * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
      HIP_TRY(hipMemsetAsync(g->drop_cfg, 0, nc, s));
      HIP_TRY(hipMemsetAsync(g->drop_cfg + nc, 1, nc, s));
    }
    float* out = (nc == B) ? logits : g->logits_tmp;    // chunked: through a buffer of the engine's own, then to the two halves of the caller's
    if (b0 == 0) g_prof.begin_forward(false);           // one guided forward = all of its chunks
    int rc = pair ? gen_forward_pair_impl(g, g->tok_cfg, g->lab_cfg, g->drop_cfg, out, nc, wmode, s)
                  : gen_forward_impl(g, g->tok_cfg, g->lab_cfg, g->drop_cfg, out, 2 * nc, s);
    if (rc) return rc;
    if (nc != B) {
      HIP_TRY(hipMemcpyAsync(logits + (size_t)b0 * P * g->C, out, nc * P * g->C * sizeof(float), hipMemcpyDeviceToDevice, s));
      HIP_TRY(hipMemcpyAsync(logits + (size_t)(B + b0) * P * g->C, out + (size_t)nc * P * g->C, nc * P * g->C * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
  }
  return 0;
}

}  // namespace

extern "C" {

int mb_abi_version(void) { return MB_ABI_VERSION; }
const char* mb_last_error(void) { return g_err.c_str(); }

int mb_set_cu_count(int n) { mb::set_cu_count(n); return 0; }
int mb_prof_enable(int on) {
  if (!on) g_prof.drain();
  g_prof.on = on != 0;
  if (on) { g_prof.acc.clear(); g_prof.stride = on; g_prof.tick[0] = g_prof.tick[1] = 0; }
  return 0;
}
int mb_prof_read(char* buf, int buflen) {
  g_prof.drain();
  std::string out;
  char line[256];
  for (auto& kv : g_prof.acc) {
    snprintf(line, sizeof line, "%s %ld %.6f\n", g_prof.names[kv.first].c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if ((int)out.size() + 1 > buflen) return fail(-3, "mb_prof_read: buffer too small (%zu needed)", out.size() + 1);
  memcpy(buf, out.c_str(), out.size() + 1);
  return (int)out.size();
}

int mb_gemm_ex(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16,
               int M, int N, int K, const float* ln_stats, const float* ln_g, const float* ln_b, int period, int variant, mb_stream stream) {
  if (!A || !W || !bias || epi < 0 || epi > 4) return fail(-1, "mb_gemm_ex: bad arguments");
  if (K % 64) return fail(-1, "mb_gemm_ex: K must be a multiple of 64");
  if (ln_stats && (!ln_g || !ln_b || epi != mb::EPI_RES_F32)) return fail(-1, "mb_gemm_ex: LayerNorm residual needs gamma, beta and the fp32+residual epilogue");
  mb::GemmArgs a{(const h16*)A, (const h16*)W, bias, residual, out_f32, (h16*)out_h16, M, N, K, period, nullptr, ln_stats, ln_g, ln_b};
  ProfScope p("gemm_diag", (hipStream_t)stream);
  if (mb::gemm_tn((hipStream_t)stream, (mb::GemmEpi)epi, a, variant)) return fail(-3, "GEMM shape M=%d N=%d K=%d is outside the kernels' shapes", a.M, a.N, a.K);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_gemm_mini(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16, void* out4,
                 void* out4_scale, int rows, int pair, int N, int K, int nlo, const void* const* lo /* nlo x {A4, a_scale, W4, w_scale} */, mb_stream stream) {
  return mb_gemm_mini_seq(epi, A, W, bias, residual, out_f32, out_h16, out4, out4_scale, nullptr, nullptr, rows, pair, 0, N, K, nlo, lo, stream);
}
int mb_gemm_mini_seq(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16, void* out4,
                     void* out4_scale, void* out4l, void* out4l_scale, int rows, int pair, int seq_rows, int N, int K, int nlo, const void* const* lo, mb_stream stream) {
  if ((out4l || out4l_scale) && (!out4l || !out4l_scale || !out4 || !out4_scale)) return fail(-1, "mb_gemm_mini_seq: the lo copy (out4l / out4l_scale) rides with the value copy (out4 / out4_scale)");
  if (!A || !W || !bias || epi < 0 || epi > 2 || rows <= 0 || K <= 0 || K % 64 || nlo < 0 || nlo > 2 || (nlo && !lo)) return fail(-1, "mb_gemm_mini: bad arguments");
  mb::GemmArgs a{(const h16*)A, (const h16*)W, bias, residual, out_f32, (h16*)out_h16, pair ? 2 * rows : rows, N, K, 0};
  if (pair) a.pair_rows = rows;
  a.seq_rows = seq_rows;
  a.nlo = nlo;
  for (int i = 0; i < nlo; ++i) a.lo[i] = {(const uint8_t*)lo[4 * i], (const uint8_t*)lo[4 * i + 1], (const uint8_t*)lo[4 * i + 2], (const uint8_t*)lo[4 * i + 3]};
  a.out4 = (uint8_t*)out4; a.out4_scale = (uint8_t*)out4_scale;
  a.out4l = (uint8_t*)out4l; a.out4l_scale = (uint8_t*)out4l_scale;
  if ((!seq_rows && a.M % 257) || !mb::gemm_ht_supported((mb::GemmEpi)epi, a)) return fail(-3, "mb_gemm_mini: shape not supported by the sequence-aligned tiles");
  ProfScope p("gemm_diag", (hipStream_t)stream);
  if (mb::gemm_tn((hipStream_t)stream, (mb::GemmEpi)epi, a, 257)) return fail(-3, "mb_gemm_mini: shape refused");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_gemm_mini_split(int epi, const void* A_hi, const void* A_lo, const void* W, const float* bias, void* out_h16, void* out4, void* out4_scale,
                       int rows, int N, int kw, const void* const* lo /* {A4, a_scale, W4, w_scale} */, mb_stream stream) {
  if (!A_hi || !A_lo || !W || !bias || !out_h16 || !lo || epi < 0 || epi > 1 || rows <= 0 || rows % 257 || kw <= 0 || kw % 128)
    return fail(-1, "mb_gemm_mini_split: bad arguments");
  mb::GemmArgs a{(const h16*)A_hi, (const h16*)W, bias, nullptr, nullptr, (h16*)out_h16, rows, N, 2 * kw, 0};
  a.A2 = (const h16*)A_lo; a.kw = kw;
  a.nlo = 1;
  a.lo[0] = {(const uint8_t*)lo[0], (const uint8_t*)lo[1], (const uint8_t*)lo[2], (const uint8_t*)lo[3]};
  a.out4 = (uint8_t*)out4; a.out4_scale = (uint8_t*)out4_scale;
  if (!mb::gemm_ht_supported((mb::GemmEpi)epi, a)) return fail(-3, "mb_gemm_mini_split: shape not supported by the sequence-aligned tiles");
  ProfScope p("gemm_diag", (hipStream_t)stream);
  if (mb::gemm_tn((hipStream_t)stream, (mb::GemmEpi)epi, a, 257)) return fail(-3, "mb_gemm_mini_split: shape refused");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_attention_pair(const void* qkv, void* out_h16, int pairs, int N, int d, int heads, mb_stream stream) {
  if (!qkv || !out_h16 || pairs <= 0 || N <= 0 || heads <= 0 || d % heads) return fail(-1, "mb_attention_pair: bad arguments");
  ProfScope p("attention", (hipStream_t)stream);
  if (mb::attention_pair((hipStream_t)stream, (const h16*)qkv, (h16*)out_h16, pairs, N, d, heads, nullptr, nullptr))
    return fail(-3, "mb_attention_pair: head width %d (N = %d tokens) is outside the attention kernels", d / heads, N);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_attention_pair_f4(const void* qkv, void* out_h16, void* out4, void* out4_scale, void* out4l, void* out4l_scale, int pairs, int N, int d, int heads, mb_stream stream) {
  if (!qkv || !out_h16 || !out4 || !out4_scale || (!out4l) != (!out4l_scale) || pairs <= 0 || N <= 0 || heads <= 0 || d % heads) return fail(-1, "mb_attention_pair_f4: bad arguments");
  ProfScope p("attention", (hipStream_t)stream);
  if (mb::attention_pair((hipStream_t)stream, (const h16*)qkv, (h16*)out_h16, pairs, N, d, heads, (uint8_t*)out4, (uint8_t*)out4_scale, (uint8_t*)out4l, (uint8_t*)out4l_scale))
    return fail(-3, "mb_attention_pair_f4: head width %d / N = %d tokens: no e2m1 copy for this shape", d / heads, N);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_gemm_act_split(int epi, const void* A_hi, const void* A_lo, const void* W, const float* bias, const float* residual, float* out_f32,
                      void* out_h16, int M, int N, int kw, int variant, mb_stream stream) {
  if (!A_hi || !A_lo || !W || !bias || epi < 0 || epi > 3 || kw <= 0 || kw % 64) return fail(-1, "mb_gemm_act_split: bad arguments");
  mb::GemmArgs a{(const h16*)A_hi, (const h16*)W, bias, residual, out_f32, (h16*)out_h16, M, N, 2 * kw, 0};
  a.A2 = (const h16*)A_lo; a.kw = kw;
  ProfScope p("gemm_diag", (hipStream_t)stream);
  if (mb::gemm_tn((hipStream_t)stream, (mb::GemmEpi)epi, a, variant)) return fail(-3, "GEMM shape M=%d N=%d K=%d is outside the kernels' shapes", a.M, a.N, a.K);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_w4_from_f32(const float* W, int N, int K, void* dst4, void* scale_out, mb_stream stream) {
  if (!W || !dst4 || !scale_out || N <= 0 || K <= 0 || N % 64 || K % 128) return fail(-1, "mb_w4_from_f32: bad arguments");
  mb::w4_from_f32((hipStream_t)stream, W, (uint8_t*)dst4, N, K, (uint8_t*)scale_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_w4lo_from_f32(const float* W, int N, int K, void* dst4, void* scale_out, mb_stream stream) {
  if (!W || !dst4 || !scale_out || N <= 0 || K <= 0 || N % 64 || K % 128) return fail(-1, "mb_w4lo_from_f32: bad arguments");
  mb::w4lo_from_f32((hipStream_t)stream, W, (uint8_t*)dst4, N, K, (uint8_t*)scale_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_layernorm_f4(const float* y, const float* gamma, const float* beta, float eps, float* x_f32, void* x_h16, void* x4, void* x4_scale, void* xl4,
                    void* xl4_scale, int M, int d, mb_stream stream) {
  if (!y || !gamma || !beta || (!x4 && !xl4) || (x4 && !x4_scale) || (xl4 && !xl4_scale) || M <= 0 || M % 257 || (d != 768 && d != 1024))
    return fail(-1, "mb_layernorm_f4: bad arguments (d must be 768 or 1024, M a multiple of 257)");
  mb::Fp4Rows f4{(uint8_t*)x4, (uint8_t*)x4_scale, (uint8_t*)xl4, (uint8_t*)xl4_scale, M / 257};
  mb::layernorm_rows((hipStream_t)stream, y, gamma, beta, eps, x_f32, (h16*)x_h16, nullptr, M, d, nullptr, f4);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}
int mb_layernorm(const float* y, const float* gamma, const float* beta, float eps, float* x_f32, void* x_h16, void* x_lo, float* stats, int M,
                 int d, mb_stream stream) {
  if (!y || !gamma || !beta || M <= 0 || d <= 0 || d > 2048) return fail(-1, "mb_layernorm: bad arguments");
  mb::layernorm_rows((hipStream_t)stream, y, gamma, beta, eps, x_f32, (h16*)x_h16, stats, M, d, (h16*)x_lo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

// Diagnostic entry: one GEMM of the trunk family on caller buffers (tests and tools/gemm_bench.py).
int mb_gemm(int epi, const void* A, const void* W, const float* bias, const float* residual, float* out_f32, void* out_h16,
            int M, int N, int K, int period, int variant, mb_stream stream) {
  if (!A || !W || !bias || epi < 0 || epi > 4) return fail(-1, "mb_gemm: bad arguments");
  if (K % 64) return fail(-1, "mb_gemm: K must be a multiple of 64");
  mb::GemmArgs a{(const h16*)A, (const h16*)W, bias, residual, out_f32, (h16*)out_h16, M, N, K, period};
  ProfScope p("gemm_diag", (hipStream_t)stream);
  if (mb::gemm_tn((hipStream_t)stream, (mb::GemmEpi)epi, a, variant)) return fail(-3, "GEMM shape M=%d N=%d K=%d is outside the kernels' shapes", a.M, a.N, a.K);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

int mb_gen_create(const mb_gen_cfg* cfg, int max_seqs, mb_gen** out) {
  if (!cfg || !out || max_seqs <= 0) return fail(-1, "mb_gen_create: bad arguments");
  const mb_gen_cfg& c = *cfg;
  if (c.splits <= 0 || c.bits % c.splits) return fail(-1, "bits (%d) must be divisible by splits (%d)", c.bits, c.splits);
  if (c.bits > 24) return fail(-1, "bits > 24 is not supported");
  if (c.hidden % 64 || c.mlp % 64) return fail(-1, "hidden (%d) and mlp (%d) must be multiples of 64", c.hidden, c.mlp);
  if (c.hidden > 2048) return fail(-1, "hidden > 2048 is not supported");
  const int dh = c.heads > 0 ? c.hidden / c.heads : 0;
  if (c.heads <= 0 || c.hidden % c.heads || (dh != 32 && dh != 64)) return fail(-1, "hidden/heads must be 32 or 64 (got %d)", dh);
  if (c.seq < 1 || c.seq > 4096) return fail(-1, "seq = %d outside [1, 4096]", c.seq);   // > 287 tokens: streaming attention kernel
  if ((size_t)c.seq * c.splits > 8192) return fail(-1, "seq * splits = %zu exceeds the step kernel's 8192 positions", (size_t)c.seq * c.splits);
  const int C = 1 << (c.bits / c.splits);
  if (C > 4096 || (c.splits * C) % 4) return fail(-1, "unsupported group codebook size %d (the fused step kernel holds up to 4096 codes per group)", C);
  if ((c.prenorm != 0 && c.prenorm != 1) || (c.embed_tables != 0 && c.embed_tables != 1)) return fail(-1, "prenorm / embed_tables must be 0 or 1");
  if (c.embed_tables && c.splits > 8) return fail(-1, "embed_tables supports up to 8 token groups");
  if (c.precision < 0 || c.precision > 4) return fail(-1, "precision must be 0 .. 4 (MB_PREC_FP16 / _DIFF / _WCORR / _ALO / _ALO_ALL)");
  mb_gen* g = new mb_gen();
  g->c = c; g->max_seqs = max_seqs; g->N = c.seq + 1; g->gbits = c.bits / c.splits; g->C = C;
  (void)hipGetDevice(&g->device);
  // rows per forward pass: 32-bit byte offsets inside the kernels need rows * max(mlp, 3 * hidden) * 4 < 2^32 (gemm_ht_supported)
  const size_t widest = (size_t)(c.mlp > 3 * c.hidden ? c.mlp : 3 * c.hidden);
  const size_t max_rows = ((1ull << 32) - 1) / (4 * widest);
  g->chunk_seqs = (int)std::min<size_t>((size_t)max_seqs, std::max<size_t>(1, max_rows / g->N));
  const size_t d = c.hidden, f = c.mlp, M = (size_t)g->chunk_seqs * g->N;
  int rc = 0;
  g->layers.resize(c.depth);
  rc |= galloc(g, &g->split_tmp, 1);
  rc |= galloc(g, &g->sat, 1);
  if (!rc) (void)hipMemset(g->sat, 0, sizeof(unsigned));
  for (auto& L : g->layers) {
    rc |= galloc(g, &L.wqkv, 3 * d * d); rc |= galloc(g, &L.bqkv, 3 * d);
    rc |= galloc(g, &L.wo, d * d); rc |= galloc(g, &L.bo, d);
    rc |= galloc(g, &L.w1, f * d); rc |= galloc(g, &L.b1, f);
    rc |= galloc(g, &L.w2, d * f); rc |= galloc(g, &L.b2, d);
    rc |= galloc(g, &L.ln1g, d); rc |= galloc(g, &L.ln1b, d); rc |= galloc(g, &L.ln2g, d); rc |= galloc(g, &L.ln2b, d);
  }
  rc |= galloc(g, &g->w_in, d * c.bits); rc |= galloc(g, &g->b_in, d);
  if (c.prenorm) { rc |= galloc(g, &g->lnag, d); rc |= galloc(g, &g->lnab, d); }
  if (c.embed_tables) { rc |= galloc(g, &g->tables, (size_t)c.splits * (C + 1) * d); rc |= galloc(g, &g->bias_pos, (size_t)c.seq * c.splits * C); }
  rc |= galloc(g, &g->class_emb, (size_t)(c.nclass + 1) * d); rc |= galloc(g, &g->pos, (size_t)g->N * d);
  rc |= galloc(g, &g->ln0g, d); rc |= galloc(g, &g->ln0b, d);
  rc |= galloc(g, &g->wl, d * d); rc |= galloc(g, &g->wl_lo, d * d); rc |= galloc(g, &g->bl, d); rc |= galloc(g, &g->lnhg, d); rc |= galloc(g, &g->lnhb, d);
  rc |= galloc(g, &g->wp, (size_t)c.splits * C * d); rc |= galloc(g, &g->bp, (size_t)c.splits * C); rc |= galloc(g, &g->head_scale, 2);
  if (!c.embed_tables) rc |= galloc(g, &g->wp_lo, (size_t)c.splits * C * d);
  rc |= galloc(g, &g->y_f32, M * d); rc |= galloc(g, &g->ln_stats, M * 2); rc |= galloc(g, &g->x_h16, M * d);
  rc |= galloc(g, &g->x_lo, M * d);   // lo halves of the LayerNorm outputs: always for the head GEMMs, precision >= 1 for QKV / FFN-up of plain forwards
  rc |= galloc(g, &g->qkv, M * 3 * d); rc |= galloc(g, &g->att, M * d); rc |= galloc(g, &g->h, M * f);
  // MX-fp4 mini-tile passes (precision 2 / 3): 257- / 1025-token sequences, vector LayerNorm widths, heads of 64 (the attention kernels' e2m1 output), whole mini-tiles
  g->mini_ok = c.precision >= 2 && (c.seq == 256 || c.seq == 1024) && (c.hidden == 768 || c.hidden == 1024) && c.mlp % 256 == 0 && c.hidden / c.heads == 64;   // (FFN-up's N = mlp: whole 256-column tiles)
  // differential CFG forward: 257-token sequences (pair tiles = 2 x 128 tokens + the class pair), vector LayerNorm widths, plain fp16 operands
  // (round 5: also the 1024 + 1-token models of 512 x 512 images -- a pair tile is 128 tokens of a sequence pair whatever the sequence length)
  if (c.precision == 3) { g->alo_mask = g->alo_mask_built = 6; }              // out-proj + FFN-up, every layer
  if (c.precision >= 4) { g->alo_mask = 14; g->alo_mask_built = 15; }         // + FFN-down (the QKV operands exist for coverage studies only)
  g->pair_ok = c.precision >= 1 && (c.seq == 256 || c.seq == 1024) && (c.hidden == 768 || c.hidden == 1024) && c.mlp % 256 == 0 && g->chunk_seqs >= 2 &&
               (c.precision == 1 || g->mini_ok);
  if (g->mini_ok) {
    // e2m1 operands: row stride of the fp16 sibling (2 * width bytes, first width / 2 used); scale bytes in lane order: [width / 64][sequences][256]
    const size_t ns = (size_t)g->chunk_seqs * c.seq;
    rc |= galloc(g, &g->x4, M * 2 * d); rc |= galloc(g, &g->x4s, (d / 64) * ns + 256);
    rc |= galloc(g, &g->att4, M * 2 * d); rc |= galloc(g, &g->att4s, (d / 64) * ns + 256);
    rc |= galloc(g, &g->h4, M * 2 * f); rc |= galloc(g, &g->h4s, (f / 64) * ns + 256);
    if (g->alo_mask_built & 5) { rc |= galloc(g, &g->xl4, M * 2 * d); rc |= galloc(g, &g->xl4s, (d / 64) * ns + 256); }
    if (g->alo_mask_built & 2) {
      rc |= galloc(g, &g->attl4, M * 2 * d); rc |= galloc(g, &g->attl4s, (d / 64) * ns + 256);
      if (!rc) { (void)hipMemset(g->attl4, 0, M * 2 * d); (void)hipMemset(g->attl4s, 0, (d / 64) * ns + 256); }
    }
    if (g->alo_mask_built & 8) {
      rc |= galloc(g, &g->hl4, M * 2 * f); rc |= galloc(g, &g->hl4s, (f / 64) * ns + 256);
      if (!rc) { (void)hipMemset(g->hl4, 0, M * 2 * f); (void)hipMemset(g->hl4s, 0, (f / 64) * ns + 256); }
    }
    if (!rc) {
      (void)hipMemset(g->x4, 0, M * 2 * d); (void)hipMemset(g->x4s, 0, (d / 64) * ns + 256);
      (void)hipMemset(g->att4, 0, M * 2 * d); (void)hipMemset(g->att4s, 0, (d / 64) * ns + 256);
      (void)hipMemset(g->h4, 0, M * 2 * f); (void)hipMemset(g->h4s, 0, (f / 64) * ns + 256);
      if (g->xl4) { (void)hipMemset(g->xl4, 0, M * 2 * d); (void)hipMemset(g->xl4s, 0, (d / 64) * ns + 256); }
    }
    g->w4lo.assign((size_t)4 * c.depth, nullptr); g->w4los.assign((size_t)4 * c.depth, nullptr);
    g->w4.assign((size_t)4 * c.depth, nullptr); g->w4s.assign((size_t)4 * c.depth, nullptr);
    for (int l = 0; l < c.depth; ++l) {
      // (mini-tile-packed: half a byte per weight; one scale byte per (weight row, 128 columns))
      rc |= galloc(g, &g->w4lo[4 * l], 3 * d * d / 2); rc |= galloc(g, &g->w4los[4 * l], 3 * d * d / 128);
      rc |= galloc(g, &g->w4lo[4 * l + 1], d * d / 2); rc |= galloc(g, &g->w4los[4 * l + 1], d * d / 128);
      rc |= galloc(g, &g->w4lo[4 * l + 2], f * d / 2); rc |= galloc(g, &g->w4los[4 * l + 2], f * d / 128);
      rc |= galloc(g, &g->w4lo[4 * l + 3], d * f / 2); rc |= galloc(g, &g->w4los[4 * l + 3], d * f / 128);
      // e2m1 of the fp16 weight VALUES for the GEMMs that carry an activation-lo set (precision >= 3)
      const size_t wn[4] = {3 * d * d, d * d, f * d, d * f};
      for (int q = 0; q < 4; ++q)
        if ((g->alo_mask_built >> q) & 1) { rc |= galloc(g, &g->w4[4 * l + q], wn[q] / 2); rc |= galloc(g, &g->w4s[4 * l + q], wn[q] / 128); }
    }
  }
  const size_t P = (size_t)c.seq * c.splits, B = max_seqs;
  rc |= galloc(g, &g->tok_a, B * P); rc |= galloc(g, &g->tok_b, B * P); rc |= galloc(g, &g->tok_cfg, B * P);
  rc |= galloc(g, &g->pred, B * P); rc |= galloc(g, &g->codes, B * c.seq);
  rc |= galloc(g, &g->lab_cfg, B); rc |= galloc(g, &g->drop_cfg, B); rc |= galloc(g, &g->logits, B * P * C);
  if (g->chunk_seqs < max_seqs) rc |= galloc(g, &g->logits_tmp, (size_t)g->chunk_seqs * P * C);
  if (rc) { mb_gen_destroy(g); return rc; }
  *out = g;
  return 0;
}

void mb_gen_destroy(mb_gen* g) {
  if (!g) return;
  for (void* p : g->owned) (void)hipFree(p);
  delete g;
}

int mb_gen_load(mb_gen* g, const char* name, const float* data, const int64_t* shape, int ndim, mb_stream stream) {
  if (!g || !name || !data) return fail(-1, "mb_gen_load: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const mb_gen_cfg& c = g->c;
  const size_t d = c.hidden, f = c.mlp;
  size_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
  float* dst_f = nullptr; h16* dst_h = nullptr; size_t want = 0;
  int wrows = 0, wcols = 0, sidx = -1;                 // GEMM weights: [rows, cols] and their output-scale slot
  int l = -1, sub = -1; char rest[96] = {0};
  std::string n(name);
  // the embedding-table generator (Bert) has neither a bit projection nor an untied prediction layer: those keys belong to LFQBert checkpoints
  if (c.embed_tables && (n.rfind("prediction_layer.", 0) == 0 || n.rfind("input_proj.", 0) == 0))
    return fail(-2, "mb_gen_load: unknown checkpoint entry '%s' (embed_tables engine: tok_emb_list.* / bias.* instead)", name);
  if (sscanf(name, "transformer.layers.%d.%d.%95s", &l, &sub, rest) == 3) {
    if (l < 0 || l >= c.depth) return fail(-2, "layer index out of range in '%s'", name);
    mb_gen::Layer& L = g->layers[l];
    std::string r(rest);
    if (sub == 0) {
      if (r == "mha.in_proj_weight") { dst_h = L.wqkv; want = 3 * d * d; wrows = 3 * d; wcols = d; sidx = 4 * l; }
      else if (r == "mha.in_proj_bias") { dst_f = L.bqkv; want = 3 * d; }
      else if (r == "mha.out_proj.weight") { dst_h = L.wo; want = d * d; wrows = d; wcols = d; sidx = 4 * l + 1; }
      else if (r == "mha.out_proj.bias") { dst_f = L.bo; want = d; }
      else if (r == "norm.weight") { dst_f = L.ln1g; want = d; }
      else if (r == "norm.bias") { dst_f = L.ln1b; want = d; }
    } else if (sub == 1) {
      if (r == "net.0.weight") { dst_h = L.w1; want = f * d; wrows = f; wcols = d; sidx = 4 * l + 2; }
      else if (r == "net.0.bias") { dst_f = L.b1; want = f; }
      else if (r == "net.2.weight") { dst_h = L.w2; want = d * f; wrows = d; wcols = f; sidx = 4 * l + 3; }
      else if (r == "net.2.bias") { dst_f = L.b2; want = d; }
      else if (r == "norm.weight") { dst_f = L.ln2g; want = d; }
      else if (r == "norm.bias") { dst_f = L.ln2b; want = d; }
    }
  } else if (n == "pos_emb") { dst_f = g->pos; want = (size_t)g->N * d; }
  else if (n == "class_emb.weight") { dst_f = g->class_emb; want = (size_t)(c.nclass + 1) * d; }
  else if (n == "input_proj.weight") { dst_f = g->w_in; want = d * c.bits; }
  else if (n == "input_proj.bias") { dst_f = g->b_in; want = d; }
  else if (n == "first_layer.0.weight") { dst_f = g->ln0g; want = d; }
  else if (n == "first_layer.0.bias") { dst_f = g->ln0b; want = d; }
  else if (n == "last_layer.0.weight") { dst_h = g->wl; want = d * d; wrows = d; wcols = d; sidx = 4 * c.depth; }
  else if (n == "last_layer.0.bias") { dst_f = g->bl; want = d; }
  else if (n == "last_layer.2.weight") { dst_f = g->lnhg; want = d; }
  else if (n == "last_layer.2.bias") { dst_f = g->lnhb; want = d; }
  else if (n == "prediction_layer.weight") { dst_h = g->wp; want = (size_t)c.splits * g->C * d; wrows = c.splits * g->C; wcols = d; sidx = 4 * c.depth + 1; }
  else if (n == "prediction_layer.bias") { dst_f = g->bp; want = (size_t)c.splits * g->C; }
  else if (n == "bits_to_indices") return 0;   // derived buffer (bert.py:383-384): recomputed on the device
  else if (c.prenorm && n == "norm_after_transformer.weight") { dst_f = g->lnag; want = d; }
  else if (c.prenorm && n == "norm_after_transformer.bias") { dst_f = g->lnab; want = d; }
  else if (c.embed_tables) {
    int q = -1;
    if (sscanf(name, "tok_emb_list.%d.weight", &q) == 1 && n == "tok_emb_list." + std::to_string(q) + ".weight") {
      // Bert (bert.py:224-226, 329-332): the table is the input embedding (fp32, gathered) AND, rows 0..C-1, the output head
      if (q < 0 || q >= c.splits) return fail(-2, "group index out of range in '%s'", name);
      const size_t rows = (size_t)g->C + 1;
      if (numel != rows * d) return fail(-4, "mb_gen_load: '%s' has %zu elements, expected %zu", name, numel, rows * d);
      HIP_TRY(hipMemcpyAsync(g->tables + (size_t)q * rows * d, data, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
      mb::cast_f32_to_h16(s, data, g->wp + (size_t)q * g->C * d, (size_t)g->C * d);
      g->loaded++;
      return 0;
    }
    if (sscanf(name, "bias.%d", &q) == 1 && n == "bias." + std::to_string(q)) {
      if (q < 0 || q >= c.splits) return fail(-2, "group index out of range in '%s'", name);
      if (numel != (size_t)c.seq * g->C) return fail(-4, "mb_gen_load: '%s' has %zu elements, expected %zu", name, numel, (size_t)c.seq * g->C);
      HIP_TRY(hipMemcpy2DAsync(g->bias_pos + (size_t)q * g->C, (size_t)c.splits * g->C * sizeof(float), data, (size_t)g->C * sizeof(float),
                               (size_t)g->C * sizeof(float), (size_t)c.seq, hipMemcpyDeviceToDevice, s));
      g->loaded++;
      return 0;
    }
  }
  if (!dst_f && !dst_h) return fail(-2, "mb_gen_load: unknown checkpoint entry '%s'", name);
  if (numel != want) return fail(-4, "mb_gen_load: '%s' has %zu elements, expected %zu", name, numel, want);
  if (dst_f == g->w_in) mb::transpose_f32(s, data, g->w_in, (int)d, c.bits);       // [d,K] -> [K,d] for the embed kernel
  else if (dst_h == g->wl) mb::split_f32_to_h16_planes(s, data, g->wl, g->wl_lo, wrows, wcols, g->head_scale, g->split_tmp);
  else if (dst_h == g->wp) mb::split_f32_to_h16_planes(s, data, g->wp, g->wp_lo, wrows, wcols, g->head_scale + 1, g->split_tmp);
  else if (dst_h) {
    mb::cast_f32_to_h16(s, data, dst_h, numel);
    if (g->mini_ok && sidx >= 0 && sidx < 4 * c.depth) {
      mb::w4lo_from_f32(s, data, g->w4lo[sidx], wrows, wcols, g->w4los[sidx]);
      if (g->w4[sidx]) mb::w4_from_f32(s, data, g->w4[sidx], wrows, wcols, g->w4s[sidx]);
    }
  }
  else HIP_TRY(hipMemcpyAsync(dst_f, data, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
  g->loaded++;
  return 0;
}

int mb_gen_set_alo(mb_gen* g, int from_layer, int gemm_mask) {
  if (!g || from_layer < 0 || from_layer > g->c.depth || gemm_mask < 0 || gemm_mask > 15) return fail(-1, "mb_gen_set_alo: layer outside [0, depth] or mask outside [0, 15]");
  if ((gemm_mask & ~g->alo_mask_built) || (gemm_mask && from_layer < g->alo_from_built))
    return fail(-1, "mb_gen_set_alo: the handle was created (precision %d) with the activation-lo operands of GEMM mask %d from layer %d on only", g->c.precision, g->alo_mask_built, g->alo_from_built);
  g->alo_mask = gemm_mask; g->alo_from = from_layer;
  return 0;
}

int mb_gen_set_wcorr(mb_gen* g, int from_layer, int gemm_mask) {
  if (!g || from_layer < 0 || from_layer > g->c.depth || gemm_mask < 0 || gemm_mask > 15) return fail(-1, "mb_gen_set_wcorr: layer outside [0, depth] or mask outside [0, 15]");
  g->wcorr_from = from_layer;
  g->wcorr_mask = gemm_mask;
  return 0;
}

int mb_gen_saturation_count(mb_gen* g, unsigned* count, int reset, mb_stream stream) {
  if (!g || !count) return fail(-1, "mb_gen_saturation_count: null argument");
  HIP_TRY(hipMemcpyAsync(count, g->sat, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (reset) HIP_TRY(hipMemsetAsync(g->sat, 0, sizeof(unsigned), (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int mb_gen_forward(mb_gen* g, const int64_t* tokens, const int64_t* labels, const uint8_t* drop, float* logits,
                   int nb, mb_stream stream) {
  if (!g || !tokens || !labels || !logits) return fail(-1, "mb_gen_forward: null argument");
  if (nb <= 0 || nb > g->max_seqs) return fail(-1, "mb_gen_forward: nb=%d outside [1, %d]", nb, g->max_seqs);
  return gen_forward(g, tokens, labels, drop, logits, nb, (hipStream_t)stream);
}

int mb_gen_forward_cfg(mb_gen* g, const int64_t* tokens, const int64_t* labels, float* logits, int B, mb_stream stream) {
  if (!g || !tokens || !labels || !logits) return fail(-1, "mb_gen_forward_cfg: null argument");
  if (B <= 0 || 2 * B > g->max_seqs) return fail(-1, "mb_gen_forward_cfg: B=%d needs %d sequences, engine holds %d", B, 2 * B, g->max_seqs);
  return gen_forward_cfg(g, tokens, labels, logits, B, (hipStream_t)stream);
}

int mb_gen_forward_attn(mb_gen* g, const int64_t* tokens, const int64_t* labels, const uint8_t* drop, float* logits, float* attn,
                        int nb, mb_stream stream) {
  if (!g || !tokens || !labels || !logits || !attn) return fail(-1, "mb_gen_forward_attn: null argument");
  if (nb <= 0 || nb > g->max_seqs) return fail(-1, "mb_gen_forward_attn: nb=%d outside [1, %d]", nb, g->max_seqs);
  return gen_forward(g, tokens, labels, drop, logits, nb, (hipStream_t)stream, attn);
}

int mb_sample_step(const float* logits_c, const float* logits_u, float scale, float temperature,
                   const float* exp_noise, const float* conf_noise, int k_mask_len, const int64_t* tokens_in,
                   int64_t* tokens_out, int64_t* pred_out, int B, int n, int m, int C, mb_stream stream) {
  if (!logits_c || !exp_noise || !conf_noise || !tokens_in || !tokens_out) return fail(-1, "mb_sample_step: null argument");
  if (tokens_in == tokens_out) return fail(-1, "mb_sample_step: tokens_in and tokens_out must not alias");
  if (B <= 0 || n <= 0 || m <= 0 || C <= 0) return fail(-1, "mb_sample_step: bad sizes");
  mb::StepArgs a{logits_c, logits_u, scale, temperature, exp_noise, conf_noise, k_mask_len, tokens_out, pred_out, B, n * m, C};
  ProfScope p("sample_step", (hipStream_t)stream);
  if (mb::sample_step((hipStream_t)stream, a, tokens_in)) return fail(-1, "mb_sample_step: C=%d or n*m=%d too large", C, n * m);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"

// ================================================================================================
// decoder handle (kernels in decoder.hip)
// ================================================================================================
extern "C" {

int mb_dec_create(const mb_dec_cfg* cfg, int max_batch, mb_dec** out) {
  if (!cfg || !out || max_batch <= 0) return fail(-1, "mb_dec_create: bad arguments");
  std::string err;
  mb_dec* d = mb::dec_create(*cfg, max_batch, err);
  if (!d) return fail(-1, "mb_dec_create: %s", err.c_str());
  *out = d;
  return 0;
}
void mb_dec_destroy(mb_dec* d) { mb::dec_destroy(d); }
int mb_dec_load(mb_dec* d, const char* name, const float* data, const int64_t* shape, int ndim, mb_stream stream) {
  if (!d || !name || !data) return fail(-1, "mb_dec_load: bad arguments");
  std::string err;
  int rc = mb::dec_load(d, name, data, shape, ndim, (hipStream_t)stream, err);
  if (rc) return fail(rc, "mb_dec_load: %s", err.c_str());
  return 0;
}
int mb_dec_decode(mb_dec* d, const int64_t* tokens, float* img_nchw, uint8_t* img_nhwc_u8, int B, mb_stream stream) {
  if (!d || !tokens) return fail(-1, "mb_dec_decode: null argument");
  std::string err;
  ProfScope p("decode", (hipStream_t)stream);
  int rc = mb::dec_decode(d, tokens, img_nchw, img_nhwc_u8, B, (hipStream_t)stream, err);
  if (rc) return fail(rc, "mb_dec_decode: %s", err.c_str());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

int mb_dec_saturation_count(mb_dec* d, unsigned* count, int reset, mb_stream stream) {
  if (!d || !count) return fail(-1, "mb_dec_saturation_count: null argument");
  if (mb::dec_saturation_count(d, count, reset != 0, (hipStream_t)stream)) return fail(-10, "mb_dec_saturation_count: copy failed");
  return 0;
}

int mb_enc_encode(mb_dec* d, const float* img_nchw, int64_t* indices, float* zq, float* zraw, int B, mb_stream stream) {
  if (!d || !img_nchw || !indices) return fail(-1, "mb_enc_encode: null argument");
  std::string err;
  ProfScope p("encode", (hipStream_t)stream);
  int rc = mb::enc_encode(d, img_nchw, indices, zq, zraw, B, (hipStream_t)stream, err);
  if (rc) return fail(rc, "mb_enc_encode: %s", err.c_str());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

// ================================================================================================
// whole loop (sampling.py:55-136)
// ================================================================================================
int mb_sample(mb_gen* g, mb_dec* d, const mb_sample_plan* plan, const int64_t* labels, int B, const float* exp_noise,
              const float* conf_noise, int64_t* step_tokens, int64_t* tokens_out, float* img_nchw,
              uint8_t* img_nhwc_u8, mb_stream stream) {
  if (!g || !plan || !labels || !exp_noise || !conf_noise) return fail(-1, "mb_sample: null argument");
  if (!plan->scale || !plan->temperature || !plan->mask_len || plan->num_steps <= 0) return fail(-1, "mb_sample: incomplete plan");
  const int nbf = plan->use_guidance ? 2 * B : B;
  if (B <= 0 || nbf > g->max_seqs) return fail(-1, "mb_sample: B=%d needs %d sequences, engine holds %d", B, nbf, g->max_seqs);
  if (!d && (img_nchw || img_nhwc_u8)) return fail(-1, "mb_sample: image requested without a decoder");
  hipStream_t s = (hipStream_t)stream;
  const mb_gen_cfg& c = g->c;
  const int n = c.seq, m = c.splits, C = g->C;
  const size_t P = (size_t)n * m;
  // state init (sampling.py:65-71): every position masked; CFG batch = [cond | label-dropped]
  // A run may be fed in step chunks (plan->step_begin / step_end: the noise of a whole 256-step run at batch 100 is 6.7 GB): chunk [0, e) starts from
  // the all-masked state, later chunks continue from the token state the engine kept; exp_noise / conf_noise / step_tokens hold THIS chunk's steps.
  const int s0 = plan->step_end > 0 ? plan->step_begin : 0, s1 = plan->step_end > 0 ? plan->step_end : plan->num_steps;
  if (s0 < 0 || s1 > plan->num_steps || s0 >= s1) return fail(-1, "mb_sample: step chunk [%d, %d) outside [0, %d)", s0, s1, plan->num_steps);
  // A run fed in chunks keeps its token state in the engine: a chunk is accepted only as the exact continuation of the run in progress (same batch,
  // plan length and guidance flag, beginning where the previous chunk ended).  The handle is not re-entrant while a run is in progress.
  if (s0 == 0) { mb::fill_i64(s, g->tok_a, (int64_t)C, (size_t)B * P); g->loop_B = B; g->loop_steps = plan->num_steps; g->loop_guided = plan->use_guidance != 0; }
  else if (g->loop_next != s0 || g->loop_B != B || g->loop_steps != plan->num_steps || g->loop_guided != (plan->use_guidance != 0))
    return fail(-1, "mb_sample: step chunk [%d, %d) of a %d-step run with B = %d does not continue the run in progress (next step %d of %d, B = %d)",
                s0, s1, plan->num_steps, B, g->loop_next, g->loop_steps, g->loop_B);
  g->loop_next = -1;                                   // (set again below when this chunk has been enqueued and more follow)
  int64_t* cur = (s0 & 1) ? g->tok_b : g->tok_a;
  int64_t* nxt = (s0 & 1) ? g->tok_a : g->tok_b;
  int64_t* last_pred = g->pred;
  g->cfg_labels_ready = nullptr;
  for (int i = s0; i < s1; ++i) {
    const float* lc = g->logits;
    const float* lu = nullptr;
    int rc;
    // sampling.py:98-99 combines c + s_i (c - u).  Where the annealed scale s_i is exactly 0 -- the first steps of the cosine schedule: (i / N)^p pi
    // is below float32's cos() resolution -- the unconditional logits do not enter the result (c + 0 (c - u) == c for finite logits), so that
    // forward is not run: the step is the plain conditional forward, bit for bit what the guided expression evaluates to.
    // (precision 4: the guided forward is also the MORE PRECISE conditional forward -- its pair tiles carry the activation-lo sets, the plain tiles do not --
    // and the first, almost fully masked steps are where near-ties flip: the zero-scale steps run it too; its unconditional half is then multiplied by 0)
    if (plan->use_guidance && (plan->scale[i] != 0.0f || (g->pair_ok && c.precision >= 4))) {
      rc = gen_forward_cfg(g, cur, labels, g->logits, B, s);
      lu = g->logits + (size_t)B * P * C;
      if (B <= g->chunk_seqs / 2) { g->cfg_labels_ready = labels; g->cfg_ready_B = B; }   // lab_cfg / drop_cfg stay valid for the rest of this call
    } else {
      rc = gen_forward(g, cur, labels, nullptr, g->logits, B, s);
    }
    if (rc) { g->cfg_labels_ready = nullptr; return rc; }
    const size_t k = (size_t)(i - s0);                 // the noise / step_tokens buffers hold this chunk's steps
    int64_t* pred = step_tokens ? step_tokens + k * B * P : g->pred;
    rc = mb_sample_step(lc, lu, plan->scale[i], plan->temperature[i], exp_noise + k * B * P * C,
                        conf_noise + k * B * P, plan->mask_len[i], cur, nxt, pred, B, n, m, C, stream);
    if (rc) { g->cfg_labels_ready = nullptr; return rc; }
    last_pred = pred;
    int64_t* t = cur; cur = nxt; nxt = t;
  }
  g->cfg_labels_ready = nullptr;
  if (s1 < plan->num_steps) {                          // more chunks follow: keep the last predictions only if they are the engine's own buffer
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
    g->loop_next = s1;
    return 0;
  }
  // combine_factorized_tokens (factorization.py:7-24) on the LAST step's predictions, kept as integers
  int64_t* codes = tokens_out ? tokens_out : g->codes;
  mb::combine_groups(s, last_pred, codes, (size_t)B * n, m, g->gbits);
  if (d) {
    int rc = mb_dec_decode(d, codes, img_nchw, img_nhwc_u8, B, stream);
    if (rc) return rc;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-11, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"
