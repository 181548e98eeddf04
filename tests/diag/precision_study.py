"""CPU emulation of the engine's dataflow with 16-bit storage (bf16 or fp16) to predict token mismatch.
Rounds exactly what the HIP engine rounds: GEMM operands (activations + weights), qkv, P, attention
output, FFN hidden; accumulates in fp32; residual stream and LayerNorm in fp32."""
import math, sys, time, os
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import maskbit_oracle as O

def make_q(kind):
    if kind == "fp32": return lambda x: x
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[kind]
    return lambda x: x.to(dt).to(torch.float32)

def fwd_emul(sd, cfg, tokens, labels, drop, q, qhead=None, wq=None):
    qhead = qhead or q
    W = lambda k: wq[k]
    b = tokens.shape[0]
    lab = torch.where(drop.bool(), torch.full_like(labels, cfg.nclass), labels)
    x_tok = F.linear(O.token_bit_vectors(tokens, cfg), sd["input_proj.weight"], sd["input_proj.bias"])
    x = torch.cat([x_tok, sd["class_emb.weight"][lab].unsqueeze(1)], 1) + sd["pos_emb"]
    x = O._ln(x, sd, "first_layer.0", 1e-12)
    d, H = cfg.hidden, cfg.heads; dh = d // H
    for l in range(cfg.depth):
        a, f = f"transformer.layers.{l}.0", f"transformer.layers.{l}.1"
        qkv = q(F.linear(q(x), W(a + ".mha.in_proj_weight"), sd[a + ".mha.in_proj_bias"]))
        qq, kk, vv = [t.reshape(b, -1, H, dh).transpose(1, 2) for t in qkv.split(d, -1)]
        s = (qq @ kk.transpose(-1, -2)) * (1 / math.sqrt(dh))
        p = torch.exp(s - s.amax(-1, keepdim=True)); den = p.sum(-1, keepdim=True)
        o = q((q(p) @ vv) / den).transpose(1, 2).reshape(b, -1, d)
        x = O._ln(F.linear(o, W(a + ".mha.out_proj.weight"), sd[a + ".mha.out_proj.bias"]) + x, sd, a + ".norm", 1e-12)
        h = q(F.gelu(F.linear(q(x), W(f + ".net.0.weight"), sd[f + ".net.0.bias"])))
        x = O._ln(F.linear(h, W(f + ".net.2.weight"), sd[f + ".net.2.bias"]) + x, sd, f + ".norm", 1e-12)
    y = F.gelu(F.linear(qhead(x), wq["last_layer.0.weight@head"], sd["last_layer.0.bias"]))
    y = O._ln(y, sd, "last_layer.2", 1e-12)
    lg = F.linear(qhead(y), wq["prediction_layer.weight@head"], sd["prediction_layer.bias"])
    return lg.reshape(b, cfg.seq + 1, cfg.splits, cfg.group_codes)[:, :cfg.seq]

def main():
    torch.set_num_threads(8)
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
    B, N = 4, int(os.environ.get("NSTEPS", "8"))
    y = torch.tensor([1, 7, 282, 604])
    rec = []
    torch.manual_seed(4321)
    O.sample_loop(lambda t, yy, dd: O.lfq_bert_forward(sd, cfg, t, yy, dd), B, y, num_steps=N, guidance_scale=7.1, guidance_annealing="cosine",
                  scale_pow=3.0, randomize_temperature=8.2, mask_schedule_strategy="arccos", mask_token=64, codebook_splits=2, record=rec)
    drop = torch.cat([torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)])
    for kind, head in [("bf16", "bf16"), ("fp16", "fp16"), ("fp16", "fp32")]:
        q, qh = make_q(kind), make_q(head)
        wq = {k: q(v) for k, v in sd.items() if v.dim() == 2 and "transformer" in k}
        wq["last_layer.0.weight@head"] = qh(sd["last_layer.0.weight"]); wq["prediction_layer.weight@head"] = qh(sd["prediction_layer.weight"])
        tm = tn = 0; errs = []
        for r in rec:
            lg = fwd_emul(sd, cfg, torch.cat([r.tokens_in, r.tokens_in]), torch.cat([y, y]), drop, q, qh, wq)
            lc, lu = lg[:B], lg[B:]
            pred, _ = O.sample_step(lc, lu, r.scale, 1.0, r.exp_noise, r.conf_noise, r.tokens_in, 64, torch.tensor(r.mask_ratio), 512)
            msk = r.tokens_in == 64
            tm += int((pred != r.pred)[msk].sum()); tn += int(msk.sum())
            errs.append(float((lc - r.logits_c).abs().mean()))
        print(f"{kind:5s} head={head:5s}: mismatch {tm}/{tn} = {tm / tn:.5f}; mean |logit err| per step: " + " ".join(f"{e:.4f}" for e in errs), flush=True)

if __name__ == "__main__":
    main()
