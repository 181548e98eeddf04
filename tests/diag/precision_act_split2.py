"""Refinement of precision_act_split.py: which consumer of the LayerNorm output x needs the hi+lo pair -- the QKV GEMM, the FFN-up GEMM, the head?
And how much do the attention output (out-proj input) and the FFN hidden (FFN-down input) add?  (Test infrastructure: CPU emulation.)"""
import math, os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import maskbit_oracle as O
h16 = lambda x: x.to(torch.float16).to(torch.float32)
ident = lambda x: x

def fwd(sd, cfg, tokens, labels, drop, q, wq):
    b = tokens.shape[0]
    lab = torch.where(drop.bool(), torch.full_like(labels, cfg.nclass), labels)
    x_tok = F.linear(O.token_bit_vectors(tokens, cfg), sd["input_proj.weight"], sd["input_proj.bias"])
    x = torch.cat([x_tok, sd["class_emb.weight"][lab].unsqueeze(1)], 1) + sd["pos_emb"]
    x = O._ln(x, sd, "first_layer.0", 1e-12)
    d, H = cfg.hidden, cfg.heads; dh = d // H
    for l in range(cfg.depth):
        a, f = f"transformer.layers.{l}.0", f"transformer.layers.{l}.1"
        qkv = h16(F.linear(q["x_qkv"](x), wq[a + ".mha.in_proj_weight"], sd[a + ".mha.in_proj_bias"]))
        qq, kk, vv = [t.reshape(b, -1, H, dh).transpose(1, 2) for t in qkv.split(d, -1)]
        s = (qq @ kk.transpose(-1, -2)) * (1 / math.sqrt(dh))
        p = torch.exp(s - s.amax(-1, keepdim=True)); den = p.sum(-1, keepdim=True)
        o = q["att"]((h16(p) @ vv) / den).transpose(1, 2).reshape(b, -1, d)
        x = O._ln(F.linear(o, wq[a + ".mha.out_proj.weight"], sd[a + ".mha.out_proj.bias"]) + x, sd, a + ".norm", 1e-12)
        h = q["h"](F.gelu(F.linear(q["x_ffn"](x), wq[f + ".net.0.weight"], sd[f + ".net.0.bias"])))
        x = O._ln(F.linear(h, wq[f + ".net.2.weight"], sd[f + ".net.2.bias"]) + x, sd, f + ".norm", 1e-12)
    y = F.gelu(F.linear(q["x_head"](x), wq["last_layer.0.weight"], sd["last_layer.0.bias"]))
    y = O._ln(y, sd, "last_layer.2", 1e-12)
    lg = F.linear(q["x_head"](y), wq["prediction_layer.weight"], sd["prediction_layer.bias"])
    return lg.reshape(b, cfg.seq + 1, cfg.splits, cfg.group_codes)[:, :cfg.seq]

def main():
    torch.set_num_threads(16)
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
    B, N = 6, 8
    y = torch.tensor([1, 7, 282, 604, 724, 179]); rec = []
    torch.manual_seed(4321)
    O.sample_loop(lambda t, yy, dd: O.lfq_bert_forward(sd, cfg, t, yy, dd), B, y, num_steps=N, guidance_scale=7.1, guidance_annealing="cosine",
                  scale_pow=3.0, randomize_temperature=8.2, mask_schedule_strategy="arccos", mask_token=64, codebook_splits=2, record=rec)
    drop = torch.cat([torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)])
    w16 = {k: h16(v) for k, v in sd.items() if v.dim() == 2}
    base = {c: h16 for c in ["x_qkv", "x_ffn", "x_head", "att", "h"]}
    cases = [("all fp16", base), ("split x -> QKV", dict(base, x_qkv=ident)), ("split x -> FFN-up", dict(base, x_ffn=ident)),
             ("split x -> head", dict(base, x_head=ident)), ("split x -> QKV, FFN-up", dict(base, x_qkv=ident, x_ffn=ident)),
             ("split x -> all three", dict(base, x_qkv=ident, x_ffn=ident, x_head=ident)),
             ("split x (all) + att", dict(base, x_qkv=ident, x_ffn=ident, x_head=ident, att=ident))]
    for name, q in cases:
        tm = tn = 0; errs = []
        for r in rec:
            lg = fwd(sd, cfg, torch.cat([r.tokens_in, r.tokens_in]), torch.cat([y, y]), drop, q, w16)
            lc, lu = lg[:B], lg[B:]
            pred, _ = O.sample_step(lc, lu, r.scale, 1.0, r.exp_noise, r.conf_noise, r.tokens_in, 64, torch.tensor(r.mask_ratio), 512)
            msk = r.tokens_in == 64
            tm += int((pred != r.pred)[msk].sum()); tn += int(msk.sum()); errs.append(float((lc - r.logits_c).abs().mean()))
        print(f"{name:26s}: mismatch {tm}/{tn} = {tm / tn:.5f}; mean |logit err| {sum(errs) / len(errs):.5f}", flush=True)

if __name__ == "__main__":
    main()
