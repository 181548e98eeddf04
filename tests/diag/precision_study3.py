"""Would folding LayerNorm into the consumer GEMM (fp16(y) . (W*gamma)^T, normalised in the epilogue from row statistics) cost
precision against the current path (fp16(LayerNorm(y)) . W^T)?  CPU emulation on the full 12-bit model, 4 CFG steps."""
import math, sys, os, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import maskbit_oracle as O
h16 = lambda x: x.to(torch.float16).to(torch.float32)

def ln_stats(y, eps=1e-12):
    mu = y.mean(-1, keepdim=True); var = ((y - mu) ** 2).mean(-1, keepdim=True)
    return mu, torch.rsqrt(var + eps)

def consumer(y, g, b, W, bias, fold):
    """LayerNorm(y; g, b) then Linear(W, bias) with fp16 GEMM operands, two ways"""
    mu, r = ln_stats(y)
    if not fold:
        x = (y - mu) * r * g + b
        return F.linear(h16(x), h16(W), bias), x
    Wg = h16(W * g)                                        # fold gamma into the weight, round once
    acc = F.linear(h16(y), Wg)                             # un-normalised fp16 activations
    s = Wg.sum(1)
    out = r * (acc - mu * s) + (bias + F.linear(b, W))
    return out, (y - mu) * r * g + b

def fwd(sd, cfg, tokens, labels, drop, fold):
    b = tokens.shape[0]
    lab = torch.where(drop.bool(), torch.full_like(labels, cfg.nclass), labels)
    x_tok = F.linear(O.token_bit_vectors(tokens, cfg), sd["input_proj.weight"], sd["input_proj.bias"])
    y = torch.cat([x_tok, sd["class_emb.weight"][lab].unsqueeze(1)], 1) + sd["pos_emb"]
    g, be = sd["first_layer.0.weight"], sd["first_layer.0.bias"]
    d, H = cfg.hidden, cfg.heads; dh = d // H
    for l in range(cfg.depth):
        a, f = f"transformer.layers.{l}.0", f"transformer.layers.{l}.1"
        qkv, x = consumer(y, g, be, sd[a + ".mha.in_proj_weight"], sd[a + ".mha.in_proj_bias"], fold)
        qkv = h16(qkv)
        qq, kk, vv = [t.reshape(b, -1, H, dh).transpose(1, 2) for t in qkv.split(d, -1)]
        s = (qq @ kk.transpose(-1, -2)) * (1 / math.sqrt(dh))
        p = torch.exp(s - s.amax(-1, keepdim=True)); den = p.sum(-1, keepdim=True)
        o = h16((h16(p) @ vv) / den).transpose(1, 2).reshape(b, -1, d)
        y = F.linear(o, h16(sd[a + ".mha.out_proj.weight"]), sd[a + ".mha.out_proj.bias"]) + x
        g, be = sd[a + ".norm.weight"], sd[a + ".norm.bias"]
        hpre, x = consumer(y, g, be, sd[f + ".net.0.weight"], sd[f + ".net.0.bias"], fold)
        h = h16(F.gelu(hpre))
        y = F.linear(h, h16(sd[f + ".net.2.weight"]), sd[f + ".net.2.bias"]) + x
        g, be = sd[f + ".norm.weight"], sd[f + ".norm.bias"]
    pre, _ = consumer(y, g, be, sd["last_layer.0.weight"], sd["last_layer.0.bias"], fold)
    yy = F.gelu(pre)
    mu, r = ln_stats(yy)
    yy = (yy - mu) * r * sd["last_layer.2.weight"] + sd["last_layer.2.bias"]
    lg = F.linear(h16(yy), h16(sd["prediction_layer.weight"]), sd["prediction_layer.bias"])
    return lg.reshape(b, cfg.seq + 1, cfg.splits, cfg.group_codes)[:, :cfg.seq]

def main():
    torch.set_num_threads(8)
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
    B, N = 4, 8
    y = torch.tensor([1, 7, 282, 604]); rec = []
    torch.manual_seed(4321)
    O.sample_loop(lambda t, yy, dd: O.lfq_bert_forward(sd, cfg, t, yy, dd), B, y, num_steps=N, guidance_scale=7.1, guidance_annealing="cosine",
                  scale_pow=3.0, randomize_temperature=8.2, mask_schedule_strategy="arccos", mask_token=64, codebook_splits=2, record=rec)
    drop = torch.cat([torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)])
    for name, fold in (("LayerNorm then fp16 (current)", False), ("fp16 then LayerNorm in the epilogue (folded)", True)):
        tm = tn = 0; errs = []
        for r in rec[::2]:
            lg = fwd(sd, cfg, torch.cat([r.tokens_in, r.tokens_in]), torch.cat([y, y]), drop, fold)
            lc, lu = lg[:B], lg[B:]
            pred, _ = O.sample_step(lc, lu, r.scale, 1.0, r.exp_noise, r.conf_noise, r.tokens_in, 64, torch.tensor(r.mask_ratio), 512)
            msk = r.tokens_in == 64
            tm += int((pred != r.pred)[msk].sum()); tn += int(msk.sum()); errs.append(float((lc - r.logits_c).abs().mean()))
        print(f"{name:46s}: mismatch {tm}/{tn} = {tm / tn:.5f}; mean |logit err| {sum(errs) / len(errs):.5f}", flush=True)
if __name__ == "__main__":
    main()
