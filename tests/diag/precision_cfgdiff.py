"""CPU emulation: "differential" CFG operands.  With classifier-free guidance the sampled logits are c + s (c - u); fp16 rounding of a GEMM
operand is independent in the conditional and the unconditional stream, so (c - u) carries both errors and s amplifies them.  If the
unconditional stream's operand is represented as  fp16(x_c) + fp16(x_u - x_c)  (its GEMM = the conditional GEMM + a GEMM over the fp16
difference), the rounding error of x_c is COMMON to both streams and cancels in (c - u); only the (much smaller) difference is rounded.
Cost on the engine: nothing extra -- the difference rows replace the unconditional rows in the same GEMM.
Cases: all fp16 (independent roundings) / x, h, att kept in fp32 (what the strict hi+lo mode approximates) / differential operands."""
import math, os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import maskbit_oracle as O
h16 = lambda x: x.to(torch.float16).to(torch.float32)
ident = lambda x: x


def diff16(x):                      # batch = [cond | uncond]: uncond operand = fp16(cond) + fp16(uncond - cond)
    b = x.shape[0] // 2
    xc = h16(x[:b])
    return torch.cat([xc, xc + h16(x[b:] - x[:b])])


def fwd(sd, cfg, tokens, labels, drop, q, wq):
    b = tokens.shape[0]
    lab = torch.where(drop.bool(), torch.full_like(labels, cfg.nclass), labels)
    x_tok = F.linear(O.token_bit_vectors(tokens, cfg), sd["input_proj.weight"], sd["input_proj.bias"])
    x = torch.cat([x_tok, sd["class_emb.weight"][lab].unsqueeze(1)], 1) + sd["pos_emb"]
    x = O._ln(x, sd, "first_layer.0", 1e-12)
    d, H = cfg.hidden, cfg.heads; dh = d // H
    rel = []
    for l in range(cfg.depth):
        a, f = f"transformer.layers.{l}.0", f"transformer.layers.{l}.1"
        rel.append(float((x[b // 2:] - x[:b // 2]).abs().mean() / x[:b // 2].abs().mean()))
        qkv = q["qkv"](F.linear(q["x"](x), wq[a + ".mha.in_proj_weight"], sd[a + ".mha.in_proj_bias"]))
        qq, kk, vv = [t.reshape(b, -1, H, dh).transpose(1, 2) for t in qkv.split(d, -1)]
        s = (qq @ kk.transpose(-1, -2)) * (1 / math.sqrt(dh))
        p = torch.exp(s - s.amax(-1, keepdim=True)); den = p.sum(-1, keepdim=True)
        o = q["att"](((q["p"](p) @ vv) / den).transpose(1, 2).reshape(b, -1, d))
        x = O._ln(F.linear(o, wq[a + ".mha.out_proj.weight"], sd[a + ".mha.out_proj.bias"]) + x, sd, a + ".norm", 1e-12)
        h = q["h"](F.gelu(F.linear(q["x"](x), wq[f + ".net.0.weight"], sd[f + ".net.0.bias"])))
        x = O._ln(F.linear(h, wq[f + ".net.2.weight"], sd[f + ".net.2.bias"]) + x, sd, f + ".norm", 1e-12)
    y = F.gelu(F.linear(q["head"](x), wq["last_layer.0.weight"], sd["last_layer.0.bias"]))
    y = O._ln(y, sd, "last_layer.2", 1e-12)
    lg = F.linear(q["head"](y), wq["prediction_layer.weight"], sd["prediction_layer.bias"])
    return lg.reshape(b, cfg.seq + 1, cfg.splits, cfg.group_codes)[:, :cfg.seq], rel


def main():
    torch.set_num_threads(8)
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
    B, N = int(os.environ.get("B", "6")), 8
    y = torch.tensor([1, 7, 282, 604, 724, 179, 751, 404][:B]); rec = []
    torch.manual_seed(4321)
    O.sample_loop(lambda t, yy, dd: O.lfq_bert_forward(sd, cfg, t, yy, dd), B, y, num_steps=N, guidance_scale=7.1, guidance_annealing="cosine",
                  scale_pow=3.0, randomize_temperature=8.2, mask_schedule_strategy="arccos", mask_token=64, codebook_splits=2, record=rec)
    drop = torch.cat([torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)])
    w16 = {k: h16(v) for k, v in sd.items() if v.dim() == 2}
    comps = ["x", "qkv", "p", "att", "h", "head"]
    base = {c: h16 for c in comps}
    base["head"] = ident                               # the engine feeds the two head GEMMs hi + lo pairs in every mode
    cases = [("all fp16 (engine default)", dict(base)),
             ("fp32 x, h, att (strict ideal)", {**base, "x": ident, "h": ident, "att": ident}),
             ("differential x, h", {**base, "x": diff16, "h": diff16}),
             ("differential x, h, att", {**base, "x": diff16, "h": diff16, "att": diff16}),
             ("differential x, h, att, qkv", {**base, "x": diff16, "h": diff16, "att": diff16, "qkv": diff16})]
    use = [r for r in rec if r.scale > 0.5]
    print("steps used: scales", [round(r.scale, 2) for r in use], "masked", [int((r.tokens_in == 64).sum()) for r in use])
    for name, q in cases:
        tm = tn = 0; e_c, e_l = [], []
        for r in use:
            lg, rel = fwd(sd, cfg, torch.cat([r.tokens_in, r.tokens_in]), torch.cat([y, y]), drop, q, w16)
            lc, lu = lg[:B], lg[B:]
            pred, _ = O.sample_step(lc, lu, r.scale, 1.0, r.exp_noise, r.conf_noise, r.tokens_in, 64, torch.tensor(r.mask_ratio), 512)
            msk = r.tokens_in == 64
            tm += int((pred != r.pred)[msk].sum()); tn += int(msk.sum())
            L = lc + r.scale * (lc - lu); Lr = r.logits_c + r.scale * (r.logits_c - r.logits_u)
            e_c.append(float((lc - r.logits_c).abs().mean())); e_l.append(float((L - Lr).abs().mean()))
        print(f"{name:32s}: mismatch {tm}/{tn} = {tm / tn:.5f}; mean |err| cond logits {sum(e_c) / len(e_c):.5f}, guided logits {sum(e_l) / len(e_l):.5f}", flush=True)
    print("mean |x_u - x_c| / mean |x_c| per layer input (last forward):", [round(v, 3) for v in rel[::4]])


if __name__ == "__main__":
    main()
