"""What would BASELINE configs[4]'s "fp8 MFMA attention" cost in token parity?  CPU emulation on the full-size 12-bit generator: everything as
the fp16 engine rounds it, and Q / K / V (per-head absmax-scaled) and / or the softmax numerator P stored as OCP e4m3 before the two attention
contractions.  Teacher-forced against the fp32 oracle, same protocol as precision_study2.py.  (Test infrastructure: imports oracle/.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import maskbit_oracle as O
from precision_study2 import fwd, h16

F8 = torch.float8_e4m3fn

def f8_heads(qkv, heads=16, dh=64):
    """[b, N, 3*d] -> fp16-rounded, then each of the 3 x heads column blocks scaled to the e4m3 range by its absmax and rounded to e4m3."""
    b, n, _ = qkv.shape
    t = h16(qkv).reshape(b, n, 3 * heads, dh)
    s = 448.0 / t.abs().amax(dim=(0, 1, 3), keepdim=True).clamp_min(1e-12)
    return ((t * s).to(F8).to(torch.float32) / s).reshape(b, n, -1)

def f8_p(p):
    return (p * 256.0).to(F8).to(torch.float32) / 256.0       # exp(s - max) in (0, 1]: scaled so that 2^-17 is still representable

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
    w16 = {k: h16(v) for k, v in sd.items() if v.dim() == 2}
    base = {c: h16 for c in ["x", "qkv", "p", "att", "h"]}
    cases = [("fp16 engine", base), ("e4m3 Q/K/V", dict(base, qkv=f8_heads)), ("e4m3 P", dict(base, p=f8_p)),
             ("e4m3 Q/K/V + P", dict(base, qkv=f8_heads, p=f8_p))]
    for name, q in cases:
        tm = tn = 0; errs = []
        for r in rec[::2]:
            lg = fwd(sd, cfg, torch.cat([r.tokens_in, r.tokens_in]), torch.cat([y, y]), drop, q, w16)
            lc, lu = lg[:B], lg[B:]
            pred, _ = O.sample_step(lc, lu, r.scale, 1.0, r.exp_noise, r.conf_noise, r.tokens_in, 64, torch.tensor(r.mask_ratio), 512)
            msk = r.tokens_in == 64
            tm += int((pred != r.pred)[msk].sum()); tn += int(msk.sum()); errs.append(float((lc - r.logits_c).abs().mean()))
        print(f"{name:16s}: mismatch {tm}/{tn} = {tm / tn:.5f}; mean |logit err| {sum(errs) / len(errs):.5f}", flush=True)

if __name__ == "__main__":
    main()
