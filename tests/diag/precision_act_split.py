"""Which 16-bit operands have to go to get under 1e-3?  CPU emulation (precision_study2.fwd) with the activation roundings removed
(= what fp16 hi+lo activation pairs would give) and the weights left in single fp16, and the other way round.  (Test infrastructure.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import maskbit_oracle as O
from precision_study2 import fwd, h16, ident

def main():
    torch.set_num_threads(16)
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
    B, N = 4, 8
    y = torch.tensor([1, 7, 282, 604]); rec = []
    torch.manual_seed(4321)
    O.sample_loop(lambda t, yy, dd: O.lfq_bert_forward(sd, cfg, t, yy, dd), B, y, num_steps=N, guidance_scale=7.1, guidance_annealing="cosine",
                  scale_pow=3.0, randomize_temperature=8.2, mask_schedule_strategy="arccos", mask_token=64, codebook_splits=2, record=rec)
    drop = torch.cat([torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)])
    w16 = {k: h16(v) for k, v in sd.items() if v.dim() == 2}
    w32 = {k: v for k, v in sd.items() if v.dim() == 2}
    allh = {c: h16 for c in ["x", "qkv", "p", "att", "h"]}
    cases = [("all fp16", allh, w16), ("fp16x2 weights", allh, w32),
             ("split x, h, att (fp16 W)", dict(allh, x=ident, h=ident, att=ident), w16),
             ("split x, h (fp16 W)", dict(allh, x=ident, h=ident), w16),
             ("split x only (fp16 W)", dict(allh, x=ident), w16),
             ("split x, h, att + fp16x2 W", dict(allh, x=ident, h=ident, att=ident), w32)]
    for name, q, wq in cases:
        tm = tn = 0; errs = []
        for r in rec:
            lg = fwd(sd, cfg, torch.cat([r.tokens_in, r.tokens_in]), torch.cat([y, y]), drop, q, wq)
            lc, lu = lg[:B], lg[B:]
            pred, _ = O.sample_step(lc, lu, r.scale, 1.0, r.exp_noise, r.conf_noise, r.tokens_in, 64, torch.tensor(r.mask_ratio), 512)
            msk = r.tokens_in == 64
            tm += int((pred != r.pred)[msk].sum()); tn += int(msk.sum()); errs.append(float((lc - r.logits_c).abs().mean()))
        print(f"{name:28s}: mismatch {tm}/{tn} = {tm / tn:.5f}; mean |logit err| {sum(errs) / len(errs):.5f}", flush=True)

if __name__ == "__main__":
    main()
