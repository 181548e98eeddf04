"""Which part of the logit error can flip a sampled token?  A shift common to all codes of a (position, group) does not change the softmax; the
CENTERED error e - mean_codes(e) does.  Per engine mode, against the CPU oracle on the same tokens: mean |e|, rms of the centered error, and the
rms error of the top-1 / top-2 logit gap (the quantity whose sign decides a near-tie).
usage: python tests/diag/logit_error_modes.py [bits]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import maskbit_oracle as O
from hip_helpers import hip_generator

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = O.GenCfg(bits=bits, splits=2)
sd = O.make_generator_weights(cfg, seed=100 + (bits - 12) // 2 if bits != 10 else 101, head_gain=12.0)
m = hip_generator(cfg, sd)
g = torch.Generator().manual_seed(5)
B = 4
t = torch.randint(0, cfg.group_codes, (B, 256, 2), generator=g)
frac = torch.tensor([1.0, 0.7, 0.4, 0.15]).view(B, 1, 1)
t = torch.where(torch.rand(B, 256, 2, generator=g) < frac, torch.full_like(t, cfg.group_codes), t)
y = torch.tensor([3, 250, 600, 999])
torch.set_num_threads(min(16, torch.get_num_threads()))
ref = O.lfq_bert_forward(sd, cfg, t, y, torch.zeros(B, dtype=torch.bool))
msk = (t == cfg.group_codes)
top2 = ref.topk(2, dim=-1)
gap_ref = top2.values[..., 0] - top2.values[..., 1]
for tag, act, pair in [("single fp16", 0, 0), ("W mode (QKV, FFN-up weights)", 0, 2), ("exact qkv + w1 weights (host-rounded rest)", -3, 0), ("hi+lo fp16 (2)", 2, 0), ("hi+lo e4m3 (3)", 3, 0), ("fp16x2 weights", -2, 0)]:
    if act == -3:
        sd2 = {k: (v.half().float() if (v.dim() == 2 and k.startswith("transformer.layers.") and not ("in_proj_weight" in k or "net.0.weight" in k)) else v) for k, v in sd.items()}
        m = hip_generator(cfg, sd2)
        m.weight_split, m.act_split, m.cfg_pair = 1, 0, 0
    elif act == -2:
        m = hip_generator(cfg, sd)
        m.weight_split, m.act_split, m.cfg_pair = 1, 0, 0
    else:
        m.weight_split, m.act_split, m.cfg_pair = 0, act, pair
    lg = m(t.cuda(), y.cuda(), torch.zeros(B, dtype=torch.bool).cuda()).cpu()
    e = (lg - ref)[msk]
    ec = e - e.mean(-1, keepdim=True)
    gap = lg.gather(-1, top2.indices[..., :1]).squeeze(-1) - lg.gather(-1, top2.indices[..., 1:2]).squeeze(-1)
    print(f"{bits}-bit {tag:30s}: mean |e| {float(e.abs().mean()):.5f}  rms centered e {float(ec.pow(2).mean().sqrt()):.5f}  "
          f"rms common shift {float(e.mean(-1).pow(2).mean().sqrt()):.5f}  rms top-2 gap error {float((gap - gap_ref)[msk].pow(2).mean().sqrt()):.5f}", flush=True)
