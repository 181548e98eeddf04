"""Logit error of the HIP forward vs the fp32 oracle for weight_split = 0 / 1 (full 12-bit model, random masked tokens)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import maskbit_oracle as O
from hip_helpers import hip_generator
cfg = O.GenCfg(bits=12, splits=2)
sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
torch.manual_seed(0)
nb = 8
tok = torch.randint(0, 64, (nb, 256, 2)); tok[torch.rand(nb, 256, 2) < 0.6] = 64
y = torch.randint(0, 1000, (nb,)); drop = torch.zeros(nb, dtype=torch.bool); drop[nb // 2:] = True
torch.set_num_threads(16)
ref = O.lfq_bert_forward(sd, cfg, tok, y, drop)
ref64 = None
m = hip_generator(cfg, sd)
for split in (0, 1):
    m.weight_split = split
    lg = m(tok.cuda(), y.cuda(), drop.cuda()).cpu()
    e = lg - ref
    print(f"weight_split={split}: mean|err| {float(e.abs().mean()):.5f}  rms {float(e.pow(2).mean().sqrt()):.5f}  max {float(e.abs().max()):.4f}  "
          f"rel-Frobenius {float(e.norm() / ref.norm()):.2e}", flush=True)
