"""Attention kernel time inside the engine forward (nb = 128) + logits checksum."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import maskbit_oracle as O
from hip_helpers import hip_generator
from maskbit_amd import _lib
cfg = O.GenCfg(bits=12, splits=2, depth=4)
sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
m = hip_generator(cfg, sd)
g = torch.Generator().manual_seed(1)
t = torch.randint(0, 65, (128, 256, 2), generator=g).cuda(); y = torch.randint(0, 1000, (128,), generator=g).cuda()
out = m(t, y); torch.cuda.synchronize()
_lib.prof_enable(True)
for _ in range(6): m(t, y)
torch.cuda.synchronize()
p = _lib.prof_read(); _lib.prof_enable(False)
c, ms = p["attention"]
print(f"attention {ms / c * 1e3:.1f} us; logits checksum {float(out.double().sum()):.4f}")
