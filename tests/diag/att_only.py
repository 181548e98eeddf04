"""Run a few generator forwards (target for rocprofv3 --pmc passes on attention_kernel)."""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import maskbit_oracle as O
from hip_helpers import hip_generator
cfg = O.GenCfg(bits=12, splits=2, depth=2)
sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
m = hip_generator(cfg, sd)
t = torch.randint(0, 65, (128, 256, 2), device="cuda"); y = torch.randint(0, 1000, (128,), device="cuda")
for _ in range(3): m(t, y)
torch.cuda.synchronize(); print("done")
