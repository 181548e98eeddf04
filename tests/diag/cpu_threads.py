"""How does the CPU oracle scale with torch threads on this host? (informs bench.py's cpu_baseline)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import maskbit_oracle as O
cfg = O.GenCfg(bits=12, splits=2)
sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
t = torch.randint(0, 65, (8, 256, 2)); y = torch.arange(8)
for n in (16, 32, 64, 128):
    torch.set_num_threads(n)
    O.lfq_bert_forward(sd, cfg, t[:2], y[:2], None)
    t0 = time.perf_counter(); O.lfq_bert_forward(sd, cfg, t, y, None); dt = time.perf_counter() - t0
    print(f"threads={n}: 8-sequence forward {dt:.2f}s -> {dt/8:.3f} s/seq", flush=True)
