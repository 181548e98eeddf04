"""Experiment (round 6, review item 6): BASELINE configs[1] (10-bit, 16 steps, no guidance) at batch 16 as ONE run of 16 samples against TWO concurrent runs of
8 samples on two HIP streams (two engine handles, two host threads) -- does co-scheduling two half batches fill what a single batch of 16 leaves idle
(QKV: 192 of 256 CUs; every launch a single tile round)?  usage: python tools/two_streams.py [batch]"""
import os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import parity_replay as PR
from maskbit_amd.sampling import build_plan, run_chunked
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
g = PR.load_run(PR.RUN_CFG1)
kw = g["kw"]
plan = build_plan(int(kw["num_steps"]), 512, 0.0, "none", 4.0, 1.0, False, kw["mask_schedule_strategy"])
rt = float(kw["randomize_temperature"])
gens = [PR.build_models(dev, with_tokenizer=False, name=PR.RUN_CFG1)[0] for _ in range(2)]
N = 8


def runs(gen, b, stream, n):
    labels = (torch.arange(b) * 37 % 1000).to(dev)
    with torch.cuda.stream(stream):
        for _ in range(n):
            run_chunked(gen, None, labels, plan, rt, want_steps=False, want_image=False)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
for rnd in range(3):
    t_one = timed(lambda: runs(gens[0], B, s0, N))

    def both():
        th = [threading.Thread(target=runs, args=(gens[i], B // 2, (s0, s1)[i], N)) for i in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
    t_two = timed(both)
    t_half = timed(lambda: runs(gens[0], B // 2, s0, N))
    print(f"round {rnd}: one run of {B}: {t_one / N * 1e3:6.1f} ms = {B * N / t_one:6.1f} images/s | two concurrent runs of {B // 2}: {t_two / N * 1e3:6.1f} ms = {B * N / t_two:6.1f} images/s"
          f" | one run of {B // 2} alone: {t_half / N * 1e3:6.1f} ms = {B // 2 * N / t_half:6.1f} images/s", flush=True)
