"""Host-side draw experiment (round 6): configs[1] at batch 16 through run_chunked with (a) the product's draw (a dedicated worker thread whose own intra-op
thread count is 1), (b) the draws on the calling thread with the process's default pool (128 threads on the GPU boxes), (c) rounds 3-5's scheme (thread count
flipped to 1 around each draw on the calling thread), (d) the whole process at one thread.  usage: python tools/host_draw_ab.py [batch]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import parity_replay as PR, sampling as S
from maskbit_amd.sampling import build_plan, run_chunked
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
g = PR.load_run(PR.RUN_CFG1)
gen, _ = PR.build_models(dev, with_tokenizer=False, name=PR.RUN_CFG1)
kw = g["kw"]
plan = build_plan(int(kw["num_steps"]), 512, 0.0, "none", 4.0, 1.0, False, kw["mask_schedule_strategy"])
labels = (torch.arange(B) * 37 % 1000).to(dev)
rt = float(kw["randomize_temperature"])


def bench(tag, n=8):
    run_chunked(gen, None, labels, plan, rt, want_steps=False, want_image=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): run_chunked(gen, None, labels, plan, rt, want_steps=False, want_image=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{tag:60s}: {dt * 1e3:7.1f} ms per batch of {B} = {B / dt:6.1f} images/s   (torch threads {torch.get_num_threads()})", flush=True)


print("host cpus", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
bench("(a) product: draws on the single-thread worker")
run_on_worker = S._DrawThread.run
S._DrawThread.run = classmethod(lambda cls, fn: fn())
bench("(b) draws on the calling thread, default pool")


def flipped(cls, fn):
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return fn()
    finally:
        torch.set_num_threads(n)
S._DrawThread.run = classmethod(flipped)
bench("(c) thread count flipped to 1 around each draw (rounds 3-5)")
S._DrawThread.run = run_on_worker
bench("(a) again")
n0 = torch.get_num_threads()
torch.set_num_threads(1)
bench("(d) the whole process at one intra-op thread")
torch.set_num_threads(n0)
bench("(a) again, pool restored")
# where does the host spend its time per chunk?
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
run_chunked(gen, None, labels, plan, rt, want_steps=False, want_image=False); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
