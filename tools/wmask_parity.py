"""Token mismatch and guided-forward time against WHICH trunk GEMMs carry the correction mini-tiles (LFQBert.wcorr_mask: 1 QKV, 2 out-proj, 4 FFN-up,
8 FFN-down) on the full-size 12-bit runs of the reference.  usage: python tools/wmask_parity.py [masks ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskbit_amd import parity_replay as R

masks = [int(a) for a in sys.argv[1:]] or [15, 5, 13, 7, 12, 0]
runs = os.environ.get("WMASK_RUNS", "sample_full12_64,sample_full12_64_s2,sample_full12_64_s3").split(",")
tot = {m: [0, 0] for m in masks}
times = {}
for name in runs:
    g = R.load_run(name)
    gen, _ = R.build_models("cuda", with_tokenizer=False, name=name)
    noise = R.reference_noise(g, gen.device)
    for m in masks:
        gen.wcorr_mask = m
        bad, n, per, _ = R.teacher_forced(gen, g, noise)
        tot[m][0] += bad; tot[m][1] += n
        if name == runs[0]:
            B = 64
            t = torch.full((B, 256, 2), g["C"], device="cuda"); y = torch.arange(B, device="cuda")
            for _ in range(2): gen.forward_cfg(t, y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): gen.forward_cfg(t, y)
            torch.cuda.synchronize(); times[m] = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{name:24s} mask {m:2d}: {bad}/{n} = {bad / n:.2e}", flush=True)
    del gen; torch.cuda.empty_cache()
for m in masks:
    b, n = tot[m]
    print(f"== mask {m:2d} (QKV {m & 1}, out {m >> 1 & 1}, up {m >> 2 & 1}, down {m >> 3 & 1}): {b}/{n} = {b / n:.2e}   guided forward of 64 pairs {times.get(m, 0):.2f} ms")
