"""Token mismatch and guided-forward time against WHICH trunk GEMMs of WHICH layers carry the activation-lo mini-tile set (precision 4 engine narrowed by
LFQBert.alo_mask / alo_from -> mb_gen_set_alo: 1 QKV, 2 out-proj, 4 FFN-up, 8 FFN-down) on full-size runs of the reference.
usage: [ALO_RUNS=name,name,...] python tools/alo_mask_parity.py [mask[:from_layer] ...]      (default runs: the four 14-bit / 256-step ones)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskbit_amd import parity_replay as R

cases = [(int(a.split(":")[0]), int(a.split(":")[1]) if ":" in a else 0) for a in sys.argv[1:]] or [(15, 0), (14, 0), (10, 0), (6, 0), (4, 12), (0, 0)]
runs = os.environ.get("ALO_RUNS", ",".join([R.RUN_CFG5, R.RUN_CFG5_S2, R.RUN_CFG5_S3, R.RUN_CFG5_S4])).split(",")
tot = {c: [0, 0] for c in cases}
times = {}
for name in runs:
    g = R.load_run(name)
    gen, _ = R.build_models("cuda", with_tokenizer=False, name=name)
    gen.precision = 4
    noise = R.reference_noise(g, gen.device)
    for c in cases:
        gen.alo_mask, gen.alo_from = c
        bad, n, per, _ = R.teacher_forced(gen, g, noise)
        tot[c][0] += bad; tot[c][1] += n
        if name == runs[0]:
            B = 64
            t = torch.full((B, 256, 2), g["C"], device="cuda"); y = torch.arange(B, device="cuda")
            for _ in range(2): gen.forward_cfg(t, y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): gen.forward_cfg(t, y)
            torch.cuda.synchronize(); times[c] = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{name:24s} mask {c[0]:2d} from layer {c[1]:2d}: {bad}/{n} = {bad / n:.2e}", flush=True)
    del gen; torch.cuda.empty_cache()
for c in cases:
    b, n = tot[c]
    m = c[0]
    print(f"== mask {m:2d} (QKV {m & 1}, out {m >> 1 & 1}, up {m >> 2 & 1}, down {m >> 3 & 1}) from layer {c[1]:2d}: {b}/{n} = {b / n:.2e}   guided forward of 64 pairs {times.get(c, 0):.2f} ms")
