"""Token mismatch against the first trunk layer that carries the weight-correction pass (LFQBert.wcorr_from; 0 = every layer = "precise",
depth = none = the plain differential form), on the three full-size 12-bit reference runs.
usage: python tools/wfrom_parity.py [wcorr_from values ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskbit_amd import parity_replay as R
vals = [int(a) for a in sys.argv[1:]] or [24, 18, 12, 6, 0]
tot_bad = {v: 0 for v in vals}; tot_pos = 0
for name in ("sample_full12_64", R.RUN_C3_S2, R.RUN_C3_S3):
    g = R.load_run(name)
    gen, _ = R.build_models("cuda", with_tokenizer=False, name=name)
    noise = R.reference_noise(g, gen.device)
    gen.precision = 2
    for wf in vals:
        gen.wcorr_from = wf
        bad, tot, per, _ = R.teacher_forced(gen, g, noise)
        tot_bad[wf] += bad
        print(f"{name:22s} wcorr_from {wf:2d}: {bad}/{tot} = {bad / tot:.2e}", flush=True)
    tot_pos += tot
    del gen; torch.cuda.empty_cache()
for wf in vals:
    print(f"== pooled wcorr_from {wf:2d}: {tot_bad[wf]}/{tot_pos} = {tot_bad[wf] / tot_pos:.2e}")
