"""Teacher-forced token mismatch of the product default and of the differential form without the weight-correction pass on EVERY recorded full-size run of the reference
(tests/golden/sample_full*.npz), with the pooled figure per BASELINE configuration.
usage: python tools/parity_all_runs.py [run names ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskbit_amd import parity_replay as R

GROUPS = {"configs[2] 12-bit / 64 steps / CFG 7.1": ["sample_full12_64", R.RUN_C3_S2, R.RUN_C3_S3],
          "configs[1] 10-bit / 16 steps / no CFG": [R.RUN_CFG1, R.RUN_CFG1_S2, R.RUN_CFG1_S3],
          "configs[4] 14-bit / 256 steps / CFG 5.8": [R.RUN_CFG5, R.RUN_CFG5_S2, R.RUN_CFG5_S3, R.RUN_CFG5_S4],
          "HELD-OUT run (recorded after round 5's coverage decisions): configs[2]": [R.RUN_C3_S4],
          "HELD-OUT run: configs[1]": [R.RUN_CFG1_S4],
          "HELD-OUT run: configs[4]": [R.RUN_CFG5_S5],
          "trained-like weights (heavy tails, massive-activation channels): configs[2]": [R.RUN_C3_OUTLIER, R.RUN_C3_OUTLIER_S2],
          "HELD-OUT trained-like run of a heavier family (round 6): configs[2]": [R.RUN_C3_OUTLIER2, R.RUN_C3_OUTLIER2_S2, R.RUN_C3_OUTLIER2_S3],
          "the demo's call site (round 6): 14-bit, guidance 3.0 with annealing none, 64 steps": [R.RUN_DEMO14],
          "trained-like weights: configs[1]": [R.RUN_CFG1_OUTLIER],
          "the other shipped codebooks (round 6): 16-bit / 18-bit, 64 steps, their own yaml's sampler": [R.RUN_16BIT, R.RUN_18BIT],
          "use_prenorm=True, configs[2]'s sampler": [R.RUN_C3_PRENORM],
          "1024 + 1 tokens (512 x 512 models), configs[2]'s sampler": [R.RUN_C3_SEQ1024]}
MODES = (("default", -1), ("differential only (precision 1)", 1))
if os.environ.get("PARITY_MODES"):            # e.g. PARITY_MODES="default:-1,fp16:0"   (tag:LFQBert.precision)
    MODES = tuple((t.split(":")[0], int(t.split(":")[1])) for t in os.environ["PARITY_MODES"].split(","))
only = set(sys.argv[1:])
for grp, names in GROUPS.items():
    pooled = {m[0]: [0, 0] for m in MODES}
    for name in names:
        if only and name not in only:
            continue
        if not os.path.exists(os.path.join(R.GOLDEN_DIR, name + ".npz")):
            continue
        g = R.load_run(name)
        gen, _ = R.build_models("cuda", with_tokenizer=False, name=name)
        noise = R.reference_noise(g, gen.device)
        for tag, prec in MODES:
            gen.precision = prec
            bad, tot, per, _ = R.teacher_forced(gen, g, noise)
            pooled[tag][0] += bad; pooled[tag][1] += tot
            S = len(per)
            print(f"{name:28s} {tag:34s} resolves to {gen.resolved_precision()}: {bad:4d}/{tot} = {bad / tot:.2e}   per eighth {[sum(per[i * S // 8:(i + 1) * S // 8]) for i in range(8)]}", flush=True)
        del gen, noise; torch.cuda.empty_cache()
    for tag, (b, t) in pooled.items():
        if t:
            print(f"== {grp}: {tag}: {b}/{t} = {b / t:.2e} (Poisson sigma {b ** 0.5 / t:.1e})", flush=True)
