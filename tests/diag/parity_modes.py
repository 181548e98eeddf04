"""Teacher-forced token mismatch of every engine precision mode against the reference's own full-size runs (tests/golden/sample_full*.npz).
usage: python tests/diag/parity_modes.py [run names ...]     (default: all three BASELINE runs)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_replay as R

MODES = [("default (-1,-1)", -1, -1, 0), ("single fp16 (0,0)", 0, 0, 0), ("differential only (0,1)", 0, 1, 0), ("precise: differential + W (3,2)", 3, 2, 0),
         ("hi+lo e4m3 (3,0)", 3, 0, 0), ("hi+lo fp4 x (4,0)", 4, 0, 0), ("fp16x2 weights (0,0,ws)", 0, 0, 1), ("differential + fp16x2 weights", 0, 1, 1)]
if os.environ.get("PM_MODES"):
    MODES = [m for i, m in enumerate(MODES) if str(i) in os.environ["PM_MODES"].split(",")]


def main():
    runs = sys.argv[1:] or ["sample_full12_64", R.RUN_CFG1, R.RUN_CFG5]
    for name in runs:
        g = R.load_run(name)
        gen, _ = R.build_models("cuda", with_tokenizer=False, name=name)
        noise = R.reference_noise(g, gen.device)
        nb = max(1, len(g["steps"]) // 8)
        for tag, act, pair, ws in MODES:
            gen.weight_split, gen.act_split, gen.cfg_pair = ws, act, pair
            t0 = time.time()
            bad, tot, per, _ = R.teacher_forced(gen, g, noise)
            print(f"{name:24s} {tag:26s}: {bad:4d}/{tot} = {bad / tot:.2e}   per eighth {[sum(per[i:i + nb]) for i in range(0, len(per), nb)]}  ({time.time() - t0:.1f}s)", flush=True)
        del gen
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
