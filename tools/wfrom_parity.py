import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskbit_amd import parity_replay as R
for name in ("sample_full12_64", R.RUN_C3_S2):
    g = R.load_run(name)
    gen, _ = R.build_models("cuda", with_tokenizer=False, name=name)
    noise = R.reference_noise(g, gen.device)
    for tag, pair, wf in (("default", -1, 0), ("precise, second half of the trunk", 2, 12), ("precise", 2, 0)):
        gen.cfg_pair, gen.wcorr_from = pair, wf
        bad, tot, per, _ = R.teacher_forced(gen, g, noise)
        print(f"{name:22s} {tag:36s}: {bad}/{tot} = {bad / tot:.2e}  per 8 steps {[sum(per[i:i + 8]) for i in range(0, 64, 8)]}", flush=True)
    del gen; torch.cuda.empty_cache()
