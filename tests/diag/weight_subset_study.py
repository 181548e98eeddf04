"""Which weights' fp16 rounding costs token parity?  The engine runs with fp16x2 (exact) weights + differential CFG operands, but a chosen subset of
the weights is pre-rounded to fp16 on the host (its lo half is then 0 = plain fp16 weights for that subset).  Teacher-forced mismatch against the
reference's own 64-step run per subset.   usage: python tests/diag/weight_subset_study.py [run name]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_replay as R
from maskbit_amd import LFQBert, synth

name = sys.argv[1] if len(sys.argv) > 1 else "sample_full12_64"
g = R.load_run(name)
z = g["z"]
bits = g["bits"]
sd0 = synth.make_generator_weights(synth.GenCfg(bits=bits, splits=2), seed=int(z["gen_seed"]), head_gain=float(z["head_gain"]))
KINDS = {"qkv": "mha.in_proj_weight", "wo": "mha.out_proj.weight", "w1": "net.0.weight", "w2": "net.2.weight"}


def rounded(pred):
    sd = dict(sd0)
    for k, v in sd0.items():
        if v.dim() == 2 and k.startswith("transformer.layers.") and pred(k):
            sd[k] = v.half().float()
    return sd


def layer_of(k):
    return int(k.split(".")[2])


CASES = [("all weights exact", lambda k: False),
         ("all trunk weights fp16 (heads exact)", lambda k: True),
         ("exact: qkv + w1 (the LayerNorm consumers)", lambda k: not (KINDS["qkv"] in k or KINDS["w1"] in k)),
         ("exact: wo + w2 (the residual GEMMs)", lambda k: not (KINDS["wo"] in k or KINDS["w2"] in k)),
         ("exact: w1 + w2 (FFN)", lambda k: not (KINDS["w1"] in k or KINDS["w2"] in k)),
         ("exact: qkv + wo (attention)", lambda k: not (KINDS["qkv"] in k or KINDS["wo"] in k)),
         ("exact: layers 12..23", lambda k: layer_of(k) < 12),
         ("exact: layers 0..11", lambda k: layer_of(k) >= 12),
         ("exact: layers 18..23", lambda k: layer_of(k) < 18)]
gen = LFQBert(img_size=256, hidden_dim=1024, codebook_size=2 ** bits, codebook_splits=2, depth=24, heads=16, mlp_dim=4096, dropout=0.1, nclass=1000, input_stride=16)
noise = None
for tag, pred in CASES:
    gen.load_state_dict(rounded(pred), strict=True)
    m = gen.eval().requires_grad_(False).to("cuda")
    m._drop_engine()
    m.weight_split, m.act_split, m.cfg_pair = 1, 0, 1
    if noise is None:
        noise = R.reference_noise(g, m.device)
    bad, tot, per, _ = R.teacher_forced(m, g, noise)
    nb = max(1, len(per) // 8)
    print(f"{name} {tag:46s}: {bad:4d}/{tot} = {bad / tot:.2e}  per eighth {[sum(per[i:i + nb]) for i in range(0, len(per), nb)]}", flush=True)
