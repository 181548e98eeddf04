"""Golden logits for the generator VARIANTS the reference also defines: LFQBert with use_prenorm=True and the
embedding-table ``Bert`` (post- and pre-norm).  Same rules as make_golden.py: the real reference is imported in the build
container, loaded (strict) with the oracle's seeded weights, and only tensors are stored -> tests/golden/gen_variants_tiny.npz.
Run:  python oracle/make_golden_variants.py"""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import maskbit_oracle as O
from oracle import make_golden as G

BASE = dict(bits=12, splits=2, hidden=128, depth=2, heads=4, mlp=256, seq=256, nclass=10)
VARIANTS = {
    "lfq_prenorm": (O.GenCfg(**BASE, prenorm=True, kind="lfq"), 31),
    "bert_postnorm": (O.GenCfg(**BASE, prenorm=False, kind="bert"), 32),
    "bert_prenorm": (O.GenCfg(**BASE, prenorm=True, kind="bert"), 33),
    "bert_3groups": (O.GenCfg(bits=12, splits=3, hidden=128, depth=1, heads=4, mlp=256, seq=256, nclass=10, kind="bert"), 34),
}


def main():
    torch.set_num_threads(8)
    G._import_reference()
    from modeling.bert import Bert, LFQBert
    out = {}
    for name, (cfg, seed) in VARIANTS.items():
        sd = O.make_generator_weights(cfg, seed=seed, head_gain=20.0)
        cls = Bert if cfg.kind == "bert" else LFQBert
        model = cls(img_size=256, hidden_dim=cfg.hidden, codebook_size=2 ** cfg.bits, codebook_splits=cfg.splits, depth=cfg.depth,
                    heads=cfg.heads, mlp_dim=cfg.mlp, dropout=0.1, nclass=cfg.nclass, input_stride=16, use_prenorm=cfg.prenorm)
        model.load_state_dict(sd, strict=True)                     # pins key names and shapes of the variant
        model = model.eval().requires_grad_(False)
        toks = G.masked_test_tokens(cfg, 4, seed=seed)
        labels = torch.tensor([0, 3, 9, 7]); drop = torch.tensor([False, True, False, True])
        logits = model(toks.clone(), labels.clone(), drop.clone())
        mine = O.lfq_bert_forward(sd, cfg, toks, labels, drop)
        print(f"{name}: logits {tuple(logits.shape)} |max| {float(logits.abs().max()):.3f}; oracle restatement max err {float((mine - logits).abs().max()):.2e}")
        out.update({f"{name}.tokens": toks.numpy(), f"{name}.labels": labels.numpy(), f"{name}.drop": drop.numpy(), f"{name}.logits": logits.numpy(),
                    f"{name}.seed": seed, f"{name}.w_sha": G.sha(sd["pos_emb"])})
    np.savez_compressed(os.path.join(G.OUT, "gen_variants_tiny.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()
