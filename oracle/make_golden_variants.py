"""Golden logits for the generator VARIANTS the reference also defines: LFQBert with use_prenorm=True and the
embedding-table ``Bert`` (post- and pre-norm); the ``return_attn=True`` outputs (logits + per-layer attention maps); and the tokenizer
with average-pool downsampling (``sample_with_conv=False``) -> tests/golden/tok_avgpool_tiny.npz.  Same rules as make_golden.py: the real reference is imported in the build
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


ATTN_VARIANTS = {"attn_lfq_postnorm": (O.GenCfg(**BASE), 35), "attn_lfq_prenorm": (O.GenCfg(**BASE, prenorm=True), 36),
                 "attn_bert_postnorm": (O.GenCfg(**BASE, kind="bert"), 37)}
AVGPOOL_TOK = O.TokCfg(token_size=12, hidden_channels=64, channel_mult=(1, 1, 2), num_resolutions=3, num_res_blocks=1, sample_with_conv=False)


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
    # return_attn=True (bert.py:461,505-508): logits + the per-layer attention maps, post- and pre-norm LFQBert and the table Bert
    for name, (cfg, seed) in ATTN_VARIANTS.items():
        sd = O.make_generator_weights(cfg, seed=seed, head_gain=20.0)
        cls = Bert if cfg.kind == "bert" else LFQBert
        model = cls(img_size=256, hidden_dim=cfg.hidden, codebook_size=2 ** cfg.bits, codebook_splits=cfg.splits, depth=cfg.depth,
                    heads=cfg.heads, mlp_dim=cfg.mlp, dropout=0.1, nclass=cfg.nclass, input_stride=16, use_prenorm=cfg.prenorm)
        model.load_state_dict(sd, strict=True)
        model = model.eval().requires_grad_(False)
        toks = G.masked_test_tokens(cfg, 2, seed=seed)
        labels = torch.tensor([4, 8]); drop = torch.tensor([False, True])
        logits, attn = model(toks.clone(), labels.clone(), drop.clone(), return_attn=True)
        mine, mattn = O.lfq_bert_forward(sd, cfg, toks, labels, drop, return_attn=True)
        assert isinstance(attn, list) and len(attn) == cfg.depth and len(mattn) == cfg.depth
        print(f"{name}: attn {tuple(attn[0].shape)} x {len(attn)}; oracle max err logits {float((mine - logits).abs().max()):.2e}, "
              f"attn {max(float((a - b).abs().max()) for a, b in zip(attn, mattn)):.2e}")
        out.update({f"{name}.tokens": toks.numpy(), f"{name}.labels": labels.numpy(), f"{name}.drop": drop.numpy(), f"{name}.logits": logits.numpy(),
                    f"{name}.attn": torch.stack(attn).numpy().astype(np.float16), f"{name}.seed": seed, f"{name}.w_sha": G.sha(sd["pos_emb"])})
    np.savez_compressed(os.path.join(G.OUT, "gen_variants_tiny.npz"), **out)

    # tokenizer with average-pool downsampling (sample_with_conv=False, autoencoder.py:179-182): encode + decode of one tiny image
    from modeling.conv_vqgan import ConvVQModel
    tcfg = AVGPOOL_TOK
    tsd = O.make_tokenizer_weights(tcfg, seed=41, with_encoder=True)
    tok = G.build_ref_tok(ConvVQModel, tcfg, tsd)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(42))
    zq, res = tok.encode(x)
    rec = tok.decode(zq)
    mzq, midx = O.encode_image(tsd, tcfg, x)
    print(f"tok_avgpool: indices {tuple(res['min_encoding_indices'].shape)}; oracle bits differing {int((mzq != zq).sum())}, "
          f"indices equal {bool(torch.equal(midx, res['min_encoding_indices']))}")
    np.savez_compressed(os.path.join(G.OUT, "tok_avgpool_tiny.npz"), seed=41, enc_input=x.numpy(), enc_zq=zq.numpy().astype(np.int8),
                        enc_indices=res["min_encoding_indices"].numpy(), recon=rec.numpy().astype(np.float16),
                        w_sha_enc_conv_in=G.sha(tsd["encoder.conv_in.weight"]), n_keys=len(tsd))
    print("done")


if __name__ == "__main__":
    main()
