"""512x512 models: 32x32 = 1024 image tokens + the class token.  One head's K/V no longer fits in LDS, so
attention runs the streaming kernel (128-key blocks, online softmax); everything else is the same code at a larger N."""
import pytest
import torch

from hip_helpers import hip_tokenizer
from oracle import maskbit_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gen(cfg, sd):
    from maskbit_amd import LFQBert
    m = LFQBert(img_size=512, hidden_dim=cfg.hidden, codebook_size=2 ** cfg.bits, codebook_splits=cfg.splits, depth=cfg.depth, heads=cfg.heads,
                mlp_dim=cfg.mlp, dropout=0.1, nclass=cfg.nclass, input_stride=16)
    assert m.seq_len == 1024
    m.load_state_dict(sd, strict=True)
    return m.eval().requires_grad_(False).to(DEV)


@pytest.mark.parametrize("hidden,heads,depth,b", [(128, 4, 2, 3), (128, 2, 1, 2), (1024, 16, 1, 2)])
def test_generator_1025_tokens_vs_oracle(hidden, heads, depth, b):
    """head dims 32 and 64; 1025 = 8 key blocks + 1 key: the last block is almost entirely masked"""
    cfg = O.GenCfg(bits=12, splits=2, hidden=hidden, depth=depth, heads=heads, mlp=2 * hidden, seq=1024, nclass=10)
    sd = O.make_generator_weights(cfg, seed=hidden + heads, head_gain=12.0)
    m = _gen(cfg, sd)
    g = torch.Generator().manual_seed(b)
    toks = torch.randint(0, 65, (b, 1024, 2), generator=g); y = torch.randint(0, 10, (b,), generator=g)
    drop = torch.tensor([False, True, False][:b])
    out = m(toks.to(DEV), y.to(DEV), drop.to(DEV))
    ref = O.lfq_bert_forward(sd, cfg, toks, y, drop)
    rel = float((out.cpu() - ref).norm() / ref.norm())
    print(f"hidden {hidden} heads {heads}: rel-Frobenius logit error {rel:.2e}")
    assert out.shape == (b, 1024, 2, 64) and torch.isfinite(out).all() and rel < 2e-3
    assert torch.equal(m(toks[:1].to(DEV), y[:1].to(DEV), drop[:1].to(DEV)), out[:1])      # batch invariance


def test_sample_512_end_to_end_tiny():
    """sample() with a 1024-token generator and a 32x32-latent decode to 128x128 (tiny tokenizer, 3 resolutions)."""
    from maskbit_amd import sample
    gcfg = O.GenCfg(bits=12, splits=2, hidden=128, depth=1, heads=4, mlp=256, seq=1024, nclass=10)
    gsd = O.make_generator_weights(gcfg, seed=5, head_gain=12.0)
    m = _gen(gcfg, gsd)
    tcfg = O.TokCfg(token_size=12, hidden_channels=64, channel_mult=(1, 1, 2), num_resolutions=3, num_res_blocks=1)
    tsd = O.make_tokenizer_weights(tcfg, seed=21, with_encoder=True)
    tk = hip_tokenizer(tcfg, tsd)
    torch.manual_seed(0)
    img, steps = sample(m, tk, num_samples=2, labels=torch.tensor([1, 7]).to(DEV), num_steps=4, guidance_scale=2.0, mask_token=64, patch_size=32,
                        codebook_size=4096, codebook_splits=2, randomize_temperature=2.0, mask_schedule_strategy="arccos")
    assert img.shape == (2, 3, 128, 128) and torch.isfinite(img).all()
    assert len(steps) == 4 and steps[-1].shape == (2, 1024, 2) and int(steps[-1].max()) < 64
    codes = O.combine_groups(steps[-1].cpu(), 12, 2)
    want = O.decode_tokens(tsd, tcfg, codes)
    assert float((img.cpu() - want).abs().max()) < 0.03
