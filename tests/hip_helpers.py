"""Shared helpers for the GPU parity tests: build HIP-backed models from oracle state dicts."""
import torch

from oracle import maskbit_oracle as O


class Cfg(dict):
    __getattr__ = dict.__getitem__


def tok_config(c: O.TokCfg) -> Cfg:
    return Cfg(quantizer_type="lookup-free", codebook_size=2 ** c.token_size, token_size=c.token_size,
               commitment_cost=0.25, entropy_loss_weight=0.02, entropy_loss_temperature=0.01, entropy_gamma=1.0,
               num_channels=c.num_channels, hidden_channels=c.hidden_channels, channel_mult=list(c.channel_mult),
               num_resolutions=c.num_resolutions, num_res_blocks=c.num_res_blocks, sample_with_conv=c.sample_with_conv)


def hip_generator(cfg: O.GenCfg, sd, device="cuda"):
    from maskbit_amd import LFQBert
    m = LFQBert(img_size=256, hidden_dim=cfg.hidden, codebook_size=2 ** cfg.bits, codebook_splits=cfg.splits,
                depth=cfg.depth, heads=cfg.heads, mlp_dim=cfg.mlp, dropout=0.1, nclass=cfg.nclass, input_stride=16)
    m.load_state_dict(sd, strict=True)
    return m.eval().requires_grad_(False).to(device)


def hip_tokenizer(cfg: O.TokCfg, sd, device="cuda"):
    from maskbit_amd import ConvVQModel
    m = ConvVQModel(tok_config(cfg))
    m.load_state_dict(sd, strict=True)
    return m.eval().requires_grad_(False).to(device)


def token_mismatch(a: torch.Tensor, b: torch.Tensor, where=None) -> float:
    ne = (a != b)
    if where is not None:
        return float(ne[where].float().mean()) if bool(where.any()) else 0.0
    return float(ne.float().mean())
