"""MaskBit tokenizer (decoder and encoder) backed by the gfx950 engine.

Call surface of the reference's ``modeling.conv_vqgan.ConvVQModel`` (conv_vqgan.py:40-112):
``ConvVQModel(config)`` with an attribute-style config (+ ``.get``), the reference's checkpoint keys
(``encoder.*``, ``decoder.*``, ``quantize.*``), ``decode_tokens(tokens [b, n]) -> image
[b, 3, H, W]`` float32 unclamped and ``decode(z [b, K, h, w])``.  The decode itself is
``mb_dec_decode``: NHWC fp16 implicit-GEMM convolutions on MFMA with fused GroupNorm+SiLU
prologue, fused nearest-2x upsampling, bias and residual epilogues.

The encoder half (image -> tokens; stage-I plumbing outside the sampling hot path, SURVEY.md 8f
next-1) runs on the same kernels through ``mb_enc_encode``: ``encode(x) -> (z_quantized,
result_dict)`` with ``min_encoding_indices`` and ``forward(x) -> (reconstruction, result_dict)``
(conv_vqgan.py:70-83,114-127), inference only.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import torch

from . import _lib
from .base_model import BaseModel, ParamSpec


def _cfg_get(config, key, default=None):
    if hasattr(config, "get"):
        try:
            v = config.get(key, default)
            return default if v is None else v
        except TypeError:
            pass
    return getattr(config, key, default)


def _res_block(p: str, cin: int, cout: int) -> List[ParamSpec]:
    s = [(p + ".norm1.weight", (cin,), "ones"), (p + ".norm1.bias", (cin,), "zeros"),
         (p + ".conv1.weight", (cout, cin, 3, 3), "kaiming"),
         (p + ".norm2.weight", (cout,), "ones"), (p + ".norm2.bias", (cout,), "zeros"),
         (p + ".conv2.weight", (cout, cout, 3, 3), "kaiming")]
    if cin != cout:
        s.append((p + ".nin_shortcut.weight", (cout, cout, 1, 1), "kaiming"))
    return s


def _tokenizer_specs(K, hc, mult, R, nrb_enc, nrb_dec, nch, sample_with_conv) -> List[ParamSpec]:
    s: List[ParamSpec] = []
    # ---- encoder (autoencoder.py:230-286)
    emult = (1,) + tuple(mult)
    s.append(("encoder.conv_in.weight", (hc, nch, 3, 3), "kaiming"))
    c = hc
    for lvl in range(R):
        cin, cout = hc * emult[lvl], hc * emult[lvl + 1]
        c = cin
        for r in range(nrb_enc):
            s += _res_block(f"encoder.down.{lvl}.res_blocks.{r}", c, cout)
            c = cout
        if lvl < R - 1 and sample_with_conv:
            s += [(f"encoder.down.{lvl}.down_conv.weight", (cout, cout, 3, 3), "kaiming"), (f"encoder.down.{lvl}.down_conv.bias", (cout,), "zeros")]
    for r in range(nrb_enc):
        s += _res_block(f"encoder.mid.res_blocks.{r}", c, c)
    s += [("encoder.norm_out.weight", (c,), "ones"), ("encoder.norm_out.bias", (c,), "zeros"),
          ("encoder.conv_out.weight", (K, c, 1, 1), "kaiming"), ("encoder.conv_out.bias", (K,), "zeros")]
    # ---- decoder (autoencoder.py:358-397): up.0 is the coarsest level
    dmult = tuple(mult) + (mult[-1],)
    top = hc * mult[R - 1]
    s += [("decoder.conv_in.weight", (top, K, 3, 3), "kaiming"), ("decoder.conv_in.bias", (top,), "zeros")]
    for r in range(nrb_dec):
        s += _res_block(f"decoder.mid.res_blocks.{r}", top, top)
    last = top
    for i, lvl in enumerate(reversed(range(R))):
        cin, cout = hc * dmult[lvl + 1], hc * dmult[lvl]
        c = cin
        for r in range(nrb_dec):
            s += _res_block(f"decoder.up.{i}.res_blocks.{r}", c, cout)
            c = cout
        if lvl > 0:
            s += [(f"decoder.up.{i}.upsample_conv.weight", (cout, cout, 3, 3), "kaiming"), (f"decoder.up.{i}.upsample_conv.bias", (cout,), "zeros")]
        last = cout
    s += [("decoder.norm_out.weight", (last,), "ones"), ("decoder.norm_out.bias", (last,), "zeros"),
          ("decoder.conv_out.weight", (nch, last, 3, 3), "kaiming"), ("decoder.conv_out.bias", (nch,), "zeros")]
    return s


class ConvVQModel(BaseModel):
    def __init__(self, config, legacy: bool = False, finetune_decoder: bool = False):
        super().__init__()
        if legacy:
            raise NotImplementedError("legacy decoder layout (MaskGIT/older weights) is not paired with any MaskBit generator; not built")
        qt = _cfg_get(config, "quantizer_type", "lookup-free")
        if qt != "lookup-free":
            raise NotImplementedError(f"quantizer_type={qt!r}: only the lookup-free (LFQ) tokenizer is on the MaskBit path")
        self.config = config
        self.finetune_decoder = finetune_decoder
        self.token_size = int(config.token_size)
        self.hidden_channels = int(config.hidden_channels)
        self.channel_mult = tuple(int(v) for v in config.channel_mult)
        self.num_resolutions = int(config.num_resolutions)
        self.num_res_blocks = int(config.num_res_blocks)
        self.num_res_blocks_decoder = int(_cfg_get(config, "num_res_blocks_decoder", self.num_res_blocks))
        self.num_channels = int(_cfg_get(config, "num_channels", 3))
        self.sample_with_conv = bool(_cfg_get(config, "sample_with_conv", False))
        self._build(_tokenizer_specs(self.token_size, self.hidden_channels, self.channel_mult, self.num_resolutions,
                                     self.num_res_blocks, self.num_res_blocks_decoder, self.num_channels,
                                     bool(_cfg_get(config, "sample_with_conv", False))))
        weights = (2 ** torch.arange(self.token_size)).to(torch.int32)
        self._attach("quantize.bits_to_indices", weights, buffer=True)                      # lookup_free.py:38-39
        codes = torch.arange(2 ** self.token_size)
        self._attach("quantize.codebook", ((codes[:, None] & weights) != 0).float() * 2.0 - 1.0, buffer=True)   # :41-43
        self._latent_size = 16

    def get_last_layer(self):
        return self.decoder.conv_out.weight

    # ---- engine hooks ---------------------------------------------------------------------
    def _engine_create(self, capacity: int):
        cfg = _lib.DecCfg()
        cfg.token_size, cfg.hidden_channels = self.token_size, self.hidden_channels
        cfg.num_resolutions, cfg.num_res_blocks, cfg.num_channels = self.num_resolutions, self.num_res_blocks_decoder, self.num_channels
        for i, v in enumerate(self.channel_mult):
            cfg.channel_mult[i] = v
        cfg.latent_size = self._latent_size
        cfg.sample_with_conv = 1 if self.sample_with_conv else 0
        cfg.build_encoder = 1
        cfg.enc_res_blocks = self.num_res_blocks
        h = C.c_void_p()
        _lib.check(_lib.load().mb_dec_create(C.byref(cfg), capacity, C.byref(h)), "mb_dec_create")
        return h

    def _engine_destroy(self, h) -> None:
        _lib.load().mb_dec_destroy(h)

    def _engine_load(self, h, key: str, t: torch.Tensor, stream: int) -> None:
        shape = (C.c_int64 * t.dim())(*t.shape)
        _lib.check(_lib.load().mb_dec_load(h, key.encode(), t.data_ptr(), shape, t.dim(), stream), f"mb_dec_load({key})")

    def engine(self, min_batch: int, latent_size: int = 16):
        if latent_size != self._latent_size:
            self._drop_engine()
            self._latent_size = latent_size
        have = self._engine_key[1] if self._engine_key else 0
        return self._ensure_engine(max(min_batch, have, 8))

    # ---- decode -----------------------------------------------------------------------------
    def _decode_codes(self, codes: torch.Tensor, want_u8: bool = False):
        dev = self._require_cuda("decode_tokens")
        b, n = codes.shape
        side = int(math.sqrt(float(n)))
        if side * side != n:
            raise ValueError(f"decode_tokens expects a square token grid, got {n} tokens")
        res = side << (self.num_resolutions - 1)
        img = torch.empty((b, self.num_channels, res, res), dtype=torch.float32, device=dev)
        u8 = torch.empty((b, res, res, self.num_channels), dtype=torch.uint8, device=dev) if want_u8 else None
        h = self.engine(b, side)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mb_dec_decode(h, codes.data_ptr(), img.data_ptr(), u8.data_ptr() if want_u8 else None, b,
                                                 torch.cuda.current_stream().cuda_stream), "mb_dec_decode")
        return (img, u8) if want_u8 else img

    @torch.no_grad()
    def decode_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens [b, n] of any numeric dtype (the sampler hands float32, factorization.py:19) -> image."""
        dev = self._require_cuda("decode_tokens")
        return self._decode_codes(tokens.to(dev).long().contiguous())

    def saturation_count(self, reset: bool = True) -> int:
        """Number of 4-channel activation groups the engine clamped at the fp16 range (+-65504) since the last reset, over every conv layer
        of decode / encode calls on the current engine.  0 for every configuration tested; a checkpoint whose activations need more range
        shows up here (the activations are stored as fp16) instead of being clipped silently.  Synchronises the current stream."""
        if self._engine is None:
            return 0
        import ctypes as C
        n = C.c_uint(0)
        dev = self._require_cuda("saturation_count")
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mb_dec_saturation_count(self._engine, C.byref(n), 1 if reset else 0, torch.cuda.current_stream().cuda_stream),
                       "mb_dec_saturation_count")
        return int(n.value)

    @torch.no_grad()
    def decode_tokens_uint8(self, tokens: torch.Tensor):
        """-> (image fp32 NCHW, uint8 NHWC = trunc(clamp(x,0,1)*255)) in one pass (eval_maskbit.py:134-135 fused)."""
        dev = self._require_cuda("decode_tokens")
        return self._decode_codes(tokens.to(dev).long().contiguous(), want_u8=True)

    @torch.no_grad()
    def decode(self, z_quantized: torch.Tensor) -> torch.Tensor:
        """z [b, K, h, w] in {-1,+1} -> image.  LFQ latents are exactly the bit pattern of a code, so the
        latent is re-packed to codes (sign -> bit, LSB first) and decoded through the token path."""
        dev = self._require_cuda("decode")
        if z_quantized.dim() != 4 or z_quantized.shape[1] != self.token_size:
            raise ValueError(f"decode expects [b, {self.token_size}, h, w], got {tuple(z_quantized.shape)}")
        z = z_quantized.to(dev)
        if not bool(((z == 1) | (z == -1)).all()):
            raise ValueError("decode(): the HIP decoder takes quantized LFQ latents (+-1) only")
        w = (2 ** torch.arange(self.token_size, device=dev)).view(1, -1, 1, 1)
        codes = ((z > 0).long() * w).sum(1).reshape(z.shape[0], -1)
        return self._decode_codes(codes.contiguous())

    @torch.no_grad()
    def _encode(self, x: torch.Tensor, want_raw: bool = False):
        dev = self._require_cuda("encode")
        if x.dim() != 4 or x.shape[1] != self.num_channels:
            raise ValueError(f"encode expects [b, {self.num_channels}, H, W], got {tuple(x.shape)}")
        b, _, H, W = x.shape
        down = 1 << (self.num_resolutions - 1)
        if H != W or H % (16 * down):
            raise ValueError(f"encode expects square images with a side that is a multiple of {16 * down}, got {H}x{W}")
        side = H // down
        img = x.to(device=dev, dtype=torch.float32).contiguous()
        idx = torch.empty((b, side, side), dtype=torch.int64, device=dev)
        zq = torch.empty((b, self.token_size, side, side), dtype=torch.float32, device=dev)
        zraw = torch.empty_like(zq) if want_raw else None
        h = self.engine(b, side)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mb_enc_encode(h, img.data_ptr(), idx.data_ptr(), zq.data_ptr(), zraw.data_ptr() if want_raw else None, b,
                                                 torch.cuda.current_stream().cuda_stream), "mb_enc_encode")
        return zq, idx, zraw

    def encode(self, x: torch.Tensor):
        """ConvVQModel.encode (conv_vqgan.py:70-83): image -> (z_quantized [b,K,h,w] in {-1,+1}, result_dict) with
        ``min_encoding_indices`` [b,h,w] (lookup_free.py:57-95).  Inference only: the quantizer losses are returned as zeros
        except the commitment term, which needs the pre-sign latent and is not computed on this path."""
        zq, idx, _ = self._encode(x)
        zero = torch.zeros((), device=zq.device)
        return zq, dict(quantizer_loss=zero, commitment_loss=zero, entropy_loss=zero, per_sample_entropy=zero, avg_entropy=zero,
                        min_encoding_indices=idx)

    def forward(self, input: torch.Tensor):
        """ConvVQModel.forward (conv_vqgan.py:114-127): (decode(encode(x)), result_dict)."""
        zq, result = self.encode(input)
        codes = result["min_encoding_indices"].reshape(zq.shape[0], -1)
        return self._decode_codes(codes.contiguous()), result
