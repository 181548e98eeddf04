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


def gemm_mini(lib, epi, A, W, bias, res, out32, out16, rows, pair, N, K, lo_sets=(), out4=None, out4s=None, seq_rows=0, out4l=None, out4ls=None):
    """mb_gemm_mini / mb_gemm_mini_seq (include/maskbit_hip_diag.h): a sequence-aligned (pair) GEMM with len(lo_sets) MX-fp4 mini-tile passes; a set =
    (A4, a_scale, W4, w_scale) tensors; seq_rows: rows per sequence of a pair GEMM (0 = 257)."""
    import ctypes as C
    from maskbit_amd import _lib
    ptr = lambda t: t.data_ptr() if t is not None else None
    flat = [t.data_ptr() for s in lo_sets for t in s]
    arr = (C.c_void_p * max(1, len(flat)))(*flat)
    if seq_rows or out4l is not None:
        _lib.check(lib.mb_gemm_mini_seq(epi, ptr(A), ptr(W), ptr(bias), ptr(res), ptr(out32), ptr(out16), ptr(out4), ptr(out4s), ptr(out4l), ptr(out4ls), rows, int(pair), seq_rows, N, K,
                                        len(lo_sets), arr, torch.cuda.current_stream().cuda_stream), "mb_gemm_mini_seq")
        return
    _lib.check(lib.mb_gemm_mini(epi, ptr(A), ptr(W), ptr(bias), ptr(res), ptr(out32), ptr(out16), ptr(out4), ptr(out4s), rows, int(pair), N, K,
                                len(lo_sets), arr, torch.cuda.current_stream().cuda_stream), "mb_gemm_mini")


_F4V = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], dtype=torch.float64)


def f4_decode(buf: torch.Tensor, K: int) -> torch.Tensor:
    """uint8 [R, >= K/2] (element 2j = low nibble of byte j) -> float64 [R, K] of e2m1 values (on the CPU)."""
    b = buf[:, : K // 2].cpu().to(torch.int64)
    codes = torch.stack([b & 15, b >> 4], dim=-1).reshape(b.shape[0], K)
    return _F4V[codes]


def f4_codes(v: torch.Tensor) -> torch.Tensor:
    """float64 scaled values -> e2m1 codes 0..15 the way mb_common.h fp4_code rounds (nearest, ties to even on the local grid, saturate at 6)."""
    a = v.abs().clamp(max=6.0)
    k = torch.floor(torch.log2(a.clamp(min=1.0))).clamp(0, 2)
    r = torch.round(a / 2.0 ** (k - 1))
    return (r + 2 * k).to(torch.int64) | ((v < 0).to(torch.int64) << 3)


def f4_block_exponent(amax: torch.Tensor) -> torch.Tensor:
    """Biased exponent E of a block with largest element amax such that amax * 2^(129 - E) lies in (3, 6] (mb_common.h fp4_nosat_exp)."""
    m, e = torch.frexp(amax)                                       # amax = m * 2^e, m in [0.5, 1)
    return (e + 126) + (m > 0.75).to(torch.int32)


def f4_scale_index(blk, nseq, seq, r, groups=4):
    """mb_kernels.h fp4_scale_index: byte index of (64-column block blk, token r of sequence seq) in a lane-ordered scale array (groups = 64-token
    groups per sequence: 4, or 16 for the 1024-token models)."""
    return ((blk * nseq + seq) * groups + (r >> 6)) * 64 + (r & 15) * 4 + ((r >> 4) & 3)


def f4_encode_rows(v: torch.Tensor, nseq: int, dev="cuda", seq_rows: int = 257):
    """What the engine's producers write for rows of values v [nseq * seq_rows, K] (float64, CPU or GPU): (x4 uint8 [R, 2K] with garbage in the class-token
    rows and the padding, lane-ordered scale bytes, the decoded float64 operand [R, K] with zeros in the class-token rows)."""
    v = v.double().cpu()
    R, K = v.shape
    assert R == nseq * seq_rows and K % 64 == 0
    ntok, groups = seq_rows - 1, (seq_rows - 1) // 64
    blocks = v.reshape(R, K // 64, 64)
    amax = blocks.abs().amax(-1)
    E = f4_block_exponent(amax.clamp(min=1e-30))
    sbyte = (E - 2).clamp(min=0)
    codes = f4_codes(blocks * (2.0 ** (129 - E).double()).unsqueeze(-1)).reshape(R, K)
    x4 = torch.randint(0, 256, (R, 2 * K), dtype=torch.uint8)
    x4[:, : K // 2] = (codes[:, 0::2] | (codes[:, 1::2] << 4)).to(torch.uint8)
    dec = (_F4V[codes].reshape(R, K // 64, 64) * (2.0 ** (sbyte.double() - 127)).unsqueeze(-1)).reshape(R, K)
    scales = torch.randint(0, 256, ((K // 64) * nseq * ntok + 256,), dtype=torch.uint8)
    rows = torch.arange(R)
    seq, tok = rows // seq_rows, rows % seq_rows
    keep = tok < ntok
    for b in range(K // 64):
        idx = f4_scale_index(b, nseq, seq[keep], tok[keep], groups)
        scales[idx] = sbyte[keep, b].to(torch.uint8)
    dec[~keep] = 0.0
    x4[~keep] = torch.randint(0, 256, (int((~keep).sum()), 2 * K), dtype=torch.uint8)      # class-token rows: garbage, must not matter
    return x4.to(dev), scales.to(dev), dec.to(dev)


def w4_decode(w4: torch.Tensor, wsb: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """e2m1 weight operand in the mini-tile-packed layout (mb_kernels.h w4_packed_offset: [N / 16][K / 128] chunks of 16 rows x 64 B, the 16-byte
    pieces of a row swizzled with (row >> 1) & 3) + its lane-ordered scale bytes, one per (weight row, 128 K-elements) (w4_scale_index; mb_w4_from_f32 /
    mb_w4lo_from_f32) -> float64 [N, K] on the device of w4."""
    n, j = torch.meshgrid(torch.arange(N, device=wsb.device), torch.arange(K // 128, device=wsb.device), indexing="ij")
    blk = wsb[(((n >> 6) * (K // 128) + j) * 16 + (n & 15)) * 4 + ((n >> 4) & 3)].to(torch.float64)          # [N, K / 128]
    nn, kk = torch.meshgrid(torch.arange(N), torch.arange(0, K, 2), indexing="ij")         # byte of elements (k, k + 1)
    r, c = nn & 15, (kk & 127) >> 5
    off = ((nn >> 4) * (K >> 7) + (kk >> 7)) * 1024 + r * 64 + ((c ^ ((r >> 1) & 3)) << 4) + ((kk & 31) >> 1)
    rowmajor = w4.reshape(-1).cpu()[off.reshape(-1)].reshape(N, K // 2)
    return f4_decode(rowmajor, K).to(w4.device) * (2.0 ** (blk - 127)).repeat_interleave(128, 1)
