"""Multi-GPU sampling: batch shards across ranks, one image gather at the end.

The sampling path is embarrassingly parallel over samples: every rank owns a
contiguous block of the batch and a full replica of the weights (0.61 GB fp16 + 59 MB), runs the whole
loop + decode on its block, and the only exchange is ONE ``all_gather`` of uint8 NHWC images
(196 608 B per image) over RCCL/xGMI.  There is no counterpart in the reference (its inference is
single-device, eval_maskbit.py:65).

For bit-parity with a single-device run of the same global batch, a rank must consume *its slice of
the batch-level noise* (noise tensors are row-major in the batch, so slices are contiguous) rather
than re-seeding per rank: ``sample_sharded(..., noise="batch")`` does that -- every rank then draws the
WHOLE batch's noise and keeps its rows (torch's generators cannot skip ahead, and the reference's CPU Gumbel
draw is ~12 ms per step at a global batch of 512), so it is the mode for REPRODUCING a single-device run, not
the throughput mode.  ``noise="rank"`` (the default) draws only the local shard's noise: the cost per rank does
not grow with the world size; the caller seeds each rank differently (``torch.manual_seed(seed + rank)``, as
bench.py does), otherwise all ranks would sample the same images.
When the batch divides evenly over the ranks a batch costs exactly ONE collective (``all_gather_into_tensor``
of the uint8 block) and no host synchronisation; only ragged batches exchange block sizes first.
"""
from __future__ import annotations

import warnings
from typing import List, Optional, Tuple

import torch


_WARNED_DEFAULT_NOISE = False


def shard_range(rank: int, world: int, batch: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the global batch owned by ``rank`` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slice_noise(exp_noise: torch.Tensor, conf_noise: torch.Tensor, lo: int, hi: int, rows_per_sample: int):
    """Rows of the global noise that belong to samples [lo, hi).
    exp_noise [steps, B*n*m, C] -> [steps, (hi-lo)*n*m, C];  conf_noise [steps, B, n, m] -> [steps, hi-lo, n, m]."""
    return (exp_noise[:, lo * rows_per_sample:hi * rows_per_sample].contiguous(), conf_noise[:, lo:hi].contiguous())


def gather_images(local: torch.Tensor, group=None, equal: Optional[bool] = None) -> torch.Tensor:
    """all_gather of per-rank image blocks [b_r, ...] -> [sum b_r, ...] in rank order.
    ``equal=True``: the caller knows every rank holds the same number of rows (the batch divides evenly over the ranks): ONE
    all_gather_into_tensor, no size exchange, no host synchronisation.  ``equal=None``: the block sizes are exchanged first (one small
    all_gather + a host sync) and equal blocks still take the single-collective path; ragged blocks are padded and gathered."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and local.is_cuda:          # test-only path: gloo collectives on host copies
        return gather_images(local.cpu(), group, equal).to(local.device)
    local = local.contiguous()
    if equal:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        out = torch.empty((world * sizes[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


@torch.no_grad()
def sample_sharded(model, vqgan_model, global_labels: torch.Tensor, *, noise: Optional[str] = None, group=None,
                   num_steps: int = 64, guidance_scale: float = 7.1, guidance_annealing: str = "cosine", scale_pow: float = 3.0,
                   softmax_temperature: float = 1.0, use_sampling_annealing: bool = False, randomize_temperature: float = 8.2,
                   mask_schedule_strategy: str = "arccos") -> torch.Tensor:
    """Sample ``len(global_labels)`` images across the process group; every rank returns all images,
    uint8 NHWC, identical on every rank (and, with ``noise="batch"``, identical to a 1-GPU run of the same seed; see the module docstring).
    ``noise`` left at None means "rank" (every rank draws its own shard's noise from ITS generators): with more than one rank this warns once,
    because a caller that seeds every rank identically -- what ``noise="batch"``, the default until round 3, wanted -- would then draw the same
    noise on every rank.  Pass ``noise="rank"`` (and seed the ranks differently) or ``noise="batch"`` explicitly."""
    import torch.distributed as dist
    from .sampling import _ForcedPlan, build_plan, draw_noise, plan_arrays, run_loop, step_chunks
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    B = int(global_labels.numel())
    lo, hi = shard_range(rank, world, B)
    n, m, C_ = model.seq_len, model.splits, model.effective_codebook_size
    plan = build_plan(num_steps, n * m, guidance_scale, guidance_annealing, scale_pow, softmax_temperature,
                      use_sampling_annealing, mask_schedule_strategy)
    if guidance_scale != 0.0 and not any(a != 0.0 for a in plan[0]):
        plan = _ForcedPlan(plan)                      # as sample(): the CFG forward runs even when every annealed scale is 0
    dev = model.device
    if noise is None:
        noise = "rank"
        global _WARNED_DEFAULT_NOISE
        if world > 1 and not _WARNED_DEFAULT_NOISE:
            _WARNED_DEFAULT_NOISE = True
            warnings.warn("sample_sharded: noise left at its default ('rank'): every rank draws its own shard's noise from its own generators -- "
                          "seed the ranks differently (torch.manual_seed(seed + rank)), or pass noise='batch' for the single-device-identical mode",
                          stacklevel=2)
    if noise not in ("batch", "rank"):
        raise ValueError("noise must be 'batch' or 'rank'")
    nb = B if noise == "batch" else hi - lo           # the batch the noise is drawn for; chunked by steps to bound its memory (sampling.step_chunks)
    if hi == lo:                                      # more ranks than samples: this rank contributes an empty block to the gather
        if noise == "batch":
            for (b0, b1) in step_chunks(nb, n, m, C_, num_steps):               # keep the generators in step with the other ranks
                draw_noise(nb, n, m, C_, num_steps, randomize_temperature, dev, b0, b1)
        side = int(round(n ** 0.5)) << (vqgan_model.num_resolutions - 1)
        return gather_images(torch.empty((0, side, side, vqgan_model.num_channels), dtype=torch.uint8, device=dev), group)
    chunks = step_chunks(nb, n, m, C_, num_steps)
    labels = global_labels[lo:hi].to(dev)
    cplan = plan_arrays(plan)
    u8 = None
    for (b0, b1) in chunks:
        e, c = draw_noise(nb, n, m, C_, num_steps, randomize_temperature, dev, b0, b1)
        if noise == "batch":
            e, c = slice_noise(e, c, lo, hi, n * m)
        _, u8, _, _ = run_loop(model, vqgan_model, labels, plan, e, c, want_steps=False, want_image=False, want_u8=True,
                               step_range=(b0, b1) if len(chunks) > 1 else None, _cplan=cplan)
    return gather_images(u8, group, equal=(B % world == 0) or None)
