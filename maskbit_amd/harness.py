"""The generation loop of the reference's evaluation driver, kept on the device.

``scripts/eval_maskbit.py:103-137`` draws ``labels = randperm(1000).repeat(50)``, then per batch calls ``sample()``, clamps to
[0, 1], multiplies by 255, permutes to NHWC, casts to uint8 (truncating) and copies to the host -- one blocking copy of a
float image per batch.  Here the uint8 NHWC image is written by the decoder's last kernel (``mb_sample`` -> ``mb_dec_decode``'s
``img_nhwc_u8`` output), and the device-to-host copy of batch *i* runs on a side stream into pinned memory while batch *i+1*
is being sampled: the host only waits for a copy that has had a whole batch time to finish.

The random-number protocol per batch is that of ``sample()`` (``maskbit_amd/sampling.py``), so for a fixed seed the images equal
what a batch-by-batch ``sample()`` + the reference's post-processing gives (tested, bit for bit).
"""
from __future__ import annotations

import math
from typing import Iterator, Optional, Text

import numpy as np
import torch

from .bert import LFQBert
from .conv_vqgan import ConvVQModel
from .sampling import build_plan, run_chunked, _ForcedPlan


def eval_labels(device, nclass: int = 1000, repeats: int = 50) -> torch.Tensor:
    """``randperm(nclass).repeat(repeats)`` drawn on ``device`` (eval_maskbit.py:107-108): every batch of the run sees a class mix."""
    return torch.randperm(nclass, dtype=torch.int, device=device).repeat(repeats)


def mask_token_for(num_codebook_entries: int, codebook_splits: int) -> int:
    """``int(2 ** (log2(entries) // splits))`` (eval_maskbit.py:77,80)."""
    return int(2 ** (math.log2(num_codebook_entries) // codebook_splits))


@torch.no_grad()
def generate_uint8(model: LFQBert, vqgan_model: ConvVQModel, labels: torch.Tensor, batchsize: int, *,
                   softmax_temperature: float = 1.0, randomize_temperature: float = 4.5, mask_schedule_strategy: Text = "linear",
                   num_steps: int = 12, guidance_scale: float = 3.0, guidance_annealing: Text = "none",
                   use_sampling_annealing: bool = False, scale_pow: float = 4.0,
                   total_samples: Optional[int] = None, return_codes: bool = False) -> Iterator[np.ndarray]:
    """Yield ``total_samples // batchsize`` arrays ``uint8 [batchsize, H, W, 3]`` (host memory), batch *i* generated for
    ``labels[batchsize * i : batchsize * (i + 1)]`` exactly as eval_maskbit.py:111-135 does.  Each yielded array is a fresh copy
    (the reference appends them to a list).  ``return_codes=True`` yields ``(images, codes int64 [batchsize, n])`` -- the combined
    tokens each image was decoded from (what the reference's evaluator takes as ``codebook_indices``, evaluator.py:536)."""
    if not isinstance(model, LFQBert) or not isinstance(vqgan_model, ConvVQModel):
        raise TypeError("generate_uint8() needs a maskbit_amd generator and tokenizer")
    dev = model._require_cuda("generate_uint8")
    model.eval()
    vqgan_model.eval()
    total = int(labels.numel()) if total_samples is None else int(total_samples)
    nbatch = total // batchsize
    if nbatch * batchsize > labels.numel():
        raise ValueError(f"{labels.numel()} labels do not cover {nbatch} batches of {batchsize}")
    model._check_labels(labels)
    n, m = model.seq_len, model.splits
    plan = build_plan(num_steps, n * m, guidance_scale, guidance_annealing, scale_pow, softmax_temperature,
                      use_sampling_annealing, mask_schedule_strategy)
    if guidance_scale != 0.0 and not any(s != 0.0 for s in plan[0]):
        plan = _ForcedPlan(plan)
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    pinned = [None, None]
    pending = None                                      # (slot, copy-done event) of the batch whose copy is in flight

    def collect(p):
        slot, done, codes = p
        done.synchronize()
        out = pinned[slot].numpy().copy()
        return (out, codes.cpu().numpy()) if return_codes else out

    for i in range(nbatch):
        y = labels[batchsize * i: batchsize * (i + 1)].long()
        _, u8, _, codes = run_chunked(model, vqgan_model, y, plan, randomize_temperature, want_steps=False, want_image=False, want_u8=True)
        slot = i & 1
        if pinned[slot] is None or pinned[slot].shape != u8.shape:
            pinned[slot] = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            pinned[slot].copy_(u8, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        u8.record_stream(side)                           # the allocator must not hand the block out before the copy has read it
        if pending is not None:
            yield collect(pending)                       # batch i-1: its copy overlapped the sampling just enqueued
        pending = (slot, done, codes)
    if pending is not None:
        yield collect(pending)


def to_evaluator_uint8(u8_nhwc: torch.Tensor) -> torch.Tensor:
    """NHWC uint8 (the decoder epilogue's output) -> the NCHW uint8 tensor ``GeneratorEvaluator.update`` builds for its Inception
    network, ``(generated_images * 255).to(torch.uint8)`` on the clamped float image (evaluator/evaluator.py:549-551): the same
    bytes without materialising the float image.  A view, no copy."""
    return u8_nhwc.permute(0, 3, 1, 2)
