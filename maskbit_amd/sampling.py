"""``sample()``: the MaskBit N-step masked bit-token sampler + decode, on the gfx950 engine.

Drop-in for the reference's ``modeling.modules.sample`` (sampling.py:13-136): same signature,
same return value ``(image [B,3,H,W] float32 unclamped on model.device, [pred tokens per step])``,
same random-number protocol -- per step one ``exponential_`` of shape [B*n*m, C] from the model
device's generator (what ``Categorical.sample`` -> ``torch.multinomial(n=1)`` draws) and one
``Gumbel(0,1).sample([B,n,m])`` from the CPU default generator -- so a fixed seed draws the same
noise the reference would on the same device.  The schedule (guidance scale, temperature, mask
length per step) is evaluated here on the host with the reference's float32 torch-scalar
arithmetic and handed to ``mb_sample`` as a plan; the loop body itself (2B-sequence forward, CFG
combine, softmax, draw, confidence, k-th-smallest re-mask, combine, decode) never leaves the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Text, Tuple

import torch

from . import _lib
from .bert import LFQBert
from .conv_vqgan import ConvVQModel
from .masking import get_masking_ratio


def build_plan(num_steps: int, num_maskable: int, guidance_scale: float, guidance_annealing: str, scale_pow: float,
               softmax_temperature: float, use_sampling_annealing: bool, mask_schedule_strategy: str):
    """Host-side per-step constants (sampling.py:82, 90-98, 103-104, 120-123)."""
    get_masking_ratio(1.0, mask_schedule_strategy)          # raises ValueError on a bad strategy before any GPU work
    scale, temp, mask_len = [], [], []
    for i in range(num_steps):
        progress = (i + 1) / num_steps
        if guidance_annealing == "none":
            a = guidance_scale * 1.0
        elif guidance_annealing == "linear":
            a = guidance_scale * (i / num_steps)
        elif guidance_annealing == "cosine":
            sp = torch.ones(1) * scale_pow                                      # float32, as in the reference
            a = float(guidance_scale * ((1 - torch.cos(((i / num_steps) ** sp) * torch.pi)) * 1 / 2))
        else:
            raise ValueError(f"guidance_annealing must be 'none', 'linear' or 'cosine', got {guidance_annealing!r}")
        scale.append(float(torch.tensor(a, dtype=torch.float32)))
        temp.append(0.5 + 0.8 * (1 - progress) if use_sampling_annealing else softmax_temperature)
        mask_len.append(int(torch.floor(get_masking_ratio(progress, mask_schedule_strategy) * num_maskable)))
    return scale, temp, mask_len


NOISE_CHUNK_BYTES = 1 << 30       # sample() / generate_uint8() draw and feed the Exp(1) noise in step chunks of at most this size
OVERLAP_CHUNKS = 8                # ... and in at least this many chunks (+ a one-step head): the host draws chunk k+1 while the device runs chunk k


class _DrawThread:
    """The host-side draws are a few small CPU tensor ops per step.  With the default intra-op pool of a many-core host (128 OpenMP / MKL threads on the
    256-CPU MI355X boxes) every such op wakes the pool, whose workers then spin -- and the HIP runtime's own host threads starve: measured on BASELINE
    configs[1] (16 steps, batch 16) 182-189 ms per run against 118 ms with the draws at one thread (tools/host_draw_ab.py; a `log` over 8 192 values takes
    2.7 ms there: MKL's vector math threads by itself, so drawing in pieces below ATen's own parallel grain does not help -- round 6 tried).
    Rounds 3-5 flipped ``torch.set_num_threads`` to 1 around every draw ON THE CALLING THREAD: the caller's own setting changed under it, per draw.  Now the
    draws run on ONE dedicated worker thread whose OWN intra-op thread count is 1 (OpenMP's and MKL's thread counts are per-thread settings); the caller
    blocks until its draws are done, so the order in which a run consumes the process-global default CPU generator is the caller's program order.  ATen
    also remembers the last value ANY thread set as the default for threads created later, so the worker's one-time 1 is followed, once, by the caller
    re-asserting the count it already has: after that no draw touches any thread setting."""
    _pool = None

    @classmethod
    def run(cls, fn):
        if cls._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            n = torch.get_num_threads()
            pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="maskbit-draw")
            # (get first: ATen initialises a thread's count lazily, from the last value set anywhere, on its first intra-op call -- that must not come later)
            pool.submit(lambda: (torch.get_num_threads(), torch.set_num_threads(1))).result()          # the worker's own setting ...
            torch.set_num_threads(n)                                # ... and the default new threads inherit is the caller's again (its own count is n already)
            cls._pool = pool
        return cls._pool.submit(fn).result()


def _draw_conf(gumbel, num_samples: int, n: int, m: int, steps, num_steps: int, randomize_temperature: float) -> torch.Tensor:
    """The reference's per-step confidence noise (sampling.py:113-117): one ``Gumbel(0, 1).sample([B, n, m])`` from the CPU default generator per step,
    scaled by randomize_temperature * (1 - progress) (in the reference's order of the two products)."""
    return torch.stack([gumbel.sample((num_samples, n, m)) * randomize_temperature * (1 - (i + 1) / num_steps) for i in steps])


_COPY_STREAMS = {}


def _to_device_early(cpu: torch.Tensor, device) -> torch.Tensor:
    """Host -> device copy of a chunk's confidence noise that neither holds the host nor sits in the launch stream's order.  A pageable-memory copy
    (``.to(device)``) holds the HOST until the device has run everything enqueued before it -- the whole previous step chunk -- and the device then
    idles while the host enqueues this chunk (BASELINE configs[1] at batch 16: eight gaps of 0.4-0.9 ms per 126 ms run, tools/launch_gaps.py); a pinned
    asynchronous copy IN the launch stream still puts its own latency (0.1-0.3 ms through the copy engine) between two chunks.  So: pinned staging, the
    copy on a side stream (it runs while the previous chunk computes), the launch stream waits for its event; the pinned block and the device
    block are kept alive by torch's allocators (non_blocking copy / record_stream).  Same-box A/B, ms per run of configs[1]: batch 16 pageable 124.2,
    pinned in-stream 121.7, side stream 121.4; batch 64 332 / 330 / 328.5."""
    device = torch.device(device)
    if device.type != "cuda":
        return cpu.to(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    main = torch.cuda.current_stream(idx)
    side = _COPY_STREAMS.get(idx)
    if side is None:
        side = _COPY_STREAMS[idx] = torch.cuda.Stream(idx)
    with torch.cuda.stream(side):
        out = cpu.pin_memory().to(device, non_blocking=True)
    main.wait_stream(side)
    out.record_stream(main)
    return out


def draw_noise(num_samples: int, n: int, m: int, C_: int, num_steps: int, randomize_temperature: float,
               device: torch.device, step_begin: int = 0, step_end: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Noise for steps [step_begin, step_end) of a run (default: the whole run), drawn in the reference's per-generator order: consecutive
    chunks consume the generators exactly as one whole-run draw does.
    Returns exp_noise [steps, B*n*m, C] (device) and conf_noise [steps, B, n, m] (device) where
    conf_noise = gumbel * randomize_temperature * (1 - progress) (sampling.py:117)."""
    step_end = num_steps if step_end is None else step_end
    exp_noise = torch.empty((step_end - step_begin, num_samples * n * m, C_), dtype=torch.float32, device=device)
    for i in range(step_end - step_begin):
        exp_noise[i].exponential_(1)
    gumbel = torch.distributions.Gumbel(loc=0.0, scale=1.0)                     # python-float params => CPU draws
    conf = _DrawThread.run(lambda: _draw_conf(gumbel, num_samples, n, m, range(step_begin, step_end), num_steps, randomize_temperature))
    return exp_noise, _to_device_early(conf, device)


def step_chunks(num_samples: int, n: int, m: int, C_: int, num_steps: int):
    """[(begin, end)] such that one chunk's Exp(1) noise stays under NOISE_CHUNK_BYTES (the reference holds one step at a time; a whole
    256-step run at batch 100 would be 6.7 GB) and that the run has a one-step head followed by >= OVERLAP_CHUNKS chunks: ``mb_sample`` only
    enqueues work, so the host-side draws of a chunk (the reference's CPU Gumbel noise: ~1.5 ms per step at batch 64, 100 ms per 64-step run
    when drawn up front) run while the device is busy with the previous chunk -- only the head's draw is exposed."""
    per_step = num_samples * n * m * C_ * 4
    k = max(1, min(num_steps, NOISE_CHUNK_BYTES // max(per_step, 1)))
    if OVERLAP_CHUNKS > 1 and num_steps > 1:
        k = max(1, min(k, -(-(num_steps - 1) // OVERLAP_CHUNKS)))
        return [(0, 1)] + [(b, min(b + k, num_steps)) for b in range(1, num_steps, k)]
    return [(b, min(b + k, num_steps)) for b in range(0, num_steps, k)]


def run_chunked(model: "LFQBert", vqgan_model, labels: torch.Tensor, plan, randomize_temperature: float, **kw):
    """run_loop over the whole run with the noise drawn chunk by chunk (same random streams as one whole-run draw)."""
    scale = plan[0]
    steps = len(scale)
    B = labels.shape[0]
    n, m = model.seq_len, model.splits
    chunks = step_chunks(B, n, m, model.effective_codebook_size, steps)
    if len(chunks) == 1:
        e, c = draw_noise(B, n, m, model.effective_codebook_size, steps, randomize_temperature, model.device)
        return run_loop(model, vqgan_model, labels, plan, e, c, **kw)
    want_steps = kw.get("want_steps", True)
    parts, out = [], None
    kw = dict(kw, _cplan=plan_arrays(plan))              # the ctypes arrays of the plan are built once per run, not once per chunk
    for (b0, b1) in chunks:
        e, c = draw_noise(B, n, m, model.effective_codebook_size, steps, randomize_temperature, model.device, b0, b1)
        out = run_loop(model, vqgan_model, labels, plan, e, c, step_range=(b0, b1), **kw)
        if want_steps:
            parts.append(out[2])
    img, u8, _, codes = out
    return img, u8, (torch.cat(parts) if want_steps else None), codes


def plan_arrays(plan):
    """The plan as the ctypes arrays ``mb_sample`` reads (+ whether any step is guided)."""
    scale, temp, mask_len = plan
    nsteps = len(scale)
    use_cfg = any(s != 0.0 for s in scale) or getattr(plan, "force_guidance", False)
    return (C.c_float * nsteps)(*scale), (C.c_float * nsteps)(*temp), (C.c_int * nsteps)(*mask_len), use_cfg


def run_loop(model: LFQBert, vqgan_model: Optional[ConvVQModel], labels: torch.Tensor, plan, exp_noise: torch.Tensor,
             conf_noise: torch.Tensor, want_steps: bool = True, want_image: bool = True, want_u8: bool = False,
             step_range: Optional[Tuple[int, int]] = None, _cplan=None):
    """One ``mb_sample`` call.  -> (image or None, uint8 NHWC or None, step tokens [steps,B,n,m] or None, codes [B,n]).
    ``step_range`` = (begin, end): only those steps of the plan, with ``exp_noise`` / ``conf_noise`` holding that chunk's noise; chunk (0, e)
    starts the run, later chunks continue from the engine's token state, the chunk ending at the last step combines and decodes (image / codes
    are meaningful only then)."""
    dev = model._require_cuda("sample")
    scale, temp, mask_len = plan
    nsteps = len(scale)
    sb, se = step_range if step_range is not None else (0, nsteps)
    steps = se - sb
    if exp_noise.shape[0] != steps or conf_noise.shape[0] != steps:
        raise ValueError(f"noise holds {exp_noise.shape[0]} steps, the step range {steps}")
    B = labels.shape[0]
    n, m = model.seq_len, model.splits
    c_scale, c_temp, c_len, use_cfg = _cplan if _cplan is not None else plan_arrays(plan)
    labels = labels.to(device=dev, dtype=torch.int64).contiguous()
    step_tokens = torch.empty((steps, B, n, m), dtype=torch.int64, device=dev) if want_steps else None
    last = se == nsteps                                  # only the chunk that ends the run combines and decodes: earlier chunks need no outputs
    codes = torch.empty((B, n), dtype=torch.int64, device=dev) if last else None
    img = u8 = None
    hdec = None
    if last and vqgan_model is not None and (want_image or want_u8):
        side = int(round(n ** 0.5))
        res = side << (vqgan_model.num_resolutions - 1)
        if want_image:
            img = torch.empty((B, vqgan_model.num_channels, res, res), dtype=torch.float32, device=dev)
        if want_u8:
            u8 = torch.empty((B, res, res, vqgan_model.num_channels), dtype=torch.uint8, device=dev)
        hdec = vqgan_model.engine(B, side)
    hgen = model.engine(2 * B if use_cfg else B)
    cplan = _lib.SamplePlan(nsteps, 1 if use_cfg else 0, c_scale, c_temp, c_len, sb if step_range is not None else 0, se if step_range is not None else 0)
    ptr = lambda t: t.data_ptr() if t is not None else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mb_sample(hgen, hdec, C.byref(cplan), labels.data_ptr(), B, exp_noise.data_ptr(),
                                         conf_noise.data_ptr(), ptr(step_tokens), ptr(codes), ptr(img), ptr(u8),
                                         torch.cuda.current_stream().cuda_stream), "mb_sample")
    return img, u8, step_tokens, codes


@torch.no_grad()
def sample(
    model,
    vqgan_model,
    num_samples: int = 10,
    labels: Optional[torch.Tensor] = None,
    softmax_temperature: float = 1.0,
    randomize_temperature: float = 4.5,
    mask_schedule_strategy: Text = "linear",
    num_steps: int = 12,
    guidance_scale: float = 3.0,
    mask_token: int = 1024,
    patch_size: int = 16,
    guidance_annealing: Text = "none",
    use_sampling_annealing: bool = False,
    scale_pow: float = 4.0,
    codebook_size: int = 1024,
    codebook_splits: int = 1,
    use_tqdm: bool = False,
) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """Generate ``num_samples`` class-conditional images.  See the module docstring; arguments as in the
    reference (sampling.py:32-54).  ``use_tqdm`` is accepted and ignored (the loop runs on the device)."""
    if not isinstance(model, LFQBert):
        raise TypeError(f"sample() needs a maskbit_amd LFQBert generator, got {type(model).__name__}")
    if not isinstance(vqgan_model, ConvVQModel):
        raise TypeError(f"sample() needs a maskbit_amd ConvVQModel tokenizer, got {type(vqgan_model).__name__}")
    device = model.device
    model.eval()
    vqgan_model.eval()
    n, m = int(patch_size ** 2), int(codebook_splits)
    if n != model.seq_len or m != model.splits:
        raise ValueError(f"patch_size/codebook_splits ({patch_size}, {m}) do not match the generator ({model.seq_len} tokens, {model.splits} groups)")
    if mask_token != model.mask_token:
        raise ValueError(f"mask_token={mask_token} but the generator masks with {model.mask_token} (= 2**(bits/splits))")
    if 2 ** model.bits != codebook_size:
        raise ValueError(f"codebook_size={codebook_size} does not match the generator's 2**{model.bits}")
    if labels is None:
        # goldfish, chicken, tiger cat, hourglass, ship, dog, race car, airliner, teddy bear, random (sampling.py:60-63)
        labels = torch.LongTensor([1, 7, 282, 604, 724, 179, 751, 404, 850, int(torch.randint(0, 999, size=(1,)))] * (num_samples // 10))
    model._check_labels(labels)
    labels = labels.to(device)
    if labels.numel() != num_samples:
        raise ValueError(f"{labels.numel()} labels for num_samples={num_samples}")
    plan = build_plan(num_steps, n * m, guidance_scale, guidance_annealing, scale_pow, softmax_temperature,
                      use_sampling_annealing, mask_schedule_strategy)
    if guidance_scale != 0.0 and not any(s != 0.0 for s in plan[0]):
        plan = _ForcedPlan(plan)                      # CFG forward still runs when every a_i happens to be 0
    img, _, step_tokens, _ = run_chunked(model, vqgan_model, labels, plan, randomize_temperature)
    return img, list(step_tokens.unbind(0))


class _ForcedPlan(tuple):
    force_guidance = True

    def __new__(cls, plan):
        return super().__new__(cls, plan)
