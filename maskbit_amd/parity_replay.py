"""Replay of the REAL reference's full-size 64-step run (tests/golden/sample_full12_64.npz) through the HIP engine.

The fixture was written by ``oracle/make_golden.py full64``: the reference's own ``sample()`` (sampling.py:55-136) on the
12-bit generator, 64 steps, CFG 7.1 cosine, arccos schedule, CPU fp32, seed 1234, with the seeded synthetic weights of
``maskbit_amd/synth.py``.  Nothing here imports ``oracle/``: the weights come from the seeds, the noise from the seed (the
reference's draw order on a CPU model: per step one ``exponential_`` [B*n*m, C], then one Gumbel [B, n, m]), the expected
tokens and pixels from the fixture.  Used by ``tests/test_hip_full64.py`` / ``test_hip_configs.py`` (through the ``tests/parity_replay.py``
shim), by the tools and by ``bench.py`` (which reports the measured token mismatch of the mode it times).

  teacher_forced(): every step restarts from the reference's masked-token state; mismatches are counted over the positions
                    sampled at that step (84 284 in the run) -- the north star's "bit-token mismatch vs reference".  With batch = 64 the
                    fixture's samples ride inside a batch of the size bench.py times.
  free_running():   one ``mb_sample`` call over all 64 steps with the same noise; reports how far the trajectories drift.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np
import torch

# recorded runs of the reference (written by oracle/make_golden.py in the build container; data only): the repository's tests/golden/, or
# wherever MASKBIT_AMD_GOLDEN_DIR points
GOLDEN_DIR = os.environ.get("MASKBIT_AMD_GOLDEN_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
GOLDEN = os.path.join(GOLDEN_DIR, "sample_full12_64.npz")
# the other full-size runs of the reference (oracle/make_golden.py RUNS): BASELINE configs[1] and configs[4] with their own sampler settings
RUN_CFG1 = "sample_full10_16_nocfg"
RUN_CFG5 = "sample_full14_256"
RUN_C3_S2 = "sample_full12_64_s2"        # configs[2] again: other generator weights (seed, head gain), noise seed and labels
RUN_C3_S3 = "sample_full12_64_s3"        # ... and a third time at twice the batch (168 568 sampled positions)
RUN_CFG1_S2 = "sample_full10_16_nocfg_s2"   # second runs of configs[1] and configs[4] (other weights, head gain, noise, labels)
RUN_CFG5_S2 = "sample_full14_256_s2"
RUN_CFG5_S3 = "sample_full14_256_s3"     # round 4: configs[4] a third time at twice the batch (334 248 positions)
RUN_CFG5_S4 = "sample_full14_256_s4"     # round 5: ... and a fourth time (batch 4, other weights / head gain / noise / labels)
RUN_CFG1_S3 = "sample_full10_16_nocfg_s3"   # ... and configs[1] a third time
RUN_CFG1_S4 = "sample_full10_16_nocfg_s4"   # round 5, end: HELD-OUT runs of configs[1] / configs[2], recorded after the round's coverage decisions
RUN_C3_S4 = "sample_full12_64_s4"           # (which GEMM / which layers run the lo refinements) were frozen on the runs above
RUN_CFG5_S5 = "sample_full14_256_s5"        # ... and of configs[4] (batch 4)
RUN_C3_OUTLIER = "sample_full12_64_outlier"          # configs[2] / configs[1] on "trained-like" weights (synth._trained_like: heavy tails,
RUN_C3_OUTLIER_S2 = "sample_full12_64_outlier_s2"    # (round 5, end: a second trained-like 12-bit run, batch 8)
RUN_CFG1_OUTLIER = "sample_full10_16_nocfg_outlier"  # massive-activation channels)
RUN_C3_PRENORM = "sample_full12_64_prenorm"          # configs[2]'s sampler on the generator variants without a differential guided forward:
RUN_C3_SEQ1024 = "sample_full12_64_seq1024"          # use_prenorm=True, and the 512 x 512 models' 1024 + 1 tokens
RUN_C3_OUTLIER2 = "sample_full12_64_outlier2"        # round 6: a third trained-like 12-bit run, held out, of a heavier family (synth style "outlier2")
RUN_C3_OUTLIER2_S2 = "sample_full12_64_outlier2_s2"  # ... and a second one, recorded after the activation-lo coverage (the round's last precision decision) was frozen
RUN_C3_OUTLIER2_S3 = "sample_full12_64_outlier2_s3"  # (the second one is an easy run -- single fp16 1.5e-4 --: a third, head gain 12)
RUN_16BIT = "sample_full16_64"                       # round 6: the other shipped codebooks (README.md:74-75), each with its own yaml's sampler: 16-bit (C = 256 per group) ...
RUN_18BIT = "sample_full18_64"                       # ... and 18-bit (C = 512)
RUN_DEMO14 = "sample_full14_demo"                    # round 6: the demo's call site (demo_utils.py:139-157, configs/demo/demo.yaml): 14-bit, guidance 3.0 with annealing "none", 64 steps


def load_run(name: str = "sample_full12_64") -> Dict[str, object]:
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    steps = torch.from_numpy(z["steps"].astype(np.int64))                      # [S, B, n, 2] predicted tokens per step (n = 256, or 1024 for the 512 x 512 models)
    S, B, n = steps.shape[0], steps.shape[1], steps.shape[2]
    masks = torch.from_numpy(np.unpackbits(z["masks"], axis=1)[:, : B * n * 2].reshape(S, B, n, 2).astype(bool))
    kw = {str(k): str(v) for k, v in zip(z["kw_keys"], z["kw_vals"])}
    bits = int(z["bits"]) if "bits" in z.files else 12
    g = {"z": z, "name": name, "steps": steps, "masks": masks, "labels": torch.from_numpy(z["labels"].astype(np.int64)), "kw": kw,
         "seed": int(z["seed"]), "bits": bits, "C": 1 << (bits // 2), "n": n, "prenorm": bool(int(z["gen_prenorm"])) if "gen_prenorm" in z.files else False}
    if "codes" in z.files:
        g["codes"] = torch.from_numpy(z["codes"].astype(np.int64))
    return g


def load_full64() -> Dict[str, object]:
    return load_run("sample_full12_64")


def tokens_in(g, i: int) -> torch.Tensor:
    """The masked-token state the reference's model saw at step i."""
    if i == 0:
        return torch.full_like(g["steps"][0], g["C"])
    return torch.where(g["masks"][i], torch.full_like(g["steps"][0], g["C"]), g["steps"][i - 1])


def reference_noise(g, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """(exp_noise [S, B*n*2, C], conf_noise [S, B, n, 2]) exactly as the reference drew them (CPU default generator)."""
    S, B, n = g["steps"].shape[0], g["steps"].shape[1], g["steps"].shape[2]
    rt = float(g["kw"]["randomize_temperature"])
    torch.manual_seed(g["seed"])
    gum = torch.distributions.Gumbel(0.0, 1.0)
    qs, cs = [], []
    for i in range(S):
        qs.append(torch.empty(B * n * 2, g["C"]).exponential_(1))
        cs.append(gum.sample((B, n, 2)) * rt * (1 - (i + 1) / S))
    return torch.stack(qs).to(device), torch.stack(cs).to(device)


def plan_of(g):
    from maskbit_amd.sampling import build_plan
    kw = g["kw"]
    return build_plan(int(kw["num_steps"]), 2 * g["steps"].shape[2], float(kw["guidance_scale"]), kw["guidance_annealing"], float(kw["scale_pow"]), 1.0, False,
                      kw["mask_schedule_strategy"])


def build_models(device, with_tokenizer: bool = True, name: str = "sample_full12_64"):
    """The fixture's generator / tokenizer on the HIP engine (weights regenerated from the seeds, sha-checked)."""
    import hashlib
    from maskbit_amd import ConvVQModel, LFQBert, synth
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    bits = int(z["bits"]) if "bits" in z.files else 12
    style = str(z["gen_style"]) if "gen_style" in z.files else "gaussian"
    seq = int(z["gen_seq"]) if "gen_seq" in z.files else 256
    prenorm = bool(int(z["gen_prenorm"])) if "gen_prenorm" in z.files else False
    gsd = synth.make_generator_weights(synth.GenCfg(bits=bits, splits=2, seq=seq, prenorm=prenorm), seed=int(z["gen_seed"]), head_gain=float(z["head_gain"]),
                                       style=style)
    sha = lambda t: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()
    assert sha(gsd["transformer.layers.0.0.mha.in_proj_weight"]) == str(z["w_sha_in_proj0"]), "synthetic generator weights changed"
    gen = LFQBert(img_size=16 * int(round(seq ** 0.5)), hidden_dim=1024, codebook_size=2 ** bits, codebook_splits=2, depth=24, heads=16, mlp_dim=4096,
                  dropout=0.1, nclass=1000, input_stride=16, use_prenorm=prenorm)
    gen.load_state_dict(gsd, strict=True)
    gen = gen.eval().requires_grad_(False).to(device)
    tok = None
    if with_tokenizer:
        tcfg = synth.TokCfg(token_size=12)
        tsd = synth.make_tokenizer_weights(tcfg, seed=int(z["tok_seed"]))
        assert sha(tsd["decoder.conv_in.weight"]) == str(z["w_sha_conv_in"]), "synthetic tokenizer weights changed"

        class Cfg(dict):
            __getattr__ = dict.__getitem__
        tok = ConvVQModel(Cfg(quantizer_type="lookup-free", codebook_size=4096, token_size=12, num_channels=3, hidden_channels=128,
                              channel_mult=[1, 1, 2, 2, 4], num_resolutions=5, num_res_blocks=2, sample_with_conv=True))
        tok.load_state_dict(tsd, strict=False)
        tok = tok.eval().requires_grad_(False).to(device)
    return gen, tok


def _embed_rows(B: int, batch: int) -> torch.Tensor:
    """Rows of a `batch`-sample batch that carry the fixture's B samples: spread over the batch (first, last and in between), so that they ride in
    different pair tiles, CU rounds and XCD chunks of the persistent GEMM grids than they do at batch B."""
    if batch == B:
        return torch.arange(B)
    rows = torch.linspace(0, batch - 1, B).round().long()
    assert rows.unique().numel() == B
    return rows


@torch.no_grad()
def teacher_forced(gen, g=None, noise=None, batch: int = 0):
    """-> (mismatches, sampled positions, per-step mismatch counts, re-mask differences).
    batch > B (round 5): the fixture's B samples ride as rows of a `batch`-sample forward -- the size bench.py TIMES (64 pairs: 2 048 pair tiles walked
    persistently in 8 rounds by 256 workgroups) -- the other rows holding random codes in the same mask state under random labels; the sampling step
    then runs on the fixture's rows alone, with the fixture's noise.  The counts must equal the batch = B replay's exactly: a sequence pair's logits do
    not depend on its batch neighbours (tests/test_hip_timed_path.py asserts the bits)."""
    from maskbit_amd import _lib
    lib = _lib.load()
    g = g or load_full64()
    dev = gen.device
    q, c = noise if noise is not None else reference_noise(g, dev)
    scale, temp, mask_len = plan_of(g)
    S, B = g["steps"].shape[0], g["steps"].shape[1]
    NB = max(int(batch), B)
    rows = _embed_rows(B, NB).to(dev)
    y = g["labels"].to(dev)
    if NB > B:
        fill = torch.Generator().manual_seed(g["seed"] + 99)
        y_all = torch.randint(0, 1000, (NB,), generator=fill).to(dev)
        y_all[rows] = y
    per_step, remask = [], 0
    total = 0
    for i in range(S):
        tin_cpu = tokens_in(g, i)
        tin = tin_cpu.to(dev).contiguous()
        if NB > B:
            src = tin_cpu[torch.arange(NB) % B]                      # filler rows: a fixture row's mask state, random codes where it is decoded
            rnd = torch.randint(0, g["C"], src.shape, generator=fill)
            tall = torch.where(src == g["C"], src, rnd).to(dev)
            tall[rows] = tin
            tall = tall.contiguous()
        else:
            tall, y_all = tin, y
        if float(g["kw"]["guidance_scale"]) != 0.0 and (scale[i] != 0.0 or gen.resolved_precision() >= 4):   # (precision 4: mb_sample runs the zero-scale steps through the guided forward too)
            lg = gen.forward_cfg(tall, y_all)              # the guided forward of the loop (cond | label-dropped); as in mb_sample, the steps whose
                                                                     # annealed scale is exactly 0 run the conditional forward alone (c + 0 (c - u) == c)
            lc, lu = lg[:NB][rows].contiguous(), lg[NB:][rows].contiguous()
        else:
            lc, lu = gen(tall, y_all, torch.zeros(NB, dtype=torch.bool, device=dev))[rows].contiguous(), None
        tout, pred = torch.empty_like(tin), torch.empty_like(tin)
        _lib.check(lib.mb_sample_step(lc.data_ptr(), lu.data_ptr() if lu is not None else None, scale[i], temp[i], q[i].data_ptr(), c[i].data_ptr(),
                                      mask_len[i], tin.data_ptr(), tout.data_ptr(), pred.data_ptr(), B, g["steps"].shape[2], 2, g["C"],
                                      torch.cuda.current_stream().cuda_stream), "mb_sample_step")
        msk = g["masks"][i]
        per_step.append(int((pred.cpu() != g["steps"][i])[msk].sum()))
        total += int(msk.sum())
        if i + 1 < S:
            remask += int((tout.cpu() != tokens_in(g, i + 1)).sum())
    return sum(per_step), total, per_step, remask


@torch.no_grad()
def free_running(gen, tok, g=None, noise=None):
    """One mb_sample call with the reference's noise.  -> dict(step_mismatch=[per step fraction], codes_mismatch, pixel_max_err,
    u8_mean_abs_diff): how far the free-running trajectory drifts from the reference's (a flipped token changes every later step)."""
    from maskbit_amd.sampling import run_loop
    g = g or load_full64()
    dev = gen.device
    q, c = noise if noise is not None else reference_noise(g, dev)
    img, u8, steps, codes = run_loop(gen, tok, g["labels"], plan_of(g), q, c, want_u8=True)
    torch.cuda.synchronize()
    steps = steps.cpu()
    z = g["z"]
    out = {"step_mismatch": [float((steps[i] != g["steps"][i]).float().mean()) for i in range(steps.shape[0])],
           "codes_mismatch": float((codes.cpu() != g["codes"]).float().mean())}
    img = img.cpu()
    same = (codes.cpu() == g["codes"]).all(dim=1)                       # images whose final codes equal the reference's: pixels comparable
    errs = []
    for (yy, xx) in ((0, 0), (120, 120), (240, 240), (37, 201)):
        ref = torch.from_numpy(z[f"crop_{yy}_{xx}"])
        if bool(same.any()):
            errs.append(float((img[same][:, :, yy:yy + 16, xx:xx + 16] - ref[same]).abs().max()))
    out["images_with_identical_codes"] = int(same.sum())
    out["pixel_max_err_identical_codes"] = max(errs) if errs else None
    out["u8_mean_abs_diff"] = float((u8.cpu()[:, ::4, ::4].float() - torch.from_numpy(z["image_u8_q"]).float()).abs().mean())
    return out
