#!/usr/bin/env python
"""Headline benchmark: images/sec of 64-step MaskBit-12bit sampling with CFG + conv-VQGAN decode.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: per GPU, B=64 class-conditional images through
the whole of sample() -- noise draw (reference RNG protocol), 64 x [2B-sequence LFQBert forward, CFG,
softmax, categorical draw, Gumbel confidence, re-mask], token combine, conv-VQGAN decode to 256x256,
uint8 NHWC -- i.e. BASELINE.json configs[2] (C3).  With N>1 (launched by torch.distributed.run, one
process per GPU) every rank samples its own B=64 shard (weak scaling, no data-path collective) and
the uint8 images are all-gathered once per batch over RCCL.  Weights are random-init of the real
architecture (304.8 M-param generator, 29.4 M-param decoder), labels synthetic; no checkpoint or
dataset is available offline.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : dominant kernel (the fp16 MFMA GEMM family) algorithmic FLOPs / HIP-event duration
  cpu_baseline : the CPU oracle (oracle/, a parity-checked port of the reference) timed on the host
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16/fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
B_PER_GPU = 64
NUM_STEPS = 64
GEN = dict(img_size=256, hidden_dim=1024, codebook_size=4096, codebook_splits=2, depth=24, heads=16, mlp_dim=4096,
           dropout=0.1, use_prenorm=False, input_stride=16)
SAMPLER = dict(guidance_scale=7.1, guidance_annealing="cosine", scale_pow=3.0, randomize_temperature=8.2,
               mask_schedule_strategy="arccos", softmax_temperature=1.0)      # configs/generator/maskbit_generator_12bit.yaml
F_SEQ = 162.33e9                        # algorithmic FLOPs per 257-token sequence forward
F_DEC = 185.97e9                        # per decoded image
DEC_BYTES_IDEAL = 426e6                  # per decoded image: 213.06 M elements x 2 B
HBM_PEAK_GBS = 8000.0
PROF_EVERY = 8                           # HIP-event timing of every 8th forward of the timed region (forwards 4, 12, .. 60 of a 64-step run: all guided)
GEN_SEED, HEAD_GAIN, TOK_SEED = 100, 12.0, 200   # maskbit_amd/synth.py: the same synthetic checkpoints the golden fixtures were made with


class Cfg(dict):
    __getattr__ = dict.__getitem__


def tok_config():
    return Cfg(quantizer_type="lookup-free", codebook_size=4096, token_size=12, num_channels=3, hidden_channels=128,
               channel_mult=[1, 1, 2, 2, 4], num_resolutions=5, num_res_blocks=2, sample_with_conv=True)


def cpu_baseline():
    """Time the CPU oracle on a bounded sample of the same workload:
    2 complete CFG steps at B=8 (16 sequences) after 1 warm-up step, and one decode of 8 images."""
    from oracle import maskbit_oracle as O          # the ONLY use of oracle/ in this file: the thing timed here is the CPU baseline itself
    gcfg, tcfg = O.GenCfg(bits=12, splits=2), O.TokCfg(token_size=12)
    gsd = O.make_generator_weights(gcfg, seed=GEN_SEED, head_gain=HEAD_GAIN)
    tsd = O.make_tokenizer_weights(tcfg, seed=TOK_SEED)
    B = 8
    labels = (torch.arange(B) * 37) % 1000
    times = []
    fwd = lambda t, y, d: O.lfq_bert_forward(gsd, gcfg, t, y, d)
    # MKL oversubscribes badly on many-core hosts (measured on the 256-CPU GPU box: 16 threads 0.19 s/sequence,
    # 128 threads 0.77 s/sequence): probe a few thread counts on a 2-sequence forward and keep the fastest.
    default_threads = torch.get_num_threads()
    probe_t = torch.full((2, 256, 2), 64, dtype=torch.int64)
    best = (float("inf"), default_threads)
    for n in sorted({min(default_threads, c) for c in (8, 16, 32, default_threads)}):
        torch.set_num_threads(n)
        fwd(probe_t, labels[:2], None)
        t0 = time.perf_counter(); fwd(probe_t, labels[:2], None); dt = time.perf_counter() - t0
        best = min(best, (dt, n))
    torch.set_num_threads(best[1])

    def timed_fwd(t, y, d):
        t0 = time.perf_counter()
        out = fwd(t, y, d)
        times.append(time.perf_counter() - t0)
        return out

    torch.manual_seed(0)
    t0 = time.perf_counter()
    preds = O.sample_loop(timed_fwd, B, labels, num_steps=3, guidance_scale=7.1, guidance_annealing="cosine", scale_pow=3.0,
                          randomize_temperature=8.2, mask_schedule_strategy="arccos", mask_token=64, codebook_splits=2)
    loop_s = time.perf_counter() - t0
    step_s = (loop_s - times[0] - (loop_s - sum(times)) / 3) / 2          # 2 timed steps: forward + that step's sampling tail
    codes = O.combine_groups(preds[-1], 12, 2)
    t0 = time.perf_counter()
    O.decode_tokens(tsd, tcfg, codes)
    dec_s = time.perf_counter() - t0
    ips = 1.0 / (NUM_STEPS * step_s / B + dec_s / B)
    used = torch.get_num_threads()
    torch.set_num_threads(default_threads)
    return {"value": ips, "unit": "images/s", "cores": used, "kind": "port",
            "sample": f"oracle (PyTorch-CPU fp32 port, parity-pinned to the reference): mean of 2 full CFG steps at B={B} "
                      f"({step_s:.2f} s/step) + decode of {B} images ({dec_s:.2f} s), extrapolated to 64 steps; fastest of the probed "
                      f"thread counts = {used}",
            "host_cpus": os.cpu_count()}


def measured_traffic(dom: str, mode: str, avg_launch_us=None):
    """HBM traffic of the dominant trunk GEMM from the hardware counters, measured by THIS run: after the timed region, rocprofv3 --pmc sub-passes
    (counters need their own passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots") over
    tools/pair_one.py, which issues that GEMM exactly as the guided forward of the timed region does (64 sequence pairs, pair tiles, the timed mode's
    mini-tile operand sets), 3 launches per pass.  -> {bytes_per_launch, algorithmic_bytes, ratio, mfma_busy, ...} or (None, reason).
    bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) KiB: the gfx950 correction of the guide's HBM section (FETCH_SIZE reports half of wide coalesced
    streams; Infinity-Cache hits are counted too, so this is traffic INTO the L2s, an upper bound of HBM traffic).  mfma_busy =
    SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): the share of the launch during which a SIMD's matrix pipe is busy."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 is not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ):
        return None, "this run is itself being profiled (rocprofv3 environment present): no nested counter passes"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as P
    shape = dom.replace("gemm_", "")
    mini = {"strict": 1, "diff": 0, "fp16": 0}.get(mode, 0)
    env = dict(os.environ, PAIR_ONE_MINI=str(mini), TMPDIR="/tmp")
    got = {}
    tmp = tempfile.mkdtemp(prefix="mb_pmc_", dir="/tmp")
    try:
        for grp in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"):
            d = os.path.join(tmp, grp.split()[0])
            cmd = [exe, "--pmc", *grp.split(), "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--",
                   sys.executable, os.path.join(ROOT, "tools", "pair_one.py"), shape, "3"]
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            c = P.gemm_counters(d)
            if r.returncode != 0 or grp.split()[0] not in c:
                return None, f"rocprofv3 --pmc {grp}: exit {r.returncode}, counters {sorted(c)}: {r.stdout.decode(errors='replace')[-300:]}"
            got.update(c)
    except Exception as e:                               # noqa: BLE001  (context only: never costs the headline line)
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    by = P.hbm_bytes(got["FETCH_SIZE"][0], got["WRITE_SIZE"][0])
    alg = P.algorithmic_bytes(shape, mini)
    out = {"bytes_per_launch": by, "algorithmic_bytes": alg, "ratio": by / alg,
           "mfma_busy": (got["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024.0) / (got["GRBM_GUI_ACTIVE"][0] / 8.0) if "GRBM_GUI_ACTIVE" in got and got["GRBM_GUI_ACTIVE"][0] else None,
           "launches_per_pass": got["FETCH_SIZE"][1], "fetch_size_kib": got["FETCH_SIZE"][0], "write_size_kib": got["WRITE_SIZE"][0],
           "how": f"rocprofv3 --pmc sub-passes of this bench run over tools/pair_one.py {shape} (PAIR_ONE_MINI={mini}); (2 FETCH_SIZE + WRITE_SIZE) KiB, "
                  "the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md; Infinity-Cache hits included"}
    if avg_launch_us:
        out["implied_gb_per_s"] = by / (avg_launch_us * 1e-6) / 1e9
    return out, None


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run on
    127.0.0.1) and pass their output through; rank 0 prints the JSON line."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and "MB_BENCH_FORCE_DEVICE" not in os.environ:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) (torch.cuda.device_count()); nothing was launched")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def build_models(dev):
    from maskbit_amd import ConvVQModel, LFQBert, synth
    gen = LFQBert(**GEN)
    tok = ConvVQModel(tok_config())
    gen.load_state_dict(synth.make_generator_weights(synth.GenCfg(bits=12, splits=2), seed=GEN_SEED, head_gain=HEAD_GAIN), strict=True)
    tok.load_state_dict(synth.make_tokenizer_weights(synth.TokCfg(token_size=12), seed=TOK_SEED), strict=False)   # decoder half only
    return gen.eval().requires_grad_(False).to(dev), tok.eval().requires_grad_(False).to(dev)


def measured_parity(gen, run=None, batch=B_PER_GPU):
    """Teacher-forced token mismatch of the engine's CURRENT precision mode against a full-size 64-step run of the real reference: by default
    tests/golden/sample_full12_64.npz (made by oracle/make_golden.py full64 with the bench's own weights; 84 284 sampled positions); `run` names
    another recorded run (`gen` must then carry that run's weights).  Round 5: measured AT THE TIMED BATCH SIZE -- the fixture's 4 (8) samples ride as
    rows 0 .. 63 (spread) of a 64-sample guided forward whose other rows hold random codes in the same mask state, i.e. through the same 2 048-tile
    persistent GEMM walks, pair attention grid and LayerNorm launches the timed region runs (parity_replay.teacher_forced(batch = 64))."""
    from maskbit_amd import parity_replay as R
    g = R.load_run(run) if run is not None else R.load_full64()
    bad, tot, _, _ = R.teacher_forced(gen, g, R.reference_noise(g, gen.device), batch=batch)
    return {"token_mismatch": bad / tot, "mismatches": bad, "positions": tot, "batch": max(batch, int(g["steps"].shape[1])),
            "fixture_samples": int(g["steps"].shape[1]),
            "against": f"the reference's own sample() run, CPU fp32 (tests/golden/{run or 'sample_full12_64'}.npz), teacher-forced per step, the fixture's "
                       f"samples embedded in a batch of {max(batch, int(g['steps'].shape[1]))}"}


def other_runs_parity(dev, mode_settings):
    """The same measurement on the OTHER full-size 12-bit runs of the reference (other generator weights, head gain, noise seed and labels:
    tests/golden/sample_full12_64_s2.npz, 84 284 positions, and _s3.npz, batch 8, 168 568 positions) for each LFQBert.precision
    in `mode_settings` -> {mode: {run: parity, "pooled_with_first_run": ...}} (the first run's count is added by the caller's own measurement)."""
    from maskbit_amd import parity_replay as R
    out = {m: {} for m in mode_settings}
    for run in (R.RUN_C3_S2, R.RUN_C3_S3):
        gen, _ = R.build_models(dev, with_tokenizer=False, name=run)
        for name, prec in mode_settings.items():
            gen.precision = prec
            out[name][run] = measured_parity(gen, run)
        del gen
        torch.cuda.empty_cache()
    return out


def other_configs(dev):
    """One batch each of the other BASELINE configurations through the same sample() path (outside the timed region, N = 1): configs[1] =
    10-bit generator, 16 steps, no guidance, batch 16 (as named) and 64; configs[4]'s per-GPU shard = 14-bit, 256 steps, CFG 5.8 cosine, batch 32.
    Generators and sampler settings are those of the reference's own recorded runs (tests/golden/sample_full10_16_nocfg / sample_full14_256);
    decode to uint8 included.  -> {name: {images_per_s, ms_per_batch, batch, precision}}"""
    from maskbit_amd import ConvVQModel, synth
    from maskbit_amd import parity_replay as R
    from maskbit_amd.sampling import build_plan, run_chunked
    out = {}
    for tag, run, batches in (("configs[1] 10-bit/16 steps/no CFG", R.RUN_CFG1, (16, 64)), ("configs[4] shard 14-bit/256 steps/CFG 5.8", R.RUN_CFG5, (32,))):
        g = R.load_run(run)
        gen, _ = R.build_models(dev, with_tokenizer=False, name=run)
        bits = int(g["bits"])
        cfg = tok_config(); cfg["codebook_size"], cfg["token_size"] = 2 ** bits, bits
        tok = ConvVQModel(cfg)
        tok.load_state_dict(synth.make_tokenizer_weights(synth.TokCfg(token_size=bits), seed=7), strict=False)
        tok = tok.eval().requires_grad_(False).to(dev)
        kw = g["kw"]
        plan = build_plan(int(kw["num_steps"]), 512, float(kw["guidance_scale"]), kw["guidance_annealing"], float(kw["scale_pow"]), 1.0, False,
                          kw["mask_schedule_strategy"])
        rt = float(kw["randomize_temperature"])
        # (round 4 also timed an opt-out for unguided sampling -- the weight correction over single fp16 activations, 7.0e-4 instead of 5.3e-4 token
        # mismatch at 1.15-1.2x the speed; removed with the one-knob precision of round 5: the default carries the margin)
        for B in batches:
          for vtag in ("",):
            labels = (torch.arange(B) * 37 % 1000).to(dev)
            torch.manual_seed(0)
            reps = 3 if int(kw["num_steps"]) < 100 else 1
            for timed in (False, True):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps if timed else 1):
                    run_chunked(gen, tok, labels, plan, rt, want_steps=False, want_image=False, want_u8=True)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out[f"{tag}, batch {B}{vtag}"] = {"images_per_s": B / dt, "ms_per_batch": dt * 1e3, "batch": B,
                                              "precision": "LFQBert.precision = %d" % gen.resolved_precision()}
        del gen, tok
        torch.cuda.empty_cache()
    out.update(variant_configs(dev))
    return out


def variant_configs(dev):
    """The generator variants of SURVEY.md section 8(f) through the same sample() path with configs[2]'s sampler (64 steps, CFG 7.1 cosine, arccos),
    decode to uint8 included, one timed batch each after a warm-up batch: use_prenorm=True (bert.py:49-59,106-123), the embedding-table `Bert` class
    (bert.py:184-340), and the 1024 + 1-token generator of the 512 x 512 models (scripts/eval_maskbit.py:125,139-144) at batch 16 with a 32 x 32-latent
    decode to 512 x 512.  All three run the differential guided forward with the weight-correction mini-tiles (resolved precision in the entry)."""
    from maskbit_amd import Bert, ConvVQModel, LFQBert, synth
    from maskbit_amd.sampling import build_plan, run_chunked
    out = {}
    tok = ConvVQModel(tok_config())
    tok.load_state_dict(synth.make_tokenizer_weights(synth.TokCfg(token_size=12), seed=TOK_SEED), strict=False)
    tok = tok.eval().requires_grad_(False).to(dev)
    for tag, cls, kw, gcfg, B, style in (
            ("variant: use_prenorm=True, 12-bit/64 steps/CFG 7.1", LFQBert, dict(use_prenorm=True), synth.GenCfg(bits=12, splits=2, prenorm=True), 64, "gaussian"),
            ("variant: Bert (embedding tables, tied head), 12-bit/64 steps/CFG 7.1", Bert, dict(), synth.GenCfg(bits=12, splits=2, kind="bert"), 64, "gaussian"),
            ("variant: 1024 + 1 tokens (512 x 512 models), 12-bit/64 steps/CFG 7.1", LFQBert, dict(img_size=512), synth.GenCfg(bits=12, splits=2, seq=1024), 16, "gaussian"),
            # the ESCALATED mode (round 6): configs[2]'s workload on a heavy-tailed ("trained-like") checkpoint, which the auto mode escalates from its own
            # statistics to precision 4 (activation-lo mini-tiles in out-proj, FFN-up and FFN-down of every layer): what that mode costs against the headline's precision 2
            ("escalated: trained-like (heavy-tailed) 12-bit checkpoint, auto precision, 64 steps/CFG 7.1", LFQBert, dict(), synth.GenCfg(bits=12, splits=2), 64, "outlier")):
        try:
            gen = cls(**dict(GEN, **kw))
            gen.load_state_dict(synth.make_generator_weights(gcfg, seed=GEN_SEED + 1, head_gain=HEAD_GAIN, style=style), strict=True)
            gen = gen.eval().requires_grad_(False).to(dev)
            plan = build_plan(NUM_STEPS, 2 * gen.seq_len, SAMPLER["guidance_scale"], SAMPLER["guidance_annealing"], SAMPLER["scale_pow"],
                              SAMPLER["softmax_temperature"], False, SAMPLER["mask_schedule_strategy"])
            labels = (torch.arange(B) * 37 % 1000).to(dev)
            torch.manual_seed(0)
            for timed in (False, True):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, u8, _, _ = run_chunked(gen, tok, labels, plan, SAMPLER["randomize_temperature"], want_steps=False, want_image=False, want_u8=True)
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[f"{tag}, batch {B}"] = {"images_per_s": B / dt, "ms_per_batch": dt * 1e3, "batch": B, "image": list(u8.shape[1:3]),
                                        "precision": "LFQBert.precision = %d" % gen.resolved_precision()}
            del gen
        except Exception as e:                              # noqa: BLE001  (context only)
            out[f"{tag}, batch {B}"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="images per GPU per step (default: the C3 batch, 64)")
    ap.add_argument("--mode", choices=("strict", "diff", "fp16"), default="strict",
                    help="precision mode of the TIMED region (LFQBert.precision): strict (default; the product default: differential guidance + MX-fp4 "
                         "correction mini-tiles for the weights' fp16 rounding on every trunk GEMM + hi/lo head weights: 5.5e-4 token mismatch over three "
                         "reference runs), diff (differential guidance alone: faster, AT the 1e-3 bound) or single fp16 (independent streams: 1.4e-3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the HIP-event kernel timing (roofline becomes null)")
    ap.add_argument("--no-modes", action="store_true", help="skip the extra (untimed-region) measurements: parity replay and the other precision mode")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc sub-passes that measure roofline.traffic (N = 1 only; about a minute)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # MB_BENCH_FORCE_DEVICE / MB_BENCH_BACKEND exist only to exercise the N>1 code path on a 1-GPU box
    # (tests/test_hip_bench.py: two ranks share cuda:0 over gloo); the driver's runs use one GPU per rank over RCCL.
    dev_index = int(os.environ.get("MB_BENCH_FORCE_DEVICE", local_rank))
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{dev_index} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MB_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # communicator warm-up: a 1-element all-reduce builds the RCCL rings (seconds on the first collective) outside every timed region, and what
        # it returns is the number of ranks that actually answered -- `ranks_seen` in the JSON line comes from here, not from the environment
        probe = torch.ones(1, device=dev if backend == "nccl" else "cpu", dtype=torch.int32)
        dist.all_reduce(probe)
        ranks_seen = int(probe.item())
        if ranks_seen != world:
            raise SystemExit(f"bench.py: {ranks_seen} of {world} ranks answered the warm-up all-reduce")

    from maskbit_amd import _lib
    from maskbit_amd.parallel import gather_images
    from maskbit_amd.sampling import build_plan, run_chunked

    B = args.batch
    gen, tok = build_models(dev)
    MODES = {"strict": -1, "diff": 1, "fp16": 0}          # LFQBert.precision; -1 = the product default (resolves to 2 for this generator)
    gen.precision = MODES[args.mode]
    torch.manual_seed(1234 + rank)
    plan = build_plan(NUM_STEPS, 512, SAMPLER["guidance_scale"], SAMPLER["guidance_annealing"], SAMPLER["scale_pow"],
                      SAMPLER["softmax_temperature"], False, SAMPLER["mask_schedule_strategy"])

    gather_ev = []                                      # (start, end) event pairs around the one collective of a batch

    def one_batch(i: int):
        labels = ((torch.arange(B) + (rank * B + i * world * B)) * 37 % 1000).to(dev)
        # the product's own path (sample() / generate_uint8()): noise drawn chunk by chunk in the reference's generator order, overlapped with the loop
        _, u8, _, _ = run_chunked(gen, tok, labels, plan, SAMPLER["randomize_temperature"], want_steps=False, want_image=False, want_u8=True)
        if world == 1:
            return u8
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out_ = gather_images(u8, equal=True)            # every rank holds B images: one collective per batch, no size exchange
        e1.record()
        gather_ev.append((e0, e1))
        return out_

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        one_batch(i)
    fence()
    gather_ev.clear()
    if not args.no_prof:
        _lib.prof_enable(True, every=PROF_EVERY)        # sampled: the event pairs themselves cost 3-5 % when every launch carries them
    from maskbit_amd.telemetry import ClockSampler
    with ClockSampler(dev_index) as clocks:             # shader clock / socket power while the timed region runs (the part is power-capped: boxes differ)
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = one_batch(args.warmup + i)
        fence()
        elapsed = time.perf_counter() - t0
    telemetry = clocks.summary()
    prof = {}
    if not args.no_prof:
        prof = _lib.prof_read()
        _lib.prof_enable(False)
    gather_ms = None
    if world > 1:
        # time of the batch's one collective on this rank's stream (it includes waiting for the slowest rank to arrive), max over ranks
        gms = sum(a.elapsed_time(b) for a, b in gather_ev) / max(1, len(gather_ev))
        t = torch.tensor([elapsed, gms], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, gather_ms = float(t[0].item()), float(t[1].item())
    assert out.shape[0] == B * world and out.dtype == torch.uint8
    # Outside the timed region, N = 1 only: (1) the MEASURED token mismatch of the timed mode against the reference's own full-size run,
    # (2) the same workload and the same parity measurement in the other precision mode, one batch.
    modes = None
    if world == 1 and not args.no_modes:
        modes = {args.mode: {"images_per_s": B * world * args.steps / elapsed, "timed": True, "parity": measured_parity(gen)}}
        # the two faster modes below the default (differential guidance without the weight-correction pass; single fp16 with independent streams),
        # one untimed-region batch each -- the first with the kernel timers on, so that the line also carries the dominant GEMM without the pass
        for other in [m for m in ("diff", "fp16", "strict") if m != args.mode][:2]:
            gen.precision = MODES[other]
            one_batch(10_000); torch.cuda.synchronize()
            if other == "diff" and not args.no_prof:
                _lib.prof_enable(True, every=PROF_EVERY)
            ts = time.perf_counter()
            one_batch(10_001); torch.cuda.synchronize()
            ips = B / (time.perf_counter() - ts)
            pr = {}
            if other == "diff" and not args.no_prof:
                pr = _lib.prof_read()
                _lib.prof_enable(False)
            modes[other] = {"images_per_s": ips, "timed": False, "parity": measured_parity(gen)}
            if "gemm_ffn_up" in pr:
                c_, ms_ = pr["gemm_ffn_up"]
                modes[other]["ffn_up_avg_launch_us"] = ms_ / c_ * 1e3
                modes[other]["ffn_up_frac_of_peak"] = 2.0 * (2 * B * 257) * 4096 * 1024 / (ms_ / c_ * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS
        gen.precision = MODES[args.mode]
        try:                                                    # the other full-size reference runs (context only: never costs the headline line)
            for name, par in other_runs_parity(dev, {m: MODES[m] for m in modes}).items():
                modes[name]["parity_other_runs"] = par
        except Exception as e:                                  # noqa: BLE001
            modes["parity_other_runs_error"] = repr(e)
    others = None
    if world == 1 and not args.no_modes and B == B_PER_GPU:
        try:
            others = other_configs(dev)
        except Exception as e:                              # noqa: BLE001  (context only: never costs the headline line)
            others = {"error": repr(e)}

    if rank == 0:
        total_images = B * world * args.steps
        value = total_images / elapsed
        M = 2 * B * 257
        gemm_flops = {"gemm_qkv": 2.0 * M * 3072 * 1024, "gemm_attn_out": 2.0 * M * 1024 * 1024,
                      "gemm_ffn_up": 2.0 * M * 4096 * 1024, "gemm_ffn_down": 2.0 * M * 1024 * 4096}
        kernels = {k: {"calls": c, "avg_us": ms / c * 1e3, "total_ms": ms} for k, (c, ms) in prof.items()}
        roofline = None
        if prof:
            dom = max(gemm_flops, key=lambda k: prof.get(k, (0, 0.0))[1])
            calls, ms = prof[dom]
            achieved = gemm_flops[dom] / (ms / calls * 1e-3) / 1e12
            fam_flops = sum(gemm_flops[k] * prof[k][0] for k in gemm_flops if k in prof)
            fam_ms = sum(prof[k][1] for k in gemm_flops if k in prof)
            # `traffic` (HBM bytes per launch from the PMC counters) cannot be collected inside a timing run -- counters need their own rocprofv3 --pmc
            # passes -- so it is measured AFTER the timed region by sub-passes of this same run (measured_traffic; N = 1, rocprofv3 on PATH); the last
            # committed counter pass of the same kernel in the same mode is quoted beside it as a REFERENCE either way.
            traffic, traffic_why = (None, "skipped (--no-traffic, N > 1 or a non-default batch)")
            if world == 1 and not args.no_traffic and B == B_PER_GPU:
                torch.cuda.synchronize()
                traffic, traffic_why = measured_traffic(dom, args.mode, ms / calls * 1e3)
            traffic_ref = None
            for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                    if B == B_PER_GPU and pmc.get("_mode", "").split(" ")[0] == args.mode:
                        traffic_ref = {"bytes_per_launch": pmc[dom.replace("gemm_", "")]["hbm_bytes_corrected"], "source": f"profiles/{name}",
                                       "what": "2*FETCH_SIZE + WRITE_SIZE of a committed rocprofv3 --pmc pass of this kernel: NOT measured in this run"}
                        break
                except Exception:
                    pass
            roofline = {"bound": "mfma", "kernel": f"gemm_ht_kernel ({dom}: M={M}, N={4096 if dom == 'gemm_ffn_up' else (3072 if dom == 'gemm_qkv' else 1024)}, "
                                                   f"K={4096 if dom == 'gemm_ffn_down' else 1024})",
                        "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                        "traffic": traffic, "traffic_unmeasured_because": traffic_why, "traffic_reference": traffic_ref,
                        "flops_per_launch": gemm_flops[dom], "avg_launch_us": ms / calls * 1e3,
                        "gemm_family_tflops": fam_flops / (fam_ms * 1e-3) / 1e12,
                        # executed sequence-forwards per image: two per guided step, one where the annealed scale is exactly 0 (the loop skips the
                        # unconditional forward there: c + 0 (c - u) == c)
                        "seq_forwards_per_image": sum(2 if a != 0.0 else 1 for a in plan[0]),
                        "end_to_end_frac": value / world * (sum(2 if a != 0.0 else 1 for a in plan[0]) * F_SEQ + F_DEC) / (MFMA_BF16_PEAK_TFLOPS * 1e12),
                        # the shader clock this run sustained (telemetry) and the same fraction against the peak AT that clock (2.4 GHz behind the nominal figure)
                        "effective_clock_mhz": telemetry.get("effective_clock_mhz"),
                        "frac_at_effective_clock": (achieved / (MFMA_BF16_PEAK_TFLOPS * telemetry["effective_clock_mhz"] / 2400.0)
                                                    if telemetry.get("effective_clock_mhz") else None)}
            # the two other rooflines the north star names: attention core on MFMA, decoder on HBM
            if "attention" in prof:
                c_, ms_ = prof["attention"]
                fl = 2 * B * 16 * 4.0 * 257 * 257 * 64           # QK^T + PV per launch (nb = 2B sequences x 16 heads)
                roofline["attention"] = {"bound": "mfma", "achieved": fl / (ms_ / c_ * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": fl / (ms_ / c_ * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                         "flops_per_launch": fl, "avg_launch_us": ms_ / c_ * 1e3}
            if "decode" in prof:
                c_, ms_ = prof["decode"]
                by = B * DEC_BYTES_IDEAL                         # ideal-fusion 16-bit NHWC traffic per decode call
                roofline["decoder"] = {"bound": "hbm", "achieved": by / (ms_ / c_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": by / (ms_ / c_ * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_call": by,
                                       "mfma_frac": B * F_DEC / (ms_ / c_ * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "avg_call_ms": ms_ / c_}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        line = {
            "metric": "images/sec (256x256, 64-step MaskBit-12bit, CFG)", "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] (C3): MaskBit-Generator 12-bit, 64 steps, CFG 7.1 cosine, arccos schedule, "
                                   f"batch {B}/GPU, conv_vqgan decode to 256x256 uint8" + (", RCCL all-gather of images" if world > 1 else ""),
                       "global_batch": B * world, "parallelism": f"dp{world} (batch shards, one process per GPU)"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
            # shader clock and socket power sampled during the timed region (maskbit_amd/telemetry.py): the nominal peaks assume 2.4 GHz; random fp16
            # data on all 256 CUs holds 1.9-2.0 GHz under the power cap, and boxes differ -- read every fraction above next to this clock
            "telemetry": telemetry,
            "precision": {"timed_mode": args.mode,
                          "strict": "the product default: fp16 MFMA, fp32 accumulate, classifier-free guidance in differential form (the unconditional "
                                    "stream's GEMM operands carried as fp16(x_u - x_c) next to fp16(x_c): operand rounding cancels in c - u) + MX-fp4 "
                                    "correction mini-tiles for the fp16 rounding of the weights on the conditional half of every trunk GEMM (and on every row of "
                                    "the plain forward of the zero-scale steps) + hi/lo head weights (LFQBert.precision = 2)",
                          "diff": "the differential form without the correction mini-tiles (LFQBert.precision = 1): ~1.17x the default's speed; its token "
                                  "mismatch over the three reference runs is ~1e-3, AT the bound (round 2's default)",
                          "fp16": "single fp16 operands, independent streams (LFQBert.precision = 0)"},
            "precision_modes": modes,
            "other_configs": others,
            "ranks_seen": ranks_seen if world > 1 else 1, "backend": dist.get_backend() if world > 1 else None,
            "gather_ms": gather_ms,       # N > 1: the batch's one all-gather of uint8 images incl. the wait for the slowest rank (max over ranks); part of ms_per_step
            "kernels_note": f"HIP events on the launch stream inside the timed region; generator kernels sampled on every {PROF_EVERY}th guided forward "
                            f"(names as they are) and every {PROF_EVERY}th plain forward (the zero-scale steps: names + '.plain'), counted separately",
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
