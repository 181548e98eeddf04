"""GPU parity for the other BASELINE configurations: 10-bit (C=32), 14-bit (C=128), 18-bit (C=512) generators,
no-CFG sampling, and BASELINE configs[1] (10-bit, 16 steps, no CFG) teacher-forced at full size."""
import os

import pytest
import torch

from hip_helpers import hip_generator, token_mismatch
from oracle import maskbit_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _teacher_forced(cfg, sd, model, B, N, labels, seed, **kw):
    from maskbit_amd import _lib
    lib = _lib.load()
    C_, m = cfg.group_codes, cfg.splits
    rec = []
    torch.manual_seed(seed)
    O.sample_loop(lambda t, yy, dd: O.lfq_bert_forward(sd, cfg, t, yy, dd), B, labels, num_steps=N, mask_token=C_,
                  codebook_splits=m, record=rec, **kw)
    if os.environ.get("MB_TEST_ALSO_FP16"):              # context: the single-fp16 mode on the same oracle run
        model.precision = 0
        r0 = _replay(lib, _lib, cfg, model, rec, B, labels, kw)
        model.precision = -1
        print(f"  [single fp16: mismatch {r0[0]:.2e} ({r0[2]}/{r0[3]}), mean |logit err| {r0[1]:.4f}]")
    r = _replay(lib, _lib, cfg, model, rec, B, labels, kw)
    print(f"  [default precision: {r[2]}/{r[3]} mismatches]")
    return r[0], r[1]


def _replay(lib, _lib, cfg, model, rec, B, labels, kw):
    C_, m = cfg.group_codes, cfg.splits
    cfgd = kw.get("guidance_scale", 3.0) != 0.0
    drop = torch.cat([torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)]).to(DEV)
    bad = tot = 0
    max_logit_err = 0.0
    for i, r in enumerate(rec):
        tin = r.tokens_in.to(DEV).contiguous()
        if cfgd:
            lg = model.forward_cfg(tin, labels.to(DEV))          # the guided forward of the loop
            lc, lu = lg[:B].contiguous(), lg[B:].contiguous()
        else:
            lc, lu = model(tin, labels.to(DEV), torch.zeros(B, dtype=torch.bool, device=DEV)), None
        max_logit_err = max(max_logit_err, float((lc.cpu() - r.logits_c).abs().mean()))
        tout, pred = torch.empty_like(tin), torch.empty_like(tin)
        qn, cn = r.exp_noise.to(DEV).contiguous(), r.conf_noise.to(DEV).contiguous()
        k = int(torch.floor(torch.tensor(r.mask_ratio) * (256 * m)))
        _lib.check(lib.mb_sample_step(lc.data_ptr(), lu.data_ptr() if lu is not None else None, r.scale, 1.0, qn.data_ptr(), cn.data_ptr(), k,
                                      tin.data_ptr(), tout.data_ptr(), pred.data_ptr(), B, 256, m, C_, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        msk = r.tokens_in == C_
        bad += int((pred.cpu() != r.pred)[msk].sum())
        tot += int(msk.sum())
    return bad / tot, max_logit_err, bad, tot


@pytest.mark.parametrize("bits,splits", [(10, 2), (14, 2), (18, 2), (12, 4), (12, 3), (8, 1), (10, 1), (12, 1)])
def test_other_bit_widths_tiny(bits, splits):
    """splits = 2: C = 32 (half a wave per row), 128 and 512 (2 / 8 logits per lane), head N = 64 / 256 / 1024.
    Other codebook_splits: 4 groups of C = 8, 3 of 16, and single-group codebooks of 256 / 1024 (the
    reference's default constructor arguments) / 4096 codes (16 / 64 logits per lane in the step kernel)."""
    cfg = O.GenCfg(bits=bits, splits=splits, hidden=128, depth=2, heads=4, mlp=256, seq=256, nclass=10)
    sd = O.make_generator_weights(cfg, seed=40 + bits, head_gain=20.0)
    model = hip_generator(cfg, sd)
    g = torch.Generator().manual_seed(bits)
    toks = torch.randint(0, cfg.group_codes + 1, (3, 256, splits), generator=g)     # includes mask tokens
    labels = torch.tensor([0, 5, 9])
    out = model(toks.to(DEV), labels.to(DEV), torch.tensor([False, True, False], device=DEV))
    ref = O.lfq_bert_forward(sd, cfg, toks, labels, torch.tensor([False, True, False]))
    assert out.shape == (3, 256, splits, cfg.group_codes)
    assert float((out.cpu() - ref).norm() / ref.norm()) < 2e-3
    mism, _ = _teacher_forced(cfg, sd, model, 3, 4, labels, 7, guidance_scale=3.0, guidance_annealing="cosine", scale_pow=2.5,
                              randomize_temperature=7.5, mask_schedule_strategy="arccos")
    assert mism < 1e-2          # tiny peaky models (head gain 20) flip easily; the step itself is bit-exact (test_hip_parity)


def _vs_reference_run(name, modes):
    """Teacher-forced replay of one of the REAL reference's full-size runs (tests/golden/<name>.npz, oracle/make_golden.py RUNS) in the given
    engine modes (tag, LFQBert.precision) -> {tag: (mismatches, positions)}."""
    from maskbit_amd import parity_replay as R
    g = R.load_run(name)
    gen, _ = R.build_models(DEV, with_tokenizer=False, name=name)
    noise = R.reference_noise(g, gen.device)
    out = {}
    for tag, prec in modes:
        gen.precision = prec
        bad, tot, per, _ = R.teacher_forced(gen, g, noise)
        nb = max(1, len(per) // 8)
        print(f"{name} [{tag}]: {bad}/{tot} = {bad / tot:.2e}; per eighth of the run {[sum(per[i:i + nb]) for i in range(0, len(per), nb)]}")
        out[tag] = (bad, tot)
    return out


@pytest.mark.timeout(900)
def test_baseline_config1_10bit_16steps_nocfg_vs_reference_runs():
    """BASELINE configs[1] as named -- 10-bit generator, 16 steps, no guidance, batch 16 -- against THREE runs of the real reference (other weights,
    head gain, noise and labels; 87 040 sampled positions each).  Without guidance the plain forward runs; its product default (the MX-fp4
    weight-correction mini-tiles on every trunk GEMM, the LayerNorm output in front of FFN-up as an fp16 hi + lo pair, hi/lo head weights) measures
    6.4e-4 / 4.6e-4 / 5.5e-4 = 5.5e-4 over all.  (Round 4 swept the lo halves in QKV as well: 5.2e-4 / 4.3e-4 / 6.4e-4 = 5.3e-4, the same within the
    counts' spread for 8-12 % of this configuration's time; no lo sweep at all: 8.2e-4 / 5.1e-4 / 7.7e-4 = 7.0e-4; single fp16 1.06e-3 / 7.1e-4;
    profiles/raw/r05/xlo_mask.log.)  Asserted: <= 7e-4 on each run and <= 6e-4 over all."""
    from maskbit_amd import parity_replay as R
    tb = tt = 0
    for name in (R.RUN_CFG1, R.RUN_CFG1_S2, R.RUN_CFG1_S3):
        r = _vs_reference_run(name, [("product default", -1), ("hi + lo FFN-up operand alone (precision 1)", 1), ("single fp16", 0)])
        bad, tot = r["product default"]
        assert tot == 87040 and bad / tot <= 7e-4, name
        tb += bad; tt += tot
    print(f"configs[1], three reference runs, product default: {tb}/{tt} = {tb / tt:.2e}")
    assert tb / tt <= 6e-4


@pytest.mark.timeout(1500)
def test_baseline_config5_14bit_256steps_vs_reference_runs():
    """BASELINE configs[4]'s generator and sampler as named -- 14-bit (C = 128 per group), 256 steps, CFG 5.8 cosine (configs/generator/
    maskbit_generator_14bit_256steps.yaml:38-44) -- against FOUR 256-step runs of the real reference (B = 2, 2, 4 and 4: 1 002 744 sampled positions).
    MEASURED: the differential form alone misses 1e-3 here (1.4e-3; single fp16: 2.0e-3); with the weight-correction mini-tiles and hi/lo head
    weights (precision 2) 5.9e-4 over the four runs with one run AT 1.0e-3; the product default at 7 bits per group (precision 3: + the activation-lo
    mini-tiles -- since round 6 in out-proj and FFN-up of every layer, chosen on these runs: profiles/r06_coverage.md) 3.8e-4 over all, every run <= 5.5e-4
    (rounds 4-5, FFN-up only: 5.1e-4 / 8.7e-4).  Asserted without allowance: <= 7e-4 on each run, <= 5e-4 over all."""
    from maskbit_amd import parity_replay as R
    r = _vs_reference_run(R.RUN_CFG5, [("product default", -1), ("weight correction + activation-lo pass", 3),
                                       ("weight correction alone", 2), ("differential operands only", 1), ("single fp16", 0)])
    bad, tot = r["differential operands only"]
    assert tot == 167124 and bad / tot <= 2e-3
    assert r["product default"] == r["weight correction + activation-lo pass"]          # what the default resolves to at 7 bits per group
    assert r["weight correction alone"][0] / r["weight correction alone"][1] <= 1e-3
    tb, tt = r["product default"]
    assert tb / tt <= 7e-4
    for name in (R.RUN_CFG5_S2, R.RUN_CFG5_S3, R.RUN_CFG5_S4):
        bad, tot = _vs_reference_run(name, [("product default", -1)])["product default"]
        assert bad / tot <= 7e-4, name
        tb += bad; tt += tot
    print(f"configs[4], four reference runs, product default: {tb}/{tt} = {tb / tt:.2e}")
    assert tt == 1002744 and tb / tt <= 5e-4


@pytest.mark.timeout(900)
def test_trained_like_weights_heavy_tails_and_massive_activation_channels():
    """Parity under the statistics trained checkpoints show and Gaussian draws do not (maskbit_amd/synth.py _trained_like: heavy-tailed Linear
    weights, six hidden channels carrying massive LayerNorm outputs in every layer): configs[2] (12-bit, 64 steps, CFG 7.1) and configs[1] (10-bit,
    16 steps, no guidance) run by the REAL reference on such weights (oracle/make_golden.py RUNS: *_outlier), replayed in the product default.
    Per-(row, 64 columns) scales keep an outlier channel from costing the resolution of the rest of its row; the fp16 stores of the trunk never
    clamp (mb_gen_saturation_count)."""
    from maskbit_amd import parity_replay as R
    # Round 5 closed with a KNOWN GAP here: the second 12-bit trained-like run measured 179 / 168 568 = 1.06e-3 at precision 2 -- over the north star's
    # 1e-3 -- and was asserted at 1.2e-3.  Round 6: the auto mode escalates such checkpoints from their own statistics (LFQBert.weight_statistics: pooled
    # row kurtosis 10.9, a LayerNorm channel at ~16x the median) to precision 4 -- activation-lo mini-tiles on every trunk GEMM, weight-error scales per
    # (row, 128 columns) -- and EVERY run is asserted at the north star's 1e-3 again (precision 2 printed beside it as context).
    pooled = [0, 0]
    for name, bound in ((R.RUN_C3_OUTLIER, 1e-3), (R.RUN_C3_OUTLIER_S2, 1e-3), (R.RUN_CFG1_OUTLIER, 1e-3)):
        g = R.load_run(name)
        gen, _ = R.build_models(DEV, with_tokenizer=False, name=name)
        assert gen.weight_statistics()["heavy_tailed"] and gen.resolved_precision() == 4
        noise = R.reference_noise(g, gen.device)
        out = {}
        for tag, prec in (("product default", -1), ("precision 2 (round 5's default here)", 2), ("single fp16", 0)):
            gen.precision = prec
            bad, tot, per, _ = R.teacher_forced(gen, g, noise)
            out[tag] = (bad, tot)
            print(f"{name} [{tag}, resolves to {gen.resolved_precision()}]: {bad}/{tot} = {bad / tot:.2e}")
        sat = gen.saturation_count()
        print(f"{name}: fp16 saturation count {sat}")
        assert sat == 0
        bad, tot = out["product default"]
        assert bad / tot <= bound, name
        assert bad <= out["single fp16"][0] and bad <= out["precision 2 (round 5's default here)"][0]
        if name != R.RUN_CFG1_OUTLIER:
            pooled[0] += bad; pooled[1] += tot
        del gen
        torch.cuda.empty_cache()
    print(f"trained-like 12-bit runs pooled: {pooled[0]}/{pooled[1]} = {pooled[0] / pooled[1]:.2e}")
    assert pooled[0] / pooled[1] <= 1e-3


@pytest.mark.timeout(900)
def test_held_out_trained_like_run_of_a_heavier_family():
    """HELD OUT (round 6): a third trained-like 12-bit run of the real reference, recorded after precision 4 and its escalation rule were built and measured on
    the two runs above -- another seed and a HEAVIER family (maskbit_amd/synth.py style "outlier2": weight kurtosis 17.2 instead of 10.9, ten
    massive-activation channels instead of six; batch 8: 168 568 sampled positions per run).  The auto mode must escalate each from its statistics and meet
    the north star's 1e-3; precision 2 and single fp16 are printed as context.  Three runs: the first (head gain 14) took part in the round's last precision
    decision (the activation-lo coverage); the other two (head gains 16 / 12, other seeds) were recorded after it was frozen -- and turned out easy (single fp16
    1.5e-4 / 1.9e-4): reported as they are."""
    from maskbit_amd import parity_replay as R
    # (the second run of the family was recorded after the round's last precision decision -- the activation-lo coverage, profiles/r06_coverage.md, a study the
    #  first one took part in -- was frozen)
    for name in (R.RUN_C3_OUTLIER2, R.RUN_C3_OUTLIER2_S2, R.RUN_C3_OUTLIER2_S3):
        g = R.load_run(name)
        gen, _ = R.build_models(DEV, with_tokenizer=False, name=name)
        st = gen.weight_statistics()
        print(f"{name}: weight statistics {st}")
        assert st["kurtosis"] > 14 and st["heavy_tailed"] and gen.resolved_precision() == 4
        noise = R.reference_noise(g, gen.device)
        out = {}
        for tag, prec in (("product default", -1), ("precision 2", 2), ("single fp16", 0)):
            gen.precision = prec
            bad, tot, per, _ = R.teacher_forced(gen, g, noise)
            out[tag] = bad
            print(f"{name} [{tag}, resolves to {gen.resolved_precision()}]: {bad}/{tot} = {bad / tot:.2e}; per eighth of the run {[sum(per[i:i + 8]) for i in range(0, 64, 8)]}")
            assert tot == 168568
        assert gen.saturation_count() == 0
        assert out["product default"] / 168568 <= 1e-3
        if out["single fp16"] >= 100:               # (the ordering of the modes is asserted where the run has enough near-ties to show it: two of the three runs of this
            assert out["product default"] <= out["precision 2"] <= out["single fp16"]       #  family are easy -- 26 / 32 mismatches in single fp16, every mode inside the counts' noise)
        del gen
        torch.cuda.empty_cache()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name,expect,bound", [("sample_full12_64_prenorm", 2, 8e-4), ("sample_full12_64_seq1024", 2, 7e-4)])
def test_generator_variants_full_width_vs_reference_runs(name, expect, bound):
    """configs[2]'s sampler (64 steps, CFG 7.1 cosine) on the two generator variants of the reference that differ in how the engine runs the guided
    forward, against full-width runs of the REAL reference (oracle/make_golden.py RUNS):
      * use_prenorm=True (bert.py:49-59, 106-123): since round 4 the guided forward runs in differential form here too (the LayerNorm pair kernel in
        front of each sub-layer, the raw residual in the stream buffer) with the weight-correction mini-tiles -- as independent streams the same
        mini-tiles measured 1.44e-3 (guidance multiplies the two streams' separate fp16 roundings), hi + lo activation pairs 8.4e-4;
      * 1024 + 1 tokens (the 512 x 512 models, scripts/eval_maskbit.py:125,139-144): since round 5 the differential form with the weight-correction
        mini-tiles here too (eight 128-token pair tiles per sequence pair, the streaming attention kernel in pair form); rounds 3-4 ran the plain forward
        over [cond | uncond] with hi + lo activation pairs (the e4m3 lo pass, retired in round 5) and measured 7.6e-4 (single fp16: 1.11e-3)."""
    from maskbit_amd import parity_replay as R
    g = R.load_run(name)
    gen, _ = R.build_models(DEV, with_tokenizer=False, name=name)
    assert gen.resolved_precision() == expect
    noise = R.reference_noise(g, gen.device)
    bad, tot, per, _ = R.teacher_forced(gen, g, noise)
    print(f"{name} [product default, resolves to {gen.resolved_precision()}]: {bad}/{tot} = {bad / tot:.2e}; per eighth {[sum(per[i:i + 8]) for i in range(0, 64, 8)]}")
    assert bad / tot <= bound


@pytest.mark.timeout(900)
def test_demo_call_site_14bit_vs_reference_run():
    """The demo's call of sample() (demo_utils.py:139-157 with configs/demo/demo.yaml): the 14-bit generator, guidance ON with guidance_annealing="none"
    and scale_pow=1.0 -- the full scale from step 0 on, while every position is still masked, where every other recorded run anneals it in from 0 (so
    no step of this run takes the conditional-only shortcut) --, arccos schedule, 64 steps; a full-size run of the REAL reference (oracle/make_golden.py
    RUNS sample_full14_demo, batch 4: 84 284 sampled positions), teacher-forced in the product default."""
    from maskbit_amd import parity_replay as R
    g = R.load_run(R.RUN_DEMO14)
    assert g["kw"]["guidance_annealing"] == "none" and float(g["kw"]["scale_pow"]) == 1.0 and float(g["kw"]["guidance_scale"]) == 3.0 and g["bits"] == 14
    assert all(s == 3.0 for s in R.plan_of(g)[0])
    r = _vs_reference_run(R.RUN_DEMO14, [("product default", -1), ("single fp16", 0)])
    bad, tot = r["product default"]
    assert tot == 84284 and bad / tot <= 1e-3 and bad < r["single fp16"][0]


@pytest.mark.timeout(900)
def test_other_shipped_codebooks_16_and_18_bit_vs_reference_runs():
    """The two shipped generator codebooks BASELINE.json does not name (README.md:74-75): 16-bit (C = 256 per group) and 18-bit (C = 512), each with the sampler
    block of its own yaml (guidance 6.5 / 5.7 cosine, scale_pow 2.5, 64 steps), full-size runs of the REAL reference (oracle/make_golden.py RUNS sample_full16_64: batch 4,
    84 284 positions; sample_full18_64: batch 2, 42 142), teacher-forced in the product default (precision 3 at 8 / 9 bits per group)."""
    from maskbit_amd import parity_replay as R
    for name, tot_want in ((R.RUN_16BIT, 84284), (R.RUN_18BIT, 42142)):
        r = _vs_reference_run(name, [("product default", -1), ("single fp16", 0)])
        bad, tot = r["product default"]
        assert tot == tot_want and bad / tot <= 1e-3 and bad < r["single fp16"][0], (name, bad, tot)


def _full_length_run(bits, num_steps, B, kw, seed):
    """A complete free-running mb_sample of a BASELINE configuration at full size and full length, checked through the size-independent
    properties of the loop (sampling.py:81-131): the run is deterministic; it equals, bit for bit, the step-by-step composition
    mb_gen_forward + mb_sample_step on the same noise; in that composition exactly clamp(k_i, 1, masked - 1) positions of every image
    stay masked after step i (k_i from the reference-derived schedule golden), decoded tokens never change again, and nothing is
    masked after the last step; the final codes are the combined groups of the last prediction."""
    import numpy as np
    from conftest import GOLDEN
    from maskbit_amd import _lib
    from maskbit_amd.sampling import build_plan, draw_noise, run_loop
    lib = _lib.load()
    cfg = O.GenCfg(bits=bits, splits=2)
    sd = O.make_generator_weights(cfg, seed=200 + bits, head_gain=12.0)
    model = hip_generator(cfg, sd)
    C_ = cfg.group_codes
    labels = ((torch.arange(B) * 37) % 1000).to(DEV)
    gs = kw["guidance_scale"]
    plan = build_plan(num_steps, 512, gs, kw.get("guidance_annealing", "none"), kw.get("scale_pow", 4.0), 1.0, False, "arccos")
    sched = np.load(os.path.join(GOLDEN, "schedule.npz"))[f"arccos_{num_steps}"]
    assert [int(v) for v in sched] == list(plan[2])                    # the host plan equals the reference's floor(ratio * 512) table
    runs = []
    for _ in range(2):
        torch.manual_seed(seed)
        torch.cuda.manual_seed(seed)
        q, c = draw_noise(B, 256, 2, C_, num_steps, kw["randomize_temperature"], torch.device(DEV))
        _, _, steps, codes = run_loop(model, None, labels, plan, q, c, want_image=False)
        torch.cuda.synchronize()
        runs.append((steps, codes))
    steps, codes = runs[0]
    assert torch.equal(steps, runs[1][0]) and torch.equal(codes, runs[1][1])          # deterministic
    assert int(steps.min()) >= 0 and int(steps.max()) < C_                             # predictions are always valid codes
    assert torch.equal(codes.cpu(), O.combine_groups(steps[-1].cpu(), bits, 2).long())
    # the same run, one C-ABI call per stage
    tok = torch.full((B, 256, 2), C_, dtype=torch.int64, device=DEV)
    drop = torch.cat([torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)]).to(DEV)
    masked = 512
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(num_steps):
        if gs != 0.0 and plan[0][i] != 0.0:                 # (where the annealed scale is exactly 0 the loop runs the conditional forward alone: c + 0 (c - u) == c)
            lg = model.forward_cfg(tok, labels)
            lc, lu = lg[:B].contiguous(), lg[B:].contiguous()
        else:
            lc, lu = model(tok, labels, torch.zeros(B, dtype=torch.bool, device=DEV)), None
        tout, pred = torch.empty_like(tok), torch.empty_like(tok)
        _lib.check(lib.mb_sample_step(lc.data_ptr(), lu.data_ptr() if lu is not None else None, plan[0][i], plan[1][i], q[i].data_ptr(),
                                      c[i].data_ptr(), plan[2][i], tok.data_ptr(), tout.data_ptr(), pred.data_ptr(), B, 256, 2, C_, stream))
        assert torch.equal(pred, steps[i]), f"step {i}: mb_sample differs from the forward + step composition"
        was_open = tok == C_
        assert torch.equal(pred[~was_open], tok[~was_open])                            # decoded tokens never change
        k = max(1, min(int(sched[i]), masked - 1))
        left = (tout == C_).reshape(B, -1).sum(1)
        if i + 1 < num_steps:
            assert int(left.min()) == k and int(left.max()) == k, f"step {i}: {left.tolist()} masked, schedule says {k}"
        masked = k
        tok = tout
    return steps


@pytest.mark.timeout(900)
def test_baseline_config1_full_length_property_run():
    """BASELINE configs[1] as named: 10-bit, 16 steps, no CFG, batch 16, through mb_sample in the product default precision."""
    _full_length_run(10, 16, 16, dict(guidance_scale=0.0, randomize_temperature=10.5), seed=11)


@pytest.mark.timeout(900)
def test_baseline_config5_full_length_property_run():
    """BASELINE configs[4]'s per-GPU shard as named: 14-bit, 256 steps, CFG 5.8 cosine, batch 32 (configs/generator/
    maskbit_generator_14bit_256steps.yaml:38-44), the whole 256-step loop on the GPU in the product default precision."""
    _full_length_run(14, 256, 32, dict(guidance_scale=5.8, guidance_annealing="cosine", scale_pow=3.0, randomize_temperature=10.3), seed=12)


@pytest.mark.timeout(900)
def test_baseline_config3_three_reference_runs_other_weights_noise_and_labels():
    """configs[2] from THREE full-size runs of the real reference (tests/golden/sample_full12_64.npz; _s2: generator seed 177, head gain 16, noise
    seed 4321, other labels; _s3: seed 180, batch 8 -- 337 136 sampled positions together; 64 steps, CFG 7.1 cosine).  The product default
    (differential guidance + weight-correction mini-tiles) measures 5.6e-4 / 5.7e-4 / 5.4e-4: asserted <= 7e-4 on each run and <= 6e-4 over all, no
    statistical allowance.  The differential form alone (round 2's default) measures ~1e-3 over all: AT the bound, which is why it is not the
    default; asserted at what it measures."""
    from maskbit_amd import parity_replay as R
    tot_all = {"product default": [0, 0], "differential only": [0, 0]}
    for name in ("sample_full12_64", R.RUN_C3_S2, R.RUN_C3_S3):
        g = R.load_run(name)
        gen, _ = R.build_models(DEV, with_tokenizer=False, name=name)
        noise = R.reference_noise(g, gen.device)
        for tag, prec in (("product default", -1), ("differential only", 1)):
            gen.precision = prec
            bad, tot, per, _ = R.teacher_forced(gen, g, noise)
            tot_all[tag][0] += bad; tot_all[tag][1] += tot
            print(f"{name}, {tag}: {bad}/{tot} = {bad / tot:.2e}; per 8 steps {[sum(per[i:i + 8]) for i in range(0, 64, 8)]}")
            assert tot in (84284, 168568)
            assert bad / tot <= (7e-4 if tag == "product default" else 1.3e-3), (name, tag)
        del gen
        torch.cuda.empty_cache()
    for tag, (b, t) in tot_all.items():
        print(f"configs[2], three reference runs, {tag}: {b}/{t} = {b / t:.2e}")
    assert tot_all["product default"][1] == 337136 and tot_all["product default"][0] / 337136 <= 6e-4
    assert tot_all["differential only"][0] / 337136 <= 1.2e-3


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("wseed,gain", [(100, 12.0), (177, 16.0)])
def test_parity_margin_over_weight_seeds_and_head_gain(wseed, gain):
    """The parity figure is not a single-draw result: 12-bit generator, two weight seeds / head gains, 12 CFG steps spread over the
    64-step schedule's guidance range (B = 4, oracle-driven), product default precision."""
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=wseed, head_gain=gain)
    model = hip_generator(cfg, sd)
    mism, logit_err = _teacher_forced(cfg, sd, model, 4, 12, torch.tensor([5, 250, 500, 750]), 1000 + wseed, guidance_scale=7.1,
                                      guidance_annealing="cosine", scale_pow=3.0, randomize_temperature=8.2, mask_schedule_strategy="arccos")
    print(f"weights seed {wseed}, head gain {gain}: teacher-forced mismatch {mism:.2e}, mean |logit err| {logit_err:.4f}")
    # 16 544 sampled positions: a true rate of 1e-3 gives 16.5 +- 4.1 mismatches, so this small sample can only be checked for CONSISTENCY with the
    # bound (two sigma); the strict <= 1e-3 assertions are the 84 k / 87 k / 167 k-position replays of the reference's own runs
    n = 16544
    assert mism * n <= 1e-3 * n + 2 * (1e-3 * n) ** 0.5


@pytest.mark.timeout(900)
def test_held_out_reference_runs_default_precision():
    """Round 5 decided WHERE the lo refinements of the LayerNorm outputs run (FFN-up only, layers >= depth / 2) on the recorded runs of the reference
    (profiles/r05_coverage.md).  These runs -- configs[1] at batch 16, configs[2] at batch 8, configs[4] at batch 4; other weights / head gain / noise / labels -- were
    recorded AFTER those decisions were frozen and took no part in them: the product default must meet the north star's 1e-3 on each, and the 7e-4 the
    other runs' tests assert per run."""
    from maskbit_amd import parity_replay as R
    for name in (R.RUN_CFG1_S4, R.RUN_C3_S4, R.RUN_CFG5_S5):
        modes = [("product default", -1)] + ([] if name == R.RUN_CFG5_S5 else [("differential / hi + lo operands alone (precision 1)", 1)])   # (256 steps: the default only; precision 1 measured 6.9e-4)
        r = _vs_reference_run(name, modes)
        bad, tot = r["product default"]
        assert tot >= 87040 and bad / tot <= 7e-4, (name, bad, tot)
        if len(modes) > 1:
            assert bad < r["differential / hi + lo operands alone (precision 1)"][0]
