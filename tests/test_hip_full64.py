"""Full-size parity against the REAL reference's own run (tests/golden/sample_full12_64.npz, written by
`oracle/make_golden.py full64`: modeling.modules.sample() of the reference, 12-bit generator, 64 steps, CFG 7.1 cosine, arccos
schedule, B = 4, CPU fp32, seed 1234).  This is the north star's parity figure inside the driver-visible suite: the HIP engine
replays the reference's noise and is compared step by step (teacher-forced: 84 284 sampled positions) and end to end (free-running).
No oracle import: weights and noise are regenerated from seeds, expectations come from the fixture."""
import pytest
import torch

from maskbit_amd import parity_replay as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    g = R.load_full64()
    gen, tok = R.build_models("cuda")
    noise = R.reference_noise(g, gen.device)
    return g, gen, tok, noise


def test_fixture_is_the_documented_run(setup):
    g = setup[0]
    assert g["steps"].shape == (64, 4, 256, 2) and int(g["masks"].sum()) == 84284 and bool(g["masks"][0].all())
    assert g["kw"]["guidance_scale"] == "7.1" and g["kw"]["num_steps"] == "64" and g["kw"]["mask_schedule_strategy"] == "arccos"


@pytest.mark.timeout(900)
def test_teacher_forced_mismatch_vs_reference_run(setup):
    """The product default precision (the mode bench.py times: differential guidance + weight-correction pass) measures 4.9e-4 on this run:
    asserted <= 7e-4 (the north star's bound is 1e-3); the single-fp16 mode is measured beside it for context."""
    g, gen, tok, noise = setup
    gen.precision = -1
    bad, tot, per_step, remask = R.teacher_forced(gen, g, noise)
    print(f"product default: teacher-forced mismatch vs the reference's run {bad}/{tot} = {bad / tot:.2e}; re-mask differences {remask}; "
          f"per 8 steps {[sum(per_step[i:i + 8]) for i in range(0, 64, 8)]}")
    assert tot == 84284
    assert bad / tot <= 7e-4
    gen.precision = 0
    bad0, _, per0, _ = R.teacher_forced(gen, g, noise)
    print(f"single fp16:     teacher-forced mismatch vs the reference's run {bad0}/{tot} = {bad0 / tot:.2e}; "
          f"per 8 steps {[sum(per0[i:i + 8]) for i in range(0, 64, 8)]}")
    gen.precision = -1
    assert bad0 / tot < 3e-3 and bad <= bad0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("prec,bound", [(1, 1e-3), (2, 7e-4), (3, 7e-4)])
def test_every_differential_mode_meets_the_bound(setup, prec, bound):
    """LFQBert.precision 1 (differential CFG operands alone: 8.4e-4 on THIS run, ~1.0e-3 over three), 2 (+ the MX-fp4 weight-correction mini-tiles on
    every trunk GEMM: the product default at this codebook), 3 (+ the activation-lo mini-tiles of the LayerNorm outputs) against the reference's run."""
    g, gen, tok, noise = setup
    gen.precision = prec
    bad, tot, per_step, _ = R.teacher_forced(gen, g, noise)
    gen.precision = -1
    print(f"precision = {prec}: teacher-forced mismatch vs the reference's run {bad}/{tot} = {bad / tot:.2e}; "
          f"per 8 steps {[sum(per_step[i:i + 8]) for i in range(0, 64, 8)]}")
    assert bad / tot <= bound


@pytest.mark.timeout(900)
def test_free_running_64_steps_vs_reference_run(setup):
    """One mb_sample call over all 64 steps + decode with the reference's noise.  A single flipped token changes every later step of
    that image, so the trajectories are compared statistically: the first step (same input for both) must agree to <= 2e-3, images
    whose final codes equal the reference's must decode to the reference's pixels, and the drift is reported."""
    g, gen, tok, noise = setup
    gen.precision = -1
    r = R.free_running(gen, tok, g, noise)
    sm = r["step_mismatch"]
    print(f"free-running token mismatch vs the reference's run: step 0 {sm[0]:.2e}, step 15 {sm[15]:.2e}, step 31 {sm[31]:.2e}, "
          f"step 47 {sm[47]:.2e}, step 63 {sm[63]:.2e}; final codes {r['codes_mismatch']:.2e}; images with identical codes "
          f"{r['images_with_identical_codes']}/4 (pixel max err {r['pixel_max_err_identical_codes']}); mean |uint8 diff| {r['u8_mean_abs_diff']:.2f}")
    assert sm[0] <= 2e-3
    if r["pixel_max_err_identical_codes"] is not None:
        assert r["pixel_max_err_identical_codes"] < 0.03
    # the trajectories stay statistically close.  Measured over the builds of rounds 2-3: final codes 0.9-1.0 % different, mean |uint8 difference| of
    # the 4x-subsampled images 7.8-8.4 (a flipped code repaints its receptive field of the random-weight decoder); bounds at 3x / 2x of that
    assert r["codes_mismatch"] <= 0.03
    assert r["u8_mean_abs_diff"] <= 16.0
    assert max(sm) <= 0.03                                  # no step drifts further than the final codes do
