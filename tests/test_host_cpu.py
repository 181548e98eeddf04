"""CPU-only tests of the host side: C-ABI library exports, reference-compatible module surface,
checkpoint key compatibility, schedule/plan arithmetic, loud failure without a GPU."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, golden_weights
from oracle import maskbit_oracle as O

TINY_GEN = O.GenCfg(bits=12, splits=2, hidden=128, depth=2, heads=4, mlp=256, seq=256, nclass=10)


def test_library_exports_every_declared_symbol():
    """include/maskbit_hip.h (the ABI) + include/maskbit_hip_diag.h (single-kernel test entries) <-> libmaskbit_hip.so <-> ctypes signatures stay
    in sync; the ABI header itself declares nothing but the reference surface, the measurement hooks and the two saturation counters."""
    from maskbit_amd import _lib
    abi = open(os.path.join(ROOT, "include", "maskbit_hip.h")).read()
    header = abi + open(os.path.join(ROOT, "include", "maskbit_hip_diag.h")).read()
    assert not re.findall(r"\b(mb_gemm[a-z0-9_]*|mb_layernorm[a-z0-9_]*|mb_w4[a-z0-9_]*|mb_set_cu_count|mb_gen_set_wcorr)\s*\(", abi)
    declared = set(re.findall(r"\b(mb_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mb_abi_version() == _lib.ABI_VERSION == 8
    assert isinstance(lib.mb_last_error(), bytes)


def test_struct_layouts_match_header():
    from maskbit_amd import _lib
    assert ctypes.sizeof(_lib.GenCfg) == 11 * 4
    assert ctypes.sizeof(_lib.DecCfg) == (5 + 8 + 1 + 3) * 4
    assert ctypes.sizeof(_lib.SamplePlan) == 8 + 3 * 8 + 8


def test_generator_state_dict_matches_reference_keys():
    """The oracle's weights were loaded with strict=True into the REAL reference when the goldens were
    made; loading the same dict strictly here pins our key names and shapes to the reference's."""
    from maskbit_amd import LFQBert
    sd = golden_weights(load_golden("gen_tiny.npz"))
    m = LFQBert(hidden_dim=128, codebook_size=4096, codebook_splits=2, depth=2, heads=4, mlp_dim=256, nclass=10)
    m.load_state_dict(sd, strict=True)
    full = LFQBert(img_size=256, hidden_dim=1024, codebook_size=4096, codebook_splits=2, depth=24, heads=16, mlp_dim=4096)
    assert len(full.state_dict()) == 301
    assert sum(p.numel() for p in full.parameters()) == 304_795_776
    assert full.mask_token == 64 and full.effective_codebook_size == 64 and full.seq_len == 256


def test_tokenizer_state_dict_matches_reference_keys():
    from maskbit_amd import ConvVQModel
    from hip_helpers import tok_config
    cfg = O.TokCfg(token_size=12)
    sd = O.make_tokenizer_weights(cfg, seed=200, with_encoder=True)
    m = ConvVQModel(tok_config(cfg))
    m.load_state_dict(sd, strict=True)
    assert len(m.state_dict()) == 177
    assert torch.equal(m.quantize.codebook, sd["quantize.codebook"])


def test_sample_signature_matches_reference():
    from maskbit_amd import sample
    sig = inspect.signature(sample)
    names = list(sig.parameters)
    assert names == ["model", "vqgan_model", "num_samples", "labels", "softmax_temperature", "randomize_temperature",
                     "mask_schedule_strategy", "num_steps", "guidance_scale", "mask_token", "patch_size",
                     "guidance_annealing", "use_sampling_annealing", "scale_pow", "codebook_size", "codebook_splits", "use_tqdm"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["num_samples"], d["randomize_temperature"], d["mask_schedule_strategy"], d["num_steps"], d["guidance_scale"],
            d["mask_token"], d["scale_pow"], d["codebook_size"], d["codebook_splits"]) == (10, 4.5, "linear", 12, 3.0, 1024, 4.0, 1024, 1)


def test_import_paths_of_the_reference_drivers():
    from modeling.bert import Bert, LFQBert          # scripts/eval_maskbit.py:11-13
    from modeling.conv_vqgan import ConvVQModel
    from modeling.modules import sample, BaseModel
    import maskbit_amd
    assert LFQBert is maskbit_amd.LFQBert and ConvVQModel is maskbit_amd.ConvVQModel and sample is maskbit_amd.sample
    assert issubclass(Bert, LFQBert)


def test_generator_variants_take_the_reference_checkpoint_keys():
    """use_prenorm and the embedding-table Bert: the oracle's seeded state dicts (whose key names / shapes were loaded strict=True
    into the REAL reference classes by oracle/make_golden_variants.py) load strict into ours."""
    from maskbit_amd import LFQBert
    from maskbit_amd.bert import Bert
    base = dict(bits=12, splits=2, hidden=128, depth=2, heads=4, mlp=256, seq=256, nclass=10)
    for cls, cfg in ((LFQBert, O.GenCfg(**base, prenorm=True)), (Bert, O.GenCfg(**base, kind="bert")), (Bert, O.GenCfg(**base, kind="bert", prenorm=True))):
        sd = O.make_generator_weights(cfg, seed=1)
        m = cls(img_size=256, hidden_dim=128, codebook_size=4096, codebook_splits=2, depth=2, heads=4, mlp_dim=256, nclass=10, use_prenorm=cfg.prenorm)
        m.load_state_dict(sd, strict=True)
        assert set(m.state_dict().keys()) == set(sd.keys())


def test_no_cpu_fallback():
    from maskbit_amd import LFQBert, ConvVQModel, sample
    from hip_helpers import tok_config
    m = LFQBert(hidden_dim=128, codebook_size=4096, codebook_splits=2, depth=1, heads=4, mlp_dim=256, nclass=10)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 256, 2, dtype=torch.long), torch.zeros(1, dtype=torch.long))
    t = ConvVQModel(tok_config(O.TokCfg(token_size=12, hidden_channels=64, channel_mult=(1, 1, 2), num_resolutions=3, num_res_blocks=1)))
    with pytest.raises(RuntimeError, match="no CPU path"):
        t.decode_tokens(torch.zeros(1, 256))
    with pytest.raises(RuntimeError):
        sample(m, t, num_samples=1, labels=torch.zeros(1, dtype=torch.long), mask_token=64, codebook_size=4096, codebook_splits=2)
    src = open(os.path.join(ROOT, "maskbit_amd", "sampling.py")).read() + open(os.path.join(ROOT, "maskbit_amd", "bert.py")).read()
    assert "oracle" not in src.replace("test infrastructure", "")


def test_plan_matches_oracle_schedule():
    from maskbit_amd.sampling import build_plan
    from maskbit_amd.masking import get_masking_ratio
    for (N, ann, sp, strat) in [(64, "cosine", 3.0, "arccos"), (16, "none", 4.0, "linear"), (8, "linear", 1.0, "cosine"), (256, "cosine", 3.0, "arccos")]:
        scale, temp, mlen = build_plan(N, 512, 7.1, ann, sp, 1.0, False, strat)
        assert mlen == [int(v) for v in O.mask_len_schedule(N, 512, strat)]
        for i in range(N):
            ref = 7.1 * O.guidance_factor(i, N, ann, sp)
            assert scale[i] == float(torch.as_tensor(ref, dtype=torch.float32).reshape(-1)[0])
        assert temp == [1.0] * N
    _, temp, _ = build_plan(4, 512, 3.0, "none", 1.0, 1.0, True, "arccos")
    assert temp == [0.5 + 0.8 * (1 - (i + 1) / 4) for i in range(4)]
    with pytest.raises(ValueError):
        build_plan(4, 512, 3.0, "none", 1.0, 1.0, False, "bogus")
    z = load_golden("schedule.npz")
    for mode in ("arccos", "cosine", "linear", "square", "root"):
        mine = np.array([float(get_masking_ratio((i + 1) / 64, mode)) for i in range(64)], dtype=np.float32)
        assert np.array_equal(mine, z[f"ratio_{mode}_64"])


def test_factorization_matches_reference_vectors():
    from maskbit_amd import combine_factorized_tokens, split_factorized_tokens
    z = load_golden("schedule.npz")
    t = torch.from_numpy(z["split_in"])
    sp = split_factorized_tokens(t, 4096, 2)
    assert torch.equal(sp, torch.from_numpy(z["split_out"]))
    comb = combine_factorized_tokens(sp, 4096, 2)
    assert comb.dtype == torch.float32 and torch.equal(comb, torch.from_numpy(z["combine_out"]))


def test_load_pretrained_roundtrip(tmp_path):
    from maskbit_amd import LFQBert
    sd = golden_weights(load_golden("gen_tiny.npz"))
    legacy = {k.replace("input_proj", "token_emb"): v for k, v in sd.items()}      # eval_maskbit.py:52-53 rename
    torch.save(legacy, tmp_path / "gen.bin")
    m = LFQBert(hidden_dim=128, codebook_size=4096, codebook_splits=2, depth=2, heads=4, mlp_dim=256, nclass=10)
    m.train()
    m.load_pretrained(str(tmp_path / "gen.bin"), rename_keys={"token_emb": "input_proj"})
    assert not m.training and torch.equal(m.input_proj.weight, sd["input_proj.weight"])
    m.save_pretrained(str(tmp_path / "out"))
    again = torch.load(tmp_path / "out" / "pytorch_model.bin")
    assert set(again) == set(sd)
    with pytest.raises(ValueError):
        m.load_pretrained(str(tmp_path / "missing.bin"))
    assert m.device == torch.device("cpu") and m.dtype == torch.float32


def test_eval_harness_host_pieces():
    """eval_maskbit.py:77-80 (mask token from the codebook size) and :107-108 (label protocol)."""
    from maskbit_amd import eval_labels, mask_token_for
    assert mask_token_for(4096, 2) == 64 and mask_token_for(1024, 2) == 32 and mask_token_for(2 ** 14, 2) == 128
    assert mask_token_for(1024, 1) == 1024 and mask_token_for(4096, 3) == 16 and mask_token_for(2 ** 18, 2) == 512
    torch.manual_seed(3)
    lab = eval_labels("cpu")
    torch.manual_seed(3)
    ref = torch.randperm(1000, dtype=torch.int).repeat(50)
    assert lab.dtype == torch.int32 and lab.shape == (50000,) and torch.equal(lab, ref)
    assert torch.equal(lab[:1000].sort().values, torch.arange(1000, dtype=torch.int32))


def test_host_noise_draws_leave_the_thread_pool_alone_and_keep_the_stream():
    """draw_noise draws the confidence noise on ONE worker thread whose own intra-op thread count is 1 (profiles/r03_parity.md section 4,
    tools/host_draw_ab.py: the pool of a many-core host starves the HIP runtime's threads): the calling thread's count is never changed, threads that
    exist or are created while draws run see the process's count, and the CPU stream is the one the reference consumes (one Gumbel draw of [B, n, m] per
    step from the default generator)."""
    import threading
    import torch
    from maskbit_amd import sampling as S
    from maskbit_amd.sampling import draw_noise
    assert not hasattr(S, "_FewCpuThreads")
    before = torch.get_num_threads()
    torch.manual_seed(5)
    _, big = draw_noise(70, 256, 2, 64, 3, 4.5, torch.device("cpu"))          # (the first call also starts the worker)
    assert S._DrawThread.run(torch.get_num_threads) == 1 and torch.get_num_threads() == before
    seen, stop = [], threading.Event()

    def watch():                                                    # another host thread, created and polling while the draws run
        while not stop.is_set():
            seen.append(torch.get_num_threads())
    w = threading.Thread(target=watch)
    w.start()
    for _ in range(3):
        draw_noise(70, 256, 2, 64, 3, 4.5, torch.device("cpu"))
    late = []
    t2 = threading.Thread(target=lambda: late.append(torch.get_num_threads()))
    t2.start(); t2.join()
    stop.set(); w.join()
    assert torch.get_num_threads() == before and seen and set(seen) == {before} and late == [before]
    torch.manual_seed(5)
    for i in range(3):
        torch.empty(70 * 512, 64).exponential_(1)
    g = torch.distributions.Gumbel(0.0, 1.0)
    assert torch.equal(big, torch.stack([g.sample((70, 256, 2)) * 4.5 * (1 - (i + 1) / 3) for i in range(3)]))
    torch.manual_seed(11)
    _, conf = draw_noise(2, 256, 2, 64, 3, 4.5, torch.device("cpu"))
    assert torch.get_num_threads() == before
    torch.manual_seed(11)
    for i in range(3):
        torch.empty(2 * 512, 64).exponential_(1)
    g = torch.distributions.Gumbel(0.0, 1.0)
    want = torch.stack([g.sample((2, 256, 2)) * 4.5 * (1 - (i + 1) / 3) for i in range(3)])
    assert torch.equal(conf, want)


def test_auto_precision_escalates_on_heavy_tailed_checkpoints(monkeypatch):
    """The auto mode's load-time rule (LFQBert.weight_statistics / resolved_precision, round 6): Gaussian-like checkpoints resolve to 2 (3 from 7 bits per
    group on); heavy-tailed Linear weights OR massive-activation LayerNorm channels -- the trained-like family of maskbit_amd/synth.py shows both --
    escalate to PREC_ALO_ALL (4); an explicit knob wins; shapes the mini-tiles do not serve still degrade; new weights re-evaluate the rule."""
    import torch
    from maskbit_amd import LFQBert, synth
    from maskbit_amd import bert as B
    monkeypatch.delenv("MASKBIT_AMD_PRECISION", raising=False)
    cfg = synth.GenCfg(bits=12, splits=2, hidden=768, depth=2, heads=12, mlp=1024)
    mk = lambda c=cfg: LFQBert(img_size=256, hidden_dim=c.hidden, codebook_size=2 ** c.bits, codebook_splits=2, depth=c.depth, heads=c.heads, mlp_dim=c.mlp)
    m = mk()
    st = m.weight_statistics()
    assert 2.5 < st["kurtosis"] < 3.2 and st["channel_ratio"] < 1.5 and not st["heavy_tailed"] and m.resolved_precision() == B.PREC_WCORR
    m.load_state_dict(synth.make_generator_weights(cfg, seed=1), strict=True)
    st = m.weight_statistics()
    assert 2.8 < st["kurtosis"] < 3.2 and not st["heavy_tailed"] and m.resolved_precision() == B.PREC_WCORR
    m.load_state_dict(synth.make_generator_weights(cfg, seed=1, style="outlier"), strict=True)          # new weights: the cached statistics are stale
    st = m.weight_statistics()
    assert st["kurtosis"] > 8 and st["channel_ratio"] > 8 and st["heavy_tailed"] and m.resolved_precision() == B.PREC_ALO_ALL
    m.precision = B.PREC_WCORR
    assert m.resolved_precision() == B.PREC_WCORR                                       # an explicit knob is not escalated
    m.precision = B.PREC_AUTO
    # either statistic alone escalates
    sd = synth.make_generator_weights(cfg, seed=2)
    for k in sd:
        if k.endswith("norm.bias"):
            sd[k][7] = 9.0                                                              # one massive-activation channel, Gaussian weights
    m.load_state_dict(sd, strict=True)
    st = m.weight_statistics()
    assert st["kurtosis"] < 3.2 and st["channel_ratio"] > B.ESCALATE_CHANNEL_RATIO and m.resolved_precision() == B.PREC_ALO_ALL
    sd = synth.make_generator_weights(cfg, seed=2)
    gq = torch.Generator().manual_seed(0)
    for k in sd:
        if sd[k].dim() == 2 and k.startswith("transformer.layers."):
            sd[k] = sd[k] * torch.where(torch.rand(sd[k].shape, generator=gq) < 0.03, 3.5, 0.9)     # heavy tails only
    m.load_state_dict(sd, strict=True)
    st = m.weight_statistics()
    assert st["kurtosis"] > B.ESCALATE_KURTOSIS and st["channel_ratio"] < 1.5 and m.resolved_precision() == B.PREC_ALO_ALL
    # a shape without mini-tiles (heads of 32) cannot run precision >= 2 whatever the statistics say
    c2 = synth.GenCfg(bits=12, splits=2, hidden=768, depth=2, heads=24, mlp=1024)
    m2 = mk(c2)
    m2.load_state_dict(synth.make_generator_weights(c2, seed=1, style="outlier"), strict=True)
    assert m2.resolved_precision() == B.PREC_DIFF


def test_precision_knob_resolution(monkeypatch):
    """LFQBert.precision is the ONE precision knob (mb_gen_cfg.precision: 0 fp16, 1 differential guidance, 2 + weight-correction mini-tiles, 3 +
    activation-lo mini-tiles): the default resolves by codebook and degrades by what the pair / mini tiles serve; the header's enum, the ctypes struct and
    the host constants agree."""
    from maskbit_amd import LFQBert, _lib
    from maskbit_amd import bert as B
    monkeypatch.delenv("MASKBIT_AMD_PRECISION", raising=False)
    full = dict(hidden_dim=1024, depth=1, heads=16, mlp_dim=4096, codebook_splits=2)
    assert LFQBert(codebook_size=4096, **full).resolved_precision() == B.PREC_WCORR == 2          # 6 bits per group
    assert LFQBert(codebook_size=2 ** 14, **full).resolved_precision() == B.PREC_ALO == 3         # 7 bits per group
    assert LFQBert(codebook_size=4096, img_size=512, **full).resolved_precision() == 2             # 1024 + 1 tokens: pair tiles since round 5
    assert LFQBert(codebook_size=4096, use_prenorm=True, **full).resolved_precision() == 2
    assert LFQBert(codebook_size=4096, hidden_dim=1024, depth=1, heads=32, mlp_dim=4096, codebook_splits=2).resolved_precision() == 1   # heads of 32: no e2m1 attention output
    assert LFQBert(codebook_size=4096, hidden_dim=128, depth=1, heads=4, mlp_dim=256, codebook_splits=2).resolved_precision() == 1      # tiny: engine falls back to [cond | uncond] with hi + lo LayerNorm outputs
    m = LFQBert(codebook_size=4096, **full)
    for p in (0, 1, 2, 3):
        m.precision = p
        assert m.resolved_precision() == p
    monkeypatch.setenv("MASKBIT_AMD_PRECISION", "1")
    assert LFQBert(codebook_size=4096, **full).resolved_precision() == 1
    hdr = open(os.path.join(ROOT, "include", "maskbit_hip.h")).read()
    assert "MB_PREC_FP16 = 0, MB_PREC_DIFF = 1, MB_PREC_WCORR = 2, MB_PREC_ALO = 3" in hdr
    assert [n for n, _ in _lib.GenCfg._fields_][-3:] == ["prenorm", "embed_tables", "precision"]
    assert not re.search(r"\b(weight_split|act_split|cfg_pair)\b", hdr)
