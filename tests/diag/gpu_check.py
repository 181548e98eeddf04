"""Diagnostic run on a GPU box: prints parity errors and timings of every engine stage.
Not a test (tests/ holds the assertions); used to look at numbers while developing."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import maskbit_oracle as O  # noqa: E402
from conftest import load_golden, golden_weights  # noqa: E402
from hip_helpers import hip_generator, hip_tokenizer, token_mismatch  # noqa: E402
from maskbit_amd import _lib  # noqa: E402

TINY_GEN = O.GenCfg(bits=12, splits=2, hidden=128, depth=2, heads=4, mlp=256, seq=256, nclass=10)
TINY_TOK = O.TokCfg(token_size=12, hidden_channels=64, channel_mult=(1, 1, 2), num_resolutions=3, num_res_blocks=1)


def stats(name, got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    print(f"[{name}] max|ref|={ref.abs().max():.4f} max_err={err.max():.5f} mean_err={err.mean():.6f} "
          f"rel_fro={(got - ref).norm() / ref.norm():.5f} nan={int(torch.isnan(got).sum())}", flush=True)


def main():
    what = sys.argv[1:] or ["gen_tiny", "gen_full", "step", "dec_tiny", "dec_full", "sample_tiny", "tf_full", "time"]
    dev = torch.device("cuda")
    print(torch.cuda.get_device_name(0), flush=True)
    lib = _lib.load()
    print("abi", lib.mb_abi_version(), flush=True)

    if "gen_tiny" in what:
        z = load_golden("gen_tiny.npz")
        sd = golden_weights(z)
        m = hip_generator(TINY_GEN, sd)
        out = m(torch.from_numpy(z["tokens"]).to(dev), torch.from_numpy(z["labels"]).to(dev), torch.from_numpy(z["drop"]).to(dev))
        torch.cuda.synchronize()
        stats("gen_tiny logits", out, torch.from_numpy(z["logits"]))

    if "gen_full" in what:
        z = load_golden("gen_full12.npz")
        cfg = O.GenCfg(bits=12, splits=2)
        sd = O.make_generator_weights(cfg, seed=int(z["seed"]), head_gain=float(z["head_gain"]))
        m = hip_generator(cfg, sd)
        t, y, d = torch.from_numpy(z["tokens"]).to(dev), torch.from_numpy(z["labels"]).to(dev), torch.from_numpy(z["drop"]).to(dev)
        out = m(t, y, d)
        torch.cuda.synchronize()
        ref = torch.from_numpy(z["logits"])
        stats("gen_full12 logits", out, ref)
        p_ref, p_got = torch.softmax(ref, -1), torch.softmax(out.cpu(), -1)
        print("  argmax agreement", float((p_ref.argmax(-1) == p_got.argmax(-1)).float().mean()),
              "max prob err", float((p_ref - p_got).abs().max()), flush=True)
        del m

    if "step" in what:
        gsd = golden_weights(load_golden("gen_tiny.npz"))
        rec = []
        torch.manual_seed(77)
        fwd = lambda tk, yy, dd: O.lfq_bert_forward(gsd, TINY_GEN, tk, yy, dd)
        O.sample_loop(fwd, 3, torch.tensor([1, 4, 8]), num_steps=8, guidance_scale=7.1, guidance_annealing="cosine", scale_pow=3.0,
                      randomize_temperature=8.2, mask_schedule_strategy="arccos", mask_token=64, codebook_splits=2, record=rec)
        bad = 0
        for i, r in enumerate(rec):
            B, n, m_ = r.tokens_in.shape
            C_ = r.logits_c.shape[-1]
            tin = r.tokens_in.to(dev).contiguous()
            tout = torch.empty_like(tin); pred = torch.empty_like(tin)
            k = int(torch.floor(torch.tensor(r.mask_ratio) * (n * m_)))
            lc, lu = r.logits_c.to(dev).contiguous(), r.logits_u.to(dev).contiguous()      # keep alive across the call
            qn, cn = r.exp_noise.to(dev).contiguous(), r.conf_noise.to(dev).contiguous()
            rc = lib.mb_sample_step(lc.data_ptr(), lu.data_ptr(), r.scale, 1.0, qn.data_ptr(), cn.data_ptr(),
                                    k, tin.data_ptr(), tout.data_ptr(), pred.data_ptr(), B, n, m_, C_, torch.cuda.current_stream().cuda_stream)
            _lib.check(rc)
            torch.cuda.synchronize()
            dp, dt = int((pred.cpu() != r.pred).sum()), int((tout.cpu() != r.tokens_out).sum())
            bad += dp + dt
            print(f"  step {i}: pred diff {dp} tokens_out diff {dt} (masked in: {int((r.tokens_in == 64).sum())})", flush=True)
        print("[step] total diffs", bad, flush=True)

    if "dec_tiny" in what:
        z = load_golden("tok_tiny.npz")
        sd = O.make_tokenizer_weights(TINY_TOK, seed=int(z["seed"]), with_encoder=True)
        tk = hip_tokenizer(TINY_TOK, sd)
        img = tk.decode_tokens(torch.from_numpy(z["tokens"]).to(dev).float())
        torch.cuda.synchronize()
        stats("dec_tiny image", img, torch.from_numpy(z["image"]))

    if "dec_full" in what:
        z = load_golden("tok_full12.npz")
        cfg = O.TokCfg(token_size=12)
        sd = O.make_tokenizer_weights(cfg, seed=int(z["seed"]), with_encoder=True)
        tk = hip_tokenizer(cfg, sd)
        img, u8 = tk.decode_tokens_uint8(torch.from_numpy(z["tokens"]).to(dev))
        torch.cuda.synchronize()
        stats("dec_full12 image(half)", img[:, :, ::2, ::2], torch.from_numpy(z["image_half"].astype(np.float32)))
        for (y, x) in ((0, 0), (120, 120), (240, 240), (37, 201)):
            stats(f"  crop {y},{x}", img[:, :, y:y + 16, x:x + 16], torch.from_numpy(z[f"crop_{y}_{x}"]))
        ref_u8 = (torch.clamp(img, 0, 1) * 255).permute(0, 2, 3, 1).to(torch.uint8)
        print("  fused uint8 == torch uint8:", bool((ref_u8 == u8).all()), flush=True)
        for bsz in (8, 32):
            toks = torch.randint(0, 4096, (bsz, 256), device=dev)
            tk.decode_tokens(toks); torch.cuda.synchronize()
            t0 = time.time(); tk.decode_tokens(toks); torch.cuda.synchronize(); dt = time.time() - t0
            print(f"  decode B={bsz}: {dt * 1e3:.2f} ms -> {bsz / dt:.1f} img/s, {bsz * 185.97e9 / dt / 1e12:.1f} TFLOP/s", flush=True)
        del tk

    if "sample_tiny" in what:
        from maskbit_amd.sampling import build_plan, run_loop
        z = load_golden("sample_tiny_cfg.npz")
        gsd = golden_weights(load_golden("gen_tiny.npz"))
        tz = load_golden("tok_tiny.npz")
        tsd = O.make_tokenizer_weights(TINY_TOK, seed=int(tz["seed"]), with_encoder=True)
        gm, tm = hip_generator(TINY_GEN, gsd), hip_tokenizer(TINY_TOK, tsd)
        kw = dict(num_steps=8, guidance_scale=7.1, guidance_annealing="cosine", scale_pow=3.0, randomize_temperature=8.2,
                  mask_schedule_strategy="arccos")
        # replay the reference's CPU noise through the HIP loop
        torch.manual_seed(1234)
        noise = []
        g = torch.distributions.Gumbel(0.0, 1.0)
        for i in range(8):
            q = torch.empty(3 * 256 * 2, 64).exponential_(1)
            noise.append((q, g.sample((3, 256, 2))))
        exp_noise = torch.stack([q for q, _ in noise]).to(dev)
        conf = torch.stack([gn * 8.2 * (1 - (i + 1) / 8) for i, (_, gn) in enumerate(noise)]).to(dev)
        plan = build_plan(8, 512, 7.1, "cosine", 3.0, 1.0, False, "arccos")
        img, _, steps, codes = run_loop(gm, tm, torch.from_numpy(z["labels"]), plan, exp_noise, conf)
        torch.cuda.synchronize()
        ref_steps = torch.from_numpy(z["steps"])
        for i in range(8):
            print(f"  free-running step {i}: token mismatch {token_mismatch(steps[i].cpu(), ref_steps[i]):.5f}", flush=True)
        stats("sample_tiny image", img, torch.from_numpy(z["image"]))

    if "tf_full" in what:
        # teacher-forced parity at full size: oracle (CPU fp32) drives the loop, HIP re-does every step from the oracle's inputs
        cfg = O.GenCfg(bits=12, splits=2)
        sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
        m = hip_generator(cfg, sd)
        B, N = int(os.environ.get("TF_B", "4")), int(os.environ.get("TF_STEPS", "8"))
        torch.set_num_threads(min(16, torch.get_num_threads()))
        y = torch.tensor([1, 7, 282, 604, 724, 179, 751, 404, 850, 13, 999, 500][:B])
        rec = []
        torch.manual_seed(4321)
        t0 = time.time()
        fwd = lambda tk, yy, dd: O.lfq_bert_forward(sd, cfg, tk, yy, dd)
        O.sample_loop(fwd, B, y, num_steps=N, guidance_scale=7.1, guidance_annealing="cosine", scale_pow=3.0,
                      randomize_temperature=8.2, mask_schedule_strategy="arccos", mask_token=64, codebook_splits=2, record=rec)
        print(f"  oracle loop {time.time() - t0:.1f}s on {torch.get_num_threads()} threads", flush=True)
        modes = [(int(v), 0) for v in os.environ.get("TF_SPLITS", "0").split(",")] + [(0, int(v)) for v in os.environ.get("TF_ACT", "").split(",") if v and v != "0"]
        quiet = bool(int(os.environ.get("TF_QUIET", "0")))
        for split, act in modes:
            m.weight_split, m.act_split = split, act
            tot_m = tot_n = 0
            for i, r in enumerate(rec):
                tin = r.tokens_in.to(dev).contiguous()
                lg = m(torch.cat([tin, tin]), torch.cat([y, y]).to(dev), torch.cat([torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)]).to(dev))
                lc, lu = lg[:B].contiguous(), lg[B:].contiguous()
                tout = torch.empty_like(tin); pred = torch.empty_like(tin)
                qn, cn = r.exp_noise.to(dev).contiguous(), r.conf_noise.to(dev).contiguous()
                k = int(torch.floor(torch.tensor(r.mask_ratio) * 512))
                _lib.check(lib.mb_sample_step(lc.data_ptr(), lu.data_ptr(), r.scale, 1.0, qn.data_ptr(), cn.data_ptr(), k, tin.data_ptr(),
                                              tout.data_ptr(), pred.data_ptr(), B, 256, 2, 64, torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                msk = r.tokens_in == 64
                mm = int((pred.cpu() != r.pred)[msk].sum()); nn_ = int(msk.sum())
                tot_m += mm; tot_n += nn_
                pmax = torch.softmax(r.logits_c + r.scale * (r.logits_c - r.logits_u), -1).max(-1).values[msk].mean()
                if not quiet:
                  print(f"  step {i}: scale={r.scale:.3f} masked={nn_} pred mismatch={mm} ({mm / max(nn_, 1):.5f}) remask diff={int((tout.cpu() != r.tokens_out).sum())} "
                      f"logit err max={float((lc.cpu() - r.logits_c).abs().max()):.3f} mean max-prob={float(pmax):.3f}", flush=True)
            print(f"[tf_full] weight_split={split} act_split={act}: teacher-forced token mismatch over masked positions: {tot_m}/{tot_n} = {tot_m / tot_n:.6f}", flush=True)
        del m

    if "time" in what:
        cfg = O.GenCfg(bits=12, splits=2)
        sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
        m = hip_generator(cfg, sd)
        for nb in (16, 128):
            t = torch.randint(0, 65, (nb, 256, 2), device=dev)
            y = torch.randint(0, 1000, (nb,), device=dev)
            m(t, y); torch.cuda.synchronize()
            _lib.prof_enable(True)
            t0 = time.time()
            for _ in range(3):
                m(t, y)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / 3
            prof = _lib.prof_read(); _lib.prof_enable(False)
            print(f"[time] forward nb={nb}: {dt * 1e3:.2f} ms  -> {nb * 162.33e9 / dt / 1e12:.1f} TFLOP/s", flush=True)
            for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                print(f"    {k:16s} calls={c:5d} total={ms:9.3f} ms avg={ms / c * 1e3:9.1f} us", flush=True)


if __name__ == "__main__":
    main()
