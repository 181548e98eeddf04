"""Throughput of the other BASELINE configurations through the product's own sample() path (bench.py times configs[2], the headline):
configs[1] = 10-bit generator, 16 steps, no guidance; configs[4]'s generator = 14-bit, 256 steps, CFG 5.8 cosine.  Synthetic weights (maskbit_amd.synth),
decode to uint8 included; with the per-kernel HIP-event averages of the timed batches.  usage: python tools/config_bench.py [batch ...]   (default 64)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from maskbit_amd import parity_replay as PR
from maskbit_amd.sampling import build_plan, run_chunked


def tokenizer(bits, dev):
    """conv-VQGAN decoder for a `bits`-bit lookup-free codebook, seeded synthetic weights."""
    from maskbit_amd import ConvVQModel, synth

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    tok = ConvVQModel(Cfg(quantizer_type="lookup-free", codebook_size=2 ** bits, token_size=bits, num_channels=3, hidden_channels=128,
                          channel_mult=[1, 1, 2, 2, 4], num_resolutions=5, num_res_blocks=2, sample_with_conv=True))
    tok.load_state_dict(synth.make_tokenizer_weights(synth.TokCfg(token_size=bits), seed=7), strict=False)
    return tok.eval().requires_grad_(False).to(dev)


def main():
    from maskbit_amd import _lib
    dev = torch.device("cuda")
    runs = (("configs[1]: 10-bit, 16 steps, no CFG", PR.RUN_CFG1), ("configs[4] generator: 14-bit, 256 steps, CFG 5.8", PR.RUN_CFG5))
    if os.environ.get("CONFIG_BENCH_ONLY"):
        runs = tuple(r for r in runs if os.environ["CONFIG_BENCH_ONLY"] in r[0])
    for B in ([int(a) for a in sys.argv[1:]] or [64]):
      for name, run in runs:
          g = PR.load_run(run)
          gen, _ = PR.build_models(dev, with_tokenizer=False, name=run)
          tok = tokenizer(int(g["bits"]), dev)
          kw = g["kw"]
          plan = build_plan(int(kw["num_steps"]), 512, float(kw["guidance_scale"]), kw["guidance_annealing"], float(kw["scale_pow"]), 1.0, False,
                            kw["mask_schedule_strategy"])
          labels = (torch.arange(B) * 37 % 1000).to(dev)
          rt = float(kw["randomize_temperature"])
          torch.manual_seed(0)
          run_chunked(gen, tok, labels, plan, rt, want_steps=False, want_image=False, want_u8=True)
          torch.cuda.synchronize()
          n = 3 if int(kw["num_steps"]) < 100 else 1
          _lib.prof_enable(True, every=1)
          t0 = time.perf_counter()
          for _ in range(n): run_chunked(gen, tok, labels, plan, rt, want_steps=False, want_image=False, want_u8=True)
          torch.cuda.synchronize()
          dt = (time.perf_counter() - t0) / n
          prof = _lib.prof_read(); _lib.prof_enable(False)
          print(f"{name}: resolved LFQBert.precision = {gen.resolved_precision()}; {B / dt:.2f} images/s ({dt * 1e3:.0f} ms per batch of {B}; with the event pairs on every launch)", flush=True)
          for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
              print(f"    {k:16s} {c:6d} launches, {ms / c * 1e3:8.1f} us each, {ms / n:8.1f} ms per batch")
          del gen, tok
          torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
