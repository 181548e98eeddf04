"""Guided-forward time per pair against the number of pairs per pass (does a smaller working set -- residual stream + operands inside the 256 MiB
Infinity Cache -- pay for the shorter tile lists?).  usage: python tools/batch_sweep.py [pairs ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from maskbit_amd import parity_replay as PR


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 32, 48, 64]
    gen, _ = PR.build_models(torch.device("cuda"), with_tokenizer=False)
    for B in sizes:
        tok = torch.randint(0, 64, (B, 256, 2), device="cuda")
        y = torch.randint(0, 1000, (B,), device="cuda")
        for _ in range(3): gen.forward_cfg(tok, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 12
        for _ in range(n): gen.forward_cfg(tok, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        print(f"B = {B:3d} pairs: {dt:8.3f} ms per guided forward = {dt / B * 1e3:7.1f} us per pair", flush=True)


if __name__ == "__main__":
    main()
