"""Time of the row LayerNorm kernel by what it writes (fp16 hi / + lo halves / + row statistics / + e2m1 copy), M rows of 1024.
usage: python tools/ln_bench.py [sequences ...]   (default 64 16)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import _lib


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream
    d = 1024
    for nb in ([int(a) for a in sys.argv[1:]] or [64, 16]):
        M = nb * 257
        y = torch.randn(M, d, device=dev); g = torch.rand(d, device=dev) + 0.5; b = torch.randn(d, device=dev) * 0.1
        hi = torch.empty(M, d, device=dev, dtype=torch.float16); lo = torch.empty_like(hi); stats = torch.empty(M, 2, device=dev)
        x4 = torch.zeros(M, 2 * d, device=dev, dtype=torch.uint8); x4s = torch.zeros((d // 64) * nb * 256 + 256, device=dev, dtype=torch.uint8)
        xl4 = torch.zeros_like(x4); xl4s = torch.zeros_like(x4s)
        P = lambda t: t.data_ptr() if t is not None else None
        cases = {
            "fp16 hi": lambda: lib.mb_layernorm(P(y), P(g), P(b), 1e-12, None, P(hi), None, None, M, d, st()),
            "hi + stats": lambda: lib.mb_layernorm(P(y), P(g), P(b), 1e-12, None, P(hi), None, P(stats), M, d, st()),
            "hi + lo + stats": lambda: lib.mb_layernorm(P(y), P(g), P(b), 1e-12, None, P(hi), P(lo), P(stats), M, d, st()),
            "hi + e2m1 values": lambda: lib.mb_layernorm_f4(P(y), P(g), P(b), 1e-12, None, P(hi), P(x4), P(x4s), None, None, M, d, st()),
            "hi + e2m1 values + e2m1 lo": lambda: lib.mb_layernorm_f4(P(y), P(g), P(b), 1e-12, None, P(hi), P(x4), P(x4s), P(xl4), P(xl4s), M, d, st()),
        }
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for name, fn in cases.items():
            for _ in range(3): assert fn() == 0
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0.record()
                for _ in range(50): fn()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 50 * 1e3)
            print(f"{nb:3d} sequences ({M} rows): {name:28s} {min(ts):6.1f} us  ({' '.join(f'{t:.1f}' for t in ts)})", flush=True)


if __name__ == "__main__":
    main()
