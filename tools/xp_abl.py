"""Main-loop ablations of the half-tile GEMM (timing only; outputs are garbage): variant 18 = DMA only, 28 = no DMA in the loop."""
import sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from xp_gemm import timeit
for v, name in ((8, "full (MT=8)"), (18, "DMA only"), (28, "LDS reads + MFMA only")):
    t1, t4 = timeit(2, 3072, 1024, v, 20), timeit(2, 3072, 4096, v, 20)
    tiles = (128 * 257 + 255) // 256 * 12
    rounds = (tiles + 255) // 256
    print(f"{name:24s}: K=1024 {t1:7.1f} us  K=4096 {t4:7.1f} us  -> per K-tile {(t4 - t1) / 48 / rounds:.3f} us ({rounds} rounds)", flush=True)
