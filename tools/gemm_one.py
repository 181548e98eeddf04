"""Run one trunk GEMM shape/variant a few times (target for rocprofv3 --pmc passes).
usage: python tools/gemm_one.py <qkv|attn_out|ffn_up|ffn_down> <variant> [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run, dev
name, variant = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
epi, N, K = {"qkv": (0, 3072, 1024), "attn_out": (2, 1024, 1024), "ffn_up": (1, 4096, 1024), "ffn_down": (2, 1024, 4096)}[name]
M = 128 * 257
torch.manual_seed(0)
A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
bias = torch.randn(N, device=dev) * 0.1
res = torch.randn(M, N, device=dev) if epi == 2 else None
for _ in range(iters):
    run(epi, A, W, bias, res, M, N, K, variant)
torch.cuda.synchronize()
print("done")
