"""Run one trunk GEMM shape as the engine's guided forward runs it (CFG pair tiles, B = 64 pairs) a few times: target for rocprofv3 --pmc passes.
usage: [PAIR_ONE_MINI=n] python tools/pair_one.py <qkv|attn_out|ffn_up|ffn_down> [iters]    (PAIR_ONE_MINI: with n MX-fp4 mini-tile operand sets:
                                                                                             1 = the weight correction of the product default)"""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskbit_amd import _lib
lib = _lib.load()
name = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
epi, N, K = {"qkv": (0, 3072, 1024), "attn_out": (2, 1024, 1024), "ffn_up": (1, 4096, 1024), "ffn_down": (2, 1024, 4096)}[name]
P = 64 * 257
M = 2 * P
dev = torch.device("cuda")
torch.manual_seed(0)
A = torch.randn(M, K, device=dev).half(); A[P:] *= 0.01
W = (torch.randn(N, K, device=dev) * 0.05).half()
bias = torch.randn(N, device=dev) * 0.1
res = torch.randn(M, N, device=dev) if epi == 2 else None
o16 = torch.empty(M, N, device=dev, dtype=torch.float16) if epi != 2 else None
ptr = lambda t: t.data_ptr() if t is not None else None
nlo = int(os.environ.get("PAIR_ONE_MINI", "0"))
sets = []
for _ in range(nlo):
    x4 = torch.randint(0, 256, (M, 2 * K), device=dev, dtype=torch.uint8)
    xs = torch.full(((K // 64) * 64 * 256 + 256,), 100, device=dev, dtype=torch.uint8)
    w4 = torch.zeros(N, K // 2, device=dev, dtype=torch.uint8); ws = torch.zeros(N * K // 128, device=dev, dtype=torch.uint8)
    _lib.check(lib.mb_w4_from_f32(W.float().data_ptr(), N, K, w4.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream))
    sets += [x4, xs, w4, ws]
arr = (C.c_void_p * max(1, len(sets)))(*[t.data_ptr() for t in sets])
h4 = torch.zeros(M, 2 * N, device=dev, dtype=torch.uint8) if (epi == 1 and nlo) else None           # FFN-up also emits the e2m1 copy of its outputs
h4s = torch.zeros((N // 64) * 64 * 256 + 256, device=dev, dtype=torch.uint8) if (epi == 1 and nlo) else None
for _ in range(iters):
    _lib.check(lib.mb_gemm_mini(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), ptr(res), ptr(res), ptr(o16), ptr(h4), ptr(h4s), P, 1, N, K, nlo, arr,
                                torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("done")
