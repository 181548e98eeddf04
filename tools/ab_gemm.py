"""In-run A/B of two builds of the library on the four trunk GEMM shapes (interleaved, so box/thermal state is shared).
usage: python tools/ab_gemm.py tools/_ab/lib_prev.so [more.so ...]   (the in-tree library is always the last column)"""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import _lib
dev = torch.device("cuda")
libs = []
for path in sys.argv[1:]:
    l = ctypes.CDLL(os.path.abspath(path)); 
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(l, name): getattr(l, name).restype = res; getattr(l, name).argtypes = args
    libs.append((os.path.basename(path), l))
libs.append(("tree", _lib.load()))
M = 128 * 257
shapes = [("qkv", 0, 3072, 1024), ("attn_out", 2, 1024, 1024), ("ffn_up", 1, 4096, 1024), ("ffn_down", 2, 1024, 4096)]
torch.manual_seed(0)
for name, epi, N, K in shapes:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev) * 0.1; res = torch.randn(M, N, device=dev) if epi == 2 else None
    o32 = torch.empty(M, N, device=dev) if epi == 2 else None; o16 = torch.empty(M, N, device=dev, dtype=torch.float16) if epi != 2 else None
    def go(l, n):
        for _ in range(n):
            rc = l.mb_gemm(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None,
                           o32.data_ptr() if o32 is not None else None, o16.data_ptr() if o16 is not None else None, M, N, K, 0, 0,
                           torch.cuda.current_stream().cuda_stream)
            assert rc == 0
    outs = []
    tot = {n: 0.0 for n, _ in libs}
    for rep in range(6):
        for n, l in libs:
            go(l, 3); torch.cuda.synchronize(); t0 = time.perf_counter(); go(l, 20); torch.cuda.synchronize()
            if rep: tot[n] += (time.perf_counter() - t0) / 20 * 1e6 / 5
    for n, l in libs:
        go(l, 1); torch.cuda.synchronize(); outs.append((o32 if o32 is not None else o16).clone())
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{name:9s}: " + "  ".join(f"{n} {t:7.1f} us" for n, t in tot.items()) + f"   outputs identical: {same}", flush=True)
