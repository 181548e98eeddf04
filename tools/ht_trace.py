"""Phase timeline of the half-tile GEMM on the trunk shapes: a copy of the library built with -DMB_HT_TRACE stamps the 100 MHz wall clock at the
phase boundaries of every workgroup's tiles (gemm_ht.hip: MB_TRACE).  Answers: how long is the K loop of a tile, what does a tile pay outside it,
and do all CUs hit their epilogues at the same time?
  [HT_DEFS="-DX=1 .."] python tools/ht_trace.py build      (here: compiles tools/_ab/libtrace{1,2,3,4}.so with the extra defines; mode 2 also waits for the stores at the end of every tile;
                                       modes 3 / 4 (results are garbage, timing only): the K loop issues no DMA / re-reads K-tiles 0 and 1 = pure L2 hits)
  python tools/ht_trace.py run [mode] (on the GPU box)"""
import ctypes as C
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "tools", "_ab")
TAG = os.environ.get("HT_TAG", "")            # suffix of the library name: several builds with different HT_DEFS side by side


def build():
    from maskbit_amd import build as B
    os.makedirs(AB, exist_ok=True)
    for mode in ([int(a) for a in sys.argv[2:]] or (1, 2, 3, 4)):
        objs, procs = [], []
        for src in B.SOURCES:
            obj = os.path.join(AB, f"trace{mode}{TAG}_{src.replace('.hip', '.o')}")
            objs.append(obj)
            procs.append(subprocess.Popen([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", f"-DMB_HT_TRACE={mode % 10}", *(["-DMB_BS_NOLOAD=1"] if mode >= 10 else []),
                                           *os.environ.get("HT_DEFS", "").split(), "-c", os.path.join(B.CSRC, src), "-o", obj]))
        assert all(p.wait() == 0 for p in procs)
        subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", os.path.join(AB, f"libtrace{mode}{TAG}.so")])
        print("built", os.path.join(AB, f"libtrace{mode}{TAG}.so"))


def run(mode):
    import numpy as np
    import torch
    from maskbit_amd import _lib
    _lib.LIB_PATH = os.path.join(AB, f"libtrace{mode}{TAG}.so")
    lib = _lib.load()
    lib.mb_debug_ht_trace.restype = C.c_int
    lib.mb_debug_ht_trace.argtypes = [C.c_void_p]
    dev = torch.device("cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream
    ptr = lambda t: t.data_ptr() if t is not None else None
    torch.manual_seed(0)
    # HT_PLAIN=n (round 6): PLAIN sequence tiles over n sequences instead of pair tiles over 64 pairs -- with HT_MINI=1 and MASKBIT_AMD_COL_SPLIT=1 / 2 / 4 the
    # fp32 + residual GEMMs then walk whole / half- / quarter-column tiles (256 x 256 / 128 / 64 outputs per tile): what a K-tile costs by tile WIDTH
    plain = int(os.environ.get("HT_PLAIN", "0"))
    P = (plain or 64) * 257
    M = P if plain else 2 * P
    shapes = [("qkv", 0, 3072, 1024), ("attn_out", 2, 1024, 1024), ("ffn_up", 1, 4096, 1024), ("ffn_down", 2, 1024, 4096)]
    if plain:
        shapes = [s_ for s_ in shapes if s_[1] == 2]
    for name, epi, N, K in shapes:
        A = torch.randn(M, K, device=dev).half()
        if not plain: A[P:] *= 0.01
        W = (torch.randn(N, K, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev) * 0.1
        res = torch.randn(M, N, device=dev) if epi == 2 else None
        o32 = torch.empty(M, N, device=dev) if epi == 2 else None
        o16 = torch.empty(M, N, device=dev, dtype=torch.float16) if epi != 2 else None
        nlo = int(os.environ.get("HT_MINI", "0"))        # with 1 / 2 MX-fp4 mini-tile operand sets on the conditional half (precision 2 / 3)
        sets = []
        for _ in range(nlo):
            x4 = torch.randint(0, 256, (M, 2 * K), device=dev, dtype=torch.uint8)
            xsb = torch.full(((K // 64) * 64 * 256 + 256,), 100, device=dev, dtype=torch.uint8)
            w4 = torch.zeros(N, 2 * K, device=dev, dtype=torch.uint8); ws = torch.zeros(N * K // 128, device=dev, dtype=torch.uint8)
            _lib.check(lib.mb_w4_from_f32(W.float().data_ptr(), N, K, w4.data_ptr(), ws.data_ptr(), st()))
            sets += [x4, xsb, w4, ws]
        arr = (C.c_void_p * max(1, len(sets)))(*[t.data_ptr() for t in sets])
        fn = lambda: _lib.check(lib.mb_gemm_mini(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), ptr(res), ptr(o32), ptr(o16), None, None, P, 0 if plain else 1, N, K, nlo, arr, st()))
        G = int(os.environ.get("MASKBIT_AMD_HT_GRID", 256))
        trace = torch.zeros(256, 8, 8, dtype=torch.int64, device=dev)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        assert lib.mb_debug_ht_trace(trace.data_ptr()) == 0
        fn(); torch.cuda.synchronize()
        assert lib.mb_debug_ht_trace(None) == 0
        t = trace.cpu().numpy().astype(np.float64)[:G] * 0.01         # us
        t = t[t[:, 0, 0] > 0]                                           # (a grid smaller than 256 workgroups leaves the other rows unstamped)
        ntile = int((t[0, :, 0] > 0).sum())
        t0 = t[:, 0, 0].min()
        print(f"== {name}: N={N} K={K}, {ntile} tiles per workgroup; kernel span {t[:, :ntile, 5 if mode == 2 else 4].max() - t0:.1f} us", flush=True)
        print("   tile | start (mean, spread over CUs) | K loop | pass 1 (bias/GELU) | DMA wait + barrier | output pass (issue) | store drain | gap to next")
        for i in range(ntile):
            d = lambda a, b: (t[:, i, b] - t[:, i, a]).mean()
            start = t[:, i, 0] - t0
            gap = (t[:, i + 1, 0] - t[:, i, 5 if mode == 2 else 4]).mean() if i + 1 < ntile else float("nan")
            drain = d(4, 5) if mode == 2 else float("nan")
            print(f"   {i:4d} | {start.mean():8.2f}  +-{start.std():5.2f} (min {start.min():7.2f} max {start.max():7.2f}) | {d(0, 1):6.2f} | {d(1, 2):6.2f} | {d(2, 3):6.2f} | {d(3, 4):6.2f} | {drain:6.2f} | {gap:6.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build": build()
    else: run(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
