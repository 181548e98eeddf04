"""A/B of two builds of the library on the four trunk pair GEMMs inside one process: tools/_ab/lib_prev.so (a copy of an earlier build) against the
product library, alternated, 64 sequence pairs (GEMM_AB_MINI = n: with n MX-fp4 mini-tile operand sets).
usage: cp maskbit_amd/libmaskbit_hip.so tools/_ab/lib_prev.so; <change, rebuild>; python tools/gemm_ab.py"""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import _lib


def main():
    sig = _lib.SIGNATURES["mb_gemm_mini"]
    nlo = int(os.environ.get("GEMM_AB_MINI", "0"))
    libs = {}
    for name, path in (("previous", os.path.join(ROOT, "tools", "_ab", "lib_prev.so")), ("product", os.path.join(ROOT, "maskbit_amd", "libmaskbit_hip.so"))):
        l = C.CDLL(path)
        l.mb_gemm_mini.restype, l.mb_gemm_mini.argtypes = sig
        l.mb_w4_from_f32.restype, l.mb_w4_from_f32.argtypes = _lib.SIGNATURES["mb_w4_from_f32"]
        libs[name] = l
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: t.data_ptr() if t is not None else None
    torch.manual_seed(0)
    P = 64 * 257
    M = 2 * P
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = {k: 0.0 for k in libs}
    for name, epi, N, K in [("qkv", 0, 3072, 1024), ("attn_out", 2, 1024, 1024), ("ffn_up", 1, 4096, 1024), ("ffn_down", 2, 1024, 4096)]:
        A = torch.randn(M, K, device=dev).half(); A[P:] *= 0.01
        W = (torch.randn(N, K, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev) * 0.1
        res = torch.randn(M, N, device=dev) if epi == 2 else None
        outs = {}
        acc = {k: [] for k in libs}
        sets = {}
        for k, l in libs.items():                        # each build packs its own e2m1 weight operand (the layout may differ between builds)
            ts = []
            for _ in range(nlo):
                x4 = torch.randint(0, 256, (M, 2 * K), device=dev, dtype=torch.uint8, generator=torch.Generator(device=dev).manual_seed(1))
                xsb = torch.full(((K // 64) * 64 * 256 + 256,), 100, device=dev, dtype=torch.uint8)
                w4 = torch.zeros(N, 2 * K, device=dev, dtype=torch.uint8); ws = torch.zeros(N * K // 128, device=dev, dtype=torch.uint8)
                assert l.mb_w4_from_f32(W.float().data_ptr(), N, K, w4.data_ptr(), ws.data_ptr(), st) == 0
                ts += [x4, xsb, w4, ws]
            sets[k] = (ts, (C.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts]))
        for rnd in range(4):
            for k, l in (list(libs.items()) if rnd % 2 == 0 else list(libs.items())[::-1]):
                o32 = res.clone() if epi == 2 else None
                o16 = torch.empty(M, N, device=dev, dtype=torch.float16) if epi != 2 else None
                h4 = h4s = None
                if epi == 1 and os.environ.get("GEMM_AB_OUT4"):          # the GELU epilogue's e2m1 copy of the conditional outputs (FFN-down's token operand)
                    h4 = torch.zeros(P, 2 * N, device=dev, dtype=torch.uint8); h4s = torch.zeros((N // 64) * 64 * 256 + 256, device=dev, dtype=torch.uint8)
                fn = lambda: l.mb_gemm_mini(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), ptr(res), ptr(o32), ptr(o16), ptr(h4), ptr(h4s), P, 1, N, K, nlo, sets[k][1], st)
                assert fn() == 0
                torch.cuda.synchronize()
                outs[k] = (o32 if epi == 2 else o16).clone()
                for _ in range(3): fn()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                acc[k].append(e0.elapsed_time(e1) / 20 * 1e3)
        same = torch.equal(outs["previous"], outs["product"])
        for k in libs:
            tot[k] += sum(acc[k]) / len(acc[k])
        print(f"{name:9s}: " + "   ".join(f"{k} {sum(v) / len(v):7.1f} us ({' '.join(f'{x:.1f}' for x in v)})" for k, v in acc.items()) + f"   outputs {'identical' if same else 'DIFFER'}")
    print("sum of the four: " + "   ".join(f"{k} {v:.1f} us" for k, v in tot.items()))


if __name__ == "__main__":
    main()
