"""Does splitting the batch over two CU-masked streams pay?  (round-2 review item 1a)

Streams from hipExtStreamCreateWithCUMask see a subset of the CUs.  The question: with the 64 CFG pairs split in two halves of 32, each half on
its own 128-CU stream and the second one started half a layer late, do the HBM-bound phases of one half (residual epilogues, LayerNorm) overlap
the matrix phases of the other -- i.e. is [two concurrent half-batch layer sequences on 128 CUs each] faster than [one full-batch sequence on 256]?

Arms (one trunk layer = QKV, attention, out-proj, LayerNorm, FFN-up, FFN-down, LayerNorm through the library's diagnostic entry points, R layers):
  full    : one ordinary stream, 64 pairs
  solo    : ONE masked stream, 32 pairs, the other half of the chip idle (per-kernel: is a kernel bound per CU or chip-wide?)
  dual    : two masked streams, 32 pairs each, the second sequence rotated by half a layer
The census kernel (tools/micro/cu_census.hip) prints which XCCs / CUs a mask really selects.
usage: python tools/cumask_probe.py [layers per arm]"""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import _lib

dev = torch.device("cuda")
lib = _lib.load()


def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 not mapped")


hip = hip_runtime()
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamDestroy.argtypes = [C.c_void_p]


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return s


def census(stream, tag):
    so = os.path.join(ROOT, "tools", "micro", "libcu_census.so")
    if not os.path.exists(so):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "micro", "cu_census.hip"), "-o", so])
    cl = C.CDLL(so)
    cl.cu_census.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    n = 2048
    out = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    cl.cu_census(out.data_ptr(), n, 20, stream)
    hip.hipStreamSynchronize(stream)
    torch.cuda.synchronize()
    o = out.cpu().view(n, 2)
    xcc = (o[:, 0] & 0xF).tolist()
    hw = o[:, 1].tolist()
    cus = sorted({(x, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15) for x, h in zip(xcc, hw)})
    per = {}
    for c in cus:
        per[c[0]] = per.get(c[0], 0) + 1
    print(f"census {tag:10s}: {len(cus)} distinct (xcc, se, sh, cu); per XCC {per}", flush=True)
    return len(cus)


class Layer:
    """Buffers + launch closures of one trunk layer for P pairs."""

    def __init__(self, pairs, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        rn = lambda *s: torch.randn(*s, device=dev, generator=g)
        self.P = P = pairs * 257
        M = 2 * P
        d, f = 1024, 4096
        self.x = rn(M, d).half(); self.x[P:] *= 0.02
        self.qkv = torch.empty(M, 3 * d, device=dev, dtype=torch.float16)
        self.att = torch.empty(M, d, device=dev, dtype=torch.float16)
        self.h = torch.empty(M, f, device=dev, dtype=torch.float16)
        self.y = rn(M, d)
        self.stats = torch.empty(M, 2, device=dev)
        self.aux = torch.empty(P, d, device=dev)
        self.w = [(rn(3 * d, d) * 0.03).half(), (rn(d, d) * 0.03).half(), (rn(f, d) * 0.03).half(), (rn(d, f) * 0.02).half()]
        self.b = [rn(3 * d) * 0.1, rn(d) * 0.1, rn(f) * 0.1, rn(d) * 0.1]
        self.g, self.bt = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        self.pairs = pairs

    def ops(self, st):
        P, d, f = self.P, 1024, 4096
        ck = _lib.check
        gp = lambda epi, A, i, res, o32, o16, N, K: ck(lib.mb_gemm_pair(epi, A.data_ptr(), self.w[i].data_ptr(), self.b[i].data_ptr(), res, o32, o16, P, N, K,
                                                                      None, None, None, None, st))
        ln = lambda: ck(lib.mb_layernorm(self.y.data_ptr(), self.g.data_ptr(), self.bt.data_ptr(), 1e-12, None, self.x.data_ptr(), None, self.stats.data_ptr(),
                                         2 * P, d, st))
        return [("qkv", lambda: gp(0, self.x, 0, None, None, self.qkv.data_ptr(), 3 * d, d)),
                ("attention", lambda: ck(lib.mb_attention_pair(self.qkv.data_ptr(), self.att.data_ptr(), self.pairs, 257, d, 16, st))),
                ("attn_out", lambda: gp(2, self.att, 1, self.y.data_ptr(), self.y.data_ptr(), None, d, d)),
                ("layernorm", ln),
                ("ffn_up", lambda: gp(1, self.x, 2, None, None, self.h.data_ptr(), f, d)),
                ("ffn_down", lambda: gp(2, self.h, 3, self.y.data_ptr(), self.y.data_ptr(), None, d, f)),
                ("layernorm2", ln)]


def run_seq(ops, layers, rot=0):
    n = len(ops)
    for i in range(layers * n):
        ops[(i + rot) % n][1]()


def per_kernel(ops, stream_sync, reps=10):
    out = {}
    for name, fn in ops:
        for _ in range(2): fn()
        stream_sync()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        stream_sync()
        out[name] = (time.perf_counter() - t0) / reps * 1e6
    return out


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cur = torch.cuda.current_stream().cuda_stream
    census(C.c_void_p(cur), "unmasked")
    patterns = {"low128": range(128), "even": range(0, 256, 2), "xcc<4": [b for b in range(256) if b % 8 < 4],
                "hi128": range(128, 256), "odd": range(1, 256, 2), "xcc>=4": [b for b in range(256) if b % 8 >= 4]}
    streams = {}
    for k, bits in patterns.items():
        try:
            streams[k] = masked_stream(list(bits))
            census(streams[k], k)
        except Exception as e:                                  # noqa: BLE001
            print(f"mask {k}: {e}", flush=True)
    full = Layer(64, 1)
    ha, hb = Layer(32, 2), Layer(32, 3)
    sync = torch.cuda.synchronize

    lib.mb_set_cu_count(0)
    ops_full = full.ops(cur)
    pk = per_kernel(ops_full, sync)
    print("full  (64 pairs, 256 CUs)        per kernel us:", {k: round(v, 1) for k, v in pk.items()}, " sum", round(sum(pk.values()), 1), flush=True)
    run_seq(ops_full, 1); sync()
    t0 = time.perf_counter(); run_seq(ops_full, layers); sync()
    t_full = (time.perf_counter() - t0) / layers * 1e6
    print(f"full  : {t_full:8.1f} us per layer (64 pairs)", flush=True)

    # half batch on the ordinary stream with full-size grids: what chunking the batch in two sequential halves would cost
    ops_half_all = ha.ops(cur)
    pk = per_kernel(ops_half_all, sync)
    print("half  (32 pairs, 256 CUs)        per kernel us:", {k: round(v, 1) for k, v in pk.items()}, " sum", round(sum(pk.values()), 1), flush=True)

    for (ka, kb) in (("low128", "hi128"), ("even", "odd"), ("xcc<4", "xcc>=4")):
        if ka not in streams or kb not in streams:
            continue
        sa, sb = streams[ka], streams[kb]
        lib.mb_set_cu_count(128)
        oa, ob = ha.ops(sa), hb.ops(sb)
        pk = per_kernel(oa, lambda: hip.hipStreamSynchronize(sa))
        print(f"solo  {ka:7s} (32 pairs, 128 CUs) per kernel us:", {k: round(v, 1) for k, v in pk.items()}, " sum", round(sum(pk.values()), 1), flush=True)
        for rot in (0, 3, 4):
            run_seq(oa, 1); run_seq(ob, 1, rot); sync()
            t0 = time.perf_counter()
            # interleave the host-side launches so neither stream starves
            n = len(oa)
            for i in range(layers * n):
                oa[i % n][1](); ob[(i + rot) % n][1]()
            sync()
            t_dual = (time.perf_counter() - t0) / layers * 1e6
            print(f"dual  {ka}/{kb} rot {rot}: {t_dual:8.1f} us per layer of 2 x 32 pairs   (full: {t_full:.1f})  ratio {t_dual / t_full:.3f}", flush=True)
        lib.mb_set_cu_count(0)


if __name__ == "__main__":
    main()
