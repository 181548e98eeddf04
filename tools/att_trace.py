"""Phase timeline of the CFG pair attention (two launches: conditional sequences, then their unconditional twins): a copy of the library built with
-DMB_ATT_TRACE stamps the 100 MHz wall clock in wave 0 of every workgroup (attention.hip: MB_ATRACE).
  python tools/att_trace.py build   (here: compiles tools/_ab/libatt_trace.so)
  python tools/att_trace.py run     (on the GPU box)"""
import ctypes as C
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "tools", "_ab")


def build():
    from maskbit_amd import build as B
    os.makedirs(AB, exist_ok=True)
    objs, procs = [], []
    for src in B.SOURCES:
        obj = os.path.join(AB, f"att_trace_{src.replace('.hip', '.o')}")
        objs.append(obj)
        procs.append(subprocess.Popen([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DMB_ATT_TRACE=1",
                                       "-c", os.path.join(B.CSRC, src), "-o", obj]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", os.path.join(AB, "libatt_trace.so")])
    print("built", os.path.join(AB, "libatt_trace.so"))


VARIANTS = {"base": []}      # (round 2: nospipe / gk2 / gk6 / gk9 = -DMB_ATT_NOSPIPE=1 / -DMB_ATT_SDEPTH=2, 6, 9)


def build_variants():
    """Experimental builds of attention.hip alone (tools/_ab/libatt_var_<name>.so), timed by `run` next to the product library."""
    from maskbit_amd import build as B
    os.makedirs(AB, exist_ok=True)
    procs = []
    for name, flags in VARIANTS.items():
        procs.append(subprocess.Popen([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-DMB_ATT_VARIANT=1", *flags,
                                       os.path.join(B.CSRC, "attention.hip"), "-o", os.path.join(AB, f"libatt_var_{name}.so")]))
    assert all(p.wait() == 0 for p in procs)


def run():
    import numpy as np
    import torch
    from maskbit_amd import _lib
    _lib.LIB_PATH = os.path.join(AB, "libatt_trace.so")
    lib = _lib.load()
    lib.mb_debug_att_trace.restype = C.c_int; lib.mb_debug_att_trace.argtypes = [C.c_void_p]
    lib.mb_debug_attention_pair.restype = C.c_int
    lib.mb_debug_attention_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    P, N, d, heads = (int(sys.argv[2]) if len(sys.argv) > 2 else 64), 257, 1024, 16
    qkv = (torch.randn(2 * P * N, 3 * d, device=dev) * 0.5).half()
    out = torch.empty(2 * P * N, d, device=dev, dtype=torch.float16)
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: lib.mb_debug_attention_pair(qkv.data_ptr(), out.data_ptr(), P, N, d, heads, st)
    plib = C.CDLL(os.path.join(ROOT, "maskbit_amd", "libmaskbit_hip.so"))           # the product library, for the uninstrumented time
    plib.mb_attention_pair.restype = C.c_int
    plib.mb_attention_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    pfn = lambda: plib.mb_attention_pair(qkv.data_ptr(), out.data_ptr(), P, N, d, heads, st)
    for _ in range(3): assert pfn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): pfn()
    e1.record(); torch.cuda.synchronize()
    print(f"product library: pair attention, {P} sequence pairs: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    prev = os.path.join(AB, "lib_prev.so")                                          # a copy of an earlier product build, for A/B in one process
    if os.path.exists(prev):
        qlib = C.CDLL(prev)
        qlib.mb_attention_pair.restype = C.c_int
        qlib.mb_attention_pair.argtypes = plib.mb_attention_pair.argtypes
        qfn = lambda: qlib.mb_attention_pair(qkv.data_ptr(), out.data_ptr(), P, N, d, heads, st)
        for rep in range(3):
            for name, f in (("previous build", qfn), ("product library", pfn)):
                for _ in range(3): assert f() == 0
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize()
                print(f"  A/B {name:16s}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    import glob
    ref = None
    for path in sorted(glob.glob(os.path.join(AB, "libatt_var_*.so"))):
        vlib = C.CDLL(path)
        vlib.mb_debug_attention_pair.restype = C.c_int
        vlib.mb_debug_attention_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        vfn = lambda: vlib.mb_debug_attention_pair(qkv.data_ptr(), out.data_ptr(), P, N, d, heads, st)
        if ref is None:
            pfn(); torch.cuda.synchronize(); ref = out.clone()
        out.zero_()
        for _ in range(3): assert vfn() == 0
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        e0.record()
        for _ in range(20): vfn()
        e1.record(); torch.cuda.synchronize()
        print(f"variant {os.path.basename(path)[11:-3]:8s}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  (output {'identical to' if same else 'DIFFERS from'} the product's)")
    if len(sys.argv) > 3 and sys.argv[3] == "notrace": return
    for _ in range(3): assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"trace library:   pair attention (two launches), {P} sequence pairs: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    G = P * heads
    trace = torch.zeros(2 * G, 32, dtype=torch.int64, device=dev)
    assert lib.mb_debug_att_trace(trace.data_ptr()) == 0
    fn(); torch.cuda.synchronize()
    assert lib.mb_debug_att_trace(None) == 0
    t = trace.cpu().numpy().astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    for name, sl in (("conditional pass", slice(0, G)), ("unconditional pass", slice(G, 2 * G))):
        x = t[sl]
        print(f"== {name}: workgroups start {x[:, 0].min() - t0:.1f} .. {x[:, 0].max() - t0:.1f} us, last ends {x[:, 22].max() - t0:.1f} us")
        print(f"   issue of K/V DMA + Q loads {np.mean(x[:, 1] - x[:, 0]):.2f} | wait until landed + barrier {np.mean(x[:, 2] - x[:, 1]):.2f} us")
        for i in range(5):
            b = 3 + 4 * i
            prev = x[:, 2] if i == 0 else x[:, b - 1]
            print(f"   query tile {i} of wave 0: scores {np.mean(x[:, b] - prev):.2f} | softmax {np.mean(x[:, b + 1] - x[:, b]):.2f} | PV {np.mean(x[:, b + 2] - x[:, b + 1]):.2f} | "
                  f"stores {np.mean(x[:, b + 3] - x[:, b + 2]):.2f} us")
        print(f"   workgroup lifetime {np.mean(x[:, 22] - x[:, 0]):.2f} us (min {np.min(x[:, 22] - x[:, 0]):.2f}, max {np.max(x[:, 22] - x[:, 0]):.2f})")
        # concurrency: how many workgroups are alive at once, sampled
        ts = np.linspace(x[:, 0].min(), x[:, 22].max(), 50)
        alive = [(np.sum((x[:, 0] <= u) & (x[:, 22] > u))) for u in ts]
        print(f"   workgroups alive (median over the launch): {int(np.median(alive))}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build": build()
    elif len(sys.argv) > 1 and sys.argv[1] == "variants": build_variants()
    else: run()
