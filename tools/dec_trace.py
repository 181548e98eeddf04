"""Phase timeline of the decoder's conv kernel: a copy of the library with decoder.hip built -DMB_CONV_TRACE stamps the 100 MHz wall clock at the
phase boundaries of every workgroup of ONE selected conv launch of a decode (decoder.hip: MB_CTRACE).  Answers: how long does a workgroup live, how
much of that is halo staging, barrier waits, tap steps, epilogue?
  python tools/dec_trace.py build        (here: tools/_ab/libdectrace.so)
  python tools/dec_trace.py run [B=64] [launch indices ...]   (on the GPU box; default: every conv launch of a decode)"""
import ctypes as C
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "tools", "_ab")
LIB = os.path.join(AB, "libdectrace.so")


def build():
    from maskbit_amd import build as B
    os.makedirs(AB, exist_ok=True)
    B.build()
    obj = os.path.join(AB, "dectrace_decoder.o")
    subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DMB_CONV_TRACE=1",
                           "-c", os.path.join(B.CSRC, "decoder.hip"), "-o", obj])
    objs = [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "decoder.hip"] + [obj]
    subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    print("built", LIB)


def run(B, sels):
    import numpy as np
    import torch
    from maskbit_amd import _lib
    _lib.LIB_PATH = LIB
    from maskbit_amd import ConvVQModel, synth

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    tok = ConvVQModel(Cfg(quantizer_type="lookup-free", codebook_size=4096, token_size=12, num_channels=3, hidden_channels=128,
                          channel_mult=[1, 1, 2, 2, 4], num_resolutions=5, num_res_blocks=2, sample_with_conv=True))
    tok.load_state_dict(synth.make_tokenizer_weights(synth.TokCfg(token_size=12), seed=200), strict=False)
    tok = tok.eval().requires_grad_(False).to("cuda")
    t = torch.randint(0, 4096, (B, 256), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    for _ in range(2): tok.decode_tokens_uint8(t)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.mb_debug_conv_trace.restype = C.c_int
    lib.mb_debug_conv_trace.argtypes = [C.c_void_p, C.c_int]
    NWG = 1 << 18
    trace = torch.zeros(NWG, 16, dtype=torch.int64, device="cuda")
    for sel in (sels or range(40)):
        trace.zero_()
        torch.cuda.synchronize()
        lib.mb_debug_conv_trace(trace.data_ptr(), sel)
        tok.decode_tokens_uint8(t)
        torch.cuda.synchronize()
        lib.mb_debug_conv_trace(None, -1)
        tr = trace.cpu().numpy().astype(np.float64) * 0.01     # us
        live = tr[:, 0] > 0
        if not live.any():
            break
        tr = tr[live]
        t0 = tr[:, 0].min()
        span = tr[:, 15].max() - t0
        life = tr[:, 15] - tr[:, 0]
        nchunk = int(sum(1 for c in range(3) if (tr[:, 4 + 4 * c] > 0).all()))
        print(f"   {len(tr)} workgroups, kernel span {span:.0f} us, workgroup lifetime mean {life.mean():.1f} us (min {life.min():.1f}, max {life.max():.1f})")
        prev = 0
        for c in range(nchunk):
            d = lambda a, b: (tr[:, b] - tr[:, a]).mean()
            print(f"   chunk {c}: wait for halo free {d(prev, 1 + 4 * c):6.2f} | stage halo (own wave) {d(1 + 4 * c, 2 + 4 * c):6.2f} | wait + barrier {d(2 + 4 * c, 3 + 4 * c):6.2f} | taps {d(3 + 4 * c, 4 + 4 * c):6.2f}")
            prev = 4 + 4 * c
        print(f"   epilogue {(tr[:, 15] - tr[:, prev]).mean():6.2f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build": build()
    else: run(int(sys.argv[2]) if len(sys.argv) > 2 else 64, [int(a) for a in sys.argv[3:]])
