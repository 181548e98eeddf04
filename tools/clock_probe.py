"""Is the trunk GEMM bound by the chip's power budget?  Runs the FFN-up pair GEMM back to back for a few seconds with persistent grids of 256, 192, 128
and 64 workgroups (mb_set_cu_count: the other CUs idle) and samples rocm-smi (shader clock, socket power) meanwhile.  If a CU were the limit, time
per launch would scale as 256 / G; it does not: with half the CUs busy every CU runs ~1.35x faster.
usage: python tools/clock_probe.py"""
import json, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import _lib

if os.environ.get("CLOCK_PROBE_LIB"):                 # e.g. tools/_ab/libbf16.so: the library built with -DMB_HALF_BF16=1
    _lib.LIB_PATH = os.path.abspath(os.environ["CLOCK_PROBE_LIB"])
lib = _lib.load()
dev = torch.device("cuda")
HALF = torch.bfloat16 if "bf16" in os.environ.get("CLOCK_PROBE_LIB", "") else torch.float16


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
        j = json.loads(out)
        card = next(iter(j.values()))
        sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
        pw = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
        return sclk, pw
    except Exception as e:      # noqa: BLE001
        return repr(e), None


def main():
    torch.manual_seed(0)
    P = 64 * 257; M = 2 * P
    st = torch.cuda.current_stream().cuda_stream
    shapes = {"ffn_up": (1, 4096, 1024), "qkv": (0, 3072, 1024)}
    for name, (epi, N, K) in shapes.items():
        A = torch.randn(M, K, device=dev).to(HALF); A[P:] *= 0.02
        W = (torch.randn(N, K, device=dev) * 0.03).to(HALF)
        bias = torch.randn(N, device=dev) * 0.1
        o16 = torch.empty(M, N, device=dev, dtype=HALF)
        fn = lambda: _lib.check(lib.mb_gemm_pair(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, None, o16.data_ptr(), P, N, K, None, None, None, None, st))
        for data in ("random", "zeros"):
            if data == "zeros":
                A.zero_(); W.zero_()
            for G in ((256, 128) if os.environ.get("CLOCK_PROBE_LIB") else (256, 192, 128, 64)):
                lib.mb_set_cu_count(G if G != 256 else 0)
                for _ in range(5): fn()
                torch.cuda.synchronize()
                samples, stop = [], False

                def sampler():
                    while not stop:
                        samples.append(smi_sample()); time.sleep(0.25)
                th = threading.Thread(target=sampler); th.start()
                n = 0
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 2.5:
                    for _ in range(50): fn()
                    torch.cuda.synchronize(); n += 50
                dt = (time.perf_counter() - t0) / n
                stop = True; th.join()
                flops = 2.0 * M * N * K
                mid = samples[len(samples) // 2:] or samples
                print(f"{name:7s} {data:6s} grid {G:3d} workgroups: {dt * 1e6:8.1f} us per launch = {flops / dt / 1e12:7.1f} TFLOP/s; per busy CU x{flops / dt / G / (1e12):.3f} TF; "
                      f"smi (sclk, W) {mid[-1] if mid else None}", flush=True)
        lib.mb_set_cu_count(0)


if __name__ == "__main__":
    main()
