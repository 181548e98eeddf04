"""Per-tile timeline of the persistent trunk GEMM. Needs the probe build: `patch -p0 < tools/patches/gemm_ht_probe.patch`, rebuild
(period = -9 then makes thread 0 of every workgroup log wall_clock64 / clock64 into the out_f32 buffer)."""
import sys, os, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from gemm_bench import dev, lib, _lib
M = 128 * 257
for epi, N, K in ((0, 3072, 1024), (1, 4096, 1024), (0, 3072, 4096)):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev) * 0.1
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    rounds = 128 * (N // 256) // 256
    tb = torch.zeros(rounds * 256 * 4, device=dev, dtype=torch.int64)
    for period in (0, 0, -9):
        _lib.check(lib.mb_gemm(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, tb.data_ptr(), o16.data_ptr(), M, N, K, period, 257,
                               torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    t = tb.cpu().reshape(rounds, 256, 4).double() * 0.01   # us (100 MHz)
    t0 = t[0, :, 0].min()
    print(f"epi={epi} N={N} K={K}: kernel span {float(t[:, :, 2].max() - t0):.1f} us")
    for r in range(rounds):
        st, ml, ep = t[r, :, 0], t[r, :, 1], t[r, :, 2]
        print(f"  round {r}: start {float((st - t0).mean()):7.1f} (min {float((st - t0).min()):6.1f} max {float((st - t0).max()):6.1f})  "
              f"main loop {float((ml - st).mean()):6.2f} (max {float((ml - st).max()):6.2f}; {float((t[r, :, 3] * 100 / (ml - st)).mean()):6.0f} MHz)  epilogue {float((ep - ml).mean()):5.2f} (max {float((ep - ml).max()):5.2f})")
