"""Correctness + timing of the 4-wave GEMM (variant 4) against the half-tile kernel (variant 0)."""
import sys, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from gemm_bench import run, reference, dev
M = 128 * 256          # whole 256-row tiles for both kernels (the prototype has no sequence-aligned mode yet)
for name, epi, N, K in [("qkv", 0, 3072, 1024), ("attn_out", 2, 1024, 1024), ("ffn_up", 1, 4096, 1024), ("ffn_down", 2, 1024, 4096)]:
    torch.manual_seed(0)
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev) * 0.1; res = torch.randn(M, N, device=dev) if epi == 2 else None
    o0 = run(epi, A, W, bias, res, M, N, K, 8); o4 = run(epi, A, W, bias, res, M, N, K, 4)
    torch.cuda.synchronize()
    ref = reference(epi, A[:2048], W, bias, res[:2048] if res is not None else None)
    e4 = float((o4[:2048].float() - ref).abs().max()); d = float((o4.float() - o0.float()).abs().max())
    ts = {}
    for v in (8, 4, 8, 4):
        for _ in range(3): run(epi, A, W, bias, res, M, N, K, v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run(epi, A, W, bias, res, M, N, K, v)
        torch.cuda.synchronize(); ts[v] = (time.perf_counter() - t0) / 20 * 1e6
    fl = 2.0 * M * N * K
    print(f"{name:9s}: ht {ts[8]:7.1f} us ({fl / ts[8] / 1e6:6.0f} TF)   w4 {ts[4]:7.1f} us ({fl / ts[4] / 1e6:6.0f} TF)   max err vs torch {e4:.4f}, vs ht {d:.4f}", flush=True)
