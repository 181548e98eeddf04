"""Follow-up of tools/cumask_probe.py: K ordinary streams, each carrying 64/K pairs through the trunk-layer kernel sequence with persistent grids of
256/K workgroups (mb_set_cu_count), the sequences rotated against each other so that HBM-bound and matrix-bound kernels of different streams
share the chip.  No CU masks: the dispatcher spreads the co-running grids over all CUs (one 512-thread GEMM workgroup fits per CU).
usage: python tools/multistream_probe.py [layers]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from maskbit_amd import _lib
import cumask_probe as CP

lib = _lib.load()


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    sync = torch.cuda.synchronize
    cur = torch.cuda.current_stream().cuda_stream
    full = CP.Layer(64, 1)
    lib.mb_set_cu_count(0)
    ops_full = full.ops(cur)
    CP.run_seq(ops_full, 2); sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); CP.run_seq(ops_full, layers); sync()
        ts.append((time.perf_counter() - t0) / layers * 1e6)
    t_full = min(ts)
    print(f"full: {t_full:8.1f} us per layer (64 pairs, one stream)   runs {[round(t, 1) for t in ts]}", flush=True)
    del full, ops_full
    torch.cuda.empty_cache()
    n = 7
    for K, grids, rots in ((2, (128, 128), [(0, r) for r in range(1, 7)]),
                           (2, (256, 256), [(0, 3), (0, 4)]),
                           (2, (160, 160), [(0, 3), (0, 4)]),
                           (2, (192, 128), [(0, 4)]),
                           (4, (64, 64, 64, 64), [(0, 2, 4, 6), (0, 1, 3, 5), (0, 4, 2, 6), (0, 3, 4, 6)]),
                           (4, (128, 128, 128, 128), [(0, 2, 4, 6)]),
                           (3, (96, 96, 96), [(0, 2, 4), (0, 3, 5)])):
        per = [64 // K + (1 if i < 64 % K else 0) for i in range(K)]
        lays = [CP.Layer(p, 10 + i) for i, p in enumerate(per)]
        streams = [torch.cuda.Stream() for _ in range(K)]
        ops = [l.ops(s.cuda_stream) for l, s in zip(lays, streams)]
        for rot in rots:
            def go(nl):
                for i in range(nl * n):
                    for k in range(K):
                        lib.mb_set_cu_count(grids[k])
                        ops[k][(i + rot[k]) % n][1]()
            go(1); sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); go(layers); sync()
                best = min(best, (time.perf_counter() - t0) / layers * 1e6)
            print(f"K = {K} streams x {per} pairs, grids {grids}, rot {rot}: {best:8.1f} us per layer  ratio {best / t_full:.3f}", flush=True)
        lib.mb_set_cu_count(0)
        del lays, ops
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
