"""Experiment (round 6): the pair attention of the guided forward (64 sequence pairs x 16 heads, N = 257) through the LDS-resident kernel (the product:
one workgroup per (pair, head), K / V of the head in 72 KiB of LDS, two workgroups of four waves per CU, 245-251 VGPRs) against the STREAMING kernel of the
1024 + 1-token models forced onto the same shape (MASKBIT_AMD_ATT_STREAM=1: 64 queries per workgroup, K / V in 128-key blocks, 144 VGPRs, five chunks per
(pair, head), every chunk re-reading the head's K / V).  Each variant runs in its own process (the switch is read once), alternated.
usage: python tools/att_stream_ab.py [pairs]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from maskbit_amd import _lib
    lib = _lib.load()
    P = int(sys.argv[2])
    N, d, heads = 257, 1024, 16
    torch.manual_seed(0)
    qc = torch.randn(P * N, 3 * d, device="cuda") * 0.7
    qkv = torch.cat([qc, qc + torch.randn_like(qc) * 0.02]).half().contiguous()
    out = torch.empty(2 * P * N, d, device="cuda", dtype=torch.float16)
    out4 = torch.zeros(2 * P * N, 2 * d, device="cuda", dtype=torch.uint8)
    out4s = torch.zeros(heads * P * 256 + 256, device="cuda", dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: lib.mb_attention_pair_f4(qkv.data_ptr(), out.data_ptr(), out4.data_ptr(), out4s.data_ptr(), None, None, P, N, d, heads, st)
    for _ in range(10): fn()
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 50 * 1e3)
    print("us per call:", " ".join(f"{t:.1f}" for t in ts), "| checksum", float(out.float().abs().sum()), flush=True)
    sys.exit(0)
P = sys.argv[1] if len(sys.argv) > 1 else "64"
for rnd in range(3):
    for tag, val in (("LDS-resident (product)", "0"), ("streaming, forced", "1")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", P], env=dict(os.environ, MASKBIT_AMD_ATT_STREAM=val), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(f"round {rnd} {tag:24s}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.returncode}", flush=True)
