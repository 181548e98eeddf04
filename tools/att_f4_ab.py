"""What the e2m1 copy of the conditional outputs costs the pair attention launch: mb_attention_pair against mb_attention_pair_f4, alternated in one
process (64 sequence pairs x 16 heads, N = 257).  usage: python tools/att_f4_ab.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskbit_amd import _lib

lib = _lib.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N, d, heads = 257, 1024, 16
torch.manual_seed(0)
qc = torch.randn(P * N, 3 * d, device="cuda") * 0.7
qkv = torch.cat([qc, qc + torch.randn_like(qc) * 0.02]).half().contiguous()
out = torch.empty(2 * P * N, d, device="cuda", dtype=torch.float16)
out4 = torch.zeros(2 * P * N, 2 * d, device="cuda", dtype=torch.uint8)
out4s = torch.zeros(heads * P * 256 + 256, device="cuda", dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
plain = lambda: lib.mb_attention_pair(qkv.data_ptr(), out.data_ptr(), P, N, d, heads, st)
f4 = lambda: lib.mb_attention_pair_f4(qkv.data_ptr(), out.data_ptr(), out4.data_ptr(), out4s.data_ptr(), None, None, P, N, d, heads, st)


def timed(fn, n=50):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for fn in (plain, f4): timed(fn, 10)
for rnd in range(4):
    print(f"round {rnd}: without the copy {timed(plain):.1f} us, with it {timed(f4):.1f} us", flush=True)
