"""The pair attention launch of the guided forward (64 sequence pairs x 16 heads, N = 257, dh = 64, with the e2m1 copy of the conditional outputs: what the engine issues
at precision >= 2) a few times: target of the rocprofv3 --pmc passes of tools/att_pmc.sh.  usage: python tools/att_only.py [pairs] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskbit_amd import _lib
lib = _lib.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, d, heads = 257, 1024, 16
torch.manual_seed(0)
qc = torch.randn(P * N, 3 * d, device="cuda") * 0.7
qkv = torch.cat([qc, qc + torch.randn_like(qc) * 0.02]).half().contiguous()
out = torch.empty(2 * P * N, d, device="cuda", dtype=torch.float16)
out4 = torch.zeros(2 * P * N, 2 * d, device="cuda", dtype=torch.uint8)
out4s = torch.zeros(heads * P * 256 + 256, device="cuda", dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
for _ in range(n):
    _lib.check(lib.mb_attention_pair_f4(qkv.data_ptr(), out.data_ptr(), out4.data_ptr(), out4s.data_ptr(), None, None, P, N, d, heads, st))
torch.cuda.synchronize()
print("done")
