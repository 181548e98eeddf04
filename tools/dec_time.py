"""Decode time of B images (uint8 path), mean of n calls.  MASKBIT_AMD_CONV_TH=8 / 16 forces the conv tile height (read once per process).
usage: [DEC_TIME_LIB=other.so] python tools/dec_time.py [B=64] [n=10]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskbit_amd import ConvVQModel, synth, _lib
if os.environ.get("DEC_TIME_LIB"):      # A/B against another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["DEC_TIME_LIB"])


class Cfg(dict):
    __getattr__ = dict.__getitem__


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tok = ConvVQModel(Cfg(quantizer_type="lookup-free", codebook_size=4096, token_size=12, num_channels=3, hidden_channels=128,
                      channel_mult=[1, 1, 2, 2, 4], num_resolutions=5, num_res_blocks=2, sample_with_conv=True))
tok.load_state_dict(synth.make_tokenizer_weights(synth.TokCfg(token_size=12), seed=200), strict=False)
tok = tok.eval().requires_grad_(False).to("cuda")
t = torch.randint(0, 4096, (B, 256), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
for _ in range(3): u8 = tok.decode_tokens_uint8(t)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(n): u8 = tok.decode_tokens_uint8(t)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / n * 1e3)
print(f"decode of {B} images, conv tile height {os.environ.get('MASKBIT_AMD_CONV_TH', 'auto')}: {' / '.join(f'{x:.2f}' for x in ts)} ms; checksum {int((u8[1] if isinstance(u8, tuple) else u8).long().sum())}")
