"""Run the conv-VQGAN decoder on B random 12-bit token maps a few times (target for rocprofv3 --pmc / --kernel-trace passes).
usage: python tools/decode_one.py [B=64] [iters=3]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskbit_amd import ConvVQModel, synth


class Cfg(dict):
    __getattr__ = dict.__getitem__


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tok = ConvVQModel(Cfg(quantizer_type="lookup-free", codebook_size=4096, token_size=12, num_channels=3, hidden_channels=128,
                      channel_mult=[1, 1, 2, 2, 4], num_resolutions=5, num_res_blocks=2, sample_with_conv=True))
tok.load_state_dict(synth.make_tokenizer_weights(synth.TokCfg(token_size=12), seed=200), strict=False)
tok = tok.eval().requires_grad_(False).to("cuda")
t = torch.randint(0, 4096, (B, 256), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
for _ in range(iters):
    tok.decode_tokens_uint8(t)
torch.cuda.synchronize()
print("done", B, iters, "saturated groups:", tok.saturation_count())
