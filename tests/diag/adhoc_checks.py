"""Ad-hoc combinations not covered one by one in the pytest suite: fp16x2 weights with the pre-norm variant and with 1025-token
sequences; encoder at batch 9 (engine regrowth) and 512x512 input with the 5-resolution tokenizer."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import maskbit_oracle as O
from hip_helpers import hip_generator, hip_tokenizer
from maskbit_amd import LFQBert
DEV = "cuda"
def rel(a, b): return float((a.cpu() - b).norm() / b.norm())
# 1. pre-norm + fp16x2 weights, full width
cfg = O.GenCfg(bits=12, splits=2, depth=2, prenorm=True)
sd = O.make_generator_weights(cfg, seed=9, head_gain=12.0)
m = LFQBert(img_size=256, hidden_dim=1024, codebook_size=4096, codebook_splits=2, depth=2, heads=16, mlp_dim=4096, nclass=1000, use_prenorm=True)
m.load_state_dict(sd, strict=True); m = m.eval().to(DEV)
g = torch.Generator().manual_seed(0)
t = torch.randint(0, 65, (6, 256, 2), generator=g); y = torch.randint(0, 1000, (6,), generator=g); d = torch.rand(6, generator=g) < 0.5
ref = O.lfq_bert_forward(sd, cfg, t, y, d)
e0 = rel(m(t.to(DEV), y.to(DEV), d.to(DEV)), ref); m.weight_split = 1; e1 = rel(m(t.to(DEV), y.to(DEV), d.to(DEV)), ref)
print(f"pre-norm full width: rel err fp16 {e0:.2e}, fp16x2 {e1:.2e}"); assert e1 < e0 < 2e-3
# 2. 1025 tokens + fp16x2
cfg = O.GenCfg(bits=12, splits=2, hidden=1024, depth=1, heads=16, mlp=2048, seq=1024, nclass=10)
sd = O.make_generator_weights(cfg, seed=10, head_gain=12.0)
m = LFQBert(img_size=512, hidden_dim=1024, codebook_size=4096, codebook_splits=2, depth=1, heads=16, mlp_dim=2048, nclass=10)
m.load_state_dict(sd, strict=True); m = m.eval().to(DEV); m.weight_split = 1
t = torch.randint(0, 65, (2, 1024, 2), generator=g); y = torch.tensor([1, 2]); d = torch.tensor([False, True])
print(f"1025 tokens, fp16x2: rel err {rel(m(t.to(DEV), y.to(DEV), d.to(DEV)), O.lfq_bert_forward(sd, cfg, t, y, d)):.2e}")
# 3. encoder: batch 9 and a 512x512 input
tcfg = O.TokCfg(token_size=12)
tsd = O.make_tokenizer_weights(tcfg, seed=200, with_encoder=True)
tk = hip_tokenizer(tcfg, tsd)
x = torch.rand(9, 3, 256, 256, generator=g)
_, r9 = tk.encode(x.to(DEV)); _, r1 = tk.encode(x[4:5].to(DEV))
assert torch.equal(r9["min_encoding_indices"][4:5], r1["min_encoding_indices"])
_, ri = O.encode_image(tsd, tcfg, x[:1])
print("encoder batch 9 ok; index mismatch vs oracle (image 0):", float((r9["min_encoding_indices"][0].cpu() != ri[0]).float().mean()))
x5 = torch.rand(1, 3, 512, 512, generator=g)
zq, r5 = tk.encode(x5.to(DEV)); rec, _ = tk(x5.to(DEV))
_, ri5 = O.encode_image(tsd, tcfg, x5)
print("512x512 encode:", tuple(r5["min_encoding_indices"].shape), "index mismatch vs oracle", float((r5["min_encoding_indices"].cpu() != ri5).float().mean()), "recon", tuple(rec.shape))
print("all ok")
