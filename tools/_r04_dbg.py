import torch, hashlib, math
sha=lambda t: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()[:12]
g=torch.Generator().manual_seed(1_000_003 * 191)
v=torch.randn(3072,1024,generator=torch.Generator().manual_seed(7))*0.02
print("v", sha(v))
zs=[torch.randn(v.shape, generator=g) for _ in range(6)]
print("z", [sha(z) for z in zs])
sq=[z**2 for z in zs]; print("sq", sha(sq[0]))
chi=sum(sq); print("chi", sha(chi))
c6=chi/6.0; print("c6", sha(c6))
r=torch.sqrt(c6); print("sqrt", sha(r))
d=v/r; print("div", sha(d))
f=d*(1.0/math.sqrt(1.5)); print("final", sha(f))
idx=torch.randint(0,64,v.shape,generator=g); print("randint", sha(idx))
print("randperm", sha(torch.randperm(1024,generator=g)), "rand", sha(torch.rand(6,generator=g)))
print(torch.__version__, torch.get_num_threads(), torch.backends.cpu.get_cpu_capability())
