"""What would the PLAIN forward measure with the activation-lo mini-tile set (e2m1 lo halves of the LayerNorm outputs, cfg_pair 3's second set)?
Probe without building it: the conditional rows of the guided (pair) forward carry exactly that arithmetic, so the unguided reference runs are
replayed through forward_cfg and sampled from its conditional logits alone.  usage: python tools/plain_via_pair_probe.py [run names ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskbit_amd import _lib, parity_replay as R


@torch.no_grad()
def replay(gen, g, noise):
    lib = _lib.load()
    dev = gen.device
    q, c = noise
    scale, temp, mask_len = R.plan_of(g)
    S, B = g["steps"].shape[0], g["steps"].shape[1]
    y = g["labels"].to(dev)
    bad = tot = 0
    for i in range(S):
        tin = R.tokens_in(g, i).to(dev).contiguous()
        lc = gen.forward_cfg(tin, y, 1.0)[:B].contiguous()
        tout, pred = torch.empty_like(tin), torch.empty_like(tin)
        _lib.check(lib.mb_sample_step(lc.data_ptr(), None, 0.0, temp[i], q[i].data_ptr(), c[i].data_ptr(), mask_len[i], tin.data_ptr(), tout.data_ptr(),
                                      pred.data_ptr(), B, g["steps"].shape[2], 2, g["C"], torch.cuda.current_stream().cuda_stream), "mb_sample_step")
        msk = g["masks"][i]
        bad += int((pred.cpu() != g["steps"][i])[msk].sum()); tot += int(msk.sum())
    return bad, tot


def main():
    names = sys.argv[1:] or [R.RUN_CFG1, R.RUN_CFG1_S2, R.RUN_CFG1_S3]
    pooled = {}
    for name in names:
        g = R.load_run(name)
        gen, _ = R.build_models("cuda", with_tokenizer=False, name=name)
        noise = R.reference_noise(g, gen.device)
        for tag, pair in (("conditional rows of the pair forward, cfg_pair 2 (weight correction)", 2), ("cfg_pair 3 (+ e2m1 lo halves of the LayerNorm outputs)", 3)):
            gen.act_split, gen.cfg_pair = 0, pair
            bad, tot = replay(gen, g, noise)
            print(f"{name:28s} {tag:72s} {bad:4d}/{tot} = {bad / tot:.2e}", flush=True)
            p = pooled.setdefault(tag, [0, 0]); p[0] += bad; p[1] += tot
        del gen; torch.cuda.empty_cache()
    for tag, (b, t) in pooled.items():
        print(f"== {tag}: {b}/{t} = {b / t:.2e}")


if __name__ == "__main__":
    main()
