"""A/B timing of the guided generator forward inside ONE process (box-to-box and run-to-run spread is +-1.5 %, more than most kernel changes move the
whole forward): alternates an environment switch that the library reads per call and reports the mean forward time of each setting.
usage: python tools/forward_ab.py MASKBIT_AMD_ATT_PAIR 2 4 [rounds] [forwards per round]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    var, vals = sys.argv[1], sys.argv[2:4]
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    nfw = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    from maskbit_amd import parity_replay as PR
    gen, _ = PR.build_models(torch.device("cuda"), with_tokenizer=False)
    B = 64
    tok = torch.randint(0, 64, (B, 256, 2), device="cuda")
    y = torch.randint(0, 1000, (B,), device="cuda")
    for v in vals:
        os.environ[var] = v
        for _ in range(3): gen.forward_cfg(tok, y)
    torch.cuda.synchronize()
    acc = {v: [] for v in vals}
    for r in range(rounds):
        for v in (vals if r % 2 == 0 else vals[::-1]):
            os.environ[var] = v
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(nfw): gen.forward_cfg(tok, y)
            torch.cuda.synchronize()
            acc[v].append((time.perf_counter() - t0) / nfw * 1e3)
    for v in vals:
        a = acc[v]
        print(f"{var}={v}: {sum(a) / len(a):8.3f} ms per guided forward (B = {B} pairs)   rounds: {' '.join(f'{x:.2f}' for x in a)}")


if __name__ == "__main__":
    main()
