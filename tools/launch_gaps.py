"""How much of a small-batch sampling run is NOT kernel time: gaps between consecutive kernels on the stream (launch latency, drain / ramp), from a
rocprofv3 kernel trace.  usage (GPU box, from the repo root):
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python $ROOT/tools/launch_gaps.py run 16
    python tools/launch_gaps.py analyze OUT
`run B` samples BASELINE configs[1] (10-bit, 16 steps, no guidance) at batch B three times through the product's own path (no event pairs)."""
import csv, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(B):
    import torch
    from maskbit_amd import parity_replay as PR
    from config_bench import tokenizer
    from maskbit_amd.sampling import build_plan, run_chunked
    dev = torch.device("cuda")
    g = PR.load_run(PR.RUN_CFG1)
    gen, _ = PR.build_models(dev, with_tokenizer=False, name=PR.RUN_CFG1)
    tok = tokenizer(int(g["bits"]), dev)
    kw = g["kw"]
    plan = build_plan(int(kw["num_steps"]), 512, float(kw["guidance_scale"]), kw["guidance_annealing"], float(kw["scale_pow"]), 1.0, False, kw["mask_schedule_strategy"])
    labels = (torch.arange(B) * 37 % 1000).to(dev)
    rt = float(kw["randomize_temperature"])
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run_chunked(gen, tok, labels, plan, rt, want_steps=False, want_image=False, want_u8=True)
        torch.cuda.synchronize()
        print(f"batch {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)


def analyze(d):
    f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # batches: split at gaps > 1 ms (the host synchronises between batches); analyse the last one
    cuts = [0] + [i for i in range(1, len(rows)) if rows[i][0] - rows[i - 1][1] > 1_000_000] + [len(rows)]
    segs = [rows[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
    segs = [s for s in segs if len(s) > 500]
    for s in segs[-2:]:
        span = s[-1][1] - s[0][0]
        busy = sum(e - b for b, e, _ in s)
        gaps = [s[i + 1][0] - s[i][1] for i in range(len(s) - 1)]
        pos = [x for x in gaps if x > 0]
        pos.sort()
        print(f"{len(s)} kernels, span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms ({busy / span:.1%}), gaps {sum(pos) / 1e6:.2f} ms: median {pos[len(pos) // 2] / 1e3:.2f} us, "
              f"p90 {pos[int(len(pos) * .9)] / 1e3:.2f} us, max {pos[-1] / 1e3:.1f} us; overlapping pairs {sum(1 for x in gaps if x <= 0)}")
        big = sorted(((s[i + 1][0] - s[i][1], s[i][2][:60], s[i + 1][2][:60]) for i in range(len(s) - 1)), reverse=True)[:6]
        for gp, a, b in big:
            print(f"    {gp / 1e3:8.1f} us between {a} -> {b}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        analyze(sys.argv[2])
