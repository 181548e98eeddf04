"""Average the rocprofv3 --pmc counters of the trunk-GEMM kernel launches (csv output of tools/profile_round.sh, or of bench.py's own counter
sub-passes: bench.py imports gemm_counters / algorithmic_bytes / hbm_bytes from here).
usage: python tools/pmc_summary.py <pmc dir> <out.json>"""
import csv, glob, json, os, sys

M_PAIR64 = 128 * 257            # rows of a guided forward over 64 sequence pairs
SHAPES = {"qkv": (3072, 1024), "attn_out": (1024, 1024), "ffn_up": (4096, 1024), "ffn_down": (1024, 4096)}      # (N, K)


def algorithmic_bytes(shape: str, mini_sets: int = 0, M: int = M_PAIR64) -> int:
    """Every operand and output byte of one launch exactly once: fp16 activations and weights, fp16 outputs (QKV, FFN-up) or the fp32 residual
    stream read + written (out-proj, FFN-down); + per mini-tile operand set (PAIR_ONE_MINI): the e2m1 weight operand (N K / 2 bytes + scales), the
    e2m1 token operand of the conditional rows (M / 2 rows x K / 2 bytes + one scale byte per 64 columns); FFN-up with mini-tiles also writes the
    e2m1 copy of its conditional outputs."""
    n, k = SHAPES[shape]
    b = M * k * 2 + n * k * 2 + (M * n * 2 if shape in ("qkv", "ffn_up") else 2 * M * n * 4)
    b += mini_sets * (n * k // 2 + n * (k // 128) + (M // 2) * k // 2 + (M // 2) * (k // 64))
    if mini_sets and shape == "ffn_up":
        b += (M // 2) * n // 2 + (M // 2) * (n // 64)
    return b


def hbm_bytes(fetch_size_kib: float, write_size_kib: float) -> float:
    """gfx950: FETCH_SIZE reports half of the bytes of wide coalesced streams (MI355X_MICROARCH.md, HBM section: double it); both counters are in KiB."""
    return (2 * fetch_size_kib + write_size_kib) * 1024


def gemm_counters(run_dir: str) -> dict:
    """{counter: (mean over the trunk-GEMM launches, launches)} of one rocprofv3 --pmc output directory."""
    acc = {}
    for f in glob.glob(os.path.join(run_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_ht_kernel" not in row["Kernel_Name"] and "gemm_tn_kernel" not in row["Kernel_Name"]:
                continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = {}
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        shape, ctrs = os.path.basename(d).split(".", 1)
        for k, (mean, n) in gemm_counters(d).items():
            res.setdefault(shape, {})[k] = mean
            res[shape][k + "_launches"] = n
    mini = int(os.environ.get("PAIR_ONE_MINI", "0") or 0)
    for s, r in res.items():
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r and s in SHAPES:
            r["hbm_bytes_corrected"] = hbm_bytes(r["FETCH_SIZE"], r["WRITE_SIZE"])
            r["algorithmic_bytes"] = algorithmic_bytes(s, mini)
            r["ratio"] = r["hbm_bytes_corrected"] / r["algorithmic_bytes"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
