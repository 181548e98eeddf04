"""Sum the rocprofv3 --pmc counters over every decoder kernel of tools/decode_one.py and divide by the number of decode calls.
usage: python tools/pmc_decoder_summary.py <pmc dir with dec.FETCH_SIZE / dec.WRITE_SIZE> <B> <iters> <out.md>"""
import csv, glob, os, sys
root, B, iters, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
DEC = ("conv_kernel", "gn_partial", "gn_finalize", "latent_kernel")
tot, per = {}, {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, "dec." + ctr, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr or not any(k in row["Kernel_Name"] for k in DEC):
                continue
            name = row["Kernel_Name"].split("(")[0][-48:]
            tot[ctr] = tot.get(ctr, 0.0) + float(row["Counter_Value"])
            per.setdefault(name, {}).setdefault(ctr, 0.0)
            per[name][ctr] += float(row["Counter_Value"])
# KiB per counter unit; FETCH_SIZE under-reports wide coalesced streams by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)
fetch = 2 * tot.get("FETCH_SIZE", 0.0) * 1024 / iters
write = tot.get("WRITE_SIZE", 0.0) * 1024 / iters
ideal = B * 426e6
lines = [f"decode of {B} images, {iters} calls averaged: corrected HBM-side bytes per call = 2 x FETCH_SIZE + WRITE_SIZE = "
         f"{fetch / 1e9:.3f} GB read + {write / 1e9:.3f} GB written = {(fetch + write) / 1e9:.3f} GB; ideal-fusion traffic (SURVEY 8d: 426 MB / image) "
         f"{ideal / 1e9:.3f} GB; ratio {(fetch + write) / ideal:.2f}", "", "| kernel | 2 x FETCH_SIZE MB / call | WRITE_SIZE MB / call |", "|---|---:|---:|"]
for name, r in sorted(per.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
    lines.append(f"| `{name}` | {2 * r.get('FETCH_SIZE', 0) * 1024 / iters / 1e6:.1f} | {r.get('WRITE_SIZE', 0) * 1024 / iters / 1e6:.1f} |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
