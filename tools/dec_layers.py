"""Per-dispatch listing of ONE decode call from a rocprofv3 --kernel-trace --output-format csv run of tools/decode_one.py.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python tools/decode_one.py 64 2; python tools/dec_layers.py <dir>"""
import csv, glob, os, sys
d = sys.argv[1]
f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode call = everything after the last latent_kernel
last = max(i for i, r in enumerate(rows) if "latent_kernel" in r["Kernel_Name"])
sel = rows[last:]
t0 = int(sel[0]["Start_Timestamp"])
tot = 0
agg = {}
for r in sel:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("void mb::", "").split("(")[0]
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    tot += dur
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {dur:8.1f} us  wgs {grid:6d} x {r['Workgroup_Size_X']:>4s}  {name[:70]}")
    agg[name] = agg.get(name, 0) + dur
print(f"sum of kernel durations {tot:.1f} us; span {(int(sel[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"  {v:9.1f} us  {k[:90]}")
