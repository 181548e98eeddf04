"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (markdown/CSV-ish).
usage: python tools/rocprof_summary.py <results.db> [out.md]"""
import sqlite3, sys
db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
cols = [c[1] for c in cur.execute("pragma table_info('kernels')")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
for n, c, s, a, mn, mx in rows:
    n = n if len(n) < 90 else n[:87] + "..."
    lines.append(f"| `{n}` | {c} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.2f} |")
lines.append(f"\ntotal kernel time: {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
