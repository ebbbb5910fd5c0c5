"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table.
usage: python tools/rocpd_stats.py results.db [out.txt]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
out = ["# kernel | calls | total ms | avg us | min us | max us | % of GPU time"]
for n, c, t, a, lo, hi in rows:
    out.append(f"{n[:150]} | {c} | {t:.2f} | {a:.1f} | {lo:.1f} | {hi:.1f} | {100 * t / tot:.1f}")
out.append(f"# total kernel time {tot:.2f} ms")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
