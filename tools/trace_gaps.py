"""Idle time between kernels from a rocprofv3 --kernel-trace csv: where the GPU waits for the host or for a launch.

    python tools/trace_gaps.py <kernel_trace.csv> [--min-us 2] [--top 25]

Kernels are sorted by start time; the gap before a kernel is its start minus the latest end seen so far (overlapping
kernels on other streams count as busy).  Gaps are summed by (previous kernel -> next kernel) and printed with the busy /
idle totals of the trace between the first and the last kernel.
"""
import argparse
import csv
import re
from collections import defaultdict


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("pnp::", "").replace("void ", "")
    return n[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--min-us", type=float, default=2.0)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--skip-ms", type=float, default=0.0, help="ignore the first part of the trace (set-up, warm-up)")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0 = rows[0][0] + int(a.skip_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    busy_end, prev = rows[0][0], "(start)"
    gaps = defaultdict(lambda: [0, 0.0])
    idle = small = 0.0
    for s, e, n in rows:
        if s > busy_end:
            g = (s - busy_end) / 1e3
            idle += g
            if g >= a.min_us:
                k = (short(prev), short(n))
                gaps[k][0] += 1
                gaps[k][1] += g
            else:
                small += g
        if e > busy_end:
            busy_end, prev = e, n
    span = (busy_end - rows[0][0]) / 1e3
    print(f"span {span / 1e3:.1f} ms, idle {idle / 1e3:.2f} ms ({100 * idle / span:.1f} %), of which gaps < {a.min_us} us: {small / 1e3:.2f} ms; {len(rows)} kernels")
    for (p, n), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"  {t / 1e3:8.3f} ms  {c:6d} x {t / c:8.1f} us   {p}  ->  {n}")


if __name__ == "__main__":
    main()
