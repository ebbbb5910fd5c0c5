#!/bin/bash
# GPU box: kernel trace of the pipelined bench; idle share and the per-kernel sum of durations against the wall time.
export TMPDIR=/tmp
R=$PWD
cd /tmp; rm -rf /tmp/tp
rocprofv3 --kernel-trace -d /tmp/tp -o tr --output-format csv -- python3 $R/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-parity-mode --no-noise12 > /tmp/tp.log 2>&1
f=$(find /tmp/tp -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/trace_pipelined.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the pipelined region: the last 40 % of the trace by time
t_lo = rows[0][0] + int(0.62 * (rows[-1][1] - rows[0][0])); t_hi = rows[-1][1] - int(0.02 * (rows[-1][1] - rows[0][0]))
sel = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
span = (sel[-1][1] - sel[0][0]) / 1e6
tot = sum(e - s for s, e, _, _ in sel) / 1e6
busy_end = sel[0][0]; idle = 0
for s, e, _, _ in sel:
    if s > busy_end: idle += s - busy_end
    busy_end = max(busy_end, e)
print(f"window {span:.1f} ms, sum of kernel durations {tot:.1f} ms (x{tot/span:.2f}), idle {idle/1e6:.2f} ms ({100*idle/1e6/span:.1f} %), queues {len(set(q for *_, q in sel))}")
PY
cat $R/gpurun_out/trace_pipelined.txt; tail -1 /tmp/tp.log | cut -c1-200
