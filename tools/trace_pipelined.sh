#!/bin/bash
# GPU box: kernel trace of the pipelined bench; idle share and the per-kernel sum of durations against the wall time.
export TMPDIR=/tmp
R=$PWD
cd /tmp; rm -rf /tmp/tp
mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace -d /tmp/tp -o tr --output-format csv -- python3 $R/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 > /tmp/tp.log 2>&1
f=$(find /tmp/tp -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/trace_pipelined.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the pipelined region: where six consecutive hist_kernel launches (1-drop + N-drop of three batches) come from three queues;
# two steps are cut off either end
hist = [(e, q) for s, e, n, q in rows if "hist_kernel" in n]
piped = [i for i in range(len(hist) - 5) if len(set(q for _, q in hist[i:i + 6])) >= 3]
if piped and piped[0] + 4 < len(hist):
    t_lo, t_hi = hist[piped[0] + 4][0], hist[piped[-1] + 1][0]
else:       # fewer than three queues overlap (--pipelines 1, or a short trace): the middle 60 % of the trace
    print("no pipelined region found (fewer than three queues overlap): using the middle 60 % of the trace")
    t0, t1 = rows[0][0], rows[-1][1]
    t_lo, t_hi = t0 + (t1 - t0) // 5, t1 - (t1 - t0) // 5
sel = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
if not sel:
    sys.exit("empty selection: trace too short")
span = (sel[-1][1] - sel[0][0]) / 1e6
tot = sum(e - s for s, e, _, _ in sel) / 1e6
busy_end = sel[0][0]; idle = 0
for s, e, _, _ in sel:
    if s > busy_end: idle += s - busy_end
    busy_end = max(busy_end, e)
gaps = []
busy_end = sel[0][0]; prev = sel[0]
for r in sel:
    s_, e_, n_, q_ = r
    if s_ > busy_end: gaps.append((s_ - busy_end, prev[2][:48], n_[:48]))
    if e_ > busy_end: busy_end, prev = e_, r
gaps.sort(reverse=True)
import collections
by = collections.Counter(); cnt = collections.Counter()
for g, a, b in gaps: by[(a, b)] += g; cnt[(a, b)] += 1
print("idle gaps: %d, > 100 us: %d (%.1f ms), 20-100 us: %d (%.1f ms), < 20 us: %d (%.1f ms)" % (len(gaps),
      sum(g > 1e5 for g, *_ in gaps), sum(g for g, *_ in gaps if g > 1e5) / 1e6,
      sum(2e4 < g <= 1e5 for g, *_ in gaps), sum(g for g, *_ in gaps if 2e4 < g <= 1e5) / 1e6,
      sum(g <= 2e4 for g, *_ in gaps), sum(g for g, *_ in gaps if g <= 2e4) / 1e6))
for (a, b), g in by.most_common(12):
    print(f"  {g / 1e6:7.2f} ms in {cnt[(a, b)]:5d} gaps  after [{a}]  before [{b}]")
print(f"window {span:.1f} ms, sum of kernel durations {tot:.1f} ms (x{tot/span:.2f}), idle {idle/1e6:.2f} ms ({100*idle/1e6/span:.1f} %), queues {len(set(q for *_, q in sel))}")
PY
cat $R/gpurun_out/trace_pipelined.txt; tail -1 /tmp/tp.log | cut -c1-200
