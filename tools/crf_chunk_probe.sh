#!/bin/bash
# GPU box: per-kernel CRF time when 1 / 2 / all images are in flight (does Infinity-Cache residency of one image pay?)
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in 1 2 0; do
  rocprofv3 --kernel-trace --stats -d /tmp/cp$c -o prof --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-noise12 --crf-chunk $c > /tmp/cp$c.log 2>&1
  echo "== chunk $c: $(tail -1 /tmp/cp$c.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["crf"])')"
  f=$(find /tmp/cp$c -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = r["Name"]
    if "crf_" in n:
        t = float(r["TotalDurationNs"]); c = int(r["Calls"]); tot += t
        print(f"  {n[:70]:70s} calls {c:6d} total {t/1e6:8.2f} ms avg {t/c/1e3:8.1f} us")
print(f"  crf total {tot/1e6:.1f} ms over 3 steps")
PY
done
