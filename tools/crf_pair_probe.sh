#!/bin/bash
# GPU box: per-kernel split of the paired post-processing (tools/crf_pair_probe.py) from a rocprofv3 kernel trace.
#   bash tools/crf_pair_probe.sh [noise=4] [tag=probe]      -> gpurun_out/crf_pair_<tag>.txt
export TMPDIR=/tmp
R=$PWD
NOISE=${1:-4}
TAG=${2:-probe}
mkdir -p $R/gpurun_out
cd /tmp
rm -rf /tmp/cpp_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/cpp_$TAG -o prof --output-format csv -- python3 $R/tools/crf_pair_probe.py 3 $NOISE > /tmp/cpp_$TAG.log 2>&1
OUT=$R/gpurun_out/crf_pair_$TAG.txt
grep "^{" /tmp/cpp_$TAG.log | tail -1 > $OUT
f=$(find /tmp/cpp_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> $OUT <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    t = float(r["TotalDurationNs"]); c = int(r["Calls"]); tot += t
    if t > 0.2e6:
        print(f"  {n[:90]:90s} calls {c:6d} total {t/1e6:8.2f} ms avg {t/c/1e3:8.1f} us")
print(f"  all kernels {tot/1e6:.1f} ms over 4 calls (1 warm-up + 3)")
PY
cat $OUT
