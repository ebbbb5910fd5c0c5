#!/bin/bash
# GPU box: PMC counters of selected kernels of any python command.
#   bash tools/pmc_probe.sh "<counters>" "<kernel regex>" tag script.py [args...]     -> gpurun_out/pmc_<tag>.txt
export TMPDIR=/tmp
R=$PWD
CNT="$1"; RX="$2"; TAG="$3"; shift 3
cd /tmp; rm -rf /tmp/pmcp_$TAG
rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex "$RX" -d /tmp/pmcp_$TAG -o pmc --output-format csv -- python3 "$@" > /tmp/pmcp_$TAG.log 2>&1
f=$(find /tmp/pmcp_$TAG -name "*counter_collection.csv" | head -1)
python3 - "$f" > $R/gpurun_out/pmc_$TAG.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in acc[k].items():
        print(f"   {c:32s} {v / cnt[(k, c)]:16.1f} per dispatch ({cnt[(k, c)]} dispatches)")
PY
cat $R/gpurun_out/pmc_$TAG.txt
