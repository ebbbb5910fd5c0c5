#!/bin/bash
# GPU box: PMC counters of the CRF kernels on the paired probe.   bash tools/crf_pmc_probe.sh "<counters>" tag [noise]
export TMPDIR=/tmp
R=$PWD
CNT="$1"; TAG=${2:-pmc}; NOISE=${3:-4}
cd /tmp; rm -rf /tmp/cpmc_$TAG
rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex "crf_" -d /tmp/cpmc_$TAG -o pmc --output-format csv -- python3 $R/tools/crf_pair_probe.py 1 $NOISE > /tmp/cpmc_$TAG.log 2>&1
f=$(find /tmp/cpmc_$TAG -name "*counter_collection.csv" | head -1)
python3 - "$f" > $R/gpurun_out/crf_pmc_$TAG.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in acc[k].items():
        print(f"   {c:32s} {v / cnt[(k, c)]:16.1f} per dispatch ({cnt[(k, c)]} dispatches)")
PY
cat $R/gpurun_out/crf_pmc_$TAG.txt
