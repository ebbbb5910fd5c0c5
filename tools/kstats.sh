#!/bin/bash
# GPU box: per-kernel stats of a command under rocprofv3.   bash tools/kstats.sh tag <python script + args>
export TMPDIR=/tmp
R=$PWD; TAG=$1; shift
SCRIPT=$1; shift
case "$SCRIPT" in /*) ;; *) SCRIPT=$R/$SCRIPT ;; esac      # rocprofv3 runs from /tmp: the script path must be absolute
set -- "$SCRIPT" "$@"
cd /tmp; rm -rf /tmp/ks_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/ks_$TAG -o prof --output-format csv -- python3 "$@" > /tmp/ks_$TAG.log 2>&1
f=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $R/gpurun_out/kstats_$TAG.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    t = float(r["TotalDurationNs"]); c = int(r["Calls"])
    print(f"  {r['Name'][:80]:80s} calls {c:6d} total {t/1e6:8.2f} ms avg {t/c/1e3:8.1f} us")
PY
cat $R/gpurun_out/kstats_$TAG.txt
