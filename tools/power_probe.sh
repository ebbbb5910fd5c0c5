#!/bin/bash
# GPU box: socket power / sclk samples (rocm-smi) while a command runs.   bash tools/power_probe.sh tag <cmd...>
TAG=$1; shift
OUT=gpurun_out/power_$TAG.txt
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Power|sclk|Average Graphics" | tr '\n' ' '; echo; sleep 0.25; done ) > $OUT &
SAMP=$!
"$@" > gpurun_out/power_${TAG}_cmd.log 2>&1
kill $SAMP
awk '{print}' $OUT | sed -n '1,400p' | cut -c1-200 | sort | uniq -c | sort -rn | head -25
