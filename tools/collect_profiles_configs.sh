#!/bin/bash
# Run on the GPU box (gpurun), second half of the round's profile collection: BASELINE configs 3-5 at their single-GPU shapes
# (psc59, coco80, ade768) in the headline mode -- rocprofv3 kernel-trace summaries and the FETCH_SIZE / WRITE_SIZE passes of the
# DenseCRF kernels (K = 59 / 81 / 2 x 150 channels per row).  tools/summarize_profiles.py <tag> turns them into
# profiles/<tag>_{psc59,coco80,ade768}_kernel_stats_summary.txt and profiles/<tag>_{...}_crf_traffic.json.
#   bash tools/collect_profiles_configs.sh r05 [configs...]
set -u
TAG=${1:-r06}
shift
CFGS=${@:-psc59 coco80 ade768}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
BARGS="--steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
B1="--steps 1 --warmup 1 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
for cfg in $CFGS; do
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace_$cfg -o prof --output-format csv -- python3 $R/bench.py --config $cfg $BARGS > $OUT/${TAG}_trace_$cfg.log 2>&1
  echo "trace $cfg rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "crf_" -d $OUT/${TAG}_pmc_${cfg}_$c -o pmc --output-format csv \
      -- python3 $R/bench.py --config $cfg $B1 > $OUT/${TAG}_pmc_${cfg}_$c.log 2>&1
    echo "pmc $cfg $c rc=$?"
  done
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "crf_" -d $OUT/${TAG}_pmc_${cfg}_tcc -o pmc --output-format csv \
    -- python3 $R/bench.py --config $cfg $B1 > $OUT/${TAG}_pmc_${cfg}_tcc.log 2>&1
  echo "pmc $cfg tcc rc=$?"
done
ls $OUT | grep "^${TAG}_" | head -60
