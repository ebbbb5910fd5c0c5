#!/bin/bash
# GPU box: per-kernel time of kernels matching a pattern in one traced bench run per config.
#   bash tools/cfg_kstats.sh 'crf_update|crf_splat' psc59 coco80 ade768
PAT=$1; shift
Q="--steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
for c in ${@:-voc psc59 coco80 ade768}; do
  bash tools/kstats.sh cfg_$c bench.py --config $c $Q > /dev/null 2>&1
  grep -E "$PAT" gpurun_out/kstats_cfg_$c.txt | sed "s/^/$c /"
done
