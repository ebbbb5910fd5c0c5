#!/bin/bash
# GPU box: time of the image-blur kernels (blur_axis_kernel<0|1>) in one traced bench run per config.  bash tools/blur_kstats.sh [configs...]
Q="--steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
for c in ${@:-voc psc59 coco80 ade768}; do
  bash tools/kstats.sh blur_$c bench.py --config $c $Q > /dev/null 2>&1
  grep -E "blur_axis" gpurun_out/kstats_blur_$c.txt | sed "s/^/$c /"
done
