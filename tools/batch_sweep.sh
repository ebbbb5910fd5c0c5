#!/bin/bash
# GPU box: (1) batch per engine against batches in flight (VERDICT r04 task 4) on the headline workload, (2) DenseCRF launch-group
# size on the ade768 config.  One compact line per run into gpurun_out/batch_sweep.txt.
#   bash tools/batch_sweep.sh
R=$PWD; OUT=$R/gpurun_out/batch_sweep.txt; : > $OUT
Q="--steps 12 --warmup 2 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
line() {  # tag, bench args...
  tag=$1; shift
  python3 $R/bench.py "$@" 2> /tmp/bs_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['pipelines']
print('$tag', 'images/s %.1f' % d['value'], 'ms/step %.1f' % d['ms_per_step'], 'one-at-a-time %.1f images/s %.1f ms/step' % (p['one_batch_at_a_time']['value'], p['one_batch_at_a_time']['ms_per_step']),
      'gemm frac %.4f' % d['roofline']['frac'], 'crf ms/step %.1f' % d['crf']['ms_per_step'], 'GiB per engine', [round(b/2**30,1) for b in p['engine_device_bytes']])
" >> $OUT || { echo "$tag FAILED" >> $OUT; tail -3 /tmp/bs_err.txt >> $OUT; }
}
line "voc B=35 P=3" $Q --pipelines 3
line "voc B=70 P=1" $Q --batch 70 --pipelines 1 --steps 6
line "voc B=70 P=2" $Q --batch 70 --pipelines 2 --steps 6
line "voc B=105 P=1" $Q --batch 105 --pipelines 1 --steps 4
line "voc B=105 P=2" $Q --batch 105 --pipelines 2 --steps 4
if [ "$1" = "all" ]; then
line "voc B=35 P=2" $Q --pipelines 2
line "voc B=35 P=1" $Q --pipelines 1
A="--config ade768 --steps 4 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
for c in 1 2 4 8; do line "ade768 crf_chunk=$c" $A --crf-chunk $c; done
fi
cat $OUT
