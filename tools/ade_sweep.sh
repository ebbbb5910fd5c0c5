R=$PWD; OUT=$R/gpurun_out/ade_sweep.txt; : > $OUT
A="--config ade768 --steps 4 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
line() { tag=$1; shift; python3 $R/bench.py "$@" 2> /tmp/bs_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipelines']
print('$tag', 'images/s %.2f' % d['value'], 'ms/step %.1f' % d['ms_per_step'], 'one-at-a-time %.2f images/s %.1f ms/step' % (p['one_batch_at_a_time']['value'], p['one_batch_at_a_time']['ms_per_step']), 'gemm frac %.4f' % d['roofline']['frac'], 'crf ms/step %.1f' % d['crf']['ms_per_step'])
" >> $OUT || { echo "$tag FAILED" >> $OUT; tail -3 /tmp/bs_err.txt >> $OUT; }; }
line "ade768 B=8 chunk=2" $A
line "ade768 B=8 chunk=1" $A --crf-chunk 1
line "ade768 B=7 chunk=1" $A --batch 7 --crf-chunk 1
line "ade768 B=7 chunk=2" $A --batch 7 --crf-chunk 2
line "ade768 B=7 chunk=1 P=3" $A --batch 7 --crf-chunk 1 --pipelines 3
line "ade768 B=14 chunk=1" $A --batch 14 --crf-chunk 1
cat $OUT
