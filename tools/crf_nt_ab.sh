#!/bin/bash
# GPU box: A/B of the non-temporal streaming accesses in the DenseCRF splat / update kernels (round 5; they are the default:
# `base` builds with -DPNP_CRF_NO_NT, which turns ld_stream / st_stream in csrc/crf.hip into plain accesses).  gpurun_out/crf_nt_ab.txt
R=$PWD; OUT=$R/gpurun_out/crf_nt_ab.txt; : > $OUT
Q="--steps 6 --warmup 2 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
line() { tag=$1; shift; python3 $R/bench.py "$@" 2> /tmp/ab_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', 'images/s %.1f' % d['value'], 'ms/step %.1f' % d['ms_per_step'], 'crf ms/step %.2f' % d['crf']['ms_per_step'])
" >> $OUT || { echo "$tag FAILED" >> $OUT; tail -3 /tmp/ab_err.txt >> $OUT; }; }
for variant in base nt base2 nt2; do
  case $variant in
    base|base2) (cd pnp-ovss_amd/csrc && touch crf.hip && make -j8 EXTRA=-DPNP_CRF_NO_NT > /tmp/mk.log 2>&1) ;;
    nt|nt2) (cd pnp-ovss_amd/csrc && touch crf.hip && make -j8 > /tmp/mk.log 2>&1) ;;
  esac
  line "$variant voc" $Q
  line "$variant noise12" $Q --noise 12
  line "$variant psc59" $Q --config psc59
  line "$variant ade768" $Q --config ade768 --steps 3 --warmup 1
done
(cd pnp-ovss_amd/csrc && touch crf.hip && make -j8 > /tmp/mk.log 2>&1)
cat $OUT
