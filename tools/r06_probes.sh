set -u
Q="--steps 4 --warmup 1 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
crf() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', 'images/s %.1f' % d['value'], 'ms/step %.1f' % d['ms_per_step'], 'crf ms/step %.2f' % d['crf']['ms_per_step'], 'frac %.3f' % d['crf']['frac'])
"; }
{
echo "# DenseCRF rows: paired (1-drop | N-drop in one row, 176 B at K = 21 / 480 B at K = 59) against two passes of half-width rows"
python bench.py $Q > gpurun_out/p_voc_pair.json 2>/dev/null; crf gpurun_out/p_voc_pair.json "voc paired"
python bench.py $Q --separate-crf > gpurun_out/p_voc_sep.json 2>/dev/null; crf gpurun_out/p_voc_sep.json "voc separate"
python bench.py $Q --config psc59 > gpurun_out/p_psc_pair.json 2>/dev/null; crf gpurun_out/p_psc_pair.json "psc59 paired"
python bench.py $Q --config psc59 --separate-crf > gpurun_out/p_psc_sep.json 2>/dev/null; crf gpurun_out/p_psc_sep.json "psc59 separate"
} > gpurun_out/r06_crf_rows.txt
python tools/gemm_text_probe.py 875 > gpurun_out/r06_text_probe.txt 2>/dev/null
python tools/gemm_text_probe.py 2975 >> gpurun_out/r06_text_probe.txt 2>/dev/null
export PNP_HIP_LIB=$PWD/pnp-ovss_amd/pnp_ovss/libpnp_hip_dev.so
for nz in 1 2 3; do PNP_TXT_NZ=$nz bash tools/kstats.sh coco_nz$nz bench.py --config coco80 --steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check > /dev/null 2>&1; grep -E "text_self_attn" gpurun_out/kstats_coco_nz$nz.txt | sed "s/^/nz=$nz /"; done > gpurun_out/r06_txt_nz.txt
unset PNP_HIP_LIB
cat gpurun_out/r06_crf_rows.txt gpurun_out/r06_text_probe.txt gpurun_out/r06_txt_nz.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.log 2>&1; tail -4 gpurun_out/r06_gpu_tests.log
