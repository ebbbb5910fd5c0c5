#!/bin/bash
# GPU box: DenseCRF update tile size (pixels per workgroup tile) against the mean-field time of a bench config (DEV library).
export PNP_HIP_LIB=$PWD/pnp-ovss_amd/pnp_ovss/libpnp_hip_dev.so
CFG=${1:-coco80}
mkdir -p gpurun_out
for tp in ${TPS:-256 128 64 32}; do
  PNP_CRF_TP=$tp python bench.py --config $CFG --no-other-modes --no-other-configs --no-cpu-baseline --no-noise12 --pipelines 1 --steps 3 --warmup 1 2> gpurun_out/tp_$tp.err > gpurun_out/tp_$tp.json || { echo "bench failed for tp $tp:"; tail -5 gpurun_out/tp_$tp.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/tp_$tp.json')); print('$CFG tp', $tp, 'img/s', round(d['value'],2), 'crf ms', round(d['crf']['ms_per_step'],1))"
done
