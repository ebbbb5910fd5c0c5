#!/bin/bash
# Run on the GPU box (gpurun): the round's bench line, the rocprofv3 kernel-trace summary of the same command and the PMC
# passes the roofline / MFMA-busy / HBM-traffic figures come from.  Outputs under gpurun_out/ (copied to profiles/ by hand).
#   bash tools/collect_profiles.sh r02
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
python3 bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$?"
cd /tmp
BARGS="--steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --no-parity-mode --no-noise12"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o prof --output-format csv -- python3 $R/bench.py $BARGS > $OUT/${TAG}_trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace_x3 -o prof --output-format csv -- python3 $R/bench.py --dtype bf16x3 $BARGS > $OUT/${TAG}_trace_x3.log 2>&1
echo "trace x3 rc=$?"
# PMC passes (own runs, kernel-trace only): MFMA busy on the dense kernels, HBM-side bytes of the GEMM and CRF kernels
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "gemm_nt|vit_attn|xattn" \
  -d $OUT/${TAG}_pmc_mfma -o pmc --output-format csv -- python3 $R/bench.py $BARGS > $OUT/${TAG}_pmc_mfma.log 2>&1
echo "pmc mfma rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "gemm_nt_wide|crf_" -d $OUT/${TAG}_pmc_$c -o pmc --output-format csv \
    -- python3 $R/bench.py --steps 1 --warmup 1 --pipelines 1 --no-cpu-baseline --no-parity-mode --no-noise12 > $OUT/${TAG}_pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "crf_" -d $OUT/${TAG}_pmc_tcc -o pmc --output-format csv \
  -- python3 $R/bench.py --steps 1 --warmup 1 --pipelines 1 --no-cpu-baseline --no-parity-mode --no-noise12 > $OUT/${TAG}_pmc_tcc.log 2>&1
echo "pmc tcc rc=$?"
ls -la $OUT | head -40
