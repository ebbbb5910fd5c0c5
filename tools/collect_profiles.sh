#!/bin/bash
# Run on the GPU box (gpurun): the round's bench line, the rocprofv3 kernel-trace summaries of the same command (headline mode
# bf16x3, and the bf16 throughput mode) and the PMC passes the roofline / MFMA-busy / HBM-traffic figures come from.  Outputs
# under gpurun_out/ (tools/summarize_profiles.py turns them into the small committed files under profiles/).
#   bash tools/collect_profiles.sh r04
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$?"
cd /tmp
BARGS="--steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o prof --output-format csv -- python3 $R/bench.py $BARGS > $OUT/${TAG}_trace.log 2>&1
echo "trace (bf16x3 headline) rc=$?"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace_bf16 -o prof --output-format csv -- python3 $R/bench.py --dtype bf16 $BARGS > $OUT/${TAG}_trace_bf16.log 2>&1
echo "trace bf16 rc=$?"
# PMC passes (own runs, kernel-trace only): MFMA busy on the dense kernels, HBM-side bytes of the GEMM and CRF kernels
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "gemm_nt|vit_attn|xattn" \
  -d $OUT/${TAG}_pmc_mfma -o pmc --output-format csv -- python3 $R/bench.py $BARGS > $OUT/${TAG}_pmc_mfma.log 2>&1
echo "pmc mfma rc=$?"
B1="--steps 1 --warmup 1 --pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "gemm_nt_wide|gemm_nt_x3|crf_" -d $OUT/${TAG}_pmc_$c -o pmc --output-format csv \
    -- python3 $R/bench.py $B1 > $OUT/${TAG}_pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "gemm_nt_wide|gemm_nt_x3" -d $OUT/${TAG}_pmc_bf16_$c -o pmc --output-format csv \
    -- python3 $R/bench.py --dtype bf16 $B1 > $OUT/${TAG}_pmc_bf16_$c.log 2>&1
  echo "pmc bf16 $c rc=$?"
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "crf_" -d $OUT/${TAG}_pmc_tcc -o pmc --output-format csv \
  -- python3 $R/bench.py $B1 > $OUT/${TAG}_pmc_tcc.log 2>&1
echo "pmc tcc rc=$?"
ls $OUT | grep "^${TAG}_" | head -40
