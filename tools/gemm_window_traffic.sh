#!/bin/bash
# GPU box: is the split-bf16 GEMM's fabric traffic above its algorithmic bytes HBM traffic or cache-served re-reads?
# The same launch (qkv shape of the bench: 15470 x 3072 x 1024, split-pair output) under different tile orders -- the group
# height GM of tile_coords (how many row tiles x how many column tiles the 32 workgroups of an XCD sweep together; GM = 0:
# plain row-major tile ids, i.e. no XCD grouping at all) -- with FETCH_SIZE / WRITE_SIZE from rocprofv3 and the launch time
# from back-to-back launches.  If the counter moves by 2x and the time does not, the bytes it counts are not what the launch
# waits for.  DEV library (PNP_GEMM_GM).    bash tools/gemm_window_traffic.sh [shape]   ->  gpurun_out/gemm_window_traffic.txt
export TMPDIR=/tmp
R=$PWD
SHAPE=${1:-qkv}
OUT=$R/gpurun_out
mkdir -p $OUT
: > $OUT/gemm_window_traffic.txt
cd /tmp
for gm in 0 1 2 4 8 16 61; do
  export PNP_GEMM_GM=$gm
  t=$(python3 $R/tools/gemm_x3_shapes.py --dev $SHAPE 2>/dev/null | grep "^$SHAPE" | awk '{print $5}')
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/gw_$c
    rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "gemm_nt_x3" -d /tmp/gw_$c -o pmc --output-format csv -- python3 $R/tools/gemm_x3_shapes.py --dev $SHAPE > /tmp/gw.log 2>&1
  done
  python3 - "$gm" "$t" >> $OUT/gemm_window_traffic.txt <<'PY'
import csv, glob, sys
gm, t = sys.argv[1], sys.argv[2]
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/gw_{c}/**/pmc_counter_collection.csv", recursive=True)[0]
    rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    v[c] = sum(rows) / len(rows)
fetch, write = 2 * v["FETCH_SIZE"] * 1024 / 1e6, v["WRITE_SIZE"] * 1024 / 1e6
print(f"GM {gm:>2}: {t:>7} us   fetch {fetch:7.1f} MB (FETCH_SIZE doubled)   write {write:7.1f} MB   total {fetch + write:7.1f} MB")
PY
done
unset PNP_GEMM_GM
cat $OUT/gemm_window_traffic.txt
