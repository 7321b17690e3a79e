#!/bin/bash
# k_fixed_multiexp at two waves per SIMD (default) against one (BP_FIXED_LDS=90000): is the gather latency-bound?
# bash benchmarks/fixed_occupancy_ab.sh
set -e
ARGS="--steps 8 --warmup 2 --cpu-sample 0 --msm-logn 20 --msm-min-logn 20 --extras 0"
for tr in device host; do
  for lds in 0 90000 0 90000; do
    BP_FIXED_LDS=$lds timeout -k 10 200 python bench.py $ARGS --transcript $tr | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tr lds=$lds', round(j['ms_per_step'],3), 'isolated', round(j['roofline']['isolated']['avg_launch_ms'],3), 'in pass', round(j['roofline']['avg_launch_ms'],3))"
  done
done
