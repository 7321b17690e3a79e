#!/bin/bash
# host-transcript pass with unequal chunks (BP_CHUNK_WEIGHTS): a smaller first chunk starts the GPU sooner, a smaller
# last chunk shortens the tail that runs alone
# bash benchmarks/chunk_weights_ab.sh
set -e
ARGS="--steps 10 --warmup 2 --cpu-sample 0 --msm-logn 20 --msm-min-logn 20 --extras 0"
for w in "4,4,4,4" "5,4,4,3" "6,5,3,2" "3,4,4,5" "2,5,5,4" "3,5,5,3" "4,4,4,4" "5,5,4,2" "2,4,5,5"; do
  BP_CHUNK_WEIGHTS=$w timeout -k 10 200 python bench.py $ARGS | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('weights $w', round(j['ms_per_step'],3))"
done
