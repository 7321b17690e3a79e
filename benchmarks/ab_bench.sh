#!/bin/bash
# same-box A/B of two builds: ab_prev/ (a copy of another commit's tree with its built libraries) against this tree.
# bash benchmarks/ab_bench.sh [rounds]
set -e
R=$(pwd)
N=${1:-3}
ARGS="--steps 12 --warmup 3 --cpu-sample 0 --msm-logn 20 --msm-min-logn 20 --extras 0"
for i in $(seq 1 $N); do
  (cd $R/ab_prev && timeout -k 10 200 python bench.py $ARGS) | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('A prev', j['ms_per_step'], j['roofline']['isolated']['avg_launch_ms'])"
  (cd $R && timeout -k 10 200 python bench.py $ARGS) | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('B this', j['ms_per_step'], j['roofline']['isolated']['avg_launch_ms'])"
done
