set -e
ARGS="--steps 10 --warmup 2 --cpu-sample 0 --msm-logn 20 --msm-min-logn 20 --extras 0"
for v in -1 1 -1 1; do
  if [ $v = -1 ]; then unset BP_ROUND_PAIR; else export BP_ROUND_PAIR=$v; fi
  timeout -k 10 200 python bench.py $ARGS | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('BP_ROUND_PAIR=$v', round(j['ms_per_step'],3))"
done
