#!/bin/bash
# Step time of the batch verifier against the batch size (4 chunks): linear growth = throughput-bound.
for b in 16384 32768 65536 131072 262144; do
  timeout -k 10 160 python bench.py --batch $b --chunks 4 --cpu-sample 0 --msm-logn 0 --extras 0 --steps 6 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', $b, 'ms_per_step %.2f' % d['ms_per_step'], 'M/s %.3f' % (d['value']/1e6), flush=True)" || exit 1
done
