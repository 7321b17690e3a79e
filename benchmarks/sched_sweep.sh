#!/bin/bash
# Sweep of the batch verifier's host scheduling policy (BP_SCHED_BIAS, rangeproof_host.hip) on one GPU.
set -e
for b in -1 0 1 2 3 100; do
  BP_SCHED_BIAS=$b timeout -k 10 120 python bench.py --cpu-sample 0 --msm-logn 0 --steps 10 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bias', $b, 'ms_per_step %.2f' % d['ms_per_step'], 'verifies/s %.0f' % d['value'], flush=True)"
done
