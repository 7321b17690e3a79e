#!/bin/bash
# A/B runs of the batch verifier's switches on one GPU: BP_ROUND_BATCHINV x BP_TEXT0, two runs each.
for rb in 0 1; do for t0 in 0 1; do for rep in 1 2; do
  BP_ROUND_BATCHINV=$rb BP_TEXT0=$t0 timeout -k 10 120 python bench.py --cpu-sample 0 --msm-logn 0 --steps 10 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round_batchinv', $rb, 'text0', $t0, 'ms_per_step %.2f' % d['ms_per_step'], flush=True)" || exit 1
done; done; done
