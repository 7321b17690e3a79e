#!/bin/bash
# rocprofv3 kernel traces of the plain MSM (bp_msm_dev, scalars resident) at N = 2^LOGN for the three scalar shapes of
# SURVEY.md 8d config 3.  Run on the GPU box from the repo root:  bash benchmarks/profile_msm.sh r05 20
# Leaves gpurun_out/prof_msm_TAG/msm_2pLOGN_KIND_kernel_stats.txt (copy into profiles/ to track them).
set -e
TAG=${1:-r05}
LOGN=${2:-20}
MODE=${3:-plain}
R=$(pwd)
OUT=$R/gpurun_out/prof_msm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for KIND in uniform 64bit 01; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_$KIND -- python3 $R/benchmarks/msm_prof.py $LOGN $MODE $KIND > $OUT/$KIND.out 2> $OUT/$KIND.err
  python3 $R/benchmarks/prof_summary.py $(find $OUT/trace_$KIND -name "*.db" | head -1) > $OUT/msm_2p${LOGN}_${KIND}_kernel_stats.txt
  find $OUT/trace_$KIND -name "*.db" -delete
  echo "== $KIND"; head -24 $OUT/msm_2p${LOGN}_${KIND}_kernel_stats.txt
done
