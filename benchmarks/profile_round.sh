#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel trace + HBM traffic (FETCH_SIZE, WRITE_SIZE in separate passes)
# + SQ issue counters of the same bench command.  Run on the GPU box from the repo root:  bash benchmarks/profile_round.sh r01d
set -e
TAG=${1:-r01}
EXTRA=${2:-}   # e.g. "--transcript device"
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --msm-logn 20 --msm-min-logn 20 --extras 0 $EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.err
cd $R
python benchmarks/prof_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/kernel_trace_stats.txt
python benchmarks/prof_overlap.py $(find $OUT/trace -name "*.db" | head -1) | head -1 > $OUT/overlap.txt
python benchmarks/prof_window.py $(find $OUT/trace -name "*.db" | head -1) 1 > $OUT/steady_window.txt || true
python benchmarks/pmc_summary.py $(find $OUT/fetch -name "*.db" | head -1) > $OUT/pmc_fetch.txt
python benchmarks/pmc_summary.py $(find $OUT/write -name "*.db" | head -1) > $OUT/pmc_write.txt
python benchmarks/pmc_summary.py $(find $OUT/sq -name "*.db" | head -1) > $OUT/pmc_sq.txt
find $OUT -name "*.db" -delete
head -25 $OUT/kernel_trace_stats.txt
cat $OUT/overlap.txt
