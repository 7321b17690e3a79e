#!/bin/bash
# kernel trace of synchronous 65,536-proof passes with the transcript on the device
set -e
TAG=${1:-r06}
R=$(pwd)
OUT=$R/gpurun_out/prof_fs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BP_FS_DEVICE=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 $R/benchmarks/pipelined_probe.py sync 6 65536 16 0 > $OUT/probe.txt 2> $OUT/trace.err
cd $R
python benchmarks/prof_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/kernel_trace_stats.txt
python benchmarks/prof_window.py $(find $OUT/trace -name "*.db" | head -1) 4 > $OUT/steady_window.txt || true
find $OUT -name "*.db" -delete
cat $OUT/probe.txt | grep -v amdgpu
head -30 $OUT/kernel_trace_stats.txt
cat $OUT/steady_window.txt
