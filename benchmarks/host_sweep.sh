#!/bin/bash
# host transcript, synchronous 65,536-proof passes: initial text on the GPU or the host, chunk count, scheduling
run() { timeout -k 10 120 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
for rep in 1 2; do
echo "BP_INIT_TEXT=1"; run sync 12 65536 16 4 || exit 1
echo "BP_INIT_TEXT=0"; BP_INIT_TEXT=0 run sync 12 65536 16 4 || exit 1
done
for c in 3 5 6 8 2; do run sync 12 65536 16 $c || exit 1; done
for b in 0 1 2; do echo "BP_SCHED_BIAS=$b"; BP_SCHED_BIAS=$b run sync 12 65536 16 4 || exit 1; done
echo "pipe2 x 2 chunks"; run pipe2 16 65536 16 2 || exit 1
echo "threads 24"; run sync 12 65536 24 4 || exit 1
echo "threads 8"; run sync 12 65536 8 4 || exit 1
