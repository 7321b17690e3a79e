#!/bin/bash
# device transcript: switches, synchronous passes and two jobs in flight
run() { timeout -k 10 120 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
for rep in 1 2; do
echo "default"; run sync 12 65536 16 0 || exit 1
echo "BP_ROUND_BATCHINV=1"; BP_ROUND_BATCHINV=1 run sync 12 65536 16 0 || exit 1
echo "pipe2 default"; run pipe2 16 65536 16 0 || exit 1
echo "pipe2 BP_ROUND_BATCHINV=1"; BP_ROUND_BATCHINV=1 run pipe2 16 65536 16 0 || exit 1
done
