#!/bin/bash
# device transcript: synchronous passes and two jobs in flight, a few sizes
run() { timeout -k 10 120 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
for rep in 1 2; do
run sync 12 65536 16 0 || exit 1
run pipe2 16 65536 16 0 || exit 1
done
for b in 1024 4096 16384 32768 262144; do run sync 8 $b 16 0 || exit 1; done
