#!/bin/bash
# host transcript: GPU-formatted initial text with the side work started per chunk, and chunk spacing
run() { timeout -k 10 120 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
for rep in 1 2; do
echo "default"; run sync 12 65536 16 4 || exit 1
echo "BP_INIT_TEXT=1"; BP_INIT_TEXT=1 run sync 12 65536 16 4 || exit 1
done
for s in 1200 1500 1800; do echo "BP_INIT_TEXT=1 BP_STAGGER_US=$s"; BP_INIT_TEXT=1 BP_STAGGER_US=$s run sync 12 65536 16 4 || exit 1; done
echo "BP_INIT_TEXT=1 pipe2x2"; BP_INIT_TEXT=1 run pipe2 16 65536 16 2 || exit 1
echo "BP_INIT_TEXT=1 threads 8"; BP_INIT_TEXT=1 run sync 12 65536 8 4 || exit 1
