#!/bin/bash
# two lanes per proof in the rounds (BP_ROUND_PAIR=1) against one, both transcript forms
run() { timeout -k 10 120 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
for rep in 1 2; do
echo "host default"; run sync 12 65536 16 4 || exit 1
echo "host BP_ROUND_PAIR=1"; BP_ROUND_PAIR=1 run sync 12 65536 16 4 || exit 1
done
echo "device default"; BP_FS_DEVICE=1 run sync 12 65536 16 0 || exit 1
echo "device BP_ROUND_PAIR=1"; BP_FS_DEVICE=1 BP_ROUND_PAIR=1 run sync 12 65536 16 0 || exit 1
echo "device 16384 default"; BP_FS_DEVICE=1 run sync 12 16384 16 0 || exit 1
echo "device 16384 BP_ROUND_PAIR=1"; BP_FS_DEVICE=1 BP_ROUND_PAIR=1 run sync 12 16384 16 0 || exit 1
echo "device 32768 BP_ROUND_PAIR=1"; BP_FS_DEVICE=1 BP_ROUND_PAIR=1 run sync 12 32768 16 0 || exit 1
echo "host 16384 default"; run sync 12 16384 16 0 || exit 1
echo "host 16384 BP_ROUND_PAIR=1"; BP_ROUND_PAIR=1 run sync 12 16384 16 0 || exit 1
