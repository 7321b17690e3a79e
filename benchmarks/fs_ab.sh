#!/bin/bash
# device transcript against host transcript, synchronous passes and two jobs in flight, by chunk count
run() { timeout -k 10 120 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
echo "side stream on"; run sync 12 65536 16 1 || exit 1
echo "BP_FS_SIDE=0"; BP_FS_SIDE=0 run sync 12 65536 16 1 || exit 1
echo "side on, 2 chunks"; run sync 12 65536 16 2 || exit 1
echo "pipe2 x 1 chunk"; run pipe2 16 65536 16 1 || exit 1
echo "pipe2 x 1 chunk BP_FS_SIDE=0"; BP_FS_SIDE=0 run pipe2 16 65536 16 1 || exit 1
echo "pipe3 x 1 chunk"; run pipe3 18 65536 16 1 || exit 1
echo "host transcript"; BP_FS_DEVICE=0 run sync 12 65536 16 4 || exit 1
echo "sizes"; for b in 1024 4096 16384 32768 131072 262144; do run sync 8 $b 16 0; done
echo "sizes host"; for b in 1024 4096 16384 32768; do BP_FS_DEVICE=0 run sync 8 $b 16 0; done
