#!/bin/bash
# pass time by batch size, both transcript forms (library defaults for chunks and lanes per proof)
run() { timeout -k 10 160 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
for b in 1024 2048 4096 8192 16384 32768 65536 131072 262144; do
  echo -n "host   "; run sync 8 $b 16 0 || exit 1
  echo -n "device "; BP_FS_DEVICE=1 run sync 8 $b 16 0 || exit 1
done
