#!/bin/bash
# how far apart should the four chunks of a host-transcript pass start?
run() { timeout -k 10 120 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
echo "default"; run sync 12 65536 16 4 || exit 1
for s in 2000 2500 3000 4000; do echo "BP_STAGGER_US=$s"; BP_STAGGER_US=$s run sync 12 65536 16 4 || exit 1; done
for s in 1000 1500 2000 3000; do echo "BP_INIT_TEXT=1 BP_STAGGER_US=$s"; BP_INIT_TEXT=1 BP_STAGGER_US=$s run sync 12 65536 16 4 || exit 1; done
for s in 3000 4000 6000; do echo "3 chunks BP_STAGGER_US=$s"; BP_STAGGER_US=$s run sync 12 65536 16 3 || exit 1; done
for s in 4000 6000 8000; do echo "2 chunks BP_STAGGER_US=$s"; BP_STAGGER_US=$s run sync 12 65536 16 2 || exit 1; done
