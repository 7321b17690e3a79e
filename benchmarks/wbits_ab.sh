#!/bin/bash
# is the fixed-base kernel bound by its table gathers?  window widths 20 / 16 / 12 (13 / 16 / 22 additions per term, 60 / 8.7 / 0.75 GB of table)
run() { timeout -k 10 160 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
for w in 20 16 12; do
  echo "BP_FB_WBITS=$w device transcript"; BP_FS_DEVICE=1 BP_FB_WBITS=$w run sync 8 65536 16 0 || exit 1
  echo "BP_FB_WBITS=$w host transcript"; BP_FB_WBITS=$w run sync 8 65536 16 4 || exit 1
done
