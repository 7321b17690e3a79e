#!/bin/bash
# J jobs in flight with C chunks per job and T hashing threads per job against the synchronous pass
set -e
run() { timeout -k 10 120 python benchmarks/pipelined_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
run pipe2 24 65536 16 2
run pipe3 24 65536 16 1
run pipe4 24 65536 16 1
run pipe3 24 65536 16 2
run pipe2 24 65536 16 1
run pipe2 24 65536 12 2
run pipe2 24 65536 24 2
run pipe4 24 65536 8 1
run pipe2 24 32768 16 2
run pipe2 24 32768 16 1
run pipe4 24 16384 16 1
run sync 12 65536 16 4
