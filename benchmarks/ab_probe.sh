#!/bin/bash
# A/B on one box: verify_probe with the in-tree library, then with benchmarks/ab/libbpgpu_prev.so swapped in
# (run on the GPU box copy only: it overwrites bulletproofs_amd/libbpgpu.so there).
set -e
timeout -k 10 200 python benchmarks/verify_probe.py --iters 12 > gpurun_out/probe_new.log 2>&1
cp bulletproofs_amd/libbpgpu.so /tmp/lib_new.so
cp benchmarks/ab/libbpgpu_prev.so bulletproofs_amd/libbpgpu.so
timeout -k 10 200 python benchmarks/verify_probe.py --iters 12 > gpurun_out/probe_prev.log 2>&1
cp /tmp/lib_new.so bulletproofs_amd/libbpgpu.so
timeout -k 10 200 python benchmarks/verify_probe.py --iters 12 > gpurun_out/probe_new2.log 2>&1
echo new; grep summary gpurun_out/probe_new.log
echo prev; grep summary gpurun_out/probe_prev.log
echo new2; grep summary gpurun_out/probe_new2.log
