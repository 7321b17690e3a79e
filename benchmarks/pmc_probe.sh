R=$(pwd); OUT=$R/gpurun_out/pmc_r02w; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --msm-logn 0 --extras 0"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INST_LEVEL_VMEM -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err
cd $R
python benchmarks/pmc_summary.py $(find $OUT/sq -name "*.db" | head -1) > $OUT/pmc_sq.txt
python benchmarks/pmc_summary.py $(find $OUT/sq2 -name "*.db" | head -1) > $OUT/pmc_sq2.txt 2>&1
find $OUT -name "*.db" -delete
grep fixed_multiexp $OUT/pmc_sq.txt $OUT/pmc_sq2.txt
