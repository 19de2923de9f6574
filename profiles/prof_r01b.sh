set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra"
rm -rf $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o bench -- $B > $R/gpurun_out/prof_trace.log 2>&1
B2="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_fetch -o bench -- $B2 > $R/gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_write -o bench -- $B2 > $R/gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $R/gpurun_out/prof/pmc_tcc -o bench -- $B2 > $R/gpurun_out/prof_pmc3.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/prof/pmc_sq -o bench -- $B2 > $R/gpurun_out/prof_pmc4.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace -d $R/gpurun_out/prof/pmc_sq2 -o bench -- $B2 > $R/gpurun_out/prof_pmc5.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/prof/pmc_sq3 -o bench -- $B2 > $R/gpurun_out/prof_pmc6.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr --kernel-trace -d $R/gpurun_out/prof/pmc_tcp -o bench -- $B2 > $R/gpurun_out/prof_pmc7.log 2>&1
grep -il "error\|invalid\|not found" $R/gpurun_out/prof_pmc*.log
ls $R/gpurun_out/prof
