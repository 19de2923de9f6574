set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o bench -- $B > $R/gpurun_out/prof_trace.log 2>&1
tail -2 $R/gpurun_out/prof_trace.log
B2="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_fetch -o bench -- $B2 > $R/gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_write -o bench -- $B2 > $R/gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/prof/pmc_tcc -o bench -- $B2 > $R/gpurun_out/prof_pmc3.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/prof/pmc_sq -o bench -- $B2 > $R/gpurun_out/prof_pmc4.log 2>&1
tail -3 $R/gpurun_out/prof_pmc4.log
find $R/gpurun_out/prof -type f | head -40
