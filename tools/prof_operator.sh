# usage: tools/prof_operator.sh <tag>  -- rocprofv3 trace + PMC passes of tools/op_driver.py -> gpurun_out/prof_<tag>/summary.md
export NRS_DEV_KNOBS=1  # the measurement knobs of libnrs are ignored without it (nrs_internal.h: dev_knob)
R=$GRAFT_REPO_ROOT
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python $R/tools/op_driver.py"
$B > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o op -- $B > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o op -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o op -- $B > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT/pmc_tcc -o op -- $B > $OUT/pmc3.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr --kernel-trace -d $OUT/pmc_tcp -o op -- $B > $OUT/pmc4.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o op -- $B > $OUT/pmc5.log 2>&1
python $R/tools/summarize_operator.py $OUT $OUT/summary.md
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.md
