# usage: tools/prof.sh <workload> <tag> [quick]   -- rocprofv3 trace + PMC passes of bench.py for one workload -> gpurun_out/prof_<tag>/
set -x
R=$GRAFT_REPO_ROOT
W=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python $R/bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $B > $OUT/trace.log 2>&1
# (a warmed-up run: the average launch the bench's HIP events report)
rocprofv3 --kernel-trace --stats -d $OUT/trace32 -o bench -- python $R/bench.py --workload $W --steps 32 --warmup 4 --no-cpu-baseline --no-extra > $OUT/trace32.log 2>&1
B2="python $R/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- $B2 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- $B2 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT/pmc_tcc -o bench -- $B2 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr --kernel-trace -d $OUT/pmc_tcp -o bench -- $B2 > $OUT/pmc7.log 2>&1
if [ "$3" != "quick" ]; then
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/pmc_sq -o bench -- $B2 > $OUT/pmc4.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/pmc_sq2 -o bench -- $B2 > $OUT/pmc5.log 2>&1
fi
python $R/profiles/summarize.py $OUT $OUT/summary.md "$W ($TAG)"
python - "$OUT" <<PY
import json, os, sys
sys.path.insert(0, "$R/tools")
from pmc_kernel import read
out = sys.argv[1]
json.dump({d: read(os.path.join(out, d)) for d in sorted(os.listdir(out)) if os.path.isdir(os.path.join(out, d))}, open(os.path.join(out, "counters.json"), "w"), indent=1)
PY
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.md | grep -v "^$" | head -70
