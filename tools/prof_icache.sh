# instruction-fetch / scalar-cache / stall counters of the render kernel: tools/prof_icache.sh <workload> <tag>  (through gpurun)
export NRS_DEV_KNOBS=1  # the measurement knobs of libnrs are ignored without it (nrs_internal.h: dev_knob)
R=$GRAFT_REPO_ROOT
W=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z0-9_]*" | sort -u > $OUT/sq_counters.txt
B2="python $R/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAIT_IFETCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o t -- $B2 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set" >> $OUT/fail.txt
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(agg):
        v = agg[k]
        o.write(f"{k:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open("$OUT/summary.txt").read())
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +1M -delete
cat $OUT/fail.txt 2>/dev/null; tail -3 $OUT/p1.log
