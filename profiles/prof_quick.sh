R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
B2="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/prof/pmc_sq -o bench -- $B2 > $R/gpurun_out/prof_pmc4.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA --kernel-trace -d $R/gpurun_out/prof/pmc_sq2 -o bench -- $B2 > $R/gpurun_out/prof_pmc5.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr --kernel-trace -d $R/gpurun_out/prof/pmc_tcp -o bench -- $B2 > $R/gpurun_out/prof_pmc7.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace -d $R/gpurun_out/prof/pmc_grbm -o bench -- $B2 > $R/gpurun_out/prof_pmc8.log 2>&1
python - <<'PY'
import sqlite3, glob, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof'
for db in sorted(glob.glob(root+'/*/bench_results.db')):
    cur=sqlite3.connect(db).cursor()
    try:
        rows=cur.execute("select counter_name, avg(value), avg(duration) from counters_collection where kernel_name like '%render_kernel%' group by counter_name").fetchall()
    except Exception as e:
        rows=[]; print(db, e)
    for r in rows: print(f"{r[0]:36s} {r[1]:.6g}   (kernel avg {r[2]/1e3:.1f} us)")
PY
