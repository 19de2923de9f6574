R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof2
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2/trace -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof2_trace.log 2>&1
python - <<'PY'
import sqlite3, glob, os
db = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof2/trace/*.db')[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select name,total_calls,average from top_kernels limit 4"): print(r)
PY
