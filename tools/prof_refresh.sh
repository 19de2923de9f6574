export NRS_DEV_KNOBS=1  # the measurement knobs of libnrs are ignored without it (nrs_internal.h: dev_knob)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ORD in 0 1; do
  rm -rf /tmp/pr$ORD
  NRS_REFRESH_ORDER=$ORD rocprofv3 --kernel-trace --stats -d /tmp/pr$ORD -o r -- python $R/profiles/bench_next_rows.py --no-cpu --reps 10 > /tmp/pr$ORD.log 2>&1
  echo "== NRS_REFRESH_ORDER=$ORD"; grep occupancy_refresh /tmp/pr$ORD.log
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pr$ORD/*/*_results.db') + glob.glob('/tmp/pr$ORD/*_results.db')
cur = sqlite3.connect(db[0]).cursor()
for n, c, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 12"):
    if True: print(f"  {n[:70]:70s} calls {c:4d} avg {avg/1e3:8.3f} ms")
PY
done
python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pr1/*/*_results.db') + glob.glob('/tmp/pr1/*_results.db')
cur = sqlite3.connect(db[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
sym = [t for t in tabs if 'kernel_symbol' in t][0]
rows = cur.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {sym} s on d.kernel_id = s.id where s.kernel_name like '%grid_refresh%' or s.kernel_name like '%occ_accel_mask%' order by d.start").fetchall()
print("  per-dispatch us:", [(n.split('(')[0][-20:], round(t / 1e3)) for n, t in rows][:12], "...", [(n.split('(')[0][-20:], round(t / 1e3)) for n, t in rows][-6:])
PY
