mkdir -p gpurun_out/r4f
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4f/trace -o t -- python $R/tools/small_launch_probe.py 8 > $R/gpurun_out/r4f/probe.log 2>&1
cd $R
python - <<'PY'
import csv, glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/r4f/trace/**/t_kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40 kernels: show name, duration, gap to previous
prev=None
out=[]
for r in rows[-60:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.append((r['Kernel_Name'][:50], (e-s)/1e3, (s-prev)/1e3 if prev else 0))
    prev=e
open('gpurun_out/r4f/timeline.txt','w').write('\n'.join(f"{n:50s} dur {d:8.1f} us  gap {g:7.1f} us" for n,d,g in out))
PY
tail -24 gpurun_out/r4f/timeline.txt; grep "share" gpurun_out/r4f/probe.log
find gpurun_out/r4f -name "*.db" -delete
