R=$GRAFT_REPO_ROOT
P=$R/tools/probe/gather_probe
[ -x $P ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/probe/gather_probe.hip -o $P
OUT=$R/gpurun_out/gather_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CASE in "65536 32" "65536 4" "128 32"; do
  D=/tmp/gp_ea2; rm -rf $D
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace -d $D -o gp -- $P $CASE 64 > /tmp/gp.log 2>&1
  python3 - "$D" "$CASE" <<'PY' | tee -a $OUT/pmc2.txt
import glob, sqlite3, sys
d, case = sys.argv[1], sys.argv[2]
db = (glob.glob(d + '/*/*_results.db') + glob.glob(d + '/*_results.db'))
if not db:
    print(case, "no db"); sys.exit()
cur = sqlite3.connect(db[0]).cursor()
rows = cur.execute("select counter_name, dispatch_id, sum(value) from counters_collection group by counter_name, dispatch_id order by dispatch_id").fetchall()
last = {}
for c, disp, v in rows:
    last[c] = v
print(case, last)
PY
done
