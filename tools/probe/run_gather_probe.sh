# usage: tools/probe/run_gather_probe.sh  -> gpurun_out/gather_probe/{plain.jsonl,pmc_*.txt}
R=$GRAFT_REPO_ROOT
P=$R/tools/probe/gather_probe
[ -x $P ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/probe/gather_probe.hip -o $P
OUT=$R/gpurun_out/gather_probe
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MB in 16 128 1024 16384 65536; do
  for B in 4 32; do
    timeout 120 $P $MB $B 64 >> $OUT/plain.jsonl 2>&1
  done
done
cat $OUT/plain.jsonl
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA_RDREQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCC_EA0_RD_UNCACHED[A-Za-z0-9_]*" | sort -u | head -20 > $OUT/counters.txt
cat $OUT/counters.txt
pmc() { # $1 tag, $2.. counters ; runs the 64 GB / 32 B and 128 MB / 32 B and 64 GB / 4 B cases
  TAG=$1; shift
  for CASE in "65536 32" "128 32" "65536 4"; do
    D=/tmp/gp_$TAG; rm -rf $D
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $D -o gp -- $P $CASE 64 > /tmp/gp.log 2>&1
    python3 - "$D" "$TAG" "$CASE" <<'PY' >> $OUT/pmc.txt
import glob, sqlite3, sys
d, tag, case = sys.argv[1], sys.argv[2], sys.argv[3]
db = (glob.glob(d + '/*/*_results.db') + glob.glob(d + '/*_results.db'))
if not db:
    print(case, tag, "no db"); sys.exit()
cur = sqlite3.connect(db[0]).cursor()
rows = cur.execute("select counter_name, dispatch_id, sum(value) from counters_collection group by counter_name, dispatch_id order by dispatch_id").fetchall()
last = {}
for c, disp, v in rows:
    last[c] = v  # the last dispatch = a timed one
print(case, {k: v for k, v in last.items()})
PY
  done
}
pmc fetch FETCH_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
cat $OUT/pmc.txt
