#!/bin/bash
# round 6, GPU session 6: packet rows from the middle outwards (A/B), where the cage move's time goes with the fine look-up table
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s6
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
V=$R/nerfshop_amd/csrc/variants
for WL in lego_cage lego_cage_varied lego_cage_membrane; do
  bash tools/ab_bench.sh $OUT/ab_roworder_$WL.txt $WL base=default rows=$V/libnrs_roworder.so
done
cat > /tmp/cage_moves.py <<PY
import sys, time
sys.path.insert(0, "$R")
import torch
from nerfshop_amd import runtime as rt, synth
ctx = rt.Context(0)
desc = synth.model_desc(1)
for n in (10, 20):
    e = synth.make_cage_edit(lattice_n=n)
    op = rt.CageDeformation(ctx, desc, e, device_authoring=True)
    op.set_mvc(e.mvc_weights)
    poses = [synth.deform_cage(e.cage_vertices, (0.10 * k / 10, 0.05, 0.0), 20.0 * k / 10) for k in range(1, 11)]
    for k in range(3):
        op.update_cage(None, poses[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20):
        op.update_cage(None, poses[k % 10])
    torch.cuda.synchronize()
    print("lattice", n, "ms per move", (time.perf_counter() - t0) * 1e3 / 20, flush=True)
PY
python /tmp/cage_moves.py
NRS_NO_FINE_LUT=1 python /tmp/cage_moves.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cm; rocprofv3 --kernel-trace --stats -d /tmp/cm -o cm -- python /tmp/cage_moves.py > /tmp/cm.log 2>&1
python - <<PY
import glob, sqlite3
db = glob.glob("/tmp/cm/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
for n, c, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 20").fetchall():
    print(f"{n[:70]:70s} calls {c:5d} total {tot/1e3:9.2f} ms avg {avg:9.1f} us {pct:6.2f}%")
PY
