#!/bin/bash
# round 6, GPU session 7: fine look-up table, third build (float box test, coalesced window): parity + cage-move cost + kernel profile
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s7
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_fine_lut.py tests/test_gpu_parity.py tests/test_gpu_cage_update.py tests/test_gpu_grid_refresh.py -x -q -m gpu ) > $OUT/tests_fine_lut.log 2>&1
tail -4 $OUT/tests_fine_lut.log
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
for n, c, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 12").fetchall():
    print(f"{n[:70]:70s} calls {c:5d} total {tot/1e3:9.2f} ms avg {avg:9.1f} us {pct:6.2f}%")
PY
cd $R
bash tools/ab_env.sh $OUT/ab_fine_lut_lego_cage.txt lego_cage "nofine=NRS_NO_FINE_LUT=1" "fine=NRS_NOTHING=1"
