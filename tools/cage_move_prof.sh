#!/bin/bash
# The per-gizmo-move chain kernel by kernel: the device-authoring tests, host-timed moves of the 6 000- and 48 000-tet cages, rocprofv3 kernel traces -> gpurun_out/r06_s15/summary.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s15; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_cage_update.py tests/test_gpu_fine_lut.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp; export TMPDIR=/tmp
python $R/tools/cage_move_driver.py 10 50 2>&1 | tail -1; python $R/tools/cage_move_driver.py 20 50 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d $O/cm6k -o cm -- python $R/tools/cage_move_driver.py 10 50 > $O/p6k.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/cm48k -o cm -- python $R/tools/cage_move_driver.py 20 50 > $O/p48k.log 2>&1
python $R/profiles/summarize.py $O $O/summary.md "cage move chain, 50 moves"
cd $O; find . -name "*.db" -delete; find . -name "*.csv" -size +2M -delete; cat summary.md | head -32
