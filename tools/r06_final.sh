#!/bin/bash
# round 6, final GPU session: the whole GPU test suite, smoke, the bench line (twice), [with "prof": rocprofv3 passes of the final build -> profiles/r06_final.md, r06_traffic.json]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_final
mkdir -p $OUT
cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $OUT/gpu_tests.log 2>&1
tail -5 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; tail -4 $OUT/bench.err
python bench.py > $OUT/bench2.json 2> $OUT/bench2.err
if [ "$1" == "prof" ]; then
bash tools/prof.sh lego_cage r06_final > $OUT/prof_lego.log 2>&1
for W in lego_cage_varied lego_cage_membrane lego_cage_tcnn_numerics garden_cage_records64; do
  bash tools/prof.sh $W r06_$W quick > $OUT/prof_$W.log 2>&1
done
python tools/make_traffic_json.py $OUT/r06_traffic.json lego_cage=$R/gpurun_out/prof_r06_final lego_cage_varied=$R/gpurun_out/prof_r06_lego_cage_varied lego_cage_membrane=$R/gpurun_out/prof_r06_lego_cage_membrane lego_cage_tcnn_numerics=$R/gpurun_out/prof_r06_lego_cage_tcnn_numerics garden_cage_records64=$R/gpurun_out/prof_r06_garden_cage_records64
fi
