mkdir -p gpurun_out/r4a
tools/probe/mixlo_probe > gpurun_out/r4a/mixlo.json 2>&1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -rf -x --timeout=900 > gpurun_out/r4a/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4a/tests.log
timeout 600 python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
timeout 400 tools/ab_env.sh gpurun_out/r4a/ab_membrane.txt lego_cage_membrane "t0_8w=NRS_X=1" "t0_12w=NRS_RENDER_CFG=124" "catchall=NRS_TEAM=1" > /dev/null 2>&1
timeout 400 tools/ab_env.sh gpurun_out/r4a/ab_numerics.txt lego_cage_tcnn_numerics "static=NRS_X=1" "runtime=NRS_RENDER_CFG=84" > /dev/null 2>&1
tail -5 gpurun_out/r4a/tests.log
