mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -rf --timeout=900 > gpurun_out/r4i/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4i/tests.log
timeout 600 python bench.py > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err
bash tools/prof_r04.sh lego_cage r04_lego > gpurun_out/r4i/prof_lego.log 2>&1
bash tools/prof_r04.sh lego_cage_membrane r04_membrane quick > gpurun_out/r4i/prof_membrane.log 2>&1
bash tools/prof_r04.sh garden_cage r04_garden quick > gpurun_out/r4i/prof_garden.log 2>&1
bash tools/prof_r04.sh lego_cage_tcnn_numerics r04_tcnn > gpurun_out/r4i/prof_tcnn.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r4i/tests.log | tail -5
