mkdir -p gpurun_out/r4d
V=nerfshop_amd/csrc/variants
timeout 900 tools/ab_bench.sh gpurun_out/r4d/ab_lego.txt lego_cage new=default alloff=$V/libnrs_alloff.so nohq=$V/libnrs_nohq.so noglobal=$V/libnrs_noglobal.so nomorton=$V/libnrs_nomorton.so nopkc=$V/libnrs_nopkc.so > /dev/null 2>&1
timeout 300 tools/ab_bench.sh gpurun_out/r4d/ab_varied.txt lego_cage_varied new=default alloff=$V/libnrs_alloff.so > /dev/null 2>&1
timeout 300 tools/ab_bench.sh gpurun_out/r4d/ab_garden.txt garden_cage new=default alloff=$V/libnrs_alloff.so > /dev/null 2>&1
timeout 300 tools/ab_bench.sh gpurun_out/r4d/ab_membrane.txt lego_cage_membrane new=default alloff=$V/libnrs_alloff.so > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -rf --timeout=900 -x tests/test_gpu_parity.py tests/test_gpu_lane_teams.py tests/test_gpu_cell_cache.py tests/test_gpu_introspection.py > gpurun_out/r4d/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4d/tests.log
cat gpurun_out/r4d/ab_*.txt; tail -3 gpurun_out/r4d/tests.log
