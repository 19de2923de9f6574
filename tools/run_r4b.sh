mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -rf --timeout=900 --ignore=tests/test_gpu_introspection.py > gpurun_out/r4b/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4b/tests.log
V=nerfshop_amd/csrc/variants
timeout 600 tools/ab_bench.sh gpurun_out/r4b/ab_numerics.txt lego_cage_tcnn_numerics new=default sel0=$V/libnrs_sel0.so mix1=$V/libnrs_mix1.so noquad3=$V/libnrs_noquad3.so > /dev/null 2>&1
timeout 300 tools/ab_bench.sh gpurun_out/r4b/ab_default.txt lego_cage new=default > /dev/null 2>&1
tail -5 gpurun_out/r4b/tests.log
