mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_introspection.py tests/test_gpu_modes.py tests/test_gpu_numerics.py tests/test_gpu_numerics_total.py -m gpu -q --maxfail=20 -rf --timeout=600 -s > gpurun_out/r4c/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4c/tests.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r4c/tests.log | tail -60
