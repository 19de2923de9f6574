mkdir -p gpurun_out/r4l
timeout 600 python -m pytest tests/test_gpu_accumulate.py -m gpu -q --timeout=600 > gpurun_out/r4l/tests.log 2>&1; echo "tests rc $?"
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r4l/tests.log | tail -12
