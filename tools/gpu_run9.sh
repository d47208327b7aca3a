set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py xcd_mode 0,1,2 > gpurun_out/sweep_xcd.log 2>&1
echo done
