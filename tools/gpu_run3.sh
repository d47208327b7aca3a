set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py event_threshold 1,4,8,16,24,32,40,48,56,64 > gpurun_out/sweep_threshold.log 2>&1
echo done
