set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py event_threshold 1,2,4,6,8,12,16,24 > gpurun_out/sweep_a.log 2>&1
timeout 600 python tools/sweep.py shade_threshold 1,8,32,64 > gpurun_out/sweep_b.log 2>&1
echo done
