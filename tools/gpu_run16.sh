mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r16.log
python tools/sweep.py jump_min_run 16777216,64,32,16,8,4 >> gpurun_out/r16.log 2>&1
