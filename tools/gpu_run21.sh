mkdir -p gpurun_out
python tools/sweep.py jump_min_run 16777216,1024,512,256,128,64 > gpurun_out/r21.log 2>&1
