mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r15.log
python tools/sweep.py event_threshold 4,6,10 >> gpurun_out/r15.log 2>&1
for w in 6 8; do cp voxel-raycaster_amd/libvrc_lb$w.so voxel-raycaster_amd/libvrc.so; echo "launch_bounds $w" >> gpurun_out/r15.log; python tools/sweep.py event_threshold 6 >> gpurun_out/r15.log 2>&1; done
