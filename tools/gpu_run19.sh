mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r19.log
python tools/sweep.py jump_min_run 16777216,32,16,8 >> gpurun_out/r19.log 2>&1
python tools/sweep.py event_threshold 16,64,256 >> gpurun_out/r19.log 2>&1
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so; cp voxel-raycaster_amd/libvrc_stats.so voxel-raycaster_amd/libvrc.so
python tools/sweep.py jump_min_run 16 >> gpurun_out/r19.log 2>&1
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
