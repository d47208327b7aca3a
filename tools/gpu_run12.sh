mkdir -p gpurun_out
python tools/sweep.py event_threshold 6 > gpurun_out/sweep_prod.log 2>&1
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so; cp voxel-raycaster_amd/libvrc_stats.so voxel-raycaster_amd/libvrc.so
python tools/sweep.py event_threshold 2,6,16,32 > gpurun_out/sweep_stats.log 2>&1
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
