mkdir -p gpurun_out
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so; cp voxel-raycaster_amd/libvrc_stats.so voxel-raycaster_amd/libvrc.so
python tools/sweep.py jump_min_run 16777216,32,16,4 > gpurun_out/r18.log 2>&1
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
