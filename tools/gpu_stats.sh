# scheduler statistics of the headline frame with the profiling build (libvrc_stats.so)
mkdir -p gpurun_out
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so; cp voxel-raycaster_amd/libvrc_stats.so voxel-raycaster_amd/libvrc.so
python tools/sweep.py burst_steps ${1:-48} > gpurun_out/stats.log 2>&1
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
