# scheduler statistics (profiling build) for a sweep: bash tools/gpu_stats2.sh <setting> <values>
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so; cp voxel-raycaster_amd/libvrc_stats.so voxel-raycaster_amd/libvrc.so
python tools/sweep.py $1 $2 2>&1
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
