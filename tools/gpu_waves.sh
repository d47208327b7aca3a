# kernel time of the headline frame for builds with different __launch_bounds__ minimum blocks: build libvrc_w<N>.so with -DVRC_MIN_BLOCKS=<N> first
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so
for w in "$@"; do
  cp voxel-raycaster_amd/libvrc_w$w.so voxel-raycaster_amd/libvrc.so
  echo "min blocks $w"; python tools/sweep.py safe_run 1,1 2>&1 | cut -c1-80
done
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
echo "min blocks 6 (product)"; python tools/sweep.py safe_run 1,1 2>&1 | cut -c1-80
