# kernel time of the headline frame for builds with different wave tile shapes: build libvrc_t<W>.so with -DVRC_TILE_W=<W> first
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so
for w in "$@"; do
  cp voxel-raycaster_amd/libvrc_t$w.so voxel-raycaster_amd/libvrc.so
  echo "tile ${w} x $((64 / w))"; python tools/sweep.py safe_run 1,1 2>&1 | cut -c1-80
done
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
echo "tile 8 x 8 (product)"; python tools/sweep.py safe_run 1,1 2>&1 | cut -c1-80
