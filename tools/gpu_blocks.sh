# kernel time of the headline frame for builds with 1/2/8 wave tiles per block: build libvrc_b<N>.so with -DVRC_TILES_PER_BLOCK=<N> first
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so
for r in 1 2; do
for t in "$@"; do
  cp voxel-raycaster_amd/libvrc_b$t.so voxel-raycaster_amd/libvrc.so
  echo -n "tiles per block $t: "; python tools/sweep.py safe_run 1 2>&1 | cut -c17-45
done
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
echo -n "tiles per block 4 (product): "; python tools/sweep.py safe_run 1 2>&1 | cut -c17-45
done
