# A/B of two library builds on the headline frame in one gpurun call: bash tools/gpu_ab.sh <libA.so> <libB.so> [rounds]
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so
for r in $(seq 1 ${3:-3}); do
  for l in "$1" "$2"; do
    cp "$l" voxel-raycaster_amd/libvrc.so
    echo -n "$l  "; python tools/sweep.py safe_run 1 2>&1 | cut -c17-60
  done
done
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
