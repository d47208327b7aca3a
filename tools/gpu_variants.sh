# usage (GPU box): bash tools/gpu_variants.sh "<frames.py args>" name1 name2 ...  : runs tools/frames.py with each gpurun_variants/libvrc_<name>.so
ARGS=$1; shift
cp voxel-raycaster_amd/libvrc.so /tmp/libvrc_prod.so
for r in 1 2; do
for n in "$@"; do
  cp gpurun_variants/libvrc_$n.so voxel-raycaster_amd/libvrc.so
  echo -n "$n  "; python tools/frames.py $ARGS 2>&1 | cut -c1-120
done
done
cp /tmp/libvrc_prod.so voxel-raycaster_amd/libvrc.so
