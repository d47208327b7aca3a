# usage (GPU box): bash tools/gpu_variants.sh "<frames.py args>" name1 name2 ...
# runs tools/frames.py with each gpurun_variants/libvrc_<name>.so (built by tools/build_variant.sh), twice, A/B inside one GPU
# call.  The variant is loaded through VRC_LIB_PATH: the product library voxel-raycaster_amd/libvrc.so is never touched.
ARGS=$1; shift
for r in 1 2; do
for n in "$@"; do
  echo -n "$n  "; VRC_LIB_PATH=gpurun_variants/libvrc_$n.so python tools/frames.py $ARGS 2>&1 | cut -c1-120
done
done
