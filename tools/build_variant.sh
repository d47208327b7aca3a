# usage: bash tools/build_variant.sh <name> <extra hipcc flags...>   -> gpurun_variants/libvrc_<name>.so  (A/B builds for tools/gpu_variants.sh)
NAME=$1; shift
CS=voxel-raycaster_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Iinclude "$@" \
  $CS/raycast_kernel.hip $CS/raycast_jump_kernel.hip $CS/svo_builder_gpu.hip $CS/empty_boxes.hip $CS/vrc_api.cpp $CS/svo_builder.cpp -o gpurun_variants/libvrc_$NAME.so
