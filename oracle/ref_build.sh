#!/bin/bash
# Builds the REAL reference for oracle pinning (outputs only into oracle/_ref/, git-ignored):
#   1. the reference kernel, UNMODIFIED, from where it lies under /root/reference, compiled for
#      gfx950 with the image's own OpenCL device libraries (no stand-ins of any kind);
#   2. a small OpenCL host runner (oracle/ref_run.c, ours) that feeds it the 16 arguments of
#      CLCaster::validate (src/CLCaster.cpp:186-202) through the AMD OpenCL runtime on the GPU box.
# The -D defines are the settings-buffer slots the reference application creates
# (src/Application.cpp:35-39, src/CLCaster.cpp:113,767-769): OCTDIM=0 OCTENABLED=1 OCTREE_ROOT_INDEX=2.
# Variants: "strict" = IEEE, no contraction (what the CPU oracle restates);
#           "shipped" = the reference's own build options (src/CLCaster.cpp:767-771).
set -euo pipefail
cd "$(dirname "$0")"
REF=${REF:-/root/reference}
CLANG=/opt/rocm/lib/llvm/bin/clang
OUT=_ref
mkdir -p "$OUT"
if [ ! -f "$REF/kernels/ray_caster_kernel.cl" ]; then
  echo "ref_build: $REF not present (GPU box?) -- keeping prebuilt files in $OUT" >&2
  exit 0
fi
COMMON="-x cl -cl-std=CL1.2 -Xclang -finclude-default-header -target amdgcn-amd-amdhsa -mcpu=gfx950 -O2 -w \
  -DOCTDIM=0 -DOCTENABLED=1 -DOCTREE_ROOT_INDEX=2"
for cov in 5 6; do
  $CLANG $COMMON -mcode-object-version=$cov -ffp-contract=off \
      "$REF/kernels/ray_caster_kernel.cl" -o "$OUT/raycaster_strict_cov$cov.co"
  $CLANG $COMMON -mcode-object-version=$cov -cl-finite-math-only -cl-fast-relaxed-math -cl-unsafe-math-optimizations \
      "$REF/kernels/ray_caster_kernel.cl" -o "$OUT/raycaster_shipped_cov$cov.co"
done
gcc -O2 -std=c11 -Wall -DCL_TARGET_OPENCL_VERSION=120 -I/opt/rocm/include ref_run.c -o "$OUT/ref_run" -lOpenCL
ls -la "$OUT"
