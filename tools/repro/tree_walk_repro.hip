// Repro harness for the advisor's round-3 finding on svo_builder_gpu.hip tree_holds(): an earlier formulation of the point
// query -- the reference's running box corner, Octree::GetVoxel (src/map/Octree.cpp:45-158), with 64-bit corners -- was said
// to give run-to-run different answers inside validate_grid_kernel on a static tree, and was replaced by the bit-per-level
// form without the cause being found.  The failing source was never committed; this file restates BOTH forms in a kernel of
// their own and runs them over EVERY voxel of static trees (page headers + far pointers included), repeatedly, at -O3 and
// -O0, against the dense grid and against each other, launches serialised:
//     hipcc --offload-arch=gfx950 -O3 tools/repro/tree_walk_repro.hip -Iinclude -Lvoxel-raycaster_amd -lvrc -Wl,-rpath,$PWD/voxel-raycaster_amd -o /tmp/twr && /tmp/twr
// Any difference -- between runs, between the forms, or from the grid -- is printed with the voxel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "vrc.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
constexpr uint64_t kFar = 0x8000ULL;

// the reference's form: running corner, compared against position (64-bit like the first device version)
__device__ int holds_corner(const uint64_t *desc, uint64_t root, int depth, int64_t x, int64_t y, int64_t z) {
    uint64_t index = root, d = desc[index];
    int64_t half = (1LL << depth) / 2, cx = 0, cy = 0, cz = 0;
    for (;;) {
        int i = 0;
        if (x >= cx + half) { i |= 1; cx += half; }
        if (y >= cy + half) { i |= 2; cy += half; }
        if (z >= cz + half) { i |= 4; cz += half; }
        if (!((d >> (16 + i)) & 1ULL)) return 0;
        if ((d >> (24 + i)) & 1ULL) return 1;
        if (half == 1) return 1;
        const uint64_t at = index + (d & 0x7fffULL);
        const uint64_t first = (d & kFar) ? desc[at] : at;
        index = first + (uint64_t)(__popcll((d >> 16) & ((2ULL << i) - 1ULL)) - 1);
        d = desc[index];
        half >>= 1;
    }
}
// the shipped form: one bit of each coordinate per level
__device__ int holds_bits(const uint64_t *desc, uint64_t root, int depth, uint32_t x, uint32_t y, uint32_t z) {
    uint64_t index = root, d = desc[index];
    for (int l = depth - 1;; l--) {
        const uint32_t i = ((x >> l) & 1u) | (((y >> l) & 1u) << 1) | (((z >> l) & 1u) << 2);
        if (!((d >> (16 + i)) & 1ULL)) return 0;
        if (((d >> (24 + i)) & 1ULL) || l == 0) return 1;
        const uint64_t at = index + (d & 0x7fffULL);
        const uint64_t first = (d & kFar) ? desc[at] : at;
        index = first + (uint64_t)(__popcll((d >> 16) & ((2ULL << i) - 1ULL)) - 1);
        d = desc[index];
    }
}
// the sampling of validate_grid_kernel (x is MODIFIED before the walk, like there), one thread per voxel
template <int kForm>
__global__ void walk_kernel(const int8_t *grid, int depth, const uint64_t *desc, uint64_t root, uint8_t *out, unsigned long long *mismatch) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t dim = 1LL << depth;
    if (s >= (uint64_t)(dim * dim * dim)) return;
    int64_t x = (int64_t)(s & (uint64_t)(dim - 1)), y = (int64_t)((s >> depth) & (uint64_t)(dim - 1)), z = (int64_t)(s >> (2 * depth));
    if (s & 1) {
        for (int k = 0; k < 64 && x + 1 < dim && !grid[x + dim * (y + dim * z)]; k++) x++;
        if ((s & 2) && x > 0) x--;
    }
    const int expect = grid[x + dim * (y + dim * z)] != 0;
    const int found = kForm == 0 ? holds_corner(desc, root, depth, x, y, z) : holds_bits(desc, root, depth, (uint32_t)x, (uint32_t)y, (uint32_t)z);
    out[s] = (uint8_t)found;
    if (found != expect) atomicAdd(mismatch, 1ULL);
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 25;
    unsigned long long total_bad = 0;
    for (int depth = 5; depth <= 7; depth++)
        for (int fill = 0; fill < 3; fill++) {
            const int64_t dim = 1LL << depth, n = dim * dim * dim;
            std::vector<int8_t> grid(n);
            uint64_t st = 0x9E3779B97F4A7C15ULL * (uint64_t)(depth * 8 + fill + 1);
            const unsigned thresh[3] = {3, 40, 128};              // sparse, medium (page headers + far pointers at depth 7), half full
            for (int64_t i = 0; i < n; i++) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; grid[i] = ((st >> 56) < thresh[fill]) ? 5 : 0; }
            uint64_t *desc = nullptr, nd = 0, root = 0;
            if (vrc_octree_generate(grid.data(), (uint32_t)dim, 0, 1, &desc, &nd, &root) != 0) { printf("vrc_octree_generate failed\n"); return 2; }
            int8_t *d_grid; uint64_t *d_desc; uint8_t *d_out; unsigned long long *d_mis;
            CK(hipMalloc(&d_grid, n)); CK(hipMalloc(&d_desc, nd * 8)); CK(hipMalloc(&d_out, n)); CK(hipMalloc(&d_mis, 8));
            CK(hipMemcpy(d_grid, grid.data(), n, hipMemcpyHostToDevice)); CK(hipMemcpy(d_desc, desc, nd * 8, hipMemcpyHostToDevice));
            std::vector<uint8_t> first[2], cur(n);
            for (int form = 0; form < 2; form++)
                for (int r = 0; r < reps; r++) {
                    CK(hipMemset(d_mis, 0, 8)); CK(hipMemset(d_out, 0xee, n));
                    if (form == 0) hipLaunchKernelGGL(walk_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_grid, depth, d_desc, root, d_out, d_mis);
                    else hipLaunchKernelGGL(walk_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_grid, depth, d_desc, root, d_out, d_mis);
                    CK(hipDeviceSynchronize());
                    unsigned long long mis = 0;
                    CK(hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(cur.data(), d_out, n, hipMemcpyDeviceToHost));
                    if (r == 0) first[form] = cur;
                    unsigned long long diff = 0;
                    for (int64_t i = 0; i < n; i++) if (cur[i] != first[form][i]) { if (diff < 4) printf("  form %d run %d voxel %lld: %d, first run %d\n", form, r, (long long)i, cur[i], first[form][i]); diff++; }
                    if (mis || diff) { printf("depth %d fill %d form %d run %d: %llu answers differ from the grid, %llu from the first run\n", depth, fill, form, r, mis, diff); total_bad += mis + diff; }
                }
            unsigned long long cross = 0;
            for (int64_t i = 0; i < n; i++) cross += first[0][i] != first[1][i];
            printf("depth %d fill %u/256: %llu descriptors, root %llu, %d runs x 2 forms x %lld voxels; corner form vs bit form: %llu differences\n", depth, thresh[fill],
                   (unsigned long long)nd, (unsigned long long)root, reps, (long long)n, cross);
            total_bad += cross;
            CK(hipFree(d_grid)); CK(hipFree(d_desc)); CK(hipFree(d_out)); CK(hipFree(d_mis)); vrc_free(desc);
        }
    printf("total differences: %llu\n", total_bad);
    return total_bad ? 1 : 0;
}
