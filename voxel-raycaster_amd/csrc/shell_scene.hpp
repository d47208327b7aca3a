// shell_scene.hpp -- the procedural scene "shell-terrain(depth, seed)" of SURVEY 8(d) as pure functions of a column,
// shared by the host builder (svo_builder.cpp) and the device builder (svo_builder_gpu.hip) so both evaluate the
// same integer arithmetic.
//
//   h(x, y)   = dim/4 + sum over octaves o = 0.. of bilinear value noise with cell 2^k, k = depth-2-o >= octave_floor,
//               lattice values hash(seed, o, i, j) mod (dim >> (o+2))   (all integer, shift instead of divide)
//   solid     iff lo(x, y) <= z <= h(x, y),  lo = max(0, min(h over the 4-neighbourhood and the column) - thickness)
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define VRC_HD __host__ __device__ inline
#else
#define VRC_HD inline
#endif

namespace vrc {

VRC_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

VRC_HD uint64_t lattice(uint64_t seed, int octave, int64_t i, int64_t j) {
    return splitmix64(seed * 0x100000001B3ULL ^ splitmix64(((uint64_t)octave << 56) ^ ((uint64_t)i << 28) ^ (uint64_t)j));
}

// height of one column, evaluated procedurally (the table-based make_heightfield of svo_builder.cpp gives the same
// numbers; tests/test_oracle_cpu.py compares them)
VRC_HD int32_t shell_height(uint32_t depth, uint64_t seed, int octave_floor, int64_t x, int64_t y) {
    const int64_t dim = 1LL << depth;
    int64_t h = dim / 4;
    int o = 0;
    for (int k = (int)depth - 2; k >= octave_floor; k--, o++) {
        const int64_t cell = 1LL << k;
        const int64_t amp = dim >> (o + 2);
        if (amp <= 0) break;
        const int64_t i = x >> k, j = y >> k, fx = x & (cell - 1), fy = y & (cell - 1);
        const uint64_t am = (uint64_t)(amp - 1);                  // amp is a power of two: "mod amp"
        const int64_t v00 = (int64_t)(lattice(seed, o, i, j) & am), v10 = (int64_t)(lattice(seed, o, i + 1, j) & am);
        const int64_t v01 = (int64_t)(lattice(seed, o, i, j + 1) & am), v11 = (int64_t)(lattice(seed, o, i + 1, j + 1) & am);
        const int64_t top = v00 * (cell - fx) + v10 * fx, bot = v01 * (cell - fx) + v11 * fx;
        h += (top * (cell - fy) + bot * fy) >> (2 * k);
    }
    return (int32_t)h;
}

// lowest solid voxel of the column: the shell reaches down to the lowest neighbour so it stays watertight on slopes
VRC_HD int32_t shell_floor(int32_t h, int32_t hxm, int32_t hxp, int32_t hym, int32_t hyp, int32_t thickness) {
    int32_t m = h;
    m = hxm < m ? hxm : m; m = hxp < m ? hxp : m; m = hym < m ? hym : m; m = hyp < m ? hyp : m;
    m -= thickness;
    return m < 0 ? 0 : m;
}

}  // namespace vrc
