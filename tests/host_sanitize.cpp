// Host-side code of libvrc.so (csrc/svo_builder.cpp: builders, scene functions, file format) under AddressSanitizer +
// UBSan (CPU only: GPU sanitizers are not available on the pool).  Built and run by
// tests/test_builder_cpu.py::test_host_builders_under_sanitizers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vrc.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "check failed: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const char *tmp = argc > 1 ? argv[1] : "/tmp/vrc_sanitize.svo";
    for (int depth = 3; depth <= 8; depth++) {
        for (uint32_t layout = 0; layout < 4; layout++) {
            uint64_t *d = nullptr, n = 0, root = 0;
            std::vector<int32_t> h((size_t)1 << (2 * depth));
            CHECK(vrc_scene_shell_terrain_ex(depth, 1, depth == 8 ? 9 : 2, layout & 1 ? 0 : 2, layout, &d, &n, &root, h.data()) == VRC_OK);
            CHECK(n > 0 && root < n);
            const uint32_t dim = 1u << depth;
            for (int s = 0; s < 2000; s++) {                    // GetVoxel vs the procedural column
                const int32_t pos[3] = {(int32_t)(rand() % dim), (int32_t)(rand() % dim), (int32_t)(rand() % dim)};
                int32_t found = -1, res = -1, sub[3], lo = 0, hi = 0;
                CHECK(vrc_octree_get_voxel(d, root, dim, pos, &found, &res, sub) == VRC_OK);
                CHECK(vrc_scene_shell_column(depth, 1, depth == 8 ? 9 : 2, layout & 1 ? 0 : 2, pos[0], pos[1], &lo, &hi) == VRC_OK);
                CHECK((found != 0) == (pos[2] >= lo && pos[2] <= hi));
            }
            uint32_t *lookup = nullptr; uint64_t *att = nullptr, na = 0;
            CHECK(vrc_scene_shell_terrain_attachments(depth, 1, 16, d, n, root, &lookup, &att, &na) == VRC_OK);
            CHECK(vrc_octree_save(tmp, dim, d, n, root, lookup, att, na) == VRC_OK);
            uint32_t dim2 = 0, *lookup2 = nullptr; uint64_t *d2 = nullptr, n2 = 0, root2 = 0, *att2 = nullptr, na2 = 0;
            CHECK(vrc_octree_load(tmp, &dim2, &d2, &n2, &root2, &lookup2, &att2, &na2) == VRC_OK);
            CHECK(dim2 == dim && n2 == n && root2 == root && na2 == na && memcmp(d, d2, n * 8) == 0 && memcmp(att, att2, na * 8) == 0);
            vrc_free(d); vrc_free(d2); vrc_free(lookup); vrc_free(lookup2); vrc_free(att); vrc_free(att2);
        }
    }
    {   // dense builder, strict and not, with a 100000-entry buffer and exactly sized
        const uint32_t dim = 32;
        std::vector<int8_t> g((size_t)dim * dim * dim);
        for (size_t i = 0; i < g.size(); i++) g[i] = (rand() % 100) < 30 ? 5 : 0;
        for (int strict = 0; strict < 2; strict++)
            for (uint64_t size : {(uint64_t)0, (uint64_t)100000}) {
                uint64_t *d = nullptr, n = 0, root = 0;
                CHECK(vrc_octree_generate(g.data(), dim, size, strict, &d, &n, &root) == VRC_OK);
                uint32_t *lookup = nullptr; uint64_t *att = nullptr, na = 0;
                CHECK(vrc_octree_attachments_from_grid(g.data(), dim, d, n, root, &lookup, &att, &na) == VRC_OK);
                vrc_free(d); vrc_free(lookup); vrc_free(att);
            }
        uint64_t *d = nullptr, n = 0, root = 0;
        CHECK(vrc_octree_generate(g.data(), dim, 100, 1, &d, &n, &root) == VRC_ERR_LIMIT);   // too small: reported, not corrupted
    }
    {   // diamond-square + atlas
        std::vector<uint8_t> h(64 * 64), a(4 * 256 * 256);
        std::vector<int8_t> g(64 * 64 * 64);
        CHECK(vrc_scene_diamond_square(64, 58.0, h.data(), g.data()) == VRC_OK);
        CHECK(vrc_scene_atlas(256, 256, a.data()) == VRC_OK);
    }
    remove(tmp);
    printf("host sanitize ok\n");
    return 0;
}
