/* C99 consumer of include/vrc.h: proves the boundary is a plain C ABI (no C++ types, no torch) and exercises
 * the host-only entry points.  Built and run by tests/test_oracle_cpu.py::test_c_consumer. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vrc.h"

int main(void) {
    enum { DIM = 16 };
    static int8_t grid[DIM * DIM * DIM];
    uint64_t *desc = NULL, n = 0, root = 0;
    int32_t pos[3] = {3, 4, 1}, found = -1, res = -1, sub[3];
    int count = -1, i;
    vrc_caster *h = NULL;

    for (i = 0; i < DIM * DIM * 2; i++) grid[i] = 5;                 /* two solid layers z = 0,1 */
    if (vrc_octree_generate(grid, DIM, 0, 0, &desc, &n, &root) != VRC_OK) return 1;
    if (n == 0 || root != 0 || (desc[root] & 0x7fff) != 1) return 2;  /* root pointer forced to 1 (Octree.cpp:27) */
    if (vrc_octree_get_voxel(desc, root, DIM, pos, &found, &res, sub) != VRC_OK || !found) return 3;
    pos[2] = 9;
    if (vrc_octree_get_voxel(desc, root, DIM, pos, &found, &res, sub) != VRC_OK || found) return 4;
    if (vrc_octree_generate(grid, 12, 0, 0, &desc, &n, &root) != VRC_ERR_INVALID_ARGUMENT) return 5;   /* not a power of two */
    vrc_free(desc);

    /* no GPU in the build container: creation must fail loudly, never fall back to a CPU path */
    if (vrc_device_count(&count) == VRC_OK && count > 0) {
        if (vrc_create(0, &h) != VRC_OK || !h) return 6;
        if (vrc_validate(h) != VRC_ERR_NOT_READY) return 7;
        if (strstr(vrc_last_error(h), "camera") == NULL) return 8;
        {   /* the host's own sin / cos of the camera angles: a retained pointer like the camera, NULL = the library's sinf / cosf;
               nothing to prepare before an octree is assigned */
            static const float trig[4] = {0.0f, 1.0f, 0.0f, 1.0f};
            if (vrc_assign_camera_trig(h, trig) != VRC_OK || vrc_assign_camera_trig(h, NULL) != VRC_OK) return 14;
            if (vrc_assign_camera_trig(NULL, trig) != VRC_ERR_INVALID_ARGUMENT) return 15;
            if (vrc_prepare(h) != VRC_ERR_NOT_READY || vrc_prepare(NULL) != VRC_ERR_INVALID_ARGUMENT) return 16;
            {   /* no frame yet: whatever counters there are, they are not a box traversal's */
                int32_t canonical = -1;
                if (vrc_counters_canonical(h, &canonical) != VRC_OK || canonical != 1) return 17;
            }
        }
        {   /* the size-versioned memory report: a caller that knows fewer fields than the library gets only what its struct holds */
            vrc_memory2 m;
            unsigned char guard[sizeof(vrc_memory2) + 8];
            vrc_caster *h2 = NULL;
            memset(&m, 0xee, sizeof(m));
            m.struct_size = (uint32_t)sizeof(m);
            if (vrc_memory_usage2(h, 0, &m) != VRC_OK || m.struct_size != sizeof(m) || m.tree_holders != 0 || m.box_bytes != 0) return 10;
            memset(guard, 0xee, sizeof(guard));
            ((vrc_memory2 *)guard)->struct_size = 24;                  /* an "older header": 24 bytes */
            if (vrc_memory_usage2(h, 0, (vrc_memory2 *)guard) != VRC_OK || guard[24] != 0xee || guard[sizeof(guard) - 1] != 0xee) return 11;
            /* a tree can only be adopted from a handle that has one */
            if (vrc_create(0, &h2) != VRC_OK) return 12;
            if (vrc_assign_octree_from(h2, h) != VRC_ERR_NOT_READY || vrc_assign_octree_from(h, h) != VRC_ERR_INVALID_ARGUMENT) return 13;
            vrc_destroy(h2);
        }
        vrc_destroy(h);
        printf("c-abi ok (gpu present)\n");
    } else {
        if (vrc_create(0, &h) == VRC_OK) return 9;
        printf("c-abi ok (no gpu)\n");
    }
    return 0;
}
