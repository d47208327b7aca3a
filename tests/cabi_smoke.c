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
        vrc_destroy(h);
        printf("c-abi ok (gpu present)\n");
    } else {
        if (vrc_create(0, &h) == VRC_OK) return 9;
        printf("c-abi ok (no gpu)\n");
    }
    return 0;
}
