// empty_boxes.hip -- a derived acceleration structure for the SVO kernels: for every EMPTY child slot of every
// descriptor, an axis-aligned box of voxels around that empty node that the tree says holds nothing.
//
// Why: the reference's working traversal (kernels/ray_caster_kernel.cl:555-570) tests one voxel per iteration; the SVO
// kernel already steps through a whole empty octree node without memory (raycast_kernel.hip), but a ray that leaves or
// approaches a surface climbs through the octree's levels -- nodes of 1, 1, 2, 2, 4, 4, ... voxels -- and pays a node event
// for each (profiles/r03_run_length_histogram.txt: the events are spread evenly over run lengths 1 .. 1023).  The octree
// node is only ONE empty box around a voxel; any other empty box serves the step loop equally well, because inside a box
// the loop needs no occupancy at all -- the float recurrence and the iteration count are untouched, so frames stay
// bit-identical.  Occupancy still comes from the descriptor array alone (Octree.h:89-94): the boxes are computed from it
// here, on the device, like the coarse top table.
//
// Layout: boxes[8 * d + k] (uint32) belongs to child slot k = x | y<<1 | z<<2 of descriptor d; meaningful where that child is
// not valid (empty).  Six 5-bit extents, in units of the node's own size s, by which the node is extended on each side:
//   bits 0-4 -x, 5-9 -y, 10-14 -z, 15-19 +x, 20-24 +y, 25-29 +z;   code c -> c (c < 4), (4 | c & 3) << (c / 4 - 1) otherwise
// (0, 1, 2, 3, 4 .. 7, 8, 10 .. 14, 16, 20 .. 28, 32 ... 448: two mantissa bits).  The kernel clamps the box to the map.
// The cells of the coarse top table get a parallel word (box_aux): the box of the empty node a cell resolves to above the
// table's level, or the index of the level-lc descriptor the descent continues from.
//
// Construction: every descriptor's position comes from a level-by-level sweep; then one thread per (descriptor, child)
// grows the box greedily -- each side in turn by one code step, the new slab checked against the tree by a region query
// (depth-first, pruned by the region) -- until every side is blocked or at the map's edge.  Any empty box is a correct
// box; the greedy order only decides how good it is.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#include "vrc_params.h"

namespace vrc {

namespace {

constexpr uint64_t kPosNone = ~0ULL;
constexpr int kPosBits = 19;              // coordinates below 2^19; bits 57-61 hold the level
constexpr int kPosLevelShift = 57;

__device__ __forceinline__ uint64_t bx_entry(const uint64_t *__restrict__ descriptors, uint64_t index, uint64_t d) {
    uint64_t base = index + (d & 0x7fffULL);
    if (d & 0x8000ULL) base = descriptors[base];          // far pointer: the slot holds an absolute index
    return (base << 16) | ((d >> 16) & 0xffffULL);        // bits 0-7 valid, 8-15 leaf, 16.. first child
}

__device__ __forceinline__ uint64_t pack_pos(int x, int y, int z, int level) {
    return (uint64_t)(unsigned)x | ((uint64_t)(unsigned)y << kPosBits) | ((uint64_t)(unsigned)z << (2 * kPosBits)) | ((uint64_t)level << kPosLevelShift);
}

__global__ void box_fill_kernel(uint64_t *p, uint64_t n, uint64_t v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// one level of the position sweep: descriptors of `level` hand their position down to their kept children
__global__ void box_positions_kernel(const uint64_t *__restrict__ descriptors, uint64_t n_desc, int n, int level, uint64_t *__restrict__ pos) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_desc) return;
    const uint64_t ps = pos[idx];
    if (ps == kPosNone || (int)(ps >> kPosLevelShift) != level) return;
    const int half = 1 << (n - level - 1);
    if (half == 1) return;                                // the children are voxels
    const uint64_t e = bx_entry(descriptors, idx, descriptors[idx]);
    const unsigned valid = (unsigned)e & 0xffu, leaf = ((unsigned)e >> 8) & 0xffu;
    const int x = (int)(ps & ((1u << kPosBits) - 1u)), y = (int)((ps >> kPosBits) & ((1u << kPosBits) - 1u)), z = (int)((ps >> (2 * kPosBits)) & ((1u << kPosBits) - 1u));
    unsigned rank = 0;
    for (int k = 0; k < 8; k++) {
        if (!(valid & (1u << k))) continue;
        if (!(leaf & (1u << k))) {
            const uint64_t child = (e >> 16) + rank;
            if (child < n_desc)
                pos[child] = pack_pos(x + ((k & 1) ? half : 0), y + ((k & 2) ? half : 0), z + ((k & 4) ? half : 0), level + 1);
        }
        rank++;
    }
}

__device__ __forceinline__ int box_decode(unsigned c) { return c < 4u ? (int)c : (int)((4u | (c & 3u)) << ((c >> 2) - 1u)); }

// children of the node at (ox, oy, oz) with child size `half` that meet the region [lo, hi)
__device__ __forceinline__ unsigned overlap_mask(int ox, int oy, int oz, int half, const int lo[3], const int hi[3]) {
    unsigned m = 0xffu;
    const int mx = ox + half, my = oy + half, mz = oz + half;
    if (!(lo[0] < mx)) m &= 0xAAu;                        // low x half not touched
    if (!(hi[0] > mx)) m &= 0x55u;
    if (!(lo[1] < my)) m &= 0xCCu;
    if (!(hi[1] > my)) m &= 0x33u;
    if (!(lo[2] < mz)) m &= 0xF0u;
    if (!(hi[2] > mz)) m &= 0x0Fu;
    return m;
}

// Does the tree hold a solid voxel inside [lo, hi)?  (the region lies inside the map and is not empty)
// The search starts at the deepest ANCESTOR of the asking node whose cube holds the whole region (path[level] = the entry of the
// ancestor at that level, (px, py, pz) a voxel of the node): a slab next to a small node near a surface lies within the node's
// parent or grandparent, and a walk from the root would spend ten dependent loads on getting there (the depth-12 build: 1.52 -> 1.12 s).
// region queries of the last box_grow_kernel launch on this device that gave up at their budget (below)
__device__ unsigned long long g_box_queries_cut;

__device__ bool region_is_empty(const uint64_t *__restrict__ descriptors, const uint64_t *path, int path_levels, int px, int py, int pz,
                                int n, const int lo[3], const int hi[3]) {
    // the node on top of the stack lives in registers (entry, children still to look at, origin); what is pushed when the search
    // descends is the parent's entry and its remaining children + the slot taken, from which the origin comes back on the way up
    uint64_t st_entry[kMaxLevels];
    unsigned st_rest[kMaxLevels];                          // bits 0-7 children still to do, 8-10 the slot the search went down into
    // bits in which some corner of the region differs from the node's voxel: the ancestor at level a has a cube of 2^(n - a) voxels
    const unsigned diff = (unsigned)((lo[0] ^ px) | ((hi[0] - 1) ^ px) | (lo[1] ^ py) | ((hi[1] - 1) ^ py) | (lo[2] ^ pz) | ((hi[2] - 1) ^ pz));
    int a = n - (32 - __clz((int)diff));                  // (diff == 0: n)
    a = a < path_levels - 1 ? a : path_levels - 1;
    a = a < 0 ? 0 : a;
    const int base = a;
    const int amask = ~((1 << (n - a)) - 1);
    int level = a;
    uint64_t e = path[a];
    int ox = px & amask, oy = py & amask, oz = pz & amask;
    unsigned todo = overlap_mask(ox, oy, oz, 1 << (n - a - 1), lo, hi) & (unsigned)e & 0xffu;
#ifndef VRC_BOX_QUERY_BUDGET
#define VRC_BOX_QUERY_BUDGET 4096
#endif
    // (a query gives up -- "not empty", the side stops growing -- after this many descents: a bound on what one thread can cost.
    // Correct -- any empty box is a correct box -- but the box is then smaller than it could be: g_box_queries_cut counts how
    // often, vrc_memory_usage2 reports it)
    int budget = VRC_BOX_QUERY_BUDGET;
    for (;;) {
        if (!todo) {
            if (level == base) return true;
            level--;
            e = st_entry[level];
            const unsigned rest = st_rest[level];
            todo = rest & 0xffu;
            const int k = (int)(rest >> 8) & 7, half = 1 << (n - level - 1);
            ox -= (k & 1) ? half : 0; oy -= (k & 2) ? half : 0; oz -= (k & 4) ? half : 0;
            continue;
        }
        const int k = __ffs((int)todo) - 1;
        todo &= todo - 1u;
        const int half = 1 << (n - level - 1);
        if ((((unsigned)e >> 8) & (1u << k)) || half == 1) return false;      // a solid leaf / voxel that meets the region
        if (--budget < 0) { atomicAdd(&g_box_queries_cut, 1ULL); return false; }
        const uint64_t child = (e >> 16) + (uint64_t)(__popc((unsigned)e & 0xffu & ((2u << k) - 1u)) - 1);
        const uint64_t ce = bx_entry(descriptors, child, descriptors[child]);
        const int cx = ox + ((k & 1) ? half : 0), cy = oy + ((k & 2) ? half : 0), cz = oz + ((k & 4) ? half : 0);
        const unsigned ctodo = overlap_mask(cx, cy, cz, half >> 1, lo, hi) & (unsigned)ce & 0xffu;
        if (!ctodo) continue;                             // nothing of the child meets the region: no need to go down
        st_entry[level] = e; st_rest[level] = todo | ((unsigned)k << 8);
        level++;
        e = ce; ox = cx; oy = cy; oz = cz; todo = ctodo;
    }
}

// The box of the empty child slot k of the descriptor at `level` whose cube starts at (nx, ny, nz): path[0 .. level] = the entries of
// the descriptor's ancestors and its own, root first.  Returns the six 5-bit extent codes.
__device__ uint32_t grow_box(const uint64_t *__restrict__ descriptors, const uint64_t *path, int level, int nx, int ny, int nz, int k, int n) {
    const int b = n - level - 1, s = 1 << b, dim = 1 << n;
    int lo[3], hi[3];
    lo[0] = nx + ((k & 1) ? s : 0);
    lo[1] = ny + ((k & 2) ? s : 0);
    lo[2] = nz + ((k & 4) ? s : 0);
    for (int a = 0; a < 3; a++) hi[a] = lo[a] + s;
    const int org_lo[3] = {lo[0], lo[1], lo[2]}, org_hi[3] = {hi[0], hi[1], hi[2]};
    unsigned code[6] = {0, 0, 0, 0, 0, 0};          // -x -y -z +x +y +z
    unsigned stride[6] = {1, 1, 1, 1, 1, 1};        // code steps the side tries next: doubled while it succeeds, halved from its first failure on
    unsigned alive = 0x3fu, doubling = 0x3fu;
    // the sides in turn, z first (open sky above a terrain costs nothing to claim).  A side gallops: 1, 2, 4, ... code steps
    // per turn while the slab is empty, then a binary search back from the first slab that is not -- at most ten queries a
    // side instead of thirty-one single steps (the build of the depth-12 scene: 1.09 -> 0.63 s, depth 10: 61 -> 36 ms)
    const int order[6] = {5, 2, 3, 0, 4, 1};
    while (alive) {
        for (int oi = 0; oi < 6; oi++) {
            const int side = order[oi];
            if (!(alive & (1u << side))) continue;
            const int a = side % 3;
            const bool positive = side >= 3;
            unsigned next = code[side] + stride[side];
            if (next > 31u && code[side] < 31u) next = 31u;
            // the box already reaches the map's edge on this side, or the code is exhausted
            if (next > 31u || (positive ? hi[a] >= dim : lo[a] <= 0)) { alive &= ~(1u << side); continue; }
            const long long ext = (long long)box_decode(next) << b;     // (448 node sizes of a 2^23-voxel node do not fit an int)
            int slo[3] = {lo[0], lo[1], lo[2]}, shi[3] = {hi[0], hi[1], hi[2]};
            int new_edge;
            if (positive) { const long long e = (long long)org_hi[a] + ext; new_edge = e > dim ? dim : (int)e; slo[a] = hi[a]; shi[a] = new_edge; }
            else { const long long e = (long long)org_lo[a] - ext; new_edge = e < 0 ? 0 : (int)e; shi[a] = lo[a]; slo[a] = new_edge; }
            if (positive ? new_edge == dim : new_edge == 0) {
                // clipped at the map's edge: the SMALLEST code that reaches the edge is stored, so that what the kernel decodes
                // (extent << b, in an int) never exceeds twice the map
                const long long need = positive ? (long long)dim - org_hi[a] : (long long)org_lo[a];
                unsigned c = code[side] + 1u;
                while (c < next && ((long long)box_decode(c) << b) < need) c++;
                next = c;
            }
            if (region_is_empty(descriptors, path, level + 1, nx, ny, nz, n, slo, shi)) {
                code[side] = next;
                if (positive) hi[a] = new_edge; else lo[a] = new_edge;
                if (doubling & (1u << side)) stride[side] <<= 1;
                else if ((stride[side] >>= 1) == 0) alive &= ~(1u << side);
            } else if (stride[side] == 1) {
                alive &= ~(1u << side);
            } else {
                stride[side] >>= 1;
                doubling &= ~(1u << side);
            }
        }
    }
    uint32_t word = 0;
    for (int side = 0; side < 6; side++) word |= code[side] << (5 * side);
    return word;
}

// (desc_of == nullptr: one box record per descriptor, record i = descriptor i; else record i belongs to descriptor desc_of[i] -- the
// upper levels only, see launch_box_build_upper)
__global__ void box_grow_kernel(const uint64_t *__restrict__ descriptors, uint64_t n_records, uint64_t root_index, int n,
                                const uint64_t *__restrict__ pos, const uint64_t *__restrict__ desc_of, uint32_t *__restrict__ boxes) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t rec = t >> 3;
    const int k = (int)(t & 7u);
    if (rec >= n_records) return;
    const uint64_t idx = desc_of ? desc_of[rec] : rec;
    const uint64_t ps = idx == kPosNone ? kPosNone : pos[rec];
    uint32_t word = 0;
    if (ps != kPosNone) {
        const unsigned valid = (unsigned)(descriptors[idx] >> 16) & 0xffu;
        if (!(valid & (1u << k))) {
            const int level = (int)(ps >> kPosLevelShift);
            const int nx = (int)(ps & ((1u << kPosBits) - 1u)), ny = (int)((ps >> kPosBits) & ((1u << kPosBits) - 1u)), nz = (int)((ps >> (2 * kPosBits)) & ((1u << kPosBits) - 1u));
            // the entries of the descriptor's ancestors and its own, root first (the canonical descent toward the node)
            uint64_t path[kMaxLevels];
            path[0] = bx_entry(descriptors, root_index, descriptors[root_index]);
            for (int l = 0; l < level; l++) {
                const int bb = n - l - 1;
                const int i = ((nx >> bb) & 1) | (((ny >> bb) & 1) << 1) | (((nz >> bb) & 1) << 2);
                const uint64_t e = path[l];
                const uint64_t child = (e >> 16) + (uint64_t)(__popc((unsigned)e & 0xffu & ((2u << i) - 1u)) - 1);
                path[l + 1] = bx_entry(descriptors, child, descriptors[child]);
            }
            word = grow_box(descriptors, path, level, nx, ny, nz, k, n);
        }
    }
    boxes[t] = word;
}

// the coarse table's parallel word (see the header comment); same descent as coarse_build_kernel (raycast_jump_kernel.hip)
// (box_child == nullptr: box records are indexed by descriptor; else by the record ids of launch_box_build_upper, which exist for the
// descriptors of the levels below box_levels: a cell that resolves at a deeper level gets 0 -- the node itself, no widening)
__global__ void box_aux_kernel(const uint64_t *__restrict__ descriptors, uint64_t root_index, int n, int lc,
                               const uint32_t *__restrict__ boxes, const uint32_t *__restrict__ box_child, int box_levels, uint32_t *__restrict__ aux) {
    const uint64_t cell = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;     // x fastest
    if (cell >> (3 * lc)) return;
    const int sh = n - lc;
    const int x = (int)((cell & ((1u << lc) - 1u)) << sh), y = (int)(((cell >> lc) & ((1u << lc) - 1u)) << sh), z = (int)((cell >> (2 * lc)) << sh);
    uint64_t own = root_index, rec = box_child ? 0 : root_index;     // the descriptor, and its box record
    uint64_t cur = bx_entry(descriptors, root_index, descriptors[root_index]);
    int top = 0;
    uint32_t out = 0;
    for (;;) {
        if (top == lc) { out = top < box_levels ? (uint32_t)rec : 0u; break; }
        const int b = n - top - 1;
        const int i = ((x >> b) & 1) | (((y >> b) & 1) << 1) | (((z >> b) & 1) << 2);
        const unsigned masks = (unsigned)cur & 0xffffu, bit = 1u << i;
        if (!(masks & bit)) { out = top < box_levels ? boxes[rec * 8 + (uint64_t)i] : 0u; break; }   // empty above the table's level: its box
        if ((masks >> 8) & bit) { out = 0; break; }                             // solid leaf
        const unsigned rank = (unsigned)__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1u;
        own = (cur >> 16) + (uint64_t)rank;
        rec = box_child ? (top + 1 < box_levels ? (uint64_t)box_child[rec] + rank : 0) : own;
        cur = bx_entry(descriptors, own, descriptors[own]);
        top++;
    }
    aux[coarse_index((unsigned)(cell & ((1u << lc) - 1u)), (unsigned)((cell >> lc) & ((1u << lc) - 1u)), (unsigned)(cell >> (2 * lc)), lc)] = out;
}

// self-check: pseudo-random voxels inside the boxes must be empty in the tree (point query from the root)
__device__ __forceinline__ uint64_t bx_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__global__ void box_check_kernel(const uint64_t *__restrict__ descriptors, uint64_t n_desc, uint64_t root_index, int n,
                                 const uint64_t *__restrict__ pos, const uint64_t *__restrict__ desc_of, const uint32_t *__restrict__ boxes,
                                 uint64_t samples, uint64_t seed, unsigned long long *__restrict__ result) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= samples) return;
    // a (descriptor, child) pair per sample; pairs that are no empty child slot are skipped
    const uint64_t r0 = bx_mix(seed ^ t), r1 = bx_mix(r0), r2 = bx_mix(r1);
    const uint64_t rec = r0 % n_desc;                     // (n_desc: the number of box records)
    const int k = (int)(r1 & 7u);
    const uint64_t idx = desc_of ? desc_of[rec] : rec;
    if (idx == kPosNone) return;
    const uint64_t ps = pos[rec];
    if (ps == kPosNone) return;
    const unsigned valid = (unsigned)(descriptors[idx] >> 16) & 0xffu;
    if (valid & (1u << k)) return;
    const int level = (int)(ps >> kPosLevelShift), b = n - level - 1, s = 1 << b, dim = 1 << n;
    int lo[3], hi[3];
    lo[0] = (int)(ps & ((1u << kPosBits) - 1u)) + ((k & 1) ? s : 0);
    lo[1] = (int)((ps >> kPosBits) & ((1u << kPosBits) - 1u)) + ((k & 2) ? s : 0);
    lo[2] = (int)((ps >> (2 * kPosBits)) & ((1u << kPosBits) - 1u)) + ((k & 4) ? s : 0);
    const uint32_t w = boxes[rec * 8 + (uint64_t)k];
    for (int a = 0; a < 3; a++) {
        hi[a] = lo[a] + s + (box_decode((w >> (15 + 5 * a)) & 31u) << b);
        lo[a] -= box_decode((w >> (5 * a)) & 31u) << b;
        if (lo[a] < 0) lo[a] = 0;
        if (hi[a] > dim) hi[a] = dim;
    }
    atomicAdd(&result[0], 1ULL);
    // a voxel of the box, biased toward its faces (every other sample lies on one)
    int v[3];
    for (int a = 0; a < 3; a++) v[a] = lo[a] + (int)((r2 >> (20 * a)) % (uint64_t)(hi[a] - lo[a]));
    if (r1 & 8u) { const int a = (int)((r1 >> 4) % 3u); v[a] = (r1 & 64u) ? hi[a] - 1 : lo[a]; }
    uint64_t cur = bx_entry(descriptors, root_index, descriptors[root_index]);
    for (int top = 0;; top++) {
        const int bb = n - top - 1;
        const int i = ((v[0] >> bb) & 1) | (((v[1] >> bb) & 1) << 1) | (((v[2] >> bb) & 1) << 2);
        const unsigned masks = (unsigned)cur & 0xffffu, bit = 1u << i;
        if (!(masks & bit)) return;                       // empty: as promised
        if (((masks >> 8) & bit) || bb == 0) { atomicAdd(&result[1], 1ULL); return; }
        const uint64_t child = (cur >> 16) + (uint64_t)(__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1);
        cur = bx_entry(descriptors, child, descriptors[child]);
    }
}

// the same self-check for the words of the table's cells (the coarse space's boxes; the only ones a very large tree has)
__global__ void box_check_cells_kernel(const uint64_t *__restrict__ descriptors, uint64_t root_index, int n, int lc,
                                       const uint32_t *__restrict__ aux, uint64_t samples, uint64_t seed, unsigned long long *__restrict__ result) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= samples) return;
    const uint64_t r0 = bx_mix(seed ^ t), r1 = bx_mix(r0), r2 = bx_mix(r1);
    const uint64_t cell = r0 & ((1ULL << (3 * lc)) - 1ULL);
    const int sh = n - lc, dim = 1 << n;
    const unsigned ccx = (unsigned)(cell & ((1u << lc) - 1u)), ccy = (unsigned)((cell >> lc) & ((1u << lc) - 1u)), ccz = (unsigned)(cell >> (2 * lc));
    const int x = (int)(ccx << sh), y = (int)(ccy << sh), z = (int)(ccz << sh);
    uint64_t cur = bx_entry(descriptors, root_index, descriptors[root_index]);
    int b = 0;
    for (int top = 0;; top++) {
        if (top == lc) return;                            // the descent goes on below the table: no word for this cell
        b = n - top - 1;
        const int i = ((x >> b) & 1) | (((y >> b) & 1) << 1) | (((z >> b) & 1) << 2);
        const unsigned masks = (unsigned)cur & 0xffffu, bit = 1u << i;
        if (!(masks & bit)) break;                        // the empty node of 2^b voxels around the cell
        if ((masks >> 8) & bit) return;
        const uint64_t child = (cur >> 16) + (uint64_t)(__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1);
        cur = bx_entry(descriptors, child, descriptors[child]);
    }
    const uint32_t w = aux[coarse_index(ccx, ccy, ccz, lc)];
    const int nm = ~((1 << b) - 1);
    long long lo[3] = {x & nm, y & nm, z & nm}, hi[3];
    for (int a = 0; a < 3; a++) {
        hi[a] = lo[a] + (1LL << b) + ((long long)box_decode((w >> (15 + 5 * a)) & 31u) << b);
        lo[a] -= (long long)box_decode((w >> (5 * a)) & 31u) << b;
        if (lo[a] < 0) lo[a] = 0;
        if (hi[a] > dim) hi[a] = dim;
    }
    atomicAdd(&result[0], 1ULL);
    int v[3];
    for (int a = 0; a < 3; a++) v[a] = (int)(lo[a] + (long long)((r2 >> (20 * a)) % (uint64_t)(hi[a] - lo[a])));
    if (r1 & 8u) { const int a = (int)((r1 >> 4) % 3u); v[a] = (int)((r1 & 64u) ? hi[a] - 1 : lo[a]); }
    cur = bx_entry(descriptors, root_index, descriptors[root_index]);
    for (int top = 0;; top++) {
        const int bb = n - top - 1;
        const int i = ((v[0] >> bb) & 1) | (((v[1] >> bb) & 1) << 1) | (((v[2] >> bb) & 1) << 2);
        const unsigned masks = (unsigned)cur & 0xffffu, bit = 1u << i;
        if (!(masks & bit)) return;
        if (((masks >> 8) & bit) || bb == 0) { atomicAdd(&result[1], 1ULL); return; }
        const uint64_t child = (cur >> 16) + (uint64_t)(__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1);
        cur = bx_entry(descriptors, child, descriptors[child]);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Boxes for the UPPER LEVELS only (round 6): trees beyond ~2^29 descriptors cannot afford 32 bytes per descriptor -- and should not:
// a box word gathered from a 45 GB array misses the TLB as well as the caches (the depth-15 bench terrain, 1.4 G descriptors: 4.8 ms
// with all words against 3.75 without any).  A breadth-first sweep from the root numbers the descriptors of the levels 0 .. L-1 -- L the
// deepest level the record budget reaches -- and only they get box words: record r holds desc[r] (descriptor index), pos[r] and
// child[r], the record of the descriptor's first child (the children of one descriptor are consecutive, in the descriptor array's
// own order: child slot k's record is child[r] + popcount(valid below k), like the descriptor index itself).  The kernel carries the
// record id down its traversal stack in place of the descriptor index; below level L an empty node is widened over its empty
// siblings as in the box-less kernels.
// ---------------------------------------------------------------------------------------------------------------
__global__ void box_bfs_count_kernel(const uint64_t *__restrict__ descriptors, const uint64_t *__restrict__ desc_of, uint64_t first, uint64_t count,
                                     unsigned long long *__restrict__ total) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nv = 0;
    if (t < count) {
        const uint64_t idx = desc_of[first + t];
        if (idx != kPosNone) nv = (unsigned)__popc((unsigned)(descriptors[idx] >> 16) & 0xffu);
    }
    for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o);
    if ((threadIdx.x & 63) == 0 && nv) atomicAdd(total, (unsigned long long)nv);
}
__global__ void box_bfs_emit_kernel(const uint64_t *__restrict__ descriptors, int n, int level, uint64_t first, uint64_t count,
                                    uint64_t *__restrict__ desc_of, uint64_t *__restrict__ pos, uint32_t *__restrict__ child,
                                    unsigned long long *__restrict__ next_free) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint64_t rec = first + t, idx = desc_of[rec];
    child[rec] = 0;
    if (idx == kPosNone) return;
    const uint64_t e = bx_entry(descriptors, idx, descriptors[idx]);
    const unsigned valid = (unsigned)e & 0xffu, leaf = ((unsigned)e >> 8) & 0xffu;
    const unsigned nv = (unsigned)__popc(valid);
    if (!nv) return;
    const uint64_t base = atomicAdd(next_free, (unsigned long long)nv);
    child[rec] = (uint32_t)base;
    const uint64_t ps = pos[rec];
    const int half = 1 << (n - level - 1);
    const int x = (int)(ps & ((1u << kPosBits) - 1u)), y = (int)((ps >> kPosBits) & ((1u << kPosBits) - 1u)), z = (int)((ps >> (2 * kPosBits)) & ((1u << kPosBits) - 1u));
    unsigned rank = 0;
    for (int k = 0; k < 8; k++) {
        if (!(valid & (1u << k))) continue;
        const bool has_record = !(leaf & (1u << k)) && half > 1;   // a kept child with a descriptor of its own
        desc_of[base + rank] = has_record ? (e >> 16) + rank : kPosNone;
        pos[base + rank] = pack_pos(x + ((k & 1) ? half : 0), y + ((k & 2) ? half : 0), z + ((k & 4) ? half : 0), level + 1);
        rank++;
    }
}

}  // namespace

// pos[n_desc]: position and level of every descriptor the root reaches (kPosNone for far-pointer slots, page headers, unused slots)
hipError_t launch_box_positions(const uint64_t *descriptors, uint64_t n_desc, uint64_t root_index, int n, uint64_t *pos, hipStream_t stream) {
    (void)hipGetLastError();
    if (n < 1 || n > kPosBits || !n_desc || root_index >= n_desc) return hipErrorInvalidValue;
    const unsigned tb = 256;
    hipLaunchKernelGGL(box_fill_kernel, dim3((unsigned)((n_desc + tb - 1) / tb)), dim3(tb), 0, stream, pos, n_desc, kPosNone);
    hipError_t e = hipMemsetAsync(pos + root_index, 0, sizeof(uint64_t), stream);     // the root: position (0, 0, 0), level 0
    if (e != hipSuccess) return e;
    for (int level = 0; level < n - 1; level++)
        hipLaunchKernelGGL(box_positions_kernel, dim3((unsigned)((n_desc + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, n_desc, n, level, pos);
    return hipGetLastError();
}

// Builds boxes[8 * n_desc] (and aux[2^(3 lc)] when lc >= 1) on `stream`.  `pos_tmp` = n_desc uint64 of scratch.  n <= 19.
hipError_t launch_box_build(const uint64_t *descriptors, uint64_t n_desc, uint64_t root_index, int n, int lc, uint64_t *pos_tmp,
                            uint32_t *boxes, uint32_t *aux, hipStream_t stream) {
    const bool timing = getenv("VRC_BOX_TIMING") != nullptr;      // (diagnostics: seconds per phase on stderr)
    auto tick = [&](const char *what, std::chrono::steady_clock::time_point &t0) {
        if (!timing) return;
        (void)hipStreamSynchronize(stream);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[vrc boxes] %s: %.4f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    };
    if (timing) (void)hipStreamSynchronize(stream);
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = launch_box_positions(descriptors, n_desc, root_index, n, pos_tmp, stream);
    if (e != hipSuccess) return e;
    tick("positions", t0);
    const unsigned tb = 256;
    const uint64_t threads = n_desc * 8;
    {
        static const unsigned long long zero = 0;
        e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_box_queries_cut), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(box_grow_kernel, dim3((unsigned)((threads + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, n_desc, root_index, n, pos_tmp, (const uint64_t *)nullptr, boxes);
    tick("grow", t0);
    if (aux && lc >= 1) {
        const uint64_t cells = 1ULL << (3 * lc);
        hipLaunchKernelGGL(box_aux_kernel, dim3((unsigned)((cells + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, root_index, n, lc, boxes, (const uint32_t *)nullptr, n, aux);
        tick("table words", t0);
    }
    return hipGetLastError();
}

// What launch_box_build_upper leaves behind (device memory, the caller frees it): records 0 .. count-1 for the descriptors of the
// levels 0 .. levels-1.
struct BoxUpper { uint64_t *desc = nullptr, *pos = nullptr; uint32_t *child = nullptr, *boxes = nullptr; uint64_t count = 0; int levels = 0; };

// max_records: the budget (52 bytes of device memory each while building, 36 after the caller frees desc / pos -- or keeps them for
// vrc_empty_boxes_check); max_levels: 0 = as many as the budget reaches.  Synchronises the stream (one count per level comes back).
hipError_t launch_box_build_upper(const uint64_t *descriptors, uint64_t root_index, int n, int lc, uint64_t max_records, int max_levels,
                                  BoxUpper *out, uint32_t *aux, hipStream_t stream) {
    (void)hipGetLastError();
    if (n < 2 || n > kPosBits || !out || max_records < 9 || max_records > 0xffffff00ULL) return hipErrorInvalidValue;
    BoxUpper u;
    unsigned long long *d_ctr = nullptr;
    auto fail = [&](hipError_t e) {
        (void)hipGetLastError();
        if (u.desc) (void)hipFree(u.desc);
        if (u.pos) (void)hipFree(u.pos);
        if (u.child) (void)hipFree(u.child);
        if (u.boxes) (void)hipFree(u.boxes);
        if (d_ctr) (void)hipFree(d_ctr);
        return e;
    };
    const unsigned tb = 256;
    // the level sizes first (count passes only, over two ping-pong id lists would cost as much as the records themselves: the
    // records are allocated at the budget and trimmed by the count)
    hipError_t e = hipMalloc((void **)&u.desc, sizeof(uint64_t) * max_records);
    if (e == hipSuccess) e = hipMalloc((void **)&u.pos, sizeof(uint64_t) * max_records);
    if (e == hipSuccess) e = hipMalloc((void **)&u.child, sizeof(uint32_t) * max_records);
    if (e == hipSuccess) e = hipMalloc((void **)&d_ctr, sizeof(unsigned long long));
    if (e != hipSuccess) return fail(e);
    const uint64_t root_pos = 0;                                   // pack_pos(0, 0, 0, level 0)
    e = hipMemcpyAsync(u.desc, &root_index, sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(u.pos, &root_pos, sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);           // (the two sources are locals)
    if (e != hipSuccess) return fail(e);
    uint64_t first = 0, count = 1, total = 1;
    int levels = 1;
    for (int level = 0; level < n - 1; level++) {
        if (max_levels > 0 && levels >= max_levels) break;
        unsigned long long next = 0;
        e = hipMemsetAsync(d_ctr, 0, sizeof(unsigned long long), stream);
        if (e != hipSuccess) return fail(e);
        hipLaunchKernelGGL(box_bfs_count_kernel, dim3((unsigned)((count + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, u.desc, first, count, d_ctr);
        e = hipMemcpyAsync(&next, d_ctr, sizeof(next), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return fail(e);
        if (!next || total + next > max_records) break;
        const unsigned long long start = total;
        e = hipMemcpyAsync(d_ctr, &start, sizeof(start), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return fail(e);
        hipLaunchKernelGGL(box_bfs_emit_kernel, dim3((unsigned)((count + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, n, level, first, count,
                           u.desc, u.pos, u.child, d_ctr);
        e = hipStreamSynchronize(stream);                           // (`start` is a local; and the next count reads what this wrote)
        if (e != hipSuccess) return fail(e);
        first = total; count = next; total += next; levels = level + 2;
    }
    // the records of the last level have no children among the records
    e = hipMemsetAsync(u.child + first, 0, sizeof(uint32_t) * count, stream);
    if (e == hipSuccess) e = hipMalloc((void **)&u.boxes, sizeof(uint32_t) * 8 * total);
    if (e != hipSuccess) return fail(e);
    {
        static const unsigned long long zero = 0;
        e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_box_queries_cut), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return fail(e);
    }
    hipLaunchKernelGGL(box_grow_kernel, dim3((unsigned)((total * 8 + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, total, root_index, n, u.pos, u.desc, u.boxes);
    if (aux && lc >= 1) {
        const uint64_t cells = 1ULL << (3 * lc);
        hipLaunchKernelGGL(box_aux_kernel, dim3((unsigned)((cells + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, root_index, n, lc, u.boxes, u.child, levels, aux);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail(e);
    (void)hipFree(d_ctr);
    u.count = total; u.levels = levels;
    *out = u;
    return hipSuccess;
}

// region queries of the last launch_box_build on the current device that gave up at their budget (call after the stream is drained)
hipError_t box_queries_cut(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_box_queries_cut), sizeof(*out));
}

hipError_t launch_box_check_cells(const uint64_t *descriptors, uint64_t root_index, int n, int lc, const uint32_t *aux, uint64_t samples,
                                  uint64_t seed, unsigned long long *result, hipStream_t stream) {
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(result, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    if (!samples) return hipSuccess;
    const unsigned tb = 256;
    hipLaunchKernelGGL(box_check_cells_kernel, dim3((unsigned)((samples + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, root_index, n, lc, aux, samples, seed, result);
    return hipGetLastError();
}

// result[0] = boxes sampled, result[1] = sampled voxels the tree calls solid (must be 0); `result` is device memory, zeroed here
hipError_t launch_box_check(const uint64_t *descriptors, uint64_t n_desc, uint64_t root_index, int n, const uint64_t *pos, const uint64_t *desc_of,
                            const uint32_t *boxes, uint64_t samples, uint64_t seed, unsigned long long *result, hipStream_t stream) {
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(result, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    if (!samples) return hipSuccess;
    const unsigned tb = 256;
    hipLaunchKernelGGL(box_check_kernel, dim3((unsigned)((samples + tb - 1) / tb)), dim3(tb), 0, stream, descriptors, n_desc, root_index, n, pos, desc_of, boxes, samples, seed, result);
    return hipGetLastError();
}

}  // namespace vrc
