// svo_builder_gpu.hip -- builds the SVO of a column scene (the procedural shell terrain, or any caller-supplied height
// field: Map::GenerateHeightBitmap's diamond-square terrain, src/map/Map.cpp:144-262) directly in HBM.
//
// Replaces the limits of the reference's builder for large scenes: Octree::buffer_size = 100000 descriptors
// (include/map/Octree.h:29) and the dense char[D^3] input of Octree::Generate (src/map/Octree.cpp:13,325-327).
// The array produced is the reference's child-descriptor format (include/map/Octree.h:89-94) in the reference's
// bottom-up order (children before parents, kept siblings ascending, 15-bit relative pointers, far-pointer slots
// with absolute indices) -- the "brick" layout of svo_builder.cpp's Emitter(pages = false): identical to what the
// sequential host emitter writes, minus the page-header slots, whose position-dependence is the only thing that
// makes the reference layout inherently serial.  tests compare the two arrays bit for bit.
//
// MI355X-first design: nothing of the scene ever exists on the host.
//   1. height field h(x,y) and shell floor lo(x,y) as uint16 in HBM (one thread per column), min/max mip pyramid;
//   2. the map is cut into bricks of 2^kb voxels per side; a COUNT pass (one thread per candidate brick, iterative
//      post-order walk pruned by the pyramid) returns each brick's slot count and root masks;
//   3. the host runs the few levels above the bricks sequentially over those results (brick bases = a prefix sum in
//      the emitter's depth-first order, top-level child blocks and far pointers in between);
//   4. an EMIT pass repeats the walk and stores every descriptor at its final index; the top-level entries are
//      scattered in; an optional VALIDATE pass is the reference's Octree::Validate (src/map/Octree.cpp:329-352) on
//      the device: tree point queries against the procedural occupancy.
// Host memory: the brick table (tens of MB at depth 16); device memory: the array + 4/3 * 4 bytes per column.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/vrc.h"
#include "shell_scene.hpp"

namespace vrc {

namespace {

constexpr int kMaxDepth = 16;
constexpr int kBrickLog2 = 6;             // largest brick: 64^3 voxels (the walk's frame arrays); the builders use 32^3 / 16^3
constexpr int kMaxFrames = kBrickLog2;    // node sizes 2^kb .. 4
constexpr uint64_t kFarBit = 0x8000ULL, kLeafAll = 0xFF000000ULL, kValidAll = 0x00FF0000ULL;

struct Pyramid {
    const uint16_t *hi[kMaxDepth + 1];    // level l: max h over cells of 2^l x 2^l columns, (dim >> l)^2 entries
    const uint16_t *lo[kMaxDepth + 1];    // level l: min lo
    int depth;
};

struct BrickRef { uint16_t bx, by, bz, pad; };
struct BrickInfo { uint32_t masks; uint32_t size; };   // masks = bits 16..31 of the brick root's descriptor

__global__ void height_kernel(uint16_t *hi0, int depth, uint64_t seed, int octave_floor) {
    const int64_t dim = 1LL << depth;
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dim) return;
    hi0[x + dim * y] = (uint16_t)shell_height((uint32_t)depth, seed, octave_floor, x, y);
}

__global__ void floor_kernel(const uint16_t *hi0, uint16_t *lo0, int depth, int thickness) {
    const int64_t dim = 1LL << depth;
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dim) return;
    const int32_t h = hi0[x + dim * y];
    const int32_t hxm = x > 0 ? hi0[x - 1 + dim * y] : h, hxp = x + 1 < dim ? hi0[x + 1 + dim * y] : h;
    const int32_t hym = y > 0 ? hi0[x + dim * (y - 1)] : h, hyp = y + 1 < dim ? hi0[x + dim * (y + 1)] : h;
    lo0[x + dim * y] = (uint16_t)shell_floor(h, hxm, hxp, hym, hyp, thickness);
}

__global__ void mip_kernel(const uint16_t *hi_prev, const uint16_t *lo_prev, uint16_t *hi, uint16_t *lo, int64_t d) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= d) return;
    const int64_t pd = 2 * d, a = 2 * x + pd * (2 * y);
    const uint16_t h0 = hi_prev[a], h1 = hi_prev[a + 1], h2 = hi_prev[a + pd], h3 = hi_prev[a + pd + 1];
    const uint16_t l0 = lo_prev[a], l1 = lo_prev[a + 1], l2 = lo_prev[a + pd], l3 = lo_prev[a + pd + 1];
    hi[x + d * y] = max(max(h0, h1), max(h2, h3));
    lo[x + d * y] = min(min(l0, l1), min(l2, l3));
}

__device__ __forceinline__ bool is_empty_leaf_masks(uint32_t masks) { return (masks & 0xffu) == 0 && (masks >> 8) == 0xffu; }

// What a brick walk asks of the scene (the Source concept of svo_builder.cpp): is the cube of 2^l voxels at (x, y, z)
// certainly empty, and which of the 2^3 voxels at (x, y, z) are solid.
struct ColumnSrc {                            // columns solid for lo <= z <= hi, min / max pyramids over (x, y)
    Pyramid pyr;
    __device__ bool certainly_empty(int x, int y, int z, int size, int l) const {
        const int64_t dim = 1LL << pyr.depth;
        const int64_t at = (int64_t)(x >> l) + (dim >> l) * (int64_t)(y >> l);
        return z > (int)pyr.hi[l][at] || z + size - 1 < (int)pyr.lo[l][at];
    }
    __device__ uint32_t leaf_mask(int x, int y, int z) const {
        const int64_t dim = 1LL << pyr.depth;
        uint32_t m = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int64_t at = (int64_t)(x + (c & 1)) + dim * (int64_t)(y + (c >> 1));
            const int lo = pyr.lo[0][at], hi = pyr.hi[0][at];
            if (z >= lo && z <= hi) m |= 1u << c;
            if (z + 1 >= lo && z + 1 <= hi) m |= 1u << (c + 4);
        }
        return m;
    }
};
constexpr int kMaxGridDepth = 12;             // a dense grid of 4096^3 bytes is 64 GiB
struct GridSrc {                              // a dense grid (Octree::Generate's input, Octree.cpp:13-43) and its occupancy pyramid
    const uint8_t *occ[kMaxGridDepth + 1];    // [1]: bit i = voxel i of the 2^3 block (i = x | y<<1 | z<<2); [l >= 2]: non-zero = a voxel inside
    int depth;
    __device__ int64_t at(int x, int y, int z, int l) const {
        const int64_t d = (1LL << depth) >> l;
        return (int64_t)(x >> l) + d * ((int64_t)(y >> l) + d * (int64_t)(z >> l));
    }
    __device__ bool certainly_empty(int x, int y, int z, int, int l) const { return occ[l][at(x, y, z, l)] == 0; }
    __device__ uint32_t leaf_mask(int x, int y, int z) const { return occ[1][at(x, y, z, 1)]; }
};

// One candidate brick per thread: the emitter's recursion (svo_builder.cpp build<Source> + Emitter::place without
// pages) as an iterative post-order walk.  kEmit = false counts slots, kEmit = true stores them.
template <bool kEmit, class Src>
__global__ __launch_bounds__(64) void brick_kernel(const Src src, const BrickRef *__restrict__ bricks, uint32_t n_bricks, int kb,
                                                   BrickInfo *__restrict__ info, const uint64_t *__restrict__ bases,
                                                   uint64_t *__restrict__ desc, uint64_t total) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bricks) return;
    const BrickRef ref = bricks[b];
    const uint64_t base = kEmit ? bases[b] : 0;
    int32_t k = 0;                                         // brick-local reversed coordinate = slots pushed so far
    auto push = [&](uint64_t v) {
        if (kEmit) desc[total - 1 - (base + (uint64_t)k)] = v;
        k++;
    };
    auto certainly_empty = [&](int x, int y, int z, int size, int l) -> bool { return src.certainly_empty(x, y, z, size, l); };
    auto leaf_mask = [&](int x, int y, int z) -> uint32_t { return src.leaf_mask(x, y, z); };

    uint32_t kept_masks[kMaxFrames][8];
    int32_t kept_k[kMaxFrames][8];
    int fx[kMaxFrames], fy[kMaxFrames], fz[kMaxFrames], fi[kMaxFrames], fn[kMaxFrames];
    uint32_t fself[kMaxFrames];                            // bits 0-7 valid, 8-15 leaf of the node being assembled

    // Emitter::place (non-strict, no pages): far-pointer slots first, then the descriptors, high sibling to low
    auto place = [&](int f) -> int32_t {
        const int n = fn[f];
        const int32_t worst = 2 * n;
        int32_t far_slot[8];
        uint32_t is_far = 0;
        for (int i = n - 1; i >= 0; i--) {
            const int32_t rel = kept_k[f][i] < 0 ? -1 : (k + worst) - kept_k[f][i];
            if (rel > 0x7fff) {
                is_far |= 1u << i;
                far_slot[i] = k;
                push(total - 1 - (base + (uint64_t)kept_k[f][i]));          // absolute index of the child block
            }
        }
        for (int i = n - 1; i >= 0; i--) {
            const int32_t rel = kept_k[f][i] < 0 ? -1 : k - kept_k[f][i];
            uint64_t d = (uint64_t)kept_masks[f][i] << 16;
            if (is_far & (1u << i)) d |= kFarBit | (uint64_t)(k - far_slot[i]);
            else if (rel > 0) d |= (uint64_t)rel;
            push(d);
        }
        return k - 1;
    };

    int f = 0;
    fx[0] = (int)ref.bx << kb; fy[0] = (int)ref.by << kb; fz[0] = (int)ref.bz << kb;
    fi[0] = 0; fn[0] = 0; fself[0] = 0;
    uint32_t root_masks = 0xff00u;                         // an empty brick: all leaf, nothing valid
    bool done = false;
    if (certainly_empty(fx[0], fy[0], fz[0], 1 << kb, kb)) done = true;
    while (!done) {
        const int size = (1 << kb) >> f;
        if (fi[f] < 8) {
            const int i = fi[f]++;
            const int h = size >> 1;
            const int cx = fx[f] + ((i & 1) ? h : 0), cy = fy[f] + ((i & 2) ? h : 0), cz = fz[f] + ((i & 4) ? h : 0);
            if (certainly_empty(cx, cy, cz, h, kb - f - 1)) { fself[f] |= 1u << (i + 8); continue; }
            if (h == 2) {                                  // bottom-level descriptor: occupancy of 2^3 voxels, all leaf
                const uint32_t m = leaf_mask(cx, cy, cz);
                if (m == 0) { fself[f] |= 1u << (i + 8); continue; }
                fself[f] |= 1u << i;
                kept_masks[f][fn[f]] = m | 0xff00u;
                kept_k[f][fn[f]] = -1;
                fn[f]++;
                continue;
            }
            f++;
            fx[f] = cx; fy[f] = cy; fz[f] = cz; fi[f] = 0; fn[f] = 0; fself[f] = 0;
            continue;
        }
        // all eight children of frame f are known
        const uint32_t masks = fself[f];
        const int32_t block = place(f);
        if (f == 0) { root_masks = masks; done = true; break; }
        f--;
        const int i = fi[f] - 1;
        if (is_empty_leaf_masks(masks)) {
            fself[f] |= 1u << (i + 8);
        } else {
            fself[f] |= 1u << i;
            kept_masks[f][fn[f]] = masks;
            kept_k[f][fn[f]] = block;
            fn[f]++;
        }
    }
    if (!kEmit) info[b] = BrickInfo{root_masks, (uint32_t)k};
}

__global__ void scatter_kernel(const uint64_t *__restrict__ kv, uint64_t n, uint64_t *__restrict__ desc, uint64_t total) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) desc[total - 1 - kv[2 * i]] = kv[2 * i + 1];
}

// Octree::GetVoxel(position).found (src/map/Octree.cpp:45-158); the octant at each level is one bit of each coordinate.
// (A first, never committed version kept the reference's running 64-bit box corner and compared against it; inside the
// validate_grid_kernel of that day it returned run-to-run different answers and was replaced by this form.  Round 4 put both
// forms into a kernel of their own -- tools/repro/tree_walk_repro.hip: every voxel of nine static trees with page headers
// and far pointers, 25 runs each, -O3 and -O0, launches serialised -- and they agree with the grid, with each other and from
// run to run (profiles/r04_tree_walk_repro.txt).  The walk was not the cause; what was unstable then was its input, i.e. the
// order of the builder's passes in that unfinished state.  Both walks read only `desc`, written by kernels that precede them
// on the same stream with a stream synchronisation in between.)
__device__ __forceinline__ int tree_holds(const uint64_t *desc, uint64_t root_index, int depth, uint32_t x, uint32_t y, uint32_t z) {
    uint64_t index = root_index, d = desc[index];
    for (int l = depth - 1;; l--) {                            // l = log2 of the child's size
        const uint32_t i = ((x >> l) & 1u) | (((y >> l) & 1u) << 1) | (((z >> l) & 1u) << 2);
        if (!((d >> (16 + i)) & 1ULL)) return 0;
        if (((d >> (24 + i)) & 1ULL) || l == 0) return 1;
        const uint64_t at = index + (d & 0x7fffULL);
        const uint64_t first = (d & kFarBit) ? desc[at] : at;
        index = first + (uint64_t)(__popcll((d >> 16) & ((2ULL << i) - 1ULL)) - 1);
        d = desc[index];
    }
}

// Octree::Validate (src/map/Octree.cpp:329-352) on the device: GetVoxel(position).found against the procedural
// occupancy, on pseudo-random voxels -- half of them within a few voxels of the shell, half anywhere in the column
__global__ void validate_kernel(const Pyramid pyr, const uint64_t *__restrict__ desc, uint64_t root_index, uint64_t samples,
                                uint64_t seed, unsigned long long *mismatches) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= samples) return;
    const int64_t dim = 1LL << pyr.depth;
    const uint64_t r = splitmix64(seed ^ splitmix64(s));
    const int64_t x = (int64_t)(r & (uint64_t)(dim - 1)), y = (int64_t)((r >> 20) & (uint64_t)(dim - 1));
    const int lo = pyr.lo[0][x + dim * y], hi = pyr.hi[0][x + dim * y];
    const uint64_t r2 = splitmix64(r);
    int64_t z;
    if (s & 1) z = (int64_t)(r2 % (uint64_t)dim);
    else z = (int64_t)lo - 4 + (int64_t)(r2 % (uint64_t)(hi - lo + 9));
    z = z < 0 ? 0 : (z >= dim ? dim - 1 : z);
    const int expect = z >= lo && z <= hi;

    const int found = tree_holds(desc, root_index, pyr.depth, (uint32_t)x, (uint32_t)y, (uint32_t)z);
    if (found != expect) atomicAdd(mismatches, 1ULL);
}

__global__ void probe_kernel(const Pyramid pyr, const int32_t *xy, uint32_t n, int32_t *lohi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t dim = 1LL << pyr.depth;
    const int64_t at = (int64_t)xy[2 * i] + dim * (int64_t)xy[2 * i + 1];
    lohi[2 * i] = pyr.lo[0][at];
    lohi[2 * i + 1] = pyr.hi[0][at];
}

// ---- dense grids: occupancy pyramid.  Level 1 = the bottom-level valid mask of every 2^3 block (Octree.cpp:195-211: any
// non-zero voxel is solid), level l >= 2 = "a voxel inside this 2^l cube"
__global__ void occ1_kernel(const int8_t *__restrict__ grid, uint8_t *__restrict__ occ1, int64_t dim) {
    const int64_t d = dim >> 1;
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= d) return;
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int64_t at = (2 * x + (i & 1)) + dim * ((2 * y + ((i >> 1) & 1)) + dim * (2 * z + ((i >> 2) & 1)));
        if (grid[at]) m |= 1u << i;
    }
    occ1[x + d * (y + d * z)] = (uint8_t)m;
}
__global__ void occ_up_kernel(const uint8_t *__restrict__ prev, uint8_t *__restrict__ out, int64_t d) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= d) return;
    const int64_t pd = 2 * d;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) any |= prev[(2 * x + (i & 1)) + pd * ((2 * y + ((i >> 1) & 1)) + pd * (2 * z + ((i >> 2) & 1)))];
    out[x + d * (y + d * z)] = any ? 1 : 0;
}
// Octree::Validate against the grid itself: uniformly random voxels, and -- so that a sparse grid is probed where it
// holds something -- the first solid voxel within 64 steps along +x of a random one, with the voxel in front of it
__global__ void validate_grid_kernel(const int8_t *__restrict__ grid, int depth, const uint64_t *__restrict__ desc, uint64_t root_index,
                                     uint64_t samples, uint64_t seed, unsigned long long *mismatches) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= samples) return;
    const int64_t dim = 1LL << depth;
    const uint64_t r = splitmix64(seed ^ splitmix64(s));
    int64_t x = (int64_t)(r & (uint64_t)(dim - 1)), y = (int64_t)((r >> 20) & (uint64_t)(dim - 1)), z = (int64_t)((r >> 40) & (uint64_t)(dim - 1));
    if (s & 1) {
        for (int k = 0; k < 64 && x + 1 < dim && !grid[x + dim * (y + dim * z)]; k++) x++;
        if ((s & 2) && x > 0) x--;
    }
    const int expect = grid[x + dim * (y + dim * z)] != 0;
    const int found = tree_holds(desc, root_index, depth, (uint32_t)x, (uint32_t)y, (uint32_t)z);
    if (found != expect) atomicAdd(mismatches, 1ULL);
}

// ---- materials of a dense grid as attachments (include/vrc.h, vrc_assign_octree_attachments): one 8-byte slot per
// non-empty 2^3 block -- byte k = the grid's value at child k -- and lookup[index of the block's bottom-level descriptor] =
// its slot.  Slots are numbered in grid order (x fastest), slot 0 is the default (all 5): three passes so that the
// arrays are the same from run to run -- non-empty blocks per chunk of kAttachChunk, exclusive scan of the chunk counts,
// and the fill, which finds each block's descriptor by walking the finished tree.
constexpr int kAttachChunk = 4096;             // blocks per workgroup of 256 threads
__global__ void attach_count_kernel(const uint8_t *__restrict__ occ1, uint64_t n_blocks, uint32_t *__restrict__ chunk_count) {
    __shared__ uint32_t acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kAttachChunk;
    uint32_t mine = 0;
    for (int k = threadIdx.x; k < kAttachChunk; k += 256)
        if (base + k < n_blocks && occ1[base + k]) mine++;
    if (mine) atomicAdd(&acc, mine);
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = acc;
}
// one workgroup: chunk_count -> exclusive prefix sums in place, the total in *total
__global__ void attach_scan_kernel(uint32_t *chunk_count, uint64_t n_chunks, unsigned long long *total) {
    __shared__ unsigned long long part[1024];
    const uint64_t per = (n_chunks + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n_chunks ? lo + per : n_chunks;
    unsigned long long sum = 0;
    for (uint64_t i = lo; i < hi; i++) sum += chunk_count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) { const unsigned long long v = part[i]; part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint64_t i = lo; i < hi; i++) { const uint32_t v = chunk_count[i]; chunk_count[i] = (uint32_t)run; run += v; }
}
// index of the bottom-level descriptor (node of 2^3 voxels) that holds voxel (x, y, z); the block is known to be non-empty
__device__ __forceinline__ uint64_t tree_block_descriptor(const uint64_t *desc, uint64_t root_index, int depth, uint32_t x, uint32_t y, uint32_t z) {
    uint64_t index = root_index, d = desc[index];
    for (int l = depth - 1; l >= 1; l--) {
        const uint32_t i = ((x >> l) & 1u) | (((y >> l) & 1u) << 1) | (((z >> l) & 1u) << 2);
        const uint64_t at = index + (d & 0x7fffULL);
        const uint64_t first = (d & kFarBit) ? desc[at] : at;
        index = first + (uint64_t)(__popcll((d >> 16) & ((2ULL << i) - 1ULL)) - 1);
        d = desc[index];
    }
    return index;
}
__global__ void attach_fill_kernel(const int8_t *__restrict__ grid, const uint8_t *__restrict__ occ1, int depth, const uint64_t *__restrict__ desc,
                                   uint64_t root_index, const uint32_t *__restrict__ chunk_base, uint32_t *__restrict__ lookup,
                                   uint64_t *__restrict__ attach) {
    __shared__ uint32_t rank[256];
    const int64_t dim = 1LL << depth, d = dim >> 1;
    const uint64_t n_blocks = (uint64_t)d * d * d, base = (uint64_t)blockIdx.x * kAttachChunk;
    constexpr int kPer = kAttachChunk / 256;                       // consecutive blocks per thread
    const uint64_t first = base + (uint64_t)threadIdx.x * kPer;
    uint32_t mine = 0;
    for (int k = 0; k < kPer; k++) mine += (first + k < n_blocks && occ1[first + k]) ? 1u : 0u;
    rank[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; i++) { const uint32_t v = rank[i]; rank[i] = run; run += v; }
    }
    __syncthreads();
    uint32_t slot = 1u + chunk_base[blockIdx.x] + rank[threadIdx.x];
    for (int k = 0; k < kPer; k++) {
        const uint64_t b = first + k;
        if (b >= n_blocks || !occ1[b]) continue;
        const int64_t bx = (int64_t)(b % (uint64_t)d), by = (int64_t)((b / (uint64_t)d) % (uint64_t)d), bz = (int64_t)(b / (uint64_t)(d * d));
        uint64_t packed = 0;
#pragma unroll
        for (int c = 0; c < 8; c++)
            packed |= (uint64_t)(uint8_t)grid[(2 * bx + (c & 1)) + dim * ((2 * by + ((c >> 1) & 1)) + dim * (2 * bz + ((c >> 2) & 1)))] << (8 * c);
        attach[slot] = packed;
        lookup[tree_block_descriptor(desc, root_index, depth, (uint32_t)(2 * bx), (uint32_t)(2 * by), (uint32_t)(2 * bz))] = slot;
        slot++;
    }
}

// ---- the levels above the bricks, on the host (sequential, a few million nodes at most)
struct TopNode { uint64_t desc = 0; int64_t k = -1; };

struct TopEmitter {
    int64_t k = 0;
    std::vector<uint64_t> kv;                  // (reversed coordinate, value) pairs of the top-level slots
    std::vector<size_t> far_fixups;            // kv value slots holding a reversed coordinate to turn into an index
    uint64_t n_far = 0;
    void push(uint64_t v) { kv.push_back((uint64_t)k); kv.push_back(v); k++; }
    int64_t place(const TopNode *kept, int n) {                // Emitter::place, non-strict, no pages
        const int worst = 2 * n;
        int64_t far_slot[8];
        bool is_far[8];
        for (int i = n - 1; i >= 0; i--) {
            const int64_t rel = kept[i].k < 0 ? -1 : (k + worst) - kept[i].k;
            is_far[i] = rel > 0x7fff;
            if (is_far[i]) {
                far_slot[i] = k;
                far_fixups.push_back(kv.size() + 1);
                push((uint64_t)kept[i].k);
                n_far++;
            }
        }
        for (int i = n - 1; i >= 0; i--) {
            const int64_t rel = kept[i].k < 0 ? -1 : k - kept[i].k;
            uint64_t d = kept[i].desc;
            if (is_far[i]) d |= kFarBit | (uint64_t)(k - far_slot[i]);
            else if (rel > 0) d |= (uint64_t)rel;
            push(d);
        }
        return k - 1;
    }
};

struct TopContext {
    int kb;
    const std::vector<BrickInfo> *info;
    std::vector<uint64_t> *bases;
    TopEmitter *em;
    std::function<bool(int, int, int, int, int)> region_empty;   // (x, y, z, size, level): certainly no voxel inside
    std::function<int64_t(int, int, int)> brick_at;              // candidate brick at voxel (x, y, z), -1: none (empty)
};

TopNode top_build(TopContext &c, int x, int y, int z, int size, int l) {
    TopNode self;
    if (l == c.kb) {                            // a brick: result of the count pass
        const int64_t b = c.brick_at(x, y, z);
        if (b < 0) { self.desc = kLeafAll; return self; }
        const BrickInfo &bi = (*c.info)[(size_t)b];
        self.desc = (uint64_t)bi.masks << 16;
        (*c.bases)[(size_t)b] = (uint64_t)c.em->k;
        c.em->k += bi.size;
        self.k = bi.size ? c.em->k - 1 : -1;
        return self;
    }
    const int h = size / 2;
    TopNode kept[8];
    int n = 0;
    for (int i = 0; i < 8; i++) {
        const int cx = x + ((i & 1) ? h : 0), cy = y + ((i & 2) ? h : 0), cz = z + ((i & 4) ? h : 0);
        TopNode child;
        if (c.region_empty(cx, cy, cz, h, l - 1)) child.desc = kLeafAll;
        else child = top_build(c, cx, cy, cz, h, l - 1);
        if ((child.desc & kValidAll) == 0 && (child.desc & kLeafAll) == kLeafAll) {
            self.desc |= 1ULL << (i + 24);
        } else {
            self.desc |= 1ULL << (i + 16);
            kept[n++] = child;
        }
    }
    self.k = c.em->place(kept, n);
    return self;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

#define GB_TRY(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            error = std::string(#call) + " failed: " + hipGetErrorString(e_);                                 \
            (void)hipGetLastError();              /* reported here, not left behind */                        \
            rc = e_ == hipErrorOutOfMemory ? VRC_ERR_OUT_OF_MEMORY : VRC_ERR_DEVICE;                          \
            goto cleanup;                                                                                     \
        }                                                                                                     \
    } while (0)

// Builds the tree on the current device.  On success *d_desc owns the array (hipFree by the caller).
// The scene is a field of columns, column (x,y) solid for lo <= z <= hi: either the procedural shell terrain (host_hi ==
// nullptr: seed / thickness / octave_floor, evaluated on the device) or caller-supplied uint16[dim*dim] arrays
// (host_lo == nullptr: solid from z = 0 up) -- the 2-D field is the only thing that crosses PCIe.
int build_columns_device(hipStream_t stream, uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor,
                         const uint16_t *host_hi, const uint16_t *host_lo,
                         uint32_t flags, uint64_t validate_samples, const int32_t *probe_xy, uint32_t n_probe,
                         int32_t *probe_lohi, uint64_t **d_desc, vrc_build_info *out, std::string &error) {
    int rc = VRC_OK;
    vrc_build_info bi;
    memset(&bi, 0, sizeof(bi));
    (void)hipGetLastError();                 // the launch checks below must not pick up an error an earlier call left behind
    const int64_t dim = 1LL << depth;
    // bricks of 32^3 voxels (one walk = one thread): measured 64^3 / 32^3 / 16^3 on the 198 GB scene of BASELINE configs[4]
    // (depth 16, thickness 33): 12.1 / 5.8 / 5.6 s with 0.19 / 0.79 / 3.9 GB of host tables; depth 12: 43 / 16 / 13 ms
    static const int column_brick_log2 = getenv("VRC_COLUMN_BRICK_LOG2") ? atoi(getenv("VRC_COLUMN_BRICK_LOG2")) : 5;
    const int kb = std::min<int>(std::min(std::max(column_brick_log2, 2), kBrickLog2), (int)depth);
    const int64_t cells = dim >> kb;
    uint16_t *d_hi = nullptr, *d_lo = nullptr;
    BrickRef *d_bricks = nullptr;
    BrickInfo *d_info = nullptr;
    uint64_t *d_bases = nullptr, *d_kv = nullptr, *desc = nullptr;
    unsigned long long *d_mis = nullptr;
    int32_t *d_pxy = nullptr, *d_plohi = nullptr;
    Pyramid pyr;
    memset(&pyr, 0, sizeof(pyr));
    pyr.depth = (int)depth;
    std::vector<std::vector<uint16_t>> h_hi, h_lo;
    std::vector<int32_t> cell_z0, cell_n;
    std::vector<uint32_t> cell_first;
    std::vector<BrickRef> bricks;
    std::vector<BrickInfo> info;
    std::vector<uint64_t> bases;
    TopEmitter em;
    uint64_t total = 0, temp_bytes = 0;
    const double t0 = now_s();
    double t1 = t0, t2 = t0, t3 = t0;
    *d_desc = nullptr;

    {
        // 1. height field, shell floor, pyramid
        size_t entries = 0;
        std::vector<size_t> off(depth + 1);
        for (uint32_t l = 0; l <= depth; l++) { off[l] = entries; entries += (size_t)(dim >> l) * (size_t)(dim >> l); }
        GB_TRY(hipMalloc((void **)&d_hi, entries * sizeof(uint16_t)));
        GB_TRY(hipMalloc((void **)&d_lo, entries * sizeof(uint16_t)));
        temp_bytes = 2 * entries * sizeof(uint16_t);
        for (uint32_t l = 0; l <= depth; l++) { pyr.hi[l] = d_hi + off[l]; pyr.lo[l] = d_lo + off[l]; }
        const dim3 tb(256), grid0((unsigned)((dim + 255) / 256), (unsigned)dim);
        if (host_hi) {
            GB_TRY(hipMemcpyAsync(d_hi, host_hi, (size_t)(dim * dim) * sizeof(uint16_t), hipMemcpyHostToDevice, stream));
            if (host_lo) GB_TRY(hipMemcpyAsync(d_lo, host_lo, (size_t)(dim * dim) * sizeof(uint16_t), hipMemcpyHostToDevice, stream));
            else GB_TRY(hipMemsetAsync(d_lo, 0, (size_t)(dim * dim) * sizeof(uint16_t), stream));
        } else {
            hipLaunchKernelGGL(height_kernel, grid0, tb, 0, stream, d_hi, (int)depth, seed, (int)octave_floor);
            hipLaunchKernelGGL(floor_kernel, grid0, tb, 0, stream, (const uint16_t *)d_hi, d_lo, (int)depth, (int)thickness);
        }
        for (uint32_t l = 1; l <= depth; l++) {
            const int64_t d = dim >> l;
            hipLaunchKernelGGL(mip_kernel, dim3((unsigned)((d + 255) / 256), (unsigned)d), tb, 0, stream, pyr.hi[l - 1], pyr.lo[l - 1],
                               d_hi + off[l], d_lo + off[l], d);
        }
        GB_TRY(hipGetLastError());
        // host copies of the levels at and above the bricks
        h_hi.resize(depth - kb + 1); h_lo.resize(depth - kb + 1);
        for (uint32_t l = (uint32_t)kb; l <= depth; l++) {
            const size_t n = (size_t)(dim >> l) * (size_t)(dim >> l);
            h_hi[l - kb].resize(n); h_lo[l - kb].resize(n);
            GB_TRY(hipMemcpyAsync(h_hi[l - kb].data(), pyr.hi[l], n * 2, hipMemcpyDeviceToHost, stream));
            GB_TRY(hipMemcpyAsync(h_lo[l - kb].data(), pyr.lo[l], n * 2, hipMemcpyDeviceToHost, stream));
        }
        GB_TRY(hipStreamSynchronize(stream));
        t1 = now_s();

        // 2. candidate bricks per column of bricks, count pass
        cell_z0.resize((size_t)(cells * cells)); cell_n.resize(cell_z0.size()); cell_first.resize(cell_z0.size());
        for (size_t c = 0; c < cell_z0.size(); c++) {
            const int z0 = (int)h_lo[0][c] >> kb, z1 = (int)h_hi[0][c] >> kb;
            cell_z0[c] = z0; cell_n[c] = z1 - z0 + 1; cell_first[c] = (uint32_t)bricks.size();
            for (int z = z0; z <= z1; z++) bricks.push_back(BrickRef{(uint16_t)(c % (size_t)cells), (uint16_t)(c / (size_t)cells), (uint16_t)z, 0});
        }
        const uint32_t nb = (uint32_t)bricks.size();
        bi.n_bricks = nb;
        GB_TRY(hipMalloc((void **)&d_bricks, (size_t)nb * sizeof(BrickRef)));
        GB_TRY(hipMalloc((void **)&d_info, (size_t)nb * sizeof(BrickInfo)));
        GB_TRY(hipMalloc((void **)&d_bases, (size_t)nb * sizeof(uint64_t)));
        temp_bytes += (size_t)nb * (sizeof(BrickRef) + sizeof(BrickInfo) + sizeof(uint64_t));
        GB_TRY(hipMemcpyAsync(d_bricks, bricks.data(), (size_t)nb * sizeof(BrickRef), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL((brick_kernel<false, ColumnSrc>), dim3((nb + 63) / 64), dim3(64), 0, stream, ColumnSrc{pyr}, (const BrickRef *)d_bricks, nb, kb, d_info,
                           (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0);
        GB_TRY(hipGetLastError());
        info.resize(nb);
        GB_TRY(hipMemcpyAsync(info.data(), d_info, (size_t)nb * sizeof(BrickInfo), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipStreamSynchronize(stream));
        t2 = now_s();

        // 3. the levels above the bricks: brick bases in emission order + the top-level slots
        bases.assign(nb, 0);
        TopContext ctx{kb, &info, &bases, &em,
                       [&](int cx, int cy, int cz, int h, int cl) {
                           const size_t at = (size_t)(cx >> cl) + (size_t)(dim >> cl) * (size_t)(cy >> cl);
                           return cz > (int)h_hi[cl - kb][at] || cz + h - 1 < (int)h_lo[cl - kb][at];
                       },
                       [&](int x, int y, int z) -> int64_t {
                           const size_t cell = (size_t)(x >> kb) + (size_t)cells * (size_t)(y >> kb);
                           const int bz = z >> kb, z0 = cell_z0[cell];
                           if (bz < z0 || bz >= z0 + cell_n[cell]) return -1;
                           return (int64_t)cell_first[cell] + (bz - z0);
                       }};
        const TopNode root = top_build(ctx, 0, 0, 0, (int)dim, (int)depth);
        em.push(root.desc | 1);                               // Octree.cpp:27-31
        total = (uint64_t)em.k;
        for (size_t at : em.far_fixups) em.kv[at] = total - 1 - em.kv[at];
        bi.n_descriptors = total;
        bi.root_index = 0;
        bi.n_top_slots = em.kv.size() / 2;
        bi.n_far_pointers_top = em.n_far;
        if (flags & VRC_BUILD_COUNT_ONLY) goto finish;

        // 4. emit
        GB_TRY(hipMalloc((void **)&desc, total * sizeof(uint64_t)));
        GB_TRY(hipMalloc((void **)&d_kv, em.kv.size() * sizeof(uint64_t)));
        temp_bytes += em.kv.size() * sizeof(uint64_t);
        GB_TRY(hipMemcpyAsync(d_bases, bases.data(), (size_t)nb * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        GB_TRY(hipMemcpyAsync(d_kv, em.kv.data(), em.kv.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL((brick_kernel<true, ColumnSrc>), dim3((nb + 63) / 64), dim3(64), 0, stream, ColumnSrc{pyr}, (const BrickRef *)d_bricks, nb, kb, d_info,
                           (const uint64_t *)d_bases, desc, total);
        {
            const uint64_t n = em.kv.size() / 2;
            hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint64_t *)d_kv, n, desc, total);
        }
        GB_TRY(hipGetLastError());
        GB_TRY(hipStreamSynchronize(stream));
        t3 = now_s();

        if (validate_samples) {
            GB_TRY(hipMalloc((void **)&d_mis, sizeof(unsigned long long)));
            GB_TRY(hipMemsetAsync(d_mis, 0, sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(validate_kernel, dim3((unsigned)((validate_samples + 255) / 256)), dim3(256), 0, stream, pyr,
                               (const uint64_t *)desc, (uint64_t)0, validate_samples, seed ^ 0x5eedULL, d_mis);
            GB_TRY(hipGetLastError());
            unsigned long long mis = 0;
            GB_TRY(hipMemcpyAsync(&mis, d_mis, sizeof(mis), hipMemcpyDeviceToHost, stream));
            GB_TRY(hipStreamSynchronize(stream));
            bi.validate_samples = validate_samples;
            bi.validate_mismatches = mis;
        }
    }
finish:
    if (n_probe && probe_xy && probe_lohi) {
        GB_TRY(hipMalloc((void **)&d_pxy, (size_t)n_probe * 2 * sizeof(int32_t)));
        GB_TRY(hipMalloc((void **)&d_plohi, (size_t)n_probe * 2 * sizeof(int32_t)));
        GB_TRY(hipMemcpyAsync(d_pxy, probe_xy, (size_t)n_probe * 2 * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(probe_kernel, dim3((n_probe + 255) / 256), dim3(256), 0, stream, pyr, (const int32_t *)d_pxy, n_probe, d_plohi);
        GB_TRY(hipGetLastError());
        GB_TRY(hipMemcpyAsync(probe_lohi, d_plohi, (size_t)n_probe * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipStreamSynchronize(stream));
    }
    bi.seconds_height = t1 - t0;
    bi.seconds_count = t2 - t1;
    bi.seconds_emit = (flags & VRC_BUILD_COUNT_ONLY) ? 0.0 : t3 - t2;
    bi.seconds_total = now_s() - t0;
    bi.device_bytes_peak = temp_bytes + total * sizeof(uint64_t) * ((flags & VRC_BUILD_COUNT_ONLY) ? 0 : 1);
    bi.host_bytes = bricks.size() * (sizeof(BrickRef) + sizeof(BrickInfo) + sizeof(uint64_t)) + em.kv.size() * sizeof(uint64_t) +
                    cell_z0.size() * 12;
    *d_desc = desc;
    desc = nullptr;

cleanup:
    if (out) *out = bi;
    (void)hipFree(d_hi); (void)hipFree(d_lo); (void)hipFree(d_bricks); (void)hipFree(d_info); (void)hipFree(d_bases);
    (void)hipFree(d_kv); (void)hipFree(d_mis); (void)hipFree(d_pxy); (void)hipFree(d_plohi);
    if (desc) (void)hipFree(desc);
    return rc;
}

// The same pipeline for Octree::Generate's own input (src/map/Octree.cpp:13-43: a dense grid, x + dim * (y + dim * z), any
// non-zero voxel solid): the grid goes to the device once, an occupancy pyramid replaces the min / max pyramids, every
// 64^3 brick that holds a voxel is a candidate.  On success *d_desc owns the array.
// resident_grid: the same grid already in this device's memory (the map of the array branch); then nothing is uploaded.
// VRC_BUILD_ATTACHMENTS: the grid's values as material attachments, *d_lookup (uint32 per descriptor) and *d_attach
// (*n_attach slots of 8 bytes), owned by the caller like *d_desc.
int build_grid_device(hipStream_t stream, uint32_t depth, const int8_t *host_grid, const int8_t *resident_grid, uint32_t flags,
                      uint64_t validate_samples, uint64_t **d_desc, uint32_t **d_lookup, uint64_t **d_attach, uint64_t *n_attach,
                      vrc_build_info *out, std::string &error) {
    int rc = VRC_OK;
    vrc_build_info bi;
    memset(&bi, 0, sizeof(bi));
    (void)hipGetLastError();
    const int64_t dim = 1LL << depth;
    // bricks of 16^3 voxels here: a dense grid is at most 4096^3, so the host's top levels stay small (256^3 cells), and a
    // brick walk -- one thread -- is at most 585 descriptors instead of 37 449 (64^3: 0.05 s per pass whatever the grid)
    static const int grid_brick_log2 = getenv("VRC_GRID_BRICK_LOG2") ? atoi(getenv("VRC_GRID_BRICK_LOG2")) : 4;
    const int kb = std::min<int>(std::min(std::max(grid_brick_log2, 2), kBrickLog2), (int)depth);
    const int64_t cells = dim >> kb;
    int8_t *d_grid = nullptr;
    uint8_t *d_occ = nullptr;
    BrickRef *d_bricks = nullptr;
    BrickInfo *d_info = nullptr;
    uint64_t *d_bases = nullptr, *d_kv = nullptr, *desc = nullptr;
    unsigned long long *d_mis = nullptr, *d_total = nullptr;
    uint32_t *lookup = nullptr, *d_chunks = nullptr;
    uint64_t *attach = nullptr;
    GridSrc src;
    memset(&src, 0, sizeof(src));
    src.depth = (int)depth;
    std::vector<std::vector<uint8_t>> h_occ;            // host copies of levels kb..depth (index l - kb)
    std::vector<int32_t> brick_index;
    std::vector<BrickRef> bricks;
    std::vector<BrickInfo> info;
    std::vector<uint64_t> bases;
    TopEmitter em;
    uint64_t total = 0, temp_bytes = 0;
    const double t0 = now_s();
    double t1 = t0, t2 = t0, t3 = t0;
    *d_desc = nullptr;
    if (d_lookup) *d_lookup = nullptr;
    if (d_attach) *d_attach = nullptr;
    if (n_attach) *n_attach = 0;

    {
        // 1. the grid and its occupancy pyramid
        const size_t voxels = (size_t)dim * (size_t)dim * (size_t)dim;
        size_t entries = 0;
        std::vector<size_t> off(depth + 1, 0);
        for (uint32_t l = 1; l <= depth; l++) { off[l] = entries; const size_t d = (size_t)(dim >> l); entries += d * d * d; }
        if (!resident_grid) {
            GB_TRY(hipMalloc((void **)&d_grid, voxels));
            GB_TRY(hipMemcpyAsync(d_grid, host_grid, voxels, hipMemcpyHostToDevice, stream));
            resident_grid = d_grid;
        }
        GB_TRY(hipMalloc((void **)&d_occ, entries));
        temp_bytes = (d_grid ? voxels : 0) + entries;
        src.occ[0] = (const uint8_t *)resident_grid;
        for (uint32_t l = 1; l <= depth; l++) src.occ[l] = d_occ + off[l];
        {
            const int64_t d = dim >> 1;
            hipLaunchKernelGGL(occ1_kernel, dim3((unsigned)((d + 255) / 256), (unsigned)d, (unsigned)d), dim3(256), 0, stream,
                               resident_grid, d_occ + off[1], dim);
        }
        for (uint32_t l = 2; l <= depth; l++) {
            const int64_t d = dim >> l;
            hipLaunchKernelGGL(occ_up_kernel, dim3((unsigned)((d + 255) / 256), (unsigned)d, (unsigned)d), dim3(256), 0, stream,
                               (const uint8_t *)(d_occ + off[l - 1]), d_occ + off[l], d);
        }
        GB_TRY(hipGetLastError());
        h_occ.resize(depth - kb + 1);
        for (uint32_t l = (uint32_t)kb; l <= depth; l++) {
            const size_t d = (size_t)(dim >> l), n = d * d * d;
            h_occ[l - kb].resize(n);
            GB_TRY(hipMemcpyAsync(h_occ[l - kb].data(), src.occ[l], n, hipMemcpyDeviceToHost, stream));
        }
        GB_TRY(hipStreamSynchronize(stream));
        t1 = now_s();

        // 2. candidate bricks (in x, y, z order), count pass
        brick_index.assign((size_t)(cells * cells * cells), -1);
        for (size_t c = 0; c < brick_index.size(); c++)
            if (h_occ[0][c]) {
                brick_index[c] = (int32_t)bricks.size();
                bricks.push_back(BrickRef{(uint16_t)(c % (size_t)cells), (uint16_t)((c / (size_t)cells) % (size_t)cells),
                                          (uint16_t)(c / (size_t)(cells * cells)), 0});
            }
        const uint32_t nb = (uint32_t)bricks.size();
        bi.n_bricks = nb;
        if (nb) {
            GB_TRY(hipMalloc((void **)&d_bricks, (size_t)nb * sizeof(BrickRef)));
            GB_TRY(hipMalloc((void **)&d_info, (size_t)nb * sizeof(BrickInfo)));
            GB_TRY(hipMalloc((void **)&d_bases, (size_t)nb * sizeof(uint64_t)));
            temp_bytes += (size_t)nb * (sizeof(BrickRef) + sizeof(BrickInfo) + sizeof(uint64_t));
            GB_TRY(hipMemcpyAsync(d_bricks, bricks.data(), (size_t)nb * sizeof(BrickRef), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL((brick_kernel<false, GridSrc>), dim3((nb + 63) / 64), dim3(64), 0, stream, src, (const BrickRef *)d_bricks, nb, kb,
                               d_info, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0);
            GB_TRY(hipGetLastError());
            info.resize(nb);
            GB_TRY(hipMemcpyAsync(info.data(), d_info, (size_t)nb * sizeof(BrickInfo), hipMemcpyDeviceToHost, stream));
            GB_TRY(hipStreamSynchronize(stream));
        }
        t2 = now_s();

        // 3. the levels above the bricks
        bases.assign(nb, 0);
        TopContext ctx{kb, &info, &bases, &em,
                       [&](int cx, int cy, int cz, int, int cl) {
                           const size_t d = (size_t)(dim >> cl);
                           return h_occ[cl - kb][(size_t)(cx >> cl) + d * ((size_t)(cy >> cl) + d * (size_t)(cz >> cl))] == 0;
                       },
                       [&](int x, int y, int z) -> int64_t {
                           return brick_index[(size_t)(x >> kb) + (size_t)cells * ((size_t)(y >> kb) + (size_t)cells * (size_t)(z >> kb))];
                       }};
        const TopNode root = top_build(ctx, 0, 0, 0, (int)dim, (int)depth);
        em.push(root.desc | 1);                               // Octree.cpp:27-31
        total = (uint64_t)em.k;
        for (size_t at : em.far_fixups) em.kv[at] = total - 1 - em.kv[at];
        bi.n_descriptors = total;
        bi.root_index = 0;
        bi.n_top_slots = em.kv.size() / 2;
        bi.n_far_pointers_top = em.n_far;
        if (flags & VRC_BUILD_COUNT_ONLY) goto finish;

        // 4. emit
        GB_TRY(hipMalloc((void **)&desc, total * sizeof(uint64_t)));
        GB_TRY(hipMalloc((void **)&d_kv, em.kv.size() * sizeof(uint64_t)));
        temp_bytes += em.kv.size() * sizeof(uint64_t);
        if (nb) {
            GB_TRY(hipMemcpyAsync(d_bases, bases.data(), (size_t)nb * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL((brick_kernel<true, GridSrc>), dim3((nb + 63) / 64), dim3(64), 0, stream, src, (const BrickRef *)d_bricks, nb, kb,
                               d_info, (const uint64_t *)d_bases, desc, total);
        }
        GB_TRY(hipMemcpyAsync(d_kv, em.kv.data(), em.kv.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        {
            const uint64_t n = em.kv.size() / 2;
            hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint64_t *)d_kv, n, desc, total);
        }
        GB_TRY(hipGetLastError());
        GB_TRY(hipStreamSynchronize(stream));
        t3 = now_s();

        if ((flags & VRC_BUILD_ATTACHMENTS) && d_lookup && d_attach && n_attach) {
            // 5. materials: one slot per non-empty 2^3 block, in grid order
            const uint64_t half = (uint64_t)(dim >> 1), n_blocks = half * half * half, n_chunks = (n_blocks + kAttachChunk - 1) / kAttachChunk;
            unsigned long long slots = 0;
            GB_TRY(hipMalloc((void **)&d_chunks, n_chunks * sizeof(uint32_t)));
            GB_TRY(hipMalloc((void **)&d_total, sizeof(unsigned long long)));
            hipLaunchKernelGGL(attach_count_kernel, dim3((unsigned)n_chunks), dim3(256), 0, stream, src.occ[1], n_blocks, d_chunks);
            hipLaunchKernelGGL(attach_scan_kernel, dim3(1), dim3(1024), 0, stream, d_chunks, n_chunks, d_total);
            GB_TRY(hipGetLastError());
            GB_TRY(hipMemcpyAsync(&slots, d_total, sizeof(slots), hipMemcpyDeviceToHost, stream));
            GB_TRY(hipStreamSynchronize(stream));
            if (slots + 1 > 0xffffffffULL) { error = "more than 2^32 attachment slots"; rc = VRC_ERR_LIMIT; goto cleanup; }
            GB_TRY(hipMalloc((void **)&lookup, total * sizeof(uint32_t)));
            GB_TRY(hipMalloc((void **)&attach, (slots + 1) * sizeof(uint64_t)));
            temp_bytes += n_chunks * sizeof(uint32_t);
            GB_TRY(hipMemsetAsync(lookup, 0, total * sizeof(uint32_t), stream));
            GB_TRY(hipMemsetAsync(attach, 0x05, sizeof(uint64_t), stream));       // slot 0: the default material
            hipLaunchKernelGGL(attach_fill_kernel, dim3((unsigned)n_chunks), dim3(256), 0, stream, resident_grid, src.occ[1], (int)depth,
                               (const uint64_t *)desc, (uint64_t)0, (const uint32_t *)d_chunks, lookup, attach);
            GB_TRY(hipGetLastError());
            GB_TRY(hipStreamSynchronize(stream));
            *n_attach = slots + 1;
        }

        if (validate_samples) {
            GB_TRY(hipMalloc((void **)&d_mis, sizeof(unsigned long long)));
            GB_TRY(hipMemsetAsync(d_mis, 0, sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(validate_grid_kernel, dim3((unsigned)((validate_samples + 255) / 256)), dim3(256), 0, stream,
                               resident_grid, (int)depth, (const uint64_t *)desc, (uint64_t)0, validate_samples, (uint64_t)0x5eedULL, d_mis);
            GB_TRY(hipGetLastError());
            unsigned long long mis = 0;
            GB_TRY(hipMemcpyAsync(&mis, d_mis, sizeof(mis), hipMemcpyDeviceToHost, stream));
            GB_TRY(hipStreamSynchronize(stream));
            bi.validate_samples = validate_samples;
            bi.validate_mismatches = mis;
        }
    }
finish:
    bi.seconds_height = t1 - t0;
    bi.seconds_count = t2 - t1;
    bi.seconds_emit = (flags & VRC_BUILD_COUNT_ONLY) ? 0.0 : t3 - t2;
    bi.seconds_total = now_s() - t0;
    bi.device_bytes_peak = temp_bytes + total * sizeof(uint64_t) * ((flags & VRC_BUILD_COUNT_ONLY) ? 0 : 1);
    bi.host_bytes = bricks.size() * (sizeof(BrickRef) + sizeof(BrickInfo) + sizeof(uint64_t)) + em.kv.size() * sizeof(uint64_t) +
                    brick_index.size() * 4;
    *d_desc = desc;
    desc = nullptr;
    if (lookup && attach) { *d_lookup = lookup; *d_attach = attach; lookup = nullptr; attach = nullptr; }

cleanup:
    if (out) *out = bi;
    (void)hipFree(lookup); (void)hipFree(attach); (void)hipFree(d_chunks); (void)hipFree(d_total);
    (void)hipFree(d_grid); (void)hipFree(d_occ); (void)hipFree(d_bricks); (void)hipFree(d_info); (void)hipFree(d_bases);
    (void)hipFree(d_kv); (void)hipFree(d_mis);
    if (desc) (void)hipFree(desc);
    return rc;
}

int build_shell_terrain_device(hipStream_t stream, uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor,
                               uint32_t flags, uint64_t validate_samples, const int32_t *probe_xy, uint32_t n_probe,
                               int32_t *probe_lohi, uint64_t **d_desc, vrc_build_info *out, std::string &error) {
    return build_columns_device(stream, depth, seed, thickness, octave_floor, nullptr, nullptr, flags, validate_samples, probe_xy,
                                n_probe, probe_lohi, d_desc, out, error);
}

}  // namespace vrc
