// svo_builder.cpp -- host-side SVO construction for the raycast path.
//
// Emits the reference's 64-bit child-descriptor array (include/map/Octree.h:89-94)
// in the layout Octree::Generate produces (src/map/Octree.cpp:13-43,171-323):
// bottom-up, buffer filled from the END downward, kept siblings ascending by
// child slot i = x | y<<1 | z<<2, a 15-bit pointer relative to the descriptor's
// own index, far-pointer slots holding absolute indices, and an all-ones page
// header every 0x8000 slots.
//
// Design (not a transcription): the array is built in *reversed coordinates*
// (k = distance from the end) in a growable vector, so nothing has to know the
// final size up front; absolute far-pointer values are patched when the array
// is flipped.  A node source abstracts where voxels come from, so the same
// emitter serves dense grids and procedural scenes that are far too large to
// materialise (4096^3 = 64 GiB as char).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/vrc.h"
#include "shell_scene.hpp"

namespace {
using vrc::lattice;
using vrc::splitmix64;

constexpr uint64_t kFarBit = 0x8000ULL;
constexpr uint64_t kLeafAll = 0xFF000000ULL;
constexpr uint64_t kValidAll = 0x00FF0000ULL;

struct Node {
    uint64_t desc = 0;
    int64_t k = -1;        // reversed coordinate of the first kept child's slot; -1 = none (bottom level)
};

// A subtree is dropped from its parent (parent leaf bit, no descriptor) when it
// is "leaf and all-invalid": util.hpp:198-235 IsLeaf && !CheckLeafSign.
inline bool is_empty_leaf(uint64_t d) {
    return (d & kValidAll) == 0 && (d & kLeafAll) == kLeafAll;
}

class Emitter {
public:
    // pages = false: the "brick" layout of the streamed / device builders (svo_builder_gpu.hip): same descriptor
    // format and the same bottom-up order, but no page-header slots, so the size of a subtree does not depend on
    // where it lands and subtrees can be counted and emitted independently
    explicit Emitter(bool strict, bool pages = true) : strict_(strict), pages_(pages) {}

    // Places the kept children of one node; returns the reversed coordinate of
    // the node's child block (Octree.cpp:245-319).
    int64_t place(const Node *kept, int n) {
        const int worst = 2 * n;
        if (pages_ && page_counter_ - worst <= 0) {       // page header (:252-262)
            skip(page_counter_);
            page_counter_ = 0x8000;
            push(~0ULL);
        }
        const int64_t far_block = k();
        int64_t far_slot[8];
        bool is_far[8];
        int nfar = 0;
        for (int i = n - 1; i >= 0; i--) {                // pessimistic far-pointer reservation (:270-285)
            const int64_t rel = kept[i].k < 0 ? -1 : (k() + worst) - kept[i].k;
            const bool far = strict_ ? rel > 0x8000 : rel > 0x7fff;
            is_far[i] = far;
            if (far) {
                far_slot[i] = k();
                far_fixups_.push_back(k());
                push((uint64_t)kept[i].k);                // patched to an absolute index in finish()
                page_counter_--;
                nfar++;
            }
        }
        int64_t next_far = far_block;
        for (int i = n - 1; i >= 0; i--) {                // the descriptors themselves (:288-315)
            const int64_t rel = kept[i].k < 0 ? -1 : k() - kept[i].k;
            uint64_t d = kept[i].desc;
            if (strict_) {
                // the reference re-derives far-ness from the actual distance and hands
                // out far slots in encounter order
                if (rel > 0x8000) {
                    d |= kFarBit;
                    d |= (uint64_t)(k() - next_far);
                    next_far++;
                } else if (rel > 0) {
                    d |= (uint64_t)rel;
                }
            } else {
                if (is_far[i]) {
                    d |= kFarBit;
                    d |= (uint64_t)(k() - far_slot[i]);
                } else if (rel > 0) {
                    d |= (uint64_t)rel;
                }
            }
            push(d);
            page_counter_--;
        }
        (void)nfar;
        return k() - 1;
    }

    void finish(uint64_t root_desc, uint64_t buffer_size, uint64_t **out, uint64_t *n_out, uint64_t *root_out) {
        push(root_desc | 1);                              // Octree.cpp:27-31
        const uint64_t used = (uint64_t)rev_.size();
        const uint64_t size = buffer_size ? buffer_size : used;
        uint64_t *buf = (uint64_t *)calloc(size ? size : 1, sizeof(uint64_t));
        *out = buf;
        *n_out = size;
        *root_out = size - used;
        if (!buf || used > size) { *out = nullptr; free(buf); return; }
        for (int64_t fk : far_fixups_) rev_[fk] = size - 1 - rev_[fk];
        for (uint64_t k = 0; k < used; k++) buf[size - 1 - k] = rev_[k];
    }

    uint64_t used() const { return rev_.size(); }

private:
    int64_t k() const { return (int64_t)rev_.size(); }
    void push(uint64_t v) { rev_.push_back(v); }
    void skip(int count) { rev_.resize(rev_.size() + (size_t)count, 0ULL); }

    std::vector<uint64_t> rev_;
    std::vector<int64_t> far_fixups_;
    int page_counter_ = 0x8000;
    bool strict_, pages_;
};

// Source concept: bool certainly_empty(x,y,z,size); uint8_t leaf_mask(x,y,z) for a 2^3 block.
template <class Source>
Node build(Emitter &em, const Source &src, int x, int y, int z, int size) {
    Node self;
    if (size == 2) {                                       // Octree.cpp:195-211
        self.desc = ((uint64_t)src.leaf_mask(x, y, z) << 16) | kLeafAll;
        return self;
    }
    const int h = size / 2;
    Node kept[8];
    int n = 0;
    for (int i = 0; i < 8; i++) {
        const int cx = x + ((i & 1) ? h : 0), cy = y + ((i & 2) ? h : 0), cz = z + ((i & 4) ? h : 0);
        Node child;
        if (src.certainly_empty(cx, cy, cz, h))
            child.desc = kLeafAll;                         // what an empty subtree evaluates to; writes nothing
        else
            child = build(em, src, cx, cy, cz, h);
        if (is_empty_leaf(child.desc)) {
            self.desc |= 1ULL << (i + 24);
        } else {
            self.desc |= 1ULL << (i + 16);
            kept[n++] = child;
        }
    }
    self.k = em.place(kept, n);
    return self;
}

struct DenseSource {
    const int8_t *grid;
    int64_t dim;
    bool certainly_empty(int, int, int, int) const { return false; }
    uint8_t leaf_mask(int x, int y, int z) const {
        uint8_t m = 0;
        for (int i = 0; i < 8; i++) {
            const int64_t at = (x + (i & 1)) + dim * ((y + ((i >> 1) & 1)) + dim * (z + ((i >> 2) & 1)));
            if (grid[at]) m |= (uint8_t)(1u << i);
        }
        return m;
    }
};

// ---- synthetic scene "shell-terrain"  ---------------------------
// Integer bilinear value noise: octaves with cell 2^k, k = depth-2 .. 2,
// amplitude dim / 2^(o+2) for the o-th octave, offset dim/4.
void make_heightfield(uint32_t depth, uint64_t seed, std::vector<int32_t> &h, int octave_floor = 2) {
    const int64_t dim = 1LL << depth;
    h.assign((size_t)(dim * dim), (int32_t)(dim / 4));
    int o = 0;
    for (int k = (int)depth - 2; k >= octave_floor; k--, o++) {
        const int64_t cell = 1LL << k;
        const int64_t amp = dim >> (o + 2);
        if (amp <= 0) break;
        const int64_t cells = dim / cell + 1;
        std::vector<int64_t> lat((size_t)(cells * cells));
        for (int64_t j = 0; j < cells; j++)
            for (int64_t i = 0; i < cells; i++) lat[(size_t)(i + cells * j)] = (int64_t)(lattice(seed, o, i, j) % (uint64_t)amp);
        for (int64_t y = 0; y < dim; y++) {
            const int64_t j = y >> k, fy = y & (cell - 1);
            for (int64_t x = 0; x < dim; x++) {
                const int64_t i = x >> k, fx = x & (cell - 1);
                const int64_t v00 = lat[(size_t)(i + cells * j)], v10 = lat[(size_t)(i + 1 + cells * j)];
                const int64_t v01 = lat[(size_t)(i + cells * (j + 1))], v11 = lat[(size_t)(i + 1 + cells * (j + 1))];
                const int64_t top = v00 * (cell - fx) + v10 * fx, bot = v01 * (cell - fx) + v11 * fx;
                h[(size_t)(x + dim * y)] += (int32_t)((top * (cell - fy) + bot * fy) >> (2 * k));
            }
        }
    }
}

// Column (x,y) is solid for lo(x,y) <= z <= hi(x,y): hi = h, lo = min(h of the
// 4-neighbourhood) - thickness, so the shell stays watertight on slopes.
struct ShellSource {
    int64_t dim;
    int levels;                                   // mip levels: level l has cells of 2^l columns
    std::vector<std::vector<int32_t>> lo_min, hi_max;   // per level: min(lo), max(hi)

    void init(uint32_t depth, const std::vector<int32_t> &h, int32_t thickness) {
        dim = 1LL << depth;
        levels = (int)depth + 1;
        lo_min.resize(levels);
        hi_max.resize(levels);
        lo_min[0].resize((size_t)(dim * dim));
        hi_max[0] = h;
        for (int64_t y = 0; y < dim; y++)
            for (int64_t x = 0; x < dim; x++) {
                int32_t m = h[(size_t)(x + dim * y)];
                if (x > 0) m = std::min(m, h[(size_t)(x - 1 + dim * y)]);
                if (x + 1 < dim) m = std::min(m, h[(size_t)(x + 1 + dim * y)]);
                if (y > 0) m = std::min(m, h[(size_t)(x + dim * (y - 1))]);
                if (y + 1 < dim) m = std::min(m, h[(size_t)(x + dim * (y + 1))]);
                lo_min[0][(size_t)(x + dim * y)] = std::max(0, m - thickness);
            }
        for (int l = 1; l < levels; l++) {
            const int64_t d = dim >> l, pd = dim >> (l - 1);
            lo_min[l].resize((size_t)(d * d));
            hi_max[l].resize((size_t)(d * d));
            for (int64_t y = 0; y < d; y++)
                for (int64_t x = 0; x < d; x++) {
                    const size_t a = (size_t)(2 * x + pd * (2 * y)), b = a + 1, c = a + (size_t)pd, e = c + 1;
                    lo_min[l][(size_t)(x + d * y)] = std::min(std::min(lo_min[l - 1][a], lo_min[l - 1][b]), std::min(lo_min[l - 1][c], lo_min[l - 1][e]));
                    hi_max[l][(size_t)(x + d * y)] = std::max(std::max(hi_max[l - 1][a], hi_max[l - 1][b]), std::max(hi_max[l - 1][c], hi_max[l - 1][e]));
                }
        }
    }
    // the same from caller-supplied columns: column (x,y) is solid for lo <= z <= hi (lo == nullptr: from z = 0 up)
    void init_columns(uint32_t depth, const uint16_t *hi, const uint16_t *lo) {
        dim = 1LL << depth;
        levels = (int)depth + 1;
        lo_min.assign(levels, {});
        hi_max.assign(levels, {});
        lo_min[0].resize((size_t)(dim * dim));
        hi_max[0].resize((size_t)(dim * dim));
        for (size_t i = 0; i < lo_min[0].size(); i++) { hi_max[0][i] = hi[i]; lo_min[0][i] = lo ? lo[i] : 0; }
        build_mips();
    }
    void build_mips() {
        for (int l = 1; l < levels; l++) {
            const int64_t d = dim >> l, pd = dim >> (l - 1);
            lo_min[l].resize((size_t)(d * d));
            hi_max[l].resize((size_t)(d * d));
            for (int64_t y = 0; y < d; y++)
                for (int64_t x = 0; x < d; x++) {
                    const size_t a = (size_t)(2 * x + pd * (2 * y)), b = a + 1, c = a + (size_t)pd, e = c + 1;
                    lo_min[l][(size_t)(x + d * y)] = std::min(std::min(lo_min[l - 1][a], lo_min[l - 1][b]), std::min(lo_min[l - 1][c], lo_min[l - 1][e]));
                    hi_max[l][(size_t)(x + d * y)] = std::max(std::max(hi_max[l - 1][a], hi_max[l - 1][b]), std::max(hi_max[l - 1][c], hi_max[l - 1][e]));
                }
        }
    }
    bool solid(int64_t x, int64_t y, int64_t z) const {
        const size_t at = (size_t)(x + dim * y);
        return z >= lo_min[0][at] && z <= hi_max[0][at];
    }
    bool certainly_empty(int x, int y, int z, int size) const {
        int l = 0;
        while ((1 << l) < size) l++;
        const int64_t d = dim >> l;
        const size_t at = (size_t)((x >> l) + d * (y >> l));
        return z > hi_max[l][at] || z + size - 1 < lo_min[l][at];
    }
    uint8_t leaf_mask(int x, int y, int z) const {
        uint8_t m = 0;
        for (int i = 0; i < 8; i++)
            if (solid(x + (i & 1), y + ((i >> 1) & 1), z + ((i >> 2) & 1))) m |= (uint8_t)(1u << i);
        return m;
    }
};

bool is_pow2(uint32_t v) { return v >= 2 && (v & (v - 1)) == 0; }

// Per-voxel materials for the SVO path ("attachments").  The reference allocates
// octree_attachment_lookup_buffer / octree_attachment_buffer (src/CLCaster.cpp:108-110) but never
// defines or reads them (ray_caster_kernel.cl:143-144); the layout is ours:
//   lookup[i]   uint32, parallel to the descriptor array: for a bottom-level descriptor i (node = 2^3
//               voxels) the slot in `attachments` holding its materials, 0 otherwise
//   attachments uint64: byte k = material (int8) of child slot k = x | y<<1 | z<<2; slot 0 is a
//               sentinel of eight 5s
template <class MaterialFn>
void walk_materials(const uint64_t *desc, uint64_t index, int x, int y, int z, int size, const MaterialFn &mat,
                    std::vector<uint32_t> &lookup, std::vector<uint64_t> &attach) {
    const uint64_t d = desc[index];
    const unsigned valid = (unsigned)(d >> 16) & 0xff, leaf = (unsigned)(d >> 24) & 0xff;
    if (size == 2) {
        uint64_t packed = 0;
        for (int k = 0; k < 8; k++)
            packed |= (uint64_t)(uint8_t)mat(x + (k & 1), y + ((k >> 1) & 1), z + ((k >> 2) & 1)) << (8 * k);
        lookup[index] = (uint32_t)attach.size();
        attach.push_back(packed);
        return;
    }
    const uint64_t base = (d & kFarBit) ? desc[index + (d & 0x7fff)] : index + (d & 0x7fff);
    const int h = size / 2;
    int before = 0;
    for (int k = 0; k < 8; k++) {
        if (!(valid & (1u << k))) continue;
        if (!(leaf & (1u << k)))
            walk_materials(desc, base + (uint64_t)before, x + ((k & 1) ? h : 0), y + ((k & 2) ? h : 0), z + ((k & 4) ? h : 0), h,
                           mat, lookup, attach);
        before++;
    }
}

template <class MaterialFn>
int build_attachments(const uint64_t *desc, uint64_t n, uint64_t root, uint32_t dim, const MaterialFn &mat,
                      uint32_t **lookup_out, uint64_t **attach_out, uint64_t *n_attach) {
    std::vector<uint32_t> lookup((size_t)n, 0u);
    std::vector<uint64_t> attach;
    attach.push_back(0x0505050505050505ULL);
    walk_materials(desc, root, 0, 0, 0, (int)dim, mat, lookup, attach);
    *lookup_out = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    *attach_out = (uint64_t *)malloc(sizeof(uint64_t) * attach.size());
    if (!*lookup_out || !*attach_out) { free(*lookup_out); free(*attach_out); return VRC_ERR_OUT_OF_MEMORY; }
    memcpy(*lookup_out, lookup.data(), sizeof(uint32_t) * (size_t)n);
    memcpy(*attach_out, attach.data(), sizeof(uint64_t) * attach.size());
    *n_attach = attach.size();
    return VRC_OK;
}

}  // namespace

extern "C" {

int vrc_octree_generate(const int8_t *grid, uint32_t dim, uint64_t buffer_size, int strict_reference,
                        uint64_t **descriptors, uint64_t *n_descriptors, uint64_t *root_index) {
    if (!grid || !descriptors || !n_descriptors || !root_index || !is_pow2(dim)) return VRC_ERR_INVALID_ARGUMENT;
    Emitter em(strict_reference != 0);
    DenseSource src{grid, (int64_t)dim};
    Node root = build(em, src, 0, 0, 0, (int)dim);
    if (buffer_size && em.used() + 1 > buffer_size) return VRC_ERR_LIMIT;
    em.finish(root.desc, buffer_size, descriptors, n_descriptors, root_index);
    return *descriptors ? VRC_OK : VRC_ERR_OUT_OF_MEMORY;
}

int vrc_octree_generate_ex(const int8_t *grid, uint32_t dim, uint32_t layout, uint64_t **descriptors, uint64_t *n_descriptors,
                           uint64_t *root_index) {
    if (!grid || !descriptors || !n_descriptors || !root_index || !is_pow2(dim)) return VRC_ERR_INVALID_ARGUMENT;
    Emitter em((layout & VRC_LAYOUT_STRICT_REFERENCE) != 0, (layout & VRC_LAYOUT_NO_PAGE_HEADERS) == 0);
    DenseSource src{grid, (int64_t)dim};
    Node root = build(em, src, 0, 0, 0, (int)dim);
    em.finish(root.desc, 0, descriptors, n_descriptors, root_index);
    return *descriptors ? VRC_OK : VRC_ERR_OUT_OF_MEMORY;
}

int vrc_scene_shell_terrain_ex(uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor, uint32_t layout,
                               uint64_t **descriptors, uint64_t *n_descriptors, uint64_t *root_index, int32_t *height) {
    if (depth < 3 || depth > 16 || octave_floor < 0 || thickness < 0 || !descriptors || !n_descriptors || !root_index)
        return VRC_ERR_INVALID_ARGUMENT;
    std::vector<int32_t> h;
    make_heightfield(depth, seed, h, octave_floor);
    if (height) memcpy(height, h.data(), h.size() * sizeof(int32_t));
    ShellSource src;
    src.init(depth, h, thickness);
    Emitter em((layout & VRC_LAYOUT_STRICT_REFERENCE) != 0, (layout & VRC_LAYOUT_NO_PAGE_HEADERS) == 0);
    Node root = build(em, src, 0, 0, 0, 1 << depth);
    em.finish(root.desc, 0, descriptors, n_descriptors, root_index);
    return *descriptors ? VRC_OK : VRC_ERR_OUT_OF_MEMORY;
}

int vrc_octree_from_columns(uint32_t depth, const uint16_t *hi, const uint16_t *lo, uint32_t layout, uint64_t **descriptors,
                            uint64_t *n_descriptors, uint64_t *root_index) {
    if (depth < 3 || depth > 13 || !hi || !descriptors || !n_descriptors || !root_index) return VRC_ERR_INVALID_ARGUMENT;
    const size_t n = (size_t)1 << (2 * depth);
    for (size_t i = 0; i < n; i++)
        if (hi[i] >= (1u << depth) || (lo && lo[i] > hi[i])) return VRC_ERR_INVALID_ARGUMENT;
    ShellSource src;
    src.init_columns(depth, hi, lo);
    Emitter em((layout & VRC_LAYOUT_STRICT_REFERENCE) != 0, (layout & VRC_LAYOUT_NO_PAGE_HEADERS) == 0);
    Node root = build(em, src, 0, 0, 0, 1 << depth);
    em.finish(root.desc, 0, descriptors, n_descriptors, root_index);
    return *descriptors ? VRC_OK : VRC_ERR_OUT_OF_MEMORY;
}

int vrc_scene_shell_terrain(uint32_t depth, uint64_t seed, int32_t thickness, int strict_reference,
                            uint64_t **descriptors, uint64_t *n_descriptors, uint64_t *root_index,
                            int32_t *height) {
    return vrc_scene_shell_terrain_ex(depth, seed, thickness, 2, strict_reference ? VRC_LAYOUT_STRICT_REFERENCE : 0u,
                                      descriptors, n_descriptors, root_index, height);
}

int vrc_scene_shell_column(uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor, int64_t x, int64_t y,
                           int32_t *lo, int32_t *hi) {
    const int64_t dim = 1LL << depth;
    if (depth < 3 || depth > 16 || octave_floor < 0 || x < 0 || y < 0 || x >= dim || y >= dim || !lo || !hi) return VRC_ERR_INVALID_ARGUMENT;
    const int32_t h = vrc::shell_height(depth, seed, octave_floor, x, y);
    const int32_t hxm = x > 0 ? vrc::shell_height(depth, seed, octave_floor, x - 1, y) : h;
    const int32_t hxp = x + 1 < dim ? vrc::shell_height(depth, seed, octave_floor, x + 1, y) : h;
    const int32_t hym = y > 0 ? vrc::shell_height(depth, seed, octave_floor, x, y - 1) : h;
    const int32_t hyp = y + 1 < dim ? vrc::shell_height(depth, seed, octave_floor, x, y + 1) : h;
    *hi = h;
    *lo = vrc::shell_floor(h, hxm, hxp, hym, hyp, thickness);
    return VRC_OK;
}

int vrc_scene_shell_terrain_dense(uint32_t depth, uint64_t seed, int32_t thickness, int8_t *grid) {
    if (depth < 3 || depth > 9 || !grid) return VRC_ERR_INVALID_ARGUMENT;
    std::vector<int32_t> h;
    make_heightfield(depth, seed, h);
    ShellSource src;
    src.init(depth, h, thickness);
    const int64_t dim = 1LL << depth;
    for (int64_t z = 0; z < dim; z++)
        for (int64_t y = 0; y < dim; y++)
            for (int64_t x = 0; x < dim; x++) grid[x + dim * (y + dim * z)] = src.solid(x, y, z) ? 5 : 0;
    return VRC_OK;
}

int vrc_octree_attachments_from_grid(const int8_t *grid, uint32_t dim, const uint64_t *descriptors, uint64_t n_descriptors,
                                     uint64_t root_index, uint32_t **lookup, uint64_t **attachments, uint64_t *n_attachments) {
    if (!grid || !descriptors || !lookup || !attachments || !n_attachments || !is_pow2(dim) || root_index >= n_descriptors)
        return VRC_ERR_INVALID_ARGUMENT;
    const int64_t d = dim;
    auto mat = [grid, d](int x, int y, int z) { return grid[x + d * (y + d * z)]; };
    return build_attachments(descriptors, n_descriptors, root_index, dim, mat, lookup, attachments, n_attachments);
}

int vrc_scene_shell_terrain_attachments(uint32_t depth, uint64_t seed, uint32_t mirror_period, const uint64_t *descriptors,
                                        uint64_t n_descriptors, uint64_t root_index, uint32_t **lookup,
                                        uint64_t **attachments, uint64_t *n_attachments) {
    if (depth < 3 || depth > 15 || !descriptors || !lookup || !attachments || !n_attachments || root_index >= n_descriptors)
        return VRC_ERR_INVALID_ARGUMENT;
    // material 6 (mirror) where hash(x,y,z,seed) % mirror_period == 0, else 5; 0 = no mirrors
    auto mat = [seed, mirror_period](int x, int y, int z) -> int8_t {
        if (!mirror_period) return 5;
        const uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)(uint32_t)x << 42) ^ ((uint64_t)(uint32_t)y << 21) ^ (uint64_t)(uint32_t)z));
        return (h % mirror_period) == 0 ? 6 : 5;
    };
    return build_attachments(descriptors, n_descriptors, root_index, 1u << depth, mat, lookup, attachments, n_attachments);
}

// the double field of Map::GenerateHeightBitmap before :248 quantises it
static void diamond_square_field(int n, double corner_seed, std::vector<double> &hm) {
    const int size = n + 1;                                    // Map.cpp:157 DATA_SIZE (samples wrap, :266-272)
    hm.assign((size_t)n * n, 0.0);
    auto at = [&](int x, int y) -> double & { return hm[(size_t)(x & (n - 1)) + (size_t)(y & (n - 1)) * n]; };
    std::mt19937 gen;                                          // :146 default seed
    std::uniform_real_distribution<double> dis(-1.0, 1.0);    // :147
    auto f_rand = [&]() { return dis(gen); };
    at(0, 0) = corner_seed; at(0, n) = corner_seed; at(n, 0) = corner_seed; at(n, n) = corner_seed;   // :163-166
    double h = 20.0;                                           // :168
    for (int side = size - 1; side >= 2; side /= 2, h /= 2.0) {   // :172-180
        const int half = side / 2;
        for (int x = 0; x < size - 1; x += side)               // squares :187-203
            for (int y = 0; y < size - 1; y += side) {
                const double avg = (at(x, y) + at(x + side, y) + at(x, y + side) + at(x + side, y + side)) / 4.0;
                at(x + half, y + half) = avg + (f_rand() * 2 * h) - h;
            }
        for (int x = 0; x < size - 1; x += half)               // diamonds :210-241
            for (int y = (x + half) % side; y < size - 1; y += side) {
                double avg = (at((x - half + size) % size, y) + at((x + half) % size, y) + at(x, (y + half) % size) +
                              at(x, (y - half + size) % size)) / 4.0;
                avg = avg + (f_rand() * 2 * h) - h;
                at(x, y) = avg;
                if (x == 0) at(size - 1, y) = avg;
                if (y == 0) at(x, size - 1) = avg;
            }
    }
}

int vrc_scene_diamond_square_f64(uint32_t dim, double corner_seed, double *field) {
    if (!is_pow2(dim) || dim > 16384u || !field) return VRC_ERR_INVALID_ARGUMENT;
    std::vector<double> hm;
    diamond_square_field((int)dim, corner_seed, hm);
    std::copy(hm.begin(), hm.end(), field);
    return VRC_OK;
}

int vrc_scene_diamond_square(uint32_t dim, double corner_seed, uint8_t *height, int8_t *grid) {
    // (the dense grid is dim^3 bytes: 4096 at most; the height field alone goes up to 16384^2 doubles = 2 GiB of host memory)
    if (!is_pow2(dim) || dim > (grid ? 4096u : 16384u) || !height) return VRC_ERR_INVALID_ARGUMENT;
    const int n = (int)dim;
    std::vector<double> hm;
    diamond_square_field(n, corner_seed, hm);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)                            // :248 clamp to [0, dimensions.z]
            height[(size_t)x + (size_t)y * n] = (uint8_t)std::min(std::max(hm[(size_t)x + (size_t)y * n], 0.0), (double)std::min(n, 255));
    if (grid)
        for (int z = 0; z < n; z++)
            for (int y = 0; y < n; y++)
                for (int x = 0; x < n; x++)
                    grid[(size_t)x + (size_t)n * ((size_t)y + (size_t)n * z)] = z <= (int)height[(size_t)x + (size_t)y * n] ? 5 : 0;
    return VRC_OK;
}

int vrc_scene_atlas(int32_t width, int32_t height, uint8_t *rgba8) {
    if (width <= 0 || height <= 0 || !rgba8) return VRC_ERR_INVALID_ARGUMENT;
    for (int32_t y = 0; y < height; y++)
        for (int32_t x = 0; x < width; x++) {
            const uint64_t v = splitmix64(((uint64_t)(uint32_t)y << 32) | (uint32_t)x);
            uint8_t *t = rgba8 + 4 * ((int64_t)x + (int64_t)width * y);
            t[0] = (uint8_t)v; t[1] = (uint8_t)(v >> 8); t[2] = (uint8_t)(v >> 16); t[3] = 255;
        }
    return VRC_OK;
}

// Octree::GetVoxel (src/map/Octree.cpp:45-158) / get_oct_vox
// (ray_caster_kernel.cl:140-251) on a host copy of the array.
int vrc_octree_get_voxel(const uint64_t *descriptors, uint64_t root_index, uint32_t dim, const int32_t position[3],
                         int32_t *found, int32_t *resolution, int32_t sub_oct_pos[3]) {
    if (!descriptors || !position || !is_pow2(dim)) return VRC_ERR_INVALID_ARGUMENT;
    uint64_t index = root_index, d = descriptors[index];
    int32_t dimension = (int32_t)dim, res = dimension / 2, corner[3] = {0, 0, 0};
    int hit = 1;
    while (dimension > 1) {
        const int32_t half = dimension / 2;
        int i = 0;
        for (int a = 0; a < 3; a++)
            if (position[a] >= corner[a] + half) { i |= 1 << a; corner[a] += half; }
        if (!((d >> 16) & (1ULL << i))) { hit = 0; break; }
        if ((d >> 24) & (1ULL << i)) break;
        dimension = half;
        res /= 2;
        const int before = __builtin_popcountll((d >> 16) & ((2ULL << i) - 1)) - 1;
        const uint64_t base = (d & kFarBit) ? descriptors[index + (d & 0x7fff)] : index + (d & 0x7fff);
        index = base + (uint64_t)before;
        d = descriptors[index];
    }
    if (found) *found = hit;
    if (resolution) *resolution = res;
    if (sub_oct_pos) for (int a = 0; a < 3; a++) sub_oct_pos[a] = corner[a];
    return VRC_OK;
}

// ---- SVO (de)serialisation: Octree::Load is declared but never defined in the reference
// (include/map/Octree.h:38, TODO src/main.cpp:30).  Little-endian file:
//   char[8] "VRCSVO01" | u32 dim | u32 flags(bit0: attachments) | u64 root | u64 n_desc | u64 n_attach |
//   u64 desc[n_desc] | (u32 lookup[n_desc] | u64 attach[n_attach])
int vrc_octree_save(const char *path, uint32_t dim, const uint64_t *descriptors, uint64_t n_descriptors, uint64_t root_index,
                    const uint32_t *lookup, const uint64_t *attachments, uint64_t n_attachments) {
    if (!path || !descriptors || !is_pow2(dim) || root_index >= n_descriptors) return VRC_ERR_INVALID_ARGUMENT;
    const bool with_att = lookup && attachments && n_attachments;
    FILE *f = fopen(path, "wb");
    if (!f) return VRC_ERR_NOT_FOUND;
    const uint32_t flags = with_att ? 1u : 0u;
    const uint64_t na = with_att ? n_attachments : 0;
    bool ok = fwrite("VRCSVO01", 1, 8, f) == 8 && fwrite(&dim, 4, 1, f) == 1 && fwrite(&flags, 4, 1, f) == 1 &&
              fwrite(&root_index, 8, 1, f) == 1 && fwrite(&n_descriptors, 8, 1, f) == 1 && fwrite(&na, 8, 1, f) == 1 &&
              fwrite(descriptors, 8, (size_t)n_descriptors, f) == (size_t)n_descriptors;
    if (ok && with_att)
        ok = fwrite(lookup, 4, (size_t)n_descriptors, f) == (size_t)n_descriptors && fwrite(attachments, 8, (size_t)na, f) == (size_t)na;
    ok = (fclose(f) == 0) && ok;
    return ok ? VRC_OK : VRC_ERR_DEVICE;
}

int vrc_octree_load(const char *path, uint32_t *dim, uint64_t **descriptors, uint64_t *n_descriptors, uint64_t *root_index,
                    uint32_t **lookup, uint64_t **attachments, uint64_t *n_attachments) {
    if (!path || !dim || !descriptors || !n_descriptors || !root_index) return VRC_ERR_INVALID_ARGUMENT;
    *descriptors = nullptr;
    if (lookup) *lookup = nullptr;
    if (attachments) *attachments = nullptr;
    if (n_attachments) *n_attachments = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return VRC_ERR_NOT_FOUND;
    char magic[8];
    uint32_t flags = 0;
    uint64_t na = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "VRCSVO01", 8) == 0 && fread(dim, 4, 1, f) == 1 &&
              fread(&flags, 4, 1, f) == 1 && fread(root_index, 8, 1, f) == 1 && fread(n_descriptors, 8, 1, f) == 1 &&
              fread(&na, 8, 1, f) == 1 && is_pow2(*dim) && *n_descriptors > 0 && *root_index < *n_descriptors;
    if (ok) {
        *descriptors = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)*n_descriptors);
        ok = *descriptors && fread(*descriptors, 8, (size_t)*n_descriptors, f) == (size_t)*n_descriptors;
    }
    if (ok && (flags & 1u) && lookup && attachments && n_attachments) {
        *lookup = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)*n_descriptors);
        *attachments = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(na ? na : 1));
        ok = *lookup && *attachments && fread(*lookup, 4, (size_t)*n_descriptors, f) == (size_t)*n_descriptors &&
             fread(*attachments, 8, (size_t)na, f) == (size_t)na;
        *n_attachments = na;
    }
    fclose(f);
    if (!ok) {
        free(*descriptors); *descriptors = nullptr;
        if (lookup) { free(*lookup); *lookup = nullptr; }
        if (attachments) { free(*attachments); *attachments = nullptr; }
        return VRC_ERR_INVALID_ARGUMENT;
    }
    return VRC_OK;
}

void vrc_free(void *p) { free(p); }

}  // extern "C"
