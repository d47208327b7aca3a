/*
 * vrc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See vrc_oracle.h for scope, pinning status and the reference citations.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * (oracle/Makefile).  Never built with -march=native / -mfma.
 */
#include "vrc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Octree.cpp:623-631, ray_caster_kernel.cl:37-46 */
static const uint8_t mask_8[8] = {0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40, 0x80};
static const uint8_t count_mask_8[8] = {0x1, 0x3, 0x7, 0xF, 0x1F, 0x3F, 0x7F, 0xFF};

/* ======================================================================= */
/* a2: Octree::Generate / GenerationRecursion  (src/map/Octree.cpp:13-43,171-323)
 * util.hpp:198-235 IsLeaf / CheckLeafSign                                   */

static int is_leaf(uint64_t d) {
    uint64_t v = d & ORC_VALID_MASK;
    if (v == ORC_VALID_MASK || v == 0)
        if ((d & ORC_LEAF_MASK) == ORC_LEAF_MASK) return 1;
    return 0;
}

static int check_leaf_sign(uint64_t d) {
    /* only reached when is_leaf() held, so valid is all-ones or all-zero */
    return (d & ORC_VALID_MASK) == ORC_VALID_MASK;
}

typedef struct {
    const int8_t *grid;
    int           dim;
    uint64_t     *buf;
    uint64_t      buffer_size;
    uint64_t      pos;                 /* descriptor_buffer_position (Octree.h:44) */
    int           page_header_counter; /* Octree.h:55 */
    int           overflow;
} gen_ctx;

typedef struct { uint64_t desc; uint64_t position; } gen_node;

static void gen_store(gen_ctx *g, uint64_t value) {
    if (g->pos >= g->buffer_size) { g->overflow = 1; return; }   /* pos wrapped below 0 */
    g->buf[g->pos] = value;
}

static gen_node generation_recursion(gen_ctx *g, int px, int py, int pz, unsigned voxel_scale) {
    gen_node self = {0, 0};
    const int s = (int)voxel_scale;
    /* child order i = x | y<<1 | z<<2   (Octree.cpp:176-185) */
    const int cx[8] = {px, px + s, px, px + s, px, px + s, px, px + s};
    const int cy[8] = {py, py, py + s, py + s, py, py, py + s, py + s};
    const int cz[8] = {pz, pz, pz, pz, pz + s, pz + s, pz + s, pz + s};

    if (g->overflow) return self;

    if (voxel_scale == 1) {                                       /* :195-211 */
        for (int i = 0; i < 8; i++) {
            int64_t at = (int64_t)cx[i] + (int64_t)g->dim * ((int64_t)cy[i] + (int64_t)g->dim * cz[i]);
            if (g->grid[at]) self.desc |= (uint64_t)1 << (i + 16);
        }
        self.desc |= 0xFF000000ULL;
        return self;
    }

    gen_node kept[8];
    int n = 0;
    for (int i = 0; i < 8; i++) {                                 /* :217-243 */
        gen_node child = generation_recursion(g, cx[i], cy[i], cz[i], voxel_scale / 2);
        if (is_leaf(child.desc) && !check_leaf_sign(child.desc)) {
            self.desc |= (uint64_t)1 << (i + 16 + 8);
        } else {
            self.desc |= (uint64_t)1 << (i + 16);
            kept[n++] = child;
        }
    }
    if (g->overflow) return self;

    int worst_case_insertion_size = n * 2;                        /* :249 */

    if (g->page_header_counter - worst_case_insertion_size <= 0) { /* :252-262 */
        g->pos -= (uint64_t)(int64_t)g->page_header_counter;
        g->page_header_counter = 0x8000;
        gen_store(g, ~(uint64_t)0);          /* current_info_section_position */
        g->pos--;
    }

    uint64_t far_pointer_block_position = g->pos;                 /* :267 */

    for (int i = n - 1; i >= 0; i--) {                            /* :270-285 */
        int relative_distance = (int)(kept[i].position - (g->pos - (uint64_t)(int64_t)worst_case_insertion_size));
        if (relative_distance > 0x8000) {
            gen_store(g, kept[i].position);
            g->pos--;
            g->page_header_counter--;
        }
    }

    for (int i = n - 1; i >= 0; i--) {                            /* :288-315 */
        int relative_distance = (int)(kept[i].position - g->pos);
        uint64_t descriptor = kept[i].desc;
        if (relative_distance > 0x8000) {
            descriptor |= ORC_FAR_BIT_MASK;
            descriptor |= far_pointer_block_position - g->pos;
            far_pointer_block_position--;
        } else if (relative_distance > 0) {
            descriptor |= (uint64_t)relative_distance;
        }
        gen_store(g, descriptor);
        g->pos--;
        g->page_header_counter--;
    }

    self.position = g->pos + 1;                                   /* :319 */
    return self;
}

int orc_octree_generate(const int8_t *grid, int dim, uint64_t *buffer, uint64_t buffer_size,
                        uint64_t *root_index, uint64_t *lowest_used) {
    gen_ctx g;
    g.grid = grid;
    g.dim = dim;
    g.buf = buffer;
    g.buffer_size = buffer_size;
    g.pos = buffer_size - 1;
    g.page_header_counter = 0x8000;
    g.overflow = 0;

    gen_node root = generation_recursion(&g, 0, 0, 0, (unsigned)dim / 2);   /* :19 */
    root.desc |= 1;                                                          /* :27 */
    gen_store(&g, root.desc);
    if (g.overflow) return -1;
    if (root_index) *root_index = g.pos;                                     /* :30 */
    if (lowest_used) *lowest_used = g.pos;
    return 0;
}

/* ======================================================================= */
/* Where descriptors come from: the flat array (kernel arg 12), or -- for trees that only exist in the GPU's memory
 * (the ~200 GB scene of BASELINE configs[4]) -- pages of ORC_PAGE_SIZE descriptors fetched on demand through a
 * callback and kept in a page table.  Either way the same indices are read.                                     */
typedef struct {
    const uint64_t   *flat;
    const uint64_t  **pages;
    orc_page_fetch_fn fetch;
    void             *user;
} desc_src;

static const uint64_t *fetch_page(const desc_src *src, uint64_t page) {
    const uint64_t *p;
#pragma omp critical(orc_desc_page)
    {
        p = __atomic_load_n(&src->pages[page], __ATOMIC_ACQUIRE);
        if (!p) {
            p = src->fetch(src->user, page);
            __atomic_store_n((const uint64_t **)&src->pages[page], p, __ATOMIC_RELEASE);
        }
    }
    return p;
}

static inline uint64_t desc_at(const desc_src *src, uint64_t i) {
    if (!src->fetch) return src->flat[i];
    const uint64_t *p = __atomic_load_n(&src->pages[i >> ORC_PAGE_SHIFT], __ATOMIC_ACQUIRE);
    if (!p) p = fetch_page(src, i >> ORC_PAGE_SHIFT);
    return p[i & (ORC_PAGE_SIZE - 1)];
}

static desc_src scene_src(const orc_scene *s) {
    desc_src src = {s->descriptors, s->desc_pages, s->desc_page_fetch, s->desc_page_user};
    return src;
}

/* ======================================================================= */
/* a3: get_oct_vox  (kernels/ray_caster_kernel.cl:140-251)                  */

static void get_oct_vox_src(const int32_t position[3], const desc_src *descriptors,
                            uint64_t root_index, int32_t dim, orc_traversal_state *ts) {
    memset(ts, 0, sizeof(*ts));
    ts->current_descriptor_index = root_index;                    /* :150-151 */
    ts->current_descriptor = desc_at(descriptors, root_index);
    ts->reads = 1;
    ts->scale = 0;
    ts->parent_stack_position = 0;
    ts->found = 0;
    ts->parent_stack[0] = ts->current_descriptor;                 /* :158-159 */
    ts->parent_stack_index[0] = ts->current_descriptor_index;

    int dimension = dim;                                          /* :162-165 */
    ts->resolution = dimension / 2;

    while (dimension > 1) {                                       /* :176 */
        for (int a = 0; a < 3; a++) ts->oct_pos[a] = ts->sub_oct_pos[a];

        uint8_t idx = 0;                                          /* :181-191 */
        for (int a = 0; a < 3; a++) {
            if (position[a] >= dimension / 2 + ts->oct_pos[a]) {
                idx |= (uint8_t)(1 << a);
                ts->sub_oct_pos[a] += dimension / 2;
            }
        }
        ts->idx_stack[ts->scale] = idx;
        int mask_index = idx;

        if ((ts->current_descriptor >> 16) & mask_8[mask_index]) {        /* :196 */
            if ((ts->current_descriptor >> 24) & mask_8[mask_index]) {    /* :199 */
                ts->found = 1;
                return;                       /* early exit: resolution not halved */
            }
            ts->scale++;                                          /* :211-214 */
            ts->parent_stack_position++;
            dimension /= 2;
            ts->resolution /= 2;

            int count = __builtin_popcount((uint8_t)(ts->current_descriptor >> 16) & count_mask_8[mask_index]) - 1;

            if (ORC_FAR_BIT_MASK & desc_at(descriptors, ts->current_descriptor_index)) {   /* :222-225 */
                uint64_t far_pointer_index = ts->current_descriptor_index + (ts->current_descriptor & ORC_CHILD_POINTER_MASK);
                ts->current_descriptor_index = desc_at(descriptors, far_pointer_index) + (uint64_t)(int64_t)count;
            } else {                                              /* :228-230 */
                ts->current_descriptor_index = ts->current_descriptor_index
                    + (ts->current_descriptor & ORC_CHILD_POINTER_MASK) + (uint64_t)(int64_t)count;
            }
            ts->current_descriptor = desc_at(descriptors, ts->current_descriptor_index);   /* :233 */
            ts->reads++;
            ts->parent_stack[ts->parent_stack_position] = ts->current_descriptor;
            ts->parent_stack_index[ts->parent_stack_position] = ts->current_descriptor_index;
        } else {
            ts->found = 0;                                        /* :245-246 */
            return;
        }
    }
    ts->found = 1;                                                /* :249 */
}

void orc_get_oct_vox(const int32_t position[3], const uint64_t *descriptors,
                     uint64_t root_index, int32_t dim, orc_traversal_state *ts) {
    const desc_src src = {descriptors, NULL, NULL, NULL};
    get_oct_vox_src(position, &src, root_index, dim, ts);
}

int64_t orc_octree_validate(const int8_t *grid, int dim, const uint64_t *descriptors,
                            uint64_t root_index) {
    int64_t bad = 0;
    for (int z = 0; z < dim; z++)
        for (int y = 0; y < dim; y++)
            for (int x = 0; x < dim; x++) {
                int32_t p[3] = {x, y, z};
                orc_traversal_state ts;
                orc_get_oct_vox(p, descriptors, root_index, dim, &ts);
                int8_t arr = grid[(int64_t)x + (int64_t)dim * ((int64_t)y + (int64_t)dim * z)];
                if ((arr != 0) != (ts.found != 0)) bad++;
            }
    return bad;
}

/* ======================================================================= */
/* a4: CLCaster::create_viewport  (src/CLCaster.cpp:233-275) + Normalize
 * (include/util.hpp:64-73)                                                 */

void orc_create_viewport(int32_t w, int32_t h, float *table) {
    memset(table, 0, sizeof(float) * 4 * (size_t)w * (size_t)h);
    const double s157 = sin(1.57), c157 = cos(1.57);
    for (int y = -h / 2; y < h / 2; y++) {
        for (int x = -w / 2; x < w / 2; x++) {
            float rx = -800.0f, ry = (float)x, rz = (float)y;
            /* float * double -> double arithmetic, cast back to float (:252-256) */
            float nx = (float)((double)rz * s157 + (double)rx * c157);
            float ny = ry;
            float nz = (float)((double)rz * c157 - (double)rx * s157);
            float multiplier = sqrtf(nx * nx + ny * ny + nz * nz);
            int64_t index = (int64_t)(x + w / 2) + (int64_t)w * (y + h / 2);
            table[4 * index + 0] = nx / multiplier;
            table[4 * index + 1] = ny / multiplier;
            table[4 * index + 2] = nz / multiplier;
            table[4 * index + 3] = 0.0f;
        }
    }
}

/* ======================================================================= */
/* helpers with OpenCL builtin semantics, restated                          */

static inline float dot3(const float a[3], const float b[3]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* OpenCL normalize(): returns v unchanged when all of v is zero */
static inline void normalize3(const float v[3], float out[3]) {
    float l2 = dot3(v, v);
    if (l2 == 0.0f) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; return; }
    float l = sqrtf(l2);
    out[0] = v[0] / l; out[1] = v[1] / l; out[2] = v[2] / l;
}

static inline float mixf(float x, float y, float a) { return x + (y - x) * a; }
static inline float fmax_cl(float a, float b) { return a < b ? b : a; }   /* max(): x < y ? y : x */
static inline float fmin_cl(float a, float b) { return b < a ? b : a; }   /* min(): y < x ? y : x */
static inline int isign(float v) { return (v > 0.0f) - (v < 0.0f); }

/* view_light  (kernels/ray_caster_kernel.cl:78-99) */
static void view_light(float out[4], const float in_color[4], const float light[3],
                       const float light_color[4], const float view[3], const int mask[3]) {
    if (light[0] == 0.0f && light[1] == 0.0f && light[2] == 0.0f) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return;
    }
    float d = sqrtf(dot3(light, light)) * 0.01f;      /* fast_length -> sqrt */
    d *= d;

    float fmask[3] = {(float)mask[0], (float)mask[1], (float)mask[2]};
    float nmask[3], nlight[3], nview[3];
    normalize3(fmask, nmask);
    normalize3(light, nlight);
    float diffuse = fmax_cl(dot3(nmask, nlight), 0.1f);
    float specular = 0.0f;
    if (diffuse > 0.0f) {
        normalize3(view, nview);
        float hsum[3] = {nlight[0] + nview[0], nlight[1] + nview[1], nlight[2] + nview[2]};
        float halfway[3];
        normalize3(hsum, halfway);
        float spec_tmp = fmax_cl(dot3(nmask, halfway), 0.0f);
        specular = spec_tmp;                          /* pow(x, 1.0f) == x */
    }
    for (int c = 0; c < 4; c++)
        out[c] = in_color[c] + (diffuse * light_color[c] + specular * light_color[c] / d);
}

void orc_view_light(float out[4], const float in_color[4], const float light[3], const float light_color[4],
                    const float view[3], const int32_t mask[3]) {
    const int m[3] = {mask[0], mask[1], mask[2]};
    view_light(out, in_color, light, light_color, view, m);
}

void orc_camera_trig(const float cam_dir[2], float trig[4]) {
    trig[0] = sinf(cam_dir[0]);
    trig[1] = cosf(cam_dir[0]);
    trig[2] = sinf(cam_dir[1]);
    trig[3] = cosf(cam_dir[1]);
}

/* read_imagef on an RGBA8 UNORM image, integer coords; out-of-range coords are
 * undefined in OpenCL (no sampler) -- the oracle clamps to the edge.          */
static void atlas_fetch(const orc_scene *s, int tx, int ty, float out[4]) {
    int w = s->atlas_dim[0], h = s->atlas_dim[1];
    if (tx < 0) tx = 0;
    if (ty < 0) ty = 0;
    if (tx >= w) tx = w - 1;
    if (ty >= h) ty = h - 1;
    const uint8_t *p = s->atlas_rgba8 + 4 * ((int64_t)tx + (int64_t)w * ty);
    for (int c = 0; c < 4; c++) out[c] = (float)p[c] / 255.0f;
}

/* ---- SVO occupancy cursor: the canonical traversal of SURVEY 8(d).
 * One descriptor read for the root per ray, one per descent into a kept
 * child; pops and steps inside a known-empty node cost nothing.            */
typedef struct {
    desc_src descriptors;
    int      n;                       /* log2(dim) */
    int      top;
    uint64_t desc[ORC_MAX_DEPTH];
    uint64_t idx[ORC_MAX_DEPTH];
    int32_t  pv[3];
    uint32_t reads;
    int      coarse_level;            /* mode B with the coarse table (0: none): see svo_locate */
} svo_cursor;

static void svo_init(svo_cursor *c, const desc_src *descriptors, uint64_t root_index, int dim) {
    c->descriptors = *descriptors;
    c->n = 0;
    while ((1 << c->n) < dim) c->n++;
    c->top = 0;
    c->idx[0] = root_index;
    c->desc[0] = desc_at(descriptors, root_index);
    c->pv[0] = c->pv[1] = c->pv[2] = 0;
    c->reads = 1;
    c->coarse_level = 0;
}

/* returns 1 when voxel v is solid (valid & leaf), 0 when it lies in an empty node */
static int svo_locate_from(svo_cursor *c, const int32_t v[3], int table);
static int svo_locate(svo_cursor *c, const int32_t v[3]) { return svo_locate_from(c, v, 0); }

/* table != 0 (mode B, after a jump): when the voxel lies in another cell of the level-L grid (L = coarse_level) than the voxel
 * located last, the product reads the cursor state of the new cell from a dense table -- ONE 8-byte read that stands for
 * the descent from the root to level L, or to the node above it whose child toward the cell is empty or a leaf
 * (raycast_jump_kernel.hip table_block / coarse_build_kernel) -- and goes on below as usual.  Restated here without a table:
 * the same descent, its reads not counted, one read for the table entry.  The nodes found and every result are those of
 * the canonical traversal; only `reads` differs.                                                                          */
static int svo_locate_from(svo_cursor *c, const int32_t v[3], int table) {
    uint32_t diff = (uint32_t)((v[0] ^ c->pv[0]) | (v[1] ^ c->pv[1]) | (v[2] ^ c->pv[2]));
    const int L = table ? c->coarse_level : 0;
    int free_until = 0;               /* descents into levels <= free_until are the table's */
    if (L > 0 && (diff >> (c->n - L)) != 0) {
        c->top = 0;
        c->reads++;
        free_until = L;
    } else {
        while (c->top > 0 && (diff >> (c->n - c->top)) != 0) c->top--;
    }
    c->pv[0] = v[0]; c->pv[1] = v[1]; c->pv[2] = v[2];
    for (;;) {
        int b = c->n - c->top - 1;
        int i = ((v[0] >> b) & 1) | (((v[1] >> b) & 1) << 1) | (((v[2] >> b) & 1) << 2);
        uint64_t d = c->desc[c->top];
        if (!((d >> 16) & mask_8[i])) return 0;
        if ((d >> 24) & mask_8[i]) return 1;
        if (b == 0) return 1;         /* get_oct_vox falls out of its loop with found = 1 (:249) */
        int count = __builtin_popcount((uint8_t)(d >> 16) & count_mask_8[i]) - 1;
        uint64_t at = c->idx[c->top];
        uint64_t child;
        if (d & ORC_FAR_BIT_MASK)
            child = desc_at(&c->descriptors, at + (d & ORC_CHILD_POINTER_MASK)) + (uint64_t)(int64_t)count;
        else
            child = at + (d & ORC_CHILD_POINTER_MASK) + (uint64_t)(int64_t)count;
        c->top++;
        c->idx[c->top] = child;
        c->desc[c->top] = desc_at(&c->descriptors, child);
        if (c->top > free_until) c->reads++;
    }
}

/* ---- the light part of the hit block (:657-679) for one light, from the FIRST STRIKE
 * (primary hit): view_light, max_distance, redirect toward the light.  With one active light this is the
 * reference's code at its place; with several (extension, SURVEY 8f-1, TODO src/main.cpp:33 "Multi light
 * support via first-strike resetting") it is re-run from the stored strike for each further light.
 * Returns 0 for the early return of :671-672 (pixel left unwritten).                                          */
typedef struct {
    int   voxel[3];        /* the solid voxel of the first strike */
    float face_position[3];
    int   nmask[3];        /* face_mask * voxel_step of the primary ray */
    int   distance;        /* distance_traveled at the strike */
} orc_strike;

static int light_from_strike(const orc_scene *s, const float *L, const orc_strike *k, const float in_color[4],
                             float color_accumulator[4], float rd[3], int voxel[3], int vstep[3],
                             float delta_t[3], float it[3], int *max_distance, int cast) {
    const float light_pos[3] = {L[4], L[5], L[6]};
    const float light_rgbi[4] = {L[0], L[1], L[2], L[3]};
    float hit_pos[3], to_light[3], to_view[3];
    for (int a = 0; a < 3; a++) {
        hit_pos[a] = (float)k->voxel[a] + k->face_position[a];
        to_light[a] = hit_pos[a] - light_pos[a];
        to_view[a] = hit_pos[a] - s->cam_pos[a];
    }
    float lit[4];
    view_light(lit, in_color, to_light, light_rgbi, to_view, k->nmask);      /* :657-664 */
    for (int c = 0; c < 4; c++) color_accumulator[c] = lit[c];
    if (!cast) return 1;
    {                                                 /* :667 int = int + float */
        float dv[3] = {(float)k->voxel[0] - light_pos[0], (float)k->voxel[1] - light_pos[1],
                       (float)k->voxel[2] - light_pos[2]};
        *max_distance = (int)((float)k->distance + sqrtf(dot3(dv, dv)));
    }
    float lv[3] = {light_pos[0] - hit_pos[0], light_pos[1] - hit_pos[1], light_pos[2] - hit_pos[2]};
    normalize3(lv, rd);                               /* :670 */
    if (rd[0] == 0.0f || rd[1] == 0.0f || rd[2] == 0.0f) return 0;           /* :671-672 */
    for (int a = 0; a < 3; a++) {
        voxel[a] = k->voxel[a] - k->nmask[a];         /* :674 voxel -= voxel_step * face_mask */
        vstep[a] = isign(rd[a]);                      /* :675 */
        delta_t[a] = fabsf(1.0f / rd[a]);             /* :677 */
        it[a] = delta_t[a] * (hit_pos[a] - floorf(hit_pos[a])) * (float)vstep[a];   /* :678 */
        it[a] += delta_t[a] * -(it[a] < 0.0f ? -1.0f : 0.0f);                       /* :679 */
    }
    return 1;
}

/* ---- stepping mode 1, "mode B" of SURVEY 7 D1: the node-sized jump the reference's unfinished octree branch was
 * heading for (ray_caster_kernel.cl:525-540: intersection_t += delta_t * jump_power * fabs(face_mask), with the
 * correction of the other axes commented out), done the stateless way: the exit of the ray  origin + t * ray_dir
 * from the empty node [corner, corner + size)^3 it is in is  t = min_a (plane_a - origin_a) * (1 / ray_dir_a);
 * the axes that reach the minimum cross to the neighbouring voxel, the others land on floor(origin_a + t * ray_dir_a)
 * clamped into the node.  NOT the reference's float sequence (the reference accumulates intersection_t by repeated
 * addition): a different renderer that agrees with the exact mode on nearly all hit voxels; this restatement is what
 * the HIP jump kernel is bit-exact against.  Definitions the two share:
 *   - steps of a jump = |dx| + |dy| + |dz| of the voxel (what the per-voxel loop would count, ties aside); the :357
 *     guard ends the ray inside the node when the cap falls into the jump;
 *   - on a hit the intersection_t the hit block (:586-618) reads is rebuilt from the exit: t + delta_t on the axes
 *     that crossed, the ray's next crossing of the hit voxel's far plane on the others;
 *   - a restarted ray (shadow / mirror) starts where the reference's restart arithmetic puts it (jump_restart_ray)
 *     and jumps out of the single voxel it starts in first;
 *   - the reference's octree bias (:353-354) has no counterpart (it biases an accumulated intersection_t).        */
typedef struct {
    float origin[3];
    float inv[3];            /* 1 / ray_dir */
    int   corner[3];
    int   size;              /* the empty node the voxel is in; 1 = extent unknown */
} jump_state;

static void jump_set_ray(jump_state *j, const float origin[3], const float rd[3]) {
    /* 1 / a denormal component is inf and 0 * inf (a ray starting exactly on a plane) NaN: clamped to the largest float,
     * as the kernel does (raycast_jump_kernel.hip set_ray), so that every jump has a minimum and makes progress */
    for (int a = 0; a < 3; a++) { j->origin[a] = origin[a]; j->inv[a] = fminf(fmaxf(1.0f / rd[a], -FLT_MAX), FLT_MAX); }
}
/* A restarted ray (:677-679 shadow, :700-702 mirror).  The reference sets intersection_t = delta_t * frac(hit_pos) *
 * voxel_step (+ delta_t where negative): the distance to the next plane is frac for rays going up an axis and 1 - frac
 * for rays going down -- the mirror image of the geometric one (the primary set-up :313-323 has it right).  Its shadow
 * and mirror rays therefore behave as if they started from hit_pos mirrored inside the restart voxel; mode B starts
 * them exactly there, so it follows the reference's rays, not the geometrically intended ones.                      */
static void jump_restart_ray(jump_state *j, const float hit_pos[3], const int voxel[3], const float rd[3]) {
    float o[3];
    for (int a = 0; a < 3; a++) o[a] = (float)voxel[a] + (1.0f - (hit_pos[a] - floorf(hit_pos[a])));
    jump_set_ray(j, o, rd);
}
static void jump_set_node(jump_state *j, const int voxel[3], int size) {
    for (int a = 0; a < 3; a++) j->corner[a] = voxel[a] & ~(size - 1);
    j->size = size;
}
/* leaves the node: updates voxel, face_mask, returns the number of voxel steps */
static int jump_step(const jump_state *j, const float rd[3], const int vstep[3], int voxel[3], int face_mask[3], float *t_exit) {
    float tx[3];
    for (int a = 0; a < 3; a++) {
        const int plane = vstep[a] > 0 ? j->corner[a] + j->size : j->corner[a];
        tx[a] = ((float)plane - j->origin[a]) * j->inv[a];
    }
    float t = tx[0];
    if (tx[1] < t) t = tx[1];
    if (tx[2] < t) t = tx[2];
    int steps = 0;
    for (int a = 0; a < 3; a++) {
        int to;
        face_mask[a] = tx[a] <= t;
        if (face_mask[a]) {
            to = vstep[a] > 0 ? j->corner[a] + j->size : j->corner[a] - 1;
        } else {
            const float p = j->origin[a] + t * rd[a];
            to = (int)floorf(p);
            if (to < j->corner[a]) to = j->corner[a];
            if (to > j->corner[a] + j->size - 1) to = j->corner[a] + j->size - 1;
        }
        steps += to > voxel[a] ? to - voxel[a] : voxel[a] - to;
        voxel[a] = to;
    }
    *t_exit = t;
    return steps;
}
/* the intersection_t of a ray that has just entered `voxel` at parameter t through the faces in face_mask */
static void jump_hit_intersection(const jump_state *j, const int vstep[3], const int voxel[3], const int face_mask[3], float t,
                                  const float delta_t[3], float it[3]) {
    for (int a = 0; a < 3; a++) {
        if (face_mask[a]) {
            it[a] = t + delta_t[a];
        } else {
            const int plane = vstep[a] > 0 ? voxel[a] + 1 : voxel[a];
            it[a] = ((float)plane - j->origin[a]) * j->inv[a];
        }
    }
}

/* ---- one pixel of `raycaster`  (kernels/ray_caster_kernel.cl:256-724) */
static void raycast_pixel(const orc_scene *s, int px, int py, const int32_t bias[3],
                          float *image, int32_t *hits, orc_counters *ctr) {
    const int W = s->resolution[0];
    const int64_t pix = (int64_t)px + (int64_t)W * py;
    int32_t *hit = hits ? hits + 8 * pix : NULL;
    if (hit) {
        hit[ORC_HIT_VX] = hit[ORC_HIT_VY] = hit[ORC_HIT_VZ] = -1;
        hit[ORC_HIT_MATERIAL] = hit[ORC_HIT_FACE] = hit[ORC_HIT_FLAGS] = 0;
        hit[ORC_HIT_STEPS] = hit[ORC_HIT_NDESC] = 0;
    }

    /* :276-291 fetch + pitch + yaw */
    const float *pm = s->viewport_matrix + 4 * pix;
    float rd[3] = {pm[0], pm[1], pm[2]};
    const float s1 = s->cam_trig[0], c1 = s->cam_trig[1], s2 = s->cam_trig[2], c2 = s->cam_trig[3];
    {
        float x = rd[2] * s1 + rd[0] * c1;
        float y = rd[1];
        float z = rd[2] * c1 - rd[0] * s1;
        rd[0] = x; rd[1] = y; rd[2] = z;
    }
    {
        float x = rd[0] * c2 - rd[1] * s2;
        float y = rd[0] * s2 + rd[1] * c2;
        float z = rd[2];
        rd[0] = x; rd[1] = y; rd[2] = z;
    }
    if (rd[0] == 0.0f || rd[1] == 0.0f || rd[2] == 0.0f) {      /* :293-294 */
        ctr->unwritten++;
        return;
    }
    ctr->primary_rays++;

    int vstep[3], voxel[3];
    float delta_t[3], it[3];
    for (int a = 0; a < 3; a++) {
        vstep[a] = isign(rd[a]);                                  /* :298 */
        float fl = floorf(s->cam_pos[a]);
        voxel[a] = (int)fl;                                       /* :302 convert_int3_rtn */
        delta_t[a] = fabsf(1.0f / rd[a]);                         /* :307 */
        float offset = delta_t[a] * (s->cam_pos[a] - fl);         /* :313 */
        it[a] = offset * -(float)vstep[a];                        /* :317 */
        it[a] += delta_t[a] * -1.0f * (it[a] < 0.0f ? -1.0f : 0.0f);   /* :323 */
    }

    int distance_traveled = 0;
    int max_distance = s->max_distance;                           /* :326 */
    unsigned bounce_count = 0;
    int face_mask[3] = {0, 0, 0};
    int voxel_data = 0;
    float face_position[3];
    float voxel_color[4] = {0, 0, 0, 0};
    float tile_face_position[2];
    float sign[3];
    float color_accumulator[4] = {0, 0, 0, 0};
    float fog_distance = 0.0f;
    int shadow_ray = 0;
    int flags = ORC_FLAG_WRITTEN;

    /* :342-354 get_oct_vox(camera voxel) is pixel-independent: the bias
     * (sub_oct_pos - voxel) * resolution / 2 is evaluated once per frame by
     * the caller; its descriptor reads are charged per ray below.            */
    const int svo = (s->using_octree == 0);
    const int jump = svo && s->stepping_mode == 1;
    if (!jump)
        for (int a = 0; a < 3; a++) it[a] += (float)bias[a];
    jump_state js;
    memset(&js, 0, sizeof(js));
    js.size = 1;

    svo_cursor cur;
    uint32_t ndesc = 0;
    if (svo) {
        const desc_src src = scene_src(s);
        svo_init(&cur, &src, (uint64_t)s->octree_root_index, (int)s->octree_dimensions);
        if (jump) {                                   /* the product's rule (vrc_params.h coarse_level_for_depth), restated */
            int L = s->coarse_log2 < 0 ? (cur.n >= 14 ? 10 : cur.n >= 5 ? (cur.n - 2 < 9 ? cur.n - 2 : 9) : 0) : s->coarse_log2;
            /* round 5 (vrc_api.cpp): by default no table of more than 16 x the descriptor array's bytes, unless it is below 1 MiB */
            while (s->coarse_log2 < 0 && L >= 1 && (8ULL << (3 * L)) > 16ULL * 8ULL * s->n_descriptors && (8ULL << (3 * L)) > (1ULL << 20)) L--;
            if (L > cur.n - 2) L = cur.n - 2;
            if (L > 10) L = 10;
            cur.coarse_level = (L >= 1 && s->n_descriptors < (1ULL << 43)) ? L : 0;
        }
        /* the per-pixel get_oct_vox(camera voxel) of the reference == the
         * cursor's first descent (only meaningful when the camera is inside
         * the map; otherwise only the root read is charged)                  */
        if (voxel[0] >= 0 && voxel[1] >= 0 && voxel[2] >= 0 && voxel[0] < s->map_dim[0] &&
            voxel[1] < s->map_dim[1] && voxel[2] < s->map_dim[2]) {
            const int solid = svo_locate(&cur, voxel);
            if (jump) jump_set_node(&js, voxel, solid ? 1 : 1 << (cur.n - cur.top - 1));
        } else if (jump) {
            jump_set_node(&js, voxel, 1);
        }
        if (jump) jump_set_ray(&js, s->cam_pos, rd);
    }

    /* the reference reads light 0 only (:660-670); active_lights > 1 is the multi-light extension */
    int nlights = s->active_lights < s->light_count ? s->active_lights : s->light_count;
    if (nlights > 8) nlights = 8;                     /* LightController.h:95: 8 slots */
    if (nlights < 1) nlights = 1;
    int light_index = 0;
    orc_strike strike;
    memset(&strike, 0, sizeof(strike));

    for (;;) {
    while (distance_traveled < max_distance && bounce_count < 2) {        /* :357 */
        float t_exit = 0.0f;
        if (jump) {
            /* mode B: one node-exit jump = `steps` iterations of the loop; the iteration that lands in the new voxel
             * is this one, the steps before it only count (and may run into the :357 guard inside the node) */
            int old[3] = {voxel[0], voxel[1], voxel[2]};
            const int steps = jump_step(&js, rd, vstep, voxel, face_mask, &t_exit);
            if (distance_traveled + steps - 1 >= max_distance) {
                ctr->n_steps += (uint64_t)(max_distance - distance_traveled);
                distance_traveled = max_distance;
                voxel[0] = old[0]; voxel[1] = old[1]; voxel[2] = old[2];
                continue;
            }
            ctr->n_steps += (uint64_t)steps;
            distance_traveled += steps - 1;
        } else {
        ctr->n_steps++;
        /* :558 ties step several axes */
        face_mask[0] = it[0] <= fmin_cl(it[1], it[2]);
        face_mask[1] = it[1] <= fmin_cl(it[2], it[0]);
        face_mask[2] = it[2] <= fmin_cl(it[0], it[1]);
        for (int a = 0; a < 3; a++) {
            it[a] += delta_t[a] * (float)face_mask[a];            /* :559 */
            voxel[a] += vstep[a] * face_mask[a];                  /* :560 */
        }
        }
        /* :563-568 */
        if (voxel[0] >= s->map_dim[0] || voxel[1] >= s->map_dim[1] || voxel[2] >= s->map_dim[2] ||
            voxel[0] < 0 || voxel[1] < 0 || voxel[2] < 0) {
            for (int a = 0; a < 3; a++) voxel[a] -= vstep[a] * face_mask[a];
            float k = 1.0f - fmax_cl((float)distance_traveled / 700.0f, 0.0f);
            for (int c = 0; c < 4; c++) color_accumulator[c] = mixf(0.0f, voxel_color[c], k);
            color_accumulator[3] *= 4.0f;
            flags |= ORC_FLAG_OOB_EXIT;
            break;
        }
        if (svo) {
            voxel_data = svo_locate_from(&cur, voxel, jump) ? 5 : 0;
            if (jump) {
                if (voxel_data) jump_hit_intersection(&js, vstep, voxel, face_mask, t_exit, delta_t, it);
                jump_set_node(&js, voxel, voxel_data ? 1 : 1 << (cur.n - cur.top - 1));
            }
            /* extension: material from the attachment buffers the reference
             * allocates but never reads; only bottom-level descriptors carry materials */
            if (voxel_data && s->attachment_lookup && s->attachments && cur.top == cur.n - 1) {
                const uint64_t a = s->attachments[s->attachment_lookup[cur.idx[cur.top]]];
                const int k = (voxel[0] & 1) | ((voxel[1] & 1) << 1) | ((voxel[2] & 1) << 2);
                voxel_data = (int8_t)(a >> (8 * k));
            }
        } else {
            /* :569 -- note dim.z is the y-stride */
            voxel_data = s->map[(int64_t)voxel[0] + (int64_t)s->map_dim[0] * ((int64_t)voxel[1] + (int64_t)s->map_dim[2] * voxel[2])];
            ctr->n_map++;
        }

        if (voxel_data == 5 || voxel_data == 6) {                 /* :575 */
            face_position[0] = face_position[1] = face_position[2] = 0.0f;
            tile_face_position[0] = tile_face_position[1] = 0.0f;
            sign[0] = sign[1] = sign[2] = 1.0f;                   /* :582 comma expr -> 1.0f */

            if (face_mask[0] == 1) {                              /* :586-599 */
                sign[0] = (float)((double)sign[0] * -1.0);
                float z_percent = (it[2] - (it[0] - delta_t[0])) / delta_t[2];
                float y_percent = (it[1] - (it[0] - delta_t[0])) / delta_t[1];
                face_position[0] = 1.00001f; face_position[1] = y_percent; face_position[2] = z_percent;
                tile_face_position[0] = face_position[1]; tile_face_position[1] = face_position[2];
            } else if (face_mask[1] == 1) {                       /* :601-608 */
                sign[1] = (float)((double)sign[1] * -1.0);
                float x_percent = (it[0] - (it[1] - delta_t[1])) / delta_t[0];
                float z_percent = (it[2] - (it[1] - delta_t[1])) / delta_t[2];
                face_position[0] = x_percent; face_position[1] = 1.00001f; face_position[2] = z_percent;
                tile_face_position[0] = face_position[0]; tile_face_position[1] = face_position[2];
            } else if (face_mask[2] == 1) {                       /* :610-618 */
                sign[2] = (float)((double)sign[2] * -1.0);
                float x_percent = (it[0] - (it[2] - delta_t[2])) / delta_t[0];
                float y_percent = (it[1] - (it[2] - delta_t[2])) / delta_t[1];
                face_position[0] = x_percent; face_position[1] = y_percent; face_position[2] = 1.00001f;
                tile_face_position[0] = face_position[0]; tile_face_position[1] = face_position[1];
            }

            /* :626-643 quadrant flips */
            if (rd[0] > 0.0f) face_position[0] = -face_position[0] + 1.0f;
            if (rd[0] < 0.0f) tile_face_position[0] = -tile_face_position[0] + 1.0f;
            if (rd[1] > 0.0f) {
                face_position[1] = -face_position[1] + 1.0f;
            } else {
                tile_face_position[0] = (float)(1.0 - (double)tile_face_position[0]);
                if (face_mask[2] == 1) {
                    tile_face_position[0] = 1.0f - tile_face_position[0];
                    tile_face_position[1] = 1.0f - tile_face_position[1];
                }
            }
            if (rd[2] > 0.0f) face_position[2] = -face_position[2] + 1.0f;
            if (rd[2] < 0.0f) tile_face_position[1] = -tile_face_position[1] + 1.0f;

            const int tiles_x = s->atlas_dim[0] / s->tile_dim[0];   /* *atlas_dim / *tile_dim (int2) */
            const int tiles_y = s->atlas_dim[1] / s->tile_dim[1];

            if (hit && hit[ORC_HIT_MATERIAL] == 0 && !shadow_ray) {
                hit[ORC_HIT_VX] = voxel[0]; hit[ORC_HIT_VY] = voxel[1]; hit[ORC_HIT_VZ] = voxel[2];
                hit[ORC_HIT_MATERIAL] = voxel_data;
                hit[ORC_HIT_FACE] = face_mask[0] | (face_mask[1] << 1) | (face_mask[2] << 2);
            }

            if (voxel_data == 5 && !shadow_ray) {                 /* :649-679 */
                shadow_ray = 1;
                int tx = (int)(tile_face_position[0] * (float)tiles_x) + (int)(5.0f * (float)tiles_x);
                int ty = (int)(tile_face_position[1] * (float)tiles_y) + (int)(0.0f * (float)tiles_y);
                float texel[4];
                atlas_fetch(s, tx, ty, texel);
                ctr->n_tex++;
                for (int c = 0; c < 3; c++) voxel_color[c] += texel[c] / 2.0f;

                for (int a = 0; a < 3; a++) {
                    strike.voxel[a] = voxel[a];
                    strike.face_position[a] = face_position[a];
                    strike.nmask[a] = face_mask[a] * vstep[a];
                }
                strike.distance = distance_traveled;
                fog_distance = (float)distance_traveled;          /* :666 */
                if (!s->shadow_rays) {          /* extension: primary rays only; every active light shades */
                    for (int l = 0; l < nlights; l++)
                        (void)light_from_strike(s, s->lights + 10 * l, &strike, l == 0 ? voxel_color : color_accumulator,
                                                color_accumulator, rd, voxel, vstep, delta_t, it, &max_distance, 0);
                    break;
                }
                if (!light_from_strike(s, s->lights, &strike, voxel_color, color_accumulator, rd, voxel, vstep,
                                       delta_t, it, &max_distance, 1)) {
                    flags &= ~ORC_FLAG_WRITTEN;
                    goto done_unwritten;
                }
                ctr->shadow_rays++;
                flags |= ORC_FLAG_SHADOW_CAST;
                if (jump) {                   /* the restarted ray starts at hit_pos, in a voxel of unknown surroundings */
                    const float hp[3] = {(float)strike.voxel[0] + strike.face_position[0], (float)strike.voxel[1] + strike.face_position[1],
                                         (float)strike.voxel[2] + strike.face_position[2]};
                    jump_restart_ray(&js, hp, voxel, rd);
                    jump_set_node(&js, voxel, 1);
                }
            } else if (voxel_data == 6 && !shadow_ray) {          /* :682-704 */
                int tx = (int)(tile_face_position[0] * (float)tiles_x) + (int)(3.0f * (float)tiles_x);
                int ty = (int)(tile_face_position[1] * (float)tiles_y) + (int)(4.0f * (float)tiles_y);
                float texel[4];
                atlas_fetch(s, tx, ty, texel);
                ctr->n_tex++;
                for (int c = 0; c < 3; c++) voxel_color[c] += texel[c] / 4.0f;
                voxel_color[3] -= 0.0f;

                float hit_pos[3];
                for (int a = 0; a < 3; a++) {
                    hit_pos[a] = (float)voxel[a] + face_position[a];
                    rd[a] *= sign[a];                             /* :693 */
                }
                if (rd[0] == 0.0f || rd[1] == 0.0f || rd[2] == 0.0f) {
                    flags &= ~ORC_FLAG_WRITTEN;
                    goto done_unwritten;
                }
                for (int a = 0; a < 3; a++) {
                    voxel[a] -= vstep[a] * face_mask[a];          /* :697 */
                    /* :698 parses as (-1*(dir>0)) - (dir<0) with vector compares = -1:
                     * +1 for both signs                                              */
                    vstep[a] = (-1 * (rd[a] > 0.0f ? -1 : 0)) - (rd[a] < 0.0f ? -1 : 0);
                    delta_t[a] = fabsf(1.0f / rd[a]);
                    it[a] = delta_t[a] * (hit_pos[a] - floorf(hit_pos[a])) * (float)vstep[a];
                    it[a] += delta_t[a] * -(it[a] < 0.0f ? -1.0f : 0.0f);
                }
                bounce_count += 1;
                if (jump) {
                    jump_restart_ray(&js, hit_pos, voxel, rd);
                    jump_set_node(&js, voxel, 1);
                }
            } else {                                              /* :707-710 */
                color_accumulator[3] = 0.1f;
                flags |= ORC_FLAG_SHADOW_HIT;
                break;
            }
        }
        distance_traveled++;                                      /* :714 */
    }
        /* multi-light extension: the shadow ray of light l ended (step cap :357, left the map :563-568 or
         * blocked :707-710); reset to the first strike and run the light block for the next light, chaining
         * the colour.  Never taken with one active light.                                                   */
        if (!(shadow_ray && s->shadow_rays && light_index + 1 < nlights)) break;
        light_index++;
        if (!light_from_strike(s, s->lights + 10 * light_index, &strike, color_accumulator, color_accumulator, rd,
                               voxel, vstep, delta_t, it, &max_distance, 1)) {
            flags &= ~ORC_FLAG_WRITTEN;
            goto done_unwritten;
        }
        ctr->shadow_rays++;
        distance_traveled = strike.distance + 1;      /* as if the strike iteration had just finished (:714) */
        if (jump) {
            const float hp[3] = {(float)strike.voxel[0] + strike.face_position[0], (float)strike.voxel[1] + strike.face_position[1],
                                 (float)strike.voxel[2] + strike.face_position[2]};
            jump_restart_ray(&js, hp, voxel, rd);
            jump_set_node(&js, voxel, 1);
        }
    }

    {
        float k = 1.0f - fmax_cl(fog_distance / 700.0f, 0.0f);    /* :716 */
        float *out = image + 4 * pix;
        for (int c = 0; c < 4; c++) out[c] = mixf(0.0f, color_accumulator[c], k);
    }
    goto done;

done_unwritten:
    ctr->unwritten++;
done:
    if (svo) ndesc = cur.reads;
    ctr->n_desc += ndesc;
    if (hit) {
        hit[ORC_HIT_FLAGS] = flags | ((int)(bounce_count & 3) << ORC_FLAG_BOUNCE_SHIFT);
        hit[ORC_HIT_STEPS] = distance_traveled;
        hit[ORC_HIT_NDESC] = (int32_t)ndesc;
    }
}

static void add_counters(orc_counters *dst, const orc_counters *src) {
    dst->primary_rays += src->primary_rays;
    dst->shadow_rays += src->shadow_rays;
    dst->n_desc += src->n_desc;
    dst->n_tex += src->n_tex;
    dst->n_map += src->n_map;
    dst->n_steps += src->n_steps;
    dst->unwritten += src->unwritten;
}

void orc_raycast(const orc_scene *s, int32_t y0, int32_t y1, float *image, int32_t *hits,
                 orc_counters *counters, int threads) {
    /* frame-constant part of :342-354 */
    int32_t cam_voxel[3], bias[3];
    for (int a = 0; a < 3; a++) cam_voxel[a] = (int32_t)floorf(s->cam_pos[a]);
    orc_traversal_state ts;
    const desc_src src = scene_src(s);
    get_oct_vox_src(cam_voxel, &src, (uint64_t)s->octree_root_index, (int32_t)s->octree_dimensions, &ts);
    for (int a = 0; a < 3; a++)
        bias[a] = s->no_bias ? 0 : (ts.sub_oct_pos[a] - cam_voxel[a]) * ts.resolution / 2;

    const int W = s->resolution[0];
    orc_counters total;
    memset(&total, 0, sizeof(total));
    const int array_mode_reads = ts.reads;   /* array branch: get_oct_vox reads per pixel */
    const int svo = (s->using_octree == 0);

    /* pixels are independent: chunks of 32 go to the threads dynamically, so even a single row keeps every thread busy */
    const int64_t p0 = (int64_t)y0 * W, p1 = (int64_t)y1 * W;
#ifdef _OPENMP
    if (threads > 1) {
#pragma omp parallel num_threads(threads)
        {
            orc_counters local;
            memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 32)
            for (int64_t q = p0; q < p1; q++) {
                const int x = (int)(q % W), y = (int)(q / W);
                uint64_t before = local.primary_rays;
                raycast_pixel(s, x, y, bias, image, hits, &local);
                if (!svo && local.primary_rays != before) {
                    local.n_desc += (uint64_t)array_mode_reads;
                    if (hits) hits[8 * q + ORC_HIT_NDESC] = array_mode_reads;
                }
            }
#pragma omp critical
            add_counters(&total, &local);
        }
    } else
#endif
    {
        (void)threads;
        for (int64_t q = p0; q < p1; q++) {
            const int x = (int)(q % W), y = (int)(q / W);
            uint64_t before = total.primary_rays;
            raycast_pixel(s, x, y, bias, image, hits, &total);
            if (!svo && total.primary_rays != before) {
                total.n_desc += (uint64_t)array_mode_reads;
                if (hits) hits[8 * q + ORC_HIT_NDESC] = array_mode_reads;
            }
        }
    }
    if (counters) add_counters(counters, &total);
}

void orc_clear_image(float *image, int64_t n_pixels) {
    for (int64_t i = 0; i < n_pixels; i++) {
        image[4 * i + 0] = 1.0f;
        image[4 * i + 1] = 1.0f;
        image[4 * i + 2] = 1.0f;
        image[4 * i + 3] = 100.0f / 255.0f;
    }
}

void orc_image_to_rgba8(const float *image, uint8_t *out, int64_t n_pixels) {
    for (int64_t i = 0; i < 4 * n_pixels; i++) {
        float v = image[i];
        if (!(v > 0.0f)) v = 0.0f;          /* NaN and negatives saturate to 0 */
        if (v > 1.0f) v = 1.0f;
        out[i] = (uint8_t)lrintf(v * 255.0f);   /* default rounding mode: nearest even */
    }
}

/* ======================================================================= */
/* a7: Ray::Cast  (src/Ray.cpp:19-146)                                      */

static uint32_t rgba(int r, int g, int b, int a) {
    return (uint32_t)(r & 255) | ((uint32_t)(g & 255) << 8) | ((uint32_t)(b & 255) << 16) | ((uint32_t)(a & 255) << 24);
}

uint32_t orc_ray_cast(const int8_t *map, const int32_t dim_in[3], const float origin[3],
                      const float direction[3], int as_written, int32_t *steps_out) {
    int32_t dimensions[3] = {0, 0, 0};                            /* Ray.cpp:15-16 */
    if (!as_written) { dimensions[0] = dim_in[0]; dimensions[1] = dim_in[1]; dimensions[2] = dim_in[2]; }

    int voxel_step[3], voxel[3];
    float delta_t[3], it[3];
    for (int a = 0; a < 3; a++) {
        voxel_step[a] = isign(direction[a]);                      /* :22-25 */
        voxel[a] = (int)floorf(origin[a]);                        /* :28-32 */
        delta_t[a] = fabsf(1.0f / direction[a]);                  /* :36-40 */
        it[a] = delta_t[a];                                       /* :44-48 */
    }
    int dist = 0;
    int face = -1;
    const uint32_t sky = rgba(172, 245, 251, 200);
    do {
        if (it[0] < it[1]) {                                      /* :57-81 */
            if (it[0] < it[2]) { face = 0; voxel[0] += voxel_step[0]; it[0] = it[0] + delta_t[0]; }
            else               { face = 2; voxel[2] += voxel_step[2]; it[2] = it[2] + delta_t[2]; }
        } else {
            if (it[1] < it[2]) { face = 1; voxel[1] += voxel_step[1]; it[1] = it[1] + delta_t[1]; }
            else               { face = 2; voxel[2] += voxel_step[2]; it[2] = it[2] + delta_t[2]; }
        }
        if (steps_out) *steps_out = dist + 1;
        /* :84-103 (note the y test against dimensions.x) */
        if (voxel[2] >= dimensions[2]) return sky;
        if (voxel[0] >= dimensions[0]) return sky;
        if (voxel[1] >= dimensions[0]) return sky;
        if (voxel[0] < 0) return sky;
        if (voxel[1] < 0) return sky;
        if (voxel[2] < 0) return sky;

        int64_t index = (int64_t)voxel[0] + (int64_t)dimensions[0] * ((int64_t)voxel[1] + (int64_t)dimensions[2] * voxel[2]);
        int voxel_data = as_written ? 0 : map[index];             /* :107-108 */

        float alpha = 0;
        (void)face;
        alpha = (float)(fmod(alpha, 0.785) * 2);                  /* :111-127 */
        alpha *= 162;
        switch (voxel_data) {                                     /* :131-138 */
            case 5: return rgba(255, 120, 255, (int)alpha);
            case 6: return rgba(150, 80, 220, (int)alpha);
            default:
                if (as_written) return rgba(150, 80, 220, (int)alpha);
                if (voxel_data != 0) return rgba(150, 80, 220, (int)alpha);
                break;   /* restored variant: empty voxel keeps stepping */
        }
        dist++;
    } while (dist < 600);                                         /* :143 */
    return rgba(0, 255, 255, 255);                                /* sf::Color::Cyan */
}

int64_t orc_ray_cast_frame(const int8_t *map, const int32_t dim[3], int32_t width, int32_t height,
                           const float *viewport_matrix, const float cam_trig[4], const float cam_pos[3],
                           uint32_t *out, int threads) {
    int64_t total = 0;
    const float s1 = cam_trig[0], c1 = cam_trig[1], s2 = cam_trig[2], c2 = cam_trig[3];
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(+ : total)
#endif
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const float *pm = viewport_matrix + 4 * ((int64_t)x + (int64_t)width * y);
            float d[3];
            {   /* same pitch / yaw as the kernel (:280-291) so both CPU paths see the same rays */
                float px = pm[2] * s1 + pm[0] * c1, py = pm[1], pz = pm[2] * c1 - pm[0] * s1;
                d[0] = px * c2 - py * s2; d[1] = px * s2 + py * c2; d[2] = pz;
            }
            int32_t steps = 0;
            out[(int64_t)x + (int64_t)width * y] = orc_ray_cast(map, dim, cam_pos, d, 0, &steps);
            total += steps;
        }
    (void)threads;
    return total;
}
