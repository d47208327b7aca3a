/*
 * vrc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C restatement of the reference's raycast hot path
 * (MitchellHansen/voxel-raycaster):
 *   a1/a2  child-descriptor format + Octree::Generate   src/map/Octree.cpp:13-43,171-323
 *   a3     get_oct_vox / Octree::GetVoxel                kernels/ray_caster_kernel.cl:114-251
 *   a4     CLCaster::create_viewport ray table           src/CLCaster.cpp:233-275
 *   a5/a6  raycaster kernel, array branch + hit block    kernels/ray_caster_kernel.cl:256-357,555-721
 *   a7     Ray::Cast                                     src/Ray.cpp:19-146
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (libvrc.so) never
 * links or calls it.
 *
 * PINNING STATUS.  Pinned against the reference's own compiled code: get_oct_vox (a3) and view_light (the shading
 * arithmetic of a6) -- kernels/ray_caster_kernel.cl compiles unmodified for gfx950, oracle/ref_probe.cl includes it
 * from /root/reference and calls the two functions from probe kernels, tests/test_reference_pin_gpu.py compares them
 * with this file on the MI355X (get_oct_vox: every field equal; view_light: 99.98 % of cases within 1e-5 relative,
 * worst 3.1e-5 -- the OpenCL library's normalize/fast_length are approximate).
 * Checked against outputs of the reference's WHOLE raycaster kernel (a4 device part, a5, a6, epilogue): the kernel is
 * run on the MI355X with its two image builtins (write_imagef / read_imagef -- CDNA4 has no image hardware, they lower
 * to nothing: profiles/r01_reference_kernel_on_gfx950.txt) redirected by macro to stores of the kernel's own locals
 * (oracle/ref_raycaster_probe.cl); live in tests/test_reference_pin_gpu.py and, as committed vectors
 * tests/golden/ref_*.npz (generator tests/make_reference_golden.py), in tests/test_oracle_cpu.py.  Hit voxel, face,
 * material, texel fetches, bounce count, written/unwritten pixels, step count and colour of rays that hit nothing:
 * equal on every pixel of 8 scenes; final step count, shadow flag, RGB (1e-5) of shaded pixels: equal on 100 %.
 * Because two I/O builtins are overridden this counts as corroboration, not as a pin in the strict sense.
 * UNPINNED: the a2 builder, the host part of a4 (ray table) and a7 -- the reference ships no tests, golden vectors or
 * fixtures and its host code (Octree.cpp, CLCaster.cpp, Ray.cpp) needs SFML/OpenCL/GL headers and
 * libraries the image lacks (writing stand-ins is not allowed).  Those parts are a line-by-line restatement (each
 * block cites the reference line it follows) checked by (a) the reference's own self-check Octree::Validate
 * (src/map/Octree.cpp:329-352), (b) a second, independently written builder in the product (bit-identical output),
 * (c) Ray::Cast's constant known answer, (d) the reference kernel above accepting the oracle-built tree and table.
 * tests/golden/orc_*.npz are regression vectors produced by THIS oracle, not by the reference.
 * See DESIGN.md "Oracle and pinning".
 *
 * Float semantics: IEEE-754 binary32, no contraction (build with
 * -ffp-contract=off), correctly rounded / and sqrt.  sin/cos of the camera
 * angles are computed once on the host with libm and handed to the kernel
 * restatement (cam_trig), so a different libm cannot perturb parity.
 */
#ifndef VRC_ORACLE_H
#define VRC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_DEPTH 32

/* ---- a1: descriptor bit-fields (ray_caster_kernel.cl:49-54, Octree.h:89-94) */
#define ORC_CHILD_POINTER_MASK 0x0000000000007fffULL
#define ORC_FAR_BIT_MASK       0x8000ULL
#define ORC_VALID_MASK         0xFF0000ULL
#define ORC_LEAF_MASK          0xFF000000ULL

/* ---- a2: Octree::Generate.  Fills `buffer` (buffer_size entries, must be
 * zero-initialised by the caller) from the END downward exactly like the
 * reference; returns 0 on success, -1 if the buffer would underflow (the
 * reference has no such check and corrupts memory instead).
 * grid: char[dim^3], index x + dim*(y + dim*z)  (Octree.cpp:325-327).       */
int orc_octree_generate(const int8_t *grid, int dim, uint64_t *buffer,
                        uint64_t buffer_size, uint64_t *root_index,
                        uint64_t *lowest_used);

/* ---- a3: get_oct_vox / TraversalState (ray_caster_kernel.cl:114-251) */
typedef struct {
    int32_t  sub_oct_pos[3];
    int32_t  parent_stack_position;
    uint64_t parent_stack[ORC_MAX_DEPTH];
    uint64_t parent_stack_index[ORC_MAX_DEPTH];
    uint8_t  scale;
    uint8_t  idx_stack[ORC_MAX_DEPTH];
    uint64_t current_descriptor;
    uint64_t current_descriptor_index;
    int32_t  oct_pos[3];
    int32_t  resolution;
    int8_t   found;
    int32_t  reads;              /* descriptor loads made (oracle counter) */
} orc_traversal_state;

void orc_get_oct_vox(const int32_t position[3], const uint64_t *descriptors,
                     uint64_t root_index, int32_t dim, orc_traversal_state *ts);

/* Octree::Validate (Octree.cpp:329-352): returns the number of voxels where
 * (grid != 0) != (get_oct_vox(found) != 0).                                  */
int64_t orc_octree_validate(const int8_t *grid, int dim, const uint64_t *descriptors,
                            uint64_t root_index);

/* ---- a4: CLCaster::create_viewport (CLCaster.cpp:233-275).  table is
 * float4[w*h], zero-initialised here; w component = 0.                       */
void orc_create_viewport(int32_t w, int32_t h, float *table);

/* Descriptors of a tree that lives only in GPU memory reach the oracle in pages: the callback returns a pointer to
 * descriptors [page * ORC_PAGE_SIZE, (page + 1) * ORC_PAGE_SIZE) that stays valid for the whole orc_raycast call
 * (the tail of the last page may be padding).  Called under a lock, from any OpenMP thread.                     */
#define ORC_PAGE_SHIFT 12
#define ORC_PAGE_SIZE  (1u << ORC_PAGE_SHIFT)
typedef const uint64_t *(*orc_page_fetch_fn)(void *user, uint64_t page);

/* ---- a5/a6: the kernel */
typedef struct {
    /* kernel args 0..15 (CLCaster.cpp:186-202) */
    const int8_t   *map;              /* 0  char[dx*dy*dz], may be NULL when using_octree == 0 */
    int32_t         map_dim[3];       /* 1  */
    int32_t         resolution[2];    /* 2  */
    const float    *viewport_matrix;  /* 3  float4[w*h] */
    float           cam_dir[2];       /* 4  */
    float           cam_pos[3];       /* 5  */
    const float    *lights;           /* 6  10 floats per light: rgbi[4] pos[3] dir[3] */
    int32_t         light_count;      /* 7  (bound, unused by the reference) */
    const uint8_t  *atlas_rgba8;      /* 9  */
    int32_t         atlas_dim[2];     /* 10 */
    int32_t         tile_dim[2];      /* 11 */
    const uint64_t *descriptors;      /* 12 */
    uint64_t        n_descriptors;
    const uint32_t *attachment_lookup;   /* 13 optional: per-descriptor slot into attachments (layout: include/vrc.h) */
    const uint64_t *attachments;         /* 14 optional: 8 int8 materials per bottom-level descriptor; NULL => material 5 */
    /* settings buffer (CLCaster.cpp:1029-1109, Application.cpp:35-39) */
    int64_t         octree_dimensions;   /* OCTDIM */
    int64_t         using_octree;        /* OCTENABLED: 0 => SVO occupancy, !=0 => dense array */
    int64_t         octree_root_index;   /* OCTREE_ROOT_INDEX */
    /* extensions; the defaults reproduce the reference */
    int32_t         max_distance;        /* ray_caster_kernel.cl:326 => 20 */
    int32_t         shadow_rays;         /* 1; 0 = shade the primary hit and stop */
    int32_t         no_bias;             /* 0: the reference (:353-354); 1: extension, drop the octree bias term */
    int32_t         active_lights;       /* <= 1: light 0 only (the reference); n: the first n lights, each
                                            from the first strike (see light_from_strike)       */
    float           cam_trig[4];         /* sin(dir.x) cos(dir.x) sin(dir.y) cos(dir.y) */
    int32_t         stepping_mode;       /* 0: the reference's per-voxel DDA (:558-560).  1: "mode B" (SURVEY D1), SVO only:
                                            stateless node-exit jumps, see raycast_pixel / jump_step in vrc_oracle.c   */
    int32_t         coarse_log2;         /* mode B only (round 4): the levels above this one are a dense table in the product
                                            (raycast_jump_kernel.hip coarse_build_kernel); -1: by depth (n >= 5: min(n - 2, 9),
                                            else none) and tree size (the finest such level whose table is at most 16 x the descriptor
                                            array or below 1 MiB), 0: none.  Changes the descriptor-read count only, see svo_locate     */
    /* optional paged descriptor source (see ORC_PAGE_SIZE): when desc_page_fetch is set, `descriptors` is not read;
     * desc_pages is a zero-initialised table of ceil(n_descriptors / ORC_PAGE_SIZE) pointers the oracle fills */
    const uint64_t  **desc_pages;
    orc_page_fetch_fn desc_page_fetch;
    void             *desc_page_user;
} orc_scene;

typedef struct {
    uint64_t primary_rays;   /* pixels passing the zero-component test (:293) */
    uint64_t shadow_rays;    /* redirects toward a light (:670-679) */
    uint64_t n_desc;         /* canonical descriptor reads */
    uint64_t n_tex;          /* atlas texels fetched */
    uint64_t n_map;          /* dense-grid bytes fetched */
    uint64_t n_steps;        /* DDA iterations */
    uint64_t unwritten;      /* pixels left untouched (early returns) */
} orc_counters;

/* hit record: 8 x int32 per pixel */
enum {
    ORC_HIT_VX = 0, ORC_HIT_VY, ORC_HIT_VZ,   /* primary-hit voxel, -1 if none */
    ORC_HIT_MATERIAL,                         /* 5 / 6 / 0 */
    ORC_HIT_FACE,                             /* bit0 x, bit1 y, bit2 z */
    ORC_HIT_FLAGS,
    ORC_HIT_STEPS,                            /* final distance_traveled */
    ORC_HIT_NDESC                             /* canonical descriptor reads of this pixel */
};
#define ORC_FLAG_WRITTEN     1
#define ORC_FLAG_SHADOW_CAST 2
#define ORC_FLAG_SHADOW_HIT  4
#define ORC_FLAG_OOB_EXIT    8
#define ORC_FLAG_BOUNCE_SHIFT 4   /* 2 bits */

/* view_light (ray_caster_kernel.cl:78-99) on its own, for pinning against the reference's compiled function
 * (tests/test_reference_pin_gpu.py)                                           */
void orc_view_light(float out[4], const float in_color[4], const float light[3], const float light_color[4],
                    const float view[3], const int32_t mask[3]);

void orc_camera_trig(const float cam_dir[2], float trig[4]);

/* Render rows [y0,y1).  image: float4[w*h] (only written pixels are touched,
 * like write_imagef); hits: int32[8*w*h] or NULL; counters accumulated (may be
 * NULL).  threads <= 1 runs single-threaded, else OpenMP over rows.          */
void orc_raycast(const orc_scene *s, int32_t y0, int32_t y1, float *image,
                 int32_t *hits, orc_counters *counters, int threads);

/* the image the reference starts from: RGBA8 (255,255,255,100)
 * (CLCaster.cpp:280-286) as normalised floats                               */
void orc_clear_image(float *image, int64_t n_pixels);

/* write_imagef to a UNORM8 target: saturate, *255, round-to-nearest-even     */
void orc_image_to_rgba8(const float *image, uint8_t *out, int64_t n_pixels);

/* ---- a7: Ray::Cast (Ray.cpp:19-146).  as_written=1 reproduces the stubbed
 * file (dimensions (0,0,0), voxel_data 0): constant colour.  as_written=0 is
 * the "restored" variant (real dimensions + grid lookup, empty voxels keep
 * stepping) used as a primary-ray CPU DDA baseline.  Returns RGBA packed
 * r | g<<8 | b<<16 | a<<24.                                                  */
uint32_t orc_ray_cast(const int8_t *map, const int32_t dim[3], const float origin[3],
                      const float direction[3], int as_written, int32_t *steps_out);

/* Ray::Cast over a whole frame (one primary ray per pixel, rays = the a4 table rotated like :276-291),
 * restored variant, OpenMP over rows: the "src/Ray.cpp CPU path" BASELINE.json names as the CPU comparison.
 * out: packed RGBA per pixel.  Returns the total number of DDA steps.                                       */
int64_t orc_ray_cast_frame(const int8_t *map, const int32_t dim[3], int32_t width, int32_t height,
                           const float *viewport_matrix, const float cam_trig[4], const float cam_pos[3],
                           uint32_t *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
