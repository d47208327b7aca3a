/*
 * vrc.h -- C ABI of the MI355X-native raycast path ("libvrc.so").
 *
 * Drop-in boundary for the reference's CLCaster subsystem
 * (include/CLCaster.h:93-329, src/CLCaster.cpp).  Every entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; no C++,
 * torch, OpenCL or GL types cross this boundary.
 *
 * Conventions
 *   - every function returns a vrc_status (0 = ok); the reference returns
 *     bool and logs through Logger (CLCaster.cpp:1011-1017).  Details of the
 *     last failure: vrc_last_error().  Nothing aborts or throws.
 *   - scene buffers are COPIED at call time (reference: CL_MEM_COPY_HOST_PTR,
 *     CLCaster.cpp:893-896); camera, lights and light_count are RETAINED
 *     pointers re-read on every vrc_compute (reference: CL_MEM_USE_HOST_PTR,
 *     CLCaster.cpp:137-139,322) -- the caller keeps them alive.
 *   - a handle is not thread-safe (the reference is single-threaded, SURVEY
 *     8b): one host thread at a time per handle.  DIFFERENT handles may be
 *     driven from different host threads, also when they hold one tree
 *     (vrc_assign_octree_from): what is derived from a tree is built once under
 *     the tree's guard, and a frame holds that guard from the moment it reads
 *     the tree's pointers until its kernel is enqueued.  Holders of one tree
 *     should agree on octree_root_index, octree_dimensions and coarse_log2: if
 *     they differ every frame rebuilds the table and the boxes for ITS settings
 *     (correct, but 0.6 s per frame at depth 12).
 *     A handle from vrc_create drives ONE GPU; multi-GPU is either one
 *     such handle per GPU and process, each rendering its own row slice
 *     (vrc_set_row_slice), or ONE group handle from vrc_create_group that
 *     drives n GPUs from one host thread -- the reference's single synchronous
 *     compute() (CLCaster.cpp:224-228,946-987) spread over the row bands of n
 *     devices.  No collective either way: the SVO is replicated, tiles are
 *     copied out by the GPU that rendered them.
 *   - the GL-shared output texture of the reference (CLCaster.cpp:278-296) is
 *     replaced by an offline float4 pixel buffer in HBM, read back with
 *     vrc_read_image_*.
 */
#ifndef VRC_H
#define VRC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrc_caster vrc_caster;

typedef enum vrc_status {
    VRC_OK = 0,
    VRC_ERR_INVALID_ARGUMENT = 1,
    VRC_ERR_NOT_READY = 2,      /* validate()/compute() before the required assign_* calls */
    VRC_ERR_DEVICE = 3,         /* a HIP call failed; see vrc_last_error */
    VRC_ERR_OUT_OF_MEMORY = 4,
    VRC_ERR_NOT_FOUND = 5,      /* unknown setting name */
    VRC_ERR_LIMIT = 6           /* > 64 settings, octree deeper than the kernel stack, ... */
} vrc_status;

/* ---- lifecycle ------------------------------------------------------- */

/* CLCaster::CLCaster + CLCaster::init (CLCaster.cpp:14-74).  The interactive
 * device prompt / device_config.bin (:23-48,495-542) is replaced by
 * device_ordinal.  Fails with VRC_ERR_DEVICE when no HIP device is present:
 * there is no CPU fallback.                                                  */
int vrc_create(int device_ordinal, vrc_caster **out);
/* CLCaster::~CLCaster (CLCaster.cpp:5-12); unlike the reference this frees all
 * device memory.                                                            */
int vrc_destroy(vrc_caster *h);
/* Logger::log(..., ERROR) + cl_err_lookup (CLCaster.cpp:1112-1310) */
const char *vrc_last_error(const vrc_caster *h);
/* aquire_hardware (CLCaster.cpp:430-493): number of usable GPUs */
int vrc_device_count(int *count);

/* ---- scene buffers (copied) ------------------------------------------- */

/* CLCaster::assign_map (CLCaster.cpp:76-87): "map" char[dx*dy*dz] indexed
 * x + dx*(y + dz*z) by the kernel (ray_caster_kernel.cl:569), "map_dimensions". */
int vrc_assign_map(vrc_caster *h, const int8_t *voxels, int32_t dx, int32_t dy, int32_t dz);
/* CLCaster::release_map (CLCaster.cpp:89-99) */
int vrc_release_map(vrc_caster *h);

/* CLCaster::assign_octree (CLCaster.cpp:102-116): "octree_descriptor_buffer"
 * (n 64-bit child descriptors, format of include/map/Octree.h:89-94) and the
 * "octree_root_index" setting.  The two attachment buffers the reference
 * uploads but never reads (ray_caster_kernel.cl:143-144) are optional.        */
int vrc_assign_octree(vrc_caster *h, const uint64_t *descriptors, uint64_t n_descriptors,
                      uint64_t root_index);
/* Attachment layout (ours; the reference never defined one): lookup[i] (parallel to
 * the descriptor array) = slot in `attachments` of bottom-level descriptor i, whose
 * byte k is the int8 material of child k.  With attachments assigned the SVO branch
 * renders exactly what the array branch renders for the same grid (materials 5 and 6
 * are solid to the renderer, everything else is passed through,
 * ray_caster_kernel.cl:575); without them every valid voxel is material 5.        */
int vrc_assign_octree_attachments(vrc_caster *h, const uint32_t *lookup, uint64_t n_lookup,
                                  const uint64_t *attachments, uint64_t n_attachments);
/* CLCaster::assign_octree for a tree ANOTHER handle on the same GPU already holds (CLCaster.cpp:102-116 uploads the same
 * array again for every caster and never frees it, :5-12): h adopts src's descriptor array, its materials and everything
 * the kernels derive from them (the dense table of the tree's top, the empty boxes) -- they exist once, and live until the
 * last handle that holds them releases them or is destroyed.  Materials assigned later through either handle are seen by
 * both from their next frame on.                                                                                          */
int vrc_assign_octree_from(vrc_caster *h, vrc_caster *src);
/* CLCaster::release_octree (CLCaster.cpp:119-131) */
int vrc_release_octree(vrc_caster *h);

/* CLCaster::create_viewport (CLCaster.cpp:233-299): builds the float4 ray table
 * exactly like the reference (v_fov / h_fov are accepted and ignored, as there)
 * and the output image, initialised to RGBA8 (255,255,255,100).              */
int vrc_create_viewport(vrc_caster *h, int32_t width, int32_t height, float v_fov, float h_fov);
/* Extension: same buffers as vrc_create_viewport but the float4[width*height]
 * ray table ("viewport_matrix", CLCaster.cpp:242-275) is supplied by the host,
 * e.g. a supersampled or differently projected lens.                          */
int vrc_create_viewport_table(vrc_caster *h, int32_t width, int32_t height, const float *table);
/* CLCaster::release_viewport (CLCaster.cpp:301-311) */
int vrc_release_viewport(vrc_caster *h);

/* CLCaster::create_texture_atlas (CLCaster.cpp:208-222): the sf::Texture is
 * replaced by its RGBA8 pixels.                                              */
int vrc_create_texture_atlas(vrc_caster *h, const uint8_t *rgba8, int32_t width, int32_t height,
                             int32_t tile_w, int32_t tile_h);

/* ---- live buffers (retained pointers) --------------------------------- */

/* CLCaster::assign_camera (CLCaster.cpp:133-143): direction = 2 floats
 * (inclination, azimuth), position = 3 floats.                              */
int vrc_assign_camera(vrc_caster *h, const float *direction2, const float *position3);
/* The sin / cos of the two camera angles, supplied by the host: trig4 = { sin(direction.x), cos(direction.x),
 * sin(direction.y), cos(direction.y) }, the four values the reference kernel evaluates per pixel with the OpenCL
 * library's sin / cos (ray_caster_kernel.cl:280-291).  RETAINED like the camera itself and re-read on every
 * vrc_compute; NULL (the default) = the library evaluates sinf / cosf of the live direction on the host once per frame
 * (SURVEY D2).  A host that evaluates them the way its reference build did -- or replays stored values, as
 * tests/test_reference_vectors_gpu.py does with the reference kernel's own -- gets exactly that build's rays: the two
 * libraries' results differ in the last bit for some angles, and one ulp of a direction is another ray table.
 * vrc_release_camera drops the pointer with the camera.                                                              */
int vrc_assign_camera_trig(vrc_caster *h, const float *trig4);
/* CLCaster::release_camera (CLCaster.cpp:145-155) */
int vrc_release_camera(vrc_caster *h);
/* CLCaster::assign_lights (CLCaster.cpp:313-328): packed = 10 floats per light
 * {rgbi[4], position[3], direction[3]} (include/LightController.h:63-73), 8
 * reserved slots in the reference; light_count points at the live count.
 * The reference kernel shades with light 0 only whatever *light_count holds
 * (ray_caster_kernel.cl:264,660-670) and so does this library by default.
 * Extension: setting "light_count" = n > 1 shades with the first
 * min(n, *light_count, 8) lights, each shadow ray restarted from the primary
 * hit ("first-strike resetting", TODO src/main.cpp:33).  */
int vrc_assign_lights(vrc_caster *h, const float *packed, const int32_t *light_count);

/* ---- settings buffer --------------------------------------------------- */

/* CLCaster::add_to_settings_buffer (CLCaster.cpp:1029-1064): appends a slot
 * (max 64, include/CLCaster.h:303).  `define` is the kernel-side macro name the
 * reference passes as -D<define>=<slot>; kept for API fidelity.  Names the
 * kernel understands:
 *   octree_dimensions  (OCTDIM)            required in SVO mode
 *   using_octree       (OCTENABLED)        0 => occupancy from the SVO, != 0 => dense map
 *   octree_root_index  (OCTREE_ROOT_INDEX) set by vrc_assign_octree
 * extensions (defaults reproduce the reference):
 *   max_distance (20: the :325/:357 step cap; |value| <= 2^30, the counter is the reference's int)
 *   shadow_rays (1)  light_count (1, see vrc_assign_lights)
 *   octree_bias (1: the :353-354 term as in the reference; 0: without it)
 *   hit_records (1: the 8 x int32 record per pixel behind vrc_read_hits; 0: none -- the reference has none)
 *   stepping_mode (0: exact per-voxel DDA, bit-identical to the array branch; 1: node-exit jumps, DESIGN.md "mode B")
 *   coarse_log2 (-1: by depth = min(depth - 2, 9) from depth 5, 10 from depth 14; 0: none; k: 2^k cells per axis, at most
 *     min(depth - 2, 10)): the levels of the tree above level k as a dense table in HBM (8 << 3k bytes: 1 GB at k = 9), built on
 *     the device from the descriptor array by vrc_prepare / vrc_validate (or by the first vrc_compute that finds it missing).
 *     Both SVO kernels read it instead of descending from the root; frames, hit records and the exact mode's counters are the same with and without it
 *   empty_boxes (-1: by the tree's size; 0: never -- the canonical traversal, canonical read counts; 1: a box word per descriptor and
 *     child, trees below 2^31 descriptors; 2: box records for the upper levels only -- any tree): an empty node is widened to the
 *     empty box a device-side pass over the descriptor array found around it (vrc_prepare builds the words); frames and hit
 *     records unchanged but for the read count (vrc_counters_canonical).  By default a word per descriptor up to 2^29 descriptors
 *     (32 bytes each), beyond that records for as many upper levels as empty_box_records (default 400 M: 36 bytes each, 52 while
 *     building; at most a quarter of the free device memory) reach; empty_box_levels caps the levels
 *   jump_tables_lds (2: the Euclid tables of the exact closed-form jumps live in LDS when they fit at full occupancy;
 *     1 / 0: always / never)   jump_min_run (closed-form jumps for runs of at least this many iterations; 1 << 24: off)
 * Settings stay live after vrc_validate (overwrite_setting needs no recompile); structural ones are re-checked by
 * every vrc_compute, which fails with an error code instead of launching on a bad value.                            */
int vrc_setting_add(vrc_caster *h, const char *name, const char *define, int64_t value);
/* CLCaster::overwrite_setting (CLCaster.cpp:1087-1109) */
int vrc_setting_set(vrc_caster *h, const char *name, int64_t value);
int vrc_setting_get(vrc_caster *h, const char *name, int64_t *value);

/* ---- validate / compute ------------------------------------------------ */

/* CLCaster::validate (CLCaster.cpp:157-206): checks that camera, map/octree,
 * viewport, lights and atlas are present and consistent.  No runtime
 * compilation happens: the kernels are prebuilt gfx950 code.  Where the
 * reference pays its one-off cost (the kernel build, :157-206) this call pays
 * ours: it ends with vrc_prepare, so the first vrc_compute costs a frame.      */
int vrc_validate(vrc_caster *h);
/* Builds what the SVO kernels derive from the tree for the current settings -- the dense table of the tree's top (setting
 * coarse_log2) and the empty boxes (setting empty_boxes): 15 ms + 0.6 s for the 20 M descriptors of a 4096^3 terrain,
 * 2.6 s at 8192^3, 10 s for the 345 M descriptors of a 16384^3 terrain or the 400 M box records of a deeper one -- so that no
 * frame has to.  Needs the octree and the settings octree_dimensions / using_octree;
 * vrc_validate calls it.  Both structures are optional: a failed allocation is not an error (the frames are rendered by the
 * kernels without them, vrc_memory_usage2().note says why).  A vrc_compute that finds them missing or built for other
 * settings (root, depth, level changed after validate) still builds them itself, synchronously, inside that call -- also
 * vrc_compute_async: it then blocks for the build.                                                                      */
int vrc_prepare(vrc_caster *h);

/* CLCaster::compute (CLCaster.cpp:224-228) -> run_kernel (:946-987).
 * Synchronous like the reference's clFinish (:970).                          */
int vrc_compute(vrc_caster *h);
/* Enqueue one frame on the handle's stream without waiting / wait for it.    */
int vrc_compute_async(vrc_caster *h);
int vrc_sync(vrc_caster *h);

/* Multi-GPU row tiling: this handle renders only the bands
 * b with b % world == rank, a band being `band_rows` consecutive image rows
 * (multiple of 8).  Default rank 0 / world 1 = the whole image.  The buffers
 * stay full-frame (rows of other ranks keep their previous contents).        */
int vrc_set_row_tiling(vrc_caster *h, int32_t rank, int32_t world, int32_t band_rows);
/* Same partition, but the ray table, the frame and the hit records hold ONLY this rank's rows (SURVEY 8e "sliced per
 * GPU: ray table rows and framebuffer rows"): per-rank memory and uploads are 1/world of the frame.  Call before
 * vrc_create_viewport(_table); the read-back calls then fill this rank's rows of the caller's full frame and leave the
 * others alone, so world handles reading into one buffer assemble the frame.                                        */
int vrc_set_row_slice(vrc_caster *h, int32_t rank, int32_t world, int32_t band_rows);

/* One host thread, n GPUs (SURVEY 8b "Compute", 8e): the returned handle is rank 0 of a group of n handles
 * (device_ordinals[r] for rank r; the same ordinal may repeat), row-sliced in bands of band_rows.  Every call on the
 * group handle is replicated on all ranks: the octree is uploaded (or built) once on rank 0's GPU and fanned out
 * device to device (hipMemcpyPeerAsync; ranks on rank 0's own GPU share its arrays), vrc_compute launches every rank
 * and returns when the frame is complete on all of them -- the one synchronous compute() of CLCaster.cpp:224-228,946-987
 * -- vrc_read_* gather every rank's rows into the caller's frame (each GPU copies its own tile, no collective),
 * vrc_get_counters sums the ranks.  vrc_destroy on the group handle destroys all ranks.                            */
int vrc_create_group(const int32_t *device_ordinals, int32_t n, int32_t band_rows, vrc_caster **out);
/* The same with flags.  VRC_GROUP_OWN_COPIES: ranks that sit on rank 0's own GPU do not share its arrays but take the
 * path a rank on another GPU takes -- own allocation, hipMemcpyPeerAsync, attachment re-copy, release -- so that the
 * multi-GPU code can be exercised on a machine with one GPU (vrc_create_group honours VRC_GROUP_OWN_COPIES=1 in the
 * environment).  After a successful creation vrc_last_error() names the ranks whose GPU has no direct peer access to
 * rank 0's (their copies are staged by the runtime); it is empty otherwise.                                          */
#define VRC_GROUP_OWN_COPIES 1u
int vrc_create_group_ex(const int32_t *device_ordinals, int32_t n, int32_t band_rows, uint32_t flags, vrc_caster **out);
int vrc_group_size(const vrc_caster *h, int32_t *n);
/* Pin / unpin a caller-owned frame buffer (hipHostRegister): a group then copies every rank's tile straight into it.
 * A pageable destination works too -- each rank stages its tile through a pinned buffer of its own so that the n copies
 * still run side by side, at the price of one more host copy.                                                        */
int vrc_pin_host_buffer(void *p, size_t bytes);
int vrc_unpin_host_buffer(void *p);
/* Device memory held by one rank of a group (rank 0 = the handle itself). */
typedef struct vrc_memory {
    int32_t  device, rows;            /* GPU ordinal; rows of the frame this rank's buffers hold */
    uint64_t viewport_bytes, image_bytes, hit_bytes, octree_bytes;
    int32_t  octree_shared;           /* 1: the arrays belong to rank 0 (same GPU) */
    int32_t  peer_access;             /* -1: rank 0 or on rank 0's GPU; 1: direct peer access to rank 0's GPU; 0: staged by the runtime */
    uint64_t coarse_bytes;            /* the dense table of the tree's top (setting coarse_log2), 0 before the first frame */
} vrc_memory;
int vrc_memory_usage(vrc_caster *h, int32_t rank, vrc_memory *out);
/* The same with a size-versioned struct: set struct_size = sizeof(vrc_memory2) before the call; the library writes no more
 * than that many bytes and stores how many it wrote, so a host compiled against an older header keeps working when fields
 * are appended.  (vrc_memory grew by coarse_bytes in round 4 without such a field: hosts compiled against the round-3 header
 * must be recompiled or move to this call.)                                                                              */
typedef struct vrc_memory2 {
    uint32_t struct_size;             /* in: sizeof the caller's struct; out: bytes written */
    int32_t  device, rows;
    int32_t  octree_shared;           /* 1: the tree is also held by another handle (group rank on the same GPU, vrc_assign_octree_from) */
    int32_t  peer_access;
    int32_t  tree_holders;            /* handles that hold this rank's tree: its arrays, its coarse table and its boxes exist ONCE */
    int32_t  coarse_log2;             /* level of the dense table of the tree's top, 0: none */
    int32_t  empty_boxes;             /* 1: this rank's last frame was rendered with the tree's empty boxes (setting empty_boxes) */
    uint64_t viewport_bytes, image_bytes, hit_bytes, octree_bytes;
    uint64_t coarse_bytes, box_bytes; /* per TREE, not per handle */
    double   box_build_seconds;
    char     note[160];               /* why an optional structure is missing (allocation failure), else empty */
    uint64_t box_queries_cut;         /* region queries of the boxes' build that gave up at their budget of 4096 descents: the
                                         boxes next to them are smaller than they could be (never wrong); 0 on every BASELINE scene */
    uint64_t box_records;             /* descriptors that carry box words: all of them, or those of the upper box_levels levels */
    int32_t  box_levels, reserved_;   /* levels of the tree (from the root) whose descriptors have box records */
} vrc_memory2;
int vrc_memory_usage2(vrc_caster *h, int32_t rank, vrc_memory2 *out);

/* Self-check of the empty boxes the last frame was rendered with (setting empty_boxes; a derived structure like the coarse
 * table, no counterpart in the reference: Octree.cpp never annotates empty space): `samples` pseudo-random (descriptor,
 * child) pairs are drawn, for each empty child one voxel of its box is looked up in the tree.  *solid_voxels must come back
 * 0.  build_seconds: device time of the boxes' construction.  Any out pointer may be NULL.                              */
int vrc_empty_boxes_check(vrc_caster *h, uint64_t samples, uint64_t seed, uint64_t *boxes_sampled, uint64_t *solid_voxels,
                          double *build_seconds);
/* The box words themselves: out[8 * i + k] belongs to child slot k = x | y<<1 | z<<2 of descriptor first_descriptor + i and
 * means something where that child is empty (not valid): six 5-bit extents, bits 0-4 -x, 5-9 -y, 10-14 -z, 15-19 +x, 20-24 +y,
 * 25-29 +z, in units of the child's own size; code c stands for c (c < 4), (4 | c & 3) << (c / 4 - 1) otherwise.  The box is
 * the child's cube widened by those extents, clamped to the map.  (tests/test_boxes_gpu.py checks every voxel of every box of
 * small trees against the dense grid on the host.)                                                                         */
int vrc_read_empty_boxes(vrc_caster *h, uint64_t first_descriptor, uint64_t count, uint32_t *out);

/* ---- output ------------------------------------------------------------ */

/* CLCaster::draw (CLCaster.cpp:330-332) has no read-back; these replace it.
 * n_floats / n_bytes / n_int32 are the capacities of the caller's buffers.   */
int vrc_read_image_f32(vrc_caster *h, float *rgba, size_t n_floats);
/* What CLCaster::draw shows: the reference's output IS an RGBA8 texture (CLCaster.cpp:278-296,330-332).  The frame is
 * quantised on the device (saturate, x255, round to nearest even, like write_imagef to CL_UNORM_INT8) and 4 bytes per
 * pixel are read back.                                                                                              */
int vrc_read_image_rgba8(vrc_caster *h, uint8_t *rgba, size_t n_bytes);
/* 8 int32 per pixel: voxel x,y,z of the primary hit (-1 if none), material,
 * face bits, flags, final step count, descriptor reads -- SURVEY 8(d)'s canonical
 * count when the frame was rendered by the canonical traversal (setting
 * empty_boxes = 0, trees without a coarse table, the array branch); with the
 * tree's empty boxes (the default where they exist) the reads THAT traversal
 * made: fewer, it skips nodes.  vrc_memory_usage2().empty_boxes says which.   */
int vrc_read_hits(vrc_caster *h, int32_t *hits, size_t n_int32);
#define VRC_HIT_FLAG_WRITTEN     1
#define VRC_HIT_FLAG_SHADOW_CAST 2
#define VRC_HIT_FLAG_SHADOW_HIT  4
#define VRC_HIT_FLAG_OOB_EXIT    8

/* Device pointers of the resident frame buffers (float4[w*h], int32[8*w*h]);
 * lets a host that owns the GPU (e.g. a torch process) consume the frame
 * without a PCIe round trip.                                                 */
int vrc_device_image(vrc_caster *h, void **dev_ptr, size_t *n_bytes);

/* Counters of the most recent frame (device-side, fetched at sync). */
typedef struct vrc_counters {
    uint64_t primary_rays;
    uint64_t shadow_rays;
    uint64_t descriptor_reads;   /* SURVEY 8(d)'s canonical count, or the box traversal's own (see vrc_read_hits) */
    uint64_t texel_reads;
    uint64_t map_reads;
    uint64_t steps;
    uint64_t unwritten_pixels;
    uint64_t watchdog_trips;     /* wavefronts the kernel's round watchdog had to stop.  Always 0; anything else is a
                                    kernel bug: vrc_get_counters then returns VRC_ERR_DEVICE, the frame is invalid */
} vrc_counters;
int vrc_get_counters(vrc_caster *h, vrc_counters *out);
/* Which count descriptor_reads (and field 7 of the hit records) of the most recent frame is: *canonical = 1 when it is SURVEY
 * 8(d)'s canonical count -- the array branch, trees without a coarse table, setting empty_boxes = 0 -- and 0 when the frame was
 * rendered with the tree's empty boxes and the fields hold that traversal's own, smaller count.  So that one frame's records
 * describe themselves (a host that compares them with the oracle or the reference reads this instead of guessing from settings). */
int vrc_counters_canonical(vrc_caster *h, int32_t *canonical);

/* Wave-scheduler statistics of the SVO kernel for the most recent frame (per wave, not per lane):
 * [0] step-loop iterations, [1] bursts, [2] node-event passes, [3] lanes serviced in them,
 * [4] hit-block passes, [5] lanes shaded in them, [6..7] reserved.                */
int vrc_get_scheduler_stats(vrc_caster *h, uint64_t out[8]);

/* hipEvent timing of the raycast kernel on the handle's stream, accumulated
 * since the last reset (replaces GraphTimer "Compute", Application.cpp:151-155). */
int vrc_timing_reset(vrc_caster *h);
int vrc_timing_get(vrc_caster *h, uint64_t *n_launches, double *total_kernel_ms);

/* ---- SVO construction (host side) -------------------------------------- */

/* Octree::Generate (src/map/Octree.cpp:13-43,171-323): bottom-up build of the
 * descriptor array from a dense char[dim^3] grid.  buffer_size == 0 sizes the
 * array exactly; otherwise the array has buffer_size entries filled from the
 * end like the reference's fixed 100000-entry buffer (Octree.h:29).
 * strict_reference != 0 reproduces the reference layout bit for bit including
 * its far-pointer corner cases; 0 emits the same format with those fixed.
 * The result is malloc'ed; release with vrc_free.                            */
int vrc_octree_generate(const int8_t *grid, uint32_t dim, uint64_t buffer_size,
                        int strict_reference, uint64_t **descriptors,
                        uint64_t *n_descriptors, uint64_t *root_index);

/* Material attachments (layout: see vrc_assign_octree_attachments) for an array built from `grid`.
 * Results are malloc'ed; release with vrc_free.                                  */
int vrc_octree_attachments_from_grid(const int8_t *grid, uint32_t dim, const uint64_t *descriptors,
                                     uint64_t n_descriptors, uint64_t root_index, uint32_t **lookup,
                                     uint64_t **attachments, uint64_t *n_attachments);
/* ... and for the procedural scene: material 6 (mirror) where hash(x,y,z,seed) % mirror_period == 0,
 * else 5; mirror_period 0 = all 5.                                               */
int vrc_scene_shell_terrain_attachments(uint32_t depth, uint64_t seed, uint32_t mirror_period,
                                        const uint64_t *descriptors, uint64_t n_descriptors,
                                        uint64_t root_index, uint32_t **lookup, uint64_t **attachments,
                                        uint64_t *n_attachments);

/* Procedural sparse builder for grids too large to materialise (SURVEY 8d
 * "shell-terrain"): solid iff h(x,y)-thickness <= z <= h(x,y).  Emits the same
 * format/layout as vrc_octree_generate on the equivalent dense grid.
 * height: optional out, int32[dim*dim] (x + dim*y).                          */
int vrc_scene_shell_terrain(uint32_t depth, uint64_t seed, int32_t thickness,
                            int strict_reference, uint64_t **descriptors,
                            uint64_t *n_descriptors, uint64_t *root_index,
                            int32_t *height);
/* Dense twin of the above for depth <= 9 (fills char[dim^3] with material 5). */
int vrc_scene_shell_terrain_dense(uint32_t depth, uint64_t seed, int32_t thickness, int8_t *grid);
/* Map::GenerateHeightBitmap (src/map/Map.cpp:144-262): the reference's diamond-square height field
 * (std::mt19937 default seed, offsets in +-h with h = 20 halved per level, wrap-around sampling :266-272), whose
 * output the reference drops in the empty Map::ApplyHeightmap (:140-142).  height: uint8[dim*dim] =
 * clamp(value, 0, dim) like :241; corner_seed is the reference's `rand() % 10 + 55` (:160; 58 with an unseeded glibc).
 * grid (optional, int8[dim^3], x + dim*(y + dim*z)): what ApplyHeightmap was meant to do -- material 5 at and
 * below the height of each column, 0 above (our definition; the reference has none).  dim <= 4096 with a grid,
 * <= 16384 without (then feed the heights to vrc_build_heightfield: no dense grid anywhere).                    */
int vrc_scene_diamond_square(uint32_t dim, double corner_seed, uint8_t *height, int8_t *grid);
/* The same field before Map.cpp:248 quantises it: double[dim*dim], x + dim*y (what `height_map` holds at :244).  For
 * checks against a second implementation (oracle/diamond_square.py): the uint8 heights hide everything below one voxel. */
int vrc_scene_diamond_square_f64(uint32_t dim, double corner_seed, double *field);

/* The same scene with its knobs exposed: octave_floor = log2 of the finest noise cell (2 in vrc_scene_shell_terrain),
 * layout = VRC_LAYOUT_* flags.  VRC_LAYOUT_NO_PAGE_HEADERS is the "brick" layout the device builder emits: same
 * descriptor format and bottom-up order, no all-ones page-header slots (they are the only position-dependent part of
 * the reference layout, Octree.cpp:251-262, and neither kernel reads them).                                          */
#define VRC_LAYOUT_STRICT_REFERENCE 1u
#define VRC_LAYOUT_NO_PAGE_HEADERS  2u
int vrc_scene_shell_terrain_ex(uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor, uint32_t layout,
                               uint64_t **descriptors, uint64_t *n_descriptors, uint64_t *root_index, int32_t *height);
/* One column of that scene, evaluated procedurally: solid for lo <= z <= hi. */
int vrc_scene_shell_column(uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor, int64_t x, int64_t y,
                           int32_t *lo, int32_t *hi);

/* Builds the scene's SVO directly in the handle's HBM and installs it as the octree (no host copy ever exists):
 * replaces the reference builder's limits -- Octree::buffer_size = 100000 descriptors (include/map/Octree.h:29) and the
 * dense char[D^3] input of Octree::Generate (src/map/Octree.cpp:13,325-327).  The array is bit-identical to
 * vrc_scene_shell_terrain_ex(..., VRC_LAYOUT_NO_PAGE_HEADERS).  flags: VRC_BUILD_COUNT_ONLY sizes the tree without
 * allocating it.  validate_samples > 0 runs the reference's Octree::Validate (Octree.cpp:329-352) on the device over
 * that many pseudo-random voxels (tree point query vs the procedural occupancy).  probe_xy (n_probe x,y pairs) /
 * probe_lohi (n_probe lo,hi pairs): optional read-back of columns of the device height field.                      */
#define VRC_BUILD_COUNT_ONLY 1u
#define VRC_BUILD_ATTACHMENTS 2u   /* vrc_build_dense_grid: the grid's values become the tree's material attachments, on the device */
typedef struct vrc_build_info {
    uint64_t n_descriptors, root_index;
    uint64_t n_bricks, n_top_slots, n_far_pointers_top;
    double   seconds_total, seconds_height, seconds_count, seconds_emit;
    uint64_t device_bytes_peak;      /* descriptor array + builder temporaries */
    uint64_t host_bytes;             /* brick tables held on the host while building */
    uint64_t validate_samples, validate_mismatches;
} vrc_build_info;
int vrc_build_shell_terrain(vrc_caster *h, uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor,
                            uint32_t flags, uint64_t validate_samples, const int32_t *probe_xy, uint32_t n_probe,
                            int32_t *probe_lohi, vrc_build_info *info);
/* The same builder for ANY column scene: column (x,y) is solid for lo[x + dim*y] <= z <= hi[x + dim*y] (uint16[dim*dim]
 * each; lo == NULL: solid from z = 0 up).  This is Octree::Generate (src/map/Octree.cpp:13-43) for terrains -- e.g. the
 * reference's own diamond-square height field, Map::GenerateHeightBitmap (src/map/Map.cpp:144-262,
 * vrc_scene_diamond_square) -- without the dense char[D^3] grid: only the 2-D field crosses PCIe, the tree is built in
 * the handle's HBM and installed as the octree.  Bit-identical to vrc_octree_from_columns(..., VRC_LAYOUT_NO_PAGE_HEADERS).
 * validate_samples as above (tree point queries against the columns).                                               */
int vrc_build_heightfield(vrc_caster *h, uint32_t depth, const uint16_t *hi, const uint16_t *lo, uint32_t flags,
                          uint64_t validate_samples, vrc_build_info *info);
/* Host twin of vrc_build_heightfield (sequential emitter, depth <= 13): the array to compare the device build with. */
int vrc_octree_from_columns(uint32_t depth, const uint16_t *hi, const uint16_t *lo, uint32_t layout, uint64_t **descriptors,
                            uint64_t *n_descriptors, uint64_t *root_index);
/* The same builder for Octree::Generate's own input (src/map/Octree.cpp:13-43): a dense grid int8[dim^3],
 * x + dim*(y + dim*z), any non-zero voxel solid -- the reference's Map::data (src/map/Map.cpp:7-30).  The grid crosses
 * PCIe once; occupancy pyramid, count, emit and validate run in the handle's HBM and the tree is installed as the octree
 * (materials: flag VRC_BUILD_ATTACHMENTS turns the grid's values into the attachment arrays on the device -- one slot
 * per non-empty 2^3 block in grid order -- so the SVO branch renders the map's materials, mirrors included; without
 * the flag every voxel is material 5).  3 <= depth <= 12 (4096^3 = 64 GiB).
 * Bit-identical to vrc_octree_generate_ex(grid, dim, VRC_LAYOUT_NO_PAGE_HEADERS).  validate_samples: tree point queries
 * against the grid, half of them on / next to solid voxels.  grid == NULL: build from the map vrc_assign_map has already
 * put into this handle's HBM (it must be dim^3) -- a host that uses both branches, like the reference's Application,
 * uploads its Map once.                                                                                             */
int vrc_build_dense_grid(vrc_caster *h, uint32_t depth, const int8_t *grid, uint32_t flags, uint64_t validate_samples,
                         vrc_build_info *info);
/* Host twin: vrc_octree_generate with the layout flags of vrc_scene_shell_terrain_ex (array sized exactly). */
int vrc_octree_generate_ex(const int8_t *grid, uint32_t dim, uint32_t layout, uint64_t **descriptors,
                           uint64_t *n_descriptors, uint64_t *root_index);
/* Read descriptors [first, first + count) of the resident octree back to the host (tests, tools). */
int vrc_read_descriptors(vrc_caster *h, uint64_t first, uint64_t count, uint64_t *out);
int vrc_octree_size(vrc_caster *h, uint64_t *n_descriptors, uint64_t *root_index);

/* Synthetic 256x256-style atlas: texel = hash(x,y) & 0xFFFFFF, alpha 255.    */
int vrc_scene_atlas(int32_t width, int32_t height, uint8_t *rgba8);

/* Octree::GetVoxel (src/map/Octree.cpp:45-158), the CPU twin of get_oct_vox
 * (kernels/ray_caster_kernel.cl:140-251), on a host copy of the array:
 * found, resolution and sub_oct_pos as that traversal leaves them.            */
int vrc_octree_get_voxel(const uint64_t *descriptors, uint64_t root_index, uint32_t dim,
                         const int32_t position[3], int32_t *found, int32_t *resolution,
                         int32_t sub_oct_pos[3]);

/* Octree::Load (declared, never defined: include/map/Octree.h:38) and its counterpart: a flat little-endian
 * file of the descriptor array (+ optional attachment buffers).  Loaded arrays are malloc'ed (vrc_free).      */
int vrc_octree_save(const char *path, uint32_t dim, const uint64_t *descriptors, uint64_t n_descriptors,
                    uint64_t root_index, const uint32_t *lookup, const uint64_t *attachments,
                    uint64_t n_attachments);
int vrc_octree_load(const char *path, uint32_t *dim, uint64_t **descriptors, uint64_t *n_descriptors,
                    uint64_t *root_index, uint32_t **lookup, uint64_t **attachments, uint64_t *n_attachments);

/* Streaming upload of a saved tree (a scene is built once, then streamed): reads the file written by
 * vrc_octree_save in chunks through pinned staging buffers straight into device memory (host memory use is two
 * chunks, whatever the size of the tree), installs the attachment buffers if the file has them, and sets the
 * octree_root_index setting like vrc_assign_octree (CLCaster.cpp:113).  *dim receives the map dimension the file
 * was saved with (the host still owns the octree_dimensions setting).                                            */
int vrc_assign_octree_file(vrc_caster *h, const char *path, uint32_t *dim);

void vrc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
