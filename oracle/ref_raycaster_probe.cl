/* ref_raycaster_probe.cl -- TEST INFRASTRUCTURE.  Runs the REFERENCE's own `raycaster` kernel on the MI355X and
 * makes its step loop observable.
 *
 * CDNA4 has no image hardware: the reference kernel's only output, write_imagef, and its atlas read, read_imagef,
 * lower to nothing on gfx950 (profiles/r01_reference_kernel_on_gfx950.txt).  This file therefore OVERRIDES exactly
 * those two I/O builtins with macros that expand, at the kernel's own call sites (ray_caster_kernel.cl:652,684,717),
 * to stores of the kernel's own locals -- voxel, face_mask, voxel_data, distance_traveled, shadow_ray, bounce_count,
 * max_distance, the texel coordinate, the colour -- into a buffer bound to the kernel's (otherwise unread)
 * octree_attachment_buffer argument, and to a texel fetch from an RGBA8 buffer bound to its (otherwise unread)
 * octree_attachment_lookup_buffer argument.  Everything else -- ray set-up, get_oct_vox bias, the step loop, the hit
 * block, shadow redirect, mirror bounce -- is the reference source, #included unmodified from /root/reference
 * (-I on the clang command line; nothing of it is copied here).
 *
 * Status: because two builtins are replaced this is CORROBORATION of the oracle's step loop and hit block, not a
 * pin of it.  tests/test_reference_pin_gpu.py compares the records with oracle/vrc_oracle.c.   */

/* record: 32 ints per pixel */
#define REC_WORDS 32
#define REC_COLOR 0      /* 4 floats: the colour handed to write_imagef (:716-721)                  */
#define REC_VOXEL 4      /* 3 ints: voxel at the end of the kernel, [7] voxel_data                    */
#define REC_FACE 8       /* 3 ints: face_mask, [11] distance_traveled                                 */
#define REC_SHADOW 12    /* shadow_ray, [13] bounce_count, [14] max_distance, [15] 1 = write_imagef ran */
#define REC_READS 16     /* number of read_imagef calls (texture hits)                                */
#define REC_HIT 17       /* first read_imagef: 3 ints voxel, [20..22] face_mask, [23] voxel_data,
                            [24] distance_traveled, [25..26] texel coordinate as passed               */

void probe_write(global ulong *rec_base, constant int2 *res, int2 px, float4 col, int3 voxel, int3 face_mask,
                 int voxel_data, int distance_traveled, int shadow_ray, int bounce_count, int max_distance) {
    global int *r = (global int *)rec_base + REC_WORDS * (px.x + (*res).x * px.y);
    global float *f = (global float *)r;
    f[REC_COLOR] = col.x; f[REC_COLOR + 1] = col.y; f[REC_COLOR + 2] = col.z; f[REC_COLOR + 3] = col.w;
    r[REC_VOXEL] = voxel.x; r[REC_VOXEL + 1] = voxel.y; r[REC_VOXEL + 2] = voxel.z; r[REC_VOXEL + 3] = voxel_data;
    r[REC_FACE] = face_mask.x; r[REC_FACE + 1] = face_mask.y; r[REC_FACE + 2] = face_mask.z; r[REC_FACE + 3] = distance_traveled;
    r[REC_SHADOW] = shadow_ray; r[REC_SHADOW + 1] = bounce_count; r[REC_SHADOW + 2] = max_distance; r[REC_SHADOW + 3] = 1;
}

float4 probe_read(global uint *atlas, global ulong *rec_base, constant int2 *res, constant int2 *atlas_dim, int2 px,
                  int2 c, int3 voxel, int3 face_mask, int voxel_data, int distance_traveled) {
    global int *r = (global int *)rec_base + REC_WORDS * (px.x + (*res).x * px.y);
    if (r[REC_READS] == 0) {
        r[REC_HIT] = voxel.x; r[REC_HIT + 1] = voxel.y; r[REC_HIT + 2] = voxel.z;
        r[REC_HIT + 3] = face_mask.x; r[REC_HIT + 4] = face_mask.y; r[REC_HIT + 5] = face_mask.z;
        r[REC_HIT + 6] = voxel_data; r[REC_HIT + 7] = distance_traveled;
        r[REC_HIT + 8] = c.x; r[REC_HIT + 9] = c.y;
    }
    r[REC_READS] += 1;
    /* sampler-less read_imagef of a CL_UNORM_INT8 RGBA image; out-of-range coordinates are undefined in OpenCL, the
     * oracle clamps them (SURVEY 8c caveat 2) and so does this */
    int w = (*atlas_dim).x, h = (*atlas_dim).y;
    int tx = c.x < 0 ? 0 : (c.x >= w ? w - 1 : c.x), ty = c.y < 0 ? 0 : (c.y >= h ? h - 1 : c.y);
    uint t = atlas[tx + w * ty];
    return (float4)((float)(t & 255u) / 255.0f, (float)((t >> 8) & 255u) / 255.0f, (float)((t >> 16) & 255u) / 255.0f,
                    (float)(t >> 24) / 255.0f);
}

#define write_imagef(img, px, col)                                                                                  \
    probe_write(octree_attachment_buffer, resolution, (px), (col), voxel, face_mask, voxel_data, distance_traveled, \
                (int)shadow_ray, (int)bounce_count, max_distance)
#define read_imagef(img, c)                                                                                          \
    probe_read(octree_attachment_lookup_buffer, octree_attachment_buffer, resolution, atlas_dim, pixel, (c), voxel, \
               face_mask, voxel_data, distance_traveled)

#include "ray_caster_kernel.cl"

/* sin / cos of the camera angles exactly as the kernel's own expressions evaluate them (:280-291) under this
 * build's math flags: handed to the oracle as its cam_trig so a different libm cannot perturb the comparison */
kernel void probe_trig(global float2 *cam_dir, global float *out) {
    out[0] = sin((*cam_dir).x); out[1] = cos((*cam_dir).x);
    out[2] = sin((*cam_dir).y); out[3] = cos((*cam_dir).y);
}
