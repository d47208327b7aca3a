/* ref_probe.cl -- TEST INFRASTRUCTURE.  Probe kernels that call the REFERENCE's own device functions.
 *
 * The first line pulls in kernels/ray_caster_kernel.cl of the reference, unmodified, from where it lies under
 * /root/reference (-I on the clang command line, oracle/Makefile target _ref); nothing of it is copied here.  The
 * `raycaster` kernel itself cannot be observed on an MI355X (its only inputs/outputs that matter are image2d_t, and
 * CDNA4 has no image hardware: profiles/r01_reference_kernel_on_gfx950.txt), but the two pure functions it is built
 * from can: view_light (:78-99, the whole shading arithmetic of the hit block) and get_oct_vox (:140-251, the octree
 * point query).  These kernels hand them buffers and store what they return, nothing else.  The code object is built
 * in the development container and travels to the GPU box under oracle/_ref/; tests/test_reference_pin_gpu.py
 * runs it there and pins oracle/vrc_oracle.c against it.                                                          */
#include "ray_caster_kernel.cl"

/* in: 14 floats per case = in_color[4] light[3] light_color[4] view[3]; mask: 3 ints per case; out: 4 floats */
kernel void probe_view_light(global const float *in, global const int *mask, global float *out, int n) {
    int i = get_global_id(0);
    if (i >= n) return;
    global const float *p = in + 14 * i;
    float4 r = view_light((float4)(p[0], p[1], p[2], p[3]), (float3)(p[4], p[5], p[6]), (float4)(p[7], p[8], p[9], p[10]),
                          (float3)(p[11], p[12], p[13]), (int3)(mask[3 * i], mask[3 * i + 1], mask[3 * i + 2]));
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

/* positions: 3 ints per case; out: 40 ints per case (layout: PROBE_* below, mirrored in tests) */
kernel void probe_get_oct_vox(global const int *positions, global ulong *descriptors, global uint *lookup,
                              global ulong *attachments, global ulong *settings, global int *out, int n) {
    int i = get_global_id(0);
    if (i >= n) return;
    struct TraversalState ts = get_oct_vox((int3)(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]),
                                           descriptors, lookup, attachments, settings);
    global int *o = out + 40 * i;
    o[0] = ts.found; o[1] = ts.scale; o[2] = ts.resolution; o[3] = ts.parent_stack_position;
    o[4] = ts.sub_oct_pos.x; o[5] = ts.sub_oct_pos.y; o[6] = ts.sub_oct_pos.z;
    o[7] = ts.oct_pos.x; o[8] = ts.oct_pos.y; o[9] = ts.oct_pos.z;
    o[10] = (int)(ts.current_descriptor_index & 0xffffffffUL); o[11] = (int)(ts.current_descriptor_index >> 32);
    o[12] = (int)(ts.current_descriptor & 0xffffffffUL); o[13] = (int)(ts.current_descriptor >> 32);
    for (int k = 0; k < 8; k++) {
        /* entries beyond scale / parent_stack_position are uninitialised in the reference: the test ignores them */
        o[14 + k] = ts.idx_stack[k];
        o[22 + k] = (int)(ts.parent_stack_index[k] & 0xffffffffUL);
        o[30 + k] = (int)(ts.parent_stack[k] & 0xffffffffUL);
    }
    o[38] = 0; o[39] = 0;
}
