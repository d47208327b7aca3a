// raycast_common.hpp -- device code shared by the two raycast kernels: the ray
// set-up, the hit block and the epilogue of kernels/ray_caster_kernel.cl, written
// once so the dense-array kernel and the SVO kernel cannot drift apart.
//
// Numerics: built with -ffp-contract=off, IEEE divide/sqrt; every float
// expression is evaluated in the order the reference source states it, so the
// results are bit-identical to the CPU oracle (oracle/vrc_oracle.c).
#pragma once

#include <hip/hip_runtime.h>

#include "vrc_params.h"

namespace vrc {

__device__ __forceinline__ float min_cl(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float max_cl(float a, float b) { return a < b ? b : a; }
__device__ __forceinline__ float mix_cl(float x, float y, float a) { return x + (y - x) * a; }
__device__ __forceinline__ int isign(float v) { return (v > 0.0f) - (v < 0.0f); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return ax * bx + ay * by + az * bz;
}

struct Vec3 { float x, y, z; };
constexpr unsigned kCountTex = 1u, kCountShadow = 0x100u, kCountBounce = 0x10000u;

// OpenCL normalize(): v / |v|, v itself when it is all zero
__device__ __forceinline__ Vec3 normalize3(Vec3 v) {
    float l2 = dot3(v.x, v.y, v.z, v.x, v.y, v.z);
    if (l2 == 0.0f) return v;
    float l = sqrtf(l2);
    return Vec3{v.x / l, v.y / l, v.z / l};
}

// view_light (ray_caster_kernel.cl:78-99)
__device__ __forceinline__ void view_light(float out[4], const float in_color[4], Vec3 light,
                                           const float light_color[4], Vec3 view, int mx, int my, int mz) {
    if (light.x == 0.0f && light.y == 0.0f && light.z == 0.0f) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return;
    }
    float d = sqrtf(dot3(light.x, light.y, light.z, light.x, light.y, light.z)) * 0.01f;
    d *= d;
    Vec3 nmask = normalize3(Vec3{(float)mx, (float)my, (float)mz});
    Vec3 nlight = normalize3(light);
    float diffuse = max_cl(dot3(nmask.x, nmask.y, nmask.z, nlight.x, nlight.y, nlight.z), 0.1f);
    float specular = 0.0f;
    if (diffuse > 0.0f) {
        Vec3 nview = normalize3(view);
        Vec3 halfway = normalize3(Vec3{nlight.x + nview.x, nlight.y + nview.y, nlight.z + nview.z});
        specular = max_cl(dot3(nmask.x, nmask.y, nmask.z, halfway.x, halfway.y, halfway.z), 0.0f);
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
        out[c] = in_color[c] + (diffuse * light_color[c] + specular * light_color[c] / d);
}

// Per-ray state of the reference kernel's locals (ray_caster_kernel.cl:276-337).
struct Ray {
    float rdx, rdy, rdz;                 // ray_dir
    int sx, sy, sz;                      // voxel_step
    int vx, vy, vz;                      // voxel
    float dtx, dty, dtz;                 // delta_t
    float itx, ity, itz;                 // intersection_t
    int fmx, fmy, fmz;                   // face_mask
    int distance_traveled, max_distance;
    unsigned counts;                     // bits 0-7 atlas texels fetched, 8-15 shadow rays cast, 16-17 bounce_count
    float voxel_color[3], color_accumulator[4];   // voxel_color.w is 0 from start to end in the reference (:684 "-= 0.0f")
    float fog_distance;
    bool shadow_ray, written;            // (as bools they live in scalar masks; packed into `flags` they cost VGPRs)
    long pix0;                           // index of the wave's first pixel in the viewport / image / hit buffers (wave-uniform: cold_pixel_index)
    int flags;                           // kFlag* bits 0-3, kFlagHasHit, face_mask of the primary hit in bits 8-10
    // the first strike (primary hit): what the light block needs to run again for a further light
    int kvx, kvy, kvz;                   // the solid voxel
    float kfx, kfy, kfz;                 // face_position
    int kmx, kmy, kmz;                   // face_mask * voxel_step of the primary ray
    int kdist;                           // distance_traveled at the strike
    int light_index;
    float hpx, hpy, hpz;                 // hit_pos of the last redirect (read by the node-exit jump kernel only)
};

__device__ __forceinline__ int cold_uniform(int v);                            // (below, with block_pixel)
__device__ __forceinline__ long cold_pixel_index(const RaycastParams &p, long pix0);

// :276-323 + the frame-constant bias of :342-354.  Returns false for the
// zero-component early return (:293-294): nothing is written for that pixel.
__device__ __forceinline__ bool ray_setup(Ray &r, const RaycastParams &p, long pix) {
    r.flags = 0;
    r.written = false;
    r.distance_traveled = 0;
    r.counts = 0;
    r.light_index = 0;

    const float4 pm = reinterpret_cast<const float4 *>(p.viewport)[pix];
    {
        const float s1 = p.trig[0], c1 = p.trig[1], s2 = p.trig[2], c2 = p.trig[3];
        float x = pm.z * s1 + pm.x * c1;                                      // pitch :280-284
        float y = pm.y;
        float z = pm.z * c1 - pm.x * s1;
        r.rdx = x * c2 - y * s2;                                              // yaw :287-291
        r.rdy = x * s2 + y * c2;
        r.rdz = z;
    }
    if (r.rdx == 0.0f || r.rdy == 0.0f || r.rdz == 0.0f) return false;        // :293-294

    r.sx = isign(r.rdx); r.sy = isign(r.rdy); r.sz = isign(r.rdz);            // :298
    const float flx = floorf(p.cam_pos[0]), fly = floorf(p.cam_pos[1]), flz = floorf(p.cam_pos[2]);
    r.vx = (int)flx; r.vy = (int)fly; r.vz = (int)flz;                        // :302
    r.dtx = fabsf(1.0f / r.rdx); r.dty = fabsf(1.0f / r.rdy); r.dtz = fabsf(1.0f / r.rdz);   // :307
    {
        float ox = r.dtx * (p.cam_pos[0] - flx), oy = r.dty * (p.cam_pos[1] - fly), oz = r.dtz * (p.cam_pos[2] - flz);
        r.itx = ox * -(float)r.sx; r.ity = oy * -(float)r.sy; r.itz = oz * -(float)r.sz;        // :317
        r.itx += r.dtx * -1.0f * (r.itx < 0.0f ? -1.0f : 0.0f);                                  // :323
        r.ity += r.dty * -1.0f * (r.ity < 0.0f ? -1.0f : 0.0f);
        r.itz += r.dtz * -1.0f * (r.itz < 0.0f ? -1.0f : 0.0f);
    }
    r.itx += (float)p.frame[0]; r.ity += (float)p.frame[1]; r.itz += (float)p.frame[2];         // :353-354

    r.max_distance = p.max_distance;                                          // :326
    r.fmx = r.fmy = r.fmz = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) r.color_accumulator[c] = 0.0f;
    r.voxel_color[0] = r.voxel_color[1] = r.voxel_color[2] = 0.0f;
    r.fog_distance = 0.0f;
    r.shadow_ray = false;
    r.written = true;
    return true;
}

// :563-568, entered with the voxel already stepped out of the map
__device__ __forceinline__ void oob_exit(Ray &r) {
    r.vx -= r.sx * r.fmx; r.vy -= r.sy * r.fmy; r.vz -= r.sz * r.fmz;
    const float k = 1.0f - max_cl((float)r.distance_traveled / 700.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 3; c++) r.color_accumulator[c] = mix_cl(0.0f, r.voxel_color[c], k);
    r.color_accumulator[3] = mix_cl(0.0f, 0.0f, k);
    r.color_accumulator[3] *= 4.0f;
    r.flags |= kFlagOob;
}

// What oob_exit() / shadow_hit() do to the colour state, owed by a segment that ended in the SVO kernel's event phase (which only
// sets kFlagOobPending / kFlagShadowPending): paid where the colours are touched anyway -- before the next light takes them as its
// input (multi-light) and before the pixel is written.  distance_traveled is what it was when the ray left the map (a finished
// segment no longer steps).  (The voxel correction of oob_exit has no reader there: the next segment starts from the strike.)
__device__ __forceinline__ void settle_segment(Ray &r) {
    if (r.flags & kFlagOobPending) {
        const float k = 1.0f - max_cl((float)r.distance_traveled / 700.0f, 0.0f);
#pragma unroll
        for (int c = 0; c < 3; c++) r.color_accumulator[c] = mix_cl(0.0f, r.voxel_color[c], k);
        r.color_accumulator[3] = mix_cl(0.0f, 0.0f, k);
        r.color_accumulator[3] *= 4.0f;
    }
    if (r.flags & kFlagShadowPending) r.color_accumulator[3] = 0.1f;
    r.flags &= ~(kFlagOobPending | kFlagShadowPending);
}

// :677-679 / :700-702: delta_t and intersection_t of a ray restarted at hit_pos (ray_dir, voxel_step already set)
__device__ __forceinline__ void restart_from(Ray &r, Vec3 hit_pos) {
    r.dtx = fabsf(1.0f / r.rdx); r.dty = fabsf(1.0f / r.rdy); r.dtz = fabsf(1.0f / r.rdz);
    r.itx = r.dtx * (hit_pos.x - floorf(hit_pos.x)) * (float)r.sx;
    r.ity = r.dty * (hit_pos.y - floorf(hit_pos.y)) * (float)r.sy;
    r.itz = r.dtz * (hit_pos.z - floorf(hit_pos.z)) * (float)r.sz;
    r.itx += r.dtx * -(r.itx < 0.0f ? -1.0f : 0.0f);
    r.ity += r.dty * -(r.ity < 0.0f ? -1.0f : 0.0f);
    r.itz += r.dtz * -(r.itz < 0.0f ? -1.0f : 0.0f);
}

// The light part of the hit block (:657-679) for light l, run from the FIRST STRIKE stored in r.k*:
// view_light, max_distance, redirect toward the light.  For l = 0 this is the reference's code at its place; the
// multi-light extension (TODO src/main.cpp:33 "first-strike resetting") runs it again for each
// further light, chaining the colour.  Returns false for the early return of :671-672 (pixel left unwritten).
__device__ __forceinline__ bool light_from_strike(Ray &r, const RaycastParams &p, int l, bool cast) {
    const Vec3 light_pos{p.lights[l][4], p.lights[l][5], p.lights[l][6]};
    const Vec3 hit_pos{(float)r.kvx + r.kfx, (float)r.kvy + r.kfy, (float)r.kvz + r.kfz};
    float in_color[4];
    if (l == 0) { in_color[0] = r.voxel_color[0]; in_color[1] = r.voxel_color[1]; in_color[2] = r.voxel_color[2]; in_color[3] = 0.0f; }
    else { in_color[0] = r.color_accumulator[0]; in_color[1] = r.color_accumulator[1]; in_color[2] = r.color_accumulator[2]; in_color[3] = r.color_accumulator[3]; }
    view_light(r.color_accumulator, in_color,
               Vec3{hit_pos.x - light_pos.x, hit_pos.y - light_pos.y, hit_pos.z - light_pos.z}, p.lights[l],
               Vec3{hit_pos.x - p.cam_pos[0], hit_pos.y - p.cam_pos[1], hit_pos.z - p.cam_pos[2]}, r.kmx, r.kmy, r.kmz);
    if (!cast) return true;
    {
        const float ddx = (float)r.kvx - light_pos.x, ddy = (float)r.kvy - light_pos.y, ddz = (float)r.kvz - light_pos.z;
        r.max_distance = (int)((float)r.kdist + sqrtf(dot3(ddx, ddy, ddz, ddx, ddy, ddz)));       // :667
    }
    const Vec3 nd = normalize3(Vec3{light_pos.x - hit_pos.x, light_pos.y - hit_pos.y, light_pos.z - hit_pos.z});
    r.rdx = nd.x; r.rdy = nd.y; r.rdz = nd.z;                                 // :670
    if (r.rdx == 0.0f || r.rdy == 0.0f || r.rdz == 0.0f) { r.written = false; return false; }   // :671-672
    r.counts += kCountShadow;
    r.flags |= kFlagShadowCast;
    r.vx = r.kvx - r.kmx; r.vy = r.kvy - r.kmy; r.vz = r.kvz - r.kmz;         // :674 voxel -= voxel_step * face_mask
    r.sx = isign(r.rdx); r.sy = isign(r.rdy); r.sz = isign(r.rdz);            // :675
    return true;                                                              // the caller restarts: :677-679
}
__device__ __forceinline__ Vec3 strike_pos(const Ray &r) {
    return Vec3{(float)r.kvx + r.kfx, (float)r.kvy + r.kfy, (float)r.kvz + r.kfz};
}

// multi-light extension: the shadow ray toward light r.light_index has ended (step cap :357, left the map
// :563-568, or blocked :707-710) and another light is waiting
__device__ __forceinline__ bool more_lights(const Ray &r, const RaycastParams &p) {
    return r.shadow_ray && r.written && p.shadow_rays && r.light_index + 1 < p.light_count;
}

// :707-710: all the hit block does for a shadow ray that struck a voxel (the face arithmetic before it has no reader);
// the kernels do it where the ray lands instead of parking the lane for a hit-block pass, then break (:710)
__device__ __forceinline__ void shadow_hit(Ray &r) {
    r.color_accumulator[3] = 0.1f;
    r.flags |= kFlagShadowHit;
}

// :575-711 for voxel_data in {5, 6}.  Returns true when the loop breaks / the
// kernel returns, false when the (redirected) ray keeps stepping.
template <bool kMulti>
__device__ __forceinline__ bool hit_block(Ray &r, int voxel_data, const RaycastParams &p) {
    float fpx = 0.f, fpy = 0.f, fpz = 0.f, tfx = 0.f, tfy = 0.f;
    float sgx = 1.0f, sgy = 1.0f, sgz = 1.0f;
    if (r.fmx == 1) {                                                         // :586-599
        sgx = (float)((double)sgx * -1.0);
        float z_percent = (r.itz - (r.itx - r.dtx)) / r.dtz;
        float y_percent = (r.ity - (r.itx - r.dtx)) / r.dty;
        fpx = 1.00001f; fpy = y_percent; fpz = z_percent;
        tfx = fpy; tfy = fpz;
    } else if (r.fmy == 1) {                                                  // :601-608
        sgy = (float)((double)sgy * -1.0);
        float x_percent = (r.itx - (r.ity - r.dty)) / r.dtx;
        float z_percent = (r.itz - (r.ity - r.dty)) / r.dtz;
        fpx = x_percent; fpy = 1.00001f; fpz = z_percent;
        tfx = fpx; tfy = fpz;
    } else if (r.fmz == 1) {                                                  // :610-618
        sgz = (float)((double)sgz * -1.0);
        float x_percent = (r.itx - (r.itz - r.dtz)) / r.dtx;
        float y_percent = (r.ity - (r.itz - r.dtz)) / r.dty;
        fpx = x_percent; fpy = y_percent; fpz = 1.00001f;
        tfx = fpx; tfy = fpy;
    }
    // :626-643 quadrant flips
    if (r.rdx > 0.0f) fpx = -fpx + 1.0f;
    if (r.rdx < 0.0f) tfx = -tfx + 1.0f;
    if (r.rdy > 0.0f) {
        fpy = -fpy + 1.0f;
    } else {
        tfx = (float)(1.0 - (double)tfx);
        if (r.fmz == 1) { tfx = 1.0f - tfx; tfy = 1.0f - tfy; }
    }
    if (r.rdz > 0.0f) fpz = -fpz + 1.0f;
    if (r.rdz < 0.0f) tfy = -tfy + 1.0f;

    // the primary hit goes straight into the pixel's hit record (it would otherwise sit in five registers until the
    // ray has finished); its face mask rides in the flags word
    if (!(r.flags & kFlagHasHit) && !r.shadow_ray) {
        if (p.hits) reinterpret_cast<int4 *>(p.hits)[2 * cold_pixel_index(p, r.pix0)] = make_int4(r.vx, r.vy, r.vz, voxel_data);
        r.flags |= kFlagHasHit | ((r.fmx | (r.fmy << 1) | (r.fmz << 2)) << 8);
    }

    if (r.shadow_ray) {                                                       // :707-710
        r.color_accumulator[3] = 0.1f;
        r.flags |= kFlagShadowHit;
        return true;
    }

    const bool mirror = (voxel_data == 6);
    // :652-656 / :684-688  tile (5,0) halved, tile (3,4) quartered
    const float tiles_x = (float)cold_uniform(p.tiles_x), tiles_y = (float)cold_uniform(p.tiles_y);
    int tx = (int)(tfx * tiles_x) + (int)((mirror ? 3.0f : 5.0f) * tiles_x);
    int ty = (int)(tfy * tiles_y) + (int)((mirror ? 4.0f : 0.0f) * tiles_y);
    tx = tx < 0 ? 0 : (tx >= p.atlas_w ? p.atlas_w - 1 : tx);                 // undefined in OpenCL: clamp
    ty = ty < 0 ? 0 : (ty >= p.atlas_h ? p.atlas_h - 1 : ty);
    const uchar4 t8 = reinterpret_cast<const uchar4 *>(p.atlas)[(long)tx + (long)p.atlas_w * ty];
    r.counts += kCountTex;
    const float div = mirror ? 4.0f : 2.0f;
    r.voxel_color[0] += ((float)t8.x / 255.0f) / div;
    r.voxel_color[1] += ((float)t8.y / 255.0f) / div;
    r.voxel_color[2] += ((float)t8.z / 255.0f) / div;

    const Vec3 hit_pos{(float)r.vx + fpx, (float)r.vy + fpy, (float)r.vz + fpz};
    if (!mirror) {                                                            // :649-679
        r.shadow_ray = true;
        r.kvx = r.vx; r.kvy = r.vy; r.kvz = r.vz;
        r.kfx = fpx; r.kfy = fpy; r.kfz = fpz;
        r.kmx = r.fmx * r.sx; r.kmy = r.fmy * r.sy; r.kmz = r.fmz * r.sz;
        r.kdist = r.distance_traveled;
        r.fog_distance = (float)r.distance_traveled;                          // :666
        const bool cast = p.shadow_rays != 0;
        if (!light_from_strike(r, p, 0, cast)) return true;
        if (!cast) {                                                          // extension: primary rays only;
            if (kMulti)                                                       // every active light shades
                for (int l = 1; l < p.light_count; l++) light_from_strike(r, p, l, false);
            return true;
        }
    } else {                                                                  // :682-704
        r.rdx *= sgx; r.rdy *= sgy; r.rdz *= sgz;                             // :693
        if (r.rdx == 0.0f || r.rdy == 0.0f || r.rdz == 0.0f) { r.written = false; return true; }
        r.vx -= r.sx * r.fmx; r.vy -= r.sy * r.fmy; r.vz -= r.sz * r.fmz;     // :697
        // :698 precedence quirk: +1 for both signs
        r.sx = (-1 * (r.rdx > 0.0f ? -1 : 0)) - (r.rdx < 0.0f ? -1 : 0);
        r.sy = (-1 * (r.rdy > 0.0f ? -1 : 0)) - (r.rdy < 0.0f ? -1 : 0);
        r.sz = (-1 * (r.rdz > 0.0f ? -1 : 0)) - (r.rdz < 0.0f ? -1 : 0);
        r.counts += kCountBounce;
    }
    restart_from(r, hit_pos);                                                 // :677-679 / :700-702
    r.hpx = hit_pos.x; r.hpy = hit_pos.y; r.hpz = hit_pos.z;
    return false;
}

// :716-721 + the hit record
__device__ __forceinline__ void ray_finish(Ray &r, const RaycastParams &p, unsigned c_desc) {
    const long pix = cold_pixel_index(p, r.pix0);
    if (r.written) {
        const float k = 1.0f - max_cl(r.fog_distance / 700.0f, 0.0f);         // :716
#ifdef VRC_NO_FRAME_STORE   // (traffic accounting only, tools/gpu_pmc_writes.sh: the frame is computed and not stored)
        if (k == -123.0f)
#endif
        reinterpret_cast<float4 *>(p.image)[pix] =
            make_float4(mix_cl(0.0f, r.color_accumulator[0], k), mix_cl(0.0f, r.color_accumulator[1], k),
                        mix_cl(0.0f, r.color_accumulator[2], k), mix_cl(0.0f, r.color_accumulator[3], k));
        r.flags |= kFlagWritten;
    }
    if (!p.hits) return;
    int4 *hp = reinterpret_cast<int4 *>(p.hits) + 2 * pix;
    if (!(r.flags & kFlagHasHit)) hp[0] = make_int4(-1, -1, -1, 0);
    hp[1] = make_int4((r.flags >> 8) & 7, (r.flags & 0xf) | ((int)((r.counts >> 16) & 3) << 4), r.distance_traveled, (int)c_desc);
}

// block id -> pixel.  Block b runs on XCD b % 8: xcd_mode 1 (default) deals the tile rows k, k+8, ... to XCD k when
// the row count is a multiple of 8 and otherwise keeps the row-major order (XCD k gets every 8th block of each row);
// either way every XCD renders the same sky/ground mix.  xcd_mode 0 gives XCD k the k-th contiguous eighth of the
// image (L2 locality, but 37 % slower: sky rows take longer than ground rows); xcd_mode 2 never remaps.
__device__ __forceinline__ void block_pixel(const RaycastParams &p, int &px, int &py, int &buffer_row) {
    const int nblocks = gridDim.x;
    int bid = blockIdx.x;
    const int per_xcd = nblocks >> 3;
    int local_ty, bx;
    if (p.xcd_mode == 1 && per_xcd > 0 && bid < (per_xcd << 3) && (p.local_tile_rows & 7) == 0) {
        // XCD k renders the tile rows k, k+8, k+16, ...: equal sky/ground mix per XCD
        const int j = bid >> 3;
        local_ty = (j / p.blocks_x) * 8 + (bid & 7);
        bx = j % p.blocks_x;
    } else {
        // XCD k renders the k-th contiguous eighth of the image (mode 0); mode 2: no remap
        if (p.xcd_mode == 0 && per_xcd > 0 && bid < (per_xcd << 3)) bid = (bid & 7) * per_xcd + (bid >> 3);
        local_ty = bid / p.blocks_x;                      // tile row among this rank's rows
        bx = bid - local_ty * p.blocks_x;
    }
    const int band = local_ty / p.band_tiles;
    const int tile_y = (band * p.tile_world + p.tile_rank) * p.band_tiles + (local_ty - band * p.band_tiles);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    px = (bx * kTilesPerBlock + wave) * kTileW + (lane & (kTileW - 1));
    py = tile_y * kTileH + lane / kTileW;
    // row of this pixel in the viewport / image / hit buffers: the image row, or its position among this rank's rows
    buffer_row = p.row_sliced ? local_ty * kTileH + lane / kTileW : py;
}

// Values the COLD code needs (hit block, epilogue) and the step loop does not: computed where they are used, behind an empty
// asm the optimiser cannot move code across -- hoisted out of the round loop they sit in registers the loop needs, i.e. in
// scratch (the pixel index and the hit-record offset took 16 B per lane, two wave-uniform floats another 8; a scratch byte is
// written to HBM once per wave whether it is read again or not).  The pixel index comes back from the index of the wave's
// first pixel (wave-uniform: scalar registers) and the lane number: a wave is one kTileW x kTileH tile.
__device__ __forceinline__ int cold_uniform(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
// threadIdx.x again after the round loop, from the wave's index in the block (scalar: readfirstlane(threadIdx.x >> 6) taken at the
// start) and the lane number
__device__ __forceinline__ int cold_thread_index(int wave_in_block) {
    unsigned zero = 0;
    asm volatile("" : "+v"(zero));
    return wave_in_block * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
}
__device__ __forceinline__ long wave_first_pixel(long pix) {                  // call with the whole wave active
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)pix), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(pix >> 32));
    return (long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long cold_pixel_index(const RaycastParams &p, long pix0) {
    int lane = (int)threadIdx.x;
    asm volatile("" : "+v"(lane));
    lane &= 63;
    return pix0 + (long)((lane & (kTileW - 1)) + p.width * (lane / kTileW));
}

// per-block counter partials (no global atomics): wave shuffle reduce, LDS, one row per block
__device__ __forceinline__ void publish_counters(const RaycastParams &p, unsigned long long *block_ctr,
                                                 const unsigned (&vals)[7], int thread) {
    const int lane = thread & 63;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        unsigned long long v = vals[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v) atomicAdd(&block_ctr[k], v);
    }
    __syncthreads();
    if (thread < kCtrCount) p.counters[(long)blockIdx.x * kCtrCount + thread] = block_ctr[thread];
}

}  // namespace vrc
