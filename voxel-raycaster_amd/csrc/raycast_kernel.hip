// raycast_kernel.hip -- the per-pixel hot path for gfx950 (MI355X, wave64).
//
// Replaces kernels/ray_caster_kernel.cl of the reference: pitch/yaw of the
// table ray (:276-294), Amanatides-Woo set-up (:298-354), the step loop
// (:357,555-570,714), the hit block (UV :575-643, atlas + view_light + shadow
// redirect :649-679, mirror :682-704, shadow hit :707-710) and the fog/write
// epilogue (:716-721).  Occupancy comes either from the dense char map (array
// branch, :569) or from the 64-bit child-descriptor array (Octree.h:89-94)
// walked with a per-ray stack held in LDS.
//
// Numerics: built with -ffp-contract=off, IEEE divide/sqrt; every float
// expression is evaluated in the order the reference source states it, so hits
// are bit-identical to the CPU oracle (oracle/vrc_oracle.c).
//
// Mapping: one wavefront = one 8x8 pixel tile (ray coherence: neighbouring
// rays walk the same octree nodes and hit the same L1/L2 lines); a 256-thread
// block = 4 adjacent tiles; block ids are remapped so that each XCD (block id
// mod 8) renders a contiguous part of the image and keeps its own L2 warm.
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

// Packed stack entry of one descriptor level:
//   bits 0-7 valid mask, 8-15 leaf mask, 16-63 absolute index of the first kept child
__device__ __forceinline__ uint64_t make_entry(const uint64_t *__restrict__ descriptors, uint64_t index,
                                               uint64_t d) {
    uint64_t base = index + (d & 0x7fffULL);
    if (d & 0x8000ULL) base = descriptors[base];          // far pointer: slot holds an absolute index
    return (base << 16) | ((d >> 16) & 0xffffULL);        // (leaf<<8 | valid) are bits 16..31 of d
}

template <bool SVO>
__global__ __launch_bounds__(kBlockThreads) void raycast_kernel(const RaycastParams p) {
    extern __shared__ uint64_t lds_stack[];               // [level-1][thread], levels 1..n-1
    __shared__ unsigned long long block_ctr[kCtrCount];

    const int tid = threadIdx.x;
    if (tid < kCtrCount) block_ctr[tid] = 0;
    __syncthreads();

    // ---- block id -> image tile (XCD-aware: block b runs on XCD b % 8) ----
    const int nblocks = gridDim.x;
    int bid = blockIdx.x;
    {
        const int per_xcd = nblocks >> 3;
        if (per_xcd > 0 && bid < (per_xcd << 3)) bid = (bid & 7) * per_xcd + (bid >> 3);
    }
    const int local_ty = bid / p.blocks_x;                // tile row among this rank's rows
    const int bx = bid - local_ty * p.blocks_x;
    const int band = local_ty / p.band_tiles;
    const int tile_y = (band * p.tile_world + p.tile_rank) * p.band_tiles + (local_ty - band * p.band_tiles);

    const int wave = tid >> 6, lane = tid & 63;
    const int px = (bx * kTilesPerBlock + wave) * kTileW + (lane & 7);
    const int py = tile_y * kTileH + (lane >> 3);

    unsigned c_primary = 0, c_shadow = 0, c_desc = 0, c_tex = 0, c_map = 0, c_steps = 0, c_unwritten = 0;

    if (px < p.width && py < p.height) {
        const long pix = (long)px + (long)p.width * py;

        int hit_vx = -1, hit_vy = -1, hit_vz = -1, hit_mat = 0, hit_face = 0;
        int flags = 0;
        int distance_traveled = 0;
        unsigned bounce_count = 0;
        bool written = false;
        float out_color[4] = {0.f, 0.f, 0.f, 0.f};

        // :276-291
        const float4 pm = reinterpret_cast<const float4 *>(p.viewport)[pix];
        float rdx, rdy, rdz;
        {
            const float s1 = p.trig[0], c1 = p.trig[1], s2 = p.trig[2], c2 = p.trig[3];
            float x = pm.z * s1 + pm.x * c1;
            float y = pm.y;
            float z = pm.z * c1 - pm.x * s1;
            rdx = x * c2 - y * s2;
            rdy = x * s2 + y * c2;
            rdz = z;
        }

        if (rdx == 0.0f || rdy == 0.0f || rdz == 0.0f) {         // :293-294 no write
            c_unwritten = 1;
        } else {
            c_primary = 1;
            int sx = isign(rdx), sy = isign(rdy), sz = isign(rdz);            // :298
            const float flx = floorf(p.cam_pos[0]), fly = floorf(p.cam_pos[1]), flz = floorf(p.cam_pos[2]);
            int vx = (int)flx, vy = (int)fly, vz = (int)flz;                  // :302
            float dtx = fabsf(1.0f / rdx), dty = fabsf(1.0f / rdy), dtz = fabsf(1.0f / rdz);   // :307
            float itx, ity, itz;
            {
                float ox = dtx * (p.cam_pos[0] - flx), oy = dty * (p.cam_pos[1] - fly), oz = dtz * (p.cam_pos[2] - flz);
                itx = ox * -(float)sx; ity = oy * -(float)sy; itz = oz * -(float)sz;           // :317
                itx += dtx * -1.0f * (itx < 0.0f ? -1.0f : 0.0f);                               // :323
                ity += dty * -1.0f * (ity < 0.0f ? -1.0f : 0.0f);
                itz += dtz * -1.0f * (itz < 0.0f ? -1.0f : 0.0f);
            }
            itx += (float)p.frame[0]; ity += (float)p.frame[1]; itz += (float)p.frame[2];      // :353-354

            int max_distance = p.max_distance;                                // :326
            int fmx = 0, fmy = 0, fmz = 0;
            float voxel_color[4] = {0.f, 0.f, 0.f, 0.f};
            float color_accumulator[4] = {0.f, 0.f, 0.f, 0.f};
            float fog_distance = 0.0f;
            bool shadow_ray = false;
            written = true;

            // ---- SVO cursor -------------------------------------------------
            const int n = p.log2_dim;
            uint64_t root_entry = 0, cur = 0;
            int top = 0, pvx = 0, pvy = 0, pvz = 0, elog = 0;
            bool in_empty = false;
            const uint64_t *__restrict__ descriptors = p.descriptors;

            auto locate = [&](int x, int y, int z) -> bool {
                unsigned diff = (unsigned)((x ^ pvx) | (y ^ pvy) | (z ^ pvz));
                if (top > 0 && (diff >> (n - top)) != 0) {
                    top = n - (31 - __clz((int)diff)) - 1;        // deepest level whose node holds both voxels
                    cur = (top == 0) ? root_entry : lds_stack[(top - 1) * kBlockThreads + tid];
                }
                pvx = x; pvy = y; pvz = z;
                for (;;) {
                    const int b = n - top - 1;
                    const int i = ((x >> b) & 1) | (((y >> b) & 1) << 1) | (((z >> b) & 1) << 2);
                    const unsigned masks = (unsigned)cur & 0xffffu;
                    const unsigned bit = 1u << i;
                    if (!(masks & bit)) { elog = b; in_empty = true; return false; }
                    in_empty = false;
                    if (((masks >> 8) & bit) || b == 0) return true;
                    const uint64_t child = (cur >> 16) + (uint64_t)(__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1);
                    const uint64_t d = descriptors[child];
                    c_desc++;
                    cur = make_entry(descriptors, child, d);
                    lds_stack[top * kBlockThreads + tid] = cur;   // level top+1 lives in slot top
                    top++;
                }
            };

            if (SVO) {
                const uint64_t d = descriptors[p.root_index];
                c_desc = 1;
                root_entry = make_entry(descriptors, p.root_index, d);
                cur = root_entry;
                if (vx >= 0 && vy >= 0 && vz >= 0 && vx < p.map_dim[0] && vy < p.map_dim[1] && vz < p.map_dim[2])
                    (void)locate(vx, vy, vz);
            }

            const Vec3 light_pos{p.light_pos[0], p.light_pos[1], p.light_pos[2]};

            while (distance_traveled < max_distance && bounce_count < 2) {    // :357
                c_steps++;
                fmx = itx <= min_cl(ity, itz);                                // :558
                fmy = ity <= min_cl(itz, itx);
                fmz = itz <= min_cl(itx, ity);
                itx += dtx * (float)fmx; ity += dty * (float)fmy; itz += dtz * (float)fmz;     // :559
                vx += sx * fmx; vy += sy * fmy; vz += sz * fmz;                                // :560

                if (vx >= p.map_dim[0] || vy >= p.map_dim[1] || vz >= p.map_dim[2] || vx < 0 || vy < 0 || vz < 0) {   // :563
                    vx -= sx * fmx; vy -= sy * fmy; vz -= sz * fmz;
                    const float k = 1.0f - max_cl((float)distance_traveled / 700.0f, 0.0f);
#pragma unroll
                    for (int c = 0; c < 4; c++) color_accumulator[c] = mix_cl(0.0f, voxel_color[c], k);
                    color_accumulator[3] *= 4.0f;
                    flags |= kFlagOob;
                    break;
                }

                int voxel_data;
                if (SVO) {
                    if (in_empty && ((((unsigned)((vx ^ pvx) | (vy ^ pvy) | (vz ^ pvz))) >> elog) == 0)) {
                        distance_traveled++;                                  // still inside the known-empty node
                        continue;
                    }
                    voxel_data = locate(vx, vy, vz) ? 5 : 0;
                } else {
                    voxel_data = p.map[(long)vx + (long)p.map_dim[0] * ((long)vy + (long)p.map_dim[2] * vz)];   // :569
                    c_map++;
                }

                if (voxel_data == 5 || voxel_data == 6) {                     // :575
                    float fpx = 0.f, fpy = 0.f, fpz = 0.f, tfx = 0.f, tfy = 0.f;
                    float sgx = 1.0f, sgy = 1.0f, sgz = 1.0f;
                    if (fmx == 1) {                                           // :586-599
                        sgx = (float)((double)sgx * -1.0);
                        float z_percent = (itz - (itx - dtx)) / dtz;
                        float y_percent = (ity - (itx - dtx)) / dty;
                        fpx = 1.00001f; fpy = y_percent; fpz = z_percent;
                        tfx = fpy; tfy = fpz;
                    } else if (fmy == 1) {                                    // :601-608
                        sgy = (float)((double)sgy * -1.0);
                        float x_percent = (itx - (ity - dty)) / dtx;
                        float z_percent = (itz - (ity - dty)) / dtz;
                        fpx = x_percent; fpy = 1.00001f; fpz = z_percent;
                        tfx = fpx; tfy = fpz;
                    } else if (fmz == 1) {                                    // :610-618
                        sgz = (float)((double)sgz * -1.0);
                        float x_percent = (itx - (itz - dtz)) / dtx;
                        float y_percent = (ity - (itz - dtz)) / dty;
                        fpx = x_percent; fpy = y_percent; fpz = 1.00001f;
                        tfx = fpx; tfy = fpy;
                    }
                    // :626-643
                    if (rdx > 0.0f) fpx = -fpx + 1.0f;
                    if (rdx < 0.0f) tfx = -tfx + 1.0f;
                    if (rdy > 0.0f) {
                        fpy = -fpy + 1.0f;
                    } else {
                        tfx = (float)(1.0 - (double)tfx);
                        if (fmz == 1) { tfx = 1.0f - tfx; tfy = 1.0f - tfy; }
                    }
                    if (rdz > 0.0f) fpz = -fpz + 1.0f;
                    if (rdz < 0.0f) tfy = -tfy + 1.0f;

                    if (hit_mat == 0 && !shadow_ray) {
                        hit_vx = vx; hit_vy = vy; hit_vz = vz; hit_mat = voxel_data;
                        hit_face = fmx | (fmy << 1) | (fmz << 2);
                    }

                    if ((voxel_data == 5 || voxel_data == 6) && !shadow_ray) {
                        const bool mirror = (voxel_data == 6);
                        // :652-656 / :684-688  tile (5,0) halved, tile (3,4) quartered
                        int tx = (int)(tfx * (float)p.tiles_x) + (int)((mirror ? 3.0f : 5.0f) * (float)p.tiles_x);
                        int ty = (int)(tfy * (float)p.tiles_y) + (int)((mirror ? 4.0f : 0.0f) * (float)p.tiles_y);
                        tx = tx < 0 ? 0 : (tx >= p.atlas_w ? p.atlas_w - 1 : tx);   // undefined in OpenCL: clamp
                        ty = ty < 0 ? 0 : (ty >= p.atlas_h ? p.atlas_h - 1 : ty);
                        const uchar4 t8 = reinterpret_cast<const uchar4 *>(p.atlas)[(long)tx + (long)p.atlas_w * ty];
                        c_tex++;
                        const float div = mirror ? 4.0f : 2.0f;
                        voxel_color[0] += ((float)t8.x / 255.0f) / div;
                        voxel_color[1] += ((float)t8.y / 255.0f) / div;
                        voxel_color[2] += ((float)t8.z / 255.0f) / div;

                        const Vec3 hit_pos{(float)vx + fpx, (float)vy + fpy, (float)vz + fpz};
                        if (!mirror) {                                        // :649-679
                            shadow_ray = true;
                            view_light(color_accumulator, voxel_color,
                                       Vec3{hit_pos.x - light_pos.x, hit_pos.y - light_pos.y, hit_pos.z - light_pos.z},
                                       p.light_rgbi,
                                       Vec3{hit_pos.x - p.cam_pos[0], hit_pos.y - p.cam_pos[1], hit_pos.z - p.cam_pos[2]},
                                       fmx * sx, fmy * sy, fmz * sz);
                            fog_distance = (float)distance_traveled;          // :666
                            if (!p.shadow_rays) break;                        // extension: primary rays only
                            {
                                const float ddx = (float)vx - light_pos.x, ddy = (float)vy - light_pos.y, ddz = (float)vz - light_pos.z;
                                max_distance = (int)((float)distance_traveled + sqrtf(dot3(ddx, ddy, ddz, ddx, ddy, ddz)));   // :667
                            }
                            const Vec3 nd = normalize3(Vec3{light_pos.x - hit_pos.x, light_pos.y - hit_pos.y, light_pos.z - hit_pos.z});
                            rdx = nd.x; rdy = nd.y; rdz = nd.z;               // :670
                            if (rdx == 0.0f || rdy == 0.0f || rdz == 0.0f) { written = false; break; }   // :671-672
                            c_shadow = 1;
                            flags |= kFlagShadowCast;
                            vx -= sx * fmx; vy -= sy * fmy; vz -= sz * fmz;   // :674
                            sx = isign(rdx); sy = isign(rdy); sz = isign(rdz);   // :675
                        } else {                                              // :682-704
                            rdx *= sgx; rdy *= sgy; rdz *= sgz;               // :693
                            if (rdx == 0.0f || rdy == 0.0f || rdz == 0.0f) { written = false; break; }
                            vx -= sx * fmx; vy -= sy * fmy; vz -= sz * fmz;   // :697
                            // :698 precedence quirk: +1 for both signs
                            sx = (-1 * (rdx > 0.0f ? -1 : 0)) - (rdx < 0.0f ? -1 : 0);
                            sy = (-1 * (rdy > 0.0f ? -1 : 0)) - (rdy < 0.0f ? -1 : 0);
                            sz = (-1 * (rdz > 0.0f ? -1 : 0)) - (rdz < 0.0f ? -1 : 0);
                            bounce_count += 1;
                        }
                        dtx = fabsf(1.0f / rdx); dty = fabsf(1.0f / rdy); dtz = fabsf(1.0f / rdz);   // :677 / :700
                        itx = dtx * (hit_pos.x - floorf(hit_pos.x)) * (float)sx;                      // :678 / :701
                        ity = dty * (hit_pos.y - floorf(hit_pos.y)) * (float)sy;
                        itz = dtz * (hit_pos.z - floorf(hit_pos.z)) * (float)sz;
                        itx += dtx * -(itx < 0.0f ? -1.0f : 0.0f);                                    // :679 / :702
                        ity += dty * -(ity < 0.0f ? -1.0f : 0.0f);
                        itz += dtz * -(itz < 0.0f ? -1.0f : 0.0f);
                    } else {                                                  // :707-710
                        color_accumulator[3] = 0.1f;
                        flags |= kFlagShadowHit;
                        break;
                    }
                }
                distance_traveled++;                                          // :714
            }

            if (written) {
                const float k = 1.0f - max_cl(fog_distance / 700.0f, 0.0f);   // :716
#pragma unroll
                for (int c = 0; c < 4; c++) out_color[c] = mix_cl(0.0f, color_accumulator[c], k);
                reinterpret_cast<float4 *>(p.image)[pix] = make_float4(out_color[0], out_color[1], out_color[2], out_color[3]);
                flags |= kFlagWritten;
            } else {
                c_unwritten = 1;
            }
            if (!SVO) c_desc = (unsigned)p.frame[3];   // array branch: the reference's per-pixel get_oct_vox
        }

        int4 *hp = reinterpret_cast<int4 *>(p.hits) + 2 * pix;
        hp[0] = make_int4(hit_vx, hit_vy, hit_vz, hit_mat);
        hp[1] = make_int4(hit_face, flags | ((int)(bounce_count & 3) << 4),
                          distance_traveled, (int)c_desc);
    }

    // ---- per-block counter partials (no global atomics) --------------------
    unsigned vals[7] = {c_primary, c_shadow, c_desc, c_tex, c_map, c_steps, c_unwritten};
#pragma unroll
    for (int k = 0; k < 7; k++) {
        unsigned long long v = vals[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v) atomicAdd(&block_ctr[k], v);
    }
    __syncthreads();
    if (tid < kCtrCount) p.counters[(long)blockIdx.x * kCtrCount + tid] = block_ctr[tid];
}

__global__ void reduce_counters_kernel(const unsigned long long *partials, int nblocks, unsigned long long *out) {
    __shared__ unsigned long long acc[kCtrCount];
    if (threadIdx.x < kCtrCount) acc[threadIdx.x] = 0;
    __syncthreads();
    const int slot = threadIdx.x & (kCtrCount - 1);
    unsigned long long v = 0;
    for (int b = threadIdx.x / kCtrCount; b < nblocks; b += blockDim.x / kCtrCount) v += partials[(long)b * kCtrCount + slot];
    atomicAdd(&acc[slot], v);
    __syncthreads();
    if (threadIdx.x < kCtrCount) out[threadIdx.x] = acc[threadIdx.x];
}

// get_oct_vox(camera voxel) (ray_caster_kernel.cl:140-251, 342-354): identical
// for every pixel, so one lane evaluates it per frame and leaves the bias
// (sub_oct_pos - voxel) * resolution / 2 plus its read count in p.frame.
__global__ void frame_setup_kernel(const RaycastParams p) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int pos[3], corner[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++) pos[a] = (int)floorf(p.cam_pos[a]);
    uint64_t index = p.root_index, d = p.descriptors[index];
    int reads = 1;
    int dimension = 1 << p.log2_dim, res = dimension / 2;
    while (dimension > 1) {
        const int half = dimension / 2;
        int i = 0;
        for (int a = 0; a < 3; a++)
            if (pos[a] >= corner[a] + half) { i |= 1 << a; corner[a] += half; }
        if (!((d >> (16 + i)) & 1ULL)) break;             // not valid: empty node
        if ((d >> (24 + i)) & 1ULL) break;                // valid leaf: early exit, resolution not halved
        dimension = half;
        res /= 2;
        const int before = __popcll((d >> 16) & ((2ULL << i) - 1ULL)) - 1;
        const uint64_t base = (d & 0x8000ULL) ? p.descriptors[index + (d & 0x7fffULL)] : index + (d & 0x7fffULL);
        index = base + (uint64_t)before;
        d = p.descriptors[index];
        reads++;
    }
    for (int a = 0; a < 3; a++) p.frame[a] = (corner[a] - pos[a]) * res / 2;
    p.frame[3] = reads;
}

hipError_t launch_frame_setup(const RaycastParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(frame_setup_kernel, dim3(1), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_raycast(const RaycastParams &p, hipStream_t stream) {
    const int nblocks = p.blocks_x * p.local_tile_rows;
    if (nblocks <= 0) return hipSuccess;
    const int levels = p.svo ? (p.log2_dim > 1 ? p.log2_dim - 1 : 1) : 0;
    const size_t lds = (size_t)levels * kBlockThreads * sizeof(uint64_t);
    if (p.svo)
        hipLaunchKernelGGL(raycast_kernel<true>, dim3(nblocks), dim3(kBlockThreads), lds, stream, p);
    else
        hipLaunchKernelGGL(raycast_kernel<false>, dim3(nblocks), dim3(kBlockThreads), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_reduce_counters(const unsigned long long *partials, int nblocks, unsigned long long *out,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(reduce_counters_kernel, dim3(1), dim3(256), 0, stream, partials, nblocks, out);
    return hipGetLastError();
}

}  // namespace vrc
