// raycast_jump_kernel.hip -- stepping mode 1 ("mode B" of SURVEY 7 D1), the memory-side twin of the exact kernel.
//
// The reference's unfinished octree branch was heading for node-sized jumps (kernels/ray_caster_kernel.cl:525-540:
// intersection_t += delta_t * jump_power * fabs(face_mask), the correction of the other axes commented out).  This
// kernel does that jump the stateless way: the exit of  origin + t * ray_dir  from the empty node it is in is
// t = min_a (plane_a - origin_a) / ray_dir_a.  One node event costs ~100 instructions instead of ~120 DDA steps, so
// the frame is bound by the chain of node events (jump arithmetic, pop, one dependent 8-byte load per level of descent)
// each ray has to get through, not by the float recurrence: this is the mode that puts the memory-side design under load.  It is NOT the reference's float sequence (the reference accumulates
// intersection_t by repeated addition and the hit block reads the accumulated rounding back, :592-614), so it is a
// labelled, opt-in mode (setting stepping_mode = 1), never the headline: its parity statement is bit-exactness against
// its own restatement in oracle/vrc_oracle.c (jump_step & co, same definitions, cited there) plus mismatch statistics
// against the exact mode on every bench frame.
//
// Shared with the exact kernel: ray set-up, hit block, epilogue (raycast_common.hpp), the packed LDS traversal stack
// [level][thread], deferred shading with wave votes, the XCD-aware block -> tile map, per-block counter partials.
#include <hip/hip_runtime.h>

#include "raycast_common.hpp"

namespace vrc {

namespace {
__device__ __forceinline__ uint64_t jump_make_entry(const uint64_t *__restrict__ descriptors, uint64_t index, uint64_t d) {
    uint64_t base = index + (d & 0x7fffULL);
    if (d & 0x8000ULL) base = descriptors[base];          // far pointer: slot holds an absolute index
    return (base << 16) | ((d >> 16) & 0xffffULL);
}
enum JumpLane { jStep = 0, jShade = 1, jDone = 2, jRelight = 3, jDescend = 4, jTable = 5 };

// (coarse_index(): vrc_params.h -- measured, headline frame, table levels 8 / 9 / 10: plain x-fastest order 0.639 / 0.581 / 0.555 ms,
// bricks of 2^3, 4^3, 8^3 cells 0.666 / 0.602 / 0.563-0.573)
}  // namespace

// Coarse table (round 4): one thread per cell walks the canonical descent from the root toward the cell and stores the
// cursor state it ends with -- the node at level `lc`, or the node above whose child toward the cell is empty or a leaf.
__global__ void coarse_build_kernel(const uint64_t *__restrict__ descriptors, uint64_t root_index, int n, int lc, uint64_t *__restrict__ out) {
    const uint64_t cell = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;     // cells in x-fastest order; stored at coarse_index()
    if (cell >> (3 * lc)) return;
    const int sh = n - lc;
    const unsigned ccx = (unsigned)(cell & ((1u << lc) - 1u)), ccy = (unsigned)((cell >> lc) & ((1u << lc) - 1u)), ccz = (unsigned)(cell >> (2 * lc));
    const int x = (int)(ccx << sh), y = (int)(ccy << sh), z = (int)(ccz << sh);
    const uint64_t idx = coarse_index(ccx, ccy, ccz, lc);
    uint64_t cur = jump_make_entry(descriptors, root_index, descriptors[root_index]);
    int top = 0;
    while (top < lc) {
        const int b = n - top - 1;
        const int i = ((x >> b) & 1) | (((y >> b) & 1) << 1) | (((z >> b) & 1) << 2);
        const unsigned masks = (unsigned)cur & 0xffffu, bit = 1u << i;
        if (!(masks & bit) || ((masks >> 8) & bit)) break;    // empty or leaf: the descent block's test finds it from here
        const uint64_t child = (cur >> 16) + (uint64_t)(__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1);
        cur = jump_make_entry(descriptors, child, descriptors[child]);
        top++;
    }
    out[idx] = cur | ((uint64_t)top << kCoarseLevelShift);
}

// blocks per CU the register budget is set for: 8 waves per SIMD (64 VGPRs, 14 dwords of scratch) measured fastest --
// 5 / 6 / 7 blocks: 0.93 / 0.91 / 0.85 ms on the headline frame with the first version of the kernel, 7 / 8 blocks
// 0.678 / 0.667 ms with this one; the kernel waits on dependent loads half of the time (profiles/r02_pmc_mode_b.txt),
// so residency beats spill-free registers.  LDS: 20 KB of stack per block at depth 12 (levels 1..10, the level-11 entry
// is never popped to and is not stored), no static LDS -- 8 blocks fill the CU's 160 KB exactly.
// The blocks one round runs, in order (see the main loop); results never depend on it.  VRC_J = the jump block,
// VRC_D = one level of descent; the argument says whether the block runs unconditionally (true) or only when a lane of
// the wave waits for it.  Measured on the headline frame (same box): J D 0.755 ms, J D D 0.700, J D D D ~0.74,
// D J D 0.700, J D J D 0.758, (J D D) x 2 / x 3 / x 4 0.673 / 0.700 / 0.670 ms.
// Dropped earlier: the jump block by majority vote of the wave (0.86-0.88 vs 0.869 ms then), votes without a descent
// after the jump (1.00 ms); child indices kept in 32 bits for trees below 2^32 descriptors, masks and index in
// separate words of the stack entry (+1 %); testing the landing node at the end of the jump block and after every
// descent load instead of in a round of its own (halves the descent blocks a jump needs, yet +3 % with the best program).
// Also dropped (round 2, s_memtime per block: 60 % of a wave's time passes in the descent blocks, 32 % in the jump
// blocks, 3 % in the hit block): the descent load issued before the jump block and used after it (+4 %: the latency is
// already covered by the other 7 waves of the SIMD); load instructions batched by a vote of the wave until 8 / 16 / 32
// lanes wait (+17 / +23 / +40 %: an idle lane costs more than a thin load).
// Round 3: the one value the compiler spills inside the round loop is the thread's stack slot (every jump / descent
// block starts with a 4-byte scratch reload of it: 6 per round); recomputed instead from the wave's number (a scalar
// register) and v_mbcnt, the hot blocks are free of scratch accesses -- and the frame 1.7 % slower (0.707 vs 0.695 ms):
// with 8 waves per SIMD the reload's latency is covered, the two extra VALU instructions are not.
#ifndef VRC_ROUND_PROGRAM
#define VRC_ROUND_PROGRAM VRC_J(true) VRC_D(true) VRC_D(true) VRC_J(true) VRC_D(true) VRC_D(true)
#endif
// ... and with the coarse table (VRC_T = the table block: one 8-byte load + the descent block's test, for the lanes whose jump
// left their level-lc cell)
#ifndef VRC_ROUND_PROGRAM_COARSE
#define VRC_ROUND_PROGRAM_COARSE VRC_J(true) VRC_T(true) VRC_D(false) VRC_J(true) VRC_T(true) VRC_D(false)
#endif
#ifndef VRC_JUMP_SHADE_THRESHOLD
// lanes that must wait for the hit block before a wave with stepping lanes runs it: 8 / 16 / 32 / 48 / 64 measured
// 0.83 / 0.81 / 0.78 / 0.80 / 0.665 ms (64 = only when no lane of the wave has anything cheaper to do); with the
// rest of the tile's primary hits shaded as soon as its last primary ray is parked (so that late lanes do not wait for
// the early lanes' shadow rays) 24 / 32 / 40 / 48: 0.75 / 0.73 / 0.73 / 0.72 ms -- one hit-block pass per ray generation
// and tile stays the fastest
#define VRC_JUMP_SHADE_THRESHOLD kDefaultShadeThreshold
#endif
#ifndef VRC_JUMP_MIN_BLOCKS
#define VRC_JUMP_MIN_BLOCKS (32 / VRC_TILES_PER_BLOCK)
#endif

// kCoarse: the levels above coarse_log2 are a dense table (RaycastParams::coarse): a jump that leaves its level-lc cell reads
//          the cursor state for the new cell with ONE load instead of popping the stack and descending level by level
template <bool kMulti, bool kCoarse>
__global__ __launch_bounds__(kBlockThreads, VRC_JUMP_MIN_BLOCKS) void raycast_jump_kernel(const RaycastParams p) {
    extern __shared__ uint64_t lds_stack[];               // [level-1][thread], levels 1..n-2
    // the counter partials of the block reuse the stack's memory once every ray of the block has ended (no static LDS:
    // 8 blocks of a depth-12 tree fill the CU's 160 KB exactly)
    unsigned long long *const block_ctr = (unsigned long long *)lds_stack;
    const int tid = threadIdx.x;

    int px, py, brow;
    block_pixel(p, px, py, brow);
    const bool in_image = px < p.width && py < p.height;
    const long pix = (long)px + (long)p.width * brow;

    Ray r;
    r.pix0 = wave_first_pixel(pix);
    unsigned c_primary = 0, c_desc = 0, c_unwritten = 0, c_steps = 0;
    int mode = jDone;
    auto ended = [&]() -> int { return (kMulti && more_lights(r, p)) ? jRelight : jDone; };

    // the ray as a line: origin + t * ray_dir, and the empty node [corner, corner + size)^3 the voxel is in
    float ox = 0.f, oy = 0.f, oz = 0.f, ivx = 0.f, ivy = 0.f, ivz = 0.f;
    int cx = 0, cy = 0, cz = 0, size = 1;
    float t_exit = 0.0f;

    const int n = p.log2_dim;
    const int lc = kCoarse ? p.coarse_log2 : 0, csh = n - lc;            // table level, log2 of the cell size
    const uint64_t *__restrict__ descriptors = p.descriptors;
    // the root's entry is the same for every ray of the frame: it lives in scalar registers
    uint64_t root_entry;
    {
        const uint64_t e = jump_make_entry(descriptors, p.root_index, descriptors[p.root_index]);
        root_entry = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(e >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)e);
    }
    uint64_t cur = 0;
    int top = 0, pvx = 0, pvy = 0, pvz = 0;

    // returns b >= 0: voxel lies in an empty node of size 2^b;  -1: voxel is solid   (canonical traversal)
    auto locate = [&](int x, int y, int z) -> int {
        const unsigned diff = (unsigned)((x ^ pvx) | (y ^ pvy) | (z ^ pvz));
        if (top > 0 && (diff >> (n - top)) != 0) {
            top = n - (31 - __clz((int)diff)) - 1;
            cur = (top == 0) ? root_entry : lds_stack[(top - 1) * kBlockThreads + tid];
        }
        pvx = x; pvy = y; pvz = z;
        for (;;) {
            const int b = n - top - 1;
            const int i = ((x >> b) & 1) | (((y >> b) & 1) << 1) | (((z >> b) & 1) << 2);
            const unsigned masks = (unsigned)cur & 0xffffu;
            const unsigned bit = 1u << i;
            if (!(masks & bit)) return b;
            if (((masks >> 8) & bit) || b == 0) return -1;
            const uint64_t child = (cur >> 16) + (uint64_t)(__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1);
            const uint64_t d = descriptors[child];
            c_desc++;
            cur = jump_make_entry(descriptors, child, d);
            if (top < n - 2) lds_stack[top * kBlockThreads + tid] = cur;     // the deepest entry is never popped to
            top++;
        }
    };
    auto solid_material = [&](int x, int y, int z) -> int {
        if (!p.attach_lookup || top != n - 1) return 5;
        uint64_t node = p.root_index;
        if (top > 0) {
            const uint64_t parent = (top == 1) ? root_entry : lds_stack[(top - 2) * kBlockThreads + tid];
            const int slot = ((x >> 1) & 1) | (((y >> 1) & 1) << 1) | (((z >> 1) & 1) << 2);
            node = (parent >> 16) + (uint64_t)(__popc((unsigned)parent & 0xffu & ((2u << slot) - 1u)) - 1);
        }
        const uint64_t a = p.attachments[p.attach_lookup[node]];
        return (int)(int8_t)(a >> (8 * ((x & 1) | ((y & 1) << 1) | ((z & 1) << 2))));
    };
    auto set_node = [&](int s) {
        size = s;
        cx = r.vx & ~(s - 1); cy = r.vy & ~(s - 1); cz = r.vz & ~(s - 1);
    };
    auto set_ray = [&](float x, float y, float z) {
        ox = x; oy = y; oz = z;
        // (a denormal component would give inf, and 0 * inf at a ray that starts exactly on a plane is NaN: the jump would
        // have no minimum and make no progress.  Clamped to the largest float: such an axis crosses at t = 0 or never.)
        ivx = fminf(fmaxf(1.0f / r.rdx, -3.402823466e38f), 3.402823466e38f);
        ivy = fminf(fmaxf(1.0f / r.rdy, -3.402823466e38f), 3.402823466e38f);
        ivz = fminf(fmaxf(1.0f / r.rdz, -3.402823466e38f), 3.402823466e38f);
    };
    // a restarted ray starts where the reference's restart arithmetic (:677-679, :700-702) puts it: hit_pos mirrored
    // inside the restart voxel (oracle/vrc_oracle.c jump_restart_ray)
    auto restart_ray = [&](float hx, float hy, float hz) {
        set_ray((float)r.vx + (1.0f - (hx - floorf(hx))), (float)r.vy + (1.0f - (hy - floorf(hy))),
                (float)r.vz + (1.0f - (hz - floorf(hz))));
        set_node(1);
    };

    if (in_image) {
        if (!ray_setup(r, p, pix)) {
            c_unwritten = 1;
        } else {
            // ray_setup added the octree bias of :353-354 to intersection_t; this mode never reads intersection_t
            c_primary = 1;
            c_desc = 1;                                   // the root read of this ray (canonical count)
            cur = root_entry;
            int b = -1;
            if (r.vx >= 0 && r.vy >= 0 && r.vz >= 0 && r.vx < p.map_dim[0] && r.vy < p.map_dim[1] && r.vz < p.map_dim[2])
                b = locate(r.vx, r.vy, r.vz);
            set_node(b >= 0 ? 1 << b : 1);
            set_ray(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
            mode = (r.distance_traveled < r.max_distance) ? jStep : jDone;    // :357 guard
        }
    }

    // (compiled in: this kernel does not read the live `shade_threshold` setting -- every value below a full tile measured
    // slower, see VRC_JUMP_SHADE_THRESHOLD above; the exact kernel's untuned instances do honour it)
    const int shade_threshold = VRC_JUMP_SHADE_THRESHOLD;
    int rounds_left = p.watchdog_rounds;
#ifdef VRC_SCHED_STATS
    // profiling build only: how often a wave ran each block and how many lanes had work in it
    unsigned s_rounds = 0, s_jwaves = 0, s_jlanes = 0, s_dwaves = 0, s_dlanes = 0, s_shade_passes = 0;
#endif
    for (;;) {
#ifdef VRC_SCHED_STATS
        if ((tid & 63) == 0) s_rounds++;
#endif
        // One round = VRC_ROUND_PROGRAM: the node-exit JUMP (pure arithmetic + the pop to the common ancestor, an LDS
        // read) for the lanes that wait for it, then levels of the DESCENT toward the voxel a lane landed in (one
        // dependent 8-byte load each).  A jump is followed by ~2 descent steps on average (tools/mode_b_stats.py).
        auto jump_block = [&]() {
#ifdef VRC_SCHED_STATS
        if ((tid & 63) == 0) s_jwaves++;
        s_jlanes += mode == jStep;
#endif
        if (mode == jStep) {
                // straight-line code on purpose: both sides of every choice are a couple of instructions, a branch costs more
                const int smx = (r.sx - 1) >> 1, smy = (r.sy - 1) >> 1, smz = (r.sz - 1) >> 1;   // 0 for step +1, -1 for step -1
                const int pix = cx + (size & ~smx), piy = cy + (size & ~smy), piz = cz + (size & ~smz);   // the exit planes
                const float tx = ((float)pix - ox) * ivx, ty = ((float)piy - oy) * ivy, tz = ((float)piz - oz) * ivz;
                float t = tx;
                t = ty < t ? ty : t;
                t = tz < t ? tz : t;
                const int mx = tx <= t, my = ty <= t, mz = tz <= t;
                const int hi_x = cx + size - 1, hi_y = cy + size - 1, hi_z = cz + size - 1;
                const int qx = min(max((int)floorf(ox + t * r.rdx), cx), hi_x), qy = min(max((int)floorf(oy + t * r.rdy), cy), hi_y),
                          qz = min(max((int)floorf(oz + t * r.rdz), cz), hi_z);
                const int nx = mx ? pix + smx : qx, ny = my ? piy + smy : qy, nz = mz ? piz + smz : qz;   // plane, or plane - 1 going down
                const int steps = abs(nx - r.vx) + abs(ny - r.vy) + abs(nz - r.vz);
                if (r.distance_traveled + steps - 1 >= r.max_distance) {     // the :357 guard ends the ray inside the node
                    c_steps += (unsigned)(r.max_distance - r.distance_traveled);
                    r.distance_traveled = r.max_distance;
                    mode = ended();
                } else {
                    c_steps += (unsigned)steps;
                    r.distance_traveled += steps - 1;
                    r.vx = nx; r.vy = ny; r.vz = nz;
                    r.fmx = mx; r.fmy = my; r.fmz = mz;
                    t_exit = t;
                    // :563 any(voxel >= map_dim) || any(voxel < 0): one unsigned compare per axis covers both sides
                    if ((unsigned)nx >= (unsigned)p.map_dim[0] || (unsigned)ny >= (unsigned)p.map_dim[1] || (unsigned)nz >= (unsigned)p.map_dim[2]) {
                        oob_exit(r);                              // :563-568
                        mode = ended();
                    } else {
                        // pop to the deepest level whose node holds both the voxel located last and the new one (after a
                        // redirect the two are not neighbours: the cursor still sits at the hit voxel)
                        const unsigned diff = (unsigned)((nx ^ pvx) | (ny ^ pvy) | (nz ^ pvz));
                        pvx = nx; pvy = ny; pvz = nz;
                        if (kCoarse && (diff >> csh) != 0) {
                            mode = jTable;                    // another level-lc cell: the cursor comes from the table
                        } else {
                            if (top > 0 && (diff >> (n - top)) != 0) {
                                top = n - (31 - __clz((int)diff)) - 1;       // (>= lc with the table: the voxels share a cell)
                                cur = (top == 0) ? root_entry : lds_stack[(top - 1) * kBlockThreads + tid];
                            }
                            mode = jDescend;
                        }
                    }
                }
            }
        };
        auto descend_body = [&]() {
                const int b = n - top - 1;
                const int i = ((r.vx >> b) & 1) | (((r.vy >> b) & 1) << 1) | (((r.vz >> b) & 1) << 2);
                const unsigned masks = (unsigned)cur & 0xffffu;
                const unsigned bit = 1u << i;
                if (!(masks & bit)) {                             // the voxel lies in an empty node of size 2^b
                    set_node(1 << b);
                    r.distance_traveled++;                        // :714
                    mode = (r.distance_traveled < r.max_distance) ? jStep : ended();
                } else if (((masks >> 8) & bit) || b == 0) {      // solid
                    set_node(1);
                    const int m5 = solid_material(r.vx, r.vy, r.vz);
                    if (m5 == 5 || m5 == 6) {                     // :575
                        mode = jShade;                            // the hit block is deferred (the material is looked up again there)
                    } else {                                      // any other material is passed through
                        r.distance_traveled++;
                        mode = (r.distance_traveled < r.max_distance) ? jStep : ended();
                    }
                } else {                                          // one level down: one dependent load
                    const uint64_t child = (cur >> 16) + (uint64_t)(__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1);
                    const uint64_t d = descriptors[child];
                    c_desc++;
                    cur = jump_make_entry(descriptors, child, d);
                    if (top < n - 2) lds_stack[top * kBlockThreads + tid] = cur;     // the deepest entry is never popped to
                    top++;
                }
        };
        auto descend_block = [&]() {
#ifdef VRC_SCHED_STATS
        if ((tid & 63) == 0) s_dwaves++;
        s_dlanes += mode == jDescend;
#endif
        if (mode == jDescend) descend_body();
        };
        // the table block: the cursor state of the cell the ray landed in (one 8-byte load, counted like a descriptor read),
        // then the descent block's test right away -- in the coarse empty space above the terrain that test ends the event
        auto table_block = [&]() {
        if (kCoarse && mode == jTable) {
                const uint64_t e = p.coarse[coarse_index((unsigned)(r.vx >> csh), (unsigned)(r.vy >> csh), (unsigned)(r.vz >> csh), lc)];
                c_desc++;
                cur = e & ((1ULL << kCoarseLevelShift) - 1ULL);
                top = (int)(e >> kCoarseLevelShift);
                if (top == lc) lds_stack[(lc - 1) * kBlockThreads + tid] = cur;      // pops inside the cell end here (lc <= n - 2)
                mode = jDescend;
                descend_body();
            }
        };
        // later blocks of the round run only if some lane of the wave waits for them (a wave-uniform branch)
#define VRC_J(always) if ((always) || __ballot(mode == jStep) != 0ULL) jump_block();
#define VRC_D(always) if ((always) || __ballot(mode == jDescend) != 0ULL) descend_block();
#define VRC_T(always) if ((always) || __ballot(mode == jTable) != 0ULL) table_block();
        if (kCoarse) { VRC_ROUND_PROGRAM_COARSE } else { VRC_ROUND_PROGRAM }
#undef VRC_J
#undef VRC_D
#undef VRC_T
        const unsigned long long st = __ballot(mode == jStep || mode == jDescend || mode == jTable);
        const unsigned long long sh = __ballot(mode == jShade || (kMulti && mode == jRelight));
        if ((st | sh) == 0ULL || --rounds_left < 0) break;

        // ---- hit block (:575-711): runs when many lanes wait for it or nothing cheaper is left
        if (sh != 0ULL && (st == 0ULL || __popcll(sh) >= shade_threshold)) {
#ifdef VRC_SCHED_STATS
            if ((tid & 63) == 0) s_shade_passes++;
#endif
            if (kMulti && mode == jRelight) {                 // back to the first strike for the next light
                r.light_index++;
                if (!light_from_strike(r, p, r.light_index, true)) {
                    mode = jDone;                             // :671-672, pixel left unwritten
                } else {
                    restart_from(r, strike_pos(r));
                    const Vec3 hp = strike_pos(r);
                    restart_ray(hp.x, hp.y, hp.z);
                    r.distance_traveled = r.kdist + 1;        // as if the strike iteration had just finished (:714)
                    mode = (r.distance_traveled < r.max_distance) ? jStep : ended();
                }
            } else if (mode == jShade) {
                // delta_t and the intersection_t the hit block reads (:586-618), rebuilt from the exit (kept out of the
                // traversal loop's registers; recomputed rather than taken from iv: iv is clamped, delta_t is not)
                r.dtx = fabsf(1.0f / r.rdx); r.dty = fabsf(1.0f / r.rdy); r.dtz = fabsf(1.0f / r.rdz);
                r.itx = r.fmx ? t_exit + r.dtx : ((float)(r.sx > 0 ? r.vx + 1 : r.vx) - ox) * ivx;
                r.ity = r.fmy ? t_exit + r.dty : ((float)(r.sy > 0 ? r.vy + 1 : r.vy) - oy) * ivy;
                r.itz = r.fmz ? t_exit + r.dtz : ((float)(r.sz > 0 ? r.vz + 1 : r.vz) - oz) * ivz;
                if (hit_block<kMulti>(r, solid_material(r.vx, r.vy, r.vz), p)) {
                    mode = ended();
                } else {
                    restart_ray(r.hpx, r.hpy, r.hpz);
                    r.distance_traveled++;                    // :714
                    mode = (r.distance_traveled < r.max_distance && (r.counts >> 16) < 2) ? jStep : ended();
                }
            }
        }
    }

    __syncthreads();
    if (tid < kCtrCount) block_ctr[tid] = 0;
    __syncthreads();
    if (rounds_left < 0 && (tid & 63) == 0) {
        atomicAdd(&block_ctr[kCtrWatchdog], 1ULL);
        if (p.watchdog_flag) __hip_atomic_store(p.watchdog_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    unsigned c_tex = 0, c_shadow = 0;
    if (in_image) {
        if (c_primary) {
            c_tex = r.counts & 0xffu; c_shadow = (r.counts >> 8) & 0xffu;
            if (!r.written) c_unwritten = 1;
        }
        ray_finish(r, p, c_desc);
    }
#ifdef VRC_SCHED_STATS
    atomicAdd(&block_ctr[kCtrWaveIters], (unsigned long long)s_rounds);
    atomicAdd(&block_ctr[kCtrBursts], (unsigned long long)s_jwaves);
    atomicAdd(&block_ctr[kCtrEventPasses], (unsigned long long)s_jlanes);
    atomicAdd(&block_ctr[kCtrEventLanes], (unsigned long long)s_dwaves);
    atomicAdd(&block_ctr[kCtrShadePasses], (unsigned long long)s_shade_passes);
    atomicAdd(&block_ctr[kCtrShadeLanes], (unsigned long long)s_dlanes);
#endif
    const unsigned vals[7] = {c_primary, c_shadow, c_desc, c_tex, 0u, c_steps, c_unwritten};
    publish_counters(p, block_ctr, vals, (int)threadIdx.x);
}

hipError_t launch_raycast_jump(const RaycastParams &p, hipStream_t stream) {
    (void)hipGetLastError();                 // an error an earlier call left behind is not this launch's
    const int nblocks = p.blocks_x * p.local_tile_rows;
    if (nblocks <= 0) return hipSuccess;
    const int levels = p.log2_dim > 2 ? p.log2_dim - 2 : 1;     // >= the counter partials that reuse the memory
    const size_t lds = (size_t)levels * kBlockThreads * sizeof(uint64_t);
    const bool coarse = p.coarse != nullptr && p.coarse_log2 >= 1 && p.coarse_log2 <= p.log2_dim - 2;
    if (coarse) {
        if (p.light_count > 1) hipLaunchKernelGGL((raycast_jump_kernel<true, true>), dim3(nblocks), dim3(kBlockThreads), lds, stream, p);
        else hipLaunchKernelGGL((raycast_jump_kernel<false, true>), dim3(nblocks), dim3(kBlockThreads), lds, stream, p);
    } else {
        if (p.light_count > 1) hipLaunchKernelGGL((raycast_jump_kernel<true, false>), dim3(nblocks), dim3(kBlockThreads), lds, stream, p);
        else hipLaunchKernelGGL((raycast_jump_kernel<false, false>), dim3(nblocks), dim3(kBlockThreads), lds, stream, p);
    }
    return hipGetLastError();
}

// the coarse table of a tree: 2^(3 lc) entries at `out` (device memory)
hipError_t launch_coarse_build(const uint64_t *descriptors, uint64_t root_index, int log2_dim, int lc, uint64_t *out, hipStream_t stream) {
    (void)hipGetLastError();
    const uint64_t cells = 1ULL << (3 * lc);
    hipLaunchKernelGGL(coarse_build_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, descriptors, root_index, log2_dim, lc, out);
    return hipGetLastError();
}

}  // namespace vrc
