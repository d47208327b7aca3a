// raycast_kernel.hip -- the per-pixel hot path for gfx950 (MI355X, wave64).
//
// Replaces kernels/ray_caster_kernel.cl of the reference: pitch/yaw of the
// table ray (:276-294), Amanatides-Woo set-up (:298-354), the step loop
// (:357,555-570,714), the hit block (:575-711) and the fog/write epilogue
// (:716-721).  Two kernels share raycast_common.hpp:
//
//   raycast_array_kernel  occupancy from the dense char map (array branch, :569)
//   raycast_svo_kernel    occupancy from the 64-bit child-descriptor array
//                         (Octree.h:89-94), the headline path
//
// SVO kernel design (MI355X-first, not a translation of the OpenCL loop):
//   * The float recurrence intersection_t += delta_t * face_mask (:559) is kept
//     step for step -- it is what makes hits bit-identical to the reference --
//     but while a ray is inside a node the octree says is empty, a step touches
//     no memory and no voxel coordinates: per axis a countdown of steps to the
//     node face replaces voxel += step, the bounds test and the lookup
//     (17 VALU ops per step, -fno-slp-vectorize: v_pk_*_f32 is slow on gfx950).  v_fma(dt, mask, t) with mask in {0,1} is exact,
//     so it equals the reference's unfused multiply-add bit for bit.
//   * Leaving a node is an EVENT: reconstruct the voxel, pop the per-ray stack
//     (LDS, [level][thread], conflict-free) to the common ancestor, descend.
//     The wave works in ROUNDS: every stepping lane runs up to burst_steps
//     iterations (one basic block, no exec juggling), then all parked lanes are
//     serviced together; the ~600-instruction hit block is deferred until
//     nothing cheaper is left, so it runs a few times per wave with most lanes
//     active.  __ballot votes pick the phases.
//   * Opt-in (setting jump_min_run): exact_jump.hpp replaces a whole empty
//     stretch by closed forms that reproduce the float recurrence and the
//     iteration count bit for bit.
//   * One wavefront = one 8x8 pixel tile (coherent rays walk the same nodes and
//     L1/L2 lines), 4 tiles per 256-thread block; blocks go to the XCDs round-robin and the block -> tiles
//     mapping (block_pixel) gives every XCD the same sky/ground mix.
#include <hip/hip_runtime.h>

#include <mutex>

#include "exact_jump.hpp"
#include "raycast_common.hpp"
#include "safe_run.hpp"

#ifndef VRC_RELIGHT_THRESHOLD
#define VRC_RELIGHT_THRESHOLD 64      // lanes that must wait for the next light before a wave with stepping lanes serves them
#endif

namespace vrc {

static_assert(3 * 4 * (int)(sizeof(JumpWord) / sizeof(uint32_t)) == kJumpTableDwordsPerLane, "vrc_api.cpp sizes the global Euclid tables (4 ring rows x 3 pairs x 8 bytes) with kJumpTableDwordsPerLane");

// Packed stack entry of one descriptor level:
//   bits 0-7 valid mask, 8-15 leaf mask, 16-63 absolute index of the first kept child
__device__ __forceinline__ uint64_t make_entry(const uint64_t *__restrict__ descriptors, uint64_t index,
                                               uint64_t d) {
    uint64_t base = index + (d & 0x7fffULL);
    if (d & 0x8000ULL) base = descriptors[base];          // far pointer: slot holds an absolute index
    return (base << 16) | ((d >> 16) & 0xffffULL);        // (leaf<<8 | valid) are bits 16..31 of d
}

// ---------------------------------------------------------------------------
// dense-array branch
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void raycast_array_kernel(const RaycastParams p) {
    __shared__ unsigned long long block_ctr[kCtrCount];
    if (threadIdx.x < kCtrCount) block_ctr[threadIdx.x] = 0;
    __syncthreads();

    int px, py, brow;
    block_pixel(p, px, py, brow);
    unsigned c_primary = 0, c_desc = 0, c_map = 0, c_steps = 0, c_unwritten = 0, c_tex = 0, c_shadow = 0;

    const long pix = (long)px + (long)p.width * brow;
    Ray r;
    r.pix0 = wave_first_pixel(pix);
    if (px < p.width && py < p.height) {
        if (!ray_setup(r, p, pix)) {
            c_unwritten = 1;
        } else {
            c_primary = 1;
            c_desc = (unsigned)p.frame[3];                // the reference's per-pixel get_oct_vox (:342)
            for (;;) {
            while (r.distance_traveled < r.max_distance && (r.counts >> 16) < 2) {          // :357
                c_steps++;
                r.fmx = r.itx <= min_cl(r.ity, r.itz);                                      // :558
                r.fmy = r.ity <= min_cl(r.itz, r.itx);
                r.fmz = r.itz <= min_cl(r.itx, r.ity);
                r.itx += r.dtx * (float)r.fmx; r.ity += r.dty * (float)r.fmy; r.itz += r.dtz * (float)r.fmz;   // :559
                r.vx += r.sx * r.fmx; r.vy += r.sy * r.fmy; r.vz += r.sz * r.fmz;           // :560
                if (r.vx >= p.map_dim[0] || r.vy >= p.map_dim[1] || r.vz >= p.map_dim[2] || r.vx < 0 || r.vy < 0 || r.vz < 0) {
                    oob_exit(r);                                                            // :563-568
                    break;
                }
                const int voxel_data =
                    p.map[(long)r.vx + (long)p.map_dim[0] * ((long)r.vy + (long)p.map_dim[2] * r.vz)];   // :569
                c_map++;
                if (voxel_data == 5 || voxel_data == 6)                                     // :575
                    if (hit_block<true>(r, voxel_data, p)) break;
                r.distance_traveled++;                                                      // :714
            }
                // multi-light extension: back to the first strike for the next light (never with one light)
                if (!more_lights(r, p)) break;
                r.light_index++;
                if (!light_from_strike(r, p, r.light_index, true)) break;
                restart_from(r, strike_pos(r));
                r.distance_traveled = r.kdist + 1;        // as if the strike iteration had just finished (:714)
            }
            if (!r.written) c_unwritten = 1;
            c_tex = r.counts & 0xffu; c_shadow = (r.counts >> 8) & 0xffu;
        }
        ray_finish(r, p, c_desc);
    }
    const unsigned vals[7] = {c_primary, c_shadow, c_desc, c_tex, c_map, c_steps, c_unwritten};
    publish_counters(p, block_ctr, vals, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------
// SVO branch
// ---------------------------------------------------------------------------
#ifndef VRC_MIN_BLOCKS
#define VRC_MIN_BLOCKS (24 / VRC_TILES_PER_BLOCK)   // blocks per CU the register budget allows: 6 waves per SIMD
#endif
enum LaneMode { kStep = 0, kEvent = 1, kShade = 2, kDone = 3, kRelight = 4 };

#ifdef VRC_SCHED_STATS
// profiling build only: lane steps by run length (iterations between two node events), tools/run_hist.py
//   [0][b] node events whose run had 2^b <= L < 2^(b+1) iterations   [1][b] the iterations of those runs
//   [2][b] the same iterations by the bucket of the ESTIMATE made when the node was entered (run_estimate)
__device__ unsigned long long g_run_hist[3][32];
// [0] jump-block passes (waves)  [1] lane jumps  [2] iterations they covered  [3] jumps that left the node  [4] jumps that
// ended at the step cap  [5] pair solves (extended Euclid)  [6] lanes that wanted a jump  [7] rounds
__device__ unsigned long long g_jump_stats[12];   // [8], [9]: wave-iterations an ungated safe-run prefix could cover / all
#endif
#ifdef VRC_TIME_STATS
// profiling build only: shader-clock ticks (s_memtime) a wave spends per phase of a round, summed over waves
//   [0] jump estimate + vote  [1] wave-wide Euclid fill  [2] jump block  [3] safe run  [4] single step + exact loop
//   [5] node events  [6] relight + hit block  [7] set-up and epilogue  [8] whole kernel
__device__ unsigned long long g_time_stats[16];
#define VRC_TICK(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if ((tid & 63) == 0) t_acc[slot] += now_ - t_last; t_last = now_; } while (0)
#else
#define VRC_TICK(slot) do { } while (0)
#endif

// kJump: lanes whose node promises a long run take it in one closed-form jump of exact_jump.hpp (setting jump_min_run)
// kMulti: multi-light extension (setting light_count > 1): a finished shadow ray parks the lane in kRelight and
//         the shade phase restarts it from the first strike toward the next light
// kTuned: the scheduling knobs are at their defaults (vrc_api.cpp), so they are compile-time constants here instead
//         of kernarg fields held in SGPRs for the whole kernel -- the kernel sits at the SGPR and VGPR limits, and a
//         single extra live scalar costs several per cent in spills
// (the jump block's temporaries on top of the step-loop state: the instances with it run one block per CU fewer -- 96
// instead of 80 registers; measured 2.81 -> 2.61 ms on the headline frame, without it the other way round)
#ifndef VRC_MIN_BLOCKS_JUMP
#define VRC_MIN_BLOCKS_JUMP (VRC_MIN_BLOCKS - 1)
#endif
// s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4): (size - 1) << 11 | offset << 6 | register id 20; the XCD this wave runs on, 0..7
constexpr int kHwRegXccId = (3 << 11) | (0 << 6) | 20;
// kLdsTab: the Euclid tables of the jumps live in LDS behind the traversal stack ([ring row][pair][thread], 9 KB per block)
//         instead of in the global table buffer -- chosen by the launch when stack + tables of 5 blocks fit the CU's LDS
//         (depth <= 12): no slot to take, 12 bytes less scratch, 2.57 -> 2.51 ms on the headline frame; deeper trees keep the
//         global tables (with the rows in LDS they would run at 4 blocks per CU: depth 13 3.36 -> 3.56 ms, depth 16 6.85 -> 7.42)
// kCoarse: the levels above p.coarse_log2 are read from the dense table of the tree's top (RaycastParams::coarse, built by
//         raycast_jump_kernel.hip coarse_build_kernel): a voxel in another level-lc cell than the voxel located last costs ONE
//         load instead of a pop and a level-by-level descent with a dependent load each.  The descriptor-read COUNT stays the
//         canonical one of SURVEY 8d (the descents the table stands for are counted by their number: level found - level of
//         the common ancestor), so counters and hit records are those of the plain traversal, bit for bit.  The LDS stack
//         then only holds the levels from lc down (slot = level - lc).
// kBox: an empty node is widened to the empty BOX the tree's builder-side pass found around it (empty_boxes.hip: RaycastParams::boxes
//         per (descriptor, empty child), box_aux per table cell) instead of over its empty siblings: a ray that leaves or approaches
//         a surface no longer pays a node event per octree level.  The step loop, its float sequence and the iteration count are
//         what they were -- inside an empty box the loop reads no occupancy -- so frames and hit records stay bit-identical; only
//         the descriptor-read counter now counts the reads this traversal makes, not SURVEY 8d's canonical ones (setting
//         empty_boxes = 0 renders with the canonical counter).  The cursor keeps the INDEX of the descriptor whose masks it holds
//         in a second LDS array beside the stack ([level - lc][thread], one dword).
template <bool kJump, bool kMulti, bool kTuned, int kLdsRows = 0, bool kCoarse = false, bool kBox = false>
#ifndef VRC_MIN_BLOCKS_JUMP_MULTI
#define VRC_MIN_BLOCKS_JUMP_MULTI VRC_MIN_BLOCKS_JUMP
#endif
__global__ __launch_bounds__(kBlockThreads, kJump ? (kMulti ? VRC_MIN_BLOCKS_JUMP_MULTI : VRC_MIN_BLOCKS_JUMP) : VRC_MIN_BLOCKS) void raycast_svo_kernel(const RaycastParams p) {
    // kLdsRows: rows of the Euclid-table ring in LDS (kLdsTab), 0 = the tables live in global memory (ring of 4).  3 rows x 3 pairs x
    // 8 bytes = 72 bytes per lane beside a stack of up to 4 levels with the boxes' index array (6 without) inside the 128 bytes a lane
    // has at 5 blocks per CU; 2 rows = 48 bytes for the deeper stacks of trees from depth 15 on WITH boxes (5 - 6 levels x 12 bytes) --
    // two rows cost 1 - 2 % against three (more row builds), tables in global memory 25 %
    constexpr bool kLdsTab = kLdsRows > 0;
    static_assert(kLdsRows == 0 || kLdsRows == 3 || (kLdsRows == 2 && kBox), "ring of 3 rows, or 2 for the box instances");
    static_assert(kJump || !kLdsTab, "tables exist for the jump instances only");
    static_assert(kCoarse || !kJump, "the jump instances read the tree's top from the coarse table (no table: depth < 5 or coarse_log2 = 0, where jumps never pay)");
    static_assert(kTuned || kMulti, "the instances with run-time scheduling knobs exist once, with the multi-light code compiled in (it renders one light too)");
    static_assert(kCoarse || !kBox, "the boxes hang on the coarse table's cells");
    extern __shared__ uint64_t lds_stack[];               // [level-1][thread], levels 1..n-1  (kCoarse: [level-lc][thread], levels lc..n-1)
    __shared__ unsigned long long block_ctr[kCtrCount];
    __shared__ int s_jump_slot;
    const int tid = threadIdx.x;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the epilogue rebuilds the thread index from it)
    if (tid < kCtrCount) block_ctr[tid] = 0;
    if (kLdsTab) {
        if (tid == 0) s_jump_slot = 0;
    } else if (kJump && tid == 0) {
        // the block's slot in the table buffer: tables exist for RESIDENT blocks only (they stay in the L2 / MALL), so a
        // block takes a free slot when it starts and gives it back at its end.  The slots are divided among the XCDs and a
        // block only ever takes one of the XCD it RUNS on (HW_REG_XCC_ID, not a guess from blockIdx): an XCD's L2 is not
        // coherent with the others', so a slot handed from a block on one XCD to a block on another could be overwritten
        // by the first L2's late write-back after the second block's lines were evicted (seen as 1 frame in 4000 with a
        // few tie counts off by one when small frames made blocks look for slots anywhere; tests/soak_jumps_gpu.py).
        // An eighth holds as many slots as an XCD holds blocks (vrc_api.cpp ensure_jump_cache sizes it from the device's CU
        // count and threads per CU), so a free slot is always there; the search is bounded all the same -- after one full
        // sweep of its eighth without a free slot (a part with more CUs per XCD, aliased XCC ids, a flag a killed launch left
        // set) the block gives up and steps voxel by voxel, which renders the same frame.
        // The tables an XCD works on are lines its L2 already holds and rewrites in place.
        const unsigned per = (unsigned)p.jump_slot_count >> 3;
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(kHwRegXccId) & 7u;
        const unsigned lo = xcc * per, hi = lo + per;
        unsigned i = lo + (unsigned)(((unsigned long long)(blockIdx.x >> 3) * 2654435761ULL) % per), tries = 0;
        while (atomicCAS(&p.jump_slots[i], 0u, 1u) != 0u) {
            if (++i == hi) i = lo;
            if (++tries >= per) { i = 0xffffffffu; break; }
        }
        s_jump_slot = (int)i;                             // -1: no slot
    }
    __syncthreads();

    int px, py, brow;
    block_pixel(p, px, py, brow);
    const bool in_image = px < p.width && py < p.height;
    const long pix = (long)px + (long)p.width * brow;

    Ray r;
    r.pix0 = wave_first_pixel(pix);
    r.flags = 0;                                          // (kFlagPrimary / kFlagUnwritten / kFlagBroke live here too)
    unsigned c_desc = 0;
    int mode = kDone, mat = 5;
    bool t_unsafe = true;                                 // see arith_mask below
    int steps_base = 0;                                   // iterations of the segments before the last reset (kMulti)
    // a ray segment ended (:357 guard, :563-568, :707-710)
    // (more_lights(r, p), with its light_index part kept as a flag bit by note_lights(): light_index is cold state, and read here it
    // was reloaded from scratch in every step phase of the multi-light instances)
    auto ended = [&]() -> int { return (kMulti && r.shadow_ray && r.written && (r.flags & kFlagMoreLights)) ? kRelight : kDone; };
    auto note_lights = [&]() {
        if (kMulti) r.flags = (r.flags & ~kFlagMoreLights) | ((p.shadow_rays && r.light_index + 1 < p.light_count) ? kFlagMoreLights : 0);
    };

    // stepping state while inside a known-empty node: countdown of steps to the
    // node face per axis, voxel_a = base_a - s_a * n_a
    float nx = 1.0f, ny = 1.0f, nz = 1.0f;               // exact: counts < 2^24
    float fxf = 0.0f, fyf = 0.0f, fzf = 0.0f;             // face_mask of the last step, as 0.0 / 1.0
    int bx = 0, by = 0, bz = 0;

    // SVO cursor (canonical traversal of SURVEY 8d)
    const int n = p.log2_dim;
    const uint64_t *__restrict__ descriptors = p.descriptors;
    uint64_t root_entry = 0, cur = 0;
    int top = 0, pvx = 0, pvy = 0, pvz = 0;

    // returns b >= 0: voxel lies in an empty node of size 2^b;  -1: voxel is solid
    const int lc = kCoarse ? p.coarse_log2 : 0, csh = n - lc;           // table level, log2 of its cell size
    const int sbase = kCoarse ? lc : 1;                   // level of stack slot 0
    // kBox: the descriptor index of every stack level, behind the stack; boxw = the box word of the empty node locate() found
    uint32_t *const lds_own = reinterpret_cast<uint32_t *>(lds_stack + (size_t)(kCoarse ? n - lc : (n > 1 ? n - 1 : 1)) * kBlockThreads);
    uint32_t boxw = 0;
    // the empty child i of a node with valid mask `valid`, widened over the empty siblings that lie ahead of the ray (enter_node's
    // rule for the box-less instances), as a box word: extent code 1 -- one node size -- on the side the ray leaves through
    auto widen_word = [&](unsigned valid, int i) -> uint32_t {
        const unsigned sgn = ((unsigned)r.flags >> kFlagStepShift) & 7u, ahead = ((unsigned)i ^ sgn) & 7u;
        auto span = [&](unsigned e) -> unsigned { return ((unsigned)(0xFF5533110F050301ULL >> (8u * e)) & 0xffu) << ((unsigned)i & ~e); };
        auto pair = [&](unsigned e) -> unsigned { return (1u << i) | (1u << ((unsigned)i ^ e)); };
        unsigned ext = 0;
        if ((span(ahead) & valid) == 0) ext = ahead;
        else if ((ahead & 2u) && (pair(2u) & valid) == 0) ext = 2u;
        else if ((ahead & 1u) && (pair(1u) & valid) == 0) ext = 1u;
        else if ((ahead & 4u) && (pair(4u) & valid) == 0) ext = 4u;
        uint32_t w = 0;
#pragma unroll
        for (unsigned a = 0; a < 3; a++) w |= ((ext >> a) & 1u) << (5u * a + (((sgn >> a) & 1u) ? 15u : 0u));
        return w;
    };
    auto locate = [&](int x, int y, int z) -> int {
        const unsigned diff = (unsigned)((x ^ pvx) | (y ^ pvy) | (z ^ pvz));
        uint32_t own = 0;                                 // kBox: index of the descriptor `cur` was made from (top >= lc), or the cell's box word (top < lc)
        if (kCoarse && (top < lc || (diff >> csh) != 0)) {
            // the cursor of the new cell comes from the table; the canonical traversal would have popped to level a (the deepest
            // node that holds both voxels, or where the cursor already sits) and made t - a descents from there
            int a = n - (32 - __clz((int)diff));          // (diff == 0: n)
            a = a < top ? a : top;
            // (the cell index in 32 bits: the table's level is at most 10 -- vrc_api.cpp -- so it has at most 30)
            const uint64_t cell = coarse_index((unsigned)(x >> csh), (unsigned)(y >> csh), (unsigned)(z >> csh), lc);   // (vrc_params.h: 32-bit arithmetic in the default layout)
            const uint64_t e = p.coarse[cell];
            if (kBox) own = p.box_aux[cell];
            cur = e & ((1ULL << kCoarseLevelShift) - 1ULL);
            top = (int)(e >> kCoarseLevelShift);
            c_desc += (unsigned)(top - a);
            if (top == lc) {                              // slot 0 = level lc: pops inside the cell end here
                lds_stack[tid] = cur;
                if (kBox) lds_own[tid] = own;
            }
        } else {
            if (top > 0 && (diff >> (n - top)) != 0) top = n - (31 - __clz((int)diff)) - 1;   // deepest level whose node holds both voxels
            // (the entry comes from the stack also when nothing is popped: one ds_read_b64 per event instead of two registers
            // carried through the round loop -- every level from sbase down to `top` was stored on the way down)
            cur = (!kCoarse && top == 0) ? root_entry : lds_stack[(top - sbase) * kBlockThreads + tid];
            if (kBox) own = lds_own[(top - sbase) * kBlockThreads + tid];   // (top >= lc here: a cursor above the table's level always takes the table)
        }
        pvx = x; pvy = y; pvz = z;
        for (;;) {
            const int b = n - top - 1;
            const int i = ((x >> b) & 1) | (((y >> b) & 1) << 1) | (((z >> b) & 1) << 2);
            const unsigned masks = (unsigned)cur & 0xffffu;
            const unsigned bit = 1u << i;
            if (!(masks & bit)) {
                // (round 5: the box word of the voxel's slot one level down loaded speculatively BESIDE every descriptor, so that it is
                // never a dependent load of its own: headline 1.510 vs 1.502 ms, 4 lights 3.69 vs 3.63 -- the event chain's latency
                // is covered by the other waves, the extra load instruction is not)
                // (the box word: the table's own for a cell that resolves above its level; the descriptor's record; or, below the levels
                // that have records -- trees too large for a word per descriptor -- the node widened over its empty siblings, as a word)
                if (kBox) boxw = top < lc ? own : (top < p.box_levels ? p.boxes[(size_t)own * 8u + (unsigned)i] : widen_word(masks & 0xffu, i));
                return b;
            }
            if (((masks >> 8) & bit) || b == 0) return -1;
            const unsigned rank = (unsigned)__popc(masks & 0xffu & ((bit << 1) - 1u)) - 1u;
            const uint64_t child = (cur >> 16) + (uint64_t)rank;
            const uint64_t d = descriptors[child];
            // the child's box record: the descriptor index itself, or (upper levels only) the parent's first-child record + the rank
            // -- a load beside the descriptor's, not behind it
            if (kBox) own = p.box_child ? (top + 1 < p.box_levels ? p.box_child[own] + rank : 0u) : (uint32_t)child;
            c_desc++;
            cur = make_entry(descriptors, child, d);
            lds_stack[(top + 1 - sbase) * kBlockThreads + tid] = cur;   // level top+1 (>= lc + 1 with the table)
            if (kBox) lds_own[(top + 1 - sbase) * kBlockThreads + tid] = own;
            top++;
        }
    };
    // material of the solid voxel the cursor just found (attachments: see include/vrc.h); 5 without them
    auto solid_material = [&](int x, int y, int z) -> int {
        if (!p.attach_lookup || top != n - 1) return 5;   // only bottom-level descriptors carry materials
        uint64_t node = p.root_index;
        if (top > 0) {
            const uint64_t parent = (!kCoarse && top == 1) ? root_entry : lds_stack[(top - 1 - sbase) * kBlockThreads + tid];
            const int slot = ((x >> 1) & 1) | (((y >> 1) & 1) << 1) | (((z >> 1) & 1) << 2);
            node = (parent >> 16) + (uint64_t)(__popc((unsigned)parent & 0xffu & ((2u << slot) - 1u)) - 1);
        }
        const uint64_t a = p.attachments[p.attach_lookup[node]];
        return (int)(int8_t)(a >> (8 * ((x & 1) | ((y & 1) << 1) | ((z & 1) << 2))));
    };
#ifndef VRC_TUNED_WIDEN
#define VRC_TUNED_WIDEN true
#endif
    const bool widen = kTuned ? VRC_TUNED_WIDEN : p.widen_nodes != 0;
    // voxel_step in the round loop: three bits of r.flags (kFlagStepShift); the voxel the cursor located last (pvx, pvy, pvz) is the
    // ray's voxel from one event to the next -- Ray::sx.. / Ray::vx.. are set for the cold code where it is entered
    auto step_pos = [&](int axis) -> bool { return (r.flags >> (kFlagStepShift + axis)) & 1; };
    auto steps_to_flags = [&]() {
        r.flags = (r.flags & ~kFlagStepMask) | ((r.sx > 0 ? 1 : 0) << kFlagStepShift) | ((r.sy > 0 ? 1 : 0) << (kFlagStepShift + 1)) | ((r.sz > 0 ? 1 : 0) << (kFlagStepShift + 2));
    };
    auto steps_from_flags = [&]() { r.sx = step_pos(0) ? 1 : -1; r.sy = step_pos(1) ? 1 : -1; r.sz = step_pos(2) ? 1 : -1; };
    // park the ray in the empty node of size 2^b around its voxel.  The parent's valid mask is at hand, so the
    // box is widened over empty siblings that lie ahead of the ray: fewer node events, same lookups (a sibling
    // the mask calls empty would have been found empty without any descriptor read).
    auto enter_node = [&](int b) {
        const int size = 1 << b;
        if (kBox) {
            // the node at (v & ~(size - 1)) extended by the box word's extents on the three sides the ray can leave through, clamped
            // to the map (beyond it everything is empty, but the :563 bounds test must see the crossing)
            auto side = [&](bool pos, int v, int axis, int dim, int &base, float &count) {
                const unsigned c = (boxw >> (unsigned)(5 * axis + (pos ? 15 : 0))) & 31u;
                const int ext = (c < 4u ? (int)c : (int)((4u | (c & 3u)) << ((c >> 2) - 1u))) << b;
                const int o = v & ~(size - 1);
                if (pos) { const int f = o + size + ext; base = f < dim ? f : dim; count = (float)(base - v); }
                else { const int f = o - ext; base = (f > 0 ? f : 0) - 1; count = (float)(v - base); }
            };
            side(step_pos(0), pvx, 0, p.map_dim[0], bx, nx);
            side(step_pos(1), pvy, 1, p.map_dim[1], by, ny);
            side(step_pos(2), pvz, 2, p.map_dim[2], bz, nz);
            return;
        }
        const unsigned valid = (unsigned)cur & 0xffu;
        const int i = ((pvx >> b) & 1) | (((pvy >> b) & 1) << 1) | (((pvz >> b) & 1) << 2);
        // axis a can be widened when the ray moves from this half of the parent toward the other half: child bit a differs
        // from the sign bit of the step (voxel_step is +1 or -1, never 0)
        const unsigned sgn = ((unsigned)r.flags >> kFlagStepShift) & 7u;
        const unsigned ahead = ((unsigned)i ^ sgn) & 7u;
        // children covered when widening over the axes in e = those that differ from i only in axes of e: the subsets of
        // e as bit positions (one byte per e in the constant), shifted to i with the axes of e cleared
        auto span = [&](unsigned e) -> unsigned {
            return ((unsigned)(0xFF5533110F050301ULL >> (8u * e)) & 0xffu) << ((unsigned)i & ~e);
        };
        auto pair = [&](unsigned e) -> unsigned { return (1u << i) | (1u << ((unsigned)i ^ e)); };   // span of one axis
        unsigned ext = 0;
        if (widen) {
            if ((span(ahead) & valid) == 0) ext = ahead;
            else if ((ahead & 2u) && (pair(2u) & valid) == 0) ext = 2u;
            else if ((ahead & 1u) && (pair(1u) & valid) == 0) ext = 1u;
            else if ((ahead & 4u) && (pair(4u) & valid) == 0) ext = 4u;
        }
        const int sx2 = (ext & 1u) ? 2 * size : size, sy2 = (ext & 2u) ? 2 * size : size, sz2 = (ext & 4u) ? 2 * size : size;
        const int cx = pvx & ~(sx2 - 1), cy = pvy & ~(sy2 - 1), cz = pvz & ~(sz2 - 1);
        bx = step_pos(0) ? cx + sx2 : cx - 1;
        by = step_pos(1) ? cy + sy2 : cy - 1;
        bz = step_pos(2) ? cz + sz2 : cz - 1;
        // countdown = |base - voxel| (the step is +-1: no multiply needed)
        nx = (float)(step_pos(0) ? bx - pvx : pvx - bx); ny = (float)(step_pos(1) ? by - pvy : pvy - by);
        nz = (float)(step_pos(2) ? bz - pvz : pvz - bz);
    };
    // extent unknown (after a redirect, or inside solid): force an event after one step from voxel (x, y, z)
    auto enter_single = [&](int x, int y, int z) {
        bx = x + (step_pos(0) ? 1 : -1); by = y + (step_pos(1) ? 1 : -1); bz = z + (step_pos(2) ? 1 : -1);
        nx = ny = nz = 1.0f;
    };

    // exact_jump.hpp: this lane's table of Euclid runs (one per binade and axis pair), interleaved over the wave's lanes,
    // and the bit mask of the rows built for the ray's current direction
    JumpWord *jtab = nullptr;
    uint32_t jrows = 0;
    // rows in LDS: behind the traversal stack, [ring row][pair][thread], one 8-byte word each (ds_read_b64 / ds_write_b64, consecutive threads)
    constexpr int kRing = kLdsTab ? kLdsRows : 4;        // table rows per ray (exact_jump.hpp)
    const int jstride = kLdsTab ? kBlockThreads : 64;
    if (kLdsTab) jtab = reinterpret_cast<JumpWord *>(lds_own + (kBox ? (size_t)(n - lc) * kBlockThreads : 0)) + tid;   // (8-byte aligned: whole multiples of 1 KB before it)
    else if (kJump && s_jump_slot >= 0) jtab = reinterpret_cast<JumpWord *>(p.jump_cache) + ((size_t)s_jump_slot * kTilesPerBlock + (tid >> 6)) * (size_t)(3 * kRing * 64) + (tid & 63);

    if (in_image) {
        if (!ray_setup(r, p, pix)) {
            r.flags |= kFlagUnwritten;
        } else {
            r.flags |= kFlagPrimary;
            const uint64_t d = descriptors[p.root_index];
            c_desc = 1;
            root_entry = make_entry(descriptors, p.root_index, d);
            cur = root_entry;
            int b = -1;
            steps_to_flags();
            if (r.vx >= 0 && r.vy >= 0 && r.vz >= 0 && r.vx < p.map_dim[0] && r.vy < p.map_dim[1] && r.vz < p.map_dim[2])
                b = locate(r.vx, r.vy, r.vz);             // the reference's per-pixel get_oct_vox (:342)
            if (b >= 0) enter_node(b); else enter_single(r.vx, r.vy, r.vz);
            mode = (r.distance_traveled < r.max_distance) ? kStep : kDone;    // :357 guard
        }
    }

    const bool use_arith = kTuned ? true : p.arith_mask != 0;
    const bool use_safe = kTuned ? true : p.safe_run != 0;
#ifndef VRC_TUNED_SINGLE
#define VRC_TUNED_SINGLE true
#endif
    const bool use_single = kTuned ? VRC_TUNED_SINGLE : p.single_step != 0;
    const int shade_threshold = kTuned ? kDefaultShadeThreshold : p.shade_threshold;
    const float jump_min_run = kTuned ? (float)(kLdsTab ? kDefaultJumpMinRunLds : kDefaultJumpMinRun) : (float)p.jump_min_run;   // estimated iterations that make a jump worth its block
    const int safe_cap = kTuned ? (kJump ? kDefaultSafeStepsJump : kDefaultSafeSteps) : p.safe_steps;   // iterations per safe run (phase 2a)
    const int burst_cap = kTuned ? kDefaultBurstSteps : p.burst_steps;   // ordinary steps per round and lane (compare/select loop)
    const int exact_cap = (use_arith && use_safe) ? (kTuned ? kDefaultExactSteps : p.exact_steps) : burst_cap;
    const float safe_limit = safe_t_limit(safe_cap);
#ifdef VRC_SCHED_STATS
    // profiling build only (libvrc_stats.so): per-wave scheduler statistics; each event is counted by the
    // first active lane, so the sum over lanes is the wave-level count
    unsigned w_iters = 0, w_bursts = 0, w_ev_passes = 0, w_ev_lanes = 0, w_sh_passes = 0, w_sh_lanes = 0, w_jumps = 0;
    const int lane_id = tid & 63;
    unsigned l_try = 0, l_ok = 0, l_cov = 0;            // lane-level: jump attempts, successes, iterations covered
    int run_start = 0, run_est_bucket = 0;              // distance_traveled when the node was entered, bucket of its estimate
    auto note_entry = [&]() {
        run_start = r.distance_traveled;
        const float T = fminf(fminf(fmaf(nx - 1.0f, r.dtx, r.itx), fmaf(ny - 1.0f, r.dty, r.ity)), fmaf(nz - 1.0f, r.dtz, r.itz));
        const float est = fmaxf(0.0f, (T - r.itx) * fabsf(r.rdx) + 1.0f) + fmaxf(0.0f, (T - r.ity) * fabsf(r.rdy) + 1.0f) +
                          fmaxf(0.0f, (T - r.itz) * fabsf(r.rdz) + 1.0f);
        run_est_bucket = est >= 1.0f ? 31 - __clz((int)fminf(est, 1e9f)) : 0;
    };
    if (mode == kStep) note_entry();
#define VRC_STAT(var, inc) do { if (lane_id == __ffsll((long long)__ballot(true)) - 1) var += (inc); } while (0)
#else
#define VRC_STAT(var, inc) do { } while (0)
#endif
    int rounds_left = p.watchdog_rounds;
#ifdef VRC_TIME_STATS
    unsigned long long t_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_last;
#endif
    for (;;) {
        // One round = every live lane advances to its next node event: a closed-form jump (long empty
        // stretch) or a burst of ordinary steps (short stretch / not yet in the closed-form regime), then all
        // parked lanes are serviced together, so each phase runs with as many lanes as possible.

        // ---- phase 1: exact closed-form jumps (exact_jump.hpp) for the lanes whose node promises a long run.  The
        // estimate is the one a safe run would make (time of the node exit, here also cut at the next binade end of each
        // axis and at a slow axis that has not had its first crossing: stretch_jump() stops there), turned into iterations.
        if (kJump) {
            bool want = false;
            if (mode == kStep) {
                float X = fminf(fminf(fmaf(nx - 1.0f, r.dtx, r.itx), fmaf(ny - 1.0f, r.dty, r.ity)), fmaf(nz - 1.0f, r.dtz, r.itz));
                X = fminf(X, fminf(fminf(jump_axis_limit(r.itx, r.dtx), jump_axis_limit(r.ity, r.dty)), jump_axis_limit(r.itz, r.dtz)));
                const float est = fmaxf(0.0f, fmaf(X - r.itx, fabsf(r.rdx), 1.0f)) + fmaxf(0.0f, fmaf(X - r.ity, fabsf(r.rdy), 1.0f)) +
                                  fmaxf(0.0f, fmaf(X - r.itz, fabsf(r.rdz), 1.0f));
                // (a block that found no table slot has no table: it never jumps.  Tested on the pointer, which is live anyway:
                // one more value held through the loop costs this kernel 2 % in spills)
                want = (kLdsTab || jtab != nullptr) && est >= jump_min_run;
                // ... and only once every t has reached the table's first binade (t >= 128): below it a pair has no table row and
                // is solved on the spot by every jump that meets it -- with the threshold at 64 that was 5.2 M on-the-spot Euclid
                // runs per headline frame (11 M lanes waiting for them), 2.28 -> 2.21 ms without them.  A ray spends ~200
                // iterations below t = 128, the binades there are short, and the step loop takes them
                want = want && fminf(fminf(r.itx, r.ity), r.itz) >= (float)(1 << VRC_JUMP_FIRST_LOG2);
                // (round 4: a second, lower threshold for lanes whose three t are all inside the table's binades -- "always jump once
                // t >= 128" -- 1 / 8 / 24 iterations: 2.57 / 2.63 / 2.56 ms against 2.50: the pass, ~500 instructions on its straight
                // path, is what a short run cannot pay for, not the Euclid runs)
            }
            const unsigned long long wj = __ballot(want);
            VRC_TICK(0);
#ifdef VRC_SCHED_STATS
            if (lane_id == 0) atomicAdd(&g_jump_stats[7], 1ULL);
            if (want) atomicAdd(&g_jump_stats[6], 1ULL);
#endif
            // (one lane is enough: holding the block back until 4 / 8 / 16 / 32 lanes want it -- the lanes that do standing
            // still meanwhile, or stepping on -- measured 2.73 / 2.82 / 2.85 / 3.10 ms against 2.59 on the headline frame)
            if (wj != 0ULL) {
#ifdef VRC_SCHED_STATS
                if (lane_id == __ffsll((long long)wj) - 1) atomicAdd(&g_jump_stats[0], 1ULL);
#endif
                {   // the table rows the jumps will read: built for every lane whose ray keeps its direction
                    uint32_t solves = 0;
                    asm volatile("; VRC_MARK jump_rows_begin");
#ifndef VRC_JUMP_NO_FILL   // (timing experiment only, with VRC_JUMP_NO_TIES)
                    jump_rows_build<kRing>(want, mode == kStep || mode == kEvent, jrows, r.itx, r.ity, r.itz, r.dtx, r.dty, r.dtz, jtab, jstride, solves);
#endif
                    asm volatile("; VRC_MARK jump_rows_end");
                    VRC_TICK(1);
#ifdef VRC_SCHED_STATS
                    if (solves) atomicAdd(&g_jump_stats[5], (unsigned long long)solves);
#endif
                }
                if (want) {
                    asm volatile("; VRC_MARK jump_block_begin");
                    int inx = (int)nx, iny = (int)ny, inz = (int)nz;
                    const JumpOut jo = stretch_jump<kRing>(r.itx, r.ity, r.itz, r.dtx, r.dty, r.dtz, inx, iny, inz,
                                                    r.max_distance - r.distance_traveled, jtab, jstride, jrows);
#ifdef VRC_SCHED_STATS
                    atomicAdd(&g_jump_stats[1], 1ULL);
                    atomicAdd(&g_jump_stats[2], (unsigned long long)jo.iterations);
                    if (jo.left_node) atomicAdd(&g_jump_stats[3], 1ULL);
                    if (jo.capped) atomicAdd(&g_jump_stats[4], 1ULL);
#endif
                    if (jo.capped) {                          // :357 the step cap ends the loop inside the stretch
                        r.distance_traveled = r.max_distance;
                        mode = ended();
                    } else {
                        nx = (float)inx; ny = (float)iny; nz = (float)inz;
                        if (jo.left_node) {
                            fxf = jo.fx; fyf = jo.fy; fzf = jo.fz;
                            r.distance_traveled += jo.iterations - 1;     // the leaving iteration's :714 follows the lookup
                            mode = kEvent;
                        } else {                              // stopped at a binade end / a frozen axis: the next round goes on
                            r.distance_traveled += jo.iterations;
                            if (r.distance_traveled >= r.max_distance) mode = ended();
                        }
                    }
                    asm volatile("; VRC_MARK jump_block_end");
                }
            }
        }

        VRC_TICK(2);
        asm volatile("; VRC_MARK safe_begin");
        // the arithmetic face_mask of the step loop needs every t to be 0 or >= 2^-100.  t only grows by
        // delta_t >= 1/2 per step, so once a lane is safe it stays safe until its ray is restarted (t_unsafe is set
        // again there); a wave with an unsafe stepping lane takes the compare/select loop for this burst.
        if (t_unsafe) t_unsafe = !(t_is_safe(r.itx) && t_is_safe(r.ity) && t_is_safe(r.itz));
        const bool arith_mask = use_arith && __ballot(mode == kStep && t_unsafe) == 0ULL;

        // ---- phase 2a: safe run (safe_run.hpp): lanes deep inside an empty node step without countdowns while
        // min(t) is below their threshold T.  All lanes run every trip (a lane whose gate is closed takes empty
        // steps): no exec masking, scalar loop control, kSafeUnroll iterations per loop trip (the vote and the scalar
        // branch are not free: 2 -> 4 -> 8 -> 16 -> 32 iterations per trip measured 3.30 -> 3.11 -> 3.09 -> 3.03 -> 3.14 ms).
        bool deep = false;                                // still far from the node face after the safe run
        constexpr int kSafeUnroll = kJump ? kSafeUnrollJump : kSafeUnrollPlain;
        if (arith_mask && use_safe && safe_cap >= kSafeUnroll) {
            SafeGate gate;
            if (kJump) {                                  // (branch-free form, see make_gate)
                const bool steps = (mode == kStep) & (r.max_distance - r.distance_traveled >= safe_cap);
                const float T = fminf(fminf(safe_threshold(r.itx, r.dtx, nx), safe_threshold(r.ity, r.dty, ny)),
                                      safe_threshold(r.itz, r.dtz, nz));
                gate = make_gate<true>(steps ? T : -1.0f, fminf(fminf(r.itx, r.ity), r.itz), safe_limit, fminf(fminf(r.dtx, r.dty), r.dtz));
            } else if (mode == kStep && r.max_distance - r.distance_traveled >= safe_cap) {
                const float T = fminf(fminf(safe_threshold(r.itx, r.dtx, nx), safe_threshold(r.ity, r.dty, ny)),
                                      safe_threshold(r.itz, r.dtz, nz));
                gate = make_gate(T, fminf(fminf(r.itx, r.ity), r.itz), safe_limit, fminf(fminf(r.dtx, r.dty), r.dtz));
            }
            if (__ballot(gate.open) != 0ULL) {
                const float x0 = r.itx, y0 = r.ity, z0 = r.itz;
                float cnt = 0.0f, alive = 0.0f;
#ifdef VRC_SCHED_STATS
                // what a prefix without the per-iteration gate could cover: an open lane stays below its threshold for
                // at least max over axes of floor((T - t) / delta_t) iterations; the wave could run the minimum of
                // that over its open lanes (whole loop trips of it) with a constant 0/1 mask per lane
                float kl = 3.0e38f;
                if (gate.open) {
                    const float T = fminf(fminf(safe_threshold(r.itx, r.dtx, nx), safe_threshold(r.ity, r.dty, ny)),
                                          safe_threshold(r.itz, r.dtz, nz));
                    kl = fmaxf(fmaxf(floorf((T - r.itx) * r.rdx * (r.rdx < 0 ? -1.0f : 1.0f)), floorf((T - r.ity) * fabsf(r.rdy))),
                               floorf((T - r.itz) * fabsf(r.rdz)));
                    kl = fmaxf(kl - 1.0f, 0.0f);
                }
                for (int o = 32; o > 0; o >>= 1) kl = fminf(kl, __shfl_xor(kl, o));
                const int prefix_trips = (int)fminf(kl, 1.0e6f) / kSafeUnroll;
                int trips_run = 0;
#endif
#pragma nounroll
                for (int trip = safe_cap / kSafeUnroll; trip > 0; trip--) {
#pragma unroll
                    for (int u = 0; u < kSafeUnroll; u++) {
                        const float m = fminf(fminf(r.itx, r.ity), r.itz);
                        alive = fma_sat(m, gate.neg_b1, gate.tb1);
                        cnt += alive;
                        const float gx = alive_if_zero(r.itx - m, alive);    // :558
                        const float gy = alive_if_zero(r.ity - m, alive);
                        const float gz = alive_if_zero(r.itz - m, alive);
                        r.itx = __builtin_fmaf(r.dtx, gx, r.itx);            // :559
                        r.ity = __builtin_fmaf(r.dty, gy, r.ity);
                        r.itz = __builtin_fmaf(r.dtz, gz, r.itz);
                    }
                    VRC_STAT(w_iters, kSafeUnroll);
#ifdef VRC_SCHED_STATS
                    if (lane_id == 0) l_try += kSafeUnroll;   // safe-run wave-iterations (reported as jump_attempts)
                    trips_run++;
                    l_ok += kSafeUnroll * (unsigned)alive;    // ~ safe-run lane-iterations (jump_successes)
#endif
                    if (__ballot(alive != 0.0f) == 0ULL) break;
                }
#ifdef VRC_SCHED_STATS
                if (lane_id == 0) {
                    atomicAdd(&g_jump_stats[8], (unsigned long long)((prefix_trips < trips_run ? prefix_trips : trips_run) * kSafeUnroll));
                    atomicAdd(&g_jump_stats[9], (unsigned long long)(trips_run * kSafeUnroll));
                }
#endif
                if (gate.open) {
                    nx -= safe_steps_taken(r.itx, x0, r.rdx);                // :560 as countdowns
                    ny -= safe_steps_taken(r.ity, y0, r.rdy);
                    nz -= safe_steps_taken(r.itz, z0, r.rdz);
                    r.distance_traveled += (int)cnt;                         // :714
                    if (r.distance_traveled >= r.max_distance) mode = ended();   // :357
                    deep = alive != 0.0f;
                }
            }
        }

        VRC_TICK(3);
        asm volatile("; VRC_MARK single_begin");
        // ---- phase 2b: one exact step for the lanes that are not deep inside a node.  A lane that has just reached
        // its safe-run threshold is almost always exactly one iteration from the node face (the threshold is within
        // n 2^-23 relative of the crossing time), and so is a lane in a freshly entered one-voxel node: one
        // straight-line iteration with countdowns sends them to the event phase, and the exact loop below only runs
        // for what is left (lanes near their step cap, t outside the arithmetic range, rare two-step leftovers).
        if (arith_mask && use_single && mode == kStep && !deep) {
            const float m = fminf(fminf(r.itx, r.ity), r.itz);
            const float gx = alive_if_zero(r.itx - m, 1.0f);              // :558
            const float gy = alive_if_zero(r.ity - m, 1.0f);
            const float gz = alive_if_zero(r.itz - m, 1.0f);
            r.itx = __builtin_fmaf(r.dtx, gx, r.itx);                     // :559
            r.ity = __builtin_fmaf(r.dty, gy, r.ity);
            r.itz = __builtin_fmaf(r.dtz, gz, r.itz);
            nx -= gx; ny -= gy; nz -= gz;                                 // :560 as countdowns
            VRC_STAT(w_iters, 1);
            if ((nx * ny) * nz == 0.0f) {                                 // left the node: bounds test + lookup pending
                fxf = gx; fyf = gy; fzf = gz;
                mode = kEvent;                                            // the leaving iteration's :714 follows the lookup
            } else {
                r.distance_traveled++;                                    // :714
                if (r.distance_traveled >= r.max_distance) mode = ended();   // :357
            }
        }

        asm volatile("; VRC_MARK exact_begin");
        // ---- phase 2: ordinary steps (:357-560) for lanes still inside their node.  A lane that is waiting
        // to jump only takes two steps (enough to settle its progressions); the others run to their node face.
        // (skipped when every stepping lane of the wave is still deep inside its node: the next safe run takes them on)
        if (mode == kStep && __ballot(mode == kStep && !deep) != 0ULL) {
            const int true_limit = r.max_distance - r.distance_traveled;     // >= 1 iterations left (:357)
            const int it_limit = true_limit < exact_cap ? true_limit : exact_cap;
            float left = (float)it_limit;                  // per-lane countdown (exact: < 2^24)
            bool go;
            if (arith_mask) {
                // face_mask by arithmetic instead of v_cmp + v_cndmask (v_sub/v_mul/v_fma issue at ~1.6x the rate
                // of compares, selects and min/max on gfx950; tools/ubench): d = t - min >= 0 is 0 exactly when
                // t == min (IEEE subtraction with denormals never rounds a non-zero difference to 0), and
                // clamp(1 - d * 2^127) is 1 for d == 0 and 0 for d >= 2^-127, which covers every non-zero d because
                // all t are 0 or >= 2^-100 here (t_safe).  Two iterations per exit test: "alive" (1.0 / 0.0: no
                // countdown reached 0 in the first iteration) replaces the constant 1.0 in the mask and the step
                // count of the second one, so a lane that has left its node takes an empty second step.  The
                // countdown test is a product (integers < 2^24: never rounds to zero) instead of min3 + min.
                float gx, gy, gz, alive;
                do {
                    float m = fminf(fminf(r.itx, r.ity), r.itz);
                    fxf = alive_if_zero(r.itx - m, 1.0f);
                    fyf = alive_if_zero(r.ity - m, 1.0f);
                    fzf = alive_if_zero(r.itz - m, 1.0f);
                    r.itx = __builtin_fmaf(r.dtx, fxf, r.itx);
                    r.ity = __builtin_fmaf(r.dty, fyf, r.ity);
                    r.itz = __builtin_fmaf(r.dtz, fzf, r.itz);
                    nx -= fxf; ny -= fyf; nz -= fzf;
                    left -= 1.0f;
                    alive = mul_sat(nx * ny, nz * left);
                    m = fminf(fminf(r.itx, r.ity), r.itz);
                    gx = alive_if_zero(r.itx - m, alive);
                    gy = alive_if_zero(r.ity - m, alive);
                    gz = alive_if_zero(r.itz - m, alive);
                    r.itx = __builtin_fmaf(r.dtx, gx, r.itx);
                    r.ity = __builtin_fmaf(r.dty, gy, r.ity);
                    r.itz = __builtin_fmaf(r.dtz, gz, r.itz);
                    nx -= gx; ny -= gy; nz -= gz;
                    left -= alive;
                    go = (nx * ny) * (nz * left) != 0.0f;
                    VRC_STAT(w_iters, 2);
                } while (go);
                if (alive != 0.0f) { fxf = gx; fyf = gy; fzf = gz; }   // the mask of the last real iteration
            } else {
                do {
                    const float m = fminf(fminf(r.itx, r.ity), r.itz);
                    fxf = r.itx <= m ? 1.0f : 0.0f;       // :558 (ties step several axes)
                    fyf = r.ity <= m ? 1.0f : 0.0f;
                    fzf = r.itz <= m ? 1.0f : 0.0f;
                    r.itx = __builtin_fmaf(r.dtx, fxf, r.itx);   // :559, exact: dt * {0,1} has no rounding
                    r.ity = __builtin_fmaf(r.dty, fyf, r.ity);
                    r.itz = __builtin_fmaf(r.dtz, fzf, r.itz);
                    nx -= fxf; ny -= fyf; nz -= fzf;      // :560 as countdowns to the node face
                    left -= 1.0f;
                    go = fminf(fminf(fminf(nx, ny), nz), left) != 0.0f;
                    VRC_STAT(w_iters, 1);
                } while (go);
            }
            const int it = it_limit - (int)left;          // iterations this lane executed in the burst
            VRC_STAT(w_bursts, 1);
            if (fminf(fminf(nx, ny), nz) == 0.0f) {       // left the node: bounds test + lookup pending
                mode = kEvent;
                r.distance_traveled += it - 1;            // the leaving iteration's :714 comes after the lookup
            } else {
                r.distance_traveled += it;                // :714
                if (it == true_limit) mode = ended();     // :357
            }
        }
        VRC_TICK(4);
        asm volatile("; VRC_MARK event_begin");
        const unsigned long long ev = __ballot(mode == kEvent);
        const unsigned long long st = __ballot(mode == kStep);
        unsigned long long sh = __ballot(mode == kShade || (kMulti && mode == kRelight));
        // (watchdog: every round advances at least one lane by a step, an event or a hit block, so a wave needs far
        // fewer rounds than this; stopping a wave that got here keeps the GPU alive and is reported by vrc_get_counters)
        if ((ev | st | sh) == 0ULL || --rounds_left < 0) break;

        // ---- phase 3: node events
        if (ev != 0ULL) {
            VRC_STAT(w_ev_passes, 1); VRC_STAT(w_ev_lanes, __popcll(ev));
            if (mode == kEvent) {
#ifdef VRC_SCHED_STATS
                {
                    const int L = r.distance_traveled + 1 - run_start;
                    const int b = L >= 1 ? 31 - __clz(L) : 0;
                    atomicAdd(&g_run_hist[0][b], 1ULL);
                    atomicAdd(&g_run_hist[1][b], (unsigned long long)(L > 0 ? L : 0));
                    atomicAdd(&g_run_hist[2][run_est_bucket], (unsigned long long)(L > 0 ? L : 0));
                }
#endif
                const int vx = step_pos(0) ? bx - (int)nx : bx + (int)nx;   // voxel = base - step * countdown, step = +-1
                const int vy = step_pos(1) ? by - (int)ny : by + (int)ny;
                const int vz = step_pos(2) ? bz - (int)nz : bz + (int)nz;
                // (the event phase touches nothing of the ray's COLD state -- colours, face_mask as the hit block wants it: what a segment
                // that ends here owes the colours is flagged and paid by settle_segment(), the hit block gets its face_mask from the
                // last step's mask when it runs)
                if (vx >= p.map_dim[0] || vy >= p.map_dim[1] || vz >= p.map_dim[2] || vx < 0 || vy < 0 || vz < 0) {
                    r.flags |= kFlagOob | kFlagOobPending | kFlagBroke;   // :563-568 (oob_exit)
                    mode = ended();
                } else {
                    const int b = locate(vx, vy, vz);
#ifdef VRC_SCHED_STATS
                    if (b >= 0) atomicAdd(&g_jump_stats[(kCoarse && top < lc) ? 10 : 11], 1ULL);   // empty nodes found above the table's level / below it
#endif
                    if (b >= 0) {
                        enter_node(b);
                        r.distance_traveled++;            // :714
                        mode = (r.distance_traveled < r.max_distance) ? kStep : ended();   // :357
#ifdef VRC_SCHED_STATS
                        note_entry();
#endif
                    } else {
                        mat = solid_material(vx, vy, vz);
                        if ((mat == 5 || mat == 6) && r.shadow_ray) {   // :575, :707-710 (shadow_hit)
                            r.flags |= kFlagShadowHit | kFlagShadowPending | kFlagBroke;
                            mode = ended();
                        } else if (mat == 5 || mat == 6) {
                            mode = kShade;                // the hit block is deferred
                        } else {                          // any other material is passed through
                            enter_single(vx, vy, vz);
                            r.distance_traveled++;        // :714
                            mode = (r.distance_traveled < r.max_distance) ? kStep : ended();   // :357
#ifdef VRC_SCHED_STATS
                            note_entry();
#endif
                        }
                    }
                }
            }
            sh = __ballot(mode == kShade || (kMulti && mode == kRelight));
        }

        VRC_TICK(5);
        asm volatile("; VRC_MARK relight_begin");
        // ---- phase 4a (multi-light): a lane whose shadow ray has ended goes back to the first strike for the next light;
        // cheap next to the hit block, so it need not wait for the whole tile (VRC_RELIGHT_THRESHOLD lanes, or nothing left
        // to step)
        if (kMulti) {
            const unsigned long long rl = __ballot(mode == kRelight);
            if (rl != 0ULL && ((int)__popcll(rl) >= VRC_RELIGHT_THRESHOLD || __ballot(mode == kStep) == 0ULL)) {
            if (mode == kRelight) {
                settle_segment(r);                        // what the ended shadow segment owes the colours: the next light's input
                r.light_index++;
                note_lights();
                if (!light_from_strike(r, p, r.light_index, true)) {
                    mode = kDone;                         // :671-672, pixel left unwritten
                } else {
                    restart_from(r, strike_pos(r));
                    steps_to_flags();
                    t_unsafe = true;
                    steps_base += r.distance_traveled + ((r.flags >> kFlagBrokeShift) & 1) - (r.kdist + 1);
                    r.flags &= ~kFlagBroke;
                    enter_single(r.vx, r.vy, r.vz);
                    jrows = 0;                            // delta_t changed: the table of exact_jump.hpp is stale
                    r.distance_traveled = r.kdist + 1;    // as if the strike iteration had just finished (:714)
                    mode = (r.distance_traveled < r.max_distance) ? kStep : ended();
#ifdef VRC_SCHED_STATS
                    note_entry();
#endif
                }
            }
            }
            sh = __ballot(mode == kShade);
        } else {
            sh = __ballot(mode == kShade);
        }
        asm volatile("; VRC_MARK shade_begin");
        // ---- phase 4b: hit block (:575-711): expensive and needed once per pixel (a shadow ray's hit is handled where it
        // lands), so it runs only when the whole tile waits for it or nothing cheaper is left to do
        if (sh != 0ULL && (__ballot(mode == kStep) == 0ULL || (int)__popcll(sh) >= shade_threshold)) {
            VRC_STAT(w_sh_passes, 1); VRC_STAT(w_sh_lanes, __popcll(sh));
            if (mode == kShade) {
                // what the hit block reads of the ray's hot state, handed over here: face_mask of the step that found the voxel (a parked
                // lane has not stepped since), the voxel itself (the one the cursor located last), voxel_step
                r.fmx = (int)fxf; r.fmy = (int)fyf; r.fmz = (int)fzf;
                r.vx = pvx; r.vy = pvy; r.vz = pvz;
                steps_from_flags();
                note_lights();                            // (light 0 is cast by the hit block: light_index is 0 here)
                if (hit_block<kMulti>(r, mat, p)) {
                    r.flags |= kFlagBroke;
                    mode = ended();
                } else {
                    steps_to_flags();
                    enter_single(r.vx, r.vy, r.vz);
                    t_unsafe = true;
                    jrows = 0;                            // delta_t changed with the redirect: the table of exact_jump.hpp is stale
                    r.distance_traveled++;                // :714
                    mode = (r.distance_traveled < r.max_distance && (r.counts >> 16) < 2) ? kStep : ended();   // :357
#ifdef VRC_SCHED_STATS
                    note_entry();
#endif
                }
            }
        }
        VRC_TICK(6);
        asm volatile("; VRC_MARK round_end");
    }

    const int tid_end = cold_thread_index(wave_in_block);  // threadIdx.x, without a register through the round loop
    if (rounds_left < 0 && (tid_end & 63) == 0) {
        atomicAdd(&block_ctr[kCtrWatchdog], 1ULL);
        if (p.watchdog_flag) __hip_atomic_store(p.watchdog_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    unsigned c_steps = 0, c_tex = 0, c_shadow = 0, c_primary = 0, c_unwritten = 0;
    if (in_image) {
        c_primary = (r.flags & kFlagPrimary) ? 1u : 0u;
        c_unwritten = (r.flags & kFlagUnwritten) ? 1u : 0u;
        if (c_primary) {
            c_steps = (unsigned)(steps_base + r.distance_traveled) + (unsigned)((r.flags >> kFlagBrokeShift) & 1);
            c_tex = r.counts & 0xffu; c_shadow = (r.counts >> 8) & 0xffu;
            if (!r.written) c_unwritten = 1;
        }
        settle_segment(r);
        ray_finish(r, p, c_desc);
    }
#ifdef VRC_SCHED_STATS
    if (w_iters) atomicAdd(&block_ctr[kCtrWaveIters], (unsigned long long)w_iters);
    if (w_bursts) atomicAdd(&block_ctr[kCtrBursts], (unsigned long long)w_bursts);
    if (w_ev_passes) atomicAdd(&block_ctr[kCtrEventPasses], (unsigned long long)w_ev_passes);
    if (w_ev_lanes) atomicAdd(&block_ctr[kCtrEventLanes], (unsigned long long)w_ev_lanes);
    if (w_sh_passes) atomicAdd(&block_ctr[kCtrShadePasses], (unsigned long long)w_sh_passes);
    if (w_sh_lanes) atomicAdd(&block_ctr[kCtrShadeLanes], (unsigned long long)w_sh_lanes);
    if (l_try) atomicAdd(&block_ctr[kCtrShadeLanes + 1], (unsigned long long)l_try);
    if (l_ok) atomicAdd(&block_ctr[kCtrShadeLanes + 2], (unsigned long long)l_ok);
    if (l_cov) atomicAdd(&block_ctr[kCtrMap], (unsigned long long)l_cov);
#endif
#ifdef VRC_TIME_STATS
    {
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();
        if ((tid_end & 63) == 0) {
            t_acc[8] = now_ - t_begin;
            for (int k = 0; k < 9; k++) atomicAdd(&g_time_stats[k], t_acc[k]);
        }
    }
#endif
    const unsigned vals[7] = {c_primary, c_shadow, c_desc, c_tex, 0u, c_steps, c_unwritten};
    publish_counters(p, block_ctr, vals, tid_end);        // (a __syncthreads inside: every wave of the block is through with its tables)
    if (kJump && !kLdsTab && tid_end == 0 && s_jump_slot >= 0) atomicExch(&p.jump_slots[s_jump_slot], 0u);
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
    // setting octree_bias = 0 (extension) drops the term: the reference's bias shears the picture whenever the camera
    // sits in an empty node whose corner is not the camera voxel
    for (int a = 0; a < 3; a++) p.frame[a] = p.octree_bias ? (corner[a] - pos[a]) * res / 2 : 0;
    p.frame[3] = reads;
}

// the image the reference starts from: RGBA8 (255,255,255,100) (CLCaster.cpp:280-286) as normalised floats
__global__ void fill_image_kernel(float4 *image, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) image[i] = make_float4(1.0f, 1.0f, 1.0f, 100.0f / 255.0f);
}

// write_imagef to the reference's CL_UNORM_INT8 target (CLCaster.cpp:278-296): saturate, scale, round to nearest even
__global__ void pack_rgba8_kernel(const float4 *__restrict__ image, uchar4 *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = image[i];
    auto q = [](float f) -> unsigned char {
        if (!(f > 0.0f)) f = 0.0f;                        // NaN and negatives saturate to 0
        if (f > 1.0f) f = 1.0f;
        return (unsigned char)__float2int_rn(f * 255.0f);
    };
    out[i] = make_uchar4(q(v.x), q(v.y), q(v.z), q(v.w));
}

#ifdef VRC_TIME_STATS
}  // namespace vrc
extern "C" int vrc_stats_time(unsigned long long *out16, int clear) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(vrc::g_time_stats), sizeof(vrc::g_time_stats)) != hipSuccess) return 1;
    if (clear) {
        static unsigned long long zero[16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(vrc::g_time_stats), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
namespace vrc {
#endif
#ifdef VRC_SCHED_STATS
}  // namespace vrc
// profiling build only (tools/run_hist.py): read / clear the run-length histogram
extern "C" int vrc_stats_run_hist(unsigned long long *out96, int clear) {   // out: 96 histogram + 12 counters + 4 tie-path counters (112 values)
    if (out96 && hipMemcpyFromSymbol(out96, HIP_SYMBOL(vrc::g_run_hist), sizeof(vrc::g_run_hist)) != hipSuccess) return 1;
    if (out96 && hipMemcpyFromSymbol(out96 + 96, HIP_SYMBOL(vrc::g_jump_stats), sizeof(vrc::g_jump_stats)) != hipSuccess) return 1;
    if (out96 && hipMemcpyFromSymbol(out96 + 108, HIP_SYMBOL(vrc::g_jump_private_solves), sizeof(vrc::g_jump_private_solves)) != hipSuccess) return 1;
    if (clear) {
        static unsigned long long zero[3][32];
        if (hipMemcpyToSymbol(HIP_SYMBOL(vrc::g_run_hist), zero, sizeof(zero)) != hipSuccess) return 1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(vrc::g_jump_stats), zero, sizeof(vrc::g_jump_stats)) != hipSuccess) return 1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(vrc::g_jump_private_solves), zero, sizeof(vrc::g_jump_private_solves)) != hipSuccess) return 1;
    }
    return 0;
}
namespace vrc {
#endif

hipError_t launch_fill_image(float *image, size_t n_pixels, hipStream_t stream) {
    (void)hipGetLastError();                 // an error an earlier call left behind is not this launch's
    if (!n_pixels) return hipSuccess;
    hipLaunchKernelGGL(fill_image_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<float4 *>(image), n_pixels);
    return hipGetLastError();
}

hipError_t launch_pack_rgba8(const float *image, uint8_t *out, size_t n_pixels, hipStream_t stream) {
    (void)hipGetLastError();                 // an error an earlier call left behind is not this launch's
    if (!n_pixels) return hipSuccess;
    hipLaunchKernelGGL(pack_rgba8_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4 *>(image), reinterpret_cast<uchar4 *>(out), n_pixels);
    return hipGetLastError();
}

hipError_t launch_frame_setup(const RaycastParams &p, hipStream_t stream) {
    (void)hipGetLastError();                 // an error an earlier call left behind is not this launch's
    hipLaunchKernelGGL(frame_setup_kernel, dim3(1), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_raycast_jump(const RaycastParams &p, hipStream_t stream);   // raycast_jump_kernel.hip

// dynamic LDS of the SVO kernel: the traversal stack, and behind it the jump tables when they live in LDS
static bool svo_uses_coarse(const RaycastParams &p) {
    return p.coarse != nullptr && p.coarse_log2 >= 1 && p.coarse_log2 <= p.log2_dim - 2;
}
static bool svo_uses_boxes(const RaycastParams &p) { return svo_uses_coarse(p) && p.boxes != nullptr && p.box_aux != nullptr; }
static size_t svo_stack_bytes(const RaycastParams &p) {
    const int levels = svo_uses_coarse(p) ? p.log2_dim - p.coarse_log2 : (p.log2_dim > 1 ? p.log2_dim - 1 : 1);
    // (with the boxes: a dword per level and thread for the descriptor index, behind the 8-byte entries)
    return (size_t)levels * kBlockThreads * (sizeof(uint64_t) + (svo_uses_boxes(p) ? sizeof(uint32_t) : 0)) + (size_t)p.lds_pad_bytes;
}
constexpr size_t lds_table_bytes(int rows) { return (size_t)(3 * rows) * kBlockThreads * sizeof(JumpWord); }   // ring rows x 3 pairs, one 8-byte word per thread

// How many rows of the jumps' Euclid tables live in LDS for this frame: 3 when the jump instance with stack + tables still reaches
// the blocks per CU its registers allow (VRC_MIN_BLOCKS_JUMP) -- asked of the runtime once per LDS size -- else 2 (the box instances
// have a two-row twin for their deeper stacks), else 0: the tables live in global memory.  Setting jump_tables_lds = 0 / 1
// overrides (never / three rows whatever the occupancy; 2 = this rule).  vrc_api.cpp asks to know whether the global table buffer is needed.
int jump_tables_lds_rows(const RaycastParams &p) {
    if (!p.svo || p.stepping_mode != 0 || !svo_uses_coarse(p)) return 0;
    if (p.jump_tables_lds == 0) return 0;
    if (p.jump_tables_lds == 1) return 3;
    static std::mutex guard;                             // (handles of several host threads may ask at the same time)
    std::lock_guard<std::mutex> lock(guard);
    // asked of the instance that will run (single- / multi-light, with / without the boxes) on the current device, once per
    // (device, instance, LDS size)
    const bool box = svo_uses_boxes(p), multi = p.light_count > 1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t stack = svo_stack_bytes(p);
    const unsigned long long key = ((unsigned long long)stack << 16) | ((unsigned long long)(dev & 0xff) << 8) | (box ? 2u : 0u) | (multi ? 4u : 0u);
    static unsigned long long cached_key[8] = {~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL};
    static int cached[8];
    const int slot = (box ? 2 : 0) | (multi ? 4 : 0);
    if (cached_key[slot] != key) {
        auto fits = [&](const void *fn, int rows) {
            int per_cu = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlockThreads, stack + lds_table_bytes(rows));
            (void)hipGetLastError();
            return e == hipSuccess && per_cu >= (multi ? VRC_MIN_BLOCKS_JUMP_MULTI : VRC_MIN_BLOCKS_JUMP);
        };
        const void *f3 = box ? (multi ? reinterpret_cast<const void *>(raycast_svo_kernel<true, true, true, 3, true, true>)
                                      : reinterpret_cast<const void *>(raycast_svo_kernel<true, false, true, 3, true, true>))
                             : (multi ? reinterpret_cast<const void *>(raycast_svo_kernel<true, true, true, 3, true>)
                                      : reinterpret_cast<const void *>(raycast_svo_kernel<true, false, true, 3, true>));
        const void *f2 = multi ? reinterpret_cast<const void *>(raycast_svo_kernel<true, true, true, 2, true, true>)
                               : reinterpret_cast<const void *>(raycast_svo_kernel<true, false, true, 2, true, true>);
        cached[slot] = fits(f3, 3) ? 3 : (box && fits(f2, 2)) ? 2 : 0;
        cached_key[slot] = key;
    }
    return cached[slot];
}

hipError_t launch_raycast(const RaycastParams &p, hipStream_t stream) {
    (void)hipGetLastError();                 // an error an earlier call left behind is not this launch's
    const int nblocks = p.blocks_x * p.local_tile_rows;
    if (nblocks <= 0) return hipSuccess;
    if (p.svo && p.stepping_mode == 1) return launch_raycast_jump(p, stream);
    if (p.svo) {
        const bool jump = p.jump_min_run < kJumpOff, multi = p.light_count > 1;
        const int lds_rows = jump ? p.jump_tables_lds : 0;       // (vrc_api.cpp has resolved the setting to 0 / 2 / 3 rows with jump_tables_lds_rows)
        if (lds_rows != 0 && lds_rows != 3 && !(lds_rows == 2 && svo_uses_boxes(p))) return hipErrorInvalidValue;
        const bool lds_tab = lds_rows > 0;
        const size_t lds = svo_stack_bytes(p) + lds_table_bytes(lds_rows);
        const bool tuned = (!jump || p.jump_min_run == (lds_tab ? kDefaultJumpMinRunLds : kDefaultJumpMinRun)) &&
                           p.widen_nodes != 0 && p.arith_mask != 0 && p.safe_run != 0 && p.single_step != 0 &&
                           p.shade_threshold == kDefaultShadeThreshold && p.safe_steps == (jump ? kDefaultSafeStepsJump : kDefaultSafeSteps) &&
                           p.exact_steps == kDefaultExactSteps && p.burst_steps == kDefaultBurstSteps;
        if (jump && !lds_tab && (!p.jump_cache || !p.jump_slots || p.jump_slot_count < 1)) return hipErrorInvalidValue;
        // 24 instances: the knobs at their defaults (kTuned) x {no jumps | Euclid tables in global memory | in LDS} x {no table | coarse
        // table | + empty boxes} x {one light | multi-light}, jumps only with the table, the box instances also with a two-row ring in
        // LDS (deep trees); and the same with run-time knobs ONCE each,
        // with the multi-light code compiled in (it renders a single light too -- the relight block never runs): those exist for the
        // tests and tools that move the knobs, and a frame rendered through them is the same frame
#define VRC_LAUNCH(...) hipLaunchKernelGGL((raycast_svo_kernel<__VA_ARGS__>), dim3(nblocks), dim3(kBlockThreads), lds, stream, p)
#define VRC_LAUNCH_MT(J, L, ...) do { if (!tuned) VRC_LAUNCH(J, true, false, L, __VA_ARGS__); else if (multi) VRC_LAUNCH(J, true, true, L, __VA_ARGS__); \
                                    else VRC_LAUNCH(J, false, true, L, __VA_ARGS__); } while (0)
        if (jump && !svo_uses_coarse(p)) return hipErrorInvalidValue;     // (vrc_api.cpp switches the jumps off where there is no table)
        if (svo_uses_boxes(p)) {
            if (lds_rows == 3) VRC_LAUNCH_MT(true, 3, true, true);
            else if (lds_rows == 2) VRC_LAUNCH_MT(true, 2, true, true);
            else if (jump) VRC_LAUNCH_MT(true, 0, true, true);
            else VRC_LAUNCH_MT(false, 0, true, true);
        } else if (svo_uses_coarse(p)) {
            if (lds_tab) VRC_LAUNCH_MT(true, 3, true);
            else if (jump) VRC_LAUNCH_MT(true, 0, true);
            else VRC_LAUNCH_MT(false, 0, true);
        } else {
            VRC_LAUNCH_MT(false, 0, false);
        }
#undef VRC_LAUNCH_MT
#undef VRC_LAUNCH
    } else {
        hipLaunchKernelGGL(raycast_array_kernel, dim3(nblocks), dim3(kBlockThreads), 0, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_reduce_counters(const unsigned long long *partials, int nblocks, unsigned long long *out,
                                  hipStream_t stream) {
    (void)hipGetLastError();                 // an error an earlier call left behind is not this launch's
    hipLaunchKernelGGL(reduce_counters_kernel, dim3(1), dim3(256), 0, stream, partials, nblocks, out);
    return hipGetLastError();
}

}  // namespace vrc
