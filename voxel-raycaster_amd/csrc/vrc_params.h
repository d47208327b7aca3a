// Kernel-argument block shared by the host layer (vrc_api.cpp) and the gfx950
// raycast kernels (raycast_kernel.hip).  One struct passed by value: it lands
// in the kernarg segment / SGPRs, replacing the 16 cl_mem arguments the
// reference binds by name (src/CLCaster.cpp:186-202).
#pragma once

#include <stdint.h>

namespace vrc {

#ifndef VRC_TILE_W
#define VRC_TILE_W 8
#endif
constexpr int kTileW = VRC_TILE_W;    // one wavefront = one kTileW x kTileH pixel tile (8x8)
constexpr int kTileH = 64 / VRC_TILE_W;
static_assert((kTileW & (kTileW - 1)) == 0 && kTileW >= 1 && kTileW <= 64, "tile width must be a power of two <= 64");
#ifndef VRC_TILES_PER_BLOCK
#define VRC_TILES_PER_BLOCK 4
#endif
constexpr int kTilesPerBlock = VRC_TILES_PER_BLOCK;   // 256-thread block = 4 horizontally adjacent tiles (32x8 px)
constexpr int kBlockThreads = 64 * kTilesPerBlock;
// defaults of the scheduling knobs (settings of the same names; the tuned kernel instances have them compiled in)
// iterations per safe run (setting safe_steps): short rounds since round 4 -- with the tree's top in the coarse table a node
// event is cheap, and what a long safe run costs is the lanes that idle through it (38 % lane utilisation at 64).  Headline
// frame, safe_steps x jump threshold: 64 x 64 2.21 ms, 32 x 32 2.13, 16 x 16 2.11, 8 x 8 2.25; without jumps (depth 11) 16 / 32 / 64:
// 1.62 / 1.53 / 1.55 ms
#ifndef VRC_DEFAULT_SAFE_STEPS
#define VRC_DEFAULT_SAFE_STEPS 32
#endif
#ifndef VRC_DEFAULT_SAFE_STEPS_JUMP
#define VRC_DEFAULT_SAFE_STEPS_JUMP 16
#endif
#ifndef VRC_DEFAULT_EXACT_STEPS
#define VRC_DEFAULT_EXACT_STEPS 16
#endif
constexpr int kDefaultBurstSteps = 48, kDefaultShadeThreshold = 64, kDefaultSafeSteps = VRC_DEFAULT_SAFE_STEPS, kDefaultExactSteps = VRC_DEFAULT_EXACT_STEPS;
constexpr int kDefaultSafeStepsJump = VRC_DEFAULT_SAFE_STEPS_JUMP;      // ... of the instances with the closed-form jumps
// exact closed-form jumps (exact_jump.hpp): estimated iterations from which a lane asks for the jump block, and the tree
// depth from which they are on by default (measured: depth 10 loses 15 %, depth 12 gains 15 %, depth 16 is 3.3x faster)
#ifndef VRC_DEFAULT_JUMP_MIN_RUN
#define VRC_DEFAULT_JUMP_MIN_RUN 96
#endif
constexpr int kDefaultJumpMinRun = VRC_DEFAULT_JUMP_MIN_RUN, kDefaultJumpMinDepth = 12;
// ... and with the empty boxes (round 5) the runs are longer: depth 11 then gains 13 % from the jumps (1.229 -> 1.069 ms, primary only
// 0.847 -> 0.730), depth 10 and 9 still lose (0.562 -> 0.619, 0.546 -> 0.660)
constexpr int kDefaultJumpMinDepthBoxes = 11;
// ... and with the tables in LDS a jump is cheaper: 48 / 64 / 96 / 128 measured 2.56 / 2.50 / 2.52 / 2.53 ms on the headline frame
// before the coarse table, 16 the best with it and 16-iteration safe runs (see kDefaultSafeStepsJump below)
#ifndef VRC_DEFAULT_JUMP_MIN_RUN_LDS
#define VRC_DEFAULT_JUMP_MIN_RUN_LDS 16
#endif
constexpr int kDefaultJumpMinRunLds = VRC_DEFAULT_JUMP_MIN_RUN_LDS;
constexpr int kJumpOff = 1 << 24;      // jump_min_run >= this: the instances without the jump block
constexpr int kJumpTableDwordsPerLane = 24;   // 4 ring rows x 3 pairs x one 8-byte word (exact_jump.hpp JumpWord; checked in raycast_kernel.hip)
constexpr int kJumpSlotsPerXcd = 256;         // table slots per XCD: 32 CUs x 8 blocks, the most 256-thread blocks an XCD can hold at any occupancy
constexpr int kJumpSlots = 8 * kJumpSlotsPerXcd;   // a block takes a slot of ITS XCD while it runs (the L2s of two XCDs are not coherent)
constexpr int kMaxLights = 8;         // light slots (include/LightController.h:95)
constexpr int kMaxLevels = 24;        // descriptor levels the LDS stack can hold (dim <= 2^24)
// mode B's coarse table: by default 512^3 cells at most (1 GB; the headline frame: 0.695 ms without a table, levels 5 .. 10:
// 0.756 / 0.727 / 0.688 / 0.635 / 0.574 / 0.550 ms -- level 10 is an 8 GB table for a 161 MB tree and stays a setting), and
// never finer than cells of 4 voxels (the bottom two levels are always descriptors)
constexpr int kCoarseMaxLog2 = 9;
constexpr int kCoarseLevelShift = 59; // entry bits 59-63: level; child indices must stay below 2^43
// index of cell (cx, cy, cz) of the coarse table (and of the boxes' parallel word) with 2^lc cells per axis: x fastest; with
// -DVRC_COARSE_BRICK=k bricks of (2^k)^3 cells are contiguous (an experiment of round 4: the index arithmetic costs more than the
// locality gives, so the default is no bricks).  ONE definition for the builders and every kernel that reads the table.
#ifndef VRC_COARSE_BRICK
#define VRC_COARSE_BRICK 0
#endif
constexpr int kCoarseBrickLog2 = VRC_COARSE_BRICK;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t coarse_index(unsigned cx, unsigned cy, unsigned cz, int lc) {
    constexpr unsigned k = kCoarseBrickLog2, m = (1u << k) - 1u;
    if (k == 0 || lc < (int)k) return cx | (cy << lc) | (cz << (2 * lc));   // (32 bits: lc <= 10, vrc_api.cpp)
    const int lb = lc - (int)k;                            // bricks per axis = 2^lb
    const uint64_t brick = (uint64_t)(cx >> k) | ((uint64_t)(cy >> k) << lb) | ((uint64_t)(cz >> k) << (2 * lb));
    return (brick << (3 * k)) | (cx & m) | ((cy & m) << k) | ((cz & m) << (2 * k));
}
// the table level for a tree of depth n (0: no table).  From depth 14 on one level finer (1024^3 cells, 8 GB -- vrc_api.cpp only
// takes it while the table is at most 16 x the descriptor array): the traversal stack then has 4 (depth 14) to 6 (depth 16) levels,
// and stack + boxes' index array + Euclid tables of the exact jumps (exact_jump.hpp) still fit the 128 bytes of LDS a lane has at
// 5 blocks per CU -- with the tables in global memory instead a depth-16 frame takes 6.3 instead of 4.7 ms, and the depth-14 frame
// WITH boxes 3.16 instead of 2.40
constexpr int coarse_level_for_depth(int n) { return n >= 14 ? kCoarseMaxLog2 + 1 : n >= 5 ? (n - 2 < kCoarseMaxLog2 ? n - 2 : kCoarseMaxLog2) : 0; }

// hit-record flag bits (include/vrc.h VRC_HIT_FLAG_*)
constexpr int kFlagWritten = 1, kFlagShadowCast = 2, kFlagShadowHit = 4, kFlagOob = 8;
constexpr int kFlagHasHit = 0x800;     // internal: the primary hit has been recorded (bits above 7 never leave the kernels)
// internal, SVO kernel: what the counters need at the end of the ray, kept in the flags word instead of a register each --
// the pixel cast a primary ray, the pixel stays unwritten (:293-294), the last segment ended by a break (one more iteration)
constexpr int kFlagPrimary = 0x1000, kFlagUnwritten = 0x2000, kFlagBroke = 0x4000, kFlagBrokeShift = 14;
// ... and what the end of a ray segment still owes the colour state (settle_segment in raycast_common.hpp): the ray left the map
// (:563-568) / a shadow ray struck a voxel (:707-710).  The colours are cold state -- they live in scratch through the round loop --
// and an assignment to them inside the event phase made the register allocator reload and re-spill them in EVERY event pass
constexpr int kFlagOobPending = 0x8000, kFlagShadowPending = 0x10000;
// ... and voxel_step as three bits (set: the ray steps +1 on that axis): the round loop of the SVO kernel reads the steps from here,
// the cold code (hit block, relight) from Ray::sx/sy/sz, which it sets itself -- three registers less through the loop
constexpr int kFlagStepShift = 17, kFlagStepMask = 7 << kFlagStepShift;
// ... and (multi-light) "another light waits after this shadow segment": what more_lights() computes from light_index, kept as a bit
// so that the step phases' `ended()` read no cold register
constexpr int kFlagMoreLights = 0x100000;

// counters[] slots (device, uint64)
enum CounterSlot {
    kCtrPrimary = 0, kCtrShadow, kCtrDesc, kCtrTex, kCtrMap, kCtrSteps, kCtrUnwritten, kCtrWatchdog,
    // wave-scheduler statistics of the SVO kernel (one count per wave, not per lane)
    kCtrWaveIters = 8, kCtrBursts, kCtrEventPasses, kCtrEventLanes, kCtrShadePasses, kCtrShadeLanes, kCtrCount = 16
};

struct RaycastParams {
    // arg 0-1: dense map
    const int8_t *map;
    int32_t map_dim[3];
    // arg 2-3: viewport
    int32_t width, height;
    const float *viewport;            // float4[w*h]
    // arg 8: offline pixel buffer + hit records
    float *image;                     // float4[w*h]
    int32_t *hits;                    // int32[8*w*h]; nullptr = no hit records (setting hit_records = 0)
    // arg 9-11: atlas
    const uint8_t *atlas;             // RGBA8
    int32_t atlas_w, atlas_h;
    int32_t tiles_x, tiles_y;         // atlas_dim / tile_dim (integer division)
    // arg 12 + settings: SVO
    const uint64_t *descriptors;
    const uint32_t *attach_lookup;    // arg 13 (optional): per-descriptor slot into attachments
    const uint64_t *attachments;      // arg 14 (optional): 8 int8 materials per bottom-level descriptor
    uint64_t root_index;
    int32_t log2_dim;                 // OCTDIM = 1 << log2_dim
    int32_t svo;                      // using_octree == 0
    // arg 4-5: camera (trig evaluated once on the host: SURVEY D2)
    float cam_pos[3];
    float trig[4];                    // sin(dir.x) cos(dir.x) sin(dir.y) cos(dir.y)
    // arg 6/7: lights[l] = rgbi[4], position[3], pad.  light_count = lights the kernel shades with: 1 (light 0
    // only) is the reference; more is the multi-light extension (setting "light_count")
    float lights[kMaxLights][8];
    int32_t light_count;
    int32_t octree_bias;              // 1: the reference's (sub_oct_pos - voxel) * resolution / 2 term (:353-354); 0: none
    int32_t arith_mask;
    int32_t watchdog_rounds;          // rounds a wave may take before the watchdog stops it
    int32_t safe_run, exact_steps, safe_steps, single_step;
    // frame constants written by frame_setup_kernel: {bias[3], reads} -- the
    // pixel-independent get_oct_vox(camera voxel) of ray_caster_kernel.cl:342-354
    int32_t *frame;
    int32_t max_distance;
    int32_t shadow_rays;
    int32_t burst_steps;              // SVO kernel: ordinary DDA steps per lane and round before node events are serviced
    int32_t shade_threshold;          // ... and before the hit block runs
    int32_t widen_nodes;              // widen an empty node over empty siblings ahead of the ray (results unchanged)
    int32_t jump_min_run;             // exact closed-form jumps for stretches of at least this many (estimated) iterations (1<<24 = off)
    uint32_t *jump_cache;             // the per-ray Euclid tables of exact_jump.hpp: kJumpTableDwordsPerLane dwords per lane of a block slot
    uint32_t *jump_slots;             // one flag per block slot (0 free / 1 taken): a block takes a slot when it starts
    int32_t jump_slot_count;
    int32_t jump_tables_lds;          // 2: the tables live in LDS when stack + tables fit at full occupancy (depth <= 12), 1 / 0: always / never
    int32_t lds_pad_bytes;            // experiment knob: extra dynamic LDS to lower occupancy
    int32_t xcd_mode;                 // block->tile map: 0 contiguous eighth per XCD, 1 tile rows interleaved over XCDs, 2 none
    // row tiling (multi-GPU)
    int32_t tile_rank, tile_world, band_tiles;   // band_tiles = band_rows / kTileH
    int32_t blocks_x;                 // ceil(width / 32)
    int32_t local_tile_rows;          // tile rows this rank renders
    // 1: viewport / image / hits hold only this rank's rows (vrc_set_row_slice): row index = position among the
    // rank's rows; 0: full-frame buffers indexed by the image row
    int32_t row_sliced;
    int32_t stepping_mode;            // 0: exact per-voxel DDA (reference parity); 1: node-exit jumps (SURVEY D1 mode B)
    // mode B only: the top of the tree as a dense grid of 2^coarse_log2 cells per axis (x fastest).  An entry is the state the
    // canonical descent toward the cell has when it reaches level coarse_log2 or meets an empty / leaf child before:
    // bits 0-15 leaf << 8 | valid, 16-58 absolute index of the first kept child, 59-63 the level of that node.  nullptr: no table
    const uint64_t *coarse;
    int32_t coarse_log2;
    // empty boxes (empty_boxes.hip; needs the coarse table): boxes[8 * descriptor + child slot] = six 5-bit extents by which an
    // empty child node may be widened, box_aux[cell] = the table's parallel word (the box of the empty node a cell resolves to
    // above the table's level, or the index of the level-coarse_log2 descriptor).  nullptr: nodes + sibling widening only
    const uint32_t *boxes;
    const uint32_t *box_aux;
    // boxes for the upper levels only (trees too large for a word per descriptor and child): box records exist for the descriptors of
    // the levels below box_levels, numbered breadth-first; box_child[record] = the record of the descriptor's first child.  nullptr:
    // a record per descriptor, record index = descriptor index (then box_levels is the tree's depth)
    const uint32_t *box_child;
    int32_t box_levels;
    unsigned long long *counters;
    // host-mapped flag the round watchdog raises (checked by vrc_sync: a truncated frame never looks like success)
    unsigned int *watchdog_flag;
};

}  // namespace vrc
