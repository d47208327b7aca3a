// vrc_api.cpp -- C-ABI host layer of libvrc.so (see include/vrc.h).
//
// Plays the role of src/CLCaster.cpp in the reference: owns the device
// buffers (its named buffer_map, :855-944), the settings buffer (:1029-1109),
// validation (:157-206) and the per-frame launch (:224-228, 946-987) -- with
// HIP streams/events on MI355X instead of an OpenCL queue + GL interop.
//
// A handle drives one GPU.  A GROUP handle (vrc_create_group) is rank 0 of a set of handles, one per GPU, that share
// one host thread: every assign_* / setting call is replicated, the octree is uploaded once and fanned out
// device-to-device, each rank owns only its row bands of the ray table / frame (vrc_set_row_slice), vrc_compute
// launches all ranks and returns when all are done, and the read-back calls gather every rank's rows into the
// caller's frame.  No collective anywhere.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vrc.h"
#include "vrc_params.h"

namespace vrc {
hipError_t launch_raycast(const RaycastParams &p, hipStream_t stream);
int jump_tables_lds_rows(const RaycastParams &p);
hipError_t launch_coarse_build(const uint64_t *descriptors, uint64_t root_index, int log2_dim, int lc, uint64_t *out, hipStream_t stream);
hipError_t launch_box_positions(const uint64_t *descriptors, uint64_t n_desc, uint64_t root_index, int n, uint64_t *pos, hipStream_t stream);
hipError_t box_queries_cut(unsigned long long *out);
struct BoxUpper { uint64_t *desc = nullptr, *pos = nullptr; uint32_t *child = nullptr, *boxes = nullptr; uint64_t count = 0; int levels = 0; };
hipError_t launch_box_build_upper(const uint64_t *descriptors, uint64_t root_index, int n, int lc, uint64_t max_records, int max_levels,
                                  BoxUpper *out, uint32_t *aux, hipStream_t stream);
hipError_t launch_box_build(const uint64_t *descriptors, uint64_t n_desc, uint64_t root_index, int n, int lc, uint64_t *pos_tmp,
                            uint32_t *boxes, uint32_t *aux, hipStream_t stream);
hipError_t launch_box_check_cells(const uint64_t *descriptors, uint64_t root_index, int n, int lc, const uint32_t *aux, uint64_t samples,
                                  uint64_t seed, unsigned long long *result, hipStream_t stream);
hipError_t launch_box_check(const uint64_t *descriptors, uint64_t n_records, uint64_t root_index, int n, const uint64_t *pos, const uint64_t *desc_of,
                            const uint32_t *boxes, uint64_t samples, uint64_t seed, unsigned long long *result, hipStream_t stream);
hipError_t launch_frame_setup(const RaycastParams &p, hipStream_t stream);
hipError_t launch_reduce_counters(const unsigned long long *partials, int nblocks, unsigned long long *out,
                                  hipStream_t stream);
hipError_t launch_fill_image(float *image, size_t n_pixels, hipStream_t stream);
hipError_t launch_pack_rgba8(const float *image, uint8_t *out, size_t n_pixels, hipStream_t stream);
int build_shell_terrain_device(hipStream_t stream, uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor,
                               uint32_t flags, uint64_t validate_samples, const int32_t *probe_xy, uint32_t n_probe,
                               int32_t *probe_lohi, uint64_t **d_desc, vrc_build_info *out, std::string &error);
int build_columns_device(hipStream_t stream, uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor,
                         const uint16_t *host_hi, const uint16_t *host_lo,
                         uint32_t flags, uint64_t validate_samples, const int32_t *probe_xy, uint32_t n_probe,
                         int32_t *probe_lohi, uint64_t **d_desc, vrc_build_info *out, std::string &error);
int build_grid_device(hipStream_t stream, uint32_t depth, const int8_t *host_grid, const int8_t *resident_grid, uint32_t flags,
                      uint64_t validate_samples, uint64_t **d_desc, uint32_t **d_lookup, uint64_t **d_attach, uint64_t *n_attach,
                      vrc_build_info *out, std::string &error);
}  // namespace vrc

struct vrc_setting { std::string name, define; int64_t value; };

// One descriptor array resident on one GPU, with its materials and with what the kernels derive from it (the dense table of
// the tree's top, the empty boxes).  Everything here is a function of the TREE, not of a caster: the handles that render the
// same array on the same GPU -- the ranks of a group that sit on rank 0's GPU, a handle that adopted another handle's tree
// (vrc_assign_octree_from) -- hold one Tree between them, so the 1 GB table and the boxes exist once.  The reference never
// frees or shares anything (CLCaster.cpp:5-12); this is footprint hygiene for trees of BASELINE configs[4]'s size.
struct vrc_tree {
    int device = 0;
    uint64_t *d_desc = nullptr; uint64_t n_desc = 0;
    uint32_t *d_attach_lookup = nullptr; uint64_t *d_attach = nullptr; uint64_t n_attach = 0;
    // derived, built on first use by whichever handle needs them first (guard: handles of several host threads)
    std::mutex guard;
    uint64_t *d_coarse = nullptr; uint64_t coarse_root = 0; int coarse_depth = 0, coarse_log2 = 0;
    uint32_t *d_boxes = nullptr, *d_box_aux = nullptr; uint64_t box_root = 0; int box_depth = 0, box_log2 = 0; double box_build_seconds = 0.0;
    unsigned long long box_queries_cut = 0;               // region queries of the build that gave up at their budget (boxes smaller than they could be)
    // box records: one per descriptor (box_mode 1: record = descriptor index) or for the upper levels only (box_mode 2: breadth-first
    // records, d_box_child = first-child record, d_box_desc / d_box_pos = descriptor and position of a record, kept for the self-check)
    int box_mode = 0, box_levels = 0, box_fail_mode = 0; int64_t box_levels_asked = 0, box_records_asked = 0; uint64_t box_records = 0;
    uint32_t *d_box_child = nullptr; uint64_t *d_box_desc = nullptr, *d_box_pos = nullptr;
    // an allocation failed: the frames go on without the structure.  Not retried every frame -- but retried as soon as what was
    // asked for changes (level, root, depth) or the host sets coarse_log2 / empty_boxes again (vrc_setting_add / _set)
    bool coarse_gave_up = false, boxes_gave_up = false;
    uint64_t coarse_fail_root = 0, box_fail_root = 0; int coarse_fail_depth = 0, coarse_fail_log2 = 0, box_fail_depth = 0, box_fail_log2 = 0;
    std::string coarse_note, box_note;                    // why (cleared when the structure is built after all)
    ~vrc_tree() {
        // (the caller's current device is left as it was: a group that lets go of its peers' trees goes on allocating on rank 0's GPU)
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        (void)hipSetDevice(device);
        if (d_coarse) (void)hipFree(d_coarse);
        if (d_boxes) (void)hipFree(d_boxes);
        if (d_box_aux) (void)hipFree(d_box_aux);
        if (d_box_child) (void)hipFree(d_box_child);
        if (d_box_desc) (void)hipFree(d_box_desc);
        if (d_box_pos) (void)hipFree(d_box_pos);
        if (d_desc) (void)hipFree(d_desc);
        if (d_attach_lookup) (void)hipFree(d_attach_lookup);
        if (d_attach) (void)hipFree(d_attach);
        if (cur >= 0) (void)hipSetDevice(cur);
        (void)hipGetLastError();
    }
};

struct vrc_caster {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;

    // scene buffers (device)
    int8_t *d_map = nullptr; int32_t map_dim[3] = {0, 0, 0};
    // the tree (shared, see vrc_tree); d_desc / n_desc / d_attach* mirror its fields for the code that reads them
    std::shared_ptr<vrc_tree> tree;
    uint64_t *d_desc = nullptr; uint64_t n_desc = 0; bool have_octree = false;
    bool owns_desc = true;                // false: the tree is another handle's too (a group rank on rank 0's GPU, vrc_assign_octree_from)
    bool own_copy = false;                // group flag VRC_GROUP_OWN_COPIES: never share, always take the device-to-device copy path
    int32_t peer_access = -1;             // -1 same GPU as rank 0 / rank 0 itself, 1 direct peer access enabled, 0 the runtime stages the copies
    void *pinned_stage = nullptr; size_t pinned_stage_bytes = 0;   // read-back staging for a pageable destination (groups)
    bool last_frame_boxes = false;        // the last enqueued frame was rendered with the tree's empty boxes (its descriptor-read counts are the box traversal's)
    bool last_frame_wrote_hits = false;   // d_hits belongs to the last enqueued frame (setting hit_records was on)
    uint32_t *d_attach_lookup = nullptr; uint64_t *d_attach = nullptr; uint64_t n_attach = 0;
    float *d_viewport = nullptr; float *d_image = nullptr; int32_t *d_hits = nullptr; uint8_t *d_rgba8 = nullptr;
    uint32_t *d_jump_cache = nullptr, *d_jump_slots = nullptr; int jump_slot_count = 0;   // Euclid tables of the exact closed-form jumps (exact_jump.hpp), per resident block
    int32_t width = 0, height = 0;
    bool sliced = false;                  // viewport / image / hits hold only this rank's rows
    int32_t buffer_rows = 0;              // rows the three buffers hold
    uint8_t *d_atlas = nullptr; int32_t atlas_w = 0, atlas_h = 0, tile_w = 0, tile_h = 0;
    unsigned long long *d_partials = nullptr; int partial_blocks = 0;
    unsigned long long *d_counters = nullptr;
    int32_t *d_frame = nullptr;           // {bias[3], reads}
    unsigned int *wd_flag = nullptr;      // host-mapped: raised by the kernel's round watchdog

    // live (retained) host pointers
    const float *cam_dir = nullptr, *cam_pos = nullptr;
    const float *cam_trig = nullptr;      // vrc_assign_camera_trig: the host's own sin / cos of the two angles (nullptr: sinf / cosf here)
    const float *lights = nullptr; const int32_t *light_count = nullptr;

    std::vector<vrc_setting> settings;    // <= 64 slots (include/CLCaster.h:303)

    int32_t tile_rank = 0, tile_world = 1, band_rows = 8;
    bool validated = false;
    int last_blocks = 0;
    uint64_t frames_enqueued = 0;
    uint64_t sched_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    // kernel timing
    struct EvPair { hipEvent_t a, b; };
    std::vector<EvPair> pending, pool;
    uint64_t timed_launches = 0; double timed_ms = 0.0;

    // group: this handle is rank 0, peers are ranks 1..n-1
    std::vector<vrc_caster *> peers;
    bool is_peer = false;
};

namespace {

int fail(vrc_caster *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->error = buf;
    return code;
}

#define HIP_TRY(h, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (void)hipGetLastError();      /* reported here: not left behind for a later launch check */ \
            return fail(h, e_ == hipErrorOutOfMemory ? VRC_ERR_OUT_OF_MEMORY : VRC_ERR_DEVICE,    \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                         \
    } while (0)

// replicate a call on every other rank of a group; the first failure is reported through rank 0
#define FOR_PEERS(h, expr)                                                                        \
    do {                                                                                          \
        for (size_t pi_ = 0; pi_ < (h)->peers.size(); pi_++) {                                    \
            vrc_caster *q = (h)->peers[pi_];                                                      \
            const int rc_ = (expr);                                                               \
            if (rc_ != VRC_OK) return fail(h, rc_, "rank %zu: %s", pi_ + 1, q->error.c_str());    \
        }                                                                                         \
    } while (0)

template <class T>
void release(T *&p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

// the tree's empty boxes, in either form
void release_boxes(vrc_tree *t) {
    release(t->d_boxes); release(t->d_box_aux); release(t->d_box_child); release(t->d_box_desc); release(t->d_box_pos);
    t->box_log2 = 0; t->box_mode = 0; t->box_records = 0; t->box_levels = 0;
}

// the handle lets go of its tree; the arrays, the table and the boxes are freed with the last handle that holds them
void release_tree(vrc_caster *h) {
    h->tree.reset();
    h->d_desc = nullptr; h->d_attach_lookup = nullptr; h->d_attach = nullptr;
    h->owns_desc = true;
    h->n_desc = 0; h->n_attach = 0; h->have_octree = false; h->validated = false;
}
// a fresh, unshared tree for this handle (its arrays are filled in by the caller)
vrc_tree *new_tree(vrc_caster *h) {
    h->tree = std::make_shared<vrc_tree>();
    h->tree->device = h->device;
    h->owns_desc = true;
    return h->tree.get();
}
// the handle's mirrors of its tree's fields
void mirror_tree(vrc_caster *h) {
    const vrc_tree *t = h->tree.get();
    h->d_desc = t ? t->d_desc : nullptr; h->n_desc = t ? t->n_desc : 0;
    h->d_attach_lookup = t ? t->d_attach_lookup : nullptr; h->d_attach = t ? t->d_attach : nullptr; h->n_attach = t ? t->n_attach : 0;
}

int prepare_one(vrc_caster *h);           // (below, beside the launch path that shares derive_from_tree with it)

int find_setting(const vrc_caster *h, const char *name) {
    for (size_t i = 0; i < h->settings.size(); i++)
        if (h->settings[i].name == name) return (int)i;
    return -1;
}

int64_t setting_or(const vrc_caster *h, const char *name, int64_t dflt) {
    int i = find_setting(h, name);
    return i < 0 ? dflt : h->settings[i].value;
}

// coarse_log2 / empty_boxes / empty_box_records / empty_box_levels set (again) by the host: a build that failed before -- a transient out-of-memory, a level asked too
// large -- is tried again by the next vrc_prepare / frame
void retry_derived(vrc_caster *h, const char *name) {
    if (!h->tree || (strcmp(name, "coarse_log2") != 0 && strcmp(name, "empty_boxes") != 0 && strcmp(name, "empty_box_records") != 0 &&
                     strcmp(name, "empty_box_levels") != 0)) return;
    std::lock_guard<std::mutex> lock(h->tree->guard);
    h->tree->coarse_gave_up = h->tree->boxes_gave_up = false;
}

int set_setting(vrc_caster *h, const char *name, const char *define, int64_t value) {
    retry_derived(h, name);
    int i = find_setting(h, name);
    if (i >= 0) { h->settings[i].value = value; return VRC_OK; }
    if (h->settings.size() >= 64) return fail(h, VRC_ERR_LIMIT, "settings buffer is full (64 slots)");
    h->settings.push_back({name, define ? define : "", value});
    return VRC_OK;
}

int log2_exact(int64_t v) {
    if (v < 2 || (v & (v - 1))) return -1;
    int n = 0;
    while ((1LL << n) < v) n++;
    return n;
}

void drain_events(vrc_caster *h) {
    for (auto &p : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { h->timed_ms += ms; h->timed_launches++; }
        h->pool.push_back(p);
    }
    h->pending.clear();
}

// The rows a handle renders, as runs: image rows [y0, y0 + n) live at buffer rows [b0, b0 + n).
struct RowRun { int32_t y0, n, b0; };
std::vector<RowRun> row_runs(const vrc_caster *h, bool buffer_sliced) {
    std::vector<RowRun> runs;
    int32_t b0 = 0;
    for (int32_t y = h->tile_rank * h->band_rows; y < h->height; y += h->tile_world * h->band_rows) {
        const int32_t n = std::min(h->band_rows, h->height - y);
        runs.push_back({y, n, buffer_sliced ? b0 : y});
        b0 += n;
    }
    return runs;
}
int32_t local_row_count(const vrc_caster *h) {
    int32_t rows = 0;
    for (const RowRun &r : row_runs(h, true)) rows += r.n;
    return rows;
}

// one row of the reference's ray table (CLCaster.cpp:233-275): base ray (-800, x, y) slewed by the literal 1.57 about
// Y in double, then normalised in float (util.hpp:64-73)
void reference_table_row(int32_t width, int32_t height, int32_t row, float *out) {
    const double s157 = std::sin(1.57), c157 = std::cos(1.57);
    const int y = row - height / 2;
    for (int x = -width / 2; x < width / 2; x++) {
        const float bx = -800.0f, by = (float)x, bz = (float)y;
        const float rx = (float)((double)bz * s157 + (double)bx * c157);
        const float ry = by;
        const float rz = (float)((double)bz * c157 - (double)bx * s157);
        const float len = std::sqrt(rx * rx + ry * ry + rz * rz);
        float *t = out + 4 * (size_t)(x + width / 2);
        t[0] = rx / len; t[1] = ry / len; t[2] = rz / len; t[3] = 0.0f;
    }
}

// allocate the viewport buffers (all rows, or this rank's rows when sliced) and upload the ray table run by run;
// table == nullptr: the reference's own table
int install_viewport(vrc_caster *h, int32_t width, int32_t height, const float *table) {
    HIP_TRY(h, hipSetDevice(h->device));
    release(h->d_viewport); release(h->d_image); release(h->d_hits); release(h->d_rgba8); release(h->d_jump_cache); release(h->d_jump_slots); h->jump_slot_count = 0;
    h->width = width; h->height = height;
    h->buffer_rows = h->sliced ? local_row_count(h) : height;
    h->validated = false;
    const size_t npix = (size_t)width * (size_t)std::max(h->buffer_rows, 1);
    HIP_TRY(h, hipMalloc((void **)&h->d_viewport, 16 * npix));
    HIP_TRY(h, hipMalloc((void **)&h->d_image, 16 * npix));
    std::vector<RowRun> runs = h->sliced ? row_runs(h, true) : std::vector<RowRun>{{0, height, 0}};
    std::vector<float> stage;
    const int32_t chunk = 64;                                     // rows per host staging buffer
    for (const RowRun &r : runs)
        for (int32_t o = 0; o < r.n; o += chunk) {
            const int32_t n = std::min(chunk, r.n - o);
            const float *src = table ? table + 4 * (size_t)width * (size_t)(r.y0 + o) : nullptr;
            if (!table) {
                // the reference's loops run x, y over [-w/2, w/2) x [-h/2, h/2): with an odd size the last column / row
                // of its (zero-initialised) table is never written (CLCaster.cpp:242-275)
                stage.assign(4 * (size_t)width * n, 0.0f);
                for (int32_t k = 0; k < n; k++)
                    if (r.y0 + o + k < 2 * (height / 2))
                        reference_table_row(width, height, r.y0 + o + k, stage.data() + 4 * (size_t)width * k);
                src = stage.data();
            }
            HIP_TRY(h, hipMemcpy(h->d_viewport + 4 * (size_t)width * (size_t)(r.b0 + o), src, 16 * (size_t)width * n, hipMemcpyHostToDevice));
        }
    // the image starts as RGBA8 (255,255,255,100)  (CLCaster.cpp:280-286)
    HIP_TRY(h, vrc::launch_fill_image(h->d_image, npix, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return VRC_OK;
}

// the per-ray Euclid tables of the exact closed-form jumps (exact_jump.hpp): kJumpTableDwordsPerLane dwords per lane of a
// block SLOT.  Only resident blocks hold a slot (the kernel takes and returns them), so the buffer is sized for the chip,
// not for the frame: at most kJumpSlots slots of 12 KB.
int ensure_jump_cache(vrc_caster *h, int nblocks) {
    // an eighth of the slots per XCD, each eighth at least as large as the number of blocks one XCD can hold at a time: its
    // CUs x the 256-thread blocks a CU holds at most (MI355X: 32 x 8 = kJumpSlotsPerXcd; a larger part gets larger eighths, and
    // the kernel's slot search is bounded in any case)
    int cus = 0, threads_per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 256;
    if (hipDeviceGetAttribute(&threads_per_cu, hipDeviceAttributeMaxThreadsPerMultiProcessor, h->device) != hipSuccess) threads_per_cu = 2048;
    (void)hipGetLastError();
    const int per_xcd = std::max(vrc::kJumpSlotsPerXcd, ((cus + 7) / 8) * std::max(1, threads_per_cu / vrc::kBlockThreads));
    const int slots = 8 * std::max(1, std::min(nblocks, per_xcd));
    if (h->d_jump_cache && h->jump_slot_count >= slots) return VRC_OK;
    release(h->d_jump_cache); release(h->d_jump_slots);
    h->jump_slot_count = 0;
    HIP_TRY(h, hipMalloc((void **)&h->d_jump_cache, (size_t)slots * vrc::kBlockThreads * vrc::kJumpTableDwordsPerLane * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc((void **)&h->d_jump_slots, (size_t)slots * sizeof(uint32_t)));
    HIP_TRY(h, hipMemsetAsync(h->d_jump_slots, 0, (size_t)slots * sizeof(uint32_t), h->stream));
    h->jump_slot_count = slots;
    return VRC_OK;
}

int ensure_hits(vrc_caster *h) {
    if (h->d_hits || !h->d_viewport) return VRC_OK;
    const size_t npix = (size_t)h->width * (size_t)std::max(h->buffer_rows, 1);
    HIP_TRY(h, hipMalloc((void **)&h->d_hits, 32 * npix));
    HIP_TRY(h, hipMemsetAsync(h->d_hits, 0, 32 * npix, h->stream));
    return VRC_OK;
}

// Is this host address page-locked (hipHostMalloc / hipHostRegister)?  hipMemcpyAsync into pageable memory blocks the
// calling thread copy by copy, which would serialise the tiles of a multi-GPU read-back.
bool host_byte_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    const hipError_t e = hipPointerGetAttributes(&a, p);
    (void)hipGetLastError();
    return e == hipSuccess && a.type == hipMemoryTypeHost;
}
// the whole destination [p, p + bytes) must be page-locked: a buffer pinned only in part (vrc_pin_host_buffer with a
// smaller size, or a sub-range of a larger array) is treated as pageable and staged.  Registered ranges are contiguous,
// and two adjacent registrations are still page-locked memory, so the first and the last byte decide.
bool host_is_pinned(const void *p, size_t bytes) {
    if (!p || !bytes) return false;
    return host_byte_is_pinned(p) && host_byte_is_pinned((const char *)p + bytes - 1);
}

// device rows -> the caller's full-frame buffer, `bpp` bytes per pixel.  stage: copy into this rank's pinned staging
// buffer instead (finish_rows_out moves them on after the stream has been waited for).
int copy_rows_out(vrc_caster *h, const void *dev, size_t bpp, void *host, bool stage) {
    const size_t row = (size_t)h->width * bpp;
    if (stage) {
        const size_t bytes = row * (size_t)std::max(h->buffer_rows, 1);
        if (h->pinned_stage_bytes < bytes) {
            if (h->pinned_stage) (void)hipHostFree(h->pinned_stage);
            h->pinned_stage = nullptr; h->pinned_stage_bytes = 0;
            HIP_TRY(h, hipHostMalloc(&h->pinned_stage, bytes, hipHostMallocDefault));
            h->pinned_stage_bytes = bytes;
        }
        HIP_TRY(h, hipMemcpyAsync(h->pinned_stage, dev, row * (size_t)(h->sliced ? h->buffer_rows : h->height), hipMemcpyDeviceToHost, h->stream));
        return VRC_OK;
    }
    if (!h->sliced) {
        HIP_TRY(h, hipMemcpyAsync(host, dev, row * (size_t)h->height, hipMemcpyDeviceToHost, h->stream));
        return VRC_OK;
    }
    for (const RowRun &r : row_runs(h, true))
        HIP_TRY(h, hipMemcpyAsync((char *)host + row * (size_t)r.y0, (const char *)dev + row * (size_t)r.b0, row * (size_t)r.n,
                                  hipMemcpyDeviceToHost, h->stream));
    return VRC_OK;
}
// second half of a staged read-back (after the rank's stream has been waited for): staging buffer -> the caller's frame
void finish_rows_out(vrc_caster *h, size_t bpp, void *host) {
    const size_t row = (size_t)h->width * bpp;
    if (!h->sliced) { memcpy(host, h->pinned_stage, row * (size_t)h->height); return; }
    for (const RowRun &r : row_runs(h, true))
        memcpy((char *)host + row * (size_t)r.y0, (const char *)h->pinned_stage + row * (size_t)r.b0, row * (size_t)r.n);
}

int sync_one(vrc_caster *h) {
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    drain_events(h);
    if (h->wd_flag && *(volatile unsigned int *)h->wd_flag) {
        *(volatile unsigned int *)h->wd_flag = 0;
        return fail(h, VRC_ERR_DEVICE, "the kernel's round watchdog stopped a wavefront: the frame is invalid (setting watchdog_rounds)");
    }
    return VRC_OK;
}

}  // namespace

extern "C" {

int vrc_device_count(int *count) {
    if (!count) return VRC_ERR_INVALID_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return VRC_ERR_DEVICE; }
    *count = n;
    return VRC_OK;
}

int vrc_create(int device_ordinal, vrc_caster **out) {
    if (!out) return VRC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VRC_ERR_DEVICE;   // no GPU: no CPU fallback
    if (device_ordinal < 0 || device_ordinal >= n) return VRC_ERR_INVALID_ARGUMENT;
    vrc_caster *h = new vrc_caster();
    h->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&h->d_counters, sizeof(unsigned long long) * vrc::kCtrCount) != hipSuccess ||
        hipMalloc((void **)&h->d_frame, sizeof(int32_t) * 4) != hipSuccess ||
        hipHostMalloc((void **)&h->wd_flag, sizeof(unsigned int), hipHostMallocMapped) != hipSuccess) {
        delete h;
        return VRC_ERR_DEVICE;
    }
    *h->wd_flag = 0;
    (void)hipMemset(h->d_counters, 0, sizeof(unsigned long long) * vrc::kCtrCount);
    (void)hipMemset(h->d_frame, 0, sizeof(int32_t) * 4);
    *out = h;
    return VRC_OK;
}

int vrc_create_group_ex(const int32_t *device_ordinals, int32_t n, int32_t band_rows, uint32_t flags, vrc_caster **out) {
    if (!out || !device_ordinals || n < 1 || band_rows < vrc::kTileH || band_rows % vrc::kTileH) return VRC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    vrc_caster *root = nullptr;
    int rc = vrc_create(device_ordinals[0], &root);
    if (rc != VRC_OK) return rc;
    root->tile_rank = 0; root->tile_world = n; root->band_rows = band_rows; root->sliced = n > 1;
    std::string notes;
    for (int32_t r = 1; r < n; r++) {
        vrc_caster *q = nullptr;
        rc = vrc_create(device_ordinals[r], &q);
        if (rc != VRC_OK) { vrc_destroy(root); return rc; }
        q->tile_rank = r; q->tile_world = n; q->band_rows = band_rows; q->sliced = true; q->is_peer = true;
        q->own_copy = (flags & VRC_GROUP_OWN_COPIES) != 0;
        root->peers.push_back(q);
        if (q->device != root->device) {                       // SVO fan-out goes device to device over xGMI
            int can = 0;
            q->peer_access = 0;
            if (hipDeviceCanAccessPeer(&can, q->device, root->device) == hipSuccess && can) {
                (void)hipSetDevice(q->device);
                const hipError_t e = hipDeviceEnablePeerAccess(root->device, 0);
                if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) q->peer_access = 1;
            }
            (void)hipGetLastError();
            if (!q->peer_access) {
                char buf[160];
                snprintf(buf, sizeof(buf), "rank %d: no direct peer access from GPU %d to GPU %d, tree copies are staged by the runtime; ",
                         r, q->device, root->device);
                notes += buf;
            }
        }
    }
    (void)hipSetDevice(root->device);
    root->error = notes;                   // not a failure: vrc_last_error right after creation tells which ranks fell back
    *out = root;
    return VRC_OK;
}

int vrc_create_group(const int32_t *device_ordinals, int32_t n, int32_t band_rows, vrc_caster **out) {
    // VRC_GROUP_OWN_COPIES=1 in the environment: rehearse the distinct-GPU code paths on a box with one GPU
    const char *env = getenv("VRC_GROUP_OWN_COPIES");
    return vrc_create_group_ex(device_ordinals, n, band_rows, (env && env[0] == '1') ? VRC_GROUP_OWN_COPIES : 0u, out);
}

int vrc_group_size(const vrc_caster *h, int32_t *n) {
    if (!h || !n) return VRC_ERR_INVALID_ARGUMENT;
    *n = (int32_t)h->peers.size() + 1;
    return VRC_OK;
}

int vrc_destroy(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    for (vrc_caster *q : h->peers) vrc_destroy(q);
    h->peers.clear();
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drain_events(h);
    for (auto &p : h->pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    release_tree(h);
    release(h->d_map);
    release(h->d_viewport); release(h->d_image); release(h->d_hits); release(h->d_rgba8); release(h->d_jump_cache); release(h->d_jump_slots); h->jump_slot_count = 0; release(h->d_atlas);
    release(h->d_partials); release(h->d_counters); release(h->d_frame);
    if (h->wd_flag) (void)hipHostFree(h->wd_flag);
    if (h->pinned_stage) (void)hipHostFree(h->pinned_stage);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return VRC_OK;
}

const char *vrc_last_error(const vrc_caster *h) { return h ? h->error.c_str() : "null handle"; }

int vrc_assign_map(vrc_caster *h, const int8_t *voxels, int32_t dx, int32_t dy, int32_t dz) {
    if (!h || !voxels || dx <= 0 || dy <= 0 || dz <= 0) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_map: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    release(h->d_map);
    const size_t bytes = (size_t)dx * dy * dz;
    HIP_TRY(h, hipMalloc((void **)&h->d_map, bytes));
    HIP_TRY(h, hipMemcpy(h->d_map, voxels, bytes, hipMemcpyHostToDevice));
    h->map_dim[0] = dx; h->map_dim[1] = dy; h->map_dim[2] = dz;
    h->validated = false;
    FOR_PEERS(h, vrc_assign_map(q, voxels, dx, dy, dz));
    return VRC_OK;
}

int vrc_release_map(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_map) return fail(h, VRC_ERR_NOT_FOUND, "release_map: no map assigned");
    release(h->d_map);
    h->map_dim[0] = h->map_dim[1] = h->map_dim[2] = 0;
    h->validated = false;
    FOR_PEERS(h, vrc_release_map(q));
    return VRC_OK;
}

}  // extern "C"

namespace {

// Give every other rank of the group rank 0's tree: shared when the rank sits on the same GPU, copied device to
// device otherwise (hipMemcpyPeerAsync: xGMI, no host staging; SURVEY 8e "peer fan-out").
int fan_out_tree(vrc_caster *h) {
    mirror_tree(h);
    if (!h->peers.empty()) {                                   // the copies run on the peers' streams: rank 0's tree must be complete
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    for (size_t i = 0; i < h->peers.size(); i++) {
        vrc_caster *q = h->peers[i];
        release_tree(q);
        if (q->device == h->device && !q->own_copy) {
            q->tree = h->tree;                                 // one tree, one table, one set of boxes for the ranks of one GPU
            q->owns_desc = false;
        } else {
            HIP_TRY(h, hipSetDevice(q->device));
            vrc_tree *t = new_tree(q);
            t->n_desc = h->n_desc;
            HIP_TRY(h, hipMalloc((void **)&t->d_desc, h->n_desc * sizeof(uint64_t)));
            HIP_TRY(h, hipMemcpyPeerAsync(t->d_desc, q->device, h->d_desc, h->device, h->n_desc * sizeof(uint64_t), q->stream));
            if (h->d_attach_lookup && h->d_attach) {
                t->n_attach = h->n_attach;
                HIP_TRY(h, hipMalloc((void **)&t->d_attach_lookup, h->n_desc * sizeof(uint32_t)));
                HIP_TRY(h, hipMalloc((void **)&t->d_attach, std::max<uint64_t>(h->n_attach, 1) * sizeof(uint64_t)));
                HIP_TRY(h, hipMemcpyPeerAsync(t->d_attach_lookup, q->device, h->d_attach_lookup, h->device, h->n_desc * sizeof(uint32_t), q->stream));
                HIP_TRY(h, hipMemcpyPeerAsync(t->d_attach, q->device, h->d_attach, h->device,
                                              std::max<uint64_t>(h->n_attach, 1) * sizeof(uint64_t), q->stream));
            }
        }
        mirror_tree(q);
        q->have_octree = true; q->validated = false;
        int rc = set_setting(q, "octree_root_index", "OCTREE_ROOT_INDEX", setting_or(h, "octree_root_index", 0));
        if (rc != VRC_OK) return rc;
    }
    for (vrc_caster *q : h->peers)
        if (q->owns_desc) { HIP_TRY(h, hipSetDevice(q->device)); HIP_TRY(h, hipStreamSynchronize(q->stream)); }
    HIP_TRY(h, hipSetDevice(h->device));
    return VRC_OK;
}

bool lookup_in_range(const uint32_t *lookup, size_t n, uint64_t n_attach) {
    const uint64_t limit = std::max<uint64_t>(n_attach, 1);
    for (size_t i = 0; i < n; i++)
        if (lookup[i] >= limit) return false;
    return true;
}

}  // namespace

extern "C" {

int vrc_assign_octree(vrc_caster *h, const uint64_t *descriptors, uint64_t n, uint64_t root_index) {
    if (!h || !descriptors || n == 0 || root_index >= n) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    for (vrc_caster *q : h->peers) release_tree(q);
    release_tree(h);                                               // a new tree never inherits the old one's materials
    vrc_tree *t = new_tree(h);
    HIP_TRY(h, hipMalloc((void **)&t->d_desc, n * sizeof(uint64_t)));
    HIP_TRY(h, hipMemcpy(t->d_desc, descriptors, n * sizeof(uint64_t), hipMemcpyHostToDevice));
    t->n_desc = n;
    mirror_tree(h);
    h->have_octree = true;
    int rc = set_setting(h, "octree_root_index", "OCTREE_ROOT_INDEX", (int64_t)root_index);   // CLCaster.cpp:113
    if (rc != VRC_OK) return rc;
    return fan_out_tree(h);
}

// The tree another handle on the same GPU already holds, adopted instead of uploaded again: one descriptor array, one coarse
// table, one set of empty boxes between the two (vrc_tree).  The arrays live until the last handle lets go of them.
int vrc_assign_octree_from(vrc_caster *h, vrc_caster *src) {
    if (!h || !src || h == src) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_from: bad argument");
    if (!src->tree || !src->have_octree) return fail(h, VRC_ERR_NOT_READY, "assign_octree_from: the source handle has no octree");
    if (src->device != h->device) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_from: the handles sit on different GPUs (%d, %d)", h->device, src->device);
    HIP_TRY(h, hipSetDevice(h->device));
    std::shared_ptr<vrc_tree> keep = src->tree;                    // (src may be a peer of h's own group)
    for (vrc_caster *q : h->peers) release_tree(q);
    release_tree(h);
    h->tree = keep;
    h->owns_desc = false;
    mirror_tree(h);
    h->have_octree = true;
    const int rc = set_setting(h, "octree_root_index", "OCTREE_ROOT_INDEX", setting_or(src, "octree_root_index", 0));
    if (rc != VRC_OK) return rc;
    return fan_out_tree(h);
}

int vrc_assign_octree_attachments(vrc_caster *h, const uint32_t *lookup, uint64_t n_lookup,
                                  const uint64_t *attachments, uint64_t n_attachments) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->have_octree) return fail(h, VRC_ERR_NOT_READY, "assign_octree_attachments: assign the octree first");
    HIP_TRY(h, hipSetDevice(h->device));
    // the arguments are checked before anything is touched: a rejected call leaves every rank as it was
    const bool have = lookup && n_lookup && attachments && n_attachments;
    if (have) {
        if (n_lookup != h->n_desc)
            return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_attachments: lookup must have one entry per descriptor");
        if (!lookup_in_range(lookup, (size_t)n_lookup, n_attachments))
            return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_attachments: a lookup entry points past the attachment buffer");
    }
    // from here on every rank drops its old buffers first, so that a device failure half-way can never leave two ranks
    // of one frame with different materials (ranks that share rank 0's arrays just forget the pointers)
    // (the buffers belong to the TREE: every handle that shares it renders with the new materials from its next frame on)
    auto drop = [](vrc_caster *q) {
        vrc_tree *t = q->tree.get();
        if (t) {
            std::lock_guard<std::mutex> lock(t->guard);            // (another holder's frame copies these pointers under the guard)
            (void)hipSetDevice(t->device); release(t->d_attach_lookup); release(t->d_attach); t->n_attach = 0;
        }
        mirror_tree(q);
        q->validated = false;
    };
    for (vrc_caster *q : h->peers) drop(q);
    drop(h);
    HIP_TRY(h, hipSetDevice(h->device));
    vrc_tree *t = h->tree.get();
    if (have) {
        std::lock_guard<std::mutex> lock(t->guard);
        HIP_TRY(h, hipMalloc((void **)&t->d_attach_lookup, n_lookup * sizeof(uint32_t)));
        HIP_TRY(h, hipMemcpy(t->d_attach_lookup, lookup, n_lookup * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMalloc((void **)&t->d_attach, n_attachments * sizeof(uint64_t)));
        HIP_TRY(h, hipMemcpy(t->d_attach, attachments, n_attachments * sizeof(uint64_t), hipMemcpyHostToDevice));
        t->n_attach = n_attachments;
    }
    mirror_tree(h);
    if (h->peers.empty()) return VRC_OK;
    // ranks with their own copy of the descriptors: only the attachment buffers change
    for (vrc_caster *q : h->peers) {
        vrc_tree *tq = q->tree.get();
        if (tq != t && h->d_attach_lookup && h->d_attach) {
            HIP_TRY(h, hipSetDevice(q->device));
            HIP_TRY(h, hipMalloc((void **)&tq->d_attach_lookup, h->n_desc * sizeof(uint32_t)));
            HIP_TRY(h, hipMalloc((void **)&tq->d_attach, h->n_attach * sizeof(uint64_t)));
            HIP_TRY(h, hipMemcpyPeer(tq->d_attach_lookup, q->device, h->d_attach_lookup, h->device, h->n_desc * sizeof(uint32_t)));
            HIP_TRY(h, hipMemcpyPeer(tq->d_attach, q->device, h->d_attach, h->device, h->n_attach * sizeof(uint64_t)));
            tq->n_attach = h->n_attach;
        }
        mirror_tree(q);
    }
    HIP_TRY(h, hipSetDevice(h->device));
    return VRC_OK;
}

}  // extern "C"

namespace {
// file -> device through two pinned staging buffers: the read of chunk k+1 overlaps the copy of chunk k.
// check (optional) looks at every staged chunk before it is sent.
int stream_to_device(vrc_caster *h, FILE *f, void *dst, size_t bytes, void *stage[2], size_t chunk,
                     bool (*check)(const void *, size_t, uint64_t), uint64_t check_arg) {
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) HIP_TRY(h, hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    int rc = VRC_OK;
    size_t off = 0;
    for (int k = 0; off < bytes; k ^= 1) {
        const size_t n = std::min(chunk, bytes - off);
        if (hipEventSynchronize(done[k]) != hipSuccess) { rc = fail(h, VRC_ERR_DEVICE, "assign_octree_file: event wait failed"); break; }
        if (fread(stage[k], 1, n, f) != n) { rc = fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_file: file is truncated"); break; }
        if (check && !check(stage[k], n, check_arg)) { rc = fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_file: a lookup entry points past the attachment buffer"); break; }
        if (hipMemcpyAsync((char *)dst + off, stage[k], n, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipEventRecord(done[k], h->stream) != hipSuccess) { rc = fail(h, VRC_ERR_DEVICE, "assign_octree_file: upload failed"); break; }
        off += n;
    }
    (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 2; i++) (void)hipEventDestroy(done[i]);
    return rc;
}
bool check_lookup_chunk(const void *p, size_t bytes, uint64_t n_attach) {
    return lookup_in_range((const uint32_t *)p, bytes / sizeof(uint32_t), n_attach);
}
}  // namespace

extern "C" {

int vrc_assign_octree_file(vrc_caster *h, const char *path, uint32_t *dim) {
    if (!h || !path || !dim) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_file: null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    FILE *f = fopen(path, "rb");
    if (!f) return fail(h, VRC_ERR_NOT_FOUND, "assign_octree_file: cannot open '%s'", path);
    char magic[8];
    uint32_t flags = 0;
    uint64_t root = 0, n = 0, na = 0;
    const bool header = fread(magic, 1, 8, f) == 8 && memcmp(magic, "VRCSVO01", 8) == 0 && fread(dim, 4, 1, f) == 1 &&
                        fread(&flags, 4, 1, f) == 1 && fread(&root, 8, 1, f) == 1 && fread(&n, 8, 1, f) == 1 &&
                        fread(&na, 8, 1, f) == 1 && n > 0 && root < n && *dim >= 2 && (*dim & (*dim - 1)) == 0;
    if (!header) { fclose(f); return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_file: '%s' is not a VRCSVO01 file", path); }
    for (vrc_caster *q : h->peers) release_tree(q);
    release_tree(h);
    const size_t chunk = (size_t)setting_or(h, "upload_chunk_bytes", 64 << 20);
    void *stage[2] = {nullptr, nullptr};
    int rc = VRC_OK;
    if (chunk < 4096 || hipHostMalloc(&stage[0], chunk, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&stage[1], chunk, hipHostMallocDefault) != hipSuccess)
        rc = fail(h, VRC_ERR_OUT_OF_MEMORY, "assign_octree_file: no pinned staging memory");
    vrc_tree *t = new_tree(h);
    if (rc == VRC_OK && hipMalloc((void **)&t->d_desc, n * sizeof(uint64_t)) != hipSuccess)
        rc = fail(h, VRC_ERR_OUT_OF_MEMORY, "assign_octree_file: %llu descriptors do not fit in device memory", (unsigned long long)n);
    if (rc == VRC_OK) rc = stream_to_device(h, f, t->d_desc, n * sizeof(uint64_t), stage, chunk, nullptr, 0);
    if (rc == VRC_OK && (flags & 1u)) {
        if (hipMalloc((void **)&t->d_attach_lookup, n * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void **)&t->d_attach, (na ? na : 1) * sizeof(uint64_t)) != hipSuccess)
            rc = fail(h, VRC_ERR_OUT_OF_MEMORY, "assign_octree_file: attachment buffers do not fit in device memory");
        if (rc == VRC_OK && na == 0 && hipMemset(t->d_attach, 0x05, sizeof(uint64_t)) != hipSuccess)
            rc = fail(h, VRC_ERR_DEVICE, "assign_octree_file: memset failed");
        if (rc == VRC_OK) rc = stream_to_device(h, f, t->d_attach_lookup, n * sizeof(uint32_t), stage, chunk, check_lookup_chunk, na);
        if (rc == VRC_OK) rc = stream_to_device(h, f, t->d_attach, na * sizeof(uint64_t), stage, chunk, nullptr, 0);
    }
    fclose(f);
    for (int i = 0; i < 2; i++) if (stage[i]) (void)hipHostFree(stage[i]);
    if (rc != VRC_OK) { release_tree(h); return rc; }
    t->n_desc = n; t->n_attach = (flags & 1u) ? std::max<uint64_t>(na, 1) : 0;
    mirror_tree(h);
    h->have_octree = true;
    rc = set_setting(h, "octree_root_index", "OCTREE_ROOT_INDEX", (int64_t)root);   // CLCaster.cpp:113
    if (rc != VRC_OK) return rc;
    return fan_out_tree(h);
}

int vrc_build_shell_terrain(vrc_caster *h, uint32_t depth, uint64_t seed, int32_t thickness, int32_t octave_floor,
                            uint32_t flags, uint64_t validate_samples, const int32_t *probe_xy, uint32_t n_probe,
                            int32_t *probe_lohi, vrc_build_info *info) {
    if (!h || depth < 3 || depth > 16 || thickness < 0 || octave_floor < 0)
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "build_shell_terrain: need 3 <= depth <= 16, thickness >= 0, octave_floor >= 0");
    HIP_TRY(h, hipSetDevice(h->device));
    const bool count_only = (flags & VRC_BUILD_COUNT_ONLY) != 0;
    if (!count_only) {
        for (vrc_caster *q : h->peers) release_tree(q);
        release_tree(h);
    }
    uint64_t *d = nullptr;
    vrc_build_info bi;
    std::string err;
    const int rc = vrc::build_shell_terrain_device(h->stream, depth, seed, thickness, octave_floor, flags, validate_samples, probe_xy,
                                                   n_probe, probe_lohi, &d, &bi, err);
    if (info) *info = bi;
    if (rc != VRC_OK) return fail(h, rc, "build_shell_terrain: %s", err.c_str());
    if (count_only) return VRC_OK;
    { vrc_tree *t = new_tree(h); t->d_desc = d; t->n_desc = bi.n_descriptors; mirror_tree(h); }
    h->have_octree = true;
    const int rs = set_setting(h, "octree_root_index", "OCTREE_ROOT_INDEX", (int64_t)bi.root_index);
    if (rs != VRC_OK) return rs;
    return fan_out_tree(h);
}

int vrc_build_heightfield(vrc_caster *h, uint32_t depth, const uint16_t *hi, const uint16_t *lo, uint32_t flags,
                          uint64_t validate_samples, vrc_build_info *info) {
    if (!h || depth < 3 || depth > 16 || !hi)
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "build_heightfield: need 3 <= depth <= 16 and a height field");
    const size_t n = (size_t)1 << (2 * depth);
    for (size_t i = 0; i < n; i++)
        if (hi[i] >= (1u << depth) || (lo && lo[i] > hi[i]))
            return fail(h, VRC_ERR_INVALID_ARGUMENT, "build_heightfield: column %zu needs lo <= hi < dim", i);
    HIP_TRY(h, hipSetDevice(h->device));
    const bool count_only = (flags & VRC_BUILD_COUNT_ONLY) != 0;
    if (!count_only) {
        for (vrc_caster *q : h->peers) release_tree(q);
        release_tree(h);
    }
    uint64_t *d = nullptr;
    vrc_build_info bi;
    std::string err;
    const int rc = vrc::build_columns_device(h->stream, depth, 0, 0, 0, hi, lo, flags, validate_samples, nullptr, 0, nullptr, &d, &bi, err);
    if (info) *info = bi;
    if (rc != VRC_OK) return fail(h, rc, "build_heightfield: %s", err.c_str());
    if (count_only) return VRC_OK;
    { vrc_tree *t = new_tree(h); t->d_desc = d; t->n_desc = bi.n_descriptors; mirror_tree(h); }
    h->have_octree = true;
    const int rs = set_setting(h, "octree_root_index", "OCTREE_ROOT_INDEX", (int64_t)bi.root_index);
    if (rs != VRC_OK) return rs;
    return fan_out_tree(h);
}

int vrc_build_dense_grid(vrc_caster *h, uint32_t depth, const int8_t *grid, uint32_t flags, uint64_t validate_samples,
                         vrc_build_info *info) {
    if (!h || depth < 3 || depth > 12)
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "build_dense_grid: need 3 <= depth <= 12");
    const int32_t dim = (int32_t)1 << depth;
    if (!grid && !(h->d_map && h->map_dim[0] == dim && h->map_dim[1] == dim && h->map_dim[2] == dim))
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "build_dense_grid: no grid given and no %d^3 map assigned (vrc_assign_map) to build from", dim);
    HIP_TRY(h, hipSetDevice(h->device));
    const bool count_only = (flags & VRC_BUILD_COUNT_ONLY) != 0;
    if (!count_only) {
        for (vrc_caster *q : h->peers) release_tree(q);
        release_tree(h);
    }
    uint64_t *d = nullptr;
    vrc_build_info bi;
    std::string err;
    uint32_t *lookup = nullptr;
    uint64_t *attach = nullptr, n_attach = 0;
    const int rc = vrc::build_grid_device(h->stream, depth, grid, grid ? nullptr : h->d_map, flags, validate_samples, &d, &lookup, &attach,
                                          &n_attach, &bi, err);
    if (info) *info = bi;
    if (rc != VRC_OK) return fail(h, rc, "build_dense_grid: %s", err.c_str());
    if (count_only) return VRC_OK;
    {
        vrc_tree *t = new_tree(h);
        t->d_desc = d; t->n_desc = bi.n_descriptors;
        t->d_attach_lookup = lookup; t->d_attach = attach; t->n_attach = n_attach;
        mirror_tree(h);
    }
    h->have_octree = true;
    const int rs = set_setting(h, "octree_root_index", "OCTREE_ROOT_INDEX", (int64_t)bi.root_index);
    if (rs != VRC_OK) return rs;
    return fan_out_tree(h);
}

int vrc_read_descriptors(vrc_caster *h, uint64_t first, uint64_t count, uint64_t *out) {
    if (!h || !out) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->have_octree) return fail(h, VRC_ERR_NOT_READY, "read_descriptors: no octree assigned");
    if (first > h->n_desc || count > h->n_desc - first) return fail(h, VRC_ERR_INVALID_ARGUMENT, "read_descriptors: range past the array");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(out, h->d_desc + first, count * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VRC_OK;
}

int vrc_octree_size(vrc_caster *h, uint64_t *n_descriptors, uint64_t *root_index) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->have_octree) return fail(h, VRC_ERR_NOT_READY, "octree_size: no octree assigned");
    if (n_descriptors) *n_descriptors = h->n_desc;
    if (root_index) *root_index = (uint64_t)setting_or(h, "octree_root_index", 0);
    return VRC_OK;
}

int vrc_release_octree(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_desc) return fail(h, VRC_ERR_NOT_FOUND, "release_octree: no octree assigned");
    for (vrc_caster *q : h->peers) release_tree(q);               // ranks sharing rank 0's arrays must let go first
    release_tree(h);
    return VRC_OK;
}

int vrc_create_viewport(vrc_caster *h, int32_t width, int32_t height, float v_fov, float h_fov) {
    (void)v_fov; (void)h_fov;              // ignored by the reference too (CLCaster.cpp:233-275)
    if (!h || width <= 0 || height <= 0) return fail(h, VRC_ERR_INVALID_ARGUMENT, "create_viewport: bad size");
    const int rc = install_viewport(h, width, height, nullptr);
    if (rc != VRC_OK) return rc;
    FOR_PEERS(h, vrc_create_viewport(q, width, height, v_fov, h_fov));
    return VRC_OK;
}

int vrc_create_viewport_table(vrc_caster *h, int32_t width, int32_t height, const float *table) {
    if (!h || !table || width <= 0 || height <= 0) return fail(h, VRC_ERR_INVALID_ARGUMENT, "create_viewport_table: bad argument");
    const int rc = install_viewport(h, width, height, table);
    if (rc != VRC_OK) return rc;
    FOR_PEERS(h, vrc_create_viewport_table(q, width, height, table));
    return VRC_OK;
}

int vrc_release_viewport(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_viewport) return fail(h, VRC_ERR_NOT_FOUND, "release_viewport: no viewport");
    release(h->d_viewport); release(h->d_image); release(h->d_hits); release(h->d_rgba8); release(h->d_jump_cache); release(h->d_jump_slots); h->jump_slot_count = 0;
    h->width = h->height = h->buffer_rows = 0; h->validated = false;
    FOR_PEERS(h, vrc_release_viewport(q));
    return VRC_OK;
}

int vrc_create_texture_atlas(vrc_caster *h, const uint8_t *rgba8, int32_t width, int32_t height,
                             int32_t tile_w, int32_t tile_h) {
    if (!h || !rgba8 || width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0)
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "create_texture_atlas: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    release(h->d_atlas);
    HIP_TRY(h, hipMalloc((void **)&h->d_atlas, (size_t)4 * width * height));
    HIP_TRY(h, hipMemcpy(h->d_atlas, rgba8, (size_t)4 * width * height, hipMemcpyHostToDevice));
    h->atlas_w = width; h->atlas_h = height; h->tile_w = tile_w; h->tile_h = tile_h;
    h->validated = false;
    FOR_PEERS(h, vrc_create_texture_atlas(q, rgba8, width, height, tile_w, tile_h));
    return VRC_OK;
}

int vrc_assign_camera(vrc_caster *h, const float *direction2, const float *position3) {
    if (!h || !direction2 || !position3) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_camera: null pointer");
    h->cam_dir = direction2; h->cam_pos = position3;
    h->validated = false;
    FOR_PEERS(h, vrc_assign_camera(q, direction2, position3));
    return VRC_OK;
}

int vrc_assign_camera_trig(vrc_caster *h, const float *trig4) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    h->cam_trig = trig4;                   // (NULL: back to sinf / cosf of the live direction)
    FOR_PEERS(h, vrc_assign_camera_trig(q, trig4));
    return VRC_OK;
}

int vrc_release_camera(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    h->cam_dir = h->cam_pos = nullptr; h->cam_trig = nullptr; h->validated = false;
    FOR_PEERS(h, vrc_release_camera(q));
    return VRC_OK;
}

int vrc_assign_lights(vrc_caster *h, const float *packed, const int32_t *light_count) {
    if (!h || !packed || !light_count) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_lights: null pointer");
    h->lights = packed; h->light_count = light_count;
    h->validated = false;
    FOR_PEERS(h, vrc_assign_lights(q, packed, light_count));
    return VRC_OK;
}

int vrc_setting_add(vrc_caster *h, const char *name, const char *define, int64_t value) {
    if (!h || !name) return VRC_ERR_INVALID_ARGUMENT;
    const int rc = set_setting(h, name, define, value);
    if (rc != VRC_OK) return rc;
    FOR_PEERS(h, vrc_setting_add(q, name, define, value));
    return VRC_OK;
}

int vrc_setting_set(vrc_caster *h, const char *name, int64_t value) {
    if (!h || !name) return VRC_ERR_INVALID_ARGUMENT;
    int i = find_setting(h, name);
    if (i < 0) return fail(h, VRC_ERR_NOT_FOUND, "overwrite_setting: no setting named '%s'", name);   // CLCaster.cpp:1096-1100
    retry_derived(h, name);
    h->settings[i].value = value;
    FOR_PEERS(h, vrc_setting_set(q, name, value));
    return VRC_OK;
}

int vrc_setting_get(vrc_caster *h, const char *name, int64_t *value) {
    if (!h || !name || !value) return VRC_ERR_INVALID_ARGUMENT;
    int i = find_setting(h, name);
    if (i < 0) return fail(h, VRC_ERR_NOT_FOUND, "no setting named '%s'", name);
    *value = h->settings[i].value;
    return VRC_OK;
}

int vrc_set_row_tiling(vrc_caster *h, int32_t rank, int32_t world, int32_t band_rows) {
    if (!h || world < 1 || rank < 0 || rank >= world || band_rows < vrc::kTileH || band_rows % vrc::kTileH)
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "set_row_tiling: need 0 <= rank < world and band_rows a multiple of 8");
    if (!h->peers.empty() || h->is_peer) return fail(h, VRC_ERR_INVALID_ARGUMENT, "set_row_tiling: the ranks of a group are tiled by vrc_create_group");
    if (h->sliced) return fail(h, VRC_ERR_INVALID_ARGUMENT, "set_row_tiling: this handle holds a row slice (vrc_set_row_slice); release the viewport first");
    h->tile_rank = rank; h->tile_world = world; h->band_rows = band_rows;
    return VRC_OK;
}

int vrc_set_row_slice(vrc_caster *h, int32_t rank, int32_t world, int32_t band_rows) {
    if (!h || world < 1 || rank < 0 || rank >= world || band_rows < vrc::kTileH || band_rows % vrc::kTileH)
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "set_row_slice: need 0 <= rank < world and band_rows a multiple of 8");
    if (!h->peers.empty() || h->is_peer) return fail(h, VRC_ERR_INVALID_ARGUMENT, "set_row_slice: the ranks of a group are sliced by vrc_create_group");
    if (h->d_viewport) return fail(h, VRC_ERR_NOT_READY, "set_row_slice: call it before create_viewport (the buffers are sized by it)");
    h->tile_rank = rank; h->tile_world = world; h->band_rows = band_rows; h->sliced = world > 1;
    return VRC_OK;
}

int vrc_validate(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    h->validated = false;
    // CLCaster.cpp:165-184: camera, map, viewport image + matrix must be set
    if (!h->cam_dir || !h->cam_pos) return fail(h, VRC_ERR_NOT_READY, "validate: camera not assigned");
    if (!h->d_viewport || !h->d_image) return fail(h, VRC_ERR_NOT_READY, "validate: viewport not created");
    if (!h->lights) return fail(h, VRC_ERR_NOT_READY, "validate: lights not assigned");
    if (!h->d_atlas) return fail(h, VRC_ERR_NOT_READY, "validate: texture atlas not created");
    if (!h->have_octree) return fail(h, VRC_ERR_NOT_READY, "validate: octree not assigned");
    if (find_setting(h, "octree_dimensions") < 0) return fail(h, VRC_ERR_NOT_READY, "validate: setting octree_dimensions missing");
    if (find_setting(h, "using_octree") < 0) return fail(h, VRC_ERR_NOT_READY, "validate: setting using_octree missing");
    const int64_t dim = setting_or(h, "octree_dimensions", 0);
    const int n = log2_exact(dim);
    if (n < 1) return fail(h, VRC_ERR_INVALID_ARGUMENT, "validate: octree_dimensions must be a power of two >= 2");
    if (n > vrc::kMaxLevels) return fail(h, VRC_ERR_LIMIT, "validate: octree deeper than %d levels", vrc::kMaxLevels);
    const int64_t root = setting_or(h, "octree_root_index", 0);
    if (root < 0 || (uint64_t)root >= h->n_desc) return fail(h, VRC_ERR_INVALID_ARGUMENT, "validate: octree_root_index out of range");
    if (setting_or(h, "using_octree", 0) != 0) {
        if (!h->d_map) return fail(h, VRC_ERR_NOT_READY, "validate: dense map not assigned (using_octree != 0 selects the array branch)");
    }
    if (h->atlas_w / h->tile_w <= 0 || h->atlas_h / h->tile_h <= 0) return fail(h, VRC_ERR_INVALID_ARGUMENT, "validate: tile larger than atlas");
    h->validated = true;
    // the reference pays its one-off cost here (the kernel build, CLCaster.cpp:157-206); so do we: the structures the SVO kernels
    // derive from the tree are built now, not inside the first frame.  (each rank of a group prepares in its own vrc_validate)
    if (setting_or(h, "using_octree", 0) == 0) {
        const int rc = prepare_one(h);
        if (rc != VRC_OK) { h->validated = false; return rc; }
    }
    FOR_PEERS(h, vrc_validate(q));
    return VRC_OK;
}

int vrc_prepare(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    const int rc = prepare_one(h);
    if (rc != VRC_OK) return rc;
    FOR_PEERS(h, prepare_one(q));
    if (!h->peers.empty()) HIP_TRY(h, hipSetDevice(h->device));
    return VRC_OK;
}

}  // extern "C"

namespace {

// The structures the SVO kernels derive from a tree, for these settings: built when missing or built for another (root, depth,
// level); the kernel parameters get the pointers.  The caller holds t->guard.  Both structures are optional accelerations: when
// their memory cannot be had the frame is rendered without them (the table-less / box-less kernel instances); the reason is kept
// with the tree and reported by vrc_memory_usage2, and the build is tried again when what is asked for changes.
void derive_from_tree(vrc_caster *h, vrc_tree *t, int log2_dim, uint64_t root_index, int stepping_mode, vrc::RaycastParams &p) {
    // the levels above coarse_log2 as a dense table (setting coarse_log2: -1 = by depth and tree size, 0 = none), read by both
    // SVO kernels.  By default the finest level of the depth rule whose table is at most 16 x the descriptor array: a sparse
    // tree in a large map does not get a table hundreds of times its own size
    int64_t lc = setting_or(h, "coarse_log2", -1);
    if (lc < 0) {
        lc = vrc::coarse_level_for_depth(log2_dim);
        while (lc >= 1 && ((uint64_t)sizeof(uint64_t) << (3 * lc)) > 16 * sizeof(uint64_t) * t->n_desc && ((uint64_t)sizeof(uint64_t) << (3 * lc)) > (1u << 20)) lc--;
    }
    lc = std::min<int64_t>(lc, std::min(log2_dim - 2, 10));
    if (lc >= 1 && t->n_desc < (1ULL << 43)) {
        if (!t->d_coarse || t->coarse_log2 != (int)lc || t->coarse_root != root_index || t->coarse_depth != log2_dim) {
            release(t->d_coarse);
            release_boxes(t);                                      // (the boxes' parallel word belongs to the table's cells)
            t->coarse_log2 = 0;
            const bool failed_before = t->coarse_gave_up && t->coarse_fail_log2 == (int)lc && t->coarse_fail_root == root_index && t->coarse_fail_depth == log2_dim;
            if (!failed_before) {
                hipError_t e = hipMalloc((void **)&t->d_coarse, sizeof(uint64_t) << (3 * lc));
                if (e == hipSuccess) e = vrc::launch_coarse_build(t->d_desc, root_index, log2_dim, (int)lc, t->d_coarse, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);          // other handles read it from their own streams
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    release(t->d_coarse);
                    t->coarse_gave_up = true; t->coarse_fail_log2 = (int)lc; t->coarse_fail_root = root_index; t->coarse_fail_depth = log2_dim;
                    t->coarse_note = std::string("no coarse table (level ") + std::to_string(lc) + "): " + hipGetErrorString(e) + "; ";
                } else {
                    t->coarse_log2 = (int)lc; t->coarse_root = root_index; t->coarse_depth = log2_dim;
                    t->coarse_gave_up = false; t->coarse_note.clear();
                }
            }
        }
        if (t->d_coarse) { p.coarse = t->d_coarse; p.coarse_log2 = (int32_t)lc; }
    } else if (t->d_coarse) {
        release(t->d_coarse); t->coarse_log2 = 0;             // the setting went to "none": the table goes too
        release_boxes(t);
    }
    // the empty boxes (empty_boxes.hip; setting empty_boxes: -1 = when the tree is small enough for them, 0 = never, 1 = always):
    // 32 bytes per descriptor + 4 per table cell, built like the table they hang on; exact mode only.
    // (Words for the table's cells ALONE -- boxes in the coarse space, octree nodes below it -- would fit any tree; measured:
    // depth 12 1.71 ms against 1.50 with all words and 1.91 without, depth 14 -2 %, depth 16 +4 %: not offered.)
    // Two forms (setting empty_boxes: -1 = by the tree's size, 0 = never, 1 = a word per descriptor and child, 2 = the upper levels only):
    //   * a word per (descriptor, child): trees of up to 2^29 descriptors (16 GB of words) by default.  Beyond that the words are
    //     not only expensive -- gathered from an array of tens of GB they miss the TLB as well as the caches (the depth-15 bench
    //     terrain, 1.4 G descriptors: 4.8 ms with its 45 GB of words, 3.75 without);
    //   * the upper levels only (round 6): box records for the descriptors of the levels the record budget reaches (setting
    //     empty_box_records, default 400 M records = 14 GB of words, at most a quarter of the free device memory; setting
    //     empty_box_levels caps the levels), numbered breadth-first, the nodes below them widened over their empty siblings -- any
    //     tree, also beyond the 2^31 descriptors a 32-bit descriptor index reaches.
    const int64_t want_boxes = setting_or(h, "empty_boxes", -1);
    const bool box_ok = p.coarse != nullptr && stepping_mode == 0 && log2_dim <= 19;
    int box_mode = 0;                                              // 0 none, 1 per descriptor, 2 upper levels
    if (box_ok && want_boxes == 1 && t->n_desc < (1ULL << 31)) box_mode = 1;
    else if (box_ok && want_boxes >= 2) box_mode = 2;
    else if (box_ok && want_boxes < 0) box_mode = t->n_desc <= (1ULL << 29) ? 1 : 2;
    if (box_mode) {
        const int64_t want_levels = box_mode == 2 ? std::max<int64_t>(0, setting_or(h, "empty_box_levels", 0)) : 0;
        const int64_t want_records = box_mode == 2 ? std::max<int64_t>(0, setting_or(h, "empty_box_records", 0)) : 0;
        if (!t->d_boxes || t->box_mode != box_mode || t->box_levels_asked != want_levels || t->box_records_asked != want_records || t->box_log2 != (int)lc || t->box_root != root_index || t->box_depth != log2_dim) {
            release_boxes(t);
            const bool failed_before = t->boxes_gave_up && t->box_fail_log2 == (int)lc && t->box_fail_root == root_index && t->box_fail_depth == log2_dim && t->box_fail_mode == box_mode;
            if (!failed_before) {
                uint64_t *pos_tmp = nullptr;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                float ms = 0.f;
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
                hipError_t e = hipMalloc((void **)&t->d_box_aux, sizeof(uint32_t) << (3 * lc));
                if (e == hipSuccess) e = hipEventCreate(&e0);
                if (e == hipSuccess) e = hipEventCreate(&e1);
                if (e == hipSuccess) e = hipEventRecord(e0, h->stream);
                if (box_mode == 1) {
                    // (an optional structure must not be what makes the next allocation fail: at most half of what is free right now)
                    if (e == hipSuccess && want_boxes < 0 && t->n_desc > (1ULL << 28) && 40ULL * t->n_desc > free_b / 2) e = hipErrorOutOfMemory;
                    if (e == hipSuccess) e = hipMalloc((void **)&t->d_boxes, sizeof(uint32_t) * 8 * t->n_desc);
                    if (e == hipSuccess) e = hipMalloc((void **)&pos_tmp, sizeof(uint64_t) * t->n_desc);
                    if (e == hipSuccess) e = vrc::launch_box_build(t->d_desc, t->n_desc, root_index, log2_dim, (int)lc, pos_tmp, t->d_boxes, t->d_box_aux, h->stream);
                    if (e == hipSuccess) { t->box_records = t->n_desc; t->box_levels = log2_dim; }
                } else {
                    uint64_t budget = (uint64_t)want_records;
                    // (measured on the depth-16 terrain, 5.7 G descriptors: 200 M records 4.59 ms, 400 M 4.47, 1.5 G 5.04 -- beyond ~15 GB the
                    // words themselves become TLB misses; without boxes 4.73)
                    if (!budget) budget = std::min<uint64_t>(400000000ULL, (uint64_t)(free_b / 4) / 52);
                    budget = std::min<uint64_t>(budget, 0xffffff00ULL);
                    vrc::BoxUpper u;
                    if (e == hipSuccess) e = vrc::launch_box_build_upper(t->d_desc, root_index, log2_dim, (int)lc, std::max<uint64_t>(budget, 9), (int)want_levels, &u, t->d_box_aux, h->stream);
                    if (e == hipSuccess) {
                        t->d_boxes = u.boxes; t->d_box_child = u.child; t->d_box_desc = u.desc; t->d_box_pos = u.pos;
                        t->box_records = u.count; t->box_levels = u.levels;
                    }
                }
                if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                if (e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
                if (e == hipSuccess && vrc::box_queries_cut(&t->box_queries_cut) != hipSuccess) { (void)hipGetLastError(); t->box_queries_cut = 0; }
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                if (pos_tmp) (void)hipFree(pos_tmp);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    release_boxes(t);
                    t->boxes_gave_up = true; t->box_fail_log2 = (int)lc; t->box_fail_root = root_index; t->box_fail_depth = log2_dim; t->box_fail_mode = box_mode;
                    t->box_note = std::string("no empty boxes: ") + hipGetErrorString(e) + "; ";
                } else {
                    t->box_build_seconds = ms * 1e-3;
                    t->box_log2 = (int)lc; t->box_root = root_index; t->box_depth = log2_dim; t->box_mode = box_mode; t->box_levels_asked = want_levels; t->box_records_asked = want_records;
                    t->boxes_gave_up = false; t->box_note.clear();
                }
            }
        }
        if (t->d_boxes) { p.boxes = t->d_boxes; p.box_aux = t->d_box_aux; p.box_child = t->d_box_child; p.box_levels = t->box_levels; }
    }
    // (a handle that switches the boxes off keeps them: they belong to the tree, go with it, and a host that toggles the setting
    // between frames -- tests/soak_jumps_gpu.py does -- must not pay the build again and again)
}

// vrc_prepare for one rank: the derived structures for the settings as they stand
int prepare_one(vrc_caster *h) {
    if (!h->have_octree || !h->tree) return fail(h, VRC_ERR_NOT_READY, "prepare: octree not assigned");
    const int n = log2_exact(setting_or(h, "octree_dimensions", 0));
    if (n < 1 || n > vrc::kMaxLevels) return fail(h, VRC_ERR_NOT_READY, "prepare: setting octree_dimensions missing or not a power of two in [2, 2^%d]", vrc::kMaxLevels);
    const int64_t root = setting_or(h, "octree_root_index", 0);
    if (root < 0 || (uint64_t)root >= h->tree->n_desc) return fail(h, VRC_ERR_INVALID_ARGUMENT, "prepare: octree_root_index out of range");
    if (setting_or(h, "using_octree", 0) != 0) return VRC_OK;      // the array branch derives nothing
    HIP_TRY(h, hipSetDevice(h->device));
    vrc::RaycastParams p;
    memset(&p, 0, sizeof(p));
    std::lock_guard<std::mutex> lock(h->tree->guard);
    derive_from_tree(h, h->tree.get(), n, (uint64_t)root, (int)setting_or(h, "stepping_mode", 0), p);
    return VRC_OK;
}

int compute_async_one(vrc_caster *h) {
    if (!h->validated) return fail(h, VRC_ERR_NOT_READY, "compute: validate() has not succeeded");
    HIP_TRY(h, hipSetDevice(h->device));
    // The tree's guard is held from here until the kernel is ENQUEUED: another holder of the tree (another host thread, other
    // settings, new materials) that replaces one of its arrays frees the old one with hipFree, which waits for the kernels already
    // enqueued -- never for a frame that has copied the pointers and not launched yet (advisor finding, round 5).
    std::unique_lock<std::mutex> tree_lock;
    if (h->tree) {
        tree_lock = std::unique_lock<std::mutex>(h->tree->guard);
        mirror_tree(h);                    // (a handle that shares the tree may have given it new materials since the last frame)
    }

    // settings stay live after validate() (CLCaster::overwrite_setting needs no recompile, CLCaster.cpp:1087-1109), so
    // the structural ones are checked again here: a bad value is an error return, never a device fault
    vrc::RaycastParams p;
    memset(&p, 0, sizeof(p));
    const bool svo = setting_or(h, "using_octree", 0) == 0;
    const int64_t dim = setting_or(h, "octree_dimensions", 0);
    p.svo = svo ? 1 : 0;
    p.log2_dim = log2_exact(dim);
    if (p.log2_dim < 1 || p.log2_dim > vrc::kMaxLevels)
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "compute: octree_dimensions must be a power of two in [2, 2^%d]", vrc::kMaxLevels);
    const int64_t root = setting_or(h, "octree_root_index", 0);
    if (!h->d_desc || root < 0 || (uint64_t)root >= h->n_desc) return fail(h, VRC_ERR_INVALID_ARGUMENT, "compute: octree_root_index out of range");
    if (!svo && !h->d_map) return fail(h, VRC_ERR_NOT_READY, "compute: dense map not assigned (using_octree != 0 selects the array branch)");
    if (!h->d_viewport || !h->d_image || !h->d_atlas) return fail(h, VRC_ERR_NOT_READY, "compute: viewport or atlas released since validate()");
    p.stepping_mode = (int32_t)setting_or(h, "stepping_mode", 0);
    if (p.stepping_mode < 0 || p.stepping_mode > 1 || (p.stepping_mode == 1 && !svo))
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "compute: stepping_mode must be 0 (exact) or 1 (node-exit jumps, SVO branch only)");
    p.map = h->d_map;
    if (svo) { p.map_dim[0] = p.map_dim[1] = p.map_dim[2] = (int32_t)dim; }
    else { p.map_dim[0] = h->map_dim[0]; p.map_dim[1] = h->map_dim[1]; p.map_dim[2] = h->map_dim[2]; }
    p.width = h->width; p.height = h->height;
    if (setting_or(h, "hit_records", 1) != 0) {
        const int rc = ensure_hits(h);
        if (rc != VRC_OK) return rc;
        p.hits = h->d_hits;
    }
    h->last_frame_wrote_hits = p.hits != nullptr;
    p.viewport = h->d_viewport; p.image = h->d_image;
    p.atlas = h->d_atlas; p.atlas_w = h->atlas_w; p.atlas_h = h->atlas_h;
    p.tiles_x = h->atlas_w / h->tile_w; p.tiles_y = h->atlas_h / h->tile_h;
    p.descriptors = h->d_desc;
    p.attach_lookup = (h->d_attach_lookup && h->d_attach) ? h->d_attach_lookup : nullptr;
    p.attachments = p.attach_lookup ? h->d_attach : nullptr;
    p.root_index = (uint64_t)root;
    // live buffers are re-read every frame (CL_MEM_USE_HOST_PTR semantics)
    for (int a = 0; a < 3; a++) p.cam_pos[a] = h->cam_pos[a];
    if (h->cam_trig) {                     // the host's own sin / cos (vrc_assign_camera_trig; ray_caster_kernel.cl:280-291)
        for (int a = 0; a < 4; a++) p.trig[a] = h->cam_trig[a];
    } else {
        p.trig[0] = sinf(h->cam_dir[0]); p.trig[1] = cosf(h->cam_dir[0]);
        p.trig[2] = sinf(h->cam_dir[1]); p.trig[3] = cosf(h->cam_dir[1]);
    }
    // the reference binds light_count but shades with light 0 only (ray_caster_kernel.cl:264,660-670); setting
    // "light_count" (default 1) switches on the multi-light extension for the first n packed lights
    p.light_count = (int32_t)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(setting_or(h, "light_count", 1), *h->light_count),
                                                                    vrc::kMaxLights));
    double light_reach = 0.0;              // steps a shadow ray may take beyond max_distance: |voxel - light| (:667)
    for (int l = 0; l < p.light_count; l++) {
        for (int k = 0; k < 7; k++) p.lights[l][k] = h->lights[10 * l + k];
        p.lights[l][7] = 0.0f;
        double far2 = 0.0;
        for (int a = 0; a < 3; a++) {
            const double lo = std::fabs((double)p.lights[l][4 + a]), hi = std::fabs((double)p.lights[l][4 + a] - (double)p.map_dim[a]);
            far2 += std::max(lo, hi) * std::max(lo, hi);
        }
        light_reach += std::sqrt(far2) + 2.0;
    }
    {   // the kernel's step counter is the reference's int (:325): a cap beyond 2^30 would overflow it on the way
        const int64_t md = setting_or(h, "max_distance", 20);
        if (md > (int64_t)1 << 30 || md < -((int64_t)1 << 30))
            return fail(h, VRC_ERR_INVALID_ARGUMENT, "compute: max_distance %lld is outside [-2^30, 2^30]", (long long)md);
        p.max_distance = (int32_t)md;
    }
    p.shadow_rays = (int32_t)setting_or(h, "shadow_rays", 1);
    // wave scheduling knobs of the SVO kernel; they never change results
    p.burst_steps = (int32_t)std::min<int64_t>(1 << 20, std::max<int64_t>(1, setting_or(h, "burst_steps", vrc::kDefaultBurstSteps)));
    p.shade_threshold = std::min<int64_t>(64, std::max<int64_t>(1, setting_or(h, "shade_threshold", vrc::kDefaultShadeThreshold)));

    p.widen_nodes = (int32_t)setting_or(h, "widen_nodes", 1);
    p.octree_bias = (int32_t)setting_or(h, "octree_bias", 1);
    p.arith_mask = (int32_t)setting_or(h, "arith_mask", 1);
    // every round advances at least one lane by a step, an event or a hit block: 64 lanes x (the longest legal ray:
    // max_distance primary steps + per light the distance to the farthest map corner) with a 2x margin
    const double legal_steps = (double)std::max(p.max_distance, 0) + light_reach + 64.0;
    p.watchdog_rounds = (int32_t)std::min<int64_t>(INT32_MAX, std::max<int64_t>(1, setting_or(h, "watchdog_rounds",
                                    (int64_t)std::min(2.0e9, 128.0 * legal_steps))));
    p.watchdog_flag = h->wd_flag;
    p.safe_run = (int32_t)setting_or(h, "safe_run", 1);
    p.single_step = (int32_t)setting_or(h, "single_step", 1);
    p.exact_steps = (int32_t)std::min<int64_t>(1 << 20, std::max<int64_t>(1, setting_or(h, "exact_steps", vrc::kDefaultExactSteps)));
    p.xcd_mode = (int32_t)setting_or(h, "xcd_mode", 1);
    p.lds_pad_bytes = (int32_t)std::min<int64_t>(120 * 1024, std::max<int64_t>(0, setting_or(h, "lds_pad_bytes", 0)));
    p.frame = h->d_frame;

    const int tile_rows = (h->height + vrc::kTileH - 1) / vrc::kTileH;
    p.band_tiles = h->band_rows / vrc::kTileH;
    p.tile_rank = h->tile_rank; p.tile_world = h->tile_world;
    p.row_sliced = h->sliced ? 1 : 0;
    const int bands = (tile_rows + p.band_tiles - 1) / p.band_tiles;
    int local_rows = 0;
    for (int b = h->tile_rank; b < bands; b += h->tile_world) {
        const int first = b * p.band_tiles;
        local_rows += std::min(p.band_tiles, tile_rows - first);
    }
    p.local_tile_rows = local_rows;
    p.blocks_x = (h->width + vrc::kTileW * vrc::kTilesPerBlock - 1) / (vrc::kTileW * vrc::kTilesPerBlock);
    const int nblocks = p.blocks_x * p.local_tile_rows;
    if (nblocks > h->partial_blocks) {
        release(h->d_partials);
        h->partial_blocks = 0;
        HIP_TRY(h, hipMalloc((void **)&h->d_partials, sizeof(unsigned long long) * vrc::kCtrCount * (size_t)nblocks));
        h->partial_blocks = nblocks;
    }
    p.counters = h->d_partials;
    // What the kernels derive from the tree (the dense table of its top, the empty boxes) lives WITH the tree and is built by
    // vrc_prepare / vrc_validate; a frame that finds it missing or built for other settings builds it here (under the guard).
    if (svo) derive_from_tree(h, h->tree.get(), p.log2_dim, p.root_index, p.stepping_mode, p);
    // exact closed-form jumps: on from depth 12; the threshold depends on where the Euclid tables live (LDS when stack + tables
    // fit at full occupancy: depth 12)
    p.jump_tables_lds = (int32_t)std::min<int64_t>(2, std::max<int64_t>(0, setting_or(h, "jump_tables_lds", 2)));
    const int lds_rows = vrc::jump_tables_lds_rows(p);             // 3, 2 (deep trees with boxes) or 0 rows of the tables' ring in LDS
    const bool tables_in_lds = lds_rows > 0;
    p.jump_tables_lds = lds_rows;                                  // resolved once, here: the launch takes it as it is
    // (the jump instances read the tree's top from the coarse table: without one -- depth < 5, coarse_log2 = 0 -- there are no jumps)
    p.jump_min_run = !p.coarse ? vrc::kJumpOff : (int32_t)std::min<int64_t>(vrc::kJumpOff, std::max<int64_t>(1, setting_or(h, "jump_min_run",
                                    p.log2_dim >= (p.boxes ? vrc::kDefaultJumpMinDepthBoxes : vrc::kDefaultJumpMinDepth) ? (tables_in_lds ? vrc::kDefaultJumpMinRunLds : vrc::kDefaultJumpMinRun)
                                                                            : vrc::kJumpOff)));
    p.safe_steps = (int32_t)std::min<int64_t>(256, std::max<int64_t>(2, setting_or(h, "safe_steps",
                                    p.jump_min_run < vrc::kJumpOff ? vrc::kDefaultSafeStepsJump : vrc::kDefaultSafeSteps)));
    if (svo && p.stepping_mode == 0 && p.jump_min_run < vrc::kJumpOff && !tables_in_lds) {
        const int rc = ensure_jump_cache(h, nblocks);
        if (rc != VRC_OK) return rc;
        p.jump_cache = h->d_jump_cache; p.jump_slots = h->d_jump_slots; p.jump_slot_count = h->jump_slot_count;
    }
    h->last_frame_boxes = p.boxes != nullptr;
    h->last_blocks = nblocks;
    h->frames_enqueued++;

    vrc_caster::EvPair ev;
    if (!h->pool.empty()) { ev = h->pool.back(); h->pool.pop_back(); }
    else { HIP_TRY(h, hipEventCreate(&ev.a)); HIP_TRY(h, hipEventCreate(&ev.b)); }

    HIP_TRY(h, vrc::launch_frame_setup(p, h->stream));
    HIP_TRY(h, hipEventRecord(ev.a, h->stream));
    HIP_TRY(h, vrc::launch_raycast(p, h->stream));
    HIP_TRY(h, hipEventRecord(ev.b, h->stream));
    h->pending.push_back(ev);
    if (h->pending.size() > 4096) { HIP_TRY(h, hipStreamSynchronize(h->stream)); drain_events(h); }
    return VRC_OK;
}

// what read_image / read_hits / read_image_rgba8 have in common: every rank copies its rows into the caller's frame
// (asynchronously, each on its own stream), then all ranks are waited for
// A group reading into pageable memory stages every rank's tile through its own pinned buffer, so that the n copies
// still run side by side (a pageable hipMemcpyAsync blocks the host thread); a single handle copies directly.
template <class Enqueue>
int gather_rows(vrc_caster *h, void *host, size_t bpp, Enqueue enqueue) {
    const bool stage = !h->peers.empty() && !host_is_pinned(host, bpp * (size_t)h->width * (size_t)h->height);
    int rc = enqueue(h, stage);
    if (rc != VRC_OK) return rc;
    for (size_t i = 0; i < h->peers.size(); i++) {
        rc = enqueue(h->peers[i], stage);
        if (rc != VRC_OK) return fail(h, rc, "rank %zu: %s", i + 1, h->peers[i]->error.c_str());
    }
    rc = sync_one(h);
    if (rc != VRC_OK) return rc;
    if (stage) finish_rows_out(h, bpp, host);
    for (size_t pi = 0; pi < h->peers.size(); pi++) {
        vrc_caster *q = h->peers[pi];
        rc = sync_one(q);
        if (rc != VRC_OK) return fail(h, rc, "rank %zu: %s", pi + 1, q->error.c_str());
        if (stage) finish_rows_out(q, bpp, host);
    }
    HIP_TRY(h, hipSetDevice(h->device));
    return VRC_OK;
}

}  // namespace

extern "C" {

int vrc_compute_async(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    const int rc = compute_async_one(h);
    if (rc != VRC_OK) return rc;
    FOR_PEERS(h, compute_async_one(q));
    if (!h->peers.empty()) HIP_TRY(h, hipSetDevice(h->device));
    return VRC_OK;
}

int vrc_sync(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    int rc = sync_one(h);
    for (size_t i = 0; i < h->peers.size(); i++) {                 // always wait for every rank, then report the first failure
        const int rq = sync_one(h->peers[i]);
        if (rq != VRC_OK && rc == VRC_OK) rc = fail(h, rq, "rank %zu: %s", i + 1, h->peers[i]->error.c_str());
    }
    if (!h->peers.empty()) (void)hipSetDevice(h->device);
    return rc;
}

int vrc_compute(vrc_caster *h) {
    int rc = vrc_compute_async(h);
    if (rc != VRC_OK) return rc;
    return vrc_sync(h);                    // clFinish (CLCaster.cpp:970): the frame is complete on every rank
}

int vrc_read_image_f32(vrc_caster *h, float *rgba, size_t n_floats) {
    if (!h || !rgba) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_image) return fail(h, VRC_ERR_NOT_READY, "read_image: no viewport");
    if (n_floats < (size_t)4 * h->width * h->height) return fail(h, VRC_ERR_INVALID_ARGUMENT, "read_image_f32: buffer too small");
    return gather_rows(h, rgba, 16, [rgba](vrc_caster *q, bool stage) -> int {
        HIP_TRY(q, hipSetDevice(q->device));
        return copy_rows_out(q, q->d_image, 16, rgba, stage);
    });
}

// CLCaster::draw's source (CLCaster.cpp:278-296,330-332) is an RGBA8 texture the kernel's write_imagef quantises into:
// saturate, scale by 255, round to nearest even.  Packed on the device, 4 bytes per pixel cross PCIe.
int vrc_read_image_rgba8(vrc_caster *h, uint8_t *rgba, size_t n_bytes) {
    if (!h || !rgba) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_image) return fail(h, VRC_ERR_NOT_READY, "read_image: no viewport");
    if (n_bytes < (size_t)4 * h->width * h->height) return fail(h, VRC_ERR_INVALID_ARGUMENT, "read_image_rgba8: buffer too small");
    return gather_rows(h, rgba, 4, [rgba](vrc_caster *q, bool stage) -> int {
        HIP_TRY(q, hipSetDevice(q->device));
        const size_t npix = (size_t)q->width * (size_t)std::max(q->buffer_rows, 1);
        if (!q->d_rgba8) HIP_TRY(q, hipMalloc((void **)&q->d_rgba8, 4 * npix));
        HIP_TRY(q, vrc::launch_pack_rgba8(q->d_image, q->d_rgba8, npix, q->stream));
        return copy_rows_out(q, q->d_rgba8, 4, rgba, stage);
    });
}

int vrc_read_hits(vrc_caster *h, int32_t *hits, size_t n_int32) {
    if (!h || !hits) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_viewport) return fail(h, VRC_ERR_NOT_READY, "read_hits: no viewport");
    // the records must belong to the frame the image belongs to: with hit_records switched off since, the buffer still
    // holds an older frame's records
    if (!h->d_hits || !h->last_frame_wrote_hits)
        return fail(h, VRC_ERR_NOT_READY, "read_hits: no hit records (setting hit_records is 0, or no frame computed yet)");
    if (n_int32 < (size_t)8 * h->width * h->height) return fail(h, VRC_ERR_INVALID_ARGUMENT, "read_hits: buffer too small");
    return gather_rows(h, hits, 32, [hits](vrc_caster *q, bool stage) -> int {
        HIP_TRY(q, hipSetDevice(q->device));
        if (!q->d_hits || !q->last_frame_wrote_hits) return fail(q, VRC_ERR_NOT_READY, "read_hits: no hit records");
        return copy_rows_out(q, q->d_hits, 32, hits, stage);
    });
}

int vrc_device_image(vrc_caster *h, void **dev_ptr, size_t *n_bytes) {
    if (!h || !dev_ptr) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_image) return fail(h, VRC_ERR_NOT_READY, "device_image: no viewport");
    *dev_ptr = h->d_image;
    if (n_bytes) *n_bytes = (size_t)16 * h->width * (size_t)h->buffer_rows;
    return VRC_OK;
}

int vrc_pin_host_buffer(void *p, size_t bytes) {
    if (!p || !bytes) return VRC_ERR_INVALID_ARGUMENT;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) return VRC_OK;
    (void)hipGetLastError();
    return VRC_ERR_DEVICE;
}
int vrc_unpin_host_buffer(void *p) {
    if (!p) return VRC_ERR_INVALID_ARGUMENT;
    if (hipHostUnregister(p) == hipSuccess) return VRC_OK;
    (void)hipGetLastError();                 // reported through the return code, not left behind for the next launch
    return VRC_ERR_DEVICE;
}

// self-check of the empty boxes the last frame used: pseudo-random voxels inside pseudo-random boxes, looked up in the tree
int vrc_empty_boxes_check(vrc_caster *h, uint64_t samples, uint64_t seed, uint64_t *boxes_sampled, uint64_t *solid_voxels, double *build_seconds) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    vrc_tree *t = h->tree.get();
    if (!t) return fail(h, VRC_ERR_NOT_READY, "empty_boxes_check: no octree");
    std::lock_guard<std::mutex> lock(t->guard);
    if (!t->d_box_aux || !t->d_desc) return fail(h, VRC_ERR_NOT_READY, "empty_boxes_check: no boxes (setting empty_boxes, or neither vrc_prepare nor a frame has run yet)");
    HIP_TRY(h, hipSetDevice(h->device));
    uint64_t *pos = nullptr; unsigned long long *res = nullptr;
    hipError_t e = hipMalloc((void **)&res, 2 * sizeof(unsigned long long));
    unsigned long long out[2] = {0, 0}, cells[2] = {0, 0};
    if (t->d_boxes && t->box_mode == 2) {                      // the records of the upper levels (their descriptors and positions were kept)
        if (e == hipSuccess) e = vrc::launch_box_check(t->d_desc, t->box_records, t->box_root, t->box_depth, t->d_box_pos, t->d_box_desc, t->d_boxes, samples, seed, res, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipMemcpy(out, res, sizeof(out), hipMemcpyDeviceToHost);
    } else if (t->d_boxes) {                                   // the words per (descriptor, child)
        if (e == hipSuccess) e = hipMalloc((void **)&pos, sizeof(uint64_t) * t->n_desc);
        if (e == hipSuccess) e = vrc::launch_box_positions(t->d_desc, t->n_desc, t->box_root, t->box_depth, pos, h->stream);
        if (e == hipSuccess) e = vrc::launch_box_check(t->d_desc, t->n_desc, t->box_root, t->box_depth, pos, nullptr, t->d_boxes, samples, seed, res, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipMemcpy(out, res, sizeof(out), hipMemcpyDeviceToHost);
    }
    // the words of the table's cells (the coarse space)
    if (e == hipSuccess) e = vrc::launch_box_check_cells(t->d_desc, t->box_root, t->box_depth, t->box_log2, t->d_box_aux, samples, seed + 1, res, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(cells, res, sizeof(cells), hipMemcpyDeviceToHost);
    out[0] += cells[0]; out[1] += cells[1];
    if (pos) (void)hipFree(pos);
    if (res) (void)hipFree(res);
    HIP_TRY(h, e);
    if (boxes_sampled) *boxes_sampled = out[0];
    if (solid_voxels) *solid_voxels = out[1];
    if (build_seconds) *build_seconds = t->box_build_seconds;
    return VRC_OK;
}

// read-back of the box words (tests: every voxel of every box of a small tree against its dense grid, on the host)
int vrc_read_empty_boxes(vrc_caster *h, uint64_t first_descriptor, uint64_t count, uint32_t *out) {
    if (!h || !out) return VRC_ERR_INVALID_ARGUMENT;
    vrc_tree *t = h->tree.get();
    if (!t) return fail(h, VRC_ERR_NOT_READY, "read_empty_boxes: no octree");
    std::lock_guard<std::mutex> lock(t->guard);
    if (!t->d_boxes) return fail(h, VRC_ERR_NOT_READY, "read_empty_boxes: no boxes (setting empty_boxes, or neither vrc_prepare nor a frame has run yet)");
    if (t->box_mode != 1) return fail(h, VRC_ERR_NOT_READY, "read_empty_boxes: this tree has box records for its upper levels only (they are not indexed by descriptor)");
    if (first_descriptor > t->n_desc || count > t->n_desc - first_descriptor) return fail(h, VRC_ERR_INVALID_ARGUMENT, "read_empty_boxes: range past the array");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(out, t->d_boxes + 8 * first_descriptor, sizeof(uint32_t) * 8 * count, hipMemcpyDeviceToHost));
    return VRC_OK;
}

// vrc_memory_usage with a size-versioned struct (the caller says how large ITS struct is; never more than that is written)
int vrc_memory_usage2(vrc_caster *h, int32_t rank, vrc_memory2 *out) {
    if (!h || !out || rank < 0 || rank > (int32_t)h->peers.size() || out->struct_size < sizeof(uint32_t)) return VRC_ERR_INVALID_ARGUMENT;
    const vrc_caster *q = rank == 0 ? h : h->peers[rank - 1];
    vrc_memory2 m;
    memset(&m, 0, sizeof(m));
    const size_t npix = q->d_viewport ? (size_t)q->width * (size_t)std::max(q->buffer_rows, 1) : 0;
    m.device = q->device; m.rows = q->buffer_rows;
    m.viewport_bytes = 16 * npix; m.image_bytes = 16 * npix; m.hit_bytes = q->d_hits ? 32 * npix : 0;
    m.octree_bytes = q->d_desc ? q->n_desc * 8 + (q->d_attach_lookup ? q->n_desc * 4 + std::max<uint64_t>(q->n_attach, 1) * 8 : 0) : 0;
    m.octree_shared = q->owns_desc ? 0 : 1;
    m.peer_access = q->peer_access;
    if (const vrc_tree *t = q->tree.get()) {
        m.tree_holders = (int32_t)q->tree.use_count();
        m.coarse_log2 = t->d_coarse ? t->coarse_log2 : 0;
        m.coarse_bytes = t->d_coarse ? (uint64_t)sizeof(uint64_t) << (3 * t->coarse_log2) : 0;
        m.empty_boxes = q->last_frame_boxes ? 1 : 0;
        m.box_bytes = (t->d_boxes ? (uint64_t)t->box_records * (32 + (t->box_mode == 2 ? 20 : 0)) : 0) + (t->d_box_aux ? (uint64_t)sizeof(uint32_t) << (3 * t->box_log2) : 0);
        m.box_records = t->d_boxes ? t->box_records : 0;
        m.box_levels = t->d_boxes ? t->box_levels : 0;
        m.box_build_seconds = t->box_build_seconds;
        snprintf(m.note, sizeof(m.note), "%s%s", t->coarse_note.c_str(), t->box_note.c_str());
        m.box_queries_cut = t->d_boxes ? t->box_queries_cut : 0;
    }
    const uint32_t n = std::min<uint32_t>(out->struct_size, (uint32_t)sizeof(m));
    m.struct_size = n;
    memcpy(out, &m, n);
    return VRC_OK;
}

int vrc_memory_usage(vrc_caster *h, int32_t rank, vrc_memory *out) {
    if (!h || !out || rank < 0 || rank > (int32_t)h->peers.size()) return VRC_ERR_INVALID_ARGUMENT;
    const vrc_caster *q = rank == 0 ? h : h->peers[rank - 1];
    memset(out, 0, sizeof(*out));
    const size_t npix = q->d_viewport ? (size_t)q->width * (size_t)std::max(q->buffer_rows, 1) : 0;
    out->device = q->device;
    out->rows = q->buffer_rows;
    out->viewport_bytes = 16 * npix;
    out->image_bytes = 16 * npix;
    out->hit_bytes = q->d_hits ? 32 * npix : 0;
    out->octree_bytes = q->d_desc ? q->n_desc * 8 + (q->d_attach_lookup ? q->n_desc * 4 + std::max<uint64_t>(q->n_attach, 1) * 8 : 0) : 0;
    out->octree_shared = q->owns_desc ? 0 : 1;
    out->peer_access = q->peer_access;
    const vrc_tree *t = q->tree.get();
    out->coarse_bytes = (t && t->d_coarse) ? (uint64_t)sizeof(uint64_t) << (3 * t->coarse_log2) : 0;
    return VRC_OK;
}

}  // extern "C"

namespace {
int counters_one(vrc_caster *h, unsigned long long c[vrc::kCtrCount]) {
    if (h->frames_enqueued && h->last_blocks == 0) {             // a rank that owns no rows of this frame (more ranks than bands)
        memset(c, 0, sizeof(unsigned long long) * vrc::kCtrCount);
        return VRC_OK;
    }
    if (!h->d_partials || h->last_blocks <= 0) return fail(h, VRC_ERR_NOT_READY, "get_counters: no frame computed");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, vrc::launch_reduce_counters(h->d_partials, h->last_blocks, h->d_counters, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(c, h->d_counters, sizeof(unsigned long long) * vrc::kCtrCount, hipMemcpyDeviceToHost));
    return VRC_OK;
}
}  // namespace

extern "C" {

int vrc_get_counters(vrc_caster *h, vrc_counters *out) {
    if (!h || !out) return VRC_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    unsigned long long c[vrc::kCtrCount], t[vrc::kCtrCount];
    int rc = counters_one(h, c);
    if (rc != VRC_OK) return rc;
    for (size_t i = 0; i < h->peers.size(); i++) {                 // a group reports the whole frame
        rc = counters_one(h->peers[i], t);
        if (rc != VRC_OK) return fail(h, rc, "rank %zu: %s", i + 1, h->peers[i]->error.c_str());
        for (int k = 0; k < vrc::kCtrCount; k++) c[k] += t[k];
    }
    if (!h->peers.empty()) HIP_TRY(h, hipSetDevice(h->device));
    out->primary_rays = c[vrc::kCtrPrimary]; out->shadow_rays = c[vrc::kCtrShadow];
    out->descriptor_reads = c[vrc::kCtrDesc]; out->texel_reads = c[vrc::kCtrTex];
    out->map_reads = c[vrc::kCtrMap]; out->steps = c[vrc::kCtrSteps];
    out->unwritten_pixels = c[vrc::kCtrUnwritten];
    out->watchdog_trips = c[vrc::kCtrWatchdog];
    for (int i = 0; i < 8; i++) h->sched_stats[i] = c[8 + i];
    if (out->watchdog_trips)
        return fail(h, VRC_ERR_DEVICE, "the kernel's round watchdog stopped %llu wavefronts: the frame is invalid",
                    (unsigned long long)out->watchdog_trips);
    return VRC_OK;
}

int vrc_counters_canonical(vrc_caster *h, int32_t *canonical) {
    if (!h || !canonical) return VRC_ERR_INVALID_ARGUMENT;
    bool boxes = h->last_frame_boxes;
    for (const vrc_caster *q : h->peers) boxes = boxes || q->last_frame_boxes;
    *canonical = boxes ? 0 : 1;
    return VRC_OK;
}

int vrc_get_scheduler_stats(vrc_caster *h, uint64_t out[8]) {
    if (!h || !out) return VRC_ERR_INVALID_ARGUMENT;
    vrc_counters tmp;
    int rc = vrc_get_counters(h, &tmp);
    if (rc != VRC_OK) return rc;
    for (int i = 0; i < 8; i++) out[i] = h->sched_stats[i];
    return VRC_OK;
}

int vrc_timing_reset(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    int rc = vrc_sync(h);
    h->timed_launches = 0; h->timed_ms = 0.0;
    for (vrc_caster *q : h->peers) { q->timed_launches = 0; q->timed_ms = 0.0; }
    return rc;
}

// a group reports rank 0's launch count and the largest per-rank kernel time (the critical path of the frame)
int vrc_timing_get(vrc_caster *h, uint64_t *n_launches, double *total_kernel_ms) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    int rc = vrc_sync(h);
    double ms = h->timed_ms;
    for (vrc_caster *q : h->peers) ms = std::max(ms, q->timed_ms);
    if (n_launches) *n_launches = h->timed_launches;
    if (total_kernel_ms) *total_kernel_ms = ms;
    return rc;
}

}  // extern "C"
