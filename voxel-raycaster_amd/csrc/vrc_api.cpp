// vrc_api.cpp -- C-ABI host layer of libvrc.so (see include/vrc.h).
//
// Plays the role of src/CLCaster.cpp in the reference: owns the device
// buffers (its named buffer_map, :855-944), the settings buffer (:1029-1109),
// validation (:157-206) and the per-frame launch (:224-228, 946-987) -- with
// HIP streams/events on one MI355X instead of an OpenCL queue + GL interop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vrc.h"
#include "vrc_params.h"

namespace vrc {
hipError_t launch_raycast(const RaycastParams &p, hipStream_t stream);
hipError_t launch_frame_setup(const RaycastParams &p, hipStream_t stream);
hipError_t launch_reduce_counters(const unsigned long long *partials, int nblocks, unsigned long long *out,
                                  hipStream_t stream);
}  // namespace vrc

struct vrc_setting { std::string name, define; int64_t value; };

struct vrc_caster {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;

    // scene buffers (device)
    int8_t *d_map = nullptr; int32_t map_dim[3] = {0, 0, 0};
    uint64_t *d_desc = nullptr; uint64_t n_desc = 0; bool have_octree = false;
    uint32_t *d_attach_lookup = nullptr; uint64_t *d_attach = nullptr;
    float *d_viewport = nullptr; float *d_image = nullptr; int32_t *d_hits = nullptr;
    int32_t width = 0, height = 0;
    uint8_t *d_atlas = nullptr; int32_t atlas_w = 0, atlas_h = 0, tile_w = 0, tile_h = 0;
    unsigned long long *d_partials = nullptr; int partial_blocks = 0;
    unsigned long long *d_counters = nullptr;
    int32_t *d_frame = nullptr;           // {bias[3], reads}

    // live (retained) host pointers
    const float *cam_dir = nullptr, *cam_pos = nullptr;
    const float *lights = nullptr; const int32_t *light_count = nullptr;

    std::vector<vrc_setting> settings;    // <= 64 slots (include/CLCaster.h:303)

    int32_t tile_rank = 0, tile_world = 1, band_rows = 8;
    bool validated = false;
    int last_blocks = 0;
    uint64_t sched_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    // kernel timing
    struct EvPair { hipEvent_t a, b; };
    std::vector<EvPair> pending, pool;
    uint64_t timed_launches = 0; double timed_ms = 0.0;
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
        if (e_ != hipSuccess)                                                                     \
            return fail(h, e_ == hipErrorOutOfMemory ? VRC_ERR_OUT_OF_MEMORY : VRC_ERR_DEVICE,    \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <class T>
void release(T *&p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

int find_setting(const vrc_caster *h, const char *name) {
    for (size_t i = 0; i < h->settings.size(); i++)
        if (h->settings[i].name == name) return (int)i;
    return -1;
}

int64_t setting_or(const vrc_caster *h, const char *name, int64_t dflt) {
    int i = find_setting(h, name);
    return i < 0 ? dflt : h->settings[i].value;
}

int set_setting(vrc_caster *h, const char *name, const char *define, int64_t value) {
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
        hipMalloc((void **)&h->d_frame, sizeof(int32_t) * 4) != hipSuccess) {
        delete h;
        return VRC_ERR_DEVICE;
    }
    (void)hipMemset(h->d_counters, 0, sizeof(unsigned long long) * vrc::kCtrCount);
    (void)hipMemset(h->d_frame, 0, sizeof(int32_t) * 4);
    *out = h;
    return VRC_OK;
}

int vrc_destroy(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drain_events(h);
    for (auto &p : h->pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    release(h->d_map); release(h->d_desc); release(h->d_attach_lookup); release(h->d_attach);
    release(h->d_viewport); release(h->d_image); release(h->d_hits); release(h->d_atlas);
    release(h->d_partials); release(h->d_counters); release(h->d_frame);
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
    return VRC_OK;
}

int vrc_release_map(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_map) return fail(h, VRC_ERR_NOT_FOUND, "release_map: no map assigned");
    release(h->d_map);
    h->map_dim[0] = h->map_dim[1] = h->map_dim[2] = 0;
    h->validated = false;
    return VRC_OK;
}

int vrc_assign_octree(vrc_caster *h, const uint64_t *descriptors, uint64_t n, uint64_t root_index) {
    if (!h || !descriptors || n == 0 || root_index >= n) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    release(h->d_desc);
    h->have_octree = false;
    HIP_TRY(h, hipMalloc((void **)&h->d_desc, n * sizeof(uint64_t)));
    HIP_TRY(h, hipMemcpy(h->d_desc, descriptors, n * sizeof(uint64_t), hipMemcpyHostToDevice));
    h->n_desc = n;
    h->have_octree = true;
    h->validated = false;
    return set_setting(h, "octree_root_index", "OCTREE_ROOT_INDEX", (int64_t)root_index);   // CLCaster.cpp:113
}

int vrc_assign_octree_attachments(vrc_caster *h, const uint32_t *lookup, uint64_t n_lookup,
                                  const uint64_t *attachments, uint64_t n_attachments) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    release(h->d_attach_lookup); release(h->d_attach);
    if (lookup && n_lookup && n_lookup != h->n_desc)
        return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_attachments: lookup must have one entry per descriptor (assign the octree first)");
    if (lookup && n_lookup) {
        HIP_TRY(h, hipMalloc((void **)&h->d_attach_lookup, n_lookup * sizeof(uint32_t)));
        HIP_TRY(h, hipMemcpy(h->d_attach_lookup, lookup, n_lookup * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (attachments && n_attachments) {
        HIP_TRY(h, hipMalloc((void **)&h->d_attach, n_attachments * sizeof(uint64_t)));
        HIP_TRY(h, hipMemcpy(h->d_attach, attachments, n_attachments * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    return VRC_OK;
}

namespace {
// file -> device through two pinned staging buffers: the read of chunk k+1 overlaps the copy of chunk k
int stream_to_device(vrc_caster *h, FILE *f, void *dst, size_t bytes, void *stage[2], size_t chunk) {
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) HIP_TRY(h, hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    int rc = VRC_OK;
    size_t off = 0;
    for (int k = 0; off < bytes; k ^= 1) {
        const size_t n = std::min(chunk, bytes - off);
        if (hipEventSynchronize(done[k]) != hipSuccess) { rc = fail(h, VRC_ERR_DEVICE, "assign_octree_file: event wait failed"); break; }
        if (fread(stage[k], 1, n, f) != n) { rc = fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_octree_file: file is truncated"); break; }
        if (hipMemcpyAsync((char *)dst + off, stage[k], n, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipEventRecord(done[k], h->stream) != hipSuccess) { rc = fail(h, VRC_ERR_DEVICE, "assign_octree_file: upload failed"); break; }
        off += n;
    }
    (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 2; i++) (void)hipEventDestroy(done[i]);
    return rc;
}
}  // namespace

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
    release(h->d_desc); release(h->d_attach_lookup); release(h->d_attach);
    h->have_octree = false; h->n_desc = 0; h->validated = false;
    const size_t chunk = (size_t)64 << 20;
    void *stage[2] = {nullptr, nullptr};
    int rc = VRC_OK;
    if (hipHostMalloc(&stage[0], chunk, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&stage[1], chunk, hipHostMallocDefault) != hipSuccess)
        rc = fail(h, VRC_ERR_OUT_OF_MEMORY, "assign_octree_file: no pinned staging memory");
    if (rc == VRC_OK && hipMalloc((void **)&h->d_desc, n * sizeof(uint64_t)) != hipSuccess)
        rc = fail(h, VRC_ERR_OUT_OF_MEMORY, "assign_octree_file: %llu descriptors do not fit in device memory", (unsigned long long)n);
    if (rc == VRC_OK) rc = stream_to_device(h, f, h->d_desc, n * sizeof(uint64_t), stage, chunk);
    if (rc == VRC_OK && (flags & 1u)) {
        if (hipMalloc((void **)&h->d_attach_lookup, n * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void **)&h->d_attach, (na ? na : 1) * sizeof(uint64_t)) != hipSuccess)
            rc = fail(h, VRC_ERR_OUT_OF_MEMORY, "assign_octree_file: attachment buffers do not fit in device memory");
        if (rc == VRC_OK) rc = stream_to_device(h, f, h->d_attach_lookup, n * sizeof(uint32_t), stage, chunk);
        if (rc == VRC_OK) rc = stream_to_device(h, f, h->d_attach, na * sizeof(uint64_t), stage, chunk);
    }
    fclose(f);
    for (int i = 0; i < 2; i++) if (stage[i]) (void)hipHostFree(stage[i]);
    if (rc != VRC_OK) { release(h->d_desc); release(h->d_attach_lookup); release(h->d_attach); return rc; }
    h->n_desc = n;
    h->have_octree = true;
    return set_setting(h, "octree_root_index", "OCTREE_ROOT_INDEX", (int64_t)root);   // CLCaster.cpp:113
}

int vrc_release_octree(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_desc) return fail(h, VRC_ERR_NOT_FOUND, "release_octree: no octree assigned");
    release(h->d_desc); release(h->d_attach_lookup); release(h->d_attach);
    h->n_desc = 0; h->have_octree = false; h->validated = false;
    return VRC_OK;
}

static int install_viewport(vrc_caster *h, int32_t width, int32_t height, std::vector<float> &table) {
    HIP_TRY(h, hipSetDevice(h->device));
    release(h->d_viewport); release(h->d_image); release(h->d_hits);
    const size_t npix = (size_t)width * height;
    HIP_TRY(h, hipMalloc((void **)&h->d_viewport, 16 * npix));
    HIP_TRY(h, hipMemcpy(h->d_viewport, table.data(), 16 * npix, hipMemcpyHostToDevice));
    // image starts as RGBA8 (255,255,255,100)  (CLCaster.cpp:280-286)
    for (size_t i = 0; i < npix; i++) {
        table[4 * i + 0] = 1.0f; table[4 * i + 1] = 1.0f; table[4 * i + 2] = 1.0f; table[4 * i + 3] = 100.0f / 255.0f;
    }
    HIP_TRY(h, hipMalloc((void **)&h->d_image, 16 * npix));
    HIP_TRY(h, hipMemcpy(h->d_image, table.data(), 16 * npix, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc((void **)&h->d_hits, 32 * npix));
    HIP_TRY(h, hipMemset(h->d_hits, 0, 32 * npix));
    h->width = width; h->height = height;
    h->validated = false;
    return VRC_OK;
}

int vrc_create_viewport(vrc_caster *h, int32_t width, int32_t height, float v_fov, float h_fov) {
    (void)v_fov; (void)h_fov;              // ignored by the reference too (CLCaster.cpp:233-275)
    if (!h || width <= 0 || height <= 0) return fail(h, VRC_ERR_INVALID_ARGUMENT, "create_viewport: bad size");
    const size_t npix = (size_t)width * height;
    std::vector<float> table(4 * npix, 0.0f);
    // base ray (-800, x, y) slewed by the literal 1.57 about Y in double, then
    // normalised in float (util.hpp:64-73)
    const double s157 = std::sin(1.57), c157 = std::cos(1.57);
    for (int y = -height / 2; y < height / 2; y++)
        for (int x = -width / 2; x < width / 2; x++) {
            const float bx = -800.0f, by = (float)x, bz = (float)y;
            const float rx = (float)((double)bz * s157 + (double)bx * c157);
            const float ry = by;
            const float rz = (float)((double)bz * c157 - (double)bx * s157);
            const float len = std::sqrt(rx * rx + ry * ry + rz * rz);
            float *t = &table[4 * ((size_t)(x + width / 2) + (size_t)width * (size_t)(y + height / 2))];
            t[0] = rx / len; t[1] = ry / len; t[2] = rz / len; t[3] = 0.0f;
        }
    return install_viewport(h, width, height, table);
}

int vrc_create_viewport_table(vrc_caster *h, int32_t width, int32_t height, const float *table) {
    if (!h || !table || width <= 0 || height <= 0) return fail(h, VRC_ERR_INVALID_ARGUMENT, "create_viewport_table: bad argument");
    std::vector<float> copy(table, table + (size_t)4 * width * height);
    return install_viewport(h, width, height, copy);
}

int vrc_release_viewport(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_viewport) return fail(h, VRC_ERR_NOT_FOUND, "release_viewport: no viewport");
    release(h->d_viewport); release(h->d_image); release(h->d_hits);
    h->width = h->height = 0; h->validated = false;
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
    return VRC_OK;
}

int vrc_assign_camera(vrc_caster *h, const float *direction2, const float *position3) {
    if (!h || !direction2 || !position3) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_camera: null pointer");
    h->cam_dir = direction2; h->cam_pos = position3;
    h->validated = false;
    return VRC_OK;
}

int vrc_release_camera(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    h->cam_dir = h->cam_pos = nullptr; h->validated = false;
    return VRC_OK;
}

int vrc_assign_lights(vrc_caster *h, const float *packed, const int32_t *light_count) {
    if (!h || !packed || !light_count) return fail(h, VRC_ERR_INVALID_ARGUMENT, "assign_lights: null pointer");
    h->lights = packed; h->light_count = light_count;
    h->validated = false;
    return VRC_OK;
}

int vrc_setting_add(vrc_caster *h, const char *name, const char *define, int64_t value) {
    if (!h || !name) return VRC_ERR_INVALID_ARGUMENT;
    return set_setting(h, name, define, value);
}

int vrc_setting_set(vrc_caster *h, const char *name, int64_t value) {
    if (!h || !name) return VRC_ERR_INVALID_ARGUMENT;
    int i = find_setting(h, name);
    if (i < 0) return fail(h, VRC_ERR_NOT_FOUND, "overwrite_setting: no setting named '%s'", name);   // CLCaster.cpp:1096-1100
    h->settings[i].value = value;
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
    h->tile_rank = rank; h->tile_world = world; h->band_rows = band_rows;
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
    return VRC_OK;
}

int vrc_compute_async(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->validated) return fail(h, VRC_ERR_NOT_READY, "compute: validate() has not succeeded");
    HIP_TRY(h, hipSetDevice(h->device));

    vrc::RaycastParams p;
    memset(&p, 0, sizeof(p));
    const bool svo = setting_or(h, "using_octree", 0) == 0;
    const int64_t dim = setting_or(h, "octree_dimensions", 0);
    p.svo = svo ? 1 : 0;
    p.log2_dim = log2_exact(dim);
    p.map = h->d_map;
    if (svo) { p.map_dim[0] = p.map_dim[1] = p.map_dim[2] = (int32_t)dim; }
    else { p.map_dim[0] = h->map_dim[0]; p.map_dim[1] = h->map_dim[1]; p.map_dim[2] = h->map_dim[2]; }
    p.width = h->width; p.height = h->height;
    p.viewport = h->d_viewport; p.image = h->d_image; p.hits = h->d_hits;
    p.atlas = h->d_atlas; p.atlas_w = h->atlas_w; p.atlas_h = h->atlas_h;
    p.tiles_x = h->atlas_w / h->tile_w; p.tiles_y = h->atlas_h / h->tile_h;
    p.descriptors = h->d_desc;
    p.attach_lookup = (h->d_attach_lookup && h->d_attach) ? h->d_attach_lookup : nullptr;
    p.attachments = p.attach_lookup ? h->d_attach : nullptr;
    p.root_index = (uint64_t)setting_or(h, "octree_root_index", 0);
    // live buffers are re-read every frame (CL_MEM_USE_HOST_PTR semantics)
    for (int a = 0; a < 3; a++) p.cam_pos[a] = h->cam_pos[a];
    p.trig[0] = sinf(h->cam_dir[0]); p.trig[1] = cosf(h->cam_dir[0]);
    p.trig[2] = sinf(h->cam_dir[1]); p.trig[3] = cosf(h->cam_dir[1]);
    // the reference binds light_count but shades with light 0 only (ray_caster_kernel.cl:264,660-670); setting
    // "light_count" (default 1) switches on the multi-light extension for the first n packed lights
    p.light_count = (int32_t)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(setting_or(h, "light_count", 1), *h->light_count),
                                                                    vrc::kMaxLights));
    for (int l = 0; l < p.light_count; l++) {
        for (int k = 0; k < 7; k++) p.lights[l][k] = h->lights[10 * l + k];
        p.lights[l][7] = 0.0f;
    }
    p.max_distance = (int32_t)setting_or(h, "max_distance", 20);
    p.shadow_rays = (int32_t)setting_or(h, "shadow_rays", 1);
    // wave scheduling knobs of the SVO kernel; they never change results
    p.burst_steps = (int32_t)std::min<int64_t>(1 << 20, std::max<int64_t>(1, setting_or(h, "burst_steps", vrc::kDefaultBurstSteps)));
    p.shade_threshold = std::min<int64_t>(64, std::max<int64_t>(1, setting_or(h, "shade_threshold", vrc::kDefaultShadeThreshold)));
    p.jump_min_run = (int32_t)std::min<int64_t>(1 << 24, std::max<int64_t>(1, setting_or(h, "jump_min_run", 1 << 24)));
    p.widen_nodes = (int32_t)setting_or(h, "widen_nodes", 1);
    p.octree_bias = (int32_t)setting_or(h, "octree_bias", 1);
    p.arith_mask = (int32_t)setting_or(h, "arith_mask", 1);
    p.watchdog_rounds = (int32_t)std::min<int64_t>(INT32_MAX, std::max<int64_t>(1, setting_or(h, "watchdog_rounds",
                                    64LL * ((int64_t)p.max_distance + 64) * (p.light_count + 1))));
    p.safe_run = (int32_t)setting_or(h, "safe_run", 1);
    p.single_step = (int32_t)setting_or(h, "single_step", 1);
    p.safe_steps = (int32_t)std::min<int64_t>(256, std::max<int64_t>(2, setting_or(h, "safe_steps", vrc::kDefaultSafeSteps)));
    p.exact_steps = (int32_t)std::min<int64_t>(1 << 20, std::max<int64_t>(1, setting_or(h, "exact_steps", vrc::kDefaultExactSteps)));
    p.xcd_mode = (int32_t)setting_or(h, "xcd_mode", 1);
    p.lds_pad_bytes = (int32_t)std::min<int64_t>(120 * 1024, std::max<int64_t>(0, setting_or(h, "lds_pad_bytes", 0)));
    p.frame = h->d_frame;

    const int tile_rows = (h->height + vrc::kTileH - 1) / vrc::kTileH;
    p.band_tiles = h->band_rows / vrc::kTileH;
    p.tile_rank = h->tile_rank; p.tile_world = h->tile_world;
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
        HIP_TRY(h, hipMalloc((void **)&h->d_partials, sizeof(unsigned long long) * vrc::kCtrCount * (size_t)nblocks));
        h->partial_blocks = nblocks;
    }
    p.counters = h->d_partials;
    h->last_blocks = nblocks;

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

int vrc_sync(vrc_caster *h) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    drain_events(h);
    return VRC_OK;
}

int vrc_compute(vrc_caster *h) {
    int rc = vrc_compute_async(h);
    if (rc != VRC_OK) return rc;
    return vrc_sync(h);                    // clFinish (CLCaster.cpp:970)
}

int vrc_read_image_f32(vrc_caster *h, float *rgba, size_t n_floats) {
    if (!h || !rgba) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_image) return fail(h, VRC_ERR_NOT_READY, "read_image: no viewport");
    const size_t need = (size_t)4 * h->width * h->height;
    if (n_floats < need) return fail(h, VRC_ERR_INVALID_ARGUMENT, "read_image_f32: buffer too small");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(rgba, h->d_image, need * sizeof(float), hipMemcpyDeviceToHost));
    return VRC_OK;
}

int vrc_read_image_rgba8(vrc_caster *h, uint8_t *rgba, size_t n_bytes) {
    if (!h || !rgba) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_image) return fail(h, VRC_ERR_NOT_READY, "read_image: no viewport");
    const size_t need = (size_t)4 * h->width * h->height;
    if (n_bytes < need) return fail(h, VRC_ERR_INVALID_ARGUMENT, "read_image_rgba8: buffer too small");
    std::vector<float> tmp(need);
    int rc = vrc_read_image_f32(h, tmp.data(), need);
    if (rc != VRC_OK) return rc;
    // write_imagef to a UNORM_INT8 target: saturate, scale, round to nearest even
    for (size_t i = 0; i < need; i++) {
        float v = tmp[i];
        if (!(v > 0.0f)) v = 0.0f;
        if (v > 1.0f) v = 1.0f;
        rgba[i] = (uint8_t)lrintf(v * 255.0f);
    }
    return VRC_OK;
}

int vrc_read_hits(vrc_caster *h, int32_t *hits, size_t n_int32) {
    if (!h || !hits) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_hits) return fail(h, VRC_ERR_NOT_READY, "read_hits: no viewport");
    const size_t need = (size_t)8 * h->width * h->height;
    if (n_int32 < need) return fail(h, VRC_ERR_INVALID_ARGUMENT, "read_hits: buffer too small");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(hits, h->d_hits, need * sizeof(int32_t), hipMemcpyDeviceToHost));
    return VRC_OK;
}

int vrc_device_image(vrc_caster *h, void **dev_ptr, size_t *n_bytes) {
    if (!h || !dev_ptr) return VRC_ERR_INVALID_ARGUMENT;
    if (!h->d_image) return fail(h, VRC_ERR_NOT_READY, "device_image: no viewport");
    *dev_ptr = h->d_image;
    if (n_bytes) *n_bytes = (size_t)16 * h->width * h->height;
    return VRC_OK;
}

int vrc_get_counters(vrc_caster *h, vrc_counters *out) {
    if (!h || !out) return VRC_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    if (!h->d_partials || h->last_blocks <= 0) return fail(h, VRC_ERR_NOT_READY, "get_counters: no frame computed");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, vrc::launch_reduce_counters(h->d_partials, h->last_blocks, h->d_counters, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned long long c[vrc::kCtrCount];
    HIP_TRY(h, hipMemcpy(c, h->d_counters, sizeof(c), hipMemcpyDeviceToHost));
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
    return rc;
}

int vrc_timing_get(vrc_caster *h, uint64_t *n_launches, double *total_kernel_ms) {
    if (!h) return VRC_ERR_INVALID_ARGUMENT;
    int rc = vrc_sync(h);
    if (n_launches) *n_launches = h->timed_launches;
    if (total_kernel_ms) *total_kernel_ms = h->timed_ms;
    return rc;
}

}  // extern "C"
