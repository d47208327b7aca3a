// clcaster.hpp -- header-only C++ host mirror of the reference's CLCaster
// (include/CLCaster.h:93-329) over the C ABI of libvrc.so (include/vrc.h).
//
// A C++ host that used the reference class keeps its call sites: same method
// names, same argument meaning, every method returns bool (true = ok) like the
// reference, details through last_error() instead of Logger::log.  What
// changed is what sits behind it: hand-written gfx950 kernels and an offline
// pixel buffer instead of OpenCL + cl_khr_gl_sharing.  SFML types at the
// boundary (sf::Vector*, sf::Texture, shared_ptr<Map/Camera>) become plain
// structs/pointers; see INTEGRATION.md for the adapter a maintainer would add.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/vrc.h"

namespace vrc_host {

// include/LightController.h:63-73
struct PackedData {
    float rgbi[4];
    float position[3];
    float direction_cartesian[3];
};

// src/map/Map.cpp:5-19 + src/map/Octree.cpp:13-43: dense ArrayMap + its Octree
struct Map {
    uint32_t dimensions = 0;
    std::vector<int8_t> array_map;          // x + dim*(y + dim*z)
    uint64_t *descriptor_buffer = nullptr;  // Octree::descriptor_buffer
    uint64_t buffer_size = 0;               // Octree::buffer_size
    uint64_t root_index = 0;                // Octree::root_index

    explicit Map(uint32_t dim, uint64_t octree_buffer_size = 100000 /* Octree.h:29 */, bool strict_reference = true)
        : dimensions(dim), array_map((size_t)dim * dim * dim, 5) {   // ArrayMap ctor fills with 5 (ArrayMap.cpp:17-23)
        regenerate(octree_buffer_size, strict_reference);
    }
    ~Map() { vrc_free(descriptor_buffer); }
    Map(const Map &) = delete;
    Map &operator=(const Map &) = delete;

    void setVoxel(int x, int y, int z, int8_t v) { array_map[(size_t)x + dimensions * ((size_t)y + (size_t)dimensions * z)] = v; }
    bool regenerate(uint64_t octree_buffer_size = 100000, bool strict_reference = true) {
        vrc_free(descriptor_buffer);
        descriptor_buffer = nullptr;
        return vrc_octree_generate(array_map.data(), dimensions, octree_buffer_size, strict_reference ? 1 : 0,
                                   &descriptor_buffer, &buffer_size, &root_index) == VRC_OK;
    }
};

// include/Camera.h:75-81: the two vectors CLCaster maps USE_HOST_PTR
struct Camera {
    float direction[2];   // inclination, azimuth
    float position[3];
    const float *get_direction_pointer() const { return direction; }   // Camera.cpp:302-312
    const float *get_position_pointer() const { return position; }
};

class CLCaster {
public:
    CLCaster() = default;
    ~CLCaster() { if (h_) vrc_destroy(h_); }
    CLCaster(const CLCaster &) = delete;
    CLCaster &operator=(const CLCaster &) = delete;

    bool init(int device_ordinal = 0) { return ok(vrc_create(device_ordinal, &h_)); }                    // CLCaster.cpp:14-74
    // the same object driving several GPUs from this one host thread: the frame is row-sliced over the devices, the
    // scene replicated, compute() returns when every GPU is done, read_image*() gathers the slices (vrc_create_group)
    bool init(const std::vector<int32_t> &device_ordinals, int band_rows = 8) {
        return ok(vrc_create_group(device_ordinals.data(), (int32_t)device_ordinals.size(), band_rows, &h_));
    }

    bool assign_map(const Map &m) {                                                                        // :76-87
        return ok(vrc_assign_map(h_, m.array_map.data(), (int32_t)m.dimensions, (int32_t)m.dimensions, (int32_t)m.dimensions));
    }
    bool release_map() { return ok(vrc_release_map(h_)); }                                                // :89-99
    bool assign_octree(const Map &m) {                                                                     // :102-116
        return ok(vrc_assign_octree(h_, m.descriptor_buffer, m.buffer_size, m.root_index));
    }
    // extension: Octree::Generate (src/map/Octree.cpp:13-43) on the device, from the map assign_map() has uploaded -- the
    // host skips Map::regenerate and the descriptor upload, and the map's materials come along as attachments (dimensions
    // must be a power of two, 8..4096)
    bool generate_octree_from_assigned_map(const Map &m) {
        uint32_t depth = 0;
        while ((1u << depth) < m.dimensions) depth++;
        return (1u << depth) == m.dimensions && ok(vrc_build_dense_grid(h_, depth, nullptr, VRC_BUILD_ATTACHMENTS, 0, nullptr));
    }
    // extension: the tree another CLCaster on the same GPU already holds (two viewports of one Map, two frames in flight):
    // adopted, not uploaded again -- one descriptor array, one coarse table, one set of empty boxes between the two
    bool assign_octree_from(CLCaster &other) { return ok(vrc_assign_octree_from(h_, other.h_)); }
    bool release_octree() { return ok(vrc_release_octree(h_)); }                                          // :119-131
    bool assign_camera(const Camera *camera) {                                                             // :133-143
        return ok(vrc_assign_camera(h_, camera->get_direction_pointer(), camera->get_position_pointer()));
    }
    // extension: the sin / cos of the camera's two angles as THIS host evaluates them (the reference kernel evaluates them per
    // pixel with the OpenCL library's, ray_caster_kernel.cl:280-291); four live floats, nullptr = the library's sinf / cosf
    bool assign_camera_trig(const float *trig4) { return ok(vrc_assign_camera_trig(h_, trig4)); }
    bool release_camera() { return ok(vrc_release_camera(h_)); }                                          // :145-155
    bool assign_lights(std::vector<PackedData> *data) {                                                    // :313-328
        light_count_ = (int32_t)data->size();
        return ok(vrc_assign_lights(h_, reinterpret_cast<const float *>(data->data()), &light_count_));
    }
    bool create_viewport(int width, int height, float v_fov, float h_fov) {                               // :233-299
        width_ = width; height_ = height;
        return ok(vrc_create_viewport(h_, width, height, v_fov, h_fov));
    }
    bool release_viewport() { return ok(vrc_release_viewport(h_)); }                                      // :301-311
    bool create_texture_atlas(const uint8_t *rgba8, int width, int height, int tile_w, int tile_h) {      // :208-222
        return ok(vrc_create_texture_atlas(h_, rgba8, width, height, tile_w, tile_h));
    }

    template <typename T>
    bool add_to_settings_buffer(const std::string &setting_name, const std::string &define_accessor_name, T value) {   // :1029-1064
        return ok(vrc_setting_add(h_, setting_name.c_str(), define_accessor_name.c_str(), (int64_t)value));
    }
    bool overwrite_setting(const std::string &setting_name, int64_t *value) {                              // :1087-1109
        return ok(vrc_setting_set(h_, setting_name.c_str(), *value));
    }

    bool validate() { return ok(vrc_validate(h_)); }                                                       // :157-206 (ends with prepare())
    bool prepare() { return ok(vrc_prepare(h_)); }        // the tree's derived structures for settings changed after validate()
    bool compute() { return ok(vrc_compute(h_)); }                                                         // :224-228

    // replaces draw(sf::RenderWindow*) (:330-332): the frame as float4 / RGBA8
    bool read_image(std::vector<float> &rgba) {
        rgba.resize((size_t)4 * width_ * height_);
        return ok(vrc_read_image_f32(h_, rgba.data(), rgba.size()));
    }
    bool read_image_rgba8(std::vector<uint8_t> &rgba) {
        rgba.resize((size_t)4 * width_ * height_);
        return ok(vrc_read_image_rgba8(h_, rgba.data(), rgba.size()));
    }
    bool read_hits(std::vector<int32_t> &hits) {
        hits.resize((size_t)8 * width_ * height_);
        return ok(vrc_read_hits(h_, hits.data(), hits.size()));
    }

    int last_status() const { return status_; }
    std::string last_error() const { return h_ ? vrc_last_error(h_) : "not initialised"; }
    vrc_caster *handle() { return h_; }

private:
    bool ok(int rc) { status_ = rc; return rc == VRC_OK; }
    vrc_caster *h_ = nullptr;
    int status_ = 0;
    int width_ = 0, height_ = 0;
    int32_t light_count_ = 0;
};

}  // namespace vrc_host
