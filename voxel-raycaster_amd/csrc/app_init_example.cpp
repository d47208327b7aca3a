// app_init_example.cpp -- the reference application's boundary call sequence
// (Application::init_clcaster, src/Application.cpp:27-88, and one game_loop
// compute(), :152) replayed against the MI355X library through the C++ mirror.
// Writes the frame as raw float4 + hit records so a test can compare it with
// the oracle.   usage: app_init_example <width> <height> <atlas.rgba> <out_prefix>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>

#include "clcaster.hpp"

using namespace vrc_host;

int main(int argc, char **argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s W H atlas.rgba out_prefix\n", argv[0]); return 2; }
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]);
    const int MAP = 16;                                              // Application.cpp:3-5

    auto raycaster = std::make_shared<CLCaster>();
    if (!raycaster->init()) { std::fprintf(stderr, "init failed\n"); return 1; }   // abort() in the reference (:32-33)

    raycaster->add_to_settings_buffer("octree_dimensions", "OCTDIM", (int)MAP);    // :35
    int oct_enabled = 0;                                                            // :38-39 (0 = octree occupancy)
    raycaster->add_to_settings_buffer("using_octree", "OCTENABLED", oct_enabled);

    auto map = std::make_shared<Map>(MAP);                                          // :42 Map(16): all 5s + octree
    if (!raycaster->assign_octree(*map)) return 1;                                  // :44
    if (!raycaster->assign_map(*map)) return 1;                                     // :45

    Camera camera{{2.424f, 3.141f}, {2.34f, 2.5f, 7.17f}};                          // :54-58 (dir, pos)
    if (!raycaster->assign_camera(&camera)) return 1;
    if (!raycaster->create_viewport(W, H, 0.625f * 90.0f, 90.0f)) return 1;         // :63 (fov ignored there too)

    std::vector<PackedData> lights(8);                                              // LightController.cpp:3-11: 8 slots
    lights[0] = PackedData{{0.01f, 0.01f, 0.01f, 0.2f}, {10.f, 10.f, 10.f}, {-1.f, -1.f, -1.5f}};   // :69-74
    if (!raycaster->assign_lights(&lights)) return 1;

    std::vector<uint8_t> atlas((size_t)256 * 256 * 4);
    FILE *f = std::fopen(argv[3], "rb");
    if (!f || std::fread(atlas.data(), 1, atlas.size(), f) != atlas.size()) { std::fprintf(stderr, "atlas?\n"); return 2; }
    std::fclose(f);
    if (!raycaster->create_texture_atlas(atlas.data(), 256, 256, 16, 16)) return 1; // :77-79

    if (!raycaster->validate()) { std::fprintf(stderr, "validate: %s\n", raycaster->last_error().c_str()); return 1; }   // :83-85
    if (!raycaster->compute()) { std::fprintf(stderr, "compute: %s\n", raycaster->last_error().c_str()); return 1; }     // :152

    std::vector<float> img;
    std::vector<int32_t> hits;
    if (!raycaster->read_image(img) || !raycaster->read_hits(hits)) return 1;
    std::string p = argv[4];
    f = std::fopen((p + ".image.f32").c_str(), "wb"); std::fwrite(img.data(), 4, img.size(), f); std::fclose(f);
    f = std::fopen((p + ".hits.i32").c_str(), "wb"); std::fwrite(hits.data(), 4, hits.size(), f); std::fclose(f);

    // live camera: the reference mutates the USE_HOST_PTR floats between frames (:147-152)
    camera.position[2] += 0.5f;
    if (!raycaster->compute()) return 1;
    if (!raycaster->read_image(img)) return 1;
    f = std::fopen((p + ".image2.f32").c_str(), "wb"); std::fwrite(img.data(), 4, img.size(), f); std::fclose(f);
    // extension: the octree generated on the device from the map that is already there (Octree::Generate without the host
    // pass and without the 100 000-entry buffer): the frame does not change
    if (!raycaster->generate_octree_from_assigned_map(*map) || !raycaster->validate() || !raycaster->compute()) {
        std::fprintf(stderr, "device octree: %s\n", raycaster->last_error().c_str());
        return 1;
    }
    std::vector<float> img3;
    if (!raycaster->read_image(img3) || img3.size() != img.size() || std::memcmp(img3.data(), img.data(), img.size() * 4) != 0) {
        std::fprintf(stderr, "the frame of the device-built octree differs\n");
        return 1;
    }
    std::printf("ok %dx%d\n", W, H);
    return 0;
}
