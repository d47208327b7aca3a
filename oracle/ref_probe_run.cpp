// ref_probe_run.cpp -- TEST INFRASTRUCTURE.  Loads oracle/_ref/ref_probe_gfx950.co (the reference's own
// kernels/ray_caster_kernel.cl compiled unmodified + the probe kernels of oracle/ref_probe.cl) with the HIP module
// API and runs the probes on the MI355X.  Used only by tests/test_reference_pin_gpu.py.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace {
std::string g_error;
int fail(const char *what, hipError_t e) {
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(#call, e_); } while (0)

struct Module {
    hipModule_t mod = nullptr;
    int load(const char *path) {
        TRY(hipSetDevice(0));
        TRY(hipModuleLoad(&mod, path));
        return 0;
    }
    ~Module() { if (mod) (void)hipModuleUnload(mod); }
};
template <class T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" {

const char *ref_probe_last_error() { return g_error.c_str(); }

// in14: 14 floats per case, mask3: 3 ints per case, out4: 4 floats per case
int ref_probe_view_light(const char *code_object, const float *in14, const int32_t *mask3, float *out4, int32_t n) {
    Module m;
    if (m.load(code_object)) return 1;
    hipFunction_t f;
    TRY(hipModuleGetFunction(&f, m.mod, "probe_view_light"));
    DevBuf<float> din, dout;
    DevBuf<int32_t> dmask;
    TRY(hipMalloc((void **)&din.p, sizeof(float) * 14 * n));
    TRY(hipMalloc((void **)&dmask.p, sizeof(int32_t) * 3 * n));
    TRY(hipMalloc((void **)&dout.p, sizeof(float) * 4 * n));
    TRY(hipMemcpy(din.p, in14, sizeof(float) * 14 * n, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dmask.p, mask3, sizeof(int32_t) * 3 * n, hipMemcpyHostToDevice));
    TRY(hipMemset(dout.p, 0xff, sizeof(float) * 4 * n));
    void *args[] = {&din.p, &dmask.p, &dout.p, &n};
    TRY(hipModuleLaunchKernel(f, (unsigned)((n + 63) / 64), 1, 1, 64, 1, 1, 0, nullptr, args, nullptr));
    TRY(hipDeviceSynchronize());
    TRY(hipMemcpy(out4, dout.p, sizeof(float) * 4 * n, hipMemcpyDeviceToHost));
    return 0;
}

// positions: 3 ints per case; out: 40 ints per case; settings slots as the reference application assigns them:
// OCTDIM = 0, OCTENABLED = 1, OCTREE_ROOT_INDEX = 2 (Application.cpp:35-39, CLCaster.cpp:113)
int ref_probe_get_oct_vox(const char *code_object, const int32_t *positions, int32_t n, const uint64_t *descriptors,
                          uint64_t n_descriptors, uint64_t root_index, int64_t dim, int32_t *out) {
    Module m;
    if (m.load(code_object)) return 1;
    hipFunction_t f;
    TRY(hipModuleGetFunction(&f, m.mod, "probe_get_oct_vox"));
    DevBuf<int32_t> dpos, dout;
    DevBuf<uint64_t> ddesc, dattach, dsettings;
    DevBuf<uint32_t> dlookup;
    uint64_t settings[64];
    memset(settings, 0, sizeof(settings));
    settings[0] = (uint64_t)dim; settings[1] = 0; settings[2] = root_index;
    TRY(hipMalloc((void **)&dpos.p, sizeof(int32_t) * 3 * n));
    TRY(hipMalloc((void **)&dout.p, sizeof(int32_t) * 40 * n));
    TRY(hipMalloc((void **)&ddesc.p, sizeof(uint64_t) * n_descriptors));
    TRY(hipMalloc((void **)&dattach.p, sizeof(uint64_t) * 8));
    TRY(hipMalloc((void **)&dlookup.p, sizeof(uint32_t) * 8));
    TRY(hipMalloc((void **)&dsettings.p, sizeof(settings)));
    TRY(hipMemcpy(dpos.p, positions, sizeof(int32_t) * 3 * n, hipMemcpyHostToDevice));
    TRY(hipMemcpy(ddesc.p, descriptors, sizeof(uint64_t) * n_descriptors, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dsettings.p, settings, sizeof(settings), hipMemcpyHostToDevice));
    TRY(hipMemset(dout.p, 0, sizeof(int32_t) * 40 * n));
    void *args[] = {&dpos.p, &ddesc.p, &dlookup.p, &dattach.p, &dsettings.p, &dout.p, &n};
    TRY(hipModuleLaunchKernel(f, (unsigned)((n + 63) / 64), 1, 1, 64, 1, 1, 0, nullptr, args, nullptr));
    TRY(hipDeviceSynchronize());
    TRY(hipMemcpy(out, dout.p, sizeof(int32_t) * 40 * n, hipMemcpyDeviceToHost));
    return 0;
}

// The reference's own `raycaster` kernel (oracle/ref_raycaster_probe.cl: image builtins redirected to buffers).
// Buffers as CLCaster binds them (CLCaster.cpp:186-202); records: 32 ints per pixel (layout in the .cl file), zeroed
// first; trig_out: sin/cos of the camera angles as this code object evaluates them.  width and height must be
// multiples of 8 (the kernel has no bounds check: the reference launches exactly W x H work-items).
int ref_probe_raycaster(const char *code_object, const int8_t *map, const int32_t map_dim[3], int32_t width, int32_t height,
                        const float *viewport_matrix, const float cam_dir[2], const float cam_pos[3], const float *lights10,
                        int32_t n_lights, const uint32_t *atlas_rgba8, const int32_t atlas_dim[2], const int32_t tile_dim[2],
                        const uint64_t *descriptors, uint64_t n_descriptors, uint64_t root_index, int64_t octree_dim,
                        int64_t oct_enabled, int32_t *records, float *trig_out) {
    if (width % 8 || height % 8) { g_error = "width and height must be multiples of 8"; return 1; }
    Module m;
    if (m.load(code_object)) return 1;
    hipFunction_t f, ftrig;
    TRY(hipModuleGetFunction(&f, m.mod, "raycaster"));
    TRY(hipModuleGetFunction(&ftrig, m.mod, "probe_trig"));
    const size_t npix = (size_t)width * height, nmap = (size_t)map_dim[0] * map_dim[1] * map_dim[2];
    const size_t natlas = (size_t)atlas_dim[0] * atlas_dim[1];
    DevBuf<int8_t> dmap;
    DevBuf<int32_t> dmapdim, dres, dlc, datlasdim, dtiledim, drec;
    DevBuf<float> dvm, dcamdir, dcampos, dlights, dtrig;
    DevBuf<uint32_t> datlas;
    DevBuf<uint64_t> ddesc, dsettings, ddummy;
    uint64_t settings[64];
    memset(settings, 0, sizeof(settings));
    settings[0] = (uint64_t)octree_dim; settings[1] = (uint64_t)oct_enabled; settings[2] = root_index;
    const int32_t mapdim4[4] = {map_dim[0], map_dim[1], map_dim[2], 0}, res2[2] = {width, height};
    const float camdir4[4] = {cam_dir[0], cam_dir[1], 0, 0}, campos4[4] = {cam_pos[0], cam_pos[1], cam_pos[2], 0};
    TRY(hipMalloc((void **)&dmap.p, nmap));
    TRY(hipMalloc((void **)&dmapdim.p, 16)); TRY(hipMalloc((void **)&dres.p, 8)); TRY(hipMalloc((void **)&dlc.p, 8));
    TRY(hipMalloc((void **)&datlasdim.p, 8)); TRY(hipMalloc((void **)&dtiledim.p, 8));
    TRY(hipMalloc((void **)&drec.p, npix * 32 * sizeof(int32_t)));
    TRY(hipMalloc((void **)&dvm.p, npix * 16)); TRY(hipMalloc((void **)&dcamdir.p, 16)); TRY(hipMalloc((void **)&dcampos.p, 16));
    TRY(hipMalloc((void **)&dlights.p, sizeof(float) * 10 * 8)); TRY(hipMalloc((void **)&dtrig.p, 16));
    TRY(hipMalloc((void **)&datlas.p, natlas * 4));
    TRY(hipMalloc((void **)&ddesc.p, n_descriptors * 8)); TRY(hipMalloc((void **)&dsettings.p, sizeof(settings)));
    TRY(hipMalloc((void **)&ddummy.p, 64));
    TRY(hipMemcpy(dmap.p, map, nmap, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dmapdim.p, mapdim4, 16, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dres.p, res2, 8, hipMemcpyHostToDevice));
    int32_t lc2[2] = {n_lights, 0};
    TRY(hipMemcpy(dlc.p, lc2, 8, hipMemcpyHostToDevice));
    TRY(hipMemcpy(datlasdim.p, atlas_dim, 8, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dtiledim.p, tile_dim, 8, hipMemcpyHostToDevice));
    TRY(hipMemset(drec.p, 0, npix * 32 * sizeof(int32_t)));
    TRY(hipMemcpy(dvm.p, viewport_matrix, npix * 16, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dcamdir.p, camdir4, 16, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dcampos.p, campos4, 16, hipMemcpyHostToDevice));
    TRY(hipMemset(dlights.p, 0, sizeof(float) * 80));
    TRY(hipMemcpy(dlights.p, lights10, sizeof(float) * 10 * (size_t)(n_lights < 8 ? n_lights : 8), hipMemcpyHostToDevice));
    TRY(hipMemcpy(datlas.p, atlas_rgba8, natlas * 4, hipMemcpyHostToDevice));
    TRY(hipMemcpy(ddesc.p, descriptors, n_descriptors * 8, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dsettings.p, settings, sizeof(settings), hipMemcpyHostToDevice));
    TRY(hipMemset(ddummy.p, 0, 64));
    {
        void *args[] = {&dcamdir.p, &dtrig.p};
        TRY(hipModuleLaunchKernel(ftrig, 1, 1, 1, 1, 1, 1, 0, nullptr, args, nullptr));
    }
    // the 16 kernel arguments in the reference's order; the two image2d_t arguments are dead after the redirection
    // and get a null descriptor pointer
    void *null_image = nullptr;
    void *args[] = {&dmap.p, &dmapdim.p, &dres.p, &dvm.p, &dcamdir.p, &dcampos.p, &dlights.p, &dlc.p, &null_image, &null_image,
                    &datlasdim.p, &dtiledim.p, &ddesc.p, &datlas.p, &drec.p, &dsettings.p};
    TRY(hipModuleLaunchKernel(f, (unsigned)(width / 8), (unsigned)(height / 8), 1, 8, 8, 1, 0, nullptr, args, nullptr));
    TRY(hipDeviceSynchronize());
    TRY(hipMemcpy(records, drec.p, npix * 32 * sizeof(int32_t), hipMemcpyDeviceToHost));
    TRY(hipMemcpy(trig_out, dtrig.p, 16, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
