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

}  // extern "C"
