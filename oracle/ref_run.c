/*
 * ref_run.c -- ORACLE PINNING TOOL (test infrastructure, not product code).
 *
 * Runs the reference's own kernel -- kernels/ray_caster_kernel.cl compiled
 * UNMODIFIED for gfx950 by oracle/ref_build.sh -- on the GPU box through the
 * AMD OpenCL runtime, binding the 16 arguments exactly as CLCaster::validate
 * does (src/CLCaster.cpp:186-202) and launching a 2-D NDRange (W,H) with a
 * NULL local size like CLCaster::run_kernel (src/CLCaster.cpp:946-987).
 * The GL-shared RGBA8 output texture is replaced by a CL_RGBA/CL_FLOAT image
 * so write_imagef's floats come back unquantised; the atlas is a
 * CL_RGBA/CL_UNORM_INT8 image like the GL texture it stands for.
 *
 * usage: ref_run <kernel.co> <scene_dir> <out_image.bin>
 * scene_dir holds raw little-endian files written by tests/make_ref_fixtures.py:
 *   params.txt  (W H dim dimx dimy dimz root using_octree atlas_w atlas_h tile_w tile_h n_desc)
 *   map.bin viewport.bin camera.bin(5 floats: dir2,pos3) lights.bin(80 floats) atlas.bin desc.bin
 */
#include <CL/cl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(err, what) do { if ((err) != CL_SUCCESS) { fprintf(stderr, "ref_run: %s failed: %d\n", what, (int)(err)); exit(2); } } while (0)

static void *slurp(const char *dir, const char *name, size_t *len) {
    char path[1024];
    snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "ref_run: cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(n > 0 ? (size_t)n : 1);
    if (n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "ref_run: short read %s\n", path); exit(2); }
    fclose(f);
    if (len) *len = (size_t)n;
    return p;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: ref_run kernel.co scene_dir out.bin\n"); return 2; }
    const char *co_path = argv[1], *dir = argv[2], *out_path = argv[3];

    long W, H, dim, dx, dy, dz, root, using_octree, aw, ah, tw, th, ndesc;
    {
        size_t n;
        char *txt = (char *)slurp(dir, "params.txt", &n);
        if (sscanf(txt, "%ld %ld %ld %ld %ld %ld %ld %ld %ld %ld %ld %ld %ld", &W, &H, &dim, &dx, &dy, &dz, &root,
                   &using_octree, &aw, &ah, &tw, &th, &ndesc) != 13) { fprintf(stderr, "ref_run: bad params.txt\n"); return 2; }
        free(txt);
    }
    size_t n_map, n_vp, n_cam, n_li, n_at, n_de, n_co;
    void *map = slurp(dir, "map.bin", &n_map);
    void *vp = slurp(dir, "viewport.bin", &n_vp);
    float *cam = (float *)slurp(dir, "camera.bin", &n_cam);
    void *li = slurp(dir, "lights.bin", &n_li);
    void *at = slurp(dir, "atlas.bin", &n_at);
    void *de = slurp(dir, "desc.bin", &n_de);
    unsigned char *co = (unsigned char *)slurp(".", co_path, &n_co);

    cl_int err;
    cl_uint np = 0;
    cl_platform_id plats[8];
    CHECK(clGetPlatformIDs(8, plats, &np), "clGetPlatformIDs");
    cl_device_id dev = NULL;
    cl_platform_id plat = NULL;
    for (cl_uint i = 0; i < np && !dev; i++) {
        cl_uint nd = 0;
        if (clGetDeviceIDs(plats[i], CL_DEVICE_TYPE_GPU, 1, &dev, &nd) == CL_SUCCESS && nd > 0) plat = plats[i];
        else dev = NULL;
    }
    if (!dev) { fprintf(stderr, "ref_run: no OpenCL GPU device\n"); return 3; }
    char name[256] = {0}, ver[256] = {0};
    clGetDeviceInfo(dev, CL_DEVICE_NAME, sizeof(name), name, NULL);
    clGetDeviceInfo(dev, CL_DRIVER_VERSION, sizeof(ver), ver, NULL);
    fprintf(stderr, "ref_run: device %s driver %s\n", name, ver);

    cl_context_properties props[] = {CL_CONTEXT_PLATFORM, (cl_context_properties)plat, 0};
    cl_context ctx = clCreateContext(props, 1, &dev, NULL, NULL, &err); CHECK(err, "clCreateContext");
    cl_command_queue q = clCreateCommandQueue(ctx, dev, 0, &err); CHECK(err, "clCreateCommandQueue");

    const unsigned char *bins[1] = {co};
    size_t lens[1] = {n_co};
    cl_int bin_status = 0;
    cl_program prog = clCreateProgramWithBinary(ctx, 1, &dev, lens, bins, &bin_status, &err);
    CHECK(err, "clCreateProgramWithBinary"); CHECK(bin_status, "binary status");
    err = clBuildProgram(prog, 1, &dev, "", NULL, NULL);
    if (err != CL_SUCCESS) {
        char log[8192] = {0};
        clGetProgramBuildInfo(prog, dev, CL_PROGRAM_BUILD_LOG, sizeof(log) - 1, log, NULL);
        fprintf(stderr, "ref_run: clBuildProgram failed %d: %s\n", (int)err, log);
        return 4;
    }
    cl_kernel k = clCreateKernel(prog, "raycaster", &err); CHECK(err, "clCreateKernel");

    const cl_mem_flags RO = CL_MEM_READ_ONLY | CL_MEM_COPY_HOST_PTR;
    int map_dim[4] = {(int)dx, (int)dy, (int)dz, 0};           /* kernel reads an int3 = 16 bytes */
    int res[2] = {(int)W, (int)H};
    float cam_dir[4] = {cam[0], cam[1], 0, 0};                 /* CLCaster.cpp:137-139: 16 bytes each */
    float cam_pos[4] = {cam[2], cam[3], cam[4], 0};
    int light_count[2] = {1, 0};
    int atlas_dim[2] = {(int)aw, (int)ah}, tile_dim[2] = {(int)tw, (int)th};
    cl_ulong settings[64];
    memset(settings, 0, sizeof(settings));
    settings[0] = (cl_ulong)dim;            /* OCTDIM            (Application.cpp:35) */
    settings[1] = (cl_ulong)using_octree;   /* OCTENABLED        (Application.cpp:38-39) */
    settings[2] = (cl_ulong)root;           /* OCTREE_ROOT_INDEX (CLCaster.cpp:113) */
    cl_uint attach_lookup[4] = {0, 0, 0, 0};
    cl_ulong attach[4] = {0, 0, 0, 0};

    cl_mem b_map = clCreateBuffer(ctx, RO, n_map, map, &err); CHECK(err, "map");
    cl_mem b_dim = clCreateBuffer(ctx, RO, sizeof(map_dim), map_dim, &err); CHECK(err, "map_dim");
    cl_mem b_res = clCreateBuffer(ctx, RO, sizeof(res), res, &err); CHECK(err, "res");
    cl_mem b_vp = clCreateBuffer(ctx, RO, n_vp, vp, &err); CHECK(err, "viewport");
    cl_mem b_cd = clCreateBuffer(ctx, RO, sizeof(cam_dir), cam_dir, &err); CHECK(err, "cam_dir");
    cl_mem b_cp = clCreateBuffer(ctx, RO, sizeof(cam_pos), cam_pos, &err); CHECK(err, "cam_pos");
    cl_mem b_li = clCreateBuffer(ctx, RO, n_li, li, &err); CHECK(err, "lights");
    cl_mem b_lc = clCreateBuffer(ctx, RO, sizeof(light_count), light_count, &err); CHECK(err, "light_count");
    cl_mem b_ad = clCreateBuffer(ctx, RO, sizeof(atlas_dim), atlas_dim, &err); CHECK(err, "atlas_dim");
    cl_mem b_td = clCreateBuffer(ctx, RO, sizeof(tile_dim), tile_dim, &err); CHECK(err, "tile_dim");
    cl_mem b_de = clCreateBuffer(ctx, RO, n_de, de, &err); CHECK(err, "desc");
    cl_mem b_al = clCreateBuffer(ctx, RO, sizeof(attach_lookup), attach_lookup, &err); CHECK(err, "attach_lookup");
    cl_mem b_ab = clCreateBuffer(ctx, RO, sizeof(attach), attach, &err); CHECK(err, "attach");
    cl_mem b_se = clCreateBuffer(ctx, RO, sizeof(settings), settings, &err); CHECK(err, "settings");

    cl_image_format ffmt = {CL_RGBA, CL_FLOAT}, afmt = {CL_RGBA, CL_UNORM_INT8};
    cl_image_desc idesc;
    memset(&idesc, 0, sizeof(idesc));
    idesc.image_type = CL_MEM_OBJECT_IMAGE2D;
    idesc.image_width = (size_t)W; idesc.image_height = (size_t)H;
    float *init = (float *)malloc(sizeof(float) * 4 * W * H);
    for (long i = 0; i < W * H; i++) { init[4*i] = 1.f; init[4*i+1] = 1.f; init[4*i+2] = 1.f; init[4*i+3] = 100.0f / 255.0f; }  /* CLCaster.cpp:280-286 */
    cl_mem img = clCreateImage(ctx, CL_MEM_WRITE_ONLY | CL_MEM_COPY_HOST_PTR, &ffmt, &idesc, init, &err); CHECK(err, "image");
    idesc.image_width = (size_t)aw; idesc.image_height = (size_t)ah;
    cl_mem atl = clCreateImage(ctx, RO, &afmt, &idesc, at, &err); CHECK(err, "atlas");

    cl_mem args[16] = {b_map, b_dim, b_res, b_vp, b_cd, b_cp, b_li, b_lc, img, atl, b_ad, b_td, b_de, b_al, b_ab, b_se};
    for (int i = 0; i < 16; i++) { err = clSetKernelArg(k, (cl_uint)i, sizeof(cl_mem), &args[i]); CHECK(err, "clSetKernelArg"); }

    size_t global[2] = {(size_t)W, (size_t)H};
    err = clEnqueueNDRangeKernel(q, k, 2, NULL, global, NULL, 0, NULL, NULL); CHECK(err, "clEnqueueNDRangeKernel");
    err = clFinish(q); CHECK(err, "clFinish");

    size_t origin[3] = {0, 0, 0}, region[3] = {(size_t)W, (size_t)H, 1};
    err = clEnqueueReadImage(q, img, CL_TRUE, origin, region, 0, 0, init, 0, NULL, NULL); CHECK(err, "clEnqueueReadImage");
    FILE *f = fopen(out_path, "wb");
    if (!f) { fprintf(stderr, "ref_run: cannot write %s\n", out_path); return 2; }
    fwrite(init, sizeof(float) * 4, (size_t)(W * H), f);
    fclose(f);
    fprintf(stderr, "ref_run: wrote %s (%ldx%ld)\n", out_path, W, H);
    return 0;
}
