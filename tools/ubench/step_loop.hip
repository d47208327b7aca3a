// Micro-benchmark: cycles per wave-iteration of the SVO kernel's step-loop variants in isolation
// (same flags as the library: -O3 -ffp-contract=off -fno-slp-vectorize), 6 waves per SIMD like the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float unit_if_zero(float d) {
    float f; asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(f) : "v"(d), "s"(-0x1p127f)); return f;
}
__device__ __forceinline__ float alive_if_zero(float d, float alive) {
    float f; asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(f) : "v"(d), "s"(-0x1p127f), "v"(alive)); return f;
}
__device__ __forceinline__ float mul_clamped(float a, float b) {
    float f; asm("v_mul_f32 %0, %1, %2 clamp" : "=v"(f) : "v"(a), "v"(b)); return f;
}

#define SETUP                                                                                              \
    const int t = blockIdx.x * blockDim.x + threadIdx.x;                                                   \
    float dtx = 1.0f + (t & 63) * 0.013f, dty = 1.7f + (t & 31) * 0.021f, dtz = 2.9f + (t & 15) * 0.017f;  \
    float itx = 0.3f * dtx, ity = 0.6f * dty, itz = 0.1f * dtz;                                            \
    float nx = 1e6f, ny = 1e6f, nz = 1e6f, left = (float)iters, fxf = 0, fyf = 0, fzf = 0; bool go;

#define FINISH out[t] = itx + ity + itz + nx + ny + nz + left + fxf + fyf + fzf;

__global__ __launch_bounds__(256, 6) void k_cmp(float *out, int iters) {
    SETUP
    do {
        const float m = fminf(fminf(itx, ity), itz);
        fxf = itx <= m ? 1.0f : 0.0f; fyf = ity <= m ? 1.0f : 0.0f; fzf = itz <= m ? 1.0f : 0.0f;
        itx = __builtin_fmaf(dtx, fxf, itx); ity = __builtin_fmaf(dty, fyf, ity); itz = __builtin_fmaf(dtz, fzf, itz);
        nx -= fxf; ny -= fyf; nz -= fzf; left -= 1.0f;
        go = fminf(fminf(fminf(nx, ny), nz), left) != 0.0f;
    } while (go);
    FINISH
}
__global__ __launch_bounds__(256, 6) void k_arith(float *out, int iters) {
    SETUP
    do {
        const float m = fminf(fminf(itx, ity), itz);
        fxf = unit_if_zero(itx - m); fyf = unit_if_zero(ity - m); fzf = unit_if_zero(itz - m);
        itx = __builtin_fmaf(dtx, fxf, itx); ity = __builtin_fmaf(dty, fyf, ity); itz = __builtin_fmaf(dtz, fzf, itz);
        nx -= fxf; ny -= fyf; nz -= fzf; left -= 1.0f;
        go = fminf(fminf(fminf(nx, ny), nz), left) != 0.0f;
    } while (go);
    FINISH
}
__global__ __launch_bounds__(256, 6) void k_arith_prod(float *out, int iters) {
    SETUP
    do {
        const float m = fminf(fminf(itx, ity), itz);
        fxf = unit_if_zero(itx - m); fyf = unit_if_zero(ity - m); fzf = unit_if_zero(itz - m);
        itx = __builtin_fmaf(dtx, fxf, itx); ity = __builtin_fmaf(dty, fyf, ity); itz = __builtin_fmaf(dtz, fzf, itz);
        nx -= fxf; ny -= fyf; nz -= fzf; left -= 1.0f;
        go = (nx * ny) * (nz * left) != 0.0f;
    } while (go);
    FINISH
}
__global__ __launch_bounds__(256, 6) void k_alive2(float *out, int iters) {
    SETUP
    float gx, gy, gz, alive;
    do {
        float m = fminf(fminf(itx, ity), itz);
        fxf = unit_if_zero(itx - m); fyf = unit_if_zero(ity - m); fzf = unit_if_zero(itz - m);
        itx = __builtin_fmaf(dtx, fxf, itx); ity = __builtin_fmaf(dty, fyf, ity); itz = __builtin_fmaf(dtz, fzf, itz);
        nx -= fxf; ny -= fyf; nz -= fzf; left -= 1.0f;
        alive = mul_clamped(nx * ny, nz * left);
        m = fminf(fminf(itx, ity), itz);
        gx = alive_if_zero(itx - m, alive); gy = alive_if_zero(ity - m, alive); gz = alive_if_zero(itz - m, alive);
        itx = __builtin_fmaf(dtx, gx, itx); ity = __builtin_fmaf(dty, gy, ity); itz = __builtin_fmaf(dtz, gz, itz);
        nx -= gx; ny -= gy; nz -= gz; left -= alive;
        go = (nx * ny) * (nz * left) != 0.0f;
    } while (go);
    if (alive != 0.0f) { fxf = gx; fyf = gy; fzf = gz; }
    FINISH
}
// lower bound: the recurrence alone (mask + t update), exit on a scalar counter
__global__ __launch_bounds__(256, 6) void k_floor(float *out, int iters) {
    SETUP
    for (int i = 0; i < iters; i++) {
        const float m = fminf(fminf(itx, ity), itz);
        fxf = unit_if_zero(itx - m); fyf = unit_if_zero(ity - m); fzf = unit_if_zero(itz - m);
        itx = __builtin_fmaf(dtx, fxf, itx); ity = __builtin_fmaf(dty, fyf, ity); itz = __builtin_fmaf(dtz, fzf, itz);
        nx -= fxf; ny -= fyf; nz -= fzf;
    }
    FINISH
}

// "safe run": no countdowns; a lane steps while the crossing time m is below a threshold T (alive = clamp((T - m) * B1)),
// all lanes execute every trip (a lane that is done takes empty steps), loop control is scalar
__global__ __launch_bounds__(256, 6) void k_safe(float *out, int iters) {
    SETUP
    const float T = 1e30f, B1 = 0x1p-70f, TB1 = T * B1;
    float cnt = 0.0f, alive = 1.0f;
    for (int trip = 0; trip < iters / 2; trip++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float m = fminf(fminf(itx, ity), itz);
            asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(alive) : "v"(m), "v"(-B1), "v"(TB1));
            cnt += alive;
            fxf = alive_if_zero(itx - m, alive); fyf = alive_if_zero(ity - m, alive); fzf = alive_if_zero(itz - m, alive);
            itx = __builtin_fmaf(dtx, fxf, itx); ity = __builtin_fmaf(dty, fyf, ity); itz = __builtin_fmaf(dtz, fzf, itz);
        }
        if (__ballot(alive != 0.0f) == 0ULL) break;
    }
    nx -= cnt;
    FINISH
}

typedef void (*kern_t)(float *, int);
int main() {
    struct { const char *name; kern_t k; int valu; } ks[] = {
        {"cmp/cndmask mask, min exit (old loop)", k_cmp, 17}, {"arith mask, min exit (arith_mask=1)", k_arith, 17},
        {"arith mask, product exit", k_arith_prod, 17}, {"2 iterations per exit test (arith_mask=2)", k_alive2, 0},
        {"recurrence only, scalar trip count", k_floor, 13}, {"safe run (threshold on t, no countdowns)", k_safe, 12}};
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 6;     // 6 blocks x 4 waves per CU = 6 waves per SIMD
    float *out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 200000;
    printf("# %s: %d CUs, 6 waves/SIMD, %d iterations per lane; cycles at the nominal %d MHz\n", prop.name,
           prop.multiProcessorCount, iters, prop.clockRate / 1000);
    for (auto &e : ks) {
        e.k<<<blocks, 256>>>(out, 1000);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        e.k<<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double cyc = ms * 1e-3 * prop.clockRate * 1e3 / ((double)iters * 6.0);
        printf("%-45s %8.3f ms  %6.1f cycles per wave-iteration per SIMD\n", e.name, ms, cyc);
    }
    return 0;
}
