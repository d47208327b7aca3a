// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the
// raycast step loop is made of.  8 waves per SIMD, long unrolled independent streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x

#define DEFK(name, body)                                                                      \
__global__ __launch_bounds__(256) void k_##name(float *out, int iters) {                      \
    float a = threadIdx.x * 0.5f + 1.0f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;            \
    float e = a + 4.0f, f = a + 5.0f, g = a + 6.0f, h = a + 7.0f;                             \
    for (int i = 0; i < iters; i++) { REP16(body) }                                           \
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;                      \
}

DEFK(fma,     asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %1, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(add,     asm volatile("v_add_f32 %0, %1, %0\n v_add_f32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(pkfma,   asm volatile("v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %2, %1, %1, %2" : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e));)
DEFK(min3,    asm volatile("v_min3_f32 %0, %1, %2, %0\n v_min3_f32 %3, %1, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(cmp,     asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %2, %3" :: "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");)
DEFK(cndmask, asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %1, %2, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
DEFK(addu32,  asm volatile("v_add_u32 %0, %1, %0\n v_add_u32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(subrev,  asm volatile("v_subbrev_co_u32 %0, vcc, 0, %0, vcc\n v_subbrev_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a), "+v"(b) :: "vcc");)
DEFK(mov,     asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %1" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(xor3,    asm volatile("v_xor_b32 %0, %1, %0\n v_or3_b32 %2, %1, %0, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(minf,    asm volatile("v_min_f32 %0, %1, %0\n v_min_f32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(cmpx,    asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n v_cmp_le_f32 s[22:23], %2, %3" :: "v"(a), "v"(b), "v"(c), "v"(d) : "s20", "s21", "s22", "s23");)
DEFK(mad_i,   asm volatile("v_mad_u32_u24 %0, %1, %2, %0\n v_mad_u32_u24 %3, %1, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(salu,    asm volatile("s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20" ::: "s20", "s21", "scc");)

typedef void (*kern_t)(float *, int);

int main() {
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_pk_fma_f32", k_pkfma}, {"v_min3_f32", k_min3},
        {"v_min_f32", k_minf}, {"v_cmp_le_f32(vcc)", k_cmp}, {"v_cmp_le_f32(sgpr)", k_cmpx}, {"v_cndmask_b32", k_cndmask},
        {"v_add_u32", k_addu32}, {"v_subbrev_co_u32", k_subrev}, {"v_mov_b32", k_mov}, {"v_xor/v_or3", k_xor3},
        {"v_mad_u32_u24", k_mad_i}, {"s_add/s_and (SALU)", k_salu}};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;            // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
    float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
    const int iters = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("# %s, %d CUs, clock %d kHz; cycles are per wave64 instruction per SIMD at the max clock\n", prop.name, cus, prop.clockRate);
    for (auto &e : ks) {
        e.k<<<blocks, 256>>>(out, 16);
        hipDeviceSynchronize();
        hipEventRecord(a);
        e.k<<<blocks, 256>>>(out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double insts_per_simd = (double)iters * 32.0 * 8.0;   // 32 instructions per iteration, 8 waves per SIMD
        const double cyc = ms * 1e-3 * prop.clockRate * 1e3 / insts_per_simd;
        printf("%-22s %8.3f ms  %6.2f cycles/inst\n", e.name, ms, cyc);
    }
    return 0;
}
