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

DEFK(mul,     asm volatile("v_mul_f32 %0, %1, %0\n v_mul_f32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(sub,     asm volatile("v_sub_f32 %0, %1, %0\n v_sub_f32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(fmac,    asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(fmaclamp,asm volatile("v_fma_f32 %0, %1, %2, 1.0 clamp\n v_fma_f32 %3, %1, %2, 1.0 clamp" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(maxf,    asm volatile("v_max_f32 %0, %1, %0\n v_max_f32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(med3,    asm volatile("v_med3_f32 %0, %1, %2, %0\n v_med3_f32 %3, %1, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(andb,    asm volatile("v_and_b32 %0, %1, %0\n v_or_b32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(lshl,    asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1" : "+v"(a), "+v"(b));)
DEFK(cvtfi,   asm volatile("v_cvt_f32_i32 %0, %1\n v_cvt_i32_f32 %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(minu,    asm volatile("v_min_u32 %0, %1, %0\n v_max_u32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(subu,    asm volatile("v_sub_u32 %0, %1, %0\n v_subrev_u32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(add3,    asm volatile("v_add3_u32 %0, %1, %2, %0\n v_lshl_add_u32 %3, %1, 2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEFK(addco,   asm volatile("v_add_co_u32 %0, vcc, %1, %0\n v_add_co_u32 %2, vcc, %1, %2" : "+v"(a), "+v"(b), "+v"(c) :: "vcc");)
DEFK(cndsg,   asm volatile("v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %2, %1, %2, s[20:21]" : "+v"(a), "+v"(b), "+v"(c) :: "s20", "s21");)
DEFK(cndvcc,  asm volatile("v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %1, %2, vcc" : "+v"(a), "+v"(b), "+v"(c) :: "vcc");)
DEFK(bfe,     asm volatile("v_bfe_u32 %0, %1, 3, 5\n v_bfe_u32 %2, %1, 3, 5" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(cmpeq,   asm volatile("v_cmp_eq_f32 vcc, 0, %0\n v_cmp_eq_u32 vcc, 0, %1" :: "v"(a), "v"(b) : "vcc");)
DEFK(movdpp,  asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(mul24,   asm volatile("v_mul_u32_u24 %0, %1, %0\n v_mul_lo_u32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(rcp,     asm volatile("v_rcp_f32 %0, %0\n v_sqrt_f32 %1, %1" : "+v"(a), "+v"(b));)
DEFK(ffbh,    asm volatile("v_ffbh_u32 %0, %1\n v_bcnt_u32_b32 %2, %1, %2" : "+v"(a), "+v"(b), "+v"(c));)
DEFK(addf64,  asm volatile("v_add_f64 %0, %1, %0\n v_fma_f64 %2, %1, %1, %2" : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e));)

DEFK(cnd64vcc,asm volatile("v_cndmask_b32_e64 %0, %1, %0, vcc\n v_cndmask_b32_e64 %2, %1, %2, vcc" : "+v"(a), "+v"(b), "+v"(c) :: "vcc");)
DEFK(cmpcnd32,asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %1, %2, vcc" : "+v"(a), "+v"(b), "+v"(c) :: "vcc");)
DEFK(cmpcnd64,asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n v_cndmask_b32_e64 %2, %1, %2, s[20:21]" : "+v"(a), "+v"(b), "+v"(c) :: "s20", "s21");)
DEFK(cnd32k,  asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, vcc\n v_cndmask_b32_e64 %1, 0, 1.0, vcc" : "+v"(a), "+v"(b) :: "vcc");)

typedef void (*kern_t)(float *, int);

int main() {
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_pk_fma_f32", k_pkfma}, {"v_min3_f32", k_min3},
        {"v_min_f32", k_minf}, {"v_cmp_le_f32(vcc)", k_cmp}, {"v_cmp_le_f32(sgpr)", k_cmpx}, {"v_cndmask_b32", k_cndmask},
        {"v_add_u32", k_addu32}, {"v_subbrev_co_u32", k_subrev}, {"v_mov_b32", k_mov}, {"v_xor/v_or3", k_xor3},
        {"v_mad_u32_u24", k_mad_i}, {"s_add/s_and (SALU)", k_salu},
        {"v_mul_f32", k_mul}, {"v_sub_f32", k_sub}, {"v_fmac_f32", k_fmac}, {"v_fma_f32 clamp", k_fmaclamp},
        {"v_max_f32", k_maxf}, {"v_med3_f32", k_med3}, {"v_and/v_or_b32", k_andb}, {"v_lshl/lshr_b32", k_lshl},
        {"v_cvt f32<->i32", k_cvtfi}, {"v_min/max_u32", k_minu}, {"v_sub/subrev_u32", k_subu},
        {"v_add3/lshl_add_u32", k_add3}, {"v_add_co_u32", k_addco}, {"v_cndmask(sgpr mask)", k_cndsg},
        {"v_cndmask(vcc) v2", k_cndvcc}, {"v_bfe_u32", k_bfe}, {"v_cmp_eq f32/u32", k_cmpeq},
        {"v_mov_b32_dpp", k_movdpp}, {"v_mul_u32_u24/mul_lo", k_mul24}, {"v_rcp/v_sqrt_f32", k_rcp},
        {"v_ffbh/v_bcnt", k_ffbh}, {"v_add_f64/v_fma_f64", k_addf64},
        {"v_cndmask_e64(vcc)", k_cnd64vcc}, {"v_cmp vcc + v_cndmask_e32", k_cmpcnd32},
        {"v_cmp sgpr + v_cndmask_e64", k_cmpcnd64}, {"v_cndmask_e64 0,1.0,vcc", k_cnd32k}};
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
