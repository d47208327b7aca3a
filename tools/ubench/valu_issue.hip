// Micro-benchmark (round 2): VALU issue cost per wave64 instruction on gfx950, measured in SHADER CYCLES with s_memtime
// inside the kernel (not wall time at the nominal clock: a chip-filling VALU loop runs far below 2.4 GHz under DVFS, which
// is what made profiles/r01_valu_issue_rate.txt read "3 cycles" for v_fma_f32).  For each instruction class: W waves per
// SIMD (1, 2, 4, 8), 8 independent dependency chains per wave, 64 x 16 instructions between two s_memtime reads; the
// figure is (cycles of the slowest wave on the SIMD) * / (instructions issued by all W waves of that SIMD), i.e. the
// SIMD's issue interval.  Also prints wall-clock derived "cycles at 2.4 GHz" so the two views can be compared, and the
// effective clock = shader cycles / wall time.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue && tools/ubench/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define BODY(OP)                                                                                           \
    REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                       \
                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k0), "v"(k1));)

#define DEFK(name, OP)                                                                                     \
__global__ __launch_bounds__(256) void k_##name(float *out, unsigned long long *cyc, int iters) {          \
    float r0 = threadIdx.x + 1.0f, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    float k0 = 1.0001f, k1 = 0.5f;                                                                          \
    const unsigned long long t0 = __builtin_readcyclecounter();                                             \
    for (int i = 0; i < iters; i++) { BODY(OP) }                                                            \
    const unsigned long long t1 = __builtin_readcyclecounter();                                             \
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                            \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                       \
}

#define OP_FMA(i)    "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_ADD(i)    "v_add_f32 %" #i ", %" #i ", %9\n"
#define OP_MUL(i)    "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_FMAC(i)   "v_fmac_f32 %" #i ", %8, %9\n"
#define OP_FMACL(i)  "v_fma_f32 %" #i ", %" #i ", %8, %9 clamp\n"
#define OP_MIN(i)    "v_min_f32 %" #i ", %" #i ", %9\n"
#define OP_MIN3(i)   "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MAX(i)    "v_max_f32 %" #i ", %" #i ", %9\n"
#define OP_CMP(i)    "v_cmp_le_f32 vcc, %" #i ", %9\n"
#define OP_CNDS(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
#define OP_CVT(i)    "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define OP_ADDU(i)   "v_add_u32 %" #i ", %" #i ", %9\n"
#define OP_AND(i)    "v_and_b32 %" #i ", %" #i ", %9\n"
#define OP_LSHL(i)   "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_MOV(i)    "v_mov_b32 %" #i ", %9\n"
#define OP_RCP(i)    "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_MADU(i)   "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_BFE(i)    "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define OP_SUB(i)    "v_sub_f32 %" #i ", %" #i ", %9\n"
#define OP_SQRT(i)   "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_FLOOR(i)  "v_floor_f32 %" #i ", %" #i "\n"
#define OP_RNDNE(i)  "v_rndne_f32 %" #i ", %" #i "\n"
#define OP_CVTF(i)   "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define OP_MULLO(i)  "v_mul_lo_u32 %" #i ", %" #i ", %9\n"
#define OP_MULHI(i)  "v_mul_hi_u32 %" #i ", %" #i ", %9\n"
#define OP_MUL24(i)  "v_mul_u32_u24 %" #i ", %" #i ", %9\n"
#define OP_SUBU(i)   "v_sub_u32 %" #i ", %" #i ", %9\n"
#define OP_CMPU(i)   "v_cmp_lt_u32 vcc, %" #i ", %9\n"
#define OP_CNDV(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_FFBH(i)   "v_ffbh_u32 %" #i ", %" #i "\n"
#define OP_BCNT(i)   "v_bcnt_u32_b32 %" #i ", %" #i ", %9\n"

DEFK(fma, OP_FMA) DEFK(add, OP_ADD) DEFK(mul, OP_MUL) DEFK(fmac, OP_FMAC) DEFK(fmaclamp, OP_FMACL) DEFK(minf, OP_MIN)
DEFK(min3, OP_MIN3) DEFK(maxf, OP_MAX) DEFK(cmp, OP_CMP) DEFK(cnds, OP_CNDS) DEFK(cvt, OP_CVT) DEFK(addu, OP_ADDU)
DEFK(andb, OP_AND) DEFK(lshl, OP_LSHL) DEFK(mov, OP_MOV) DEFK(rcp, OP_RCP) DEFK(madu, OP_MADU) DEFK(bfe, OP_BFE) DEFK(sub, OP_SUB)
DEFK(sqrtf, OP_SQRT) DEFK(floorf, OP_FLOOR) DEFK(rndne, OP_RNDNE) DEFK(cvtf, OP_CVTF) DEFK(mullo, OP_MULLO) DEFK(mulhi, OP_MULHI)
DEFK(mul24, OP_MUL24) DEFK(subu, OP_SUBU) DEFK(cmpu, OP_CMPU) DEFK(cndv, OP_CNDV) DEFK(ffbh, OP_FFBH) DEFK(bcnt, OP_BCNT)

// compare + select pairs (round 5): the select reading VCC (what the compiler's VOP2 forms use) against the same pair through
// another SGPR pair; two instructions per OP, so the printed interval is per PAIR.  s[20:23] and vcc are declared clobbered.
#define BODYC(OP)                                                                                          \
    REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                       \
                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k0), "v"(k1)  \
                      : "vcc", "s20", "s21", "s22", "s23");)
#define DEFKC(name, OP)                                                                                    \
__global__ __launch_bounds__(256) void k_##name(float *out, unsigned long long *cyc, int iters) {          \
    float r0 = threadIdx.x + 1.0f, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    float k0 = 1.0001f, k1 = 0.5f;                                                                          \
    const unsigned long long t0 = __builtin_readcyclecounter();                                             \
    for (int i = 0; i < iters; i++) { BODYC(OP) }                                                           \
    const unsigned long long t1 = __builtin_readcyclecounter();                                             \
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                            \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                       \
}
#define OP_PAIRV(i)  "v_cmp_lt_f32 vcc, %" #i ", %9\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_PAIRS(i)  "v_cmp_lt_f32 s[20:21], %" #i ", %9\n v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
#define OP_PAIRS2(i) "v_cmp_lt_f32 s[20:21], %" #i ", %9\n v_cmp_gt_f32 s[22:23], %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n v_cndmask_b32 %" #i ", %" #i ", %9, s[22:23]\n"
#define OP_PAIRV2(i) "v_cmp_lt_f32 vcc, %" #i ", %9\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n v_cmp_gt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define OP_CNDV64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMPVADD(i) "v_cmp_lt_f32 vcc, %" #i ", %9\n v_add_f32 %" #i ", %" #i ", %8\n"
#define OP_CNDVADD(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n v_add_f32 %" #i ", %" #i ", %9\n"
#define OP_CNDV64ADD(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n v_add_f32 %" #i ", %" #i ", %9\n"
#define OP_CNDVADD3(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n v_add_f32 %" #i ", %" #i ", %9\n v_mul_f32 %" #i ", %" #i ", %8\n v_add_f32 %" #i ", %" #i ", %9\n"
#define OP_CNDV64ADD3(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n v_add_f32 %" #i ", %" #i ", %9\n v_mul_f32 %" #i ", %" #i ", %8\n v_add_f32 %" #i ", %" #i ", %9\n"
DEFKC(cndvadd, OP_CNDVADD) DEFKC(cndv64add, OP_CNDV64ADD) DEFKC(cndvadd3, OP_CNDVADD3) DEFKC(cndv64add3, OP_CNDV64ADD3)
DEFKC(pairv, OP_PAIRV) DEFKC(pairs, OP_PAIRS) DEFKC(pairs2, OP_PAIRS2) DEFKC(pairv2, OP_PAIRV2) DEFKC(cndv64, OP_CNDV64) DEFKC(cmpvadd, OP_CMPVADD)

// packed f32 (two f32 operations per lane and instruction, 64-bit register pairs)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define BODY2(OP)                                                                                          \
    REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                       \
                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k0), "v"(k1));)
#define DEFK2(name, OP)                                                                                    \
__global__ __launch_bounds__(256) void k_##name(float *out, unsigned long long *cyc, int iters) {          \
    f32x2 r0 = {threadIdx.x + 1.0f, 2.0f}, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    f32x2 k0 = {1.0001f, 1.0002f}, k1 = {0.5f, 0.25f};                                                      \
    const unsigned long long t0 = __builtin_readcyclecounter();                                             \
    for (int i = 0; i < iters; i++) { BODY2(OP) }                                                           \
    const unsigned long long t1 = __builtin_readcyclecounter();                                             \
    const f32x2 s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                                  \
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                                        \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                       \
}
#define OP_PKFMA(i)   "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_PKFMACL(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9 clamp\n"
#define OP_PKADD(i)   "v_pk_add_f32 %" #i ", %" #i ", %9\n"
#define OP_PKMUL(i)   "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
DEFK2(pkfma, OP_PKFMA) DEFK2(pkfmacl, OP_PKFMACL) DEFK2(pkadd, OP_PKADD) DEFK2(pkmul, OP_PKMUL)

// fp64 (64-bit register pairs): the closed-form jumps' modular products, floors and reciprocal refinements (exact_jump.hpp)
#define DEFK3(name, OP)                                                                                    \
__global__ __launch_bounds__(256) void k_##name(float *out, unsigned long long *cyc, int iters) {          \
    double r0 = threadIdx.x + 1.0, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    double k0 = 1.0001, k1 = 0.5;                                                                           \
    const unsigned long long t0 = __builtin_readcyclecounter();                                             \
    for (int i = 0; i < iters; i++) { BODY2(OP) }                                                           \
    const unsigned long long t1 = __builtin_readcyclecounter();                                             \
    out[blockIdx.x * 256 + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);                   \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                       \
}
#define OP_FMA64(i)   "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define OP_MUL64(i)   "v_mul_f64 %" #i ", %" #i ", %8\n"
#define OP_ADD64(i)   "v_add_f64 %" #i ", %" #i ", %9\n"
#define OP_FLOOR64(i) "v_floor_f64 %" #i ", %" #i "\n"
#define OP_FRACT64(i) "v_fract_f64 %" #i ", %" #i "\n"
DEFK3(fma64, OP_FMA64) DEFK3(mul64, OP_MUL64) DEFK3(add64, OP_ADD64) DEFK3(floor64, OP_FLOOR64) DEFK3(fract64, OP_FRACT64)
// conversions between 32- and 64-bit registers: the result of one feeds the other, so a chain stays a chain
#define DEFK4(name, OPS)                                                                                   \
__global__ __launch_bounds__(256) void k_##name(float *out, unsigned long long *cyc, int iters) {          \
    double d0 = threadIdx.x + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;                                  \
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;                                           \
    const unsigned long long t0 = __builtin_readcyclecounter();                                             \
    for (int i = 0; i < iters; i++) {                                                                       \
        REP8(asm volatile(OPS OPS : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }   \
    const unsigned long long t1 = __builtin_readcyclecounter();                                             \
    out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3) + (float)(i0 + i1 + i2 + i3);         \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                       \
}
// 8 instructions per asm statement like the other bodies: four conversions each way
#define OPS_CVT64 "v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7\n"
#define OPS_CVT32 "v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_cvt_i32_f64 %6, %2\n v_cvt_i32_f64 %7, %3\n"
DEFK4(cvt_f64_i32, OPS_CVT64) DEFK4(cvt_i32_f64, OPS_CVT32)

typedef void (*kern_t)(float *, unsigned long long *, int);

int main() {
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_mul_f32", k_mul}, {"v_fmac_f32", k_fmac},
        {"v_fma_f32 clamp", k_fmaclamp}, {"v_min_f32", k_minf}, {"v_min3_f32", k_min3}, {"v_max_f32", k_maxf},
        {"v_cmp_le_f32 vcc", k_cmp}, {"v_cndmask_b32 (sgpr)", k_cnds}, {"v_cvt_i32_f32", k_cvt}, {"v_add_u32", k_addu},
        {"v_and_b32", k_andb}, {"v_lshlrev_b32", k_lshl}, {"v_mov_b32", k_mov}, {"v_rcp_f32", k_rcp},
        {"v_mad_u32_u24", k_madu}, {"v_bfe_u32", k_bfe},
        {"v_sqrt_f32", k_sqrtf}, {"v_floor_f32", k_floorf}, {"v_rndne_f32", k_rndne}, {"v_cvt_f32_i32", k_cvtf},
        {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_u32_u24", k_mul24}, {"v_sub_u32", k_subu},
        {"v_cmp_lt_u32 vcc", k_cmpu}, {"v_cndmask_b32 (vcc)", k_cndv}, {"v_ffbh_u32", k_ffbh}, {"v_bcnt_u32_b32", k_bcnt},
        {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_floor_f64", k_floor64}, {"v_fract_f64", k_fract64},
        {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_i32_f64", k_cvt_i32_f64},
        {"cmp vcc + cndmask vcc (pair)", k_pairv}, {"cmp s[20:21] + cndmask s[20:21] (pair)", k_pairs},
        {"2 x (cmp, cndmask) via vcc (4 instr)", k_pairv2}, {"2 cmp s[20:23] then 2 cndmask (4 instr)", k_pairs2},
        {"v_cndmask_b32_e64 ..., vcc", k_cndv64}, {"cmp vcc + v_add_f32 (pair)", k_cmpvadd},
        {"cndmask e32 vcc + add (pair)", k_cndvadd}, {"cndmask e64 vcc + add (pair)", k_cndv64add},
        {"cndmask e32 vcc + add, mul, add (4 instr)", k_cndvadd3}, {"cndmask e64 vcc + add, mul, add (4 instr)", k_cndv64add3},
        {"v_pk_fma_f32", k_pkfma}, {"v_pk_fma_f32 clamp", k_pkfmacl}, {"v_pk_add_f32", k_pkadd}, {"v_pk_mul_f32", k_pkmul}};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 256;                                  // 256 x 64 = 16384 instructions per wave between the two clock reads
    float *out; hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    unsigned long long *cyc; hipMalloc(&cyc, sizeof(unsigned long long) * cus * 8 * 4);
    std::vector<unsigned long long> h(cus * 8 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("# %s, %d CUs, nominal clock %d kHz.  Per class and W = waves per SIMD: issue interval in SHADER cycles per wave64\n"
           "# instruction (median over waves of: s_memtime delta * / (16384 * W)); [wall-clock cycles at the nominal clock]; (effective GHz)\n",
           prop.name, cus, prop.clockRate);
    for (auto &e : ks) {
        printf("%-22s", e.name);
        for (int w = 1; w <= 8; w *= 2) {
            const int blocks = cus * w;                     // w blocks of 4 waves per CU = w waves per SIMD
            e.k<<<blocks, 256>>>(out, cyc, 8);
            hipDeviceSynchronize();
            hipEventRecord(a);
            e.k<<<blocks, 256>>>(out, cyc, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + blocks * 4);
            const double med = (double)h[blocks * 2];
            const double insts = (double)iters * 64.0;
            const double shader = med / (insts * w);
            const double wall = ms * 1e-3 * prop.clockRate * 1e3 / (insts * w);
            printf("  W=%d %5.2f [%5.2f] (%.2f)", w, shader, wall, med / (ms * 1e6));
        }
        printf("\n");
    }
    return 0;
}
