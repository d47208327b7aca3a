// exact_jump.hpp -- O(1) emulation of the reference's DDA float recurrence across an empty node.
//
// The reference advances a ray with (kernels/ray_caster_kernel.cl:558-560)
//     face_mask = (t.xyz <= min(t.yzx, t.zxy));   t += delta_t * face_mask;   voxel += step * face_mask;
// once per loop iteration, and the iteration count feeds the fog factor (:716), the out-of-map colour (:565) and the
// shadow-ray cap (:667).  Bit-exact parity therefore needs the k-fold float accumulation of t AND the exact number of
// iterations, including iterations in which two or three axes tie and step together.  This is what the reference's own
// unfinished octree branch was reaching for at :525-540 (intersection_t += delta_t * jump_power * fabs(face_mask)).
//
// Inside one binade [2^e, 2^(e+1)) a float t is an integer mantissa M times u = 2^(e-23), and RNE(t + d) = t + inc*u
// with a constant integer inc (d rounded to a multiple of u; when d sits exactly half-way the increment is the even
// neighbour once M is even).  So, until it leaves its binade, each axis is an exact arithmetic progression of integers
// -- of the float's BIT PATTERN even: bits(v_k) = bits(t) + k*inc -- and
//   * the value after k steps,
//   * the number of steps with value <= X (one integer division),
//   * the number of EQUAL values of two axes (a linear congruence: one extended-Euclid run per axis pair, binade and
//     ray direction, kept in a per-ray table)
// are closed forms.  stretch_jump() consumes every iteration up to and including the one that leaves the node -- or up
// to the end of the binade of an axis whose node face lies beyond it, whichever comes first -- and reports how many loop
// iterations that was.  It ALWAYS makes progress (>= 1 iteration): an axis outside the simple regime (t <= 0 or tiny, t not above
// delta_t's binade, unsettled half-way case, increment below 64 ulps) is FROZEN, i.e. a one-element progression whose
// single step is taken with one real float add and which bounds the stretch.  A stretch that ends early is simply
// continued by the next call (or by the ordinary step loop, which is always correct).
//
// The Euclid runs are the expensive part (a loop of ~10-16 dependent divisions, least-absolute-remainder steps) and a ray
// needs one per axis pair and binade.  The increments depend on (delta_t, binade) only, so the runs are shared by the wave: when the first lane needs
// binade e, the three pairs of EVERY lane whose ray keeps its direction are solved for e in one loop (jump_rows_build) and
// kept in a per-ray table -- the rays of a tile reach a binade within a few rounds of each other.
//
// Host+device header: tools/jumptest/jump_vs_loop.cpp drives the host build against the plain loop on millions of
// random states (tests/test_exact_jump.py); raycast_kernel.hip uses the device build.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VRC_HD __host__ __device__ __forceinline__
#else
#define VRC_HD inline
#endif
// wave votes: the warm and cold sections run only when a lane needs them; the host build is a "wave" of one lane
#if defined(__HIP_DEVICE_COMPILE__)
#define VRC_BALLOT(c) __ballot(c)
#define VRC_WAVE_ANY(c) (__ballot(c) != 0ULL)
#define VRC_FIRST_LANE_VALUE(v, mask) __builtin_amdgcn_readlane((v), __ffsll((long long)(mask)) - 1)
#else
#define VRC_BALLOT(c) ((c) ? 1ULL : 0ULL)
#define VRC_WAVE_ANY(c) (c)
#define VRC_FIRST_LANE_VALUE(v, mask) (v)
#endif

namespace vrc {

#if defined(VRC_SCHED_STATS) && defined(__HIPCC__)
// profiling build only: [0] lanes that had to solve a pair on the spot (no usable table entry), [1] lanes that sat through
// such a solve for another lane, [2] warm tie-count passes (lanes with a solution index inside the stretch)
__device__ unsigned long long g_jump_private_solves[4];
#endif

VRC_HD uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
VRC_HD float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

VRC_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);          // 1 ulp
#elif defined(VRC_JUMP_FUZZ_RCP)
    // host test mode: perturb the reciprocal like a 1-ulp hardware approximation would
    const float r = 1.0f / x;
    const uint32_t h = (f2u(x) * 2654435761u) >> 30;
    return u2f(f2u(r) + (h == 0 ? 1u : (h == 1 ? (uint32_t)-1 : 0u)));
#else
    return 1.0f / x;
#endif
}
// a * b for 0 <= a, b < 2^24 (v_mul_u32_u24: full rate; v_mul_lo_u32 is not)
VRC_HD int32_t mul24(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int32_t)__umul24((unsigned)a, (unsigned)b);
#else
    return (int32_t)((uint32_t)a * (uint32_t)b);
#endif
}

// reciprocal of an integer 1 <= b < 2^24 to ~2^-44 relative: hardware estimate + one Newton step in fp64
VRC_HD double recip_d(int32_t b) {
    const double bd = (double)b, r0 = (double)fast_rcp((float)b);
    return __builtin_fma(r0, __builtin_fma(-bd, r0, 1.0), r0);
}
// floor(a / b) and a - floor(a/b) * b for |a| < 2^31, 1 <= b < 2^24, |a / b| < 2^26; rb = recip_d(b).  Exact: the
// estimate is within 2^-18 of the quotient, so the floor is at most one off, and the remainder check settles it.
VRC_HD int32_t floordiv(int32_t a, int32_t b, double rb, int32_t &rem) {
    int32_t q = (int32_t)__builtin_floor((double)a * rb);
    int32_t r = a - q * b;
    const int32_t dn = (r < 0) ? 1 : 0;
    q -= dn; r += dn ? b : 0;
    const int32_t up = (r >= b) ? 1 : 0;
    q += up; r -= up ? b : 0;
    rem = r;
    return q;
}
// floor(a / b) for 0 <= a < 2^24, 64 <= b < 2^24 (quotient < 2^18: the float estimate is within one)
VRC_HD int32_t floordiv_small(int32_t a, int32_t b) {
    int32_t q = (int32_t)((float)a * fast_rcp((float)b));
    const int32_t r = a - mul24(q, b);
    q -= (r < 0) ? 1 : 0;
    q += (r >= b) ? 1 : 0;
    return q;
}
// (c * s) mod m in [0, m) for |c| < 2^24, 0 <= s < m < 2^24: the 48-bit product is exact in fp64; rm = recip_d(m)
VRC_HD int32_t mulmod(int32_t c, int32_t s, int32_t m, double rm) {
    const double prod = (double)c * (double)s, dm = (double)m;
    const double q = __builtin_floor(prod * rm);                  // |true quotient| < 2^24: within 2^-20 of it
    double r = __builtin_fma(-q, dm, prod);                       // exact
    r += (r < 0.0) ? dm : 0.0;
    r -= (r >= dm) ? dm : 0.0;
    return (int32_t)r;
}

// ---------------------------------------------------------------------------------------------------------------
// The increment of one axis in the binade with exponent field e: bits(RNE(2^(e-127) + d)) - bits(2^(e-127)), i.e. the
// FPU's own rounding of d to a multiple of the binade's ulp, started from an even mantissa.  It depends on (d, e) only.
// `halfway`: d lies exactly half-way between two multiples (then the progression needs an even mantissa to start from).
// Valid for jump_binade_ok(e, d).
// ---------------------------------------------------------------------------------------------------------------
VRC_HD bool jump_binade_ok(int32_t e, float d) {
    const int32_t sh = e - (int32_t)(f2u(d) >> 23);
    // 25 <= e: half an ulp is a normal float; sh >= 1: t + d can stay inside the binade at all; sh <= 17 keeps the
    // increment at 64 ulps or more (quotients stay small enough for float estimates)
    return (uint32_t)(e - 25) <= (uint32_t)(253 - 25) && (uint32_t)(sh - 1) <= 16u;
}
VRC_HD int32_t jump_axis_inc(int32_t e, float d, bool &halfway) {
    const float P = u2f((uint32_t)e << 23);
    const float S = P + d;                                        // one real rounding; < 2P or exactly 2P
    const float err = d - (S - P);                                // both subtractions are exact
    halfway = __builtin_fabsf(err) == u2f((uint32_t)(e - 24) << 23);
    return (int32_t)(f2u(S) - ((uint32_t)e << 23));
}

// ---------------------------------------------------------------------------------------------------------------
// per-ray table of the Euclid runs (jump_rows_build): one 8-byte word per (binade row, axis pair),
//   row = e - kJumpFirstBinade (0 <= row < kJumpBinades), pairs xy, xz, yz:   tab[(3 * (row % R) + pair) * stride]
// A ray's intersection_t only grow, so it needs the row of the binade it is in and those ahead: the table is a ring of
// R rows, and `rows` (a bit per row, kept by the caller) says which rows it holds right now.
//   With s * ia == g (mod ib), g = gcd(ia, ib), 0 <= s < ib from the Euclid run, the word holds THETA = s / ib as a double and g
//   (jump_entry_pack); whole word 0: no entry.  The hot path wants (c * s) mod ib for a mantissa difference c: that is
//   ib * fract(c * theta), ONE fp64 product and a v_fract_f64 -- the modulus' reciprocal is a property of the row, not of the
//   jump (rounds 3-5 kept s in a dword and every probe rebuilt 1 / ib: v_rcp_f32, a Newton step, a 48-bit product, floor, fma --
//   a dependent fp64 chain of ~20 instructions per pair, 81 M times per headline frame).
// The increments depend on (delta_t, binade) only.  In HBM (RaycastParams::jump_cache), one table per lane, interleaved
// over the 64 lanes of a wave (stride 64) so that a row is coalesced 512-byte lines; in LDS [row][pair][thread].
// ---------------------------------------------------------------------------------------------------------------
#ifndef VRC_JUMP_FIRST_LOG2
#define VRC_JUMP_FIRST_LOG2 7
#endif
constexpr int kJumpFirstBinade = 127 + VRC_JUMP_FIRST_LOG2;   // no table below t = 128: a binade of fewer voxels than a Euclid run costs
constexpr int kJumpBinades = 19 - VRC_JUMP_FIRST_LOG2;   // t < 2^19
#ifndef VRC_JUMP_RING
#define VRC_JUMP_RING 4
#endif
// rows kept per ray: a ring indexed by row % R (intersection_t only grows).  The functions below take R as a template
// argument: the kernel uses 4 rows everywhere (in LDS, [row][pair][thread]: 12 KB per block, or in the global table buffer);
// 3 rows -- 9 KB -- were what fitted beside the 11-level traversal stack of a depth-12 tree before the coarse top table shrank
// the stack (same speed on the headline frame, 3-4 % slower on deep trees whose rays cross more binades).
constexpr int kJumpRing = VRC_JUMP_RING;      // the default R (host harness: -DVRC_JUMP_RING=3 / 4)
struct alignas(8) JumpWord { uint32_t lo, hi; };   // one table entry (ds_read_b64 / global_load_dwordx2)
template <int R> struct JumpRingMask;
template <> struct JumpRingMask<2> { static constexpr uint32_t value = 0x55555555u; };   // a bit every R rows
template <> struct JumpRingMask<3> { static constexpr uint32_t value = 0x09249249u; };
template <> struct JumpRingMask<4> { static constexpr uint32_t value = 0x11111111u; };
constexpr int kJumpTableWords = 3 * kJumpRing;    // one word per pair and row (default ring)

// s and g with s * ia == g (mod ib), g = gcd(ia, ib), 0 <= s < ib; 1 <= ia, ib < 2^24.  Extended Euclid on exact integers
// held in floats, two steps per trip with the roles of the two remainders alternating (no conditional swaps):
//     (r0, s0) -= q  * (r1, s1)     q  = nearest(r0 / r1)      then        (r1, s1) -= q' * (r0, s0)     q' = nearest(r1 / r0)
// with the invariant r_i == s_i * ia (mod ib), remainders folded to >= 0 (see euclid_reduce).  A zero remainder makes every
// later step a no-op, so lanes simply idle until the last chain of the wave is through.  (The floor form of rounds 2-3: the
// quotient estimate rounded DOWN and raised by one where the remainder allows; a quotient above 2^20 may come out short, the
// step is then partial and the next trip finishes it.)
struct EuclidChain { float r0, s0, r1, s1; };
VRC_HD void euclid_init(EuclidChain &c, bool active, int32_t ia, int32_t ib) {
    c.r0 = (float)ib; c.s0 = 0.0f; c.r1 = active ? (float)ia : 0.0f; c.s1 = 1.0f;
}
// (a, sa) -= floor(a / b) * (b, sb); nothing when b == 0
VRC_HD void euclid_reduce(float &a, float &sa, float b, float sb) {
#ifndef VRC_EUCLID_FLOOR
    // LEAST-ABSOLUTE-REMAINDER step (round 4): q = the integer nearest to a / b, the remainder n = a - q b lies in about
    // [-b/2, b/2] and is folded back to >= 0 together with its cofactor: -n == (-t) * ia (mod ib).  About 30 % fewer steps than
    // the floor form below (rounds 2-3; -DVRC_EUCLID_FLOOR), one instruction less per step, and the headline frame 2.10 -> 1.99 ms.
    // Any integer q keeps the invariant r == s * ia (mod ib); the estimate is within q 2^-22 + 1/2 of a / b, so |n| <= b (1/2 +
    // q 2^-22) < b keeps the chain shrinking and far below 2^24: both fused multiply-adds are exact.
    float q = __builtin_rintf(a * fast_rcp(b));                   // (b == 0: inf or NaN)
    q = (b == 0.0f) ? 0.0f : q;
    const float n = __builtin_fmaf(-q, b, a);                     // exact
    const float t = __builtin_fmaf(-q, sb, sa);                   // exact
    a = __builtin_fabsf(n);
    sa = u2f(f2u(t) ^ (f2u(n) & 0x80000000u));                    // -t where n < 0 (two full-rate bit operations instead of compare + select)
#else
    float q = __builtin_floorf(a * (fast_rcp(b) * 0.99999952f));  // <= floor(a / b), short by < 2^-20 relative (b == 0: inf or NaN)
    q = (b == 0.0f) ? 0.0f : q;
    float n = __builtin_fmaf(-q, b, a);                           // exact, >= 0
    const bool up = n >= b && b != 0.0f;                          // an exact-integer ratio k floors to k - 1: one fix
    q += up ? 1.0f : 0.0f;
    n -= up ? b : 0.0f;
    sa = __builtin_fmaf(-q, sb, sa);                              // exact: the cofactors stay below ib in magnitude
    a = n;
#endif
}
VRC_HD void euclid_step(EuclidChain &c) {
    euclid_reduce(c.r0, c.s0, c.r1, c.s1);
    euclid_reduce(c.r1, c.s1, c.r0, c.s0);
}
VRC_HD bool euclid_busy(const EuclidChain &c) { return c.r0 != 0.0f && c.r1 != 0.0f; }
VRC_HD void euclid_finish(const EuclidChain &c, int32_t ib, int32_t &s_out, int32_t &g_out) {
    const bool first = c.r1 == 0.0f;                              // the remainder that is left holds the gcd
    g_out = (int32_t)(first ? c.r0 : c.r1);
    int32_t s = (int32_t)(first ? c.s0 : c.s1);
    s += (s < 0) ? ib : 0;
    s_out = (s >= ib) ? 0 : s;
}
VRC_HD void pair_solve(bool active, int32_t ia, int32_t ib, int32_t &s_out, int32_t &g_out) {
    EuclidChain c;
    euclid_init(c, active, ia, ib);
    while (VRC_WAVE_ANY(euclid_busy(c))) euclid_step(c);
    euclid_finish(c, ib, s_out, g_out);
}
// One table entry, decoded: theta = s / ib, g = gcd; g == 0: no entry, g == kJumpGcdEscape: the gcd did not fit (solve on the spot).
struct JumpEntry { double theta; int32_t g; };
constexpr int32_t kJumpGcdEscape = 2047;
VRC_HD double bits2d(uint32_t lo, uint32_t hi) { union { double d; uint64_t u; } c; c.u = ((uint64_t)hi << 32) | lo; return c.d; }
// 1 / b for an integer 1 <= b < 2^24 to ~2^-52 relative: hardware estimate + two Newton steps in fp64
VRC_HD double recip_dd(int32_t b) {
    const double bd = (double)b, r1 = recip_d(b);
    return __builtin_fma(r1, __builtin_fma(-bd, r1, 1.0), r1);
}
// theta = s / ib to ~2^-52 (rb = recip_dd(ib)): product + one residual correction.  What the probes compute from it are
// INTEGERS (pair_probe), and any theta within 2^-48 of s / ib gives the same ones: host and device need not agree on its last bits.
VRC_HD double jump_theta(int32_t s, int32_t ib, double rb) {
    const double sd = (double)s, q = sd * rb;
    return __builtin_fma(__builtin_fma(-q, (double)ib, sd), rb, q);
}
// The word: theta's bit pattern with g in the 11 bits theta does not need.  0 <= theta < 1 and theta >= 2^-24 unless it is 0, so
// the sign and the top five exponent bits are 0 01111 for every non-zero theta (biased exponents 999 .. 1022): bits 26-31 of the
// high dword hold g >> 5.  The low five mantissa bits hold g & 31: below 2^-48 for every theta < 1, which is the precision
// pair_probe needs (|c| < 2^23 and ib <= 2^23: |c| * ib * 2^-48 <= 1/4 of an integer).  theta == 0 decodes to ~2^-63: as good.
// g above 2046 (3 pairs in 10 000: the gcd of two "random" increments is k with probability 6 / (pi k)^2): the escape code --
// stretch_jump() solves such a pair afresh (as it does a pair whose row the ring has evicted).
VRC_HD JumpWord jump_entry_pack(int32_t s, int32_t g, int32_t ib, double rb) {
    union { double d; uint64_t u; } c;
    c.d = jump_theta(s, ib, rb);
    const uint32_t gc = (uint32_t)(g < kJumpGcdEscape ? g : kJumpGcdEscape);
    JumpWord w;
    w.lo = ((uint32_t)c.u & ~31u) | (gc & 31u);
    w.hi = ((uint32_t)(c.u >> 32) & 0x03ffffffu) | ((gc >> 5) << 26);
    return w;
}
VRC_HD JumpEntry jump_entry_unpack(JumpWord d) {
    JumpEntry en;
    en.g = (int32_t)(((d.hi >> 26) << 5) | (d.lo & 31u));
    en.theta = bits2d(d.lo, (d.hi & 0x03ffffffu) | 0x3c000000u);
    return en;
}

// Builds one row (binade kJumpFirstBinade + row, the three pairs) of the table of every lane with `active` set; the
// other lanes idle through the loop.  A pair one of whose axes cannot have a progression in that binade gets "no entry".
template <int R = kJumpRing>
VRC_HD void jump_table_build_row(bool active, int row, float dx, float dy, float dz, JumpWord *tab, int stride, uint32_t &solves) {
    const int32_t e = kJumpFirstBinade + row;
    const bool okx = active && jump_binade_ok(e, dx), oky = active && jump_binade_ok(e, dy), okz = active && jump_binade_ok(e, dz);
    bool hw;
    const int32_t ix = okx ? jump_axis_inc(e, dx, hw) : 64, iy = oky ? jump_axis_inc(e, dy, hw) : 64,
                  iz = okz ? jump_axis_inc(e, dz, hw) : 64;
    const bool vxy = okx && oky, vxz = okx && okz, vyz = oky && okz;
    EuclidChain cxy, cxz, cyz;
    euclid_init(cxy, vxy, ix, iy); euclid_init(cxz, vxz, ix, iz); euclid_init(cyz, vyz, iy, iz);
    while (VRC_WAVE_ANY(euclid_busy(cxy) || euclid_busy(cxz) || euclid_busy(cyz))) {
        euclid_step(cxy); euclid_step(cxz); euclid_step(cyz);
    }
    if (active) solves += (vxy ? 1u : 0u) + (vxz ? 1u : 0u) + (vyz ? 1u : 0u);
    if (active) {
        int32_t s, g;
        const int slot = row % R;
        const double ry = recip_dd(iy), rz = recip_dd(iz);        // the two moduli's reciprocals (xz and yz share iz)
        const JumpWord none = {0u, 0u};
        euclid_finish(cxy, iy, s, g); tab[(3 * slot + 0) * stride] = vxy ? jump_entry_pack(s, g, iy, ry) : none;
        euclid_finish(cxz, iz, s, g); tab[(3 * slot + 1) * stride] = vxz ? jump_entry_pack(s, g, iz, rz) : none;
        euclid_finish(cyz, iz, s, g); tab[(3 * slot + 2) * stride] = vyz ? jump_entry_pack(s, g, iz, rz) : none;
    }
}
// rows stretch_jump() will read for these intersection_t: one per axis pair that shares a binade inside the table
VRC_HD uint32_t jump_rows_needed(float tx, float ty, float tz) {
    const uint32_t rx = (f2u(tx) >> 23) - (uint32_t)kJumpFirstBinade, ry = (f2u(ty) >> 23) - (uint32_t)kJumpFirstBinade,
                   rz = (f2u(tz) >> 23) - (uint32_t)kJumpFirstBinade;
    uint32_t need = 0;
    need |= (rx == ry && rx < (uint32_t)kJumpBinades) ? (1u << rx) : 0u;
    need |= (rx == rz && rx < (uint32_t)kJumpBinades) ? (1u << rx) : 0u;
    need |= (ry == rz && ry < (uint32_t)kJumpBinades) ? (1u << ry) : 0u;
    return need;
}
// Rows are built on demand, wave-wide: `want` lanes are about to jump, `live` lanes have a ray that keeps its direction.
// While a wanting lane lacks a row it needs, that row is built for EVERY live lane that lacks it (the rays of a tile reach
// a binade within a few rounds of each other, and the Euclid loop costs the same for one lane as for 64).  `rows`: this
// lane's bit mask of built rows (cleared when the ray changes direction).
template <int R = kJumpRing>
VRC_HD void jump_rows_build(bool want, bool live, uint32_t &rows, float tx, float ty, float tz, float dx, float dy, float dz,
                            JumpWord *tab, int stride, uint32_t &solves) {
    uint32_t need = want ? (jump_rows_needed(tx, ty, tz) & ~rows) : 0u;
    // the lowest row this ray can still ask for: the binade of its smallest intersection_t (negative / tiny t: row 0)
    const float tmin = tx < ty ? (tx < tz ? tx : tz) : (ty < tz ? ty : tz);
    const int32_t lo_e = (int32_t)(f2u(tmin) >> 23);
    const int lo_row = (lo_e >= kJumpFirstBinade && lo_e < 256) ? lo_e - kJumpFirstBinade : 0;
    unsigned long long m;
    while ((m = VRC_BALLOT(need != 0u)) != 0ULL) {
        const uint32_t first = VRC_FIRST_LANE_VALUE(need, m);
        const int row = __builtin_ctz(first);
        // a lane that asked for the row gets it whatever it evicts (the loop must end; a pair whose row was evicted is
        // solved on the spot by stretch_jump); a lane that merely rides along takes it only if the row lies in the window of
        // kJumpRing rows from the binade it is in -- further ahead it would evict a row the ray still uses
        const bool build = live && !((rows >> row) & 1u) && (((need >> row) & 1u) || (row >= lo_row && row - lo_row < R));
        jump_table_build_row<R>(build, row, dx, dy, dz, tab, stride, solves);
        if (build) rows = (rows & ~(JumpRingMask<R>::value << (row % R))) | (1u << row);   // the rows that shared its ring slot are gone
        need &= ~rows;
    }
}
// The packed entry of (pair, binade e), 0 for a row the table does not hold ("no entry").  The load is unconditional -- the ring
// slot row % R always exists -- and the word stays packed until the probes need it: a guarded load with the decode right behind
// it made every jump wait for three LDS round trips one after the other before its arithmetic started.  (A scheduling barrier
// behind the three loads, so that none is sunk to its use: headline the same, the multi-light instances 3 % slower.)
template <int R = kJumpRing>
VRC_HD JumpWord jump_table_word(const JumpWord *tab, int stride, uint32_t rows, int pair, int32_t e) {
    const uint32_t row = (uint32_t)(e - kJumpFirstBinade);
    const bool have = row < (uint32_t)kJumpBinades && ((rows >> (row & 31u)) & 1u);
    JumpWord d = tab[(3 * (int)(row % (uint32_t)R) + pair) * stride];
    d.lo = have ? d.lo : 0u; d.hi = have ? d.hi : 0u;
    return d;
}

// One regular axis pair in a common binade: the consumed events are Ma + i*ia (0 <= i < ma) and Mb + j*ib (0 <= j < mb),
// equal where i*ia - j*ib == c := Mb - Ma.  With the table's s*ia == g (mod ib): w = (c*s) mod ib equals g * i0 whenever g
// divides c, i0 the smallest solution index -- so no solution lies below ma unless w < g * ma.
// pair_probe() is the hot half: w = ib * fract(c * theta) with the row's theta = s / ib -- a conversion, one fp64 product, a
// v_fract_f64, one fma and a conversion back, no reciprocal and no floor / remainder fix-up (theta is good to 2^-48, |c| < 2^23, ib
// <= 2^23: the product before rounding is within 0.27 of the integer w).  pair_count() is the warm half, entered only when a
// solution index may lie inside the stretch.
VRC_HD double fract_d(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fract(x);                             // v_fract_f64: x - floor(x), never 1.0
#else
    const double f = x - __builtin_floor(x);
    return f < 1.0 ? f : 0x1.fffffffffffffp-1;
#endif
}
struct PairProbe { int32_t w, g; bool may; };
VRC_HD PairProbe pair_probe(bool active, JumpEntry entry, int32_t Ma, int32_t ma, int32_t Mb, int32_t ib) {
    PairProbe p;
    p.g = active ? entry.g : 1;
    const uint32_t m = (uint32_t)(active ? ib : 64);
    const double f = fract_d((double)(Mb - Ma) * entry.theta);    // w / m, within 2^-25 (in [0, 1))
    uint32_t r = (uint32_t)(int32_t)__builtin_fma(f, (double)m, 0.5);   // the nearest integer: 0 .. m
    r = r < r - m ? r : r - m;                                    // (m -> 0: unsigned minimum of r and r - m)
    p.w = (int32_t)r;
    p.may = active && r < (uint32_t)ma * (uint32_t)p.g;           // (ma * g <= 2^23 + g: the consumed events lie in one binade)
    return p;
}
// number of equal values; first_i / step_i / the count describe them as i = first_i + k * step_i
struct PairTies { int32_t count, first_i, step_i; };
// The usual tie case, decided on the hot path without a division or a vote: coprime increments and a stretch shorter than one
// period of the solutions (ma <= ib and mb <= ia) -- i0 = w is the only candidate, and it counts when its partner index
// j0 = (i0*ia - c) / ib lies in 0 .. mb-1.  j0 * ib = i0 * ia - c holds exactly (the congruence: g == 1 divides every c), so the test
// is two products and two compares; i0 < ma and mb - 1 both index events inside the binade: the products stay below 2^24.
// `general`: a solution index may lie inside the stretch and the case is not the usual one (pair_count_general, behind a vote).
VRC_HD int32_t pair_count_quick(const PairProbe &q, int32_t Ma, int32_t ia, int32_t ma, int32_t Mb, int32_t ib, int32_t mb, bool &general) {
    const bool simple = (q.g == 1) & (ma <= ib) & (mb <= ia);
    general = q.may & !simple;
    const int32_t v = mul24(q.w, ia) - (Mb - Ma);                 // (meaningful for may & simple: w < ma)
    return (q.may & simple & (v >= 0) & (v <= mul24((mb > 0 ? mb : 1) - 1, ib))) ? 1 : 0;
}
// The general count: a gcd above 1, or a stretch longer than one period of the solutions (in the terms below the usual case has
// k_a = 0, k_b = 0 or negative, k_lo = 0 or positive).  Lanes without q.may idle through it.
VRC_HD PairTies pair_count_general(const PairProbe &p, int32_t Ma, int32_t ia, int32_t ma, int32_t Mb, int32_t ib, int32_t mb) {
    PairTies out;
    out.count = 0; out.first_i = 0; out.step_i = 1;
    if (p.may) {
        const int32_t c = Mb - Ma, g = p.g;
        const double rg = recip_d(g);
        int32_t rem, r2;
        const int32_t cq = floordiv(c, g, rg, rem);               // g must divide c
        if (rem == 0) {
            const int32_t i0 = floordiv(p.w, g, rg, r2), iar = floordiv(ia, g, rg, r2), ibr = floordiv(ib, g, rg, r2);
            // i = i0 + k*ibr pairs with j = j0 + k*iar, j0 = (i0*iar - cq) / ibr (exact; i0 < ma keeps i0*iar below 2^24)
            const double rbr = recip_d(ibr), rar = recip_d(iar);
            const int32_t j0 = floordiv(i0 * iar - cq, ibr, rbr, r2);
            const int32_t k_lo = (j0 < 0) ? -floordiv(j0, iar, rar, r2) : 0;        // ceil(-j0 / iar)
            const int32_t k_a = floordiv(ma - 1 - i0, ibr, rbr, r2);                 // >= 0
            const int32_t k_b = floordiv(mb - 1 - j0, iar, rar, r2);                 // < 0 when j0 > mb - 1
            const int32_t k_hi = k_a < k_b ? k_a : k_b;
            if (k_hi >= k_lo) { out.count = k_hi - k_lo + 1; out.first_i = i0 + k_lo * ibr; out.step_i = ibr; }
        }
    }
    return out;
}
// both halves for one pair (the host harness checks them against merged progressions; stretch_jump interleaves the three pairs)
VRC_HD PairTies pair_count(const PairProbe &q, int32_t Ma, int32_t ia, int32_t ma, int32_t Mb, int32_t ib, int32_t mb) {
    bool general;
    PairTies out;
    out.count = pair_count_quick(q, Ma, ia, ma, Mb, ib, mb, general); out.first_i = q.w; out.step_i = ib;
    if (VRC_WAVE_ANY(general)) { PairProbe g = q; g.may = general; const PairTies t = pair_count_general(g, Ma, ia, ma, Mb, ib, mb); if (general) out = t; }
    return out;
}
// a pair without a usable table entry -- below t = 128, beyond the table, evicted from the ring, a gcd the word cannot hold -- is
// solved afresh (rare; behind a vote)
VRC_HD bool jump_entry_missing(bool active, const JumpEntry &e) { return active & ((e.g == 0) | (e.g == kJumpGcdEscape)); }
VRC_HD void jump_entry_solve(bool solve, JumpEntry &e, int32_t ia, int32_t ib) {
#if defined(VRC_SCHED_STATS) && defined(__HIP_DEVICE_COMPILE__)
    if (solve) atomicAdd(&g_jump_private_solves[0], 1ULL);
#endif
    if (VRC_WAVE_ANY(solve)) {
        int32_t s2, g2;
        pair_solve(solve, solve ? ia : 64, solve ? ib : 64, s2, g2);
        if (solve) { e.theta = jump_theta(s2, ib, recip_dd(ib)); e.g = g2; }
    }
}

struct JumpAxis {
    int32_t tb;        // bit pattern of t
    int32_t inc;       // increment of the bit pattern per step
    int32_t c;         // last usable event index (0 for a frozen axis)
    int32_t e;         // biased exponent field (>= 256 for negative t)
    int32_t m;         // events consumed
    float E;           // value of event c
    float last;        // value of the last consumed event
    bool reg;          // the progression is valid (not frozen)
    bool hitX;         // m > 0 and the last consumed value is X
};

VRC_HD JumpAxis jump_axis(float t, float d, int32_t n) {
    JumpAxis a;
    a.tb = (int32_t)f2u(t);
    a.e = (int32_t)((uint32_t)a.tb >> 23);
    bool reg = jump_binade_ok(a.e, d), halfway;
    const int32_t ec = reg ? a.e : 64;
    a.inc = jump_axis_inc(ec, reg ? d : 1.0f, halfway);
    reg = reg && !(halfway && (a.tb & 1));                        // half-way case: needs an even mantissa (one real step settles it)
#ifndef VRC_JUMP_ROOM
    // (round 4; -DVRC_JUMP_ROOM: the form of round 3, with a division per axis for the steps the binade has room for)
    // Does the value at the node face (n - 1 steps on) still lie in this binade?  An exact 48-bit test instead of a division: if it
    // does, that value is the axis' bound E; if not, E is the LARGEST FLOAT OF THE BINADE -- not an event of the axis, only a bound:
    // every event of the axis up to it is consumed (jump_count gets their number by its one division) and the axis' next event
    // lies in the next binade, beyond every value the stretch consumes.
    // (a frozen axis takes no step: lo = 0, the test passes and E = t, c = 0 come out of the same selects -- written as nested
    // conditionals the compiler made three exec-masked branches per axis out of them)
    const uint32_t mant = (uint32_t)a.tb & 0x7fffffu, steps = reg ? (uint32_t)(n - 1) : 0u, uinc = reg ? (uint32_t)a.inc : 64u;
#if defined(__HIP_DEVICE_COMPILE__)
    // (a full 32-bit low word, not __umul24: a countdown may reach 2^24 -- the top-level node of a depth-24 tree widened over a
    // sibling, an empty box across a 2^19 map -- and a truncated low word under a zero high word would pass the test below)
    const uint32_t lo = steps * uinc, hi = __umulhi(steps, uinc);
#else
    const uint64_t p64 = (uint64_t)steps * uinc;
    const uint32_t lo = (uint32_t)p64, hi = (uint32_t)(p64 >> 32);
#endif
    const bool fits = hi == 0u && lo <= 0x7fffffu - mant;
    a.reg = reg;
    a.c = fits ? (int32_t)steps : -1;                             // -1: E is the binade's end, the event count comes from jump_count's division
    a.E = u2f(fits ? (uint32_t)a.tb + lo : ((uint32_t)a.tb | 0x7fffffu));
#else
    // steps that stay inside the binade: floor((2^24 - 1 - M) / inc), exact (a short estimate would cost a whole extra jump)
    const int32_t room = floordiv_small(0x7fffff - (a.tb & 0x7fffff), reg ? a.inc : 64);
    a.reg = reg;
    a.c = reg ? (n - 1 < room ? n - 1 : room) : 0;
    a.E = reg ? u2f((uint32_t)(a.tb + mul24(a.c, a.inc))) : t;
#endif
    a.m = 0; a.last = t; a.hitX = false;
    return a;
}

// events of the axis with value <= X, where X <= a.E
VRC_HD void jump_count(JumpAxis &a, float t, float X) {
    const int32_t diff = (int32_t)f2u(X) - a.tb;                  // meaningful for t <= X < E: 0 <= diff < c*inc < 2^24
#ifndef VRC_JUMP_ROOM
    const bool mid = a.reg & (X >= t) & ((X < a.E) | (a.c < 0));  // (X == E with E the binade's end: counted, not known; no short circuits: they compile to branches)
    const int32_t k = floordiv_small(mid ? diff : 0, mid ? a.inc : 64);
    int32_t m = mid ? k + 1 : a.c + 1;
#else
    const bool mid = a.reg && X >= t && X < a.E;
    const int32_t k = floordiv_small(mid ? diff : 0, mid ? a.inc : 64);
    int32_t m = (X == a.E) ? a.c + 1 : k + 1;
#endif
    m = (X >= t) ? m : 0;
    a.m = m;
    a.last = u2f((uint32_t)(a.tb + mul24((a.reg & (m > 0)) ? m - 1 : 0, a.inc)));   // (t itself when nothing was consumed)
    a.hitX = m > 0 && a.last == X;
}

// upper bound one axis puts on the end of a stretch, for the estimate the kernel makes before it decides to jump: the
// next binade end of t, or t itself while the axis has not had its first crossing (t below delta_t: frozen)
VRC_HD float jump_axis_limit(float t, float d) {
    return t < d ? t : u2f((f2u(t) & 0x7f800000u) + 0x00800000u);
}

struct JumpOut {
    int32_t iterations;    // loop iterations consumed (always >= 1)
    bool left_node;        // the last consumed iteration zeroed a countdown (node exit / lookup pending)
    bool capped;           // the step cap was reached before that iteration: the ray's loop has ended
    float fx, fy, fz;      // face mask of the last consumed iteration as 0.0 / 1.0
};

// One exact multi-iteration jump.  t*, n* (countdowns >= 1) are updated in place.  `left` = iterations the loop may
// still run (max_distance - distance_traveled, >= 1).  `tab`, `stride`, `rows`: the ray's table (jump_rows_build).
template <int R = kJumpRing>
VRC_HD JumpOut stretch_jump(float &tx, float &ty, float &tz, float dx, float dy, float dz, int32_t &nx, int32_t &ny,
                            int32_t &nz, int32_t left, const JumpWord *tab, int stride, uint32_t rows) {
    // the table dwords first: their addresses need the exponents only, and the loads have the whole decode to arrive
    const int32_t ex0 = (int32_t)(f2u(tx) >> 23), ey0 = (int32_t)(f2u(ty) >> 23), ez0 = (int32_t)(f2u(tz) >> 23);
    const JumpWord txy = jump_table_word<R>(tab, stride, ex0 == ey0 ? rows : 0u, 0, ex0);
    const JumpWord txz = jump_table_word<R>(tab, stride, ex0 == ez0 ? rows : 0u, 1, ex0);
    const JumpWord tyz = jump_table_word<R>(tab, stride, ey0 == ez0 ? rows : 0u, 2, ey0);
    JumpAxis ax = jump_axis(tx, dx, nx), ay = jump_axis(ty, dy, ny), az = jump_axis(tz, dz, nz);
    float X = ax.E < ay.E ? ax.E : ay.E;
    X = X < az.E ? X : az.E;
    jump_count(ax, tx, X); jump_count(ay, ty, X); jump_count(az, tz, X);

    // iterations = distinct values among the consumed events = sum - pair ties + triple ties.  Two regular axes of one
    // binade are two integer progressions (congruence); any other pair can only meet in the value X itself.
    const bool gxy = ax.reg && ay.reg && ax.e == ay.e, gxz = ax.reg && az.reg && ax.e == az.e, gyz = ay.reg && az.reg && ay.e == az.e;
    const bool nxy = gxy && ax.m > 0 && ay.m > 0, nxz = gxz && ax.m > 0 && az.m > 0, nyz = gyz && ay.m > 0 && az.m > 0;
    int32_t ties = 0, triple = 0;
    const int32_t Mx = ax.tb & 0x7fffff, My = ay.tb & 0x7fffff, Mz = az.tb & 0x7fffff;   // (the hidden bit cancels in differences)
    PairTies pxy;
    pxy.count = 0; pxy.first_i = 0; pxy.step_i = 1;
#ifndef VRC_JUMP_NO_TIES   // (timing experiment only: wrong iteration counts)
    {
        // The three probes side by side -- each a conversion, an fp64 product, a fract, an fma: three independent chains -- and the
        // usual tie case settled by pair_count_quick; votes only for what is rare: a pair without a usable entry, a tie case with a
        // gcd or a long stretch.  (Rounds 3-5 took one pair after the other, each behind its own votes, "for the registers' sake":
        // with the reciprocal chain gone from the probe the flat form needs no more registers, and the multi-light instances --
        // whose spills it happened to move out of the round loop -- render 4 lights in 3.19 instead of 3.59 ms.)
        JumpEntry exy = jump_entry_unpack(txy), exz = jump_entry_unpack(txz), eyz = jump_entry_unpack(tyz);
        const bool sxy = jump_entry_missing(nxy, exy), sxz = jump_entry_missing(nxz, exz), syz = jump_entry_missing(nyz, eyz);
        if (VRC_WAVE_ANY(sxy | sxz | syz)) {
            jump_entry_solve(sxy, exy, ax.inc, ay.inc); jump_entry_solve(sxz, exz, ax.inc, az.inc); jump_entry_solve(syz, eyz, ay.inc, az.inc);
        }
        const PairProbe qxy = pair_probe(nxy, exy, Mx, ax.m, My, ay.inc), qxz = pair_probe(nxz, exz, Mx, ax.m, Mz, az.inc),
                        qyz = pair_probe(nyz, eyz, My, ay.m, Mz, az.inc);
        bool gen_xy, gen_xz, gen_yz;
        pxy.count = pair_count_quick(qxy, Mx, ax.inc, ax.m, My, ay.inc, ay.m, gen_xy); pxy.first_i = qxy.w; pxy.step_i = ay.inc;
        int32_t cxz = pair_count_quick(qxz, Mx, ax.inc, ax.m, Mz, az.inc, az.m, gen_xz);
        int32_t cyz = pair_count_quick(qyz, My, ay.inc, ay.m, Mz, az.inc, az.m, gen_yz);
#if defined(VRC_SCHED_STATS) && defined(__HIP_DEVICE_COMPILE__)
        atomicAdd(&g_jump_private_solves[2], (unsigned long long)((qxy.may ? 1 : 0) + (qxz.may ? 1 : 0) + (qyz.may ? 1 : 0)));
        if (gen_xy | gen_xz | gen_yz) atomicAdd(&g_jump_private_solves[3], 1ULL);
#endif
        if (VRC_WAVE_ANY(gen_xy | gen_xz | gen_yz)) {
            if (VRC_WAVE_ANY(gen_xy)) { PairProbe q = qxy; q.may = gen_xy; const PairTies t = pair_count_general(q, Mx, ax.inc, ax.m, My, ay.inc, ay.m); if (gen_xy) pxy = t; }
            if (VRC_WAVE_ANY(gen_xz)) { PairProbe q = qxz; q.may = gen_xz; const PairTies t = pair_count_general(q, Mx, ax.inc, ax.m, Mz, az.inc, az.m); if (gen_xz) cxz = t.count; }
            if (VRC_WAVE_ANY(gen_yz)) { PairProbe q = qyz; q.may = gen_yz; const PairTies t = pair_count_general(q, My, ay.inc, ay.m, Mz, az.inc, az.m); if (gen_yz) cyz = t.count; }
        }
        ties += pxy.count + cxz + cyz;
    }
#endif
    // pairs outside the congruence: both last values equal X
    ties += (!gxy && ax.hitX && ay.hitX) ? 1 : 0;
    ties += (!gxz && ax.hitX && az.hitX) ? 1 : 0;
    ties += (!gyz && ay.hitX && az.hitX) ? 1 : 0;
    // values common to all three axes
    const bool g3 = gxy && gxz;
    triple = (!g3 && ax.hitX && ay.hitX && az.hitX) ? 1 : 0;
    const bool cold3 = g3 && pxy.count > 0 && az.m > 0;
    if (VRC_WAVE_ANY(cold3)) {
        if (cold3) {
            const double rz = recip_d(az.inc);
            for (int32_t k = 0, i = pxy.first_i; k < pxy.count; k++, i += pxy.step_i) {
                const int32_t v = Mx + i * ax.inc - Mz;
                if (v >= 0) {
                    int32_t rem;
                    const int32_t qz = floordiv(v, az.inc, rz, rem);
                    if (rem == 0 && qz < az.m) triple++;
                }
            }
        }
    }

    JumpOut res;
    res.iterations = ax.m + ay.m + az.m - ties + triple;
    res.capped = false; res.left_node = false;
    res.fx = ax.hitX ? 1.0f : 0.0f; res.fy = ay.hitX ? 1.0f : 0.0f; res.fz = az.hitX ? 1.0f : 0.0f;
    if (res.iterations > left) {                                  // :357 the cap ends the loop inside this stretch
        res.iterations = left;
        res.capped = true;
        return res;
    }
    // the last step of each axis is one real float add (it may leave the binade)
    if (ax.m > 0) tx = ax.last + dx;
    if (ay.m > 0) ty = ay.last + dy;
    if (az.m > 0) tz = az.last + dz;
    nx -= ax.m; ny -= ay.m; nz -= az.m;
    res.left_node = (nx == 0) || (ny == 0) || (nz == 0);
    return res;
}

}  // namespace vrc
