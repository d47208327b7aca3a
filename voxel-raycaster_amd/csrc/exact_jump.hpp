// exact_jump.hpp -- O(1) emulation of the reference's DDA float recurrence across an empty node.
//
// The reference advances a ray with (kernels/ray_caster_kernel.cl:558-560)
//     face_mask = (t.xyz <= min(t.yzx, t.zxy));   t += delta_t * face_mask;   voxel += step * face_mask;
// once per loop iteration, and the iteration count feeds the fog factor (:716) and the shadow-ray cap
// (:667).  Bit-exact parity therefore needs the k-fold float accumulation of t AND the exact number of
// iterations, including iterations in which two or three axes tie and step together.
//
// Inside one binade [2^e, 2^(e+1)) a float t is an integer mantissa M times u = 2^(e-23), and
// RNE(t + d) = t + inc*u with a constant integer inc (d rounded to a multiple of u; when d sits exactly
// half-way the increment is the even neighbour once M is even).  So each axis is an exact arithmetic
// progression of integers until it leaves its binade, and
//   * the value after k steps,
//   * the number of steps with value <= X,
//   * the number of EQUAL values of two axes (a linear congruence, solved with one modular inverse
//     per axis pair and binade)
// are all closed forms.  try_jump() consumes every iteration up to the one that leaves the node -- or up to
// (just before) the first binade boundary of any axis, whichever comes first -- and reports how many loop
// iterations that was.  Anything outside the simple regime (t <= 0, t < d, unsettled half-way case) is
// left to the ordinary step loop, which is always correct; so is any jump whose integer estimates fail
// their own validity check.
//
// Host+device header: tools/jumptest/jump_vs_loop.cpp drives the host build against the plain loop on
// millions of random states (tests/test_exact_jump.py); raycast_kernel.hip uses the device build.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define VRC_HD __host__ __device__ __forceinline__
#define VRC_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define VRC_HD inline
#define VRC_HD_NOINLINE inline
#endif

namespace vrc {

VRC_HD uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
VRC_HD float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
// float with biased exponent e and 24-bit mantissa M (hidden bit included)
VRC_HD float mk_float(int32_t e, int32_t M) { return u2f(((uint32_t)e << 23) | ((uint32_t)M & 0x7fffffu)); }

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

// floor(a / b), a mod b for 0 <= a < 2^25, 1 <= b < 2^24.  Reciprocal estimate, then the exact integer
// remainder corrected by up to two units each way, branch-free.  `ok` is cleared if that was not enough
// (tiny b with huge quotient): the caller then refuses the jump.
VRC_HD int32_t idivmod(int32_t a, int32_t b, int32_t &rem, bool &ok) {
    int32_t q = (int32_t)((float)a * fast_rcp((float)b));
    int32_t r = a - q * b;
    int32_t adj = (r < 0) ? -1 : 0;  q += adj; r -= adj * b;
    adj = (r < 0) ? -1 : 0;          q += adj; r -= adj * b;
    adj = (r >= b) ? 1 : 0;          q += adj; r -= adj * b;
    adj = (r >= b) ? 1 : 0;          q += adj; r -= adj * b;
    ok = ok && (r >= 0) && (r < b);
    rem = r;
    return q;
}
// exact variants with unbounded correction (cold paths only)
VRC_HD int32_t idivmod_loop(int32_t a, int32_t b, int32_t &rem) {
    int32_t q = (int32_t)((float)a * fast_rcp((float)b));
    int32_t r = a - q * b;
    while (r < 0) { q--; r += b; }
    while (r >= b) { q++; r -= b; }
    rem = r;
    return q;
}
VRC_HD int32_t idiv_loop(int32_t a, int32_t b) { int32_t r; return idivmod_loop(a, b, r); }
VRC_HD int32_t imod_loop(int32_t a, int32_t b) {                 // |a| < 2^25, result in [0, b)
    int32_t r;
    if (a >= 0) { idivmod_loop(a, b, r); return r; }
    idivmod_loop(-a, b, r);
    return r ? b - r : 0;
}

// (x * y) mod m for 0 <= x, y < m < 2^24: the 48-bit product is exact in fp64; the quotient is estimated in
// fp32 and the remainder corrected by up to two multiples of m each way
VRC_HD int32_t mulmod(int32_t x, int32_t y, int32_t m, bool &ok) {
    const double prod = (double)x * (double)y;
    const double dm = (double)m;
    const double q = (double)__builtin_truncf((float)prod * fast_rcp((float)m));
    double r = __builtin_fma(-q, dm, prod);                       // exact
    r += (r < 0.0) ? dm : 0.0;
    r += (r < 0.0) ? dm : 0.0;
    r -= (r >= dm) ? dm : 0.0;
    r -= (r >= dm) ? dm : 0.0;
    ok = ok && (r >= 0.0) && (r < dm);
    return (int32_t)r;
}

struct AxisProg {          // one axis inside its current binade
    int32_t M;             // mantissa of t in units of u (2^23 <= M < 2^24)
    int32_t inc;           // exact increment per step, in units of u
    int32_t e;             // biased exponent field of t
    int32_t room;          // LOWER BOUND on the steps j for which M + j*inc stays below 2^24
    bool ok;               // axis is in the simple regime
};

VRC_HD AxisProg make_prog(float t, float d) {
    AxisProg p;
    const uint32_t tb = f2u(t), db = f2u(d);
    p.e = (int32_t)(tb >> 23);                                    // sign bit set => e >= 256 => rejected below
    const int32_t ed = (int32_t)(db >> 23);
    p.M = (int32_t)((tb & 0x7fffffu) | 0x800000u);
    const int32_t Md = (int32_t)((db & 0x7fffffu) | 0x800000u);
    const int32_t sh = p.e - ed;
    p.ok = (p.e >= 1 && p.e <= 254 && ed >= 1 && ed <= 254 && sh >= 0 && sh <= 23);
    const int32_t shc = p.ok ? sh : 0;
    const int32_t D = Md >> shc, rem = Md & ((1 << shc) - 1), half = (shc > 0) ? (1 << (shc - 1)) : 0;
    const bool halfway = shc > 0 && rem == half;                  // d exactly half-way between two multiples of u
    p.ok = p.ok && !(halfway && (p.M & 1));                       // ... needs an even M (one ordinary step settles it)
    p.inc = halfway ? D + (D & 1) : D + ((shc > 0 && rem > half) ? 1 : 0);
    // conservative estimate of floor((2^24 - 1 - M) / inc): never too large, at most a couple too small
    const float est = (float)(0xffffff - p.M) * fast_rcp((float)p.inc) * 0.999999f;
    const int32_t rl = (int32_t)est - 1;
    p.room = rl > 0 ? rl : 0;
    return p;
}

// number of j in [0, cap] with value_j <= X, value_j = (M + j*inc) * 2^(e-23)
VRC_HD int32_t count_le(const AxisProg &p, int32_t cap, float X, bool &ok) {
    const uint32_t xb = f2u(X);
    const int32_t ex = (int32_t)(xb >> 23);                       // X > 0 here
    const int32_t Mx = (int32_t)((xb & 0x7fffffu) | 0x800000u);
    int32_t rem;
    const int32_t diff = Mx - p.M;
    const int32_t j = idivmod(diff > 0 ? diff : 0, p.inc, rem, ok);
    int32_t n = (j < cap ? j : cap) + 1;
    n = (diff < 0) ? 0 : n;
    n = (ex > p.e) ? cap + 1 : n;
    n = (ex < p.e) ? 0 : n;
    return n;
}

// (g, inverse of a/g modulo b/g) by the extended Euclidean algorithm; 1 <= a, b < 2^24.
// Runs once per axis pair and binade: kept out of line so the hot jump path stays small.
VRC_HD_NOINLINE void gcd_inverse(int32_t a, int32_t b, int32_t &g, int32_t &inv) {
    int32_t r0 = b, r1 = imod_loop(a, b), s0 = 0, s1 = 1;         // invariant: r_i == s_i * a (mod b)
    while (r1 != 0) {
        int32_t r2;
        const int32_t q = idivmod_loop(r0, r1, r2);
        const int32_t s2 = s0 - q * s1;
        r0 = r1; r1 = r2; s0 = s1; s1 = s2;
    }
    g = r0;                                                       // gcd(a, b); s0 * a == g (mod b)
    const int32_t bg = (g == 1) ? b : idiv_loop(b, g);
    inv = imod_loop(s0, bg);                                      // (a/g) * inv == 1 (mod b/g)
}

// General (cold) tie count: pairs (i, j), 0 <= i < la, 0 <= j < lb, with Ma + i*ia == Mb + j*ib.
// first_i / step_i describe the solutions in i (first_i = -1 when there is none).
VRC_HD_NOINLINE int32_t count_ties_general(int32_t Ma, int32_t ia, int32_t la, int32_t Mb, int32_t ib, int32_t lb,
                                           int32_t *first_i, int32_t *step_i) {
    *first_i = -1; *step_i = 1;
    if (la <= 0 || lb <= 0) return 0;
    int32_t g, inv;
    gcd_inverse(ia, ib, g, inv);
    int32_t c = Mb - Ma;                                          // need i*ia - j*ib == c
    int32_t ibg = ib, iag = ia;
    if (g != 1) {
        int32_t rem;
        const int32_t cq = idivmod_loop(c < 0 ? -c : c, g, rem);
        if (rem != 0) return 0;
        c = c < 0 ? -cq : cq;
        ibg = idiv_loop(ib, g);
        iag = idiv_loop(ia, g);
    }
    bool ok = true;
    int32_t i0 = mulmod(imod_loop(c, ibg), inv, ibg, ok);
    if (!ok) {                                                    // cannot happen for m < 2^24; keep exactness anyway
        const uint64_t pr = (uint64_t)(uint32_t)imod_loop(c, ibg) * (uint64_t)(uint32_t)inv;
        i0 = (int32_t)(pr % (uint32_t)ibg);
    }
    int32_t lo = 0;
    if (c > 0) lo = idiv_loop(c + iag - 1, iag);                  // i*iag >= c
    const int32_t first = lo + imod_loop(i0 - lo, ibg);           // smallest i >= lo with i == i0 (mod ibg)
    if (first >= la) return 0;
    const int32_t top = c + (lb - 1) * ibg;                       // i*iag <= c + (lb-1)*ibg
    if (top < 0) return 0;
    int32_t hi = idiv_loop(top, iag);
    if (hi > la - 1) hi = la - 1;
    if (first > hi) return 0;
    *first_i = first; *step_i = ibg;
    return idiv_loop(hi - first, ibg) + 1;
}

// Cold path of try_jump: exact number of pair ties minus triple ties over the consumed events.
VRC_HD_NOINLINE int32_t ties_minus_triples(int32_t ex, int32_t Mx, int32_t ix, int32_t mx, int32_t ey, int32_t My, int32_t iy,
                                           int32_t my, int32_t ez, int32_t Mz, int32_t iz, int32_t mz) {
    int32_t ties = 0, triple = 0, first, step;
    if (ex == ey) {
        const int32_t t_xy = count_ties_general(Mx, ix, mx, My, iy, my, &first, &step);
        ties += t_xy;
        if (t_xy > 0 && ez == ex && mz > 0) {                     // values common to all three axes
            for (int32_t k = 0, i = first; k < t_xy; k++, i += step) {
                const int32_t v = Mx + i * ix - Mz;
                if (v >= 0) {
                    int32_t rem;
                    const int32_t qz = idivmod_loop(v, iz, rem);
                    if (rem == 0 && qz < mz) triple++;
                }
            }
        }
    }
    if (ex == ez) ties += count_ties_general(Mx, ix, mx, Mz, iz, mz, &first, &step);
    if (ey == ez) ties += count_ties_general(My, iy, my, Mz, iz, mz, &first, &step);
    return ties - triple;
}

// Per axis pair and binade `key`: inverse of inc_a modulo inc_b and of inc_b modulo inc_a, valid when the two
// increments are coprime (flag); the increments depend on (delta_t, binade) only, so the cache holds
// until the ray changes direction (jump_cache_reset).
struct PairCache { int32_t key, inv_ab, inv_ba; };               // key = binade | coprime << 16, -1 = empty
struct JumpCache { PairCache xy, xz, yz; };
VRC_HD void jump_cache_reset(JumpCache &c) { c.xy.key = c.xz.key = c.yz.key = -1; }

VRC_HD_NOINLINE void pair_fill(PairCache &pc, int32_t e, int32_t ia, int32_t ib) {
    int32_t g, inv;
    gcd_inverse(ia, ib, g, inv);
    pc.inv_ab = inv;
    int32_t g2 = 1, inv2 = 0;
    if (g == 1) gcd_inverse(ib, ia, g2, inv2);
    pc.inv_ba = inv2;
    pc.key = e | ((g == 1) ? (1 << 16) : 0);
}

// Hot-path question for one axis pair in a common binade: can two consumed events be equal at all?
// Returns false when certainly not (the overwhelmingly common case), true when the cold path must count.
VRC_HD bool pair_may_tie(PairCache &pc, int32_t e, int32_t Ma, int32_t ia, int32_t la, int32_t Mb, int32_t ib, int32_t lb,
                         bool &ok) {
    if ((pc.key & 0xffff) != e || pc.key < 0) pair_fill(pc, e, ia, ib);
    if (!(pc.key >> 16)) return true;                             // increments share a factor: let the cold path decide
    // orient so that c <= 0: the progression that starts higher is indexed by i, i*inc_i - j*inc_j == c
    const int32_t c = Mb - Ma;
    const bool sw = c > 0;
    const int32_t cc = sw ? -c : c;                               // <= 0
    const int32_t ij = sw ? ia : ib, li = sw ? lb : la;
    const int32_t inv = sw ? pc.inv_ba : pc.inv_ab;               // inverse of inc_i modulo inc_j
    int32_t cm;
    idivmod(-cc, ij, cm, ok);                                     // (-cc) mod ij
    cm = cm ? ij - cm : 0;                                        // cc mod ij
    const int32_t i0 = mulmod(cm, inv, ij, ok);                   // smallest i >= 0 with i*inc_i == cc (mod inc_j)
    return i0 < li;                                               // a solution index inside the stretch: count exactly
}

struct JumpResult {
    int32_t iterations;    // loop iterations consumed (0 = no jump taken)
    bool left_node;        // the last consumed iteration zeroed a countdown (node exit / lookup pending)
    bool capped;           // the step cap was reached before that iteration: the ray's loop has ended
    int32_t fx, fy, fz;    // face mask of the last consumed iteration (valid when left_node)
};

// One exact multi-iteration jump.  t*, n* (countdowns >= 1) are updated in place.  `left` = iterations the
// loop may still run (max_distance - distance_traveled, >= 1).
VRC_HD JumpResult try_jump(float &tx, float &ty, float &tz, float dx, float dy, float dz,
                           int32_t &nx, int32_t &ny, int32_t &nz, int32_t left, JumpCache &cache) {
    JumpResult res;
    res.iterations = 0; res.left_node = false; res.capped = false; res.fx = res.fy = res.fz = 0;
    const AxisProg px = make_prog(tx, dx), py = make_prog(ty, dy), pz = make_prog(tz, dz);
    if (!(px.ok && py.ok && pz.ok)) return res;
    bool ok = true;

    // last usable event index per axis: the node face (n-1) or (a lower bound of) the end of the binade
    const int32_t cx = (nx - 1 < px.room) ? nx - 1 : px.room;
    const int32_t cy = (ny - 1 < py.room) ? ny - 1 : py.room;
    const int32_t cz = (nz - 1 < pz.room) ? nz - 1 : pz.room;
    const float bxv = mk_float(px.e, px.M + cx * px.inc);
    const float byv = mk_float(py.e, py.M + cy * py.inc);
    const float bzv = mk_float(pz.e, pz.M + cz * pz.inc);
    float X = bxv < byv ? bxv : byv;
    X = X < bzv ? X : bzv;

    // events consumed per axis: all values <= X
    const int32_t mx = count_le(px, cx, X, ok), my = count_le(py, cy, X, ok), mz = count_le(pz, cz, X, ok);

    // iterations = distinct values among the consumed events = sum - pair ties + triple ties
    bool may = false;
    if (px.e == py.e && mx > 0 && my > 0) may = may | pair_may_tie(cache.xy, px.e, px.M, px.inc, mx, py.M, py.inc, my, ok);
    if (px.e == pz.e && mx > 0 && mz > 0) may = may | pair_may_tie(cache.xz, px.e, px.M, px.inc, mx, pz.M, pz.inc, mz, ok);
    if (py.e == pz.e && my > 0 && mz > 0) may = may | pair_may_tie(cache.yz, py.e, py.M, py.inc, my, pz.M, pz.inc, mz, ok);
    if (!ok) return res;                                          // an estimate failed its own check: no jump
    int32_t iters = mx + my + mz;
    if (may) iters -= ties_minus_triples(px.e, px.M, px.inc, mx, py.e, py.M, py.inc, my, pz.e, pz.M, pz.inc, mz);

    if (iters > left) {                                           // :357 the cap ends the loop inside this stretch
        res.iterations = left;
        res.capped = true;
        return res;
    }

    // face mask of the last iteration: axes whose last consumed value is X
    const bool lx = mx > 0 && mk_float(px.e, px.M + (mx - 1) * px.inc) == X;
    const bool ly = my > 0 && mk_float(py.e, py.M + (my - 1) * py.inc) == X;
    const bool lz = mz > 0 && mk_float(pz.e, pz.M + (mz - 1) * pz.inc) == X;

    // new intersection_t: inside the binade by the closed form, across its end by one real float add
    const int32_t Nx = px.M + mx * px.inc, Ny = py.M + my * py.inc, Nz = pz.M + mz * pz.inc;
    if (mx > 0) tx = (Nx <= 0xffffff) ? mk_float(px.e, Nx) : mk_float(px.e, Nx - px.inc) + dx;
    if (my > 0) ty = (Ny <= 0xffffff) ? mk_float(py.e, Ny) : mk_float(py.e, Ny - py.inc) + dy;
    if (mz > 0) tz = (Nz <= 0xffffff) ? mk_float(pz.e, Nz) : mk_float(pz.e, Nz - pz.inc) + dz;
    nx -= mx; ny -= my; nz -= mz;

    res.iterations = iters;
    res.left_node = (nx == 0) || (ny == 0) || (nz == 0);
    res.fx = lx; res.fy = ly; res.fz = lz;
    return res;
}

}  // namespace vrc
