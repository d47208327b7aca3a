// safe_run.hpp -- the SVO kernel's step loop with the face mask and the loop control done in full-rate
// arithmetic (v_sub/v_mul/v_fma issue ~1.6x faster than compares, selects and min/max on gfx950; tools/ubench).
//
// The reference's iteration (kernels/ray_caster_kernel.cl:558-560)
//     face_mask = (t.xyz <= min(t.yzx, t.zxy));   t += delta_t * face_mask;   voxel += step * face_mask;
// is kept step for step and bit for bit; only how the mask is formed and how the loop decides to stop changes.
//
//  * mask:  d = t_a - min(t) >= 0 is 0 exactly when t_a is the minimum (IEEE subtraction with denormals never
//    rounds a non-zero difference to 0) and clamp(alive - d * 2^127) is alive for d == 0 and 0 for d >= 2^-127;
//    every non-zero d is that large when all t are 0 or >= 2^-100 (t_is_safe: differences of such floats are at
//    least ulp(2^-100) = 2^-123).
//  * safe run: while a ray is deep inside an empty node the per-axis countdowns to the node face are not needed
//    step by step.  The iteration that takes the n_a-th step on axis a has min(t) = E_a, the value of t_a after
//    n_a - 1 steps of the float recurrence, and E_a >= T_a := fl(fl(t_a + (n_a-1) dt_a) * (1 - n_a 2^-23)) because
//    each step rounds by at most 2^-24 relative.  Every iteration with min(t) < T := min_a T_a therefore stays
//    inside the node: the lane steps while alive = clamp((T - min t) * B1) is 1, with B1 = 2^(24-e) for
//    2^e <= T < 2^(e+1) (floats below T are at least 2^(e-24) away, so the product is >= 1), and counts its
//    iterations.  Afterwards the steps taken per axis are rint((t_a - t_a0) * |ray_dir_a|): a stepped t stays below
//    2T, so k iterations drift by at most k * ulp(2T) / 2 <= 1/4 step when T < 2^22 / k (safe_t_limit) and delta_t >= 1.
//
// Host+device header: tools/jumptest/safe_vs_loop.cpp drives the host build against the plain loop
// (tests/test_safe_run.py); raycast_kernel.hip uses the device build.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VRC_SR __host__ __device__ __forceinline__
#else
#define VRC_SR inline
#endif

namespace vrc {

constexpr int kSafeMaxSteps = 256;     // most iterations per safe run the settings allow
#ifndef VRC_SAFE_UNROLL
#define VRC_SAFE_UNROLL 16
#endif
#ifndef VRC_SAFE_UNROLL_PLAIN
#define VRC_SAFE_UNROLL_PLAIN 8
#endif
// iterations per loop trip of a safe run (setting safe_steps is rounded down to it): the jump instances run ONE trip of 16 per round;
// the plain instances (trees below depth 12) up to four trips of 8 with the exit vote between them (depth 10: 0.369 -> 0.349 ms;
// the same in the jump instances: 1.91 -> 1.95 ms)
constexpr int kSafeUnrollJump = VRC_SAFE_UNROLL, kSafeUnrollPlain = VRC_SAFE_UNROLL_PLAIN;

// largest threshold T a safe run of at most `steps` iterations may use (see the recovery bound above)
VRC_SR float safe_t_limit(int steps) {
    int k = 1;
    while (k < steps) k <<= 1;
    return 0x1p22f / (float)k;
}

// clamp(a * b + c) to [0, 1], NaN -> 0: one v_fma_f32 with the clamp modifier on the device
VRC_SR float fma_sat(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float f;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(f) : "v"(a), "v"(b), "v"(c));
    return f;
#else
    const float v = fmaf(a, b, c);
    return v >= 1.0f ? 1.0f : (v > 0.0f ? v : 0.0f);
#endif
}
// min(a * b, 1) for a, b >= 0
VRC_SR float mul_sat(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float f;
    asm("v_mul_f32 %0, %1, %2 clamp" : "=v"(f) : "v"(a), "v"(b));
    return f;
#else
    const float v = a * b;
    return v >= 1.0f ? 1.0f : (v > 0.0f ? v : 0.0f);
#endif
}
// alive (1.0 or 0.0) when d == 0, 0.0 when d >= 2^-127
VRC_SR float alive_if_zero(float d, float alive) { return fma_sat(d, -0x1p127f, alive); }

VRC_SR bool t_is_safe(float t) { return t == 0.0f || t >= 0x1p-100f; }

// lower bound of min(t) in the iteration that takes the n-th step on this axis (n >= 1 steps to the node face)
VRC_SR float safe_threshold(float t, float dt, float n) { return fmaf(n - 1.0f, dt, t) * (1.0f - n * 0x1p-23f); }

// alive = fma_sat(min_t, neg_b1, tb1).  The default gate is closed for every value of t, NaN and inf included.
struct SafeGate { float neg_b1 = 0.0f, tb1 = -1.0f; bool open = false; };

// min_dt: the smallest delta_t of the ray; the recovery bound needs delta_t >= 1 (|ray_dir| <= 1: the reference's rays
// are normalised, a host-supplied ray table need not be)
// kSelects: no branches -- every lane computes, the tests pick (a NaN fails them all).  The jump instances of the SVO kernel are
// 0.4-1.3 % faster with it (the nested conditions compile to three exec-masked branches per round), the plain instance 1.4 % slower.
template <bool kSelects = false>
VRC_SR SafeGate make_gate(float T, float min_t, float t_limit, float min_dt) {
    SafeGate g;
    if (kSelects) {
        union { float f; uint32_t u; } c;
        c.f = T;
        c.u = (278u - (c.u >> 23)) << 23;          // 2^(24-e) for 2^e <= T < 2^(e+1)
        g.open = (T >= 0x1p-60f) & (T < t_limit) & (min_t < T) & (min_dt >= 0.999f);
        g.neg_b1 = g.open ? -c.f : 0.0f;
        g.tb1 = g.open ? T * c.f : -1.0f;
    } else if (T >= 0x1p-60f && T < t_limit && min_t < T && min_dt >= 0.999f) {
        union { float f; uint32_t u; } c;
        c.f = T;
        c.u = (278u - (c.u >> 23)) << 23;          // 2^(24-e) for 2^e <= T < 2^(e+1)
        g.neg_b1 = -c.f;
        g.tb1 = T * c.f;
        g.open = true;
    }
    return g;
}

// steps taken on one axis during a safe run
VRC_SR float safe_steps_taken(float t, float t0, float ray_dir) { return rintf((t - t0) * fabsf(ray_dir)); }

}  // namespace vrc
