// Host check of voxel-raycaster_amd/csrc/safe_run.hpp against the plain float loop of
// kernels/ray_caster_kernel.cl:558-560: random ray states inside an empty node (countdowns nx, ny, nz to the
// node face); the safe run must (a) never take the step that leaves the node, (b) leave t bit-identical to the
// plain loop after the same number of iterations, (c) report exactly the steps the plain loop took per axis.
// Usage: safe_vs_loop <cases> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../voxel-raycaster_amd/csrc/safe_run.hpp"

using namespace vrc;

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 100000;
    std::mt19937_64 rng(argc > 2 ? atol(argv[2]) : 1);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    long mismatches = 0, opened = 0, safe_iters = 0, ties = 0, stopped_early = 0;
    for (long c = 0; c < cases; c++) {
        // ray direction: random, sometimes nearly axis parallel, sometimes two equal components (tie generator)
        float rd[3];
        for (int a = 0; a < 3; a++) rd[a] = (U(rng) < 0.15f ? 1e-4f : 1.0f) * (U(rng) * 2.0f - 1.0f);
        const int kind = (int)(U(rng) * 6.0f);
        if (kind == 0) rd[1] = rd[0];
        if (kind == 1) { rd[1] = rd[0]; rd[2] = -rd[0]; }
        float len = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
        bool ok = len > 0.0f;
        const float scale = kind == 4 ? 0.3f + 2.7f * U(rng) : 1.0f;                   // host-supplied tables need not be normalised
        for (int a = 0; a < 3; a++) { rd[a] = rd[a] / len * scale; ok = ok && rd[a] != 0.0f; }
        if (!ok) { c--; continue; }
        float dt[3], t[3], n[3];
        const float travelled = (kind == 2 ? 140000.0f : 6000.0f) * U(rng) * U(rng);    // also beyond safe_t_limit
        const int node = 1 << (int)(U(rng) * 11.0f);
        for (int a = 0; a < 3; a++) {
            dt[a] = fabsf(1.0f / rd[a]);
            const float frac = (kind <= 1) ? 0.25f : U(rng);                           // equal offsets keep the ties alive
            t[a] = dt[a] * (floorf(travelled * fabsf(rd[a])) + frac);
            if (kind == 3 && a == 0) t[a] = 0.0f;                                      // camera on a voxel boundary
            n[a] = (float)(1 + (int)(U(rng) * (float)node));
        }
        const int cap = 2 * (1 + (int)(U(rng) * (kSafeMaxSteps / 2)));

        // safe run, as the kernel does it
        float s[3] = {t[0], t[1], t[2]};
        if (!(t_is_safe(s[0]) && t_is_safe(s[1]) && t_is_safe(s[2]))) continue;
        const float T = fminf(fminf(safe_threshold(s[0], dt[0], n[0]), safe_threshold(s[1], dt[1], n[1])),
                              safe_threshold(s[2], dt[2], n[2]));
        const SafeGate gate = make_gate(T, fminf(fminf(s[0], s[1]), s[2]), safe_t_limit(cap), fminf(fminf(dt[0], dt[1]), dt[2]));
        float cnt = 0.0f;
        for (int i = 0; i < cap; i++) {
            const float m = fminf(fminf(s[0], s[1]), s[2]);
            const float alive = fma_sat(m, gate.neg_b1, gate.tb1);
            cnt += alive;
            for (int a = 0; a < 3; a++) s[a] = fmaf(dt[a], alive_if_zero(s[a] - m, alive), s[a]);
        }
        if (!gate.open) { mismatches += cnt != 0.0f || bits(s[0]) != bits(t[0]) || bits(s[1]) != bits(t[1]) || bits(s[2]) != bits(t[2]); continue; }
        opened++;
        safe_iters += (long)cnt;
        stopped_early += (int)cnt < cap;

        // the plain loop for the same number of iterations
        float p[3] = {t[0], t[1], t[2]}, left[3] = {n[0], n[1], n[2]};
        bool left_node = false;
        for (int i = 0; i < (int)cnt; i++) {
            const float m = fminf(fminf(p[0], p[1]), p[2]);
            int stepped = 0;
            for (int a = 0; a < 3; a++) {
                const float f = p[a] <= m ? 1.0f : 0.0f;
                p[a] += dt[a] * f;
                left[a] -= f;
                stepped += (int)f;
            }
            ties += stepped > 1;
            left_node = left_node || left[0] == 0.0f || left[1] == 0.0f || left[2] == 0.0f;
        }
        bool bad = left_node;
        for (int a = 0; a < 3; a++) {
            bad = bad || bits(p[a]) != bits(s[a]);
            bad = bad || (n[a] - safe_steps_taken(s[a], t[a], rd[a])) != left[a];
        }
        if (bad && mismatches++ < 10)
            printf("case %ld: n=(%g,%g,%g) cnt=%g left=(%g,%g,%g) left_node=%d\n", c, n[0], n[1], n[2], cnt, left[0], left[1], left[2], left_node);
    }
    printf("cases %ld opened %ld safe iterations %ld stopped_before_cap %ld tie_iterations %ld mismatches %ld\n", cases, opened,
           safe_iters, stopped_early, ties, mismatches);
    return mismatches ? 1 : 0;
}
