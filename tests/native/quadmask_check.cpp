// Host check of splatfields_amd/csrc/quadmask.h: for random splat records around a 16x16 tile, the quad mask must
// contain every quad that holds a pixel passing the per-pixel alpha >= 1/255 test of the blend kernels (evaluated
// with the kernels' own fp32 formula).  Prints: cases, missed quads (must be 0), mask quads, true quads.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include "../../splatfields_amd/csrc/quadmask.h"

static float alpha_unclamped(float dx, float dy, float p, float s, float q, float nlo) {
    const float t = p * fmaf(s, dy, dx), u = q * dy;
    const float w = fmaf(u, u, fmaf(t, t, nlo));
    return exp2f(-w);
}

int main(int argc, char** argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 200000;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const float kLog2e = 1.4426950408889634f;
    long long missed = 0, mask_quads = 0, true_quads = 0, nonempty = 0;
    for (int it = 0; it < cases; ++it) {
        // random 2D covariance (with the 0.3 dilation), sizes from sub-pixel to several tiles, any orientation
        const float sig1 = expf(logf(0.05f) + U(rng) * (logf(40.f) - logf(0.05f)));
        const float sig2 = sig1 * expf(-U(rng) * 5.0f);
        const float th_ = U(rng) * 6.2831853f;
        const float c_ = cosf(th_), s_ = sinf(th_);
        const float a = c_ * c_ * sig1 * sig1 + s_ * s_ * sig2 * sig2 + 0.3f;
        const float b = c_ * s_ * (sig1 * sig1 - sig2 * sig2);
        const float c = s_ * s_ * sig1 * sig1 + c_ * c_ * sig2 * sig2 + 0.3f;
        const float det = a * c - b * b;
        const float opac = it % 7 == 0 ? 0.0035f + U(rng) * 0.003f : 0.004f + U(rng) * 0.996f;
        const float tau = 2.0f * logf(255.0f * opac);
        const float tau2 = tau > 0.f ? tau * kLog2e : -1.0f;
        const float p = sqrtf(0.5f * kLog2e * c / det), s = -b / c, q = sqrtf(0.5f * kLog2e / c);
        const float nlo = -log2f(opac);
        const float tx0 = 16.0f * (float)(it % 50), ty0 = 16.0f * (float)((it / 50) % 50);
        const float spread = 8.0f + 3.0f * sig1;
        const float cx = tx0 + 7.5f + (U(rng) * 2.f - 1.f) * spread, cy = ty0 + 7.5f + (U(rng) * 2.f - 1.f) * spread;
        const uint32_t mask = sr_quad_mask(cx, cy, tau2, p, s, q, tx0, ty0);
        uint32_t truth = 0;
        for (int y = 0; y < 16; ++y)
            for (int x = 0; x < 16; ++x) {
                const float al = alpha_unclamped(cx - (tx0 + x), cy - (ty0 + y), p, s, q, nlo);
                if (al >= 1.0f / 255.0f) truth |= 1u << ((y >> 2) * 4 + (x >> 2));
            }
        missed += __builtin_popcount(truth & ~mask);
        mask_quads += __builtin_popcount(mask);
        true_quads += __builtin_popcount(truth);
        nonempty += truth != 0;
    }
    printf("%d %lld %lld %lld %lld\n", cases, missed, mask_quads, true_quads, nonempty);
    return missed == 0 ? 0 : 1;
}
