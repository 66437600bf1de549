// Which 4x4-pixel quads of a 16x16 tile can a splat reach with alpha >= 1/255?  (Backward blend, blend_bwd.hip.)
//
// The per-splat record holds the exponent in completed-square form (common.h, pair_alpha_unclamped):
//   alpha_unclamped = exp2(-(t^2 + u^2 + nlo)),  t = p (dx + s dy),  u = q dy,  dx = cx - x, dy = cy - y,
// so alpha >= 1/255  <=>  t^2 + u^2 <= th := tau2 / 2  (tau2 = 2 ln(255 o) log2(e), record q0.z).
// For a pixel row at offset dy the reachable columns are  |x - (cx + s dy)| <= h(dy) = sqrt(th - q^2 dy^2) / p.
// Over a band of rows (continuous dy in [lo, hi]) the right end  cx + s dy + h(dy)  is concave in dy, so its maximum is
// at its stationary point dy* = s p sqrt(th) / (q sqrt(q^2 + s^2 p^2)) clamped into the band (the left end: -dy*).
// The result is a conservative superset of the exact pixel set (continuous band instead of 4 integer rows, and the
// same safety margins as the forward's sub-tile test), never a subset: a missing quad would drop gradient terms.
//
// Plain C++ (no device intrinsics unless SR_QM_DEVICE is defined) so that tests/ can compile it for the host and
// compare it with a brute-force evaluation of the per-pixel test.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef SR_QM_DEVICE
#define SR_QM_FN __device__ __forceinline__
#define SR_QM_RCP(x) __builtin_amdgcn_rcpf(x)
#define SR_QM_RSQ(x) __builtin_amdgcn_rsqf(x)
#define SR_QM_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define SR_QM_FN static inline
#define SR_QM_RCP(x) (1.0f / (x))
#define SR_QM_RSQ(x) (1.0f / sqrtf(x))
#define SR_QM_SQRT(x) sqrtf(x)
#endif

// cx, cy: splat centre in pixel coordinates; tau2, p, s, q: record fields; (tx0, ty0): pixel coordinates of the
// tile's first pixel.  Bit 4*qy + qx of the result: quad (qx, qy) of the tile may hold a pixel with alpha >= 1/255.
SR_QM_FN uint32_t sr_quad_mask(float cx, float cy, float tau2, float p, float s, float q, float tx0, float ty0) {
    if (!(tau2 > 0.0f)) return 0u;
    const float kMargin = 0.02f;                       // pixels
    const float th = 0.5f * tau2 * 1.002f + 0.015f;    // the forward's sub-tile test uses the same relative slack
    const float sth = SR_QM_SQRT(th);
    const float inv_q = SR_QM_RCP(q), inv_p = SR_QM_RCP(p);
    const float ymax = sth * inv_q;                    // |dy| <= ymax
    const float sp = s * p;
    const float dystar = sp * sth * inv_q * SR_QM_RSQ(q * q + sp * sp);
    const float q2 = q * q;
    const float xrel = cx - tx0;                       // centre relative to the tile
    uint32_t mask = 0u;
#pragma unroll
    for (int qy = 0; qy < 4; ++qy) {
        // rows y = ty0 + 4 qy + j, j = 0..3:  dy = cy - y in [cy - ty0 - 4 qy - 3, cy - ty0 - 4 qy]
        const float dhi = (cy - ty0) - (float)(4 * qy) + kMargin;
        const float dlo = dhi - 3.0f - 2.0f * kMargin;
        const float lo = fmaxf(dlo, -ymax), hi = fminf(dhi, ymax);
        if (!(lo <= hi)) continue;
        const float d1 = fminf(hi, fmaxf(lo, dystar));
        const float h1 = SR_QM_SQRT(fmaxf(0.0f, th - q2 * d1 * d1)) * inv_p;
        const float xhi = xrel + s * d1 + h1 + kMargin;
        const float d2 = fminf(hi, fmaxf(lo, -dystar));
        const float h2 = SR_QM_SQRT(fmaxf(0.0f, th - q2 * d2 * d2)) * inv_p;
        const float xlo = xrel + s * d2 - h2 - kMargin;
        // tile-local integer columns in [xlo, xhi], clipped to the tile
        const float f0 = fmaxf(ceilf(xlo), 0.0f), f1 = fminf(floorf(xhi), 15.0f);
        if (!(f0 <= f1)) continue;
        const int c0 = (int)f0 >> 2, c1 = (int)f1 >> 2;
        const uint32_t bits = ((2u << c1) - 1u) & ~((1u << c0) - 1u);
        mask |= bits << (4 * qy);
    }
    return mask;
}
