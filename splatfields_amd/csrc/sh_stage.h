// LDS staging of the per-splat SH blocks (shared by preprocess.hip and sh.hip).
#pragma once
#include "common.h"

namespace sr {

#ifdef __HIPCC__
// SH staging: with K = 16 coefficients a splat's SH block is 192 contiguous bytes, so the 256 splats of a
// workgroup own one contiguous 48 KiB span.  It is moved with fully coalesced 16-byte loads/stores through LDS
// (rows padded to 13 float4 = 208 B) instead of 48 strided 4-byte accesses per lane.
constexpr int kShRowF4 = 13;

// 256 splats x 12 float4: every thread moves 12 float4, all 12 global loads issued before the first LDS write
// (the naive loop compiled to load -> s_waitcnt vmcnt(0) -> ds_write per iteration: one load in flight per lane).
__device__ __forceinline__ void stage_sh_in(float4* s_sh, const float* shs, size_t first_splat, int n_here) {
    const float4* src = reinterpret_cast<const float4*>(shs + first_splat * 48);
    const int total = n_here * 12;
    float4 tmp[12];
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int i = it * kBlock + (int)threadIdx.x;
        tmp[it] = i < total ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int i = it * kBlock + (int)threadIdx.x;
        const int sp = i / 12;
        if (i < total) s_sh[sp * kShRowF4 + (i - sp * 12)] = tmp[it];
    }
}
__device__ __forceinline__ void stage_sh_out(const float4* s_sh, float* dst_base, size_t first_splat, int n_here) {
    float4* dst = reinterpret_cast<float4*>(dst_base + first_splat * 48);
    const int total = n_here * 12;
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int i = it * kBlock + (int)threadIdx.x;
        const int sp = i / 12;
        if (i < total) dst[i] = s_sh[sp * kShRowF4 + (i - sp * 12)];
    }
}

#endif  // __HIPCC__

}  // namespace sr
