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
        tmp[it] = i < total ? load_stream(src + i) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        if (i < total) store_stream(dst + i, s_sh[sp * kShRowF4 + (i - sp * 12)]);
    }
}

// ---- two source arrays: the reference keeps the SH coefficients as two parameters, `_features_dc` [N,1,3] and
// `_features_rest` [N,15,3] (scene/gaussian_model.py:40-41), and concatenates them every iteration (:79-82).  With the
// two-tensor input the workgroup's LDS holds the two spans exactly as they lie in memory -- dc: 256 x 3 floats at float 0,
// rest: 256 x 45 floats at float kShSplitRest -- so staging in and out is a linear 16-byte copy, and splat t reads / writes
// floats dc[3t..3t+2] and rest[45t..45t+44] (odd strides: conflict-free 4-byte LDS accesses).
constexpr int kRestFloats = 45;                 // 15 coefficients x 3 channels
constexpr int kShSplitRest = kBlock * 3;        // float offset of the rest span inside the LDS array (768; 16-byte aligned)
static_assert((kShSplitRest + kBlock * kRestFloats) * 4 <= kBlock * kShRowF4 * 16, "split layout must fit the padded-row array");

template <int ITERS>
__device__ __forceinline__ void stage_in_linear(float* lds, const float* src, int total_f) {
    float4 tmp[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int f0 = 4 * (it * kBlock + (int)threadIdx.x);
        if (f0 + 3 < total_f) tmp[it] = load_stream(reinterpret_cast<const float4*>(src + f0));
        else  // last, partial float4 of the array: never read past its end
            tmp[it] = make_float4(f0 < total_f ? src[f0] : 0.f, f0 + 1 < total_f ? src[f0 + 1] : 0.f,
                                  f0 + 2 < total_f ? src[f0 + 2] : 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int f0 = 4 * (it * kBlock + (int)threadIdx.x);
        if (f0 < total_f) *reinterpret_cast<float4*>(lds + f0) = tmp[it];   // the pad floats behind the span are never read
    }
}
template <int ITERS>
__device__ __forceinline__ void stage_out_linear(const float* lds, float* dst, int total_f) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int f0 = 4 * (it * kBlock + (int)threadIdx.x);
        if (f0 >= total_f) continue;
        const float4 v = *reinterpret_cast<const float4*>(lds + f0);
        if (f0 + 3 < total_f) store_stream(reinterpret_cast<float4*>(dst + f0), v);
        else {
            dst[f0] = v.x;
            if (f0 + 1 < total_f) dst[f0 + 1] = v.y;
            if (f0 + 2 < total_f) dst[f0 + 2] = v.z;
        }
    }
}
__device__ __forceinline__ void stage_sh_in_split(float4* s_sh, const float* dc, const float* rest, size_t first_splat, int n_here) {
    float* lds = reinterpret_cast<float*>(s_sh);
    stage_in_linear<1>(lds, dc + first_splat * 3, n_here * 3);                                            // <= 192 float4
    stage_in_linear<12>(lds + kShSplitRest, rest + first_splat * kRestFloats, n_here * kRestFloats);      // <= 2880 float4
}
__device__ __forceinline__ void stage_sh_out_split(const float4* s_sh, float* d_dc, float* d_rest, size_t first_splat, int n_here) {
    const float* lds = reinterpret_cast<const float*>(s_sh);
    stage_out_linear<1>(lds, d_dc + first_splat * 3, n_here * 3);
    stage_out_linear<12>(lds + kShSplitRest, d_rest + first_splat * kRestFloats, n_here * kRestFloats);
}

// Where splat `t` of the workgroup finds float j = 3k + c of its 48 coefficients: j < 3 in `lo`, the rest in `hi` (both
// indexed with j).  Padded rows (one source tensor): lo == hi == the splat's row.
__device__ __forceinline__ void sh_row_pointers(float4* s_sh, bool split, int t, float*& lo, float*& hi) {
    float* base = reinterpret_cast<float*>(s_sh);
    if (split) { lo = base + 3 * t; hi = base + kShSplitRest + kRestFloats * t - 3; }
    else lo = hi = base + 4 * kShRowF4 * t;
}

#endif  // __HIPCC__

}  // namespace sr
