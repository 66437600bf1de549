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

// ---- two source arrays: the reference keeps the SH coefficients as two parameters, `_features_dc` [N,1,3] and
// `_features_rest` [N,15,3] (scene/gaussian_model.py:40-41), and concatenates them every iteration (:79-82).  These variants
// fill / drain the same LDS rows (floats 0..2 = dc, 3..47 = rest) straight from / into the two arrays: the 256 splats of a
// workgroup own one contiguous 3 KiB span of dc and one contiguous 45 KiB span of rest, both moved with 16-byte accesses.
constexpr int kShRowFloats = kShRowF4 * 4;   // 52
constexpr int kRestFloats = 45;              // 15 coefficients x 3 channels

// splat-local float index f of a PER floats-per-splat array -> its LDS float slot (row of the splat, offset OFF inside it)
template <int PER, int OFF>
__device__ __forceinline__ void split_slots(int f0, int (&slot)[4]) {
    const int sp0 = f0 / PER, k0 = f0 - sp0 * PER;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = k0 + i, wrap = kk >= PER ? 1 : 0;   // a float4 never spans more than two splats (PER >= 3, handled below)
        slot[i] = (sp0 + wrap) * kShRowFloats + OFF + kk - wrap * PER;
    }
}
template <>
__device__ __forceinline__ void split_slots<3, 0>(int f0, int (&slot)[4]) {   // dc: 3 floats per splat, a float4 touches 2 splats
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int f = f0 + i, sp = f / 3; slot[i] = sp * kShRowFloats + (f - sp * 3); }
}

template <int PER, int OFF, int ITERS>
__device__ __forceinline__ void stage_in_part(float* s_rows, const float* src, int total_f) {
    float4 tmp[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int f0 = 4 * (it * kBlock + (int)threadIdx.x);
        if (f0 + 3 < total_f) tmp[it] = *reinterpret_cast<const float4*>(src + f0);
        else {  // last, partial float4 of the array: never read past its end
            tmp[it] = make_float4(f0 < total_f ? src[f0] : 0.f, f0 + 1 < total_f ? src[f0 + 1] : 0.f,
                                  f0 + 2 < total_f ? src[f0 + 2] : 0.f, 0.f);
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int f0 = 4 * (it * kBlock + (int)threadIdx.x);
        if (f0 >= total_f) continue;
        int slot[4];
        split_slots<PER, OFF>(f0, slot);
        const float v[4] = {tmp[it].x, tmp[it].y, tmp[it].z, tmp[it].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) if (f0 + i < total_f) s_rows[slot[i]] = v[i];
    }
}
template <int PER, int OFF, int ITERS>
__device__ __forceinline__ void stage_out_part(const float* s_rows, float* dst, int total_f) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int f0 = 4 * (it * kBlock + (int)threadIdx.x);
        if (f0 >= total_f) continue;
        int slot[4];
        split_slots<PER, OFF>(f0, slot);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = f0 + i < total_f ? s_rows[slot[i]] : 0.f;
        if (f0 + 3 < total_f) *reinterpret_cast<float4*>(dst + f0) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int i = 0; i < 3; ++i) if (f0 + i < total_f) dst[f0 + i] = v[i];
        }
    }
}
__device__ __forceinline__ void stage_sh_in_split(float4* s_sh, const float* dc, const float* rest, size_t first_splat, int n_here) {
    float* rows = reinterpret_cast<float*>(s_sh);
    stage_in_part<3, 0, 1>(rows, dc + first_splat * 3, n_here * 3);                               // <= 192 float4
    stage_in_part<kRestFloats, 3, 12>(rows, rest + first_splat * kRestFloats, n_here * kRestFloats);  // <= 2880 float4
}
__device__ __forceinline__ void stage_sh_out_split(const float4* s_sh, float* d_dc, float* d_rest, size_t first_splat, int n_here) {
    const float* rows = reinterpret_cast<const float*>(s_sh);
    stage_out_part<3, 0, 1>(rows, d_dc + first_splat * 3, n_here * 3);
    stage_out_part<kRestFloats, 3, 12>(rows, d_rest + first_splat * kRestFloats, n_here * kRestFloats);
}

#endif  // __HIPCC__

}  // namespace sr
