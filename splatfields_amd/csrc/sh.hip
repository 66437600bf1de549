// SH colour evaluation as a stand-alone differentiable stage, and its multi-view backward.
//
// Why it exists (multi-GPU): in the view-parallel step every rank renders one view and the per-splat
// gradients must be summed over ranks.  The SH gradient is 192 of the 236 bytes per splat, but for ONE view it has rank-1
// structure: dL/dsh[k][c] = basis_k(direction of the splat seen from that camera) * dL/dcolour[c].  So instead of
// all-reducing 192 B/splat, ranks all-gather the 12-byte colour gradient of every view and each rank rebuilds
// sum_v basis(dir_v) (x) dL/dcolour_v locally with this kernel -- 4x less traffic over xGMI for the same result.
//
// Semantics = the SH part of the rasterizer's preprocess (reference utils/sh_utils.py:57-112,
// extract_geo.py:40-44): dir = normalize(mean - campos), colour = max(sum_k basis_k sh_k + 0.5, 0), gradient zero where clamped.
#include "kernels.h"
#include "sh_stage.h"

namespace sr {

__device__ __forceinline__ void sh_basis16(int deg, float x, float y, float z, float B[16]) {
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = SH_C2_0 * xy; B[5] = SH_C2_1 * yz; B[6] = SH_C2_2 * (2.f * zz - xx - yy); B[7] = SH_C2_3 * xz; B[8] = SH_C2_4 * (xx - yy);
            if (deg > 2) {
                B[9] = SH_C3_0 * y * (3.f * xx - yy); B[10] = SH_C3_1 * xy * z; B[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
                B[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy); B[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
                B[14] = SH_C3_5 * z * (xx - yy); B[15] = SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

// d(sum_k basis_k g_k)/d(unit direction)
__device__ __forceinline__ float3 sh_dir_grad(int deg, float x, float y, float z, const float gk[16]) {
    float3 d = make_float3(0.f, 0.f, 0.f);
    if (deg > 0) {
        d.x += -SH_C1 * gk[3]; d.y += -SH_C1 * gk[1]; d.z += SH_C1 * gk[2];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            d.x += SH_C2_0 * y * gk[4] + SH_C2_2 * -2.f * x * gk[6] + SH_C2_3 * z * gk[7] + SH_C2_4 * 2.f * x * gk[8];
            d.y += SH_C2_0 * x * gk[4] + SH_C2_1 * z * gk[5] + SH_C2_2 * -2.f * y * gk[6] + SH_C2_4 * -2.f * y * gk[8];
            d.z += SH_C2_1 * y * gk[5] + SH_C2_2 * 4.f * z * gk[6] + SH_C2_3 * x * gk[7];
            if (deg > 2) {
                d.x += SH_C3_0 * 6.f * xy * gk[9] + SH_C3_1 * yz * gk[10] + SH_C3_2 * -2.f * xy * gk[11] + SH_C3_3 * -6.f * xz * gk[12] +
                       SH_C3_4 * (4.f * zz - 3.f * xx - yy) * gk[13] + SH_C3_5 * 2.f * xz * gk[14] + SH_C3_6 * (3.f * xx - 3.f * yy) * gk[15];
                d.y += SH_C3_0 * (3.f * xx - 3.f * yy) * gk[9] + SH_C3_1 * xz * gk[10] + SH_C3_2 * (4.f * zz - xx - 3.f * yy) * gk[11] +
                       SH_C3_3 * -6.f * yz * gk[12] + SH_C3_4 * -2.f * xy * gk[13] + SH_C3_5 * -2.f * yz * gk[14] + SH_C3_6 * -6.f * xy * gk[15];
                d.z += SH_C3_1 * xy * gk[10] + SH_C3_2 * 8.f * yz * gk[11] + SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy) * gk[12] +
                       SH_C3_4 * 8.f * xz * gk[13] + SH_C3_5 * (xx - yy) * gk[14];
            }
        }
    }
    return d;
}

template <bool STAGE>  // STAGE: K == 16 -> the workgroup's contiguous 48 KiB of coefficients arrive through LDS (coalesced 16-byte loads)
__global__ void __launch_bounds__(kBlock) k_sh_forward(int N, int K, int deg, const float* __restrict__ means3D, const float* __restrict__ shs,
                                                       const float* __restrict__ campos, float* __restrict__ colors, unsigned char* __restrict__ clamped) {
    __shared__ float4 s_sh[STAGE ? kBlock * kShRowF4 : 1];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if constexpr (STAGE) {
        const size_t first = (size_t)blockIdx.x * kBlock;
        stage_sh_in(s_sh, shs, first, min(kBlock, N - (int)first));
        __syncthreads();
    }
    if (idx >= N) return;
    float dx = means3D[3 * (size_t)idx] - campos[0], dy = means3D[3 * (size_t)idx + 1] - campos[1], dz = means3D[3 * (size_t)idx + 2] - campos[2];
    const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    dx *= il; dy *= il; dz *= il;
    float B[16];
    sh_basis16(deg, dx, dy, dz, B);
    const float* sh = STAGE ? reinterpret_cast<const float*>(&s_sh[threadIdx.x * kShRowF4]) : shs + (size_t)idx * K * 3;
    const int nb = (deg + 1) * (deg + 1);
    float r = 0.f, g = 0.f, b = 0.f;
    for (int k = 0; k < nb; ++k) { r += B[k] * sh[3 * k]; g += B[k] * sh[3 * k + 1]; b += B[k] * sh[3 * k + 2]; }
    r += 0.5f; g += 0.5f; b += 0.5f;
    unsigned char fl = 0;
    if (r < 0.f) { fl |= 1; r = 0.f; }
    if (g < 0.f) { fl |= 2; g = 0.f; }
    if (b < 0.f) { fl |= 4; b = 0.f; }
    colors[3 * (size_t)idx] = r; colors[3 * (size_t)idx + 1] = g; colors[3 * (size_t)idx + 2] = b;
    clamped[idx] = fl;
}

// The same evaluation for V cameras in one pass over the coefficients (the SH-sharded view-parallel step evaluates its
// shard of the splats for every view of the step): colors [V][N][3]; keep [V][N][3] (may be NULL) = 1 where the channel was
// not clamped, 0 where it was -- the factor its colour gradient gets in the backward.
template <bool STAGE>
__global__ void __launch_bounds__(kBlock) k_sh_forward_views(int N, int K, int deg, int V, const float* __restrict__ means3D,
                                                             const float* __restrict__ shs, const float* __restrict__ campos,
                                                             float* __restrict__ colors, float* __restrict__ keep) {
    __shared__ float4 s_sh[STAGE ? kBlock * kShRowF4 : 1];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if constexpr (STAGE) {
        const size_t first = (size_t)blockIdx.x * kBlock;
        stage_sh_in(s_sh, shs, first, min(kBlock, N - (int)first));
        __syncthreads();
    }
    if (idx >= N) return;
    const float px = means3D[3 * (size_t)idx], py = means3D[3 * (size_t)idx + 1], pz = means3D[3 * (size_t)idx + 2];
    const float* sh = STAGE ? reinterpret_cast<const float*>(&s_sh[threadIdx.x * kShRowF4]) : shs + (size_t)idx * K * 3;
    const int nb = (deg + 1) * (deg + 1);
    for (int v = 0; v < V; ++v) {
        float dx = px - campos[3 * v], dy = py - campos[3 * v + 1], dz = pz - campos[3 * v + 2];
        const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        dx *= il; dy *= il; dz *= il;
        float B[16];
        sh_basis16(deg, dx, dy, dz, B);
        float r = 0.f, g = 0.f, b = 0.f;
        for (int k = 0; k < nb; ++k) { r += B[k] * sh[3 * k]; g += B[k] * sh[3 * k + 1]; b += B[k] * sh[3 * k + 2]; }
        r += 0.5f; g += 0.5f; b += 0.5f;
        const size_t o = ((size_t)v * N + idx) * 3;
        if (keep) { keep[o] = r < 0.f ? 0.f : 1.f; keep[o + 1] = g < 0.f ? 0.f : 1.f; keep[o + 2] = b < 0.f ? 0.f : 1.f; }
        colors[o] = fmaxf(r, 0.f); colors[o + 1] = fmaxf(g, 0.f); colors[o + 2] = fmaxf(b, 0.f);
    }
}

// dcol: [V][N][3] colour gradients, ALREADY zeroed where the colour was clamped in that view.  campos: [V][3].
// d_shs (may be NULL): [N][K][3] = scale * sum_v basis(dir_v) (x) dcol_v  (bands above `deg` get zeros).
// d_means (may be NULL): [N][3] (+)= scale * sum_v d(colour_v . dcol_v)/d(mean)   (through the normalised direction).
template <bool STAGE>  // STAGE: K == 16 and d_shs != NULL -> gradient rows leave through LDS with coalesced 16-byte stores
__global__ void __launch_bounds__(kBlock) k_sh_backward(int N, int K, int deg, int V, const float* __restrict__ means3D,
                                                        const float* __restrict__ shs, const float* __restrict__ campos,
                                                        const float* __restrict__ dcol, float scale, float* __restrict__ d_shs,
                                                        float* __restrict__ d_means, int accumulate_means) {
    __shared__ float4 s_sh[STAGE ? kBlock * kShRowF4 : 1];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx < N) {
    const float px = means3D[3 * (size_t)idx], py = means3D[3 * (size_t)idx + 1], pz = means3D[3 * (size_t)idx + 2];
    const int nb = (deg + 1) * (deg + 1);
    const float* sh = shs + (size_t)idx * K * 3;
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = 0.f;
    float3 dm = make_float3(0.f, 0.f, 0.f);
    for (int v = 0; v < V; ++v) {
        const float* dc = dcol + ((size_t)v * N + idx) * 3;
        const float cr = dc[0], cg = dc[1], cb = dc[2];
        if (cr == 0.f && cg == 0.f && cb == 0.f) continue;  // culled or fully clamped in this view
        float dx = px - campos[3 * v], dy = py - campos[3 * v + 1], dz = pz - campos[3 * v + 2];
        const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        dx *= il; dy *= il; dz *= il;
        float B[16];
        sh_basis16(deg, dx, dy, dz, B);
        if (d_shs) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < nb) { acc[3 * k] += B[k] * cr; acc[3 * k + 1] += B[k] * cg; acc[3 * k + 2] += B[k] * cb; }
        }
        if (d_means && deg > 0) {
            float gk[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) gk[k] = k < nb ? sh[3 * k] * cr + sh[3 * k + 1] * cg + sh[3 * k + 2] * cb : 0.f;
            const float3 dd = sh_dir_grad(deg, dx, dy, dz, gk);
            const float proj = dx * dd.x + dy * dd.y + dz * dd.z;
            dm.x += (dd.x - dx * proj) * il; dm.y += (dd.y - dy * proj) * il; dm.z += (dd.z - dz * proj) * il;
        }
    }
    if (d_shs) {
        float* out = STAGE ? reinterpret_cast<float*>(&s_sh[threadIdx.x * kShRowF4]) : d_shs + (size_t)idx * K * 3;
#pragma unroll
        for (int k = 0; k < 16; ++k) {  // static register indices (a runtime-indexed acc[] would live in scratch)
            if (k < K) {
                const bool live = k < nb;
                out[3 * k] = live ? scale * acc[3 * k] : 0.f; out[3 * k + 1] = live ? scale * acc[3 * k + 1] : 0.f; out[3 * k + 2] = live ? scale * acc[3 * k + 2] : 0.f;
            }
        }
        for (int k = 16; k < K; ++k) { out[3 * k] = 0.f; out[3 * k + 1] = 0.f; out[3 * k + 2] = 0.f; }
    }
    if (d_means) {
        float* o = d_means + 3 * (size_t)idx;
        if (accumulate_means) { o[0] += scale * dm.x; o[1] += scale * dm.y; o[2] += scale * dm.z; }
        else { o[0] = scale * dm.x; o[1] = scale * dm.y; o[2] = scale * dm.z; }
    }
    }  // idx < N
    if constexpr (STAGE) {
        __syncthreads();
        const size_t first = (size_t)blockIdx.x * kBlock;
        stage_sh_out(s_sh, d_shs, first, min(kBlock, N - (int)first));
    }
}

void launch_sh_forward(int N, int K, int deg, const float* means3D, const float* shs, const float* campos, float* colors,
                       unsigned char* clamped, hipStream_t st) {
    if (N <= 0) return;
    if (K == 16 && deg >= 2)  // below degree 2 only <= 48 of a splat's 192 bytes are needed
        hipLaunchKernelGGL(k_sh_forward<true>, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, K, deg, means3D, shs, campos, colors, clamped);
    else
        hipLaunchKernelGGL(k_sh_forward<false>, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, K, deg, means3D, shs, campos, colors, clamped);
}

void launch_sh_forward_views(int N, int K, int deg, int V, const float* means3D, const float* shs, const float* campos, float* colors,
                             float* keep, hipStream_t st) {
    if (N <= 0 || V <= 0) return;
    if (K == 16)
        hipLaunchKernelGGL(k_sh_forward_views<true>, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, K, deg, V, means3D, shs, campos, colors, keep);
    else
        hipLaunchKernelGGL(k_sh_forward_views<false>, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, K, deg, V, means3D, shs, campos, colors, keep);
}

void launch_sh_backward(int N, int K, int deg, int V, const float* means3D, const float* shs, const float* campos, const float* dcol,
                        float scale, float* d_shs, float* d_means, int accumulate_means, hipStream_t st) {
    if (N <= 0) return;
    if (d_shs && K == 16)
        hipLaunchKernelGGL(k_sh_backward<true>, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, K, deg, V, means3D, shs, campos, dcol,
                           scale, d_shs, d_means, accumulate_means);
    else
        hipLaunchKernelGGL(k_sh_backward<false>, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, K, deg, V, means3D, shs, campos, dcol,
                           scale, d_shs, d_means, accumulate_means);
}

}  // namespace sr
