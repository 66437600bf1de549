// Tri-plane feature lookup of the deform network's encoder, forward and backward, for gfx950.
//
// Semantics: the per-point half of the reference's VarTriPlaneEncoder.forward (scene/tripFields.py:430-436):
//   coord = stack([pts[..., (0,1)], pts[..., (1,2)], pts[..., (2,0)]]);  F.grid_sample(planes[3,C,H,W], coord)
// i.e. bilinear interpolation with zero padding and align_corners = False (pixel = ((v + 1) S - 1) / 2), the features of
// the xy / yz / zx planes concatenated plane-major into [N, 3 C].  The plane GENERATOR (the reference's diffusers-based
// decoder) is not part of this library; it hands over the [3, C, H, W] planes.
//
// MI355X mapping: the planes arrive channel-major ([C][H][W]: a corner's C channels are H W floats apart, C cache lines per
// corner).  They are transposed once per call to texel-major [H][W][C] (20 MB at 16 x 320 x 320: a few microseconds), after
// which a corner is C contiguous floats -- one 64-byte line at C = 16 -- and a point touches 12 lines.
// Backward: the gradient with respect to the points is a gather (no conflicts).  The gradient with respect to the planes is
// a scatter of 4 corners x C values per (point, plane); it is accumulated in 64-bit FIXED POINT with integer atomics --
// integer addition is associative, so the result is bit-reproducible whatever order the hardware serves the atomics in, and
// there is no floating-point atomic anywhere.  The scale is a power of two chosen from max |dL/dout| and the point count so
// that the worst case (every point on one texel) cannot overflow: at 100 k points that leaves 2^-43 of the largest upstream
// value as resolution, 2^19 times finer than fp32.
#include "kernels.h"

namespace sr {

namespace {

constexpr int kAxisX[3] = {0, 1, 2};   // plane p samples (x, y) = (pts[kAxisX[p]], pts[kAxisY[p]]): xy, yz, zx
constexpr int kAxisY[3] = {1, 2, 0};

struct Corners {
    int idx[4];      // texel index y * W + x of nw, ne, sw, se; -1 outside the plane (zero padding)
    float w[4];      // bilinear weights, PyTorch's formulas: nw = (x1 - ix)(y1 - iy), ne = (ix - x0)(y1 - iy), ...
    float ix, iy, x0, y0;
};

__device__ __forceinline__ Corners corners_of(float gx, float gy, int H, int W) {
    Corners c;
    c.ix = ((gx + 1.0f) * (float)W - 1.0f) * 0.5f;
    c.iy = ((gy + 1.0f) * (float)H - 1.0f) * 0.5f;
    c.x0 = floorf(c.ix); c.y0 = floorf(c.iy);
    const float x1 = c.x0 + 1.0f, y1 = c.y0 + 1.0f;
    c.w[0] = (x1 - c.ix) * (y1 - c.iy); c.w[1] = (c.ix - c.x0) * (y1 - c.iy);
    c.w[2] = (x1 - c.ix) * (c.iy - c.y0); c.w[3] = (c.ix - c.x0) * (c.iy - c.y0);
    // NaN / huge coordinates fall outside every comparison: all four corners invalid
    const bool xin0 = c.x0 >= 0.0f && c.x0 <= (float)(W - 1), xin1 = x1 >= 0.0f && x1 <= (float)(W - 1);
    const bool yin0 = c.y0 >= 0.0f && c.y0 <= (float)(H - 1), yin1 = y1 >= 0.0f && y1 <= (float)(H - 1);
    const int xi = xin0 || xin1 ? (int)c.x0 : 0, yi = yin0 || yin1 ? (int)c.y0 : 0;
    c.idx[0] = (xin0 && yin0) ? yi * W + xi : -1;
    c.idx[1] = (xin1 && yin0) ? yi * W + xi + 1 : -1;
    c.idx[2] = (xin0 && yin1) ? (yi + 1) * W + xi : -1;
    c.idx[3] = (xin1 && yin1) ? (yi + 1) * W + xi + 1 : -1;
    return c;
}

// [3][C][HW] -> [3][HW][C]: one thread per texel; loads are coalesced per channel, stores are the thread's C contiguous floats
__global__ void __launch_bounds__(kBlock) k_tp_to_hwc(int C, int HW, const float* __restrict__ chw, float* __restrict__ hwc) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int p = blockIdx.y;
    if (t >= HW) return;
    const float* src = chw + (size_t)p * C * HW + t;
    float4* dst = reinterpret_cast<float4*>(hwc + ((size_t)p * HW + t) * C);
    for (int c = 0; c < C; c += 4)
        dst[c >> 2] = make_float4(src[(size_t)c * HW], src[(size_t)(c + 1) * HW], src[(size_t)(c + 2) * HW], src[(size_t)(c + 3) * HW]);
}

// thread = (point, plane, group of 4 channels): consecutive threads write consecutive 16-byte pieces of out[N][3 C]
__global__ void __launch_bounds__(kBlock) k_tp_forward(int N, int C, int H, int W, const float* __restrict__ hwc,
                                                       const float* __restrict__ pts, float* __restrict__ out) {
    const int cq = C >> 2;
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)N * 3 * cq) return;
    const int q = (int)(t % cq), p = (int)((t / cq) % 3), n = (int)(t / (3 * cq));
    const Corners c = corners_of(pts[3 * (size_t)n + kAxisX[p]], pts[3 * (size_t)n + kAxisY[p]], H, W);
    const float4* plane = reinterpret_cast<const float4*>(hwc + (size_t)p * H * W * C) + q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c.idx[k] >= 0) {
            const float4 v = plane[(size_t)c.idx[k] * cq];
            acc.x += v.x * c.w[k]; acc.y += v.y * c.w[k]; acc.z += v.z * c.w[k]; acc.w += v.w * c.w[k];
        }
    }
    reinterpret_cast<float4*>(out)[t] = acc;
}

// max |g| as float bits (non-negative floats order like unsigned integers): integer atomicMax, order-independent
__global__ void __launch_bounds__(kBlock) k_tp_absmax(long long count, const float* __restrict__ g, uint32_t* __restrict__ out_bits) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const float a = fabsf(g[i]);
        m = a > m ? a : m;     // a NaN never wins: its gradient contribution is dropped below as well
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if (lane_id() == 0 && m > 0.0f) atomicMax(out_bits, __float_as_uint(m));
}

// dL/dplanes: 4 corners x 4 channels per thread, 64-bit fixed-point integer atomics (see the header comment)
__global__ void __launch_bounds__(kBlock) k_tp_scatter(int N, int C, int H, int W, const float* __restrict__ pts,
                                                       const float* __restrict__ g, const uint32_t* __restrict__ gmax_bits,
                                                       unsigned long long* __restrict__ acc) {
    const int cq = C >> 2;
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)N * 3 * cq) return;
    const float gmax = __uint_as_float(*gmax_bits);
    if (!(gmax > 0.0f)) return;
    int eg, en;
    frexpf(gmax, &eg);                         // gmax < 2^eg
    frexpf(4.0f * (float)N + 1.0f, &en);       // contributions per texel <= 4 N < 2^en
    const double scale = ldexp(1.0, 62 - eg - en);
    const int q = (int)(t % cq), p = (int)((t / cq) % 3), n = (int)(t / (3 * cq));
    const Corners c = corners_of(pts[3 * (size_t)n + kAxisX[p]], pts[3 * (size_t)n + kAxisY[p]], H, W);
    const float4 gv = reinterpret_cast<const float4*>(g)[t];
    unsigned long long* plane = acc + (size_t)p * H * W * C + 4 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c.idx[k] >= 0) {
            unsigned long long* dst = plane + (size_t)c.idx[k] * C;
            const float v[4] = {gv.x * c.w[k], gv.y * c.w[k], gv.z * c.w[k], gv.w * c.w[k]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long fx = __double2ll_rn((double)v[j] * scale);   // NaN -> 0
                if (fx != 0) atomicAdd(dst + j, (unsigned long long)fx);      // two's complement: signed sums wrap correctly
            }
        }
    }
}

// fixed point [3][HW][C] -> float [3][C][HW]
__global__ void __launch_bounds__(kBlock) k_tp_finish(int N, int C, int HW, const unsigned long long* __restrict__ acc,
                                                      const uint32_t* __restrict__ gmax_bits, float* __restrict__ d_chw) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int p = blockIdx.y;
    if (t >= HW) return;
    const float gmax = __uint_as_float(*gmax_bits);
    double inv = 0.0;
    if (gmax > 0.0f) {
        int eg, en;
        frexpf(gmax, &eg);
        frexpf(4.0f * (float)N + 1.0f, &en);
        inv = ldexp(1.0, -(62 - eg - en));
    }
    const long long* src = reinterpret_cast<const long long*>(acc) + ((size_t)p * HW + t) * C;
    float* dst = d_chw + (size_t)p * C * HW + t;
    for (int c = 0; c < C; ++c) dst[(size_t)c * HW] = (float)((double)src[c] * inv);
}

// dL/dpoints: one thread per point, a gather over its 12 corners
__global__ void __launch_bounds__(kBlock) k_tp_backward_points(int N, int C, int H, int W, const float* __restrict__ hwc,
                                                               const float* __restrict__ pts, const float* __restrict__ g,
                                                               float* __restrict__ d_pts) {
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int cq = C >> 2;
    float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const Corners c = corners_of(pts[3 * (size_t)n + kAxisX[p]], pts[3 * (size_t)n + kAxisY[p]], H, W);
        const float4* plane = reinterpret_cast<const float4*>(hwc + (size_t)p * H * W * C);
        const float4* gv = reinterpret_cast<const float4*>(g + ((size_t)n * 3 + p) * C);
        // d out / d ix = (ne - nw)(y1 - iy) + (se - sw)(iy - y0);  d out / d iy = (sw - nw)(x1 - ix) + (se - ne)(ix - x0)
        const float fx1 = c.ix - c.x0, fx0 = 1.0f - fx1, fy1 = c.iy - c.y0, fy0 = 1.0f - fy1;
        float gx = 0.f, gy = 0.f;
        for (int q = 0; q < cq; ++q) {
            const float4 u = gv[q];
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = c.idx[k] >= 0 ? plane[(size_t)c.idx[k] * cq + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float dxv = (u.x * ((v[1].x - v[0].x) * fy0 + (v[3].x - v[2].x) * fy1) + u.y * ((v[1].y - v[0].y) * fy0 + (v[3].y - v[2].y) * fy1)) +
                              (u.z * ((v[1].z - v[0].z) * fy0 + (v[3].z - v[2].z) * fy1) + u.w * ((v[1].w - v[0].w) * fy0 + (v[3].w - v[2].w) * fy1));
            const float dyv = (u.x * ((v[2].x - v[0].x) * fx0 + (v[3].x - v[1].x) * fx1) + u.y * ((v[2].y - v[0].y) * fx0 + (v[3].y - v[1].y) * fx1)) +
                              (u.z * ((v[2].z - v[0].z) * fx0 + (v[3].z - v[1].z) * fx1) + u.w * ((v[2].w - v[0].w) * fx0 + (v[3].w - v[1].w) * fx1));
            gx += dxv; gy += dyv;
        }
        d[kAxisX[p]] += gx * (0.5f * (float)W);   // d ix / d x = W / 2
        d[kAxisY[p]] += gy * (0.5f * (float)H);
    }
    d_pts[3 * (size_t)n] = d[0]; d_pts[3 * (size_t)n + 1] = d[1]; d_pts[3 * (size_t)n + 2] = d[2];
}

}  // namespace

int launch_triplane_forward(int N, int C, int H, int W, const float* planes_chw, float* planes_hwc, const float* pts, float* out,
                            hipStream_t st) {
    if (C <= 0 || (C & 3) || H <= 0 || W <= 0 || (size_t)H * W > (1u << 30)) return 1;
    const int HW = H * W;
    hipLaunchKernelGGL(k_tp_to_hwc, dim3((HW + kBlock - 1) / kBlock, 3), dim3(kBlock), 0, st, C, HW, planes_chw, planes_hwc);
    if (N > 0) {
        const long long threads = (long long)N * 3 * (C >> 2);
        hipLaunchKernelGGL(k_tp_forward, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, N, C, H, W, planes_hwc, pts, out);
    }
    return 0;
}

// `fixed`: 3 H W C 64-bit accumulators + one 8-byte word for max |g| behind them (cleared here)
int launch_triplane_backward(int N, int C, int H, int W, const float* planes_hwc, const float* pts, const float* g, float* d_planes_chw,
                             float* d_pts, void* fixed, hipStream_t st) {
    if (C <= 0 || (C & 3) || H <= 0 || W <= 0 || (size_t)H * W > (1u << 30)) return 1;
    const int HW = H * W;
    if (d_planes_chw) {
        const size_t words = (size_t)3 * HW * C;
        unsigned long long* acc = static_cast<unsigned long long*>(fixed);
        uint32_t* gmax = reinterpret_cast<uint32_t*>(acc + words);
        if (hipMemsetAsync(fixed, 0, (words + 1) * sizeof(unsigned long long), st) != hipSuccess) return 2;
        if (N > 0) {
            const long long count = (long long)N * 3 * C;
            const int blocks = (int)((count + kBlock * 8 - 1) / (kBlock * 8) < 2048 ? (count + kBlock * 8 - 1) / (kBlock * 8) : 2048);
            hipLaunchKernelGGL(k_tp_absmax, dim3(blocks > 0 ? blocks : 1), dim3(kBlock), 0, st, count, g, gmax);
            const long long threads = (long long)N * 3 * (C >> 2);
            hipLaunchKernelGGL(k_tp_scatter, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, N, C, H, W, pts, g, gmax, acc);
        }
        hipLaunchKernelGGL(k_tp_finish, dim3((HW + kBlock - 1) / kBlock, 3), dim3(kBlock), 0, st, N, C, HW, acc, gmax, d_planes_chw);
    }
    if (d_pts && N > 0)
        hipLaunchKernelGGL(k_tp_backward_points, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, C, H, W, planes_hwc, pts, g, d_pts);
    return 0;
}

}  // namespace sr
