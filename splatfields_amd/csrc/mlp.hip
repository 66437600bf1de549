// Fused forward of the small MLPs of the SplatFields deform network for gfx950.
//
// What it replaces: the forward of reference utils/time_utils.py:123-191 (`GeneralMLP`: Linear -> activation for every
// layer, the input concatenated back in front of the hidden state after the `skips` layers), which PyTorch-ROCm runs as
// one GEMM + several elementwise kernels per layer with every activation round-tripping through HBM (DESIGN.md section 8:
// the network is 98 % of a 4-D step).  Here a workgroup carries 128 points through ALL layers: activations never leave
// the registers, weights stream through LDS once per 128 points.
//
// MI355X mapping: Y^T[out x points] = W[out x in] . X^T[in x points] on `v_mfma_f32_16x16x4_f32` (exact fp32 fma chains),
// points in the N dimension (lanes n = lane & 15), so the accumulator of one layer -- lane (k, n), tile t, register i =
// output channel 16 t + 4 k + i of point n -- is EXACTLY the B operand of the next layer once the weights are packed with
// their input index permuted to (t, i, k): step (t, i) of the K loop takes channel 16 t + 4 k + i from lane group k.
// Bias is the accumulator's initial value, the activation a per-register operation.  A wavefront holds 2 x 16 points;
// a workgroup's 4 wavefronts share the weight chunks (2 channel tiles = 32 input channels at a time, double-buffered).
//
// Packed weights of a layer with MT output tiles and KT input tiles (host side: splatfields_amd/fused_mlp.py):
//   float index ((((c * MT + mt) * 2 + tl) * 64 + lane) * 4 + i)  =  W[16 mt + (lane & 15)][16 (2 c + tl) + 4 (lane >> 4) + i]
// (zero beyond the layer's true sizes), bias padded to 16 MT floats.
#include "kernels.h"

namespace sr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMlpMaxLayers = 12;
struct MlpLayerK { const float* w; const float* b; int mt, mem_t, reg_t; };   // output tiles; input tiles from x0 / from the previous layer
struct MlpK { int n_layers; MlpLayerK layer[kMlpMaxLayers]; };

template <int HT>
__global__ void __launch_bounds__(kBlock) k_mlp_forward(const MlpK net, int n_points, const float* __restrict__ x0, int x0_row,
                                                        float* __restrict__ y, int out_features, float slope) {
    __shared__ float4 s_w[2][HT * 2 * 64];   // two chunks of packed weights: [mt][tl][lane]
    const int wave = wave_id(), lane = lane_id();
    const int k = lane >> 4, n = lane & 15;
    const int p0 = (blockIdx.x * 4 + wave) * 32;              // first point of this wavefront
    int prow[2];                                              // this lane's point of each of the two point tiles (clamped)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) prow[nt] = min(p0 + 16 * nt + n, n_points - 1);

    f32x4 prev[2][HT], acc[2][HT];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < HT; ++t) prev[nt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int buf = 0;
    for (int l = 0; l < net.n_layers; ++l) {
        const MlpLayerK L = net.layer[l];
        const float4* w4 = reinterpret_cast<const float4*>(L.w);
        const int chunk_f4 = L.mt * 128;                       // float4 per chunk: mt x 2 x 64
        const int n_chunks = (L.mem_t + L.reg_t) / 2;
        auto stage = [&](int c, int into) {
            for (int j = (int)threadIdx.x; j < chunk_f4; j += kBlock) s_w[into][j] = w4[(size_t)c * chunk_f4 + j];
        };
        // accumulators start at the bias: rows 4k..4k+3 of output tile mt
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (mt < L.mt) { const float4 t4 = reinterpret_cast<const float4*>(L.b)[4 * mt + k]; b4 = (f32x4){t4.x, t4.y, t4.z, t4.w}; }
            acc[0][mt] = b4; acc[1][mt] = b4;
        }
        __syncthreads();          // everyone has left the previous layer's last chunk
        stage(0, buf);
        int c = 0;
        // ---- input channels that come from memory (the network input; also the skip connection) ----
        for (; c < L.mem_t / 2; ++c) {
            __syncthreads();      // chunk c has landed; the other buffer is free
            if (c + 1 < n_chunks) stage(c + 1, buf ^ 1);
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int t = 2 * c + tl;
                float4 b4[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) b4[nt] = *reinterpret_cast<const float4*>(x0 + (size_t)prow[nt] * x0_row + 16 * t + 4 * k);
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) {
                    if (mt < L.mt) {
                        const float4 a4 = s_w[buf][(mt * 2 + tl) * 64 + lane];
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4[nt].x, acc[nt][mt], 0, 0, 0);
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4[nt].y, acc[nt][mt], 0, 0, 0);
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4[nt].z, acc[nt][mt], 0, 0, 0);
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4[nt].w, acc[nt][mt], 0, 0, 0);
                        }
                    }
                }
            }
            buf ^= 1;
        }
        // ---- input channels that are the previous layer's accumulators (compile-time register indices) ----
#pragma unroll
        for (int cr = 0; cr < HT / 2; ++cr) {
            if (cr < L.reg_t / 2) {
                __syncthreads();
                if (c + 1 < n_chunks) stage(c + 1, buf ^ 1);
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int t = 2 * cr + tl;
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) {
                        if (mt < L.mt) {
                            const float4 a4 = s_w[buf][(mt * 2 + tl) * 64 + lane];
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, prev[nt][t][0], acc[nt][mt], 0, 0, 0);
                                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, prev[nt][t][1], acc[nt][mt], 0, 0, 0);
                                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, prev[nt][t][2], acc[nt][mt], 0, 0, 0);
                                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, prev[nt][t][3], acc[nt][mt], 0, 0, 0);
                            }
                        }
                    }
                }
                buf ^= 1;
                ++c;
            }
        }
        // ---- activation (after EVERY layer, the last one included: time_utils.py:185-186); becomes the next layer's input ----
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < HT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = mt < L.mt ? acc[nt][mt][i] : 0.f;
                    prev[nt][mt][i] = fmaxf(x, slope * x);   // leaky ReLU, 0 <= slope < 1
                }
    }
    // ---- the last layer's activations leave: channel 16 mt + 4 k + i of point (nt, n) ----
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int p = p0 + 16 * nt + n;
        if (p < n_points) {
#pragma unroll
            for (int mt = 0; mt < HT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 16 * mt + 4 * k + i;
                    if (ch < out_features) y[(size_t)p * out_features + ch] = prev[nt][mt][i];
                }
        }
    }
}

}  // namespace

int launch_mlp_forward(int n_points, int hidden_tiles, int n_layers, const SrMlpLayer* layers, const float* x0, int x0_row,
                       float* y, int out_features, float slope, hipStream_t st) {
    if (n_layers < 1 || n_layers > kMlpMaxLayers) return 1;
    MlpK net;
    net.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        const SrMlpLayer& s = layers[l];
        if (s.out_tiles < 1 || s.out_tiles > hidden_tiles || s.mem_tiles < 0 || s.reg_tiles < 0 || s.reg_tiles > hidden_tiles ||
            (s.mem_tiles & 1) || (s.reg_tiles & 1) || s.mem_tiles + s.reg_tiles < 2 || !s.w_packed || !s.bias) return 1;
        if (s.mem_tiles * 16 > x0_row) return 1;
        net.layer[l] = MlpLayerK{s.w_packed, s.bias, s.out_tiles, s.mem_tiles, s.reg_tiles};
    }
    if (out_features < 1 || out_features > 16 * layers[n_layers - 1].out_tiles) return 1;
    if (n_points <= 0) return 0;
    const int blocks = (n_points + 127) / 128;
    if (hidden_tiles == 8) hipLaunchKernelGGL((k_mlp_forward<8>), dim3(blocks), dim3(kBlock), 0, st, net, n_points, x0, x0_row, y, out_features, slope);
    else if (hidden_tiles == 4) hipLaunchKernelGGL((k_mlp_forward<4>), dim3(blocks), dim3(kBlock), 0, st, net, n_points, x0, x0_row, y, out_features, slope);
    else return 1;
    return 0;
}

}  // namespace sr
