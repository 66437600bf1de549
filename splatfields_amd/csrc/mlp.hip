// Fused MLP kernels of the SplatFields deform network for gfx950: the layer chain (forward, and the
// activation-gradient chain of the backward), the weight packer, and the weight-gradient kernel.
//
// What it replaces: reference utils/time_utils.py:123-191 (`GeneralMLP`: Linear -> activation for every layer, the
// input concatenated back in front of the hidden state after the `skips` layers), which PyTorch-ROCm runs as one GEMM +
// several elementwise kernels per layer with every activation round-tripping through HBM (the
// network is 98 % of a 4-D step).  Here a workgroup carries 128 points through ALL layers: activations never leave the
// registers, weights stream through LDS once per 128 points.
//
// MI355X mapping: Y^T[out x points] = W[out x in] . X^T[in x points] on `v_mfma_f32_16x16x4_f32` (exact fp32 fma chains),
// points in the N dimension (lanes n = lane & 15), so the accumulator of one layer -- lane (k, n), tile t, register i =
// output channel 16 t + 4 k + i of point n -- is EXACTLY the B operand of the next layer once the weights are packed with
// their input index permuted to (t, i, k): step (t, i) of the K loop takes channel 16 t + 4 k + i from lane group k.
// Bias is the accumulator's initial value, the activation a per-register operation.  A wavefront holds 2 x 16 points;
// a workgroup's 4 wavefronts share the weight chunks (2 channel tiles = 32 input channels at a time, double-buffered).
//
// The kernel executes a list of OPS; an op is one matrix applied to [memory channels | the register state]:
//   forward layer     acc = b + W [x0 | h];  h <- leaky(acc);  optionally stored (the backward's saved activations, the output)
//   backward, hidden  acc = W_h^T dZ;        dZ <- acc * leaky'(saved activation of the layer below);  stored for dW = dZ^T X
//   backward, input   acc = W_x^T dZ;        added to dL/dx0 in memory, the register state is kept
// so the same code walks the network in both directions (the backward's matrices are the transposed blocks, packed the same
// way); the weight gradients are a second kernel over the stored dZ and activations (k_mlp_weight_grad below).
//
// Packed matrix with MT output tiles and KT input tiles (host side: splatfields_amd/fused_mlp.py):
//   float index ((((c * MT + mt) * 2 + tl) * 64 + lane) * 4 + i)  =  A[16 mt + (lane & 15)][16 (2 c + tl) + 4 (lane >> 4) + i]
// (zero beyond the matrix's true sizes), bias padded to 16 MT floats.
#include "kernels.h"

namespace sr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMlpMaxOps = SR_MLP_MAX_OPS;
struct MlpK { int n_ops; SrMlpOp op[kMlpMaxOps]; };

// One 16-channel input tile (4 MFMA K steps) against all output tiles.  `w` points at this lane's float4 of output tile 0
// (consecutive output tiles are 128 float4 apart).  FULL (every one of the HT output tiles is live, the common case): the LDS
// reads of SR_MLP_GROUP output tiles are issued together and the MFMAs rotate over twice as many accumulators, so neither the LDS latency
// nor the dependent accumulate chains stall the matrix pipe; otherwise tile by tile under a wave-uniform test.
#ifndef SR_MLP_GROUP
#define SR_MLP_GROUP 2      // output tiles whose LDS reads are issued together (the MFMAs rotate over 2 x this many accumulators);
                            // 4 costs 8 more registers at the 256-register ceiling of the 128-wide kernel: measured 3-5 % slower
#endif
#ifndef SR_MLP_WAVES8
#define SR_MLP_WAVES8 2     // wavefronts per SIMD the 128-wide kernel is compiled for (256 / 168 VGPRs)
#endif
template <int HT, bool FULL>
__device__ __forceinline__ void mlp_tile(f32x4 (&acc)[2][HT], const float4* w, int out_tiles, const f32x4 b0, const f32x4 b1) {
    if constexpr (FULL) {
#pragma unroll
        for (int h = 0; h < HT; h += SR_MLP_GROUP) {
            float4 a[SR_MLP_GROUP];
#pragma unroll
            for (int m = 0; m < SR_MLP_GROUP; ++m) a[m] = w[(h + m) * 128];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < SR_MLP_GROUP; ++m) {
                    const float av = i == 0 ? a[m].x : i == 1 ? a[m].y : i == 2 ? a[m].z : a[m].w;
                    acc[0][h + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0[i], acc[0][h + m], 0, 0, 0);
                    acc[1][h + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1[i], acc[1][h + m], 0, 0, 0);
                }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
            if (mt < out_tiles) {
                const float4 a4 = w[mt * 128];
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0[0], acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1[0], acc[1][mt], 0, 0, 0);
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0[1], acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1[1], acc[1][mt], 0, 0, 0);
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0[2], acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1[2], acc[1][mt], 0, 0, 0);
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0[3], acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1[3], acc[1][mt], 0, 0, 0);
            }
        }
    }
}

// Epilogue of the common ops -- every output tile live, all 32 points of the wavefront valid, the store (if any) covering whole
// aligned float4 rows and not accumulating, the result replacing the state: per tile 4 activations + one 16-byte store, the
// only branches wave-uniform (scalar) ones on the op's kind.  (The generic epilogue below decides validity, vector or scalar
// store, accumulation and the channel bound per tile and per channel at run time; the compiler turned that into ~20 branches
// and exec-mask sequences per tile, a quarter of the kernel's time with stores.)
template <int HT>
__device__ __forceinline__ void mlp_epilogue_plain(const f32x4 (&acc)[2][HT], f32x4 (&prev)[2][HT], const SrMlpOp& L, float slope,
                                                   const uint32_t (&mbits)[2], int p0, int n, int k) {
    const bool leaky = L.epilogue == SR_MLP_LEAKY, has_store = L.store != nullptr, has_signs = L.sign_store != nullptr;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* dst = L.store + (size_t)(p0 + 16 * nt + n) * L.store_row + 4 * k;
        uint32_t signs = 0u;                                // bit 4 mt + i: channel 16 mt + 4 k + i of this point is > 0
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
            f32x4 r = acc[nt][mt];
            if (leaky) {
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = fmaxf(r[i], slope * r[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] *= (mbits[nt] >> (4 * mt + i)) & 1u ? 1.0f : slope;
            }
            if (has_signs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) signs |= (r[i] > 0.f ? 1u : 0u) << (4 * mt + i);
            }
            if (has_store) *reinterpret_cast<float4*>(dst + 16 * mt) = make_float4(r[0], r[1], r[2], r[3]);
            prev[nt][mt] = r;
        }
        if (has_signs) L.sign_store[(size_t)(p0 + 16 * nt + n) * 4 + k] = signs;
    }
}

template <int HT>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(HT == 8 ? SR_MLP_WAVES8 : 3, HT == 8 ? SR_MLP_WAVES8 : 3))) k_mlp_chain(const MlpK net, int n_points, float slope) {
    __shared__ float4 s_w[2][HT * 2 * 64];   // two chunks of a packed matrix: [mt][tl][lane]
    const int wave = wave_id(), lane = lane_id();
    const int k = lane >> 4, n = lane & 15;
    const int p0 = (blockIdx.x * 4 + wave) * 32;              // first point of this wavefront
    int prow[2];                                              // this lane's point of each of the two point tiles (clamped)
    bool pvalid[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { prow[nt] = min(p0 + 16 * nt + n, n_points - 1); pvalid[nt] = p0 + 16 * nt + n < n_points; }

    f32x4 prev[2][HT], acc[2][HT];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < HT; ++t) prev[nt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int buf = 0;
    for (int l = 0; l < net.n_ops; ++l) {
        const SrMlpOp L = net.op[l];
        const float4* w4 = reinterpret_cast<const float4*>(L.w_packed);
        const int chunk_f4 = L.out_tiles * 128;                // float4 per chunk: mt x 2 x 64
        const int n_chunks = (L.mem_tiles + L.reg_tiles) / 2;
        // A chunk travels memory -> LDS by LDS-direct loads (global_load_lds_dwordx4: 16 bytes per lane land at consecutive LDS
        // addresses, no staging registers, no ds_write) that are REQUESTED here and waited for at the top of the next chunk
        // (lds_copy_wait + barrier): the copy of chunk c + 1 runs under the 128 MFMAs of chunk c.  (Rounds 2-5 loaded the chunk into
        // registers and stored it: an in-order wavefront sat through the L2 round trip in front of every chunk's MFMAs -- it was
        // in its MFMA phases 55 % of its life.)
        auto stage = [&](int c, int into) {
            const float4* src = w4 + (size_t)c * chunk_f4;
            for (int j0 = 0; j0 < chunk_f4; j0 += kBlock) {           // chunk_f4 is a multiple of 128: whole wavefronts
                if (j0 + 64 * wave < chunk_f4)   // (the LDS base is a scalar operand: wave-uniform by construction, told to the compiler)
                    lds_copy16_async(src + j0 + (int)threadIdx.x, (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(&s_w[into][j0 + 64 * wave])));
            }
        };
        // accumulators start at the bias: rows 4k..4k+3 of output tile mt
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (L.bias && mt < L.out_tiles) { const float4 t4 = reinterpret_cast<const float4*>(L.bias)[4 * mt + k]; b4 = (f32x4){t4.x, t4.y, t4.z, t4.w}; }
            acc[0][mt] = b4; acc[1][mt] = b4;
        }
        // backward: leaky'(.) of the layer below as one bit per channel (written by the forward's sign_store): one dword per
        // point tile, requested here and used in the epilogue, instead of the 16 float4 of the saved activations themselves
        uint32_t mbits[2] = {0u, 0u};
        if (L.mask_bits) { mbits[0] = L.mask_bits[(size_t)prow[0] * 4 + k]; mbits[1] = L.mask_bits[(size_t)prow[1] * 4 + k]; }
        __syncthreads();          // everyone has left the previous op's last chunk
        stage(0, buf);
        int c = 0;
        // ---- input channels that come from memory (the network input / skip connection; the top gradient) ----
        const bool full = L.out_tiles == HT;                   // wave-uniform
        const bool store_vec = (L.store_row & 3) == 0 && (reinterpret_cast<uintptr_t>(L.store) & 15u) == 0;
        for (; c < L.mem_tiles / 2; ++c) {
            // this chunk's input channels: both tiles' loads go out before the barrier and the staging of the next chunk
            f32x4 bm[2][2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float4 t4 = *reinterpret_cast<const float4*>(L.src + (size_t)prow[nt] * L.src_row + 16 * (2 * c + tl) + 4 * k);
                    bm[tl][nt] = (f32x4){t4.x, t4.y, t4.z, t4.w};
                }
            lds_copy_wait();      // this wavefront's pieces of chunk c have landed ...
            __syncthreads();      // ... and everybody's; the other buffer is free
            if (c + 1 < n_chunks) stage(c + 1, buf ^ 1);
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                if (full) mlp_tile<HT, true>(acc, &s_w[buf][tl * 64 + lane], L.out_tiles, bm[tl][0], bm[tl][1]);
                else mlp_tile<HT, false>(acc, &s_w[buf][tl * 64 + lane], L.out_tiles, bm[tl][0], bm[tl][1]);
            }
            buf ^= 1;
        }
        // ---- input channels that are the register state (compile-time register indices) ----
#pragma unroll
        for (int cr = 0; cr < HT / 2; ++cr) {
            if (cr < L.reg_tiles / 2) {
                lds_copy_wait();
                __syncthreads();
                if (c + 1 < n_chunks) stage(c + 1, buf ^ 1);
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int t = 2 * cr + tl;
                    if (full) mlp_tile<HT, true>(acc, &s_w[buf][tl * 64 + lane], L.out_tiles, prev[0][t], prev[1][t]);
                    else mlp_tile<HT, false>(acc, &s_w[buf][tl * 64 + lane], L.out_tiles, prev[0][t], prev[1][t]);
                }
                buf ^= 1;
                ++c;
            }
        }
        // ---- epilogue: channel 16 mt + 4 k + i of point (nt, n) ----
        const bool plain = full && p0 + 32 <= n_points && !L.keep_state &&
                           (L.epilogue == SR_MLP_LEAKY || (L.epilogue == SR_MLP_MASK && L.mask_bits)) &&
                           (!L.store || (store_vec && L.store_channels == 16 * HT && !L.store_accumulate));   // wave-uniform
        if (plain) { mlp_epilogue_plain<HT>(acc, prev, L, slope, mbits, p0, n, k); continue; }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            uint32_t signs = 0u;                                // bit 4 mt + i: channel 16 mt + 4 k + i of this point is > 0
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) {
                f32x4 r = {0.f, 0.f, 0.f, 0.f};
                if (mt < L.out_tiles) {
                    r = acc[nt][mt];
                    if (L.epilogue == SR_MLP_LEAKY) {            // activation of a forward layer (leaky ReLU, 0 <= slope < 1)
#pragma unroll
                        for (int i = 0; i < 4; ++i) r[i] = fmaxf(r[i], slope * r[i]);
                    } else if (L.epilogue == SR_MLP_MASK) {      // backward: times leaky'(x), read off the saved activation's sign
                        if (L.mask_bits) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) r[i] *= (mbits[nt] >> (4 * mt + i)) & 1u ? 1.0f : slope;
                        } else {
                            const float4 m4 = *reinterpret_cast<const float4*>(L.mask + (size_t)prow[nt] * L.mask_row + 16 * mt + 4 * k);
                            r[0] *= m4.x > 0.f ? 1.0f : slope; r[1] *= m4.y > 0.f ? 1.0f : slope;
                            r[2] *= m4.z > 0.f ? 1.0f : slope; r[3] *= m4.w > 0.f ? 1.0f : slope;
                        }
                    }
                    if (L.sign_store) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) signs |= (r[i] > 0.f ? 1u : 0u) << (4 * mt + i);
                    }
                    if (L.store && pvalid[nt]) {
                        float* dst = L.store + (size_t)(p0 + 16 * nt + n) * L.store_row + 16 * mt + 4 * k;
                        if (store_vec && 16 * mt + 4 * k + 3 < L.store_channels) {       // whole float4 inside the row: one 16-byte access
                            float4 v = make_float4(r[0], r[1], r[2], r[3]);
                            if (L.store_accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                            *reinterpret_cast<float4*>(dst) = v;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (16 * mt + 4 * k + i < L.store_channels) dst[i] = L.store_accumulate ? dst[i] + r[i] : r[i];
                        }
                    }
                }
                if (!L.keep_state) prev[nt][mt] = r;
            }
            if (L.sign_store && pvalid[nt]) L.sign_store[(size_t)(p0 + 16 * nt + n) * 4 + k] = signs;
        }
    }
}

struct MlpPackK { SrMlpPackJob job[SR_MLP_MAX_PACK_JOBS]; };

// One packed float per thread step: thread index = the destination index, decoded to (c, mt, tl, lane, i) -> (row, column).
__global__ void __launch_bounds__(kBlock) k_mlp_pack(const MlpPackK jobs) {
    const SrMlpPackJob J = jobs.job[blockIdx.y];
    const int kt = (J.mem_pad + J.reg_width) / 16;
    const int total = 16 * J.out_tiles * 16 * kt;
    for (int d = blockIdx.x * kBlock + (int)threadIdx.x; d < total; d += gridDim.x * kBlock) {
        const int i = d & 3, lane = (d >> 2) & 63, tl = (d >> 8) & 1;
        const int rest = d >> 9;                       // c * MT + mt
        const int mt = rest % J.out_tiles, c = rest / J.out_tiles;
        const int r = 16 * mt + (lane & 15);
        const int cp = 16 * (2 * c + tl) + 4 * (lane >> 4) + i;
        int col = -1;
        if (cp < J.mem_pad) { if (cp < J.n_mem) col = J.mem_col0 + cp; }
        else if (cp - J.mem_pad < J.n_reg) col = J.reg_col0 + cp - J.mem_pad;
        float v = 0.0f;
        if (r < J.n_rows && col >= 0) v = J.transposed ? J.w[(size_t)col * J.ld + J.row0 + r] : J.w[(size_t)(J.row0 + r) * J.ld + col];
        J.dst[d] = v;
    }
    if (J.bias_dst && blockIdx.x == 0)
        for (int j = (int)threadIdx.x; j < 16 * J.out_tiles; j += kBlock) J.bias_dst[j] = j < J.n_bias ? J.bias_src[j] : 0.0f;
}

// ---- weight gradients: dW[m][c] = sum over points of dZ[p][m] X[p][c], db[m] = sum over points of dZ[p][m] -------------------------
// The contraction runs over the POINTS (10^5), the result is tiny (128 x 222): library GEMMs serialise it (0.27 ms for one
// 128 x 128 x 100k product on MI355X, 12 TFLOP/s; eight of them were 2.7 of the 3.9 ms of a GeneralMLP training step).  Here
// the points are cut into slabs, one workgroup column per slab; a wavefront owns one 64 x 64 block of one layer's dW and walks its
// slab four points per MFMA step.  Operands come straight from memory with NO transpose: lane (kk, n) loads float4 = channels
// 4n..4n+3 of point s + kk from dZ and from X (256 contiguous bytes per point row), and register i of the dZ load with
// register j of the X load feed tile (i, j), whose MFMA row m' stands for dZ channel 4 m' + i and column n' for X channel
// 4 n' + j: 2 loads per 16 MFMAs.  Partial blocks go to a workspace and a second kernel adds them over the slabs in a fixed
// order (deterministic, no float atomics).
constexpr int kGradMaxSlabs = 256;
struct GradTask { unsigned char job, bm, bk, bias; };
struct MlpGradK {
    int n_tasks, n_points, slab, n_slabs;
    float* part;          // [task][slab][64][64]
    float* bias_part;     // [task][slab][64]
    SrMlpGradJob job[SR_MLP_MAX_GRAD_JOBS];
    GradTask task[SR_MLP_MAX_GRAD_TASKS];
};

__global__ void __launch_bounds__(kBlock) k_mlp_weight_grad(const MlpGradK P) {
    const int t = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + wave_id());   // wave-uniform: the job's fields stay in SGPRs
    if (t >= P.n_tasks) return;                       // no barriers below
    const GradTask T = P.task[t];
    const SrMlpGradJob J = P.job[T.job];
    const int lane = lane_id(), kk = lane >> 4, n = lane & 15;
    const int s_begin = blockIdx.x * P.slab, s_end = min(P.n_points, s_begin + P.slab);
    // Buffer loads: a byte offset beyond the buffer reads as 0, so rows past the slab's end and channel groups past the row
    // width need no branch (the compiler turns "load, then zero if invalid" back into a branch around the load, which
    // serialises the four steps kept in flight below).
    const bool zok = 64 * T.bm + 4 * n + 3 < J.dz_row, xok = 64 * T.bk + 4 * n + 3 < J.x_row;
    const unsigned zcol = 4u * (64 * T.bm + 4 * n), xcol = 4u * (64 * T.bk + 4 * n);
    const unsigned zbytes = 4u * (unsigned)J.dz_row, xbytes = 4u * (unsigned)J.x_row;
    const __amdgpu_buffer_rsrc_t zsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(J.dz), 0, (int)(zbytes * (unsigned)P.n_points), 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(J.x), 0, (int)(xbytes * (unsigned)P.n_points), 0x00020000);
    auto ldz = [&](int p) -> float4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(zsrc, (zok && p < s_end) ? (unsigned)p * zbytes + zcol : 0xfffffff0u, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto ldx = [&](int p) -> float4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrc, (xok && p < s_end) ? (unsigned)p * xbytes + xcol : 0xfffffff0u, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
#define SR_GRAD_ROW(i, ai, b)                                                                \
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, b.x, acc[i][0], 0, 0, 0);        \
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, b.y, acc[i][1], 0, 0, 0);        \
        acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, b.z, acc[i][2], 0, 0, 0);        \
        acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, b.w, acc[i][3], 0, 0, 0);
#define SR_GRAD_STEP(a, b, next)                                                              \
        colsum.x += a.x; colsum.y += a.y; colsum.z += a.z; colsum.w += a.w;                    \
        SR_GRAD_ROW(0, a.x, b) SR_GRAD_ROW(1, a.y, b) SR_GRAD_ROW(2, a.z, b) SR_GRAD_ROW(3, a.w, b) \
        a = ldz(next); b = ldx(next);
    float4 a0 = ldz(s_begin + kk), b0 = ldx(s_begin + kk);
    float4 a1 = ldz(s_begin + 4 + kk), b1 = ldx(s_begin + 4 + kk);
    float4 a2 = ldz(s_begin + 8 + kk), b2 = ldx(s_begin + 8 + kk);
    float4 a3 = ldz(s_begin + 12 + kk), b3 = ldx(s_begin + 12 + kk);
    for (int s = s_begin; s < s_end; s += 16) {       // four steps of four points; steps past the slab's end add zeros
        SR_GRAD_STEP(a0, b0, s + 16 + kk)
        SR_GRAD_STEP(a1, b1, s + 20 + kk)
        SR_GRAD_STEP(a2, b2, s + 24 + kk)
        SR_GRAD_STEP(a3, b3, s + 28 + kk)
    }
#undef SR_GRAD_STEP
#undef SR_GRAD_ROW
    // lane (kk, n), tile (i, j), register r  =  block row 16 kk + 4 r + i, block column 4 n + j
    float* out = P.part + ((size_t)t * P.n_slabs + blockIdx.x) * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(out + (16 * kk + 4 * r + i) * 64 + 4 * n) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
    if (T.bias) {                                     // column sums of dZ: add the four point phases kk
        colsum.x += __shfl_xor(colsum.x, 16); colsum.y += __shfl_xor(colsum.y, 16); colsum.z += __shfl_xor(colsum.z, 16); colsum.w += __shfl_xor(colsum.w, 16);
        colsum.x += __shfl_xor(colsum.x, 32); colsum.y += __shfl_xor(colsum.y, 32); colsum.z += __shfl_xor(colsum.z, 32); colsum.w += __shfl_xor(colsum.w, 32);
        if (kk == 0) *reinterpret_cast<float4*>(P.bias_part + ((size_t)t * P.n_slabs + blockIdx.x) * 64 + 4 * n) = colsum;
    }
}

// Adds the partial blocks over the slabs in a FIXED order: a workgroup owns 64 consecutive entries of a task's block (or its
// 64 bias entries); wavefront q adds slabs q, q + 4, q + 8, ... (eight loads in flight), the four partial sums are added in
// order q = 0..3.  (One thread per entry walking all 256 slabs kept a 48 x 48 gradient waiting 65 us on 17 workgroups.)
__global__ void __launch_bounds__(kBlock) k_mlp_weight_grad_reduce(const MlpGradK P) {
    __shared__ float s_part[4][64];
    const int t = blockIdx.y;
    const GradTask T = P.task[t];
    const SrMlpGradJob J = P.job[T.job];
    const int q = wave_id(), e = blockIdx.x * 64 + lane_id();
    const bool bias_row = e >= 4096;
    if (bias_row && !T.bias) return;                  // block-uniform
    const float* src = bias_row ? P.bias_part + (size_t)t * P.n_slabs * 64 + (e - 4096) : P.part + (size_t)t * P.n_slabs * 4096 + e;
    const size_t stride = bias_row ? 64 : 4096;
    float sum = 0.0f;
    int sl = q;
    for (; sl + 28 < P.n_slabs; sl += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(sl + 4 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; sl < P.n_slabs; sl += 4) sum += src[(size_t)sl * stride];
    s_part[q][lane_id()] = sum;
    __syncthreads();
    if (q != 0) return;
    sum = ((s_part[0][lane_id()] + s_part[1][lane_id()]) + s_part[2][lane_id()]) + s_part[3][lane_id()];
    if (bias_row) {
        const int m = 64 * T.bm + (e - 4096);
        if (m < J.m) J.db[m] = sum;
    } else {
        const int m = 64 * T.bm + (e >> 6), c = 64 * T.bk + (e & 63);
        if (m < J.m && c < J.k) J.dw[(size_t)m * J.dw_row + J.dw_col0 + c] = sum;
    }
}

// Fills the task table; returns the number of tasks or -1.
static int grad_tasks(int n_points, int n_jobs, const SrMlpGradJob* jobs, MlpGradK* k) {
    if (n_jobs < 1 || n_jobs > SR_MLP_MAX_GRAD_JOBS) return -1;
    int nt = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const SrMlpGradJob& s = jobs[j];
        if (!s.dz || !s.x || !s.dw || s.m < 1 || s.k < 1 || s.m > s.dz_row || s.k > s.x_row || (s.dz_row & 3) || (s.x_row & 3) ||
            (reinterpret_cast<uintptr_t>(s.dz) & 15u) || (reinterpret_cast<uintptr_t>(s.x) & 15u) || s.dw_col0 < 0 || s.dw_col0 + s.k > s.dw_row ||
            s.m > 64 * 255 || s.k > 64 * 255) return -1;
        if (n_points > 0 && ((size_t)n_points * s.dz_row * 4 >= (1ull << 31) || (size_t)n_points * s.x_row * 4 >= (1ull << 31))) return -1;   // 32-bit buffer offsets
        if (k) k->job[j] = s;
        for (int bm = 0; bm < (s.m + 63) / 64; ++bm)
            for (int bk = 0; bk < (s.k + 63) / 64; ++bk) {
                if (nt >= SR_MLP_MAX_GRAD_TASKS) return -1;
                if (k) k->task[nt] = GradTask{(unsigned char)j, (unsigned char)bm, (unsigned char)bk, (unsigned char)(s.db && bk == 0)};
                ++nt;
            }
    }
    return nt;
}

static void grad_slabs(int n_points, int* slab, int* n_slabs) {
    const int per = (n_points + kGradMaxSlabs - 1) / kGradMaxSlabs;
    *slab = max(64, (per + 3) / 4 * 4);
    *n_slabs = max(1, (n_points + *slab - 1) / *slab);
}

}  // namespace

size_t mlp_weight_grad_workspace(int n_points, int n_jobs, const SrMlpGradJob* jobs) {
    const int nt = grad_tasks(n_points, n_jobs, jobs, nullptr);
    if (nt < 0 || n_points < 0) return 0;
    int slab, n_slabs;
    grad_slabs(n_points, &slab, &n_slabs);
    return (size_t)nt * n_slabs * (4096 + 64) * sizeof(float);
}

int launch_mlp_weight_grad(int n_points, int n_jobs, const SrMlpGradJob* jobs, void* workspace, size_t workspace_bytes, hipStream_t st) {
    MlpGradK k;
    const int nt = grad_tasks(n_points, n_jobs, jobs, &k);
    if (nt < 0 || n_points < 1) return 1;
    grad_slabs(n_points, &k.slab, &k.n_slabs);
    if (!workspace || workspace_bytes < (size_t)nt * k.n_slabs * (4096 + 64) * sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15u)) return 1;
    k.n_tasks = nt; k.n_points = n_points;
    k.part = static_cast<float*>(workspace);
    k.bias_part = k.part + (size_t)nt * k.n_slabs * 4096;
    hipLaunchKernelGGL(k_mlp_weight_grad, dim3(k.n_slabs, (nt + 3) / 4), dim3(kBlock), 0, st, k);
    hipLaunchKernelGGL(k_mlp_weight_grad_reduce, dim3((4096 + 64) / 64, nt), dim3(kBlock), 0, st, k);
    return 0;
}

int launch_mlp_pack(int n_jobs, const SrMlpPackJob* jobs, hipStream_t st) {
    if (n_jobs < 0 || n_jobs > SR_MLP_MAX_PACK_JOBS) return 1;
    if (n_jobs == 0) return 0;
    MlpPackK k;
    int most = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const SrMlpPackJob& s = jobs[j];
        if (!s.w || !s.dst || s.out_tiles < 1 || s.ld < 1 || s.row0 < 0 || s.n_rows < 0 || s.n_rows > 16 * s.out_tiles ||
            s.n_mem < 0 || s.n_mem > s.mem_pad || s.n_reg < 0 || s.n_reg > s.reg_width || (s.mem_pad & 31) || (s.reg_width & 15) ||
            (((s.mem_pad + s.reg_width) / 16) & 1) || s.mem_pad + s.reg_width < 32 || (reinterpret_cast<uintptr_t>(s.dst) & 15u) ||
            (s.bias_dst && (!s.bias_src || s.n_bias < 0 || s.n_bias > 16 * s.out_tiles))) return 1;
        k.job[j] = s;
        most = max(most, 16 * s.out_tiles * (s.mem_pad + s.reg_width));
    }
    const int bx = min(64, (most + kBlock * 4 - 1) / (kBlock * 4));
    hipLaunchKernelGGL(k_mlp_pack, dim3(bx, n_jobs), dim3(kBlock), 0, st, k);
    return 0;
}

int launch_mlp_chain(int n_points, int hidden_tiles, int n_ops, const SrMlpOp* ops, float slope, hipStream_t st) {
    if (n_ops < 1 || n_ops > kMlpMaxOps) return 1;
    MlpK net;
    net.n_ops = n_ops;
    for (int l = 0; l < n_ops; ++l) {
        const SrMlpOp& s = ops[l];
        if (s.out_tiles < 1 || s.out_tiles > hidden_tiles || s.mem_tiles < 0 || s.reg_tiles < 0 || s.reg_tiles > hidden_tiles ||
            (s.mem_tiles & 1) || (s.reg_tiles & 1) || s.mem_tiles + s.reg_tiles < 2 || !s.w_packed ||
            (reinterpret_cast<uintptr_t>(s.w_packed) & 15u) || (reinterpret_cast<uintptr_t>(s.bias) & 15u)) return 1;
        if (s.mem_tiles > 0 && (!s.src || s.mem_tiles * 16 > s.src_row || (s.src_row & 3) || (reinterpret_cast<uintptr_t>(s.src) & 15u))) return 1;
        if (s.epilogue != SR_MLP_NONE && s.epilogue != SR_MLP_LEAKY && s.epilogue != SR_MLP_MASK) return 1;
        if (s.epilogue == SR_MLP_MASK && !s.mask_bits && (!s.mask || s.out_tiles * 16 > s.mask_row || (s.mask_row & 3) || (reinterpret_cast<uintptr_t>(s.mask) & 15u))) return 1;
        if ((reinterpret_cast<uintptr_t>(s.mask_bits) & 3u) || (reinterpret_cast<uintptr_t>(s.sign_store) & 3u)) return 1;
        if (s.store && (s.store_channels < 1 || s.store_channels > s.store_row || s.store_channels > 16 * s.out_tiles)) return 1;
        net.op[l] = s;
    }
    if (n_points <= 0) return 0;
    const int blocks = (n_points + 127) / 128;
    if (hidden_tiles == 8) hipLaunchKernelGGL((k_mlp_chain<8>), dim3(blocks), dim3(kBlock), 0, st, net, n_points, slope);
    else if (hidden_tiles == 4) hipLaunchKernelGGL((k_mlp_chain<4>), dim3(blocks), dim3(kBlock), 0, st, net, n_points, slope);
    else return 1;
    return 0;
}

}  // namespace sr

// ------------------------------------------------------------------------------------------------------------------------
// ResField weights of ONE frame for all layers of a network (reference utils/resfields.py:185,229,294-300,378-405 in the
// configuration GeneralMLP builds: compression 'vm', mode 'lookup', fuse 'add'):
//     W_eff[j] = W[j] + sum_k weights_t[frame][k] * matrix_t[k][j],      j < out * in
// The reference materialises the [capacity, out * in] matrix of EVERY frame per call and indexes it; composing only the
// current frame with PyTorch ops costs an index, a [1 x rank] x [rank x out in] product and an add per layer, and twice that
// backward -- about 240 small launches per step of the six-network deform field.  Here: one launch per network forward,
// two backward (d matrix_t = coeff (x) dW_eff written in place of an outer-product kernel per layer; d weights_t[frame] =
// matrix_t . dW_eff by a fixed-order two-stage reduction; d W = dW_eff needs no kernel at all).  Memory-bound: matrix_t is
// rank x out x in floats (4.5 MB for a 128 x 222 layer at the reference's rank 40).
// ------------------------------------------------------------------------------------------------------------------------
namespace sr {

struct ResFieldK { SrResFieldJob job[SR_RESFIELD_MAX_JOBS]; const long long* frame; float* part; int blocks_per_job; };

constexpr int kRfPerBlock = kBlock * 4;   // elements of W per workgroup

// frame index as the reference's `mat[frame_id]` reads it: indices in [-capacity, 0) wrap (Python indexing; general_mlp.py's
// host-side check wraps the same way), everything else outside [0, capacity) is an error
__device__ __forceinline__ long long resfield_frame(long long f, int capacity) {
    return (f < 0 && f >= -(long long)capacity) ? f + (long long)capacity : f;
}

__global__ void __launch_bounds__(kBlock) k_resfield_compose(const ResFieldK P) {
    const SrResFieldJob J = P.job[blockIdx.y];
    const int j4 = blockIdx.x * kBlock + (int)threadIdx.x;
    if (4 * j4 >= J.count) return;
    const long long f = resfield_frame(*P.frame, J.capacity);
    if (f < 0 || f >= (long long)J.capacity) {
        // the reference's `mat[frame_id]` raises IndexError here; a kernel cannot, and reading weights_t out of bounds would
        // compose garbage silently: poison the weights instead, so that the network's outputs are NaN from this step on
        const float nan = __builtin_nanf("");
        reinterpret_cast<float4*>(J.out)[j4] = make_float4(nan, nan, nan, nan);
        return;
    }
    const float* coeff = J.weights_t + (size_t)f * J.rank;
    const int q = J.count >> 2;
    const float4* M = reinterpret_cast<const float4*>(J.matrix_t) + j4;
    float4 acc = reinterpret_cast<const float4*>(J.w)[j4];
    int k = 0;
    for (; k + 4 <= J.rank; k += 4) {      // four independent loads in flight
        const float4 m0 = M[(size_t)k * q], m1 = M[(size_t)(k + 1) * q], m2 = M[(size_t)(k + 2) * q], m3 = M[(size_t)(k + 3) * q];
        const float c0 = coeff[k], c1 = coeff[k + 1], c2 = coeff[k + 2], c3 = coeff[k + 3];
        acc.x += c0 * m0.x + c1 * m1.x + c2 * m2.x + c3 * m3.x; acc.y += c0 * m0.y + c1 * m1.y + c2 * m2.y + c3 * m3.y;
        acc.z += c0 * m0.z + c1 * m1.z + c2 * m2.z + c3 * m3.z; acc.w += c0 * m0.w + c1 * m1.w + c2 * m2.w + c3 * m3.w;
    }
    for (; k < J.rank; ++k) {
        const float4 m = M[(size_t)k * q];
        const float c = coeff[k];
        acc.x += c * m.x; acc.y += c * m.y; acc.z += c * m.z; acc.w += c * m.w;
    }
    reinterpret_cast<float4*>(J.out)[j4] = acc;
}

// d_matrix_t[k][j] = coeff[k] * g[j]; partial sums of matrix_t[k][.] . g over this workgroup's 1024 elements -> part
__global__ void __launch_bounds__(kBlock) k_resfield_backward(const ResFieldK P) {
    __shared__ float s_part[4][SR_RESFIELD_MAX_RANK];
    const SrResFieldJob J = P.job[blockIdx.y];
    if (blockIdx.x * kRfPerBlock >= J.count) return;   // whole workgroup
    const int j4 = blockIdx.x * kBlock + (int)threadIdx.x;
    const bool in = 4 * j4 < J.count;
    const long long f_raw = resfield_frame(*P.frame, J.capacity);
    const bool f_ok = f_raw >= 0 && f_raw < (long long)J.capacity;   // out of range: the forward poisoned W_eff with NaN (above);
    const long long f = f_ok ? f_raw : 0;                             //   here: no out-of-bounds read, NaN gradients
    const float* coeff = J.weights_t + (size_t)f * J.rank;
    const int q = J.count >> 2;
    float4 g = in ? reinterpret_cast<const float4*>(J.d_out)[j4] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (!f_ok) g.x = g.y = g.z = g.w = __builtin_nanf("");
    for (int k = 0; k < J.rank; ++k) {
        float dot = 0.0f;
        if (in) {
            const float4 m = reinterpret_cast<const float4*>(J.matrix_t)[(size_t)k * q + j4];
            const float c = coeff[k];
            if (J.d_matrix_t) reinterpret_cast<float4*>(J.d_matrix_t)[(size_t)k * q + j4] = make_float4(c * g.x, c * g.y, c * g.z, c * g.w);
            dot = (m.x * g.x + m.y * g.y) + (m.z * g.z + m.w * g.w);
        }
        const float tot = wave_sum_to_lane63(dot);
        if (lane_id() == 63) s_part[wave_id()][k] = tot;
    }
    __syncthreads();
    if ((int)threadIdx.x < J.rank)
        P.part[((size_t)blockIdx.y * P.blocks_per_job + blockIdx.x) * SR_RESFIELD_MAX_RANK + threadIdx.x] =
            (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

// d_weights_t [capacity, rank]: zero except row `frame` = sum of the workgroups' partial sums in launch order
__global__ void __launch_bounds__(kBlock) k_resfield_finish(const ResFieldK P) {
    const SrResFieldJob J = P.job[blockIdx.x];
    if (!J.d_weights_t) return;
    const long long f = resfield_frame(*P.frame, J.capacity);
    const int nb = (J.count + kRfPerBlock - 1) / kRfPerBlock;
    for (int i = threadIdx.x; i < J.capacity * J.rank; i += kBlock) {
        float v = 0.0f;
        if (i / J.rank == (int)f) {
            const int k = i - (int)f * J.rank;
            for (int b = 0; b < nb; ++b) v += P.part[((size_t)blockIdx.x * P.blocks_per_job + b) * SR_RESFIELD_MAX_RANK + k];
        }
        J.d_weights_t[i] = v;
    }
}

static int resfield_check(int n_jobs, const SrResFieldJob* jobs, bool backward, int* max_count) {
    if (n_jobs < 1 || n_jobs > SR_RESFIELD_MAX_JOBS || !jobs) return 1;
    int mc = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const SrResFieldJob& s = jobs[j];
        if (s.count < 4 || (s.count & 3) || s.rank < 1 || s.rank > SR_RESFIELD_MAX_RANK || s.capacity < 1 || !s.weights_t || !s.matrix_t) return 1;
        if ((reinterpret_cast<uintptr_t>(s.matrix_t) & 15u)) return 1;
        if (!backward && (!s.w || !s.out || ((reinterpret_cast<uintptr_t>(s.w) | reinterpret_cast<uintptr_t>(s.out)) & 15u))) return 1;
        if (backward && (!s.d_out || (reinterpret_cast<uintptr_t>(s.d_out) & 15u) || (reinterpret_cast<uintptr_t>(s.d_matrix_t) & 15u))) return 1;
        mc = s.count > mc ? s.count : mc;
    }
    *max_count = mc;
    return 0;
}

int launch_resfield_compose(int n_jobs, const SrResFieldJob* jobs, const long long* frame, hipStream_t st) {
    int mc;
    if (!frame || resfield_check(n_jobs, jobs, false, &mc)) return 1;
    ResFieldK P;
    for (int j = 0; j < n_jobs; ++j) P.job[j] = jobs[j];
    P.frame = frame; P.part = nullptr; P.blocks_per_job = (mc + kRfPerBlock - 1) / kRfPerBlock;
    hipLaunchKernelGGL(k_resfield_compose, dim3(P.blocks_per_job, n_jobs), dim3(kBlock), 0, st, P);
    return 0;
}

size_t resfield_backward_workspace(int n_jobs, const SrResFieldJob* jobs) {
    int mc;
    if (resfield_check(n_jobs, jobs, true, &mc)) return 0;
    return (size_t)n_jobs * ((mc + kRfPerBlock - 1) / kRfPerBlock) * SR_RESFIELD_MAX_RANK * sizeof(float);
}

int launch_resfield_backward(int n_jobs, const SrResFieldJob* jobs, const long long* frame, void* workspace, size_t workspace_bytes, hipStream_t st) {
    int mc;
    if (!frame || !workspace || resfield_check(n_jobs, jobs, true, &mc)) return 1;
    if (workspace_bytes < resfield_backward_workspace(n_jobs, jobs)) return 1;
    ResFieldK P;
    for (int j = 0; j < n_jobs; ++j) P.job[j] = jobs[j];
    P.frame = frame; P.part = static_cast<float*>(workspace); P.blocks_per_job = (mc + kRfPerBlock - 1) / kRfPerBlock;
    hipLaunchKernelGGL(k_resfield_backward, dim3(P.blocks_per_job, n_jobs), dim3(kBlock), 0, st, P);
    hipLaunchKernelGGL(k_resfield_finish, dim3(n_jobs), dim3(kBlock), 0, st, P);
    return 0;
}

}  // namespace sr

// ------------------------------------------------------------------------------------------------------------------------
// Network input of a GeneralMLP (reference utils/time_utils.py:9-57 `get_embedder` + :178-181): row p of the padded input
// matrix = [ x (3) | sin(2^0 x) (3) | cos(2^0 x) (3) | ... | sin(2^(L-1) x) | cos(2^(L-1) x) | features (F) | time encoding | 0 ... ],
// time encoding = [ t | sin(2^0 t) | cos(2^0 t) | ... | sin(2^(TL-1) t) | cos(2^(TL-1) t) ] when a time per point is given
// (reference utils/time_utils.py:455-456: the time embedding is the tail of the feature vector every network receives).
// One launch instead of the reference's 2 L multiplies, 2 L sin / cos, two concatenations and the padding copy (and as many
// again backward): ~60 small kernels per network and step.
// ------------------------------------------------------------------------------------------------------------------------
namespace sr {

__global__ void __launch_bounds__(kBlock) k_mlp_input_forward(int N, int L, int F, int TL, int row, const float* __restrict__ xyz,
                                                              const float* __restrict__ feat, const float* __restrict__ time,
                                                              float* __restrict__ x0) {
    const int q = row >> 2;
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)N * q) return;
    const int p = (int)(t / q), c0 = 4 * (int)(t % q);
    const float x[3] = {xyz[3 * (size_t)p], xyz[3 * (size_t)p + 1], xyz[3 * (size_t)p + 2]};
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + i;
        float o = 0.0f;
        if (c < 3) o = x[c];
        else if (c < 3 + 6 * L) {
            const int k = c - 3, j = k / 6, r = k - 6 * j;
            const float a = x[r < 3 ? r : r - 3] * (float)(1 << j);
            o = r < 3 ? sinf(a) : cosf(a);
        } else if (c < 3 + 6 * L + F) o = feat[(size_t)p * F + (c - 3 - 6 * L)];
        else if (time && c < 3 + 6 * L + F + 1 + 2 * TL) {
            const int m = c - (3 + 6 * L + F);
            const float tv = time[p];
            if (m == 0) o = tv;
            else { const float a = tv * (float)(1 << ((m - 1) >> 1)); o = ((m - 1) & 1) ? cosf(a) : sinf(a); }
        }
        v[i] = o;
    }
    reinterpret_cast<float4*>(x0)[t] = make_float4(v[0], v[1], v[2], v[3]);
}

// d_xyz[p] = g[0:3] + sum_j 2^j (cos(2^j x) g_sin_j - sin(2^j x) g_cos_j);  d_feat[p] = g[3 + 6 L : 3 + 6 L + F]
__global__ void __launch_bounds__(kBlock) k_mlp_input_backward(int N, int L, int F, int row, const float* __restrict__ xyz,
                                                               const float* __restrict__ g, float* __restrict__ d_xyz, float* __restrict__ d_feat) {
    const int per = 1 + (F + 3) / 4;   // thread 0 of a point: the position gradient; the others: 4 feature columns each
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)N * per) return;
    const int p = (int)(t / per), part = (int)(t % per);
    const float* gp = g + (size_t)p * row;
    if (part == 0) {
        if (!d_xyz) return;
        float d[3] = {gp[0], gp[1], gp[2]};
        const float x[3] = {xyz[3 * (size_t)p], xyz[3 * (size_t)p + 1], xyz[3 * (size_t)p + 2]};
        for (int j = 0; j < L; ++j) {
            const float f = (float)(1 << j);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float sn, cs;
                sincosf(x[r] * f, &sn, &cs);
                d[r] += f * (cs * gp[3 + 6 * j + r] - sn * gp[3 + 6 * j + 3 + r]);
            }
        }
        d_xyz[3 * (size_t)p] = d[0]; d_xyz[3 * (size_t)p + 1] = d[1]; d_xyz[3 * (size_t)p + 2] = d[2];
    } else if (d_feat) {
        const int f0 = 4 * (part - 1);
        for (int i = f0; i < min(f0 + 4, F); ++i) d_feat[(size_t)p * F + i] = gp[3 + 6 * L + i];
    }
}

int launch_mlp_input_forward(int N, int L, int F, int TL, int row, const float* xyz, const float* feat, const float* time, float* x0,
                             hipStream_t st) {
    if (N < 0 || L < 0 || L > 16 || F < 0 || TL < 0 || TL > 16 || (row & 3) || row < 3 + 6 * L + F + (time ? 1 + 2 * TL : 0) ||
        (F > 0 && !feat) || !xyz || !x0) return 1;
    if (N == 0) return 0;
    const long long threads = (long long)N * (row >> 2);
    hipLaunchKernelGGL(k_mlp_input_forward, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, N, L, F, TL, row, xyz, feat, time, x0);
    return 0;
}

// Top of a network's backward: G[p][c] = dL/dy[p][c] * leaky'(y[p][c]) for c < out, 0 up to the padded row (the memory input of
// the backward chain's first op and the dZ operand of the last layer's weight gradient).
__global__ void __launch_bounds__(kBlock) k_mlp_top_gradient(int N, int out, int row, const float* __restrict__ y, const float* __restrict__ dy,
                                                             float slope, float* __restrict__ G) {
    const int q = row >> 2;
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)N * q) return;
    const int p = (int)(t / q), c0 = 4 * (int)(t % q);
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + i;
        v[i] = c < out ? dy[(size_t)p * out + c] * (y[(size_t)p * out + c] > 0.f ? 1.0f : slope) : 0.0f;
    }
    reinterpret_cast<float4*>(G)[t] = make_float4(v[0], v[1], v[2], v[3]);
}

int launch_mlp_top_gradient(int N, int out, int row, const float* y, const float* dy, float slope, float* G, hipStream_t st) {
    if (N < 0 || out < 1 || (row & 3) || row < out || !y || !dy || !G) return 1;
    if (N == 0) return 0;
    const long long threads = (long long)N * (row >> 2);
    hipLaunchKernelGGL(k_mlp_top_gradient, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, N, out, row, y, dy, slope, G);
    return 0;
}

int launch_mlp_input_backward(int N, int L, int F, int row, const float* xyz, const float* g, float* d_xyz, float* d_feat, hipStream_t st) {
    if (N < 0 || L < 0 || L > 16 || F < 0 || row < 3 + 6 * L + F || !xyz || !g) return 1;
    if (N == 0) return 0;
    const long long threads = (long long)N * (1 + (F + 3) / 4);
    hipLaunchKernelGGL(k_mlp_input_backward, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, N, L, F, row, xyz, g, d_xyz, d_feat);
    return 0;
}

}  // namespace sr
