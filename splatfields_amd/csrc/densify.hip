// Densification / pruning of the splat set on the device.
//
// Restates reference scene/gaussian_model.py:411-425 (`densify_and_prune`) with its callees :394-409
// (`densify_and_clone`), :355-380 (`densify_and_split`), :306-353 (`cat_tensors_to_optimizer`, `densification_postfix`) and
// :272-304 (`_prune_optimizer`, `prune_points`).  The reference runs these every 100 iterations as ~60 boolean-mask
// indexing / cat / repeat kernels with several host synchronisations, and rebuilds every parameter and both Adam moments
// three times (after the clone, after the split, after the prune).  Here the three steps are planned together:
//
//   plan    one pass over the N splats decides, per splat, what the reference's sequence leaves of it --
//           itself (kept unless split or pruned), its clone, its two split children -- and prefix sums give every survivor
//           its row in the final tensors, in exactly the order the reference's cat / mask sequence produces:
//           [surviving originals][surviving clones][surviving first children][surviving second children];
//   apply   one gather per tensor writes the final rows directly (parameters, and Adam moments with zeros for new rows).
//
// One host read (the new count, to allocate the outputs) instead of the reference's several.
#include "kernels.h"

namespace sr {

namespace {

constexpr int kSeg = 4;   // row kinds: 0 = the splat itself, 1 = its clone, 2 / 3 = its split children

// what survives of splat i (bit k: a row of kind k exists in the result)
__device__ __forceinline__ uint32_t densify_decide(int i, const float* __restrict__ log_scales, int scale_cols,
                                                   const float* __restrict__ opacity_logit, const float* __restrict__ grad_accum,
                                                   const float* __restrict__ denom, const float* __restrict__ max_radii2D,
                                                   const DensifyArgs a) {
    float ms = -__builtin_inff();
    for (int c = 0; c < scale_cols; ++c) ms = fmaxf(ms, expf(log_scales[(size_t)i * scale_cols + c]));   // get_scaling.max(dim=1)
    float grad = grad_accum[i] / denom[i];
    if (grad != grad) grad = 0.0f;                          // grads[grads.isnan()] = 0
    const bool hot = grad >= a.grad_threshold;
    const bool clone = hot && ms <= a.percent_dense * a.extent;
    const bool split = hot && ms > a.percent_dense * a.extent;
    const float opac = 1.0f / (1.0f + expf(-opacity_logit[i]));
    // the final prune: opacity, screen size (on max_radii2D as densification_postfix left it -- the caller passes what the
    // reference would test, see densify.py), world size
    const bool small_op = opac < a.min_opacity;
    const bool big_vs = a.max_screen_size > 0.0f && max_radii2D && max_radii2D[i] > a.max_screen_size;
    const bool prune_self = small_op || big_vs || (a.max_screen_size > 0.0f && ms > 0.1f * a.extent);
    const bool prune_child = small_op || (a.max_screen_size > 0.0f && ms / (0.8f * 2.0f) > 0.1f * a.extent);
    uint32_t m = 0u;
    if (!split && !prune_self) m |= 1u;
    if (clone && !prune_self) m |= 2u;
    if (split && !prune_child) m |= 4u | 8u;
    return m;
}

__global__ void __launch_bounds__(kBlock) k_densify_count(int n, const float* log_scales, int scale_cols, const float* opacity_logit,
                                                          const float* grad_accum, const float* denom, const float* max_radii2D,
                                                          const DensifyArgs a, uint8_t* __restrict__ flags,
                                                          uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t s_scan[8];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t m = i < n ? densify_decide(i, log_scales, scale_cols, opacity_logit, grad_accum, denom, max_radii2D, a) : 0u;
    if (i < n) flags[i] = (uint8_t)m;
#pragma unroll
    for (int k = 0; k < kSeg; ++k) {
        uint32_t total;
        block_exclusive_scan((m >> k) & 1u, s_scan, total);
        if (threadIdx.x == 0) block_counts[(size_t)k * gridDim.x + blockIdx.x] = total;
    }
}

// exclusive prefix over the [kSeg][blocks] counts in segment-major order (= the final row order); totals[k] = rows of kind k
__global__ void __launch_bounds__(1024) k_densify_scan(int blocks, uint32_t* __restrict__ block_counts, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (int k = 0; k < kSeg; ++k) {
        const uint32_t seg_start = s_carry;
        for (int base = 0; base < blocks; base += 1024) {
            const int j = base + (int)threadIdx.x;
            const uint32_t v = j < blocks ? block_counts[(size_t)k * blocks + j] : 0u;
            s_part[threadIdx.x] = v;
            __syncthreads();
            for (int d = 1; d < 1024; d <<= 1) {   // Hillis-Steele inclusive scan
                const uint32_t add = threadIdx.x >= (unsigned)d ? s_part[threadIdx.x - d] : 0u;
                __syncthreads();
                s_part[threadIdx.x] += add;
                __syncthreads();
            }
            const uint32_t carry = s_carry;
            if (j < blocks) block_counts[(size_t)k * blocks + j] = carry + s_part[threadIdx.x] - v;
            __syncthreads();
            if (threadIdx.x == 1023) s_carry = carry + s_part[1023];
            __syncthreads();
        }
        if (threadIdx.x == 0) totals[k] = s_carry - seg_start;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[kSeg] = s_carry;
}

// dest[k][i] = final row of splat i's row of kind k, or -1
__global__ void __launch_bounds__(kBlock) k_densify_place(int n, const uint8_t* __restrict__ flags,
                                                          const uint32_t* __restrict__ block_offsets, int32_t* __restrict__ dest) {
    __shared__ uint32_t s_scan[8];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t m = i < n ? flags[i] : 0u;
#pragma unroll
    for (int k = 0; k < kSeg; ++k) {
        uint32_t total;
        const uint32_t bit = (m >> k) & 1u;
        const uint32_t excl = block_exclusive_scan(bit, s_scan, total);
        if (i < n) dest[(size_t)k * n + i] = bit ? (int32_t)(block_offsets[(size_t)k * gridDim.x + blockIdx.x] + excl) : -1;
    }
}

// One gather per tensor.  mode: 0 = every surviving row is a copy of the source row (features, opacity, rotation);
// 1 = Adam moment: the splat's own row is copied, new rows are zero; 2 = positions: children are resampled
// (xyz + R(q) (unit normal * scale)); 3 = log-scales: children get log(scale / 1.6)
__global__ void __launch_bounds__(kBlock) k_densify_gather(int n, int row, const float* __restrict__ src, float* __restrict__ dst,
                                                           const int32_t* __restrict__ dest, int mode,
                                                           const float* __restrict__ log_scales, int scale_cols,
                                                           const float* __restrict__ rotations, const float* __restrict__ unit) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;   // one float of one source row
    if (t >= (size_t)n * row) return;
    const int i = (int)(t / row), c = (int)(t % row);
    const float v = src[t];
    const int32_t d0 = dest[i], d1 = dest[(size_t)n + i], d2 = dest[(size_t)2 * n + i], d3 = dest[(size_t)3 * n + i];
    if (d0 >= 0) dst[(size_t)d0 * row + c] = v;
    if (d1 >= 0) dst[(size_t)d1 * row + c] = mode == 1 ? 0.0f : v;
    if (d2 < 0 && d3 < 0) return;
    float c2 = v, c3 = v;
    if (mode == 1) { c2 = 0.0f; c3 = 0.0f; }
    else if (mode == 3) { c2 = c3 = v - logf(0.8f * 2.0f); }   // scaling_inverse_activation(get_scaling / (0.8 N)), N = 2
    else if (mode == 2) {
        // new_xyz = build_rotation(_rotation) @ normal(0, get_scaling) + xyz   (reference utils/general_utils.py:138-159)
        const float4 q4 = reinterpret_cast<const float4*>(rotations)[i];
        const float inv = 1.0f / sqrtf(q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w);
        const float r = q4.x * inv, x = q4.y * inv, y = q4.z * inv, z = q4.w * inv;
        float R0, R1, R2;   // row c of R
        if (c == 0) { R0 = 1.f - 2.f * (y * y + z * z); R1 = 2.f * (x * y - r * z); R2 = 2.f * (x * z + r * y); }
        else if (c == 1) { R0 = 2.f * (x * y + r * z); R1 = 1.f - 2.f * (x * x + z * z); R2 = 2.f * (y * z - r * x); }
        else { R0 = 2.f * (x * z - r * y); R1 = 2.f * (y * z + r * x); R2 = 1.f - 2.f * (x * x + y * y); }
        const float s0 = expf(log_scales[(size_t)i * scale_cols]);
        const float s1 = scale_cols > 1 ? expf(log_scales[(size_t)i * scale_cols + 1]) : s0;
        const float s2 = scale_cols > 2 ? expf(log_scales[(size_t)i * scale_cols + 2]) : s0;
        const float* u2 = unit + (size_t)i * 3;                    // unit normals of the first children: unit[0][i]
        const float* u3 = unit + ((size_t)n + i) * 3;              // second children: unit[1][i]
        c2 = v + (R0 * (u2[0] * s0) + R1 * (u2[1] * s1) + R2 * (u2[2] * s2));
        c3 = v + (R0 * (u3[0] * s0) + R1 * (u3[1] * s1) + R2 * (u3[2] * s2));
    }
    if (d2 >= 0) dst[(size_t)d2 * row + c] = c2;
    if (d3 >= 0) dst[(size_t)d3 * row + c] = c3;
}

}  // namespace

size_t densify_workspace_bytes(int n) {
    const size_t nb = (size_t)((n > 0 ? n : 1) + kBlock - 1) / kBlock;
    return align_up((size_t)(n > 0 ? n : 1), 256) + align_up(sizeof(uint32_t) * kSeg * nb, 256) + 256;
}

// flags [N] | block counts / offsets [4][blocks] | totals [5]
void launch_densify_plan(int n, const float* log_scales, int scale_cols, const float* opacity_logit, const float* grad_accum,
                         const float* denom, const float* max_radii2D, const DensifyArgs& a, void* workspace, int32_t* dest,
                         uint32_t** totals_out, hipStream_t st) {
    const int blocks = (n + kBlock - 1) / kBlock;
    char* base = static_cast<char*>(workspace);
    uint8_t* flags = reinterpret_cast<uint8_t*>(base);
    uint32_t* counts = reinterpret_cast<uint32_t*>(base + align_up((size_t)n, 256));
    uint32_t* totals = reinterpret_cast<uint32_t*>(base + align_up((size_t)n, 256) + align_up(sizeof(uint32_t) * kSeg * (size_t)blocks, 256));
    *totals_out = totals;
    hipLaunchKernelGGL(k_densify_count, dim3(blocks), dim3(kBlock), 0, st, n, log_scales, scale_cols, opacity_logit, grad_accum, denom,
                       max_radii2D, a, flags, counts);
    hipLaunchKernelGGL(k_densify_scan, dim3(1), dim3(1024), 0, st, blocks, counts, totals);
    hipLaunchKernelGGL(k_densify_place, dim3(blocks), dim3(kBlock), 0, st, n, flags, counts, dest);
}

void launch_densify_gather(int n, int row, const float* src, float* dst, const int32_t* dest, int mode, const float* log_scales,
                           int scale_cols, const float* rotations, const float* unit, hipStream_t st) {
    const size_t total = (size_t)n * row;
    if (total == 0) return;
    hipLaunchKernelGGL(k_densify_gather, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, row, src, dst, dest,
                       mode, log_scales, scale_cols, rotations, unit);
}

}  // namespace sr
