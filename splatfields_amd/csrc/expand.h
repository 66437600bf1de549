// Block-cooperative expansion of (splat -> tile instances): the 256 splats of a workgroup publish
// their exclusive instance offsets and tile rectangles in LDS, then all 256 threads walk the
// concatenated instance list with a stride of 256, locating the owning splat by binary search.
// A splat covering hundreds of tiles therefore costs every lane the same, instead of serialising
// one lane of a wavefront (the per-thread rect loop of the published CUDA duplicateWithKeys).
#pragma once
#include "common.h"

namespace sr {

#ifdef __HIPCC__
// s_off: [kBlock+1] exclusive offsets, s_off[kBlock] = block total.  s_rect: [kBlock] tile rects.
// slice / slices: only the slice-th of `slices` equal parts of the instance list is walked.
// s_rect holds the rectangles as Geom::rect stores them (top nibbles = reach mask of a small rectangle, common.h).
// f(local_splat, k_within_splat, tile_index, local_instance, tile_x, tile_y, the rectangle's 16 mask bits)
template <typename F>
__device__ __forceinline__ void for_each_block_instance(const uint32_t* s_off, const ushort4* s_rect, int gx, F&& f, uint32_t slice = 0, uint32_t slices = 1) {
    const uint32_t all = s_off[kBlock];
    // this workgroup's share of the sub-batch's instances (Chunking::slices): [first, total)
    const uint32_t per = (all + slices - 1u) / slices, first = min(all, slice * per), total = min(all, first + per);
    // SR_EXPAND_ILP instances per thread and round: their binary searches (8 dependent LDS reads each) are independent chains
    // the compiler interleaves.  A workgroup walks its instances alone (N / 256 workgroups: 1.5 wavefronts per SIMD at 100 k
    // splats), so nothing else hides that latency: k_count_tiles ran at 24 % of the VALU issue rate in the dense regimes.
#ifndef SR_EXPAND_ILP
#define SR_EXPAND_ILP 4
#endif
    for (uint32_t i0 = first + threadIdx.x; i0 < total; i0 += SR_EXPAND_ILP * kBlock) {
        int lo[SR_EXPAND_ILP], hi[SR_EXPAND_ILP];  // invariant: s_off[lo] <= i < s_off[hi]
#pragma unroll
        for (int u = 0; u < SR_EXPAND_ILP; ++u) { lo[u] = 0; hi[u] = kBlock; }
#pragma unroll
        for (int step = 0; step < 8; ++step) {
#pragma unroll
            for (int u = 0; u < SR_EXPAND_ILP; ++u) {
                const int mid = (lo[u] + hi[u]) >> 1;
                if (s_off[mid] <= min(i0 + (uint32_t)u * kBlock, total - 1u)) lo[u] = mid; else hi[u] = mid;
            }
        }
#pragma unroll
        for (int u = 0; u < SR_EXPAND_ILP; ++u) {
            const uint32_t i = i0 + (uint32_t)u * kBlock;
            if (i >= total) break;
            const uint32_t k = i - s_off[lo[u]];
            const ushort4 packed = s_rect[lo[u]];
            const ushort4 r = rect_clean(packed);
            const uint32_t w = (uint32_t)(r.z - r.x);
            // k / w without the 25-instruction integer division: k < 2^22 (a rectangle has at most gx * gy tiles), so the
            // float quotient of k + 0.5 is off by less than the 0.5 / w that separates it from the next integer; corrected anyway
            uint32_t ty = (uint32_t)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)w));
            if (ty * w > k) --ty; else if ((ty + 1u) * w <= k) ++ty;
            const uint32_t tx = k - ty * w;
            f(lo[u], k, (uint32_t)(r.y + ty) * (uint32_t)gx + (uint32_t)r.x + tx, i, (uint32_t)r.x + tx, (uint32_t)r.y + ty, rect_mask16(packed));
        }
    }
}
#endif

}  // namespace sr
