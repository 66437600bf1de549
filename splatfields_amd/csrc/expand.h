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
// f(local_splat, k_within_splat, tile_index, local_instance, tile_x, tile_y, tiles in the splat's rect)
template <typename F>
__device__ __forceinline__ void for_each_block_instance(const uint32_t* s_off, const ushort4* s_rect, int gx, F&& f) {
    const uint32_t total = s_off[kBlock];
    for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
        int lo = 0, hi = kBlock;  // invariant: s_off[lo] <= i < s_off[hi]
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t k = i - s_off[lo];
        const ushort4 r = s_rect[lo];
        const uint32_t w = (uint32_t)(r.z - r.x);
        const uint32_t ty = k / w, tx = k - ty * w;
        f(lo, k, (uint32_t)(r.y + ty) * (uint32_t)gx + (uint32_t)r.x + tx, i, (uint32_t)r.x + tx, (uint32_t)r.y + ty, w * (uint32_t)(r.w - r.y));
    }
}
#endif

}  // namespace sr
