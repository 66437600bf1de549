// Mean squared distance to the 3 nearest neighbours of every point -- the MI355X counterpart of
// [EXT] simple_knn._C.distCUDA2 (gitlab.inria.fr/bkerbl/simple-knn@44f7642, reference README.md:29), which the
// reference calls once at initialisation to size the splats (scene/gaussian_model.py:25, :105).
// SURVEY.md §8f row 3 ("next" row; not on the measured fwd+bwd path).
//
// Exact k-NN (k = 3, self excluded by index) on a uniform grid: points are counting-sorted into ~N/2 cells, and each
// point searches growing cubes of cells until its 3rd-best squared distance is <= (ring * cell_size)^2, which
// no point outside the cube can beat.  The published CUDA code does the same job with a Morton sort + box pruning.
#include "kernels.h"

namespace sr {

struct KnnGrid { float3 lo; float inv_cell, cell; int gx, gy, gz; };

__device__ __forceinline__ int3 cell_of(const KnnGrid& g, float x, float y, float z) {
    int cx = (int)((x - g.lo.x) * g.inv_cell), cy = (int)((y - g.lo.y) * g.inv_cell), cz = (int)((z - g.lo.z) * g.inv_cell);
    cx = min(max(cx, 0), g.gx - 1); cy = min(max(cy, 0), g.gy - 1); cz = min(max(cz, 0), g.gz - 1);
    return make_int3(cx, cy, cz);
}

// one workgroup: bounding box of all finite points -> bounds[0..5]
__global__ void __launch_bounds__(1024) k_knn_bounds(int n, const float* __restrict__ pts, float* __restrict__ bounds) {
    __shared__ float s_lo[3][16], s_hi[3][16];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = threadIdx.x; i < n; i += 1024)
        for (int c = 0; c < 3; ++c) { const float v = pts[3 * (size_t)i + c]; if (v == v) { lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); } }
    for (int c = 0; c < 3; ++c) {
        for (int d = 32; d > 0; d >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], d, 64)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], d, 64)); }
        if ((threadIdx.x & 63) == 0) { s_lo[c][threadIdx.x >> 6] = lo[c]; s_hi[c][threadIdx.x >> 6] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float l = 3.0e38f, h = -3.0e38f;
        for (int w = 0; w < 16; ++w) { l = fminf(l, s_lo[threadIdx.x][w]); h = fmaxf(h, s_hi[threadIdx.x][w]); }
        bounds[threadIdx.x] = l; bounds[3 + threadIdx.x] = h;
    }
}

__device__ __forceinline__ KnnGrid make_grid(const float* bounds, int n, int max_cells) {
    KnnGrid g;
    g.lo = make_float3(bounds[0], bounds[1], bounds[2]);
    float ex = bounds[3] - bounds[0], ey = bounds[4] - bounds[1], ez = bounds[5] - bounds[2];
    const float emax = fmaxf(ex, fmaxf(ey, ez));
    g.gx = g.gy = g.gz = 1; g.cell = 1.0f; g.inv_cell = 0.0f;  // degenerate cloud (all points equal / one point): one cell
    if (!(emax > 0.0f) || !(emax < 3.0e38f)) return g;
    // flat or linear clouds: give the thin axes a floor so that the cell size follows the populated extent
    ex = fmaxf(ex, 1e-3f * emax); ey = fmaxf(ey, 1e-3f * emax); ez = fmaxf(ez, 1e-3f * emax);
    // ~2 points per cell on average, never more cells than the workspace holds
    const float target = fminf(fmaxf(0.5f * n, 1.0f), (float)max_cells);
    float cell = fmaxf(cbrtf(ex) * cbrtf(ey) * cbrtf(ez) / cbrtf(target), emax / 1024.0f);
    bool ok = false;
    for (int it = 0; it < 64 && !ok; ++it) {
        g.gx = (int)(ex / cell) + 1; g.gy = (int)(ey / cell) + 1; g.gz = (int)(ez / cell) + 1;
        ok = (long long)g.gx * g.gy * g.gz <= (long long)max_cells;
        if (!ok) cell *= 1.3f;
    }
    if (!ok) { g.gx = g.gy = g.gz = 1; cell = 2.0f * emax; }
    g.cell = cell; g.inv_cell = 1.0f / cell;
    return g;
}

__global__ void k_knn_count(int n, const float* __restrict__ pts, const float* __restrict__ bounds, int max_cells, uint32_t* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KnnGrid g = make_grid(bounds, n, max_cells);
    const int3 c = cell_of(g, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]);
    atomicAdd(&count[(c.z * g.gy + c.y) * g.gx + c.x], 1u);
}

// exclusive scan of count[0..cells) by one workgroup -> start[0..cells]; cursor := 0
__global__ void __launch_bounds__(1024) k_knn_scan(const float* __restrict__ bounds, int n, int max_cells, const uint32_t* __restrict__ count,
                                                   uint32_t* __restrict__ start, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t s_wave[16];
    const KnnGrid g = make_grid(bounds, n, max_cells);
    const int cells = g.gx * g.gy * g.gz;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < cells; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < cells ? count[i] : 0u;
        const uint32_t inc = wave_inclusive_scan(v);
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        uint32_t wave_base = 0, all = 0;
        for (int k = 0; k < 16; ++k) { const uint32_t t = s_wave[k]; if (k < w) wave_base += t; all += t; }
        if (i < cells) { start[i] = carry + wave_base + inc - v; cursor[i] = 0u; }
        carry += all;
        __syncthreads();
    }
    if (tid == 0) start[cells] = carry;
}

__global__ void k_knn_fill(int n, const float* __restrict__ pts, const float* __restrict__ bounds, int max_cells, const uint32_t* __restrict__ start,
                           uint32_t* __restrict__ cursor, uint32_t* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KnnGrid g = make_grid(bounds, n, max_cells);
    const int3 c = cell_of(g, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]);
    const int cell = (c.z * g.gy + c.y) * g.gx + c.x;
    order[start[cell] + atomicAdd(&cursor[cell], 1u)] = (uint32_t)i;
}

__device__ __forceinline__ void knn_insert(float d, float best[3]) {
    if (d < best[2]) {
        if (d < best[1]) { best[2] = best[1]; if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d; }
        else best[2] = d;
    }
}

__global__ void k_knn_search(int n, const float* __restrict__ pts, const float* __restrict__ bounds, int max_cells, const uint32_t* __restrict__ start,
                             const uint32_t* __restrict__ order, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KnnGrid g = make_grid(bounds, n, max_cells);
    const float px = pts[3 * (size_t)i], py = pts[3 * (size_t)i + 1], pz = pts[3 * (size_t)i + 2];
    const int3 c = cell_of(g, px, py, pz);
    float best[3] = {3.0e38f, 3.0e38f, 3.0e38f};
    const int rmax = max(g.gx, max(g.gy, g.gz));
    for (int r = 0; r <= rmax; ++r) {
        // visit the shell of cells at Chebyshev distance r from the point's cell
        for (int dz = -r; dz <= r; ++dz) {
            const int z = c.z + dz; if (z < 0 || z >= g.gz) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = c.y + dy; if (y < 0 || y >= g.gy) continue;
                const bool face = (dz == -r || dz == r || dy == -r || dy == r);
                const int step = face ? 1 : max(2 * r, 1);  // interior rows of the shell only have the two x ends
                for (int dx = -r; dx <= r; dx += step) {
                    const int x = c.x + dx; if (x < 0 || x >= g.gx) continue;
                    const int cell = (z * g.gy + y) * g.gx + x;
                    for (uint32_t k = start[cell]; k < start[cell + 1]; ++k) {
                        const uint32_t j = order[k];
                        if ((int)j == i) continue;
                        const float ddx = pts[3 * (size_t)j] - px, ddy = pts[3 * (size_t)j + 1] - py, ddz = pts[3 * (size_t)j + 2] - pz;
                        knn_insert(ddx * ddx + ddy * ddy + ddz * ddz, best);
                    }
                }
            }
        }
        const float safe = r * g.cell;  // every unvisited point is farther than this
        if (best[2] <= safe * safe) break;
    }
    // fewer than 4 points: missing neighbours contribute 0, as the published code's zero-initialised best[]... it starts
    // from FLT_MAX; keep the mean over the neighbours that exist
    float sum = 0.f; int cnt = 0;
    for (int k = 0; k < 3; ++k) if (best[k] < 3.0e38f) { sum += best[k]; ++cnt; }
    out[i] = cnt == 3 ? sum / 3.0f : (cnt > 0 ? sum / 3.0f : 0.0f);
}

size_t knn_workspace_bytes(int n) {
    const size_t cells = (size_t)(n > 2 ? n : 2);  // max_cells = n
    return align_up(8 * sizeof(float), 256) + 3 * align_up((cells + 1) * sizeof(uint32_t), 256) + align_up((size_t)(n > 0 ? n : 1) * sizeof(uint32_t), 256);
}

void launch_knn3(int n, const float* pts, float* out, void* workspace, hipStream_t st) {
    if (n <= 0) return;
    Carver c{static_cast<char*>(workspace), 0};
    const int max_cells = n > 2 ? n : 2;
    float* bounds = c.take<float>(8);
    uint32_t* count = c.take<uint32_t>((size_t)max_cells + 1);
    uint32_t* start = c.take<uint32_t>((size_t)max_cells + 1);
    uint32_t* cursor = c.take<uint32_t>((size_t)max_cells + 1);
    uint32_t* order = c.take<uint32_t>((size_t)n);
    const int nb = (n + 255) / 256;
    hipMemsetAsync(count, 0, sizeof(uint32_t) * ((size_t)max_cells + 1), st);
    hipLaunchKernelGGL(k_knn_bounds, dim3(1), dim3(1024), 0, st, n, pts, bounds);
    hipLaunchKernelGGL(k_knn_count, dim3(nb), dim3(256), 0, st, n, pts, bounds, max_cells, count);
    hipLaunchKernelGGL(k_knn_scan, dim3(1), dim3(1024), 0, st, bounds, n, max_cells, count, start, cursor);
    hipLaunchKernelGGL(k_knn_fill, dim3(nb), dim3(256), 0, st, n, pts, bounds, max_cells, start, cursor, order);
    hipLaunchKernelGGL(k_knn_search, dim3(nb), dim3(256), 0, st, n, pts, bounds, max_cells, start, order, out);
}

}  // namespace sr
