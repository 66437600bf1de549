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
// Where the atomics run: the (point, plane) pairs are first BINNED by tile of the plane (16 x 16 texels up to 32 channels; a
// pair whose 2 x 2 footprint straddles a border goes to every tile it touches: one counting pass, a scan, one scattering pass
// with a cursor per tile -- ~0.34 M global atomics at 100 k points), then one workgroup per tile accumulates its pairs into a
// tile of 64-bit accumulators in LDS (ds_add_u64) and writes the tile's gradient out.  The 19 M atomics of a step stay inside
// the CUs (they were 0.43 ms of device-scope atomics on a 39 MB accumulator, plus its clearing and conversion passes), and
// because the sums are integer the arbitrary order of a tile's list never reaches the result.
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

// max |g| as float bits (non-negative floats order like unsigned integers): integer atomicMax, order-independent.
// One atomic per workgroup (one per wavefront on a single word cost 80 us for 8 k wavefronts).
__global__ void __launch_bounds__(kBlock) k_tp_absmax(long long count, const float* __restrict__ g, uint32_t* __restrict__ out_bits) {
    __shared__ float s_m[kBlock / 64];
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock) {
        const float a = fabsf(g[i]);
        m = a > m ? a : m;     // a NaN never wins: its gradient contribution is dropped below as well
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if (lane_id() == 0) s_m[wave_id()] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_m[w]);
        if (m > 0.0f) atomicMax(out_bits, __float_as_uint(m));
    }
}

// ---- binning of the (point, plane) pairs by plane tile -------------------------------------------------------------
// Every tile has kTpCopies counters, one per residue of the counting workgroup's index: 0.34 M atomics on 1200 words were 66 us
// per pass (same-address serialisation in the L2), spread over 38 k words they are not.  A tile's list is the concatenation of
// its copies' pieces (the scan runs over [tile][copy]).
constexpr int kTpCopies = 32;
struct TpBins {
    int tile, tiles_x, tiles_y;   // tile edge in texels; tiles per plane = tiles_x * tiles_y
    uint32_t* count;              // [3 tiles][kTpCopies] pairs per tile and copy, then (k_tp_bin_scan) the write cursors of the scattering pass
    uint32_t* start;              // [3 tiles * kTpCopies + 1]
    uint32_t* list;               // [<= 12 N] point indices, tile after tile
    uint32_t* gmax;               // max |g| as float bits
};

// the (up to 4) distinct tiles a pair's valid corners lie in: f(tile index inside the plane)
template <typename F>
__device__ __forceinline__ void for_each_tile_of(const Corners& c, int W, const TpBins& b, F&& f) {
    int seen[4];
    int ns = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c.idx[k] < 0) continue;
        const int y = c.idx[k] / W, x = c.idx[k] - y * W;
        const int t = (y / b.tile) * b.tiles_x + x / b.tile;
        bool dup = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) dup = dup || (j < ns && seen[j] == t);
        if (!dup) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j == ns) seen[j] = t;   // no dynamic indexing (scratch)
            ++ns;
            f(t);
        }
    }
}

template <bool EMIT>
__global__ void __launch_bounds__(kBlock) k_tp_bin(int N, int H, int W, const float* __restrict__ pts, const TpBins b) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)N * 3) return;
    const int p = (int)(t % 3), n = (int)(t / 3);
    const Corners c = corners_of(pts[3 * (size_t)n + kAxisX[p]], pts[3 * (size_t)n + kAxisY[p]], H, W);
    const int nt = b.tiles_x * b.tiles_y;
    const uint32_t copy = blockIdx.x & (kTpCopies - 1);
    for_each_tile_of(c, W, b, [&](int tile) {
        uint32_t* word = b.count + (size_t)(p * nt + tile) * kTpCopies + copy;
        if (EMIT) b.list[atomicAdd(word, 1u)] = (uint32_t)n;   // count[] holds the cursors by now
        else atomicAdd(word, 1u);
    });
}

// exclusive prefix of the 3 tiles x kTpCopies counts (one workgroup); count[] becomes the cursor array of the scattering pass
__global__ void __launch_bounds__(1024) k_tp_bin_scan(int bins, const TpBins b) {   // bins: a multiple of 4 (kTpCopies)
    __shared__ uint32_t s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < bins; base += 4096) {   // four consecutive counters per thread
        const int i = base + 4 * tid;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < bins) v = *reinterpret_cast<const uint4*>(b.count + i);
        const uint32_t sum = v.x + v.y + v.z + v.w;
        const uint32_t inc = wave_inclusive_scan(sum);
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        uint32_t wave_base = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t x = s_wave[k]; if (k < w) wave_base += x; all += x; }
        if (i < bins) {
            const uint32_t e0 = carry + wave_base + inc - sum;
            const uint4 ex = make_uint4(e0, e0 + v.x, e0 + v.x + v.y, e0 + v.x + v.y + v.z);
            *reinterpret_cast<uint4*>(b.start + i) = ex;
            *reinterpret_cast<uint4*>(b.count + i) = ex;
        }
        carry += all;
        __syncthreads();
    }
    if (tid == 0) b.start[bins] = carry;
}

// One workgroup per (tile, plane): the tile's pairs -> TILE x TILE x C 64-bit accumulators in LDS -> dL/dplanes [3][C][HW].
// Every texel of the plane is written (zeros where nothing landed): no clearing pass, no conversion pass.
template <int TILE>
__global__ void __launch_bounds__(kBlock) k_tp_accumulate(int N, int C, int H, int W, const float* __restrict__ pts,
                                                          const float* __restrict__ g, const TpBins b, float* __restrict__ d_chw) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_acc[];   // [TILE * TILE][C]
    const int p = blockIdx.y, tile = blockIdx.x, nt = b.tiles_x * b.tiles_y;
    const int tx0 = (tile % b.tiles_x) * TILE, ty0 = (tile / b.tiles_x) * TILE;
    const int cq = C >> 2;
    for (int i = threadIdx.x; i < TILE * TILE * C; i += kBlock) s_acc[i] = 0ull;
    const float gmax = __uint_as_float(*b.gmax);
    int eg = 0, en = 0;
    frexpf(gmax, &eg);                         // gmax < 2^eg
    frexpf(4.0f * (float)N + 1.0f, &en);       // contributions per texel <= 4 N < 2^en
    const double scale = ldexp(1.0, 62 - eg - en);
    const uint32_t first = b.start[(size_t)(p * nt + tile) * kTpCopies], cnt = b.start[(size_t)(p * nt + tile + 1) * kTpCopies] - first;
    __syncthreads();
    if (gmax > 0.0f) {
        for (uint32_t i = threadIdx.x; i < cnt * (uint32_t)cq; i += kBlock) {
            const uint32_t e = i / (uint32_t)cq, q = i - e * (uint32_t)cq;
            const uint32_t n = b.list[first + e];
            const Corners c = corners_of(pts[3 * (size_t)n + kAxisX[p]], pts[3 * (size_t)n + kAxisY[p]], H, W);
            const float4 gv = reinterpret_cast<const float4*>(g)[((size_t)n * 3 + p) * cq + q];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (c.idx[k] < 0) continue;
                const int y = c.idx[k] / W, x = c.idx[k] - y * W;
                const int lx = x - tx0, ly = y - ty0;
                if (lx < 0 || lx >= TILE || ly < 0 || ly >= TILE) continue;   // this corner belongs to a neighbouring tile
                unsigned long long* dst = s_acc + (size_t)(ly * TILE + lx) * C + 4 * q;
                const float v[4] = {gv.x * c.w[k], gv.y * c.w[k], gv.z * c.w[k], gv.w * c.w[k]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long fx = __double2ll_rn((double)v[j] * scale);   // NaN -> 0
                    if (fx != 0) atomicAdd(dst + j, (unsigned long long)fx);      // LDS; two's complement: signed sums wrap correctly
                }
            }
        }
    }
    __syncthreads();
    const double inv = gmax > 0.0f ? ldexp(1.0, -(62 - eg - en)) : 0.0;
    const size_t HW = (size_t)H * W;
    for (int i = threadIdx.x; i < TILE * TILE * C; i += kBlock) {
        const int ch = i / (TILE * TILE), local = i - ch * (TILE * TILE);
        const int x = tx0 + local % TILE, y = ty0 + local / TILE;
        if (x < W && y < H)
            d_chw[((size_t)p * C + ch) * HW + (size_t)y * W + x] = (float)((double)(long long)s_acc[(size_t)local * C + ch] * inv);
    }
}

// dL/dpoints: a gather over the point's 12 corners.  Sixteen lanes (one DPP row) per point: lane = (plane, channel quad), the
// fourth plane slot idle; the partial derivatives are summed over the row in a fixed tree.  (One thread per point left
// the kernel with N / 64 wavefronts walking 12 x C / 4 dependent gathers each: 49 us at 100 k points.)
__global__ void __launch_bounds__(kBlock) k_tp_backward_points(int N, int C, int H, int W, const float* __restrict__ hwc,
                                                               const float* __restrict__ pts, const float* __restrict__ g,
                                                               float* __restrict__ d_pts) {
    const int n = blockIdx.x * (kBlock / 16) + ((int)threadIdx.x >> 4);
    const int s = (int)threadIdx.x & 15, p = s >> 2, qi = s & 3;
    const int cq = C >> 2;
    float d[3] = {0.f, 0.f, 0.f};
    if (n < N && p < 3) {
        const Corners c = corners_of(pts[3 * (size_t)n + kAxisX[p]], pts[3 * (size_t)n + kAxisY[p]], H, W);
        const float4* plane = reinterpret_cast<const float4*>(hwc + (size_t)p * H * W * C);
        const float4* gv = reinterpret_cast<const float4*>(g + ((size_t)n * 3 + p) * C);
        // d out / d ix = (ne - nw)(y1 - iy) + (se - sw)(iy - y0);  d out / d iy = (sw - nw)(x1 - ix) + (se - ne)(ix - x0)
        const float fx1 = c.ix - c.x0, fx0 = 1.0f - fx1, fy1 = c.iy - c.y0, fy0 = 1.0f - fy1;
        float gx = 0.f, gy = 0.f;
        for (int q = qi; q < cq; q += 4) {
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
        gx *= 0.5f * (float)W;   // d ix / d x = W / 2
        gy *= 0.5f * (float)H;
        // plane p contributes to axes kAxisX[p] = p and kAxisY[p] = (p + 1) % 3
        d[0] = p == 0 ? gx : (p == 2 ? gy : 0.f);
        d[1] = p == 1 ? gx : (p == 0 ? gy : 0.f);
        d[2] = p == 2 ? gx : (p == 1 ? gy : 0.f);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // sum over the row of 16 lanes (every lane ends up with it)
        float t = d[a];
        t = dpp_add<0xB1>(t); t = dpp_add<0x4E>(t); t = dpp_add<0x124>(t); t = dpp_add<0x128>(t);
        d[a] = t;
    }
    if (n < N && s == 0) { d_pts[3 * (size_t)n] = d[0]; d_pts[3 * (size_t)n + 1] = d[1]; d_pts[3 * (size_t)n + 2] = d[2]; }
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

// tile edge of the binning: TILE^2 x C 64-bit accumulators must fit 64 KiB of LDS
static int tp_tile(int C) { return C <= 32 ? 16 : (C <= 128 ? 8 : 0); }

static size_t tp_carve(void* base, int N, int C, int H, int W, TpBins* b) {
    const int tile = tp_tile(C);
    const int tiles_x = (W + tile - 1) / tile, tiles_y = (H + tile - 1) / tile;
    const size_t bins = (size_t)3 * tiles_x * tiles_y * kTpCopies;
    size_t off = 0;
    auto take = [&](size_t words) { const size_t at = off; off += (words * 4 + 15) / 16 * 16; return at; };
    const size_t o_gmax = take(4), o_count = take(bins), o_start = take(bins + 1), o_list = take((size_t)12 * (N > 0 ? N : 1));
    if (b) {
        char* c = static_cast<char*>(base);
        b->tile = tile; b->tiles_x = tiles_x; b->tiles_y = tiles_y;
        b->gmax = reinterpret_cast<uint32_t*>(c + o_gmax); b->count = reinterpret_cast<uint32_t*>(c + o_count);
        b->start = reinterpret_cast<uint32_t*>(c + o_start); b->list = reinterpret_cast<uint32_t*>(c + o_list);
    }
    return off;
}

size_t triplane_backward_workspace(int N, int C, int H, int W) {
    if (C <= 0 || (C & 3) || tp_tile(C) == 0 || H <= 0 || W <= 0 || N < 0) return 0;
    return tp_carve(nullptr, N, C, H, W, nullptr);
}

// `workspace`: triplane_backward_workspace(N, C, H, W) bytes (tile counters, tile starts, the binned point lists, max |g|)
int launch_triplane_backward(int N, int C, int H, int W, const float* planes_hwc, const float* pts, const float* g, float* d_planes_chw,
                             float* d_pts, void* workspace, hipStream_t st) {
    if (C <= 0 || (C & 3) || tp_tile(C) == 0 || H <= 0 || W <= 0 || (size_t)H * W > (1u << 30) || (long long)N * 12 > 0xffffffffll) return 1;
    if (d_planes_chw) {
        TpBins b;
        tp_carve(workspace, N, C, H, W, &b);
        const int nt = b.tiles_x * b.tiles_y, bins = 3 * nt * kTpCopies;
        // max |g| and the tile counters are contiguous at the front of the workspace
        if (hipMemsetAsync(workspace, 0, (size_t)(reinterpret_cast<char*>(b.start) - static_cast<char*>(workspace)), st) != hipSuccess) return 2;
        if (N > 0) {
            const long long count = (long long)N * 3 * C;
            const int blocks = (int)((count + kBlock * 16 - 1) / (kBlock * 16) < 1024 ? (count + kBlock * 16 - 1) / (kBlock * 16) : 1024);
            hipLaunchKernelGGL(k_tp_absmax, dim3(blocks > 0 ? blocks : 1), dim3(kBlock), 0, st, count, g, b.gmax);
            const unsigned pair_blocks = (unsigned)(((long long)N * 3 + kBlock - 1) / kBlock);
            hipLaunchKernelGGL(k_tp_bin<false>, dim3(pair_blocks), dim3(kBlock), 0, st, N, H, W, pts, b);
            hipLaunchKernelGGL(k_tp_bin_scan, dim3(1), dim3(1024), 0, st, bins, b);
            hipLaunchKernelGGL(k_tp_bin<true>, dim3(pair_blocks), dim3(kBlock), 0, st, N, H, W, pts, b);
        } else {
            hipLaunchKernelGGL(k_tp_bin_scan, dim3(1), dim3(1024), 0, st, bins, b);   // all starts 0
        }
        const size_t lds = (size_t)b.tile * b.tile * C * sizeof(unsigned long long);
        if (b.tile == 16) hipLaunchKernelGGL(k_tp_accumulate<16>, dim3(nt, 3), dim3(kBlock), lds, st, N, C, H, W, pts, g, b, d_planes_chw);
        else hipLaunchKernelGGL(k_tp_accumulate<8>, dim3(nt, 3), dim3(kBlock), lds, st, N, C, H, W, pts, g, b, d_planes_chw);
    }
    if (d_pts && N > 0)
        hipLaunchKernelGGL(k_tp_backward_points, dim3((N + kBlock / 16 - 1) / (kBlock / 16)), dim3(kBlock), 0, st, N, C, H, W, planes_hwc, pts, g, d_pts);
    return 0;
}

}  // namespace sr
