// Internal definitions shared by the gfx950 kernels of libsplatraster.so.
// Semantics restated from SURVEY.md Appendix A (published 3DGS tile rasterizer + depth output).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/splatraster.h"

namespace sr {

// ---- constants of the algorithm (one place; the oracle mirrors them) ----------------------
constexpr int kTile = SR_TILE;          // binning tile edge
constexpr int kSub = 8;                 // one wavefront renders an 8x8 sub-tile
constexpr float kNearCullZ = 0.2f;
constexpr float kDilation = 0.3f;
constexpr float kClampFov = 1.3f;
constexpr float kAlphaMax = 0.99f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTStop = 0.0001f;
constexpr float kWEps = 0.0000001f;

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

constexpr int kBlock = 256;             // threads per workgroup everywhere (4 wavefronts of 64)
constexpr int kWave = 64;

// per-splat flag bits written by the forward preprocess
constexpr uint8_t kFlagClampR = 1, kFlagClampG = 2, kFlagClampB = 4, kFlagClampTx = 8, kFlagClampTy = 16;

// ---- carved views of the caller-owned opaque buffers ---------------------------------------
struct Geom {
    // [N][4] one 64-byte, 64-byte-aligned record per splat, so a gather touches exactly one cache line:
    //   q0 = (pix.x, pix.y, tau = 2 ln(255 o) * log2(e) [support: d^T Q' d <= tau; < 0: never visible], view depth)
    //   q1 = (p, s, q, -log2 o): completed-square factors of Q' = conic * log2(e), see pair_alpha_unclamped()      q2 = (r, g, b, view depth)
    //   q3 = (bits: tile rect xmin | ymin << 16, bits: rect width, unused, unused)  -- instance index of (splat, tile)
    float4* rec;
    ushort4* rect;       // [N] tile rect (xmin, ymin, xmax, ymax), max exclusive
    uint32_t* touched;   // [N] instances emitted by this splat
    uint32_t* offsets;   // [N] exclusive prefix of `touched`
    float* dcol_ddir;    // (SH path, degree > 0) J[3c + axis] = d colour_c / d unit-direction_axis before the >= 0 clamp, stored as
                         //     float4[N] (J0..J3), float4[N] (J4..J7), float[N] (J8):
                         //     written by the forward so that the backward never re-reads the 192-byte SH block of a splat
    uint32_t* depth_bits; // [N] view depth as float bits (> 0.2, so they sort as integers): the scatter reads 4 B instead of a 64-B record
    uint8_t* flags;      // [N]
    uint32_t* block_sums;    // [ceil(N/256)]
    uint32_t* block_offsets; // [ceil(N/256)]
    uint32_t* tile_count;    // [tiles]
    uint32_t* tile_start;    // [tiles+1]
    uint32_t* tile_cursor;   // [tiles]
    uint32_t* tile_order;    // [tiles] tiles by decreasing list length (coarse classes): launch order of the blend kernels
    uint32_t* total;         // [0] number of instances, [1] longest tile list, [4..6] slots of tile_order that can hold lists > 2048 / 4096 / 8192
    // written by the forward blend, read by the backward kernels:
    uint32_t* tile_qlast;    // [tiles][16] per 4x4-pixel quad (row-major in the tile): list entries in front of the stop of its
                             //     last pixel (max over the quad's pixels of n_contrib); the backward replays [0, max over quads)
    // atomic-free bucketing (images up to kMaxMatrixTiles tiles): per-chunk x per-tile instance counts
    uint32_t* cnt;           // [chunks][tiles_padded] counts, then exclusive prefix over the chunks of a segment
    uint32_t* segtot;        // [tiles_padded] column totals of the count matrix (k_colscan); the carve keeps the [segments][tiles_padded]
    uint32_t* segbase;       // extent of rounds 2-4 for both tables (workspace sizes are part of the callers' contract); segbase is unused
};

constexpr int kMaxChunks = 1024;        // chunk = consecutive 256-splat sub-batches handled by one workgroup
constexpr int kSegRows = 128;   // chunks per column-scan segment
constexpr int kMaxMatrixTiles = 16384;  // LDS histogram of 64 KiB; larger images use the global-atomic fallback

// A chunk = `sub_per_chunk` consecutive sub-batches, or -- when there are fewer than kMaxChunks / 2 sub-batches -- one of
// `slices` equal parts of the instance range of ONE sub-batch (chunk = sub-batch * slices + slice).  Small scenes with large
// footprints (100 k splats x 70 tiles) otherwise leave the two instance passes with N / 256 workgroups, each walking 17 k
// instances alone: 1.5 wavefronts per SIMD, 24 % of the VALU issue rate.
struct Chunking { int n_sub, chunks, sub_per_chunk, slices, segments, tiles_padded; };
inline Chunking make_chunking(int N, int tiles) {
    Chunking c;
    c.n_sub = (N + 255) / 256; if (c.n_sub < 1) c.n_sub = 1;
    const int b = c.n_sub < kMaxChunks ? c.n_sub : kMaxChunks;
    c.sub_per_chunk = (c.n_sub + b - 1) / b;
    c.chunks = (c.n_sub + c.sub_per_chunk - 1) / c.sub_per_chunk;
    c.slices = 1;
    if (c.sub_per_chunk == 1) { c.slices = kMaxChunks / c.chunks; if (c.slices > 8) c.slices = 8; if (c.slices < 1) c.slices = 1; }
    c.chunks *= c.slices;
    c.segments = (c.chunks + kSegRows - 1) / kSegRows;
    c.tiles_padded = (tiles + 63) / 64 * 64;
    return c;
}

struct Binning {
    uint64_t* ent;       // [R] bucketed per tile, unsorted: sort key = (view-depth bits << 32) | splat index
                         //     -- ONE 8-byte store per instance from the scatter; the payload is the key's low half
    uint64_t* keys;      // [R] ping-pong buffer A of the long-list merge path
    uint64_t* keys_tmp;  // [R] ping-pong buffer B
    uint32_t* sorted_id; // [R] splat index, per tile front-to-back.  The instance index a (splat, tile) pair owns in the backward scratch
                         //     is not stored: offsets[splat] + row-major position of the tile inside the splat's tile rect (record
                         //     quarter 3).  (Precomputing it per list entry in the per-tile sort was measured: +45 us there for the
                         //     two gathers per entry at the tail of a VALU-bound kernel, nothing gained in the backward.)
    uint32_t* qmask;     // [R] per list entry (same order as sorted_id): which 4x4-pixel quads of its tile the splat can reach
                         //     with alpha >= 1/255 (bit 4 qy + qx, quadmask.h; 32-bit words: the backward fetches them with
                         //     LDS-direct loads, whose granule is a dword).  Computed once by the forward blend while it stages
                         //     the entry, used there for the sub-tile culling and by the backward blend for its bucketing
    uint8_t* reached;    // [R] per tile-splat INSTANCE (index = offsets[splat] + position of the tile in the splat's rect): 1 once
                         //     the backward blend has written its gradient slot.  Cleared by the scatter (k_emit walks the
                         //     instances in exactly this order: coalesced byte stores, no separate clearing pass), set by the
                         //     backward blend for the list entries in front of the stop of the tile's last pixel, tested by
                         //     k_preprocess_backward: unreached slots are neither written nor read
    uint32_t capacity;   // instances the arrays above were carved for; stage-2 kernels exit if total > capacity
    uint32_t sorted_up_to;   // longest tile list the launched sort classes cover (0xffffffff: every class the lists need was
                             // launched -- the host knew the longest list).  sr_forward_async launches the classes a HINT asks for
                             // without waiting for the real figure: k_render_forward exits if a longer list exists (its ids were
                             // never written), and the caller finds out when it redeems the ticket
};

struct Image {
    float* final_T;      // [H*W]
    uint32_t* n_contrib; // [H*W]
};

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Carver {
    char* base; size_t off;
    template <typename T> __host__ T* take(size_t n) {
        off = align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

inline int tiles_x(int W) { return (W + kTile - 1) / kTile; }
inline int tiles_y(int H) { return (H + kTile - 1) / kTile; }

inline size_t carve_geom(void* base, int N, int H, int W, Geom* g) {
    Carver c{static_cast<char*>(base), 0};
    const size_t n = (size_t)(N > 0 ? N : 1);
    const size_t nb = (n + kBlock - 1) / kBlock;
    const size_t tiles = (size_t)tiles_x(W) * tiles_y(H);
    Geom t;
    t.rec = c.take<float4>(4 * n);
    t.rect = c.take<ushort4>(n);
    t.touched = c.take<uint32_t>(n); t.offsets = c.take<uint32_t>(n); t.depth_bits = c.take<uint32_t>(n);
    t.dcol_ddir = c.take<float>(9 * n);
    t.flags = c.take<uint8_t>(n);
    t.block_sums = c.take<uint32_t>(nb); t.block_offsets = c.take<uint32_t>(nb);
    t.tile_count = c.take<uint32_t>(tiles); t.tile_start = c.take<uint32_t>(tiles + 1);
    t.tile_cursor = c.take<uint32_t>(tiles); t.tile_order = c.take<uint32_t>(tiles); t.total = c.take<uint32_t>(8);
    t.tile_qlast = c.take<uint32_t>(16 * tiles);
    t.cnt = t.segtot = t.segbase = nullptr;
    if (tiles <= (size_t)kMaxMatrixTiles) {
        const Chunking ch = make_chunking(N, (int)tiles);
        t.cnt = c.take<uint32_t>((size_t)ch.chunks * ch.tiles_padded);
        t.segtot = c.take<uint32_t>((size_t)ch.segments * ch.tiles_padded);
        t.segbase = c.take<uint32_t>((size_t)ch.segments * ch.tiles_padded);
    }
    if (g) *g = t;
    return align_up(c.off, 256);
}

inline size_t carve_binning(void* base, long long R, Binning* b) {
    Carver c{static_cast<char*>(base), 0};
    const size_t r = (size_t)(R > 0 ? R : 1);
    Binning t;
    t.ent = c.take<uint64_t>(r);
    t.keys = c.take<uint64_t>(r);
    t.keys_tmp = c.take<uint64_t>(r);
    t.sorted_id = c.take<uint32_t>(r);
    t.qmask = c.take<uint32_t>(r);
    t.reached = c.take<uint8_t>(r);
    t.capacity = (uint32_t)(R > 0 ? R : 0);
    t.sorted_up_to = 0xffffffffu;
    if (b) *b = t;
    return align_up(c.off, 256);
}

inline size_t carve_image(void* base, int H, int W, Image* im) {
    Carver c{static_cast<char*>(base), 0};
    const size_t px = (size_t)H * W;
    Image t;
    t.final_T = c.take<float>(px); t.n_contrib = c.take<uint32_t>(px);
    if (im) *im = t;
    return align_up(c.off, 256);
}

// One record of the backward scratch: per tile-splat instance, the tile-reduced moment sums.
// (S0, Sx, Sy, Sxx | Sxy, Syy, dr, dg | db, ddepth, pad, pad)
// Only the instances the forward reached (list position in front of the stop of the tile's last pixel) are written by the
// backward blend and read by k_preprocess_backward; Binning::reached says which.
// Slot stride = kSlotF4 quarters of 16 bytes.  Memory is written and fetched in 64-byte sectors, and a 48-byte stride puts half
// of the slots across two of them (backward blend: WRITE_SIZE 168 MB for 82 MB of slots).  A 64-byte stride was measured
// (round 4): the backward blend does not care (0.2556 vs 0.2558 ms), and k_preprocess_backward gets SLOWER
// (0.0978 vs 0.0945 ms) -- it reads a splat's consecutive instances, which the 48-byte stride packs into fewer sectors.
constexpr int kSlotF4 = 3;
constexpr int kSlotFloats = 4 * kSlotF4;


// ---- per-view constants, passed to kernels by value ----------------------------------------
struct ViewK {
    int H, W, gx, gy;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    int sh_degree, sh_coeffs;
    const float* viewmatrix; const float* projmatrix; const float* campos; const float* bg;
};

// ---- device helpers -------------------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Streaming (read-once) 16-byte load: marked non-temporal so that it does not displace the lines the kernel is still
// assembling in the L2 (k_preprocess: 0.089 -> 0.077 ms from this alone on its 192 MB SH stream; plain stores must stay
// temporal, they rely on the L2 to combine 16-byte pieces into whole lines -- non-temporal stores doubled the kernel time).
__device__ __forceinline__ float4 load_stream(const float4* p) {
    typedef float v4f_nt __attribute__((ext_vector_type(4)));
    const v4f_nt x = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(p));
    return make_float4(x.x, x.y, x.z, x.w);
}

// Streaming 16-byte store for data written in WHOLE lines by consecutive lanes (nothing left for the L2 to combine) and
// not read again by this kernel: keeps a 192 MB gradient stream from flushing the partially written lines of the other
// outputs out of the L2 (k_preprocess_backward: -8 us).
__device__ __forceinline__ void store_stream(float4* p, const float4 v) {
    typedef float v4f_nt __attribute__((ext_vector_type(4)));
    const v4f_nt x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<v4f_nt*>(p));
}

// Asynchronous 16-byte-per-lane copy memory -> LDS (global_load_lds_dwordx4): lane l's 16 bytes at `src` land at LDS byte
// address `lds_base` + 16 l, with no register staging and no ds_write.  Written as inline assembly on purpose: hipcc drains
// every outstanding memory operation at the next barrier / first use when it tracks such a copy itself, whereas these
// requests are meant to stay in flight across a whole batch of blending; the kernel waits for them with lds_copy_wait().
// M0 (the LDS base) is reserved by the compiler: saved and restored inside the statement.
__device__ __forceinline__ void lds_copy16_async(const void* src, uint32_t lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_base) : "memory");
}
// the 4-byte form: lane l's dword lands at `lds_base` + 4 l
__device__ __forceinline__ void lds_copy4_async(const void* src, uint32_t lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void lds_copy_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Workgroup barrier that orders LDS traffic only (this wavefront's LDS operations are complete before it arrives): unlike
// __syncthreads() it carries no release fence, so requests to global memory stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)p; }   // low half of the flat address

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
    // lanes whose row is masked off (or whose source is invalid) add `old` = 0
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}

// Sum over the 64 lanes of a wavefront; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_add<0x124>(v);       // row_ror:4
    v = dpp_add<0x128>(v);       // row_ror:8   -> every lane holds its row-of-16 sum
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// Inclusive prefix sum across a wavefront (unsigned).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// Exclusive prefix sum over the 256 threads of a workgroup; `total` is returned to every thread.
// `scratch` must hold >= 8 uint32_t of LDS.  Contains two __syncthreads().
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch, uint32_t& total) {
    const uint32_t inc = wave_inclusive_scan(v);
    const int w = wave_id();
    __syncthreads();  // protect scratch reuse across calls
    if (lane_id() == 63) scratch[w] = inc;
    __syncthreads();
    const uint32_t s0 = scratch[0], s1 = scratch[1], s2 = scratch[2], s3 = scratch[3];
    total = s0 + s1 + s2 + s3;
    const uint32_t wave_base = (w > 0 ? s0 : 0) + (w > 1 ? s1 : 0) + (w > 2 ? s2 : 0);
    return wave_base + inc - v;
}


// Two prefixes in one pass (same two barriers): returns the exclusive prefix of a, b's through b_excl.  scratch: [8].
__device__ __forceinline__ uint32_t block_exclusive_scan2(uint32_t a, uint32_t b, uint32_t* scratch, uint32_t& a_total, uint32_t& b_excl, uint32_t& b_total) {
    const uint32_t ia = wave_inclusive_scan(a), ib = wave_inclusive_scan(b);
    const int w = wave_id();
    __syncthreads();  // protect scratch reuse across calls
    if (lane_id() == 63) { scratch[w] = ia; scratch[4 + w] = ib; }
    __syncthreads();
    const uint32_t s0 = scratch[0], s1 = scratch[1], s2 = scratch[2], s3 = scratch[3];
    const uint32_t t0 = scratch[4], t1 = scratch[5], t2 = scratch[6], t3 = scratch[7];
    a_total = s0 + s1 + s2 + s3;
    b_total = t0 + t1 + t2 + t3;
    b_excl = (w > 0 ? t0 : 0) + (w > 1 ? t1 : 0) + (w > 2 ? t2 : 0) + ib - b;
    return (w > 0 ? s0 : 0) + (w > 1 ? s1 : 0) + (w > 2 ? s2 : 0) + ia - a;
}

// The alpha of one (pixel, splat) pair -- the same instruction sequence in forward and backward, so both make the same
// skip decision (alpha < 1/255).  The per-splat record holds the exponent in completed-square form, in log2 units:
//   -0.5 d^T Q d * log2(e) + log2(o) = -(p (dx + s dy))^2 - (q dy)^2 - nlo,
//   p = sqrt(log2e/2 * c/det), s = -b/c, q = sqrt(log2e/2 / c), nlo = -log2(o)        (cov2D = [[a, b], [b, c]])
// i.e. 5 multiply-adds and ONE v_exp_f32 give opacity * exp(power); the exponent is a negated sum of squares, so the
// published "skip if power > 0" guard (a defence against rounding in the expanded quadratic form) can never fire here.
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float pair_alpha_unclamped(float dx, float dy, const float4 f) {
    const float t = f.x * fmaf(f.y, dy, dx), u = f.z * dy;
    const float w = fmaf(u, u, fmaf(t, t, f.w));
    return __builtin_amdgcn_exp2f(-w);   // = opacity * G
}
// The same exponent in TILE-relative coordinates, split into a per-(tile, entry) part -- computed once by the thread that stages
// the entry -- and a per-pixel part:
//   t = p (dx + s dy) = E0 - p X - (p s) Y,   u = q dy = F0 - q Y,      E0 = p (cx' + s cy'),  F0 = q cy',
// with (cx', cy') the splat centre and (X, Y) the pixel, both relative to the tile's first pixel (|cx'|, |cy'| <= support radius
// + 16, X, Y in 0..15: everything of order one, no cancellation of image-sized coordinates).  Per pair: 5 multiply-adds
// and one v_exp_f32 instead of 7 + v_exp_f32 in the forward blend, and 2 + v_exp_f32 per pixel column in the entry-per-lane
// backward, where the Y terms are per bucket (forward -6 %, backward replay -8 % VALU instructions).  Forward and backward
// evaluate the SAME expression tree on the same inputs (both through these two functions), so they make the same
// alpha >= 1/255 decisions bit for bit.
__device__ __forceinline__ void exponent_terms(float cx, float cy, const float4 f, float tx0f, float ty0f, float& E0, float& F0, float& ps) {
    const float cxr = cx - tx0f, cyr = cy - ty0f;
    E0 = f.x * fmaf(f.y, cyr, cxr);
    F0 = f.z * cyr;
    ps = f.x * f.y;
}
// row part (shared by the pixels of a row): G0 = E0 - ps Y,  K = (F0 - q Y)^2 + nlo;   pixel: opacity * G = exp2(-((G0 - p X)^2 + K))
__device__ __forceinline__ void pair_alpha_row(float Y, float E0, float F0, float ps, float q, float nlo, float& G0, float& K) {
    G0 = fmaf(-ps, Y, E0);
    const float L2 = fmaf(-q, Y, F0);
    K = fmaf(L2, L2, nlo);
}
__device__ __forceinline__ float pair_alpha_px(float X, float p, float G0, float K) {
    const float L1 = fmaf(-p, X, G0);
    return __builtin_amdgcn_exp2f(-fmaf(L1, L1, K));   // = opacity * G
}
// Q' = conic * log2(e) back from the factors (the support test works on the quadratic form)
// (The tile-reach decision built on this -- tile_test_prepare / tile_reached / edge_min -- is evaluated by k_count_tiles and
// again by k_emit, and the two MUST agree bit for bit, or a chunk's piece of a tile segment overflows into its neighbour's:
// floating-point contraction is switched off inside these functions, so that the result does not depend on which
// multiply-adds the compiler fuses in each inlining context.)
__device__ __forceinline__ void conic_from_factors(const float4 f, float& A, float& B, float& C) {
#pragma clang fp contract(off)
    A = 2.0f * f.x * f.x;
    B = f.y * A;
    C = fmaf(2.0f * f.z, f.z, f.y * B);
}

// Exact (up to a safety margin) test: is there a point of the 8x8 pixel-centre box starting at (sx, sy) where
// the splat's alpha can reach 1/255, i.e. min over the box of d^T Q d <= tau?  The minimum of a convex
// quadratic over a box is at the centre if it is inside, else on one of the 4 edges (1-D clamped minima).
__device__ __forceinline__ float edge_min(float a, float b2, float c, float inv_c, float fixed, float lo, float hi) {
#pragma clang fp contract(off)
    // min over t in [lo,hi] of a*fixed^2 + b2*fixed*t + c*t^2   (b2 = 2B).  The minimiser only needs to be
    // approximately right (the value is evaluated exactly at it; the error is second order), so 1/c is v_rcp_f32.
    const float t = fminf(hi, fmaxf(lo, -0.5f * b2 * fixed * inv_c));
    return a * fixed * fixed + (b2 * fixed + c * t) * t;
}
// box: pixel centres [sx, sx + extent] x [sy, sy + extent]
__device__ __forceinline__ bool box_overlap(const float4 r0, const float4 r1, float sx, float sy, float extent) {
    const float tau = r0.z;
    if (!(tau > 0.0f)) return false;
    const float x0 = sx - r0.x, x1 = x0 + extent, y0 = sy - r0.y, y1 = y0 + extent;  // box relative to the centre
    const bool in_x = x0 <= 0.0f && x1 >= 0.0f, in_y = y0 <= 0.0f && y1 >= 0.0f;
    if (in_x && in_y) return true;
    float A, B, C;
    conic_from_factors(r1, A, B, C);
    const float B2 = 2.0f * B;
    const float inv_a = __builtin_amdgcn_rcpf(A), inv_c = __builtin_amdgcn_rcpf(C);
    float fmin_ = edge_min(A, B2, C, inv_c, x0, y0, y1);
    fmin_ = fminf(fmin_, edge_min(A, B2, C, inv_c, x1, y0, y1));
    fmin_ = fminf(fmin_, edge_min(C, B2, A, inv_a, y0, x0, x1));
    fmin_ = fminf(fmin_, edge_min(C, B2, A, inv_a, y1, x0, x1));
    return fmin_ <= tau * 1.002f + 0.03f;
}
__device__ __forceinline__ bool subtile_overlap(const float4 r0, const float4 r1, float sx, float sy) {
    return box_overlap(r0, r1, sx, sy, (float)(kSub - 1));
}
// The same test for a whole 16x16 tile: the bucketing drops the (splat, tile) instances of a splat's tile rectangle whose tile
// holds no pixel the splat can reach with alpha >= 1/255 (the corners of the rectangle around a slanted or faint ellipse).
// Such an instance is blended by the published pipeline too, with every pixel failing the alpha test: it changes no output and
// no gradient, only the list lengths every later stage works through.
// Applied to splats whose rectangle has at least kCullMinTiles tiles: below that almost every tile is reached (the rectangle
// is already the bounding box of the support), and the test -- plus 32 bytes of record per splat in the two passes that
// apply it -- costs more than the few entries it removes (headline workload, test on every instance: +18 us for -15 us;
// from 4 tiles on: step -0.5 %, dense scenes -3.5 %, their sort -27 %).
constexpr uint32_t kCullMinTiles = 4;
constexpr uint32_t kDirectTiles = 9;   // rectangles up to this many tiles are walked by their own thread in the two instance passes
                                                    // (round 4, headline scan + emit: 4 -> +1.3 us, 6 -> -3.3, 9 -> -4.4, 16 -> -4.4; dense 100 k x 0.05: +1)
// Rectangles of up to kMaskTiles tiles get the outcome of the test once, in k_preprocess: bit k of the mask = tile k of the
// rectangle (row-major) can be reached (all ones below kCullMinTiles).  The two instance passes read the bit instead of the
// ellipse -- one evaluation instead of three, no 32-byte record read per splat in either pass, and the count pass and the
// scatter cannot disagree.  The 16 bits (bit 15 = mask present, bits 0..14 = mask) ride in the top nibbles of the four
// 16-bit fields of Geom::rect (tile coordinates are < 4096: images up to 65520 pixels a side, checked by the C ABI).
// (Packed into Geom::touched they cost k_preprocess_backward 2.4 us: its loads queued behind the decode of that word.)
constexpr uint32_t kMaskTiles = 9;   // <= 15 (the mask has 15 bits).  Measured 9 vs 15: headline equal; 100 k x 0.05: k_preprocess +1.9 vs +4.5 us
                                                 // (a wavefront pays the longest mask loop of its 64 splats)
constexpr uint32_t kMaskMinTiles = 1;   // masked rectangles of at least this many tiles are tested tile by tile: every one.
                                                       // (With the test paid once, the one- to three-tile rectangles are worth it too -- 4 / 2 / 1: k_preprocess
                                                       // +0 / +1.7 / +1.7 us, scatter + sort + blend pair -0 / -2.3 / -4.3 us at the headline.)
constexpr uint32_t kRectMasked = 0x8000u;
constexpr int kMaxTilesPerSide = 4095;
static_assert(kDirectTiles <= kMaskTiles, "the per-thread walk reads the mask");
__host__ __device__ inline ushort4 rect_pack(ushort4 r, uint32_t m16) {
    return make_ushort4((unsigned short)(r.x | ((m16 & 0xfu) << 12)), (unsigned short)(r.y | (((m16 >> 4) & 0xfu) << 12)),
                        (unsigned short)(r.z | (((m16 >> 8) & 0xfu) << 12)), (unsigned short)(r.w | (((m16 >> 12) & 0xfu) << 12)));
}
__host__ __device__ inline uint32_t rect_mask16(ushort4 p) {
    return (uint32_t)(p.x >> 12) | ((uint32_t)(p.y >> 12) << 4) | ((uint32_t)(p.z >> 12) << 8) | ((uint32_t)(p.w >> 12) << 12);
}
__host__ __device__ inline ushort4 rect_clean(ushort4 p) {
    return make_ushort4((unsigned short)(p.x & 0xfffu), (unsigned short)(p.y & 0xfffu), (unsigned short)(p.z & 0xfffu), (unsigned short)(p.w & 0xfffu));
}
// The per-splat half of the test is done once (tile_test_prepare, by the thread that stages the splat of a sub-batch): e0 =
// (centre x, centre y, threshold, A), e1 = (2 B, C, 1 / A, 1 / C) of the quadratic form; tile_reached then costs the four
// edge minima only.  A splat that can never be seen has threshold < 0.
__device__ __forceinline__ void tile_test_prepare(const float4 r0, const float4 r1, float4& e0, float4& e1) {
#pragma clang fp contract(off)
    float A, B, C;
    conic_from_factors(r1, A, B, C);
    e0 = make_float4(r0.x, r0.y, r0.z > 0.0f ? r0.z * 1.002f + 0.03f : -1.0f, A);
    e1 = make_float4(2.0f * B, C, __builtin_amdgcn_rcpf(A), __builtin_amdgcn_rcpf(C));
}
__device__ __forceinline__ bool tile_reached(const float4 e0, const float4 e1, uint32_t tile_x, uint32_t tile_y) {
#pragma clang fp contract(off)
    if (!(e0.z > 0.0f)) return false;
    const float x0 = (float)(tile_x * kTile) - e0.x, x1 = x0 + (float)(kTile - 1);  // the tile's pixel-centre box relative to the centre
    const float y0 = (float)(tile_y * kTile) - e0.y, y1 = y0 + (float)(kTile - 1);
    if (x0 <= 0.0f && x1 >= 0.0f && y0 <= 0.0f && y1 >= 0.0f) return true;
    const float A = e0.w, B2 = e1.x, C = e1.y, inv_a = e1.z, inv_c = e1.w;
    float fmin_ = edge_min(A, B2, C, inv_c, x0, y0, y1);
    fmin_ = fminf(fmin_, edge_min(A, B2, C, inv_c, x1, y0, y1));
    fmin_ = fminf(fmin_, edge_min(C, B2, A, inv_a, y0, x0, x1));
    fmin_ = fminf(fmin_, edge_min(C, B2, A, inv_a, y1, x0, x1));
    return fmin_ <= e0.z;
}

#endif  // __HIPCC__

}  // namespace sr
