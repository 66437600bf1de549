// Backward of the alpha compositing for gfx950 -- "entry-per-lane" formulation with MFMA moment reduction.
//
// Semantics: SURVEY.md Appendix A "Backward blend" (the renderCUDA backward of the published rasterizer reached from
// reference gaussian_renderer/__init__.py:94-102 through train.py:252), identical to what render.hip's forward blended.
//
// Why not one lane per pixel (the published mapping, and round 1's kernel): each list entry's ten gradient sums are
// sums over PIXELS, so a pixel-per-lane wavefront pays a 64-lane butterfly per entry (half of that kernel's issue
// slots), and with ~7 px supports only a quarter of the lanes of an 8x8 sub-tile are inside a splat at all.
//
// MI355X mapping used here:
//   * work unit = (4x4 pixel quad, bucket of 16 list entries that can reach the quad).  A wavefront holds the bucket's
//     entries in lanes n = lane & 15 and the quad's four pixel rows in k = lane >> 4; it walks the quad's four pixel
//     columns t = 0..3, so every step evaluates 16 entries x 4 pixels.
//   * transmittance / "colour behind" recurrences along the list become two 16-lane prefix scans per step (DPP row
//     shifts, carry between buckets in lane 15) -- no per-pixel serial loop, no early-exit divergence.
//   * the ten sums over pixels are a dense contraction with operands SHARED by all entries of the quad:
//       [M0 MX MY MXX MXY MYY](entry) = W(6 x 16 pixels)  . g1(16 pixels x entry)      (pixel monomials 1, X, Y, X^2, XY, Y^2)
//       [dr dg db dd](entry)          = G(4 x 16 pixels)  . wgt(16 pixels x entry)     (upstream colour/depth gradients)
//     = two v_mfma_f32_16x16x4_f32 per step (exact fp32 fma chains), accumulated over the four steps: no butterflies,
//     no products with dx, dy in the VALU stream.  Moments are taken about the TILE centre and shifted to the splat
//     centre once per (tile, entry) when the quads' partial sums are combined.
//   * the quads' entry lists are built per chunk of the tile list from an exact-support test (quadmask.h), bucketed
//     with ballots + prefix counts; a slot of the LDS scratch first carries the entry's record to the wavefront that
//     owns the quad and then carries the 10 sums back.  Everything is summed in a fixed order: bit-reproducible,
//     no floating-point atomics (as before).
#include "kernels.h"
#define SR_QM_DEVICE 1
#include "quadmask.h"

namespace sr {

namespace {

constexpr int kBucket = 16;                 // list entries per bucket = N of the MFMA
constexpr int kBlkEntries = 64;             // list entries tested by one wavefront pass ("block")
#ifndef SR_BWD_MAXE
#define SR_BWD_MAXE 2
#endif
constexpr int kMaxE = SR_BWD_MAXE;          // blocks per wavefront per chunk (chunk <= 256 * kMaxE list entries)
constexpr int kMaxBlocks = 4 * kMaxE;
#ifndef SR_BWD_CAP
#define SR_BWD_CAP 1024
#endif
constexpr int kCap = SR_BWD_CAP;            // (quad, entry) slots per chunk; one block can need 64 x 16 = 1024
static_assert(kCap >= 1024 && kCap % 16 == 0, "one block must always fit");

typedef float f32x4 __attribute__((ext_vector_type(4)));

// DPP controls: row_shr:n = 0x110 + n (lane l reads lane l - n of its row of 16), row_mirror = 0x140, row_ror:n = 0x120 + n
constexpr int kShr1 = 0x111, kShr2 = 0x112, kShr4 = 0x114, kShr8 = 0x118, kMirror = 0x140;

// value of lane (l - shift) of the row; lanes without a source (and nothing else) keep `old`
template <int CTRL> __device__ __forceinline__ float dpp_f(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, 0xf, false);
}

#ifdef SR_BWD_SCAN_SHFL
// reference implementation of the row scans (ds_bpermute): debugging aid, selected at build time
__device__ __forceinline__ float row_scan_add(float x) {
    const int n = threadIdx.x & 15;
    for (int d = 1; d < 16; d <<= 1) { const float y = __shfl_up(x, d, 16); if (n >= d) x += y; }
    return x;
}
__device__ __forceinline__ float row_scan_mul(float x) {
    const int n = threadIdx.x & 15;
    for (int d = 1; d < 16; d <<= 1) { const float y = __shfl_up(x, d, 16); if (n >= d) x *= y; }
    return x;
}
// lane 0 of each row: `carry` of lane 15; lane l >= 1: v of lane l - 1
__device__ __forceinline__ float shift_in_carry(float carry, float v) {
    const int n = threadIdx.x & 15;
    const float c = __shfl(carry, 15, 16), y = __shfl_up(v, 1, 16);
    return n == 0 ? c : y;
}
#else
// inclusive prefix over each row of 16 lanes (Kogge-Stone, 4 DPP steps)
__device__ __forceinline__ float row_scan_add(float x) {
    x += dpp_f<kShr1>(0.f, x); x += dpp_f<kShr2>(0.f, x); x += dpp_f<kShr4>(0.f, x); x += dpp_f<kShr8>(0.f, x);
    return x;
}
// The product scan needs "lanes without a source keep their own value", i.e. v_mul_f32_dpp with the destination as
// `old` operand; the compiler only fuses DPP moves whose fill value is 0 (it emits v_mov 1.0 + v_mov_dpp + v_mul, three
// issue slots per level), so the four levels are written out.  s_nop 1 = the two wait states a DPP read needs after the
// VALU write of its source (the hazard recogniser does not look inside inline assembly).
__device__ __forceinline__ float row_scan_mul(float x) {
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    return x;
}
__device__ __forceinline__ float shift_in_carry(float carry, float v) {
    return dpp_f<kShr1>(dpp_f<kMirror>(carry, carry), v);   // row_mirror: lane 0 <- lane 15
}
#endif

__device__ __forceinline__ uint32_t row_allsum_u32(uint32_t v) {   // every lane of a row: the row's sum
    v += dpp_u<0xB1>(0u, v); v += dpp_u<0x4E>(0u, v); v += dpp_u<0x124>(0u, v); v += dpp_u<0x128>(0u, v);
    return v;
}
__device__ __forceinline__ uint32_t row_scan_add_u32(uint32_t x) {
    x += dpp_u<kShr1>(0u, x); x += dpp_u<kShr2>(0u, x); x += dpp_u<kShr4>(0u, x); x += dpp_u<kShr8>(0u, x);
    return x;
}

}  // namespace

__global__ void __launch_bounds__(kBlock) k_render_backward_mfma(const ViewK v, const Geom g, const Binning b, const Image im,
                                                                 const float* __restrict__ dL_dcolor,
                                                                 const float* __restrict__ dL_ddepth,
                                                                 const float* __restrict__ dL_dalpha,
                                                                 float* __restrict__ slots, uint8_t* __restrict__ reached) {
    // slot i: [3 i] = (cx, cy, list position, -) -> (M0, MX, MY, MXX); [3 i + 1] = (p, s, q, -log2 o) -> (MXY, MYY, dr, dg);
    //         [3 i + 2] = (r, g, b, depth) -> (db, dd, 0, 0)
    __shared__ float4 s_slot[kCap * 3];
    __shared__ float4 s_pixA[256];              // per pixel of the tile: dL/d(r, g, b, depth)
    __shared__ float4 s_pixB[256];              // (dL/dalpha, last contributor + 1 [int bits], T state, "behind . g" state)
    __shared__ uint64_t s_mask[16][kMaxBlocks]; // [quad][block]: lanes of the block whose entry reaches the quad
    __shared__ uint16_t s_bb[16][kMaxBlocks];   // slot of the first entry of (quad, block)
    __shared__ uint32_t s_qlen[16], s_qbase[16], s_qlast[16];
    __shared__ uint32_t s_ctl[2];               // blocks used by this chunk, (quad, entry) pairs in them

    const int tile = (int)g.tile_order[blockIdx.x];  // longest lists first
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int wave = wave_id(), lane = lane_id();
    const uint32_t start = g.tile_start[tile], end = g.tile_start[tile + 1];
    const int n = (int)(end - start);
    const float tx0f = (float)(tx * kTile), ty0f = (float)(ty * kTile);

    // ---- per-pixel inputs: thread i <-> pixel (i & 15, i >> 4) of the tile ----
    {
        const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
        const int px = tx * kTile + lx, py = ty * kTile + ly;
        const bool inside = px < v.W && py < v.H;
        const size_t hw = (size_t)v.H * v.W, pix = (size_t)py * v.W + px;
        uint32_t my_last = 0;
        float T_final = 0.f, gR = 0.f, gG = 0.f, gB = 0.f, gD = 0.f, gA = 0.f;
        if (inside) {
            my_last = im.n_contrib[pix];
            T_final = im.final_T[pix];
            gR = dL_dcolor[pix]; gG = dL_dcolor[hw + pix]; gB = dL_dcolor[2 * hw + pix];
            if (dL_ddepth) gD = dL_ddepth[pix];
            if (dL_dalpha) gA = dL_dalpha[pix];
        }
        const float bg_dot = v.bg[0] * gR + v.bg[1] * gG + v.bg[2] * gB;
        s_pixA[threadIdx.x] = make_float4(gR, gG, gB, gD);
        s_pixB[threadIdx.x] = make_float4(gA, __uint_as_float(my_last), T_final, T_final * bg_dot);
        if (threadIdx.x < 16) s_qlast[threadIdx.x] = 0u;
        __syncthreads();
        atomicMax(&s_qlast[(ly >> 2) * 4 + (lx >> 2)], my_last);   // LDS, integer: order-independent
        __syncthreads();
    }
    uint32_t bmax_u = 0u, qlast_min = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const uint32_t x = s_qlast[q]; bmax_u = max(bmax_u, x); qlast_min = min(qlast_min, x); }
    const int bmax = __builtin_amdgcn_readfirstlane((int)bmax_u);   // uniform by construction; tell the compiler
    qlast_min = (uint32_t)__builtin_amdgcn_readfirstlane((int)qlast_min);

    float4* slot4 = reinterpret_cast<float4*>(slots);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // List entries behind every pixel's last contributor receive no gradient.  With `flags` their slots are not written
    // at all and their `reached` byte stays 0 (the buffer is cleared before the launch); otherwise they are zero-filled.
    const bool flags = use_reached_flags(g.total);
    if (!flags) {
        for (int i = bmax + (int)threadIdx.x; i < n; i += kBlock) {
            const uint32_t id = b.sorted_id[start + i];
            const ushort4 rc = g.rect[id];
            const size_t inst = g.offsets[id] + (uint32_t)(ty - rc.y) * (uint32_t)(rc.z - rc.x) + (uint32_t)(tx - rc.x);
            slot4[inst * 3] = zero4; slot4[inst * 3 + 1] = zero4; slot4[inst * 3 + 2] = zero4;
        }
    }

    const int k = lane >> 4, nl = lane & 15;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    int hi = bmax;   // list entries [0, hi) are still to be replayed (back to front)
    int E = 1;       // blocks per wavefront in the next chunk (adapted to the density of (quad, entry) pairs)
    while (hi > 0) {
        // ---------------- (A) test: which quads does each entry of the chunk reach? ----------------
        uint32_t qm[kMaxE], inst_of[kMaxE];
        int pos_of[kMaxE];
        float4 r0_of[kMaxE], r1_of[kMaxE], r2_of[kMaxE];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            qm[e] = 0u; inst_of[e] = 0u; pos_of[e] = -1;
            r0_of[e] = zero4; r1_of[e] = zero4; r2_of[e] = zero4;
            if (e < E) {
                const int blk = e * 4 + wave;
                const int pos = hi - 1 - (blk * kBlkEntries + lane);   // descending list position
                pos_of[e] = pos;
                if (pos >= 0) {
                    const uint32_t id = b.sorted_id[start + (uint32_t)pos];
                    const float4* rec = g.rec + 4 * (size_t)id;
                    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                    const uint32_t first = g.offsets[id];
                    const uint32_t xy = __float_as_uint(r3.x), rw = __float_as_uint(r3.y);
                    inst_of[e] = first + ((uint32_t)ty - (xy >> 16)) * rw + ((uint32_t)tx - (xy & 0xffffu));
                    r0_of[e] = r0; r1_of[e] = r1; r2_of[e] = r2;
                    uint32_t m = sr_quad_mask(r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, tx0f, ty0f);
                    if ((uint32_t)pos >= qlast_min) {   // behind the last contributor of every pixel of some quad
#pragma unroll
                        for (int q = 0; q < 16; ++q) if ((uint32_t)pos >= s_qlast[q]) m &= ~(1u << q);
                    }
                    qm[e] = m;
                }
                // one ballot per quad; lane q (< 16) collects quad q's mask
                uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint64_t bal = __builtin_amdgcn_ballot_w64(((qm[e] >> q) & 1u) != 0u);
                    const bool mine = lane == q;
                    mlo = mine ? (uint32_t)bal : mlo;
                    mhi = mine ? (uint32_t)(bal >> 32) : mhi;
                }
                if (lane < 16) s_mask[lane][blk] = ((uint64_t)mhi << 32) | mlo;
            }
        }
        __syncthreads();
        // ---------------- (B) slot assignment (first wavefront): longest prefix of blocks that fits ----------------
        if (wave == 0) {
            const int q = nl;   // the four rows of the wavefront compute the same thing
            uint32_t len = 0u, used = 0u;
            uint32_t first_of[kMaxBlocks];
            bool open = true;
#pragma unroll
            for (int blk = 0; blk < kMaxBlocks; ++blk) {
                first_of[blk] = len;
                if (blk < 4 * E && open) {
                    const uint32_t c = (uint32_t)__popcll(s_mask[q][blk]);
                    const uint32_t padded = (len + c + 15u) & ~15u;
                    if (row_allsum_u32(padded) <= (uint32_t)kCap) { len += c; used = blk + 1; }
                    else open = false;
                }
            }
            const uint32_t padded = (len + 15u) & ~15u;
            const uint32_t base = row_scan_add_u32(padded) - padded;
            const uint32_t pairs = row_allsum_u32(len);
            if (lane < 16) {
                s_qlen[q] = len; s_qbase[q] = base;
#pragma unroll
                for (int blk = 0; blk < kMaxBlocks; ++blk) s_bb[q][blk] = (uint16_t)(base + first_of[blk]);
            }
            if (lane == 0) { s_ctl[0] = used; s_ctl[1] = pairs; }
        }
        __syncthreads();
        const int used = __builtin_amdgcn_readfirstlane((int)s_ctl[0]);
        const uint32_t pairs = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ctl[1]);
        // ---------------- (C) scatter the records into the quads' slot runs ----------------
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const int blk = e * 4 + wave;
            if (e < E && blk < used) {
                uint32_t m = qm[e];
                while (m) {
                    const int q = __builtin_ctz(m);
                    m &= m - 1u;
                    const uint32_t slot = (uint32_t)s_bb[q][blk] + (uint32_t)__popcll(s_mask[q][blk] & lt_mask);
                    s_slot[3 * slot] = make_float4(r0_of[e].x, r0_of[e].y, __int_as_float(pos_of[e]), 0.f);
                    s_slot[3 * slot + 1] = r1_of[e];
                    s_slot[3 * slot + 2] = r2_of[e];
                }
            }
        }
        __syncthreads();
        // ---------------- (D) replay: each wavefront walks the buckets of its four quads ----------------
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int qy = j, qx = (wave - j) & 3;   // one quad per quad-row and quad-column: balances the wavefronts
            const int q = qy * 4 + qx;
            const int len = __builtin_amdgcn_readfirstlane((int)s_qlen[q]);
            if (len == 0) continue;
            const int base = __builtin_amdgcn_readfirstlane((int)s_qbase[q]);
            const int prow = (4 * qy + k) * 16 + 4 * qx;   // tile-local index of pixel (t = 0, row k) of the quad
            float pxf[4], gR[4], gG[4], gB[4], gD[4], gA[4], ST[4], SB[4], A1[4], A2[4];
            int last[4];
            const float pyf = ty0f + (float)(4 * qy + k);
            const float Y = (float)(4 * qy + k) - 7.5f;
            // MFMA A operands: this lane supplies row m = lane & 15 of the 16 x 4 operand for pixel row k.
            // rows 0-5: pixel monomials 1, X, Y, X^2, XY, Y^2 = c0 + X (c1 + X c2); rows 6-9: dL/d(r, g, b, depth)
            const float c0 = nl == 0 ? 1.0f : nl == 2 ? Y : nl == 5 ? Y * Y : 0.0f;
            const float c1 = nl == 1 ? 1.0f : nl == 4 ? Y : 0.0f;
            const float c2 = nl == 3 ? 1.0f : 0.0f;
            const float w6 = nl == 6 ? 1.0f : 0.0f, w7 = nl == 7 ? 1.0f : 0.0f, w8 = nl == 8 ? 1.0f : 0.0f, w9 = nl == 9 ? 1.0f : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 a = s_pixA[prow + t], c = s_pixB[prow + t];
                gR[t] = a.x; gG[t] = a.y; gB[t] = a.z; gD[t] = a.w;
                gA[t] = c.x; last[t] = __float_as_int(c.y); ST[t] = c.z; SB[t] = c.w;
                pxf[t] = tx0f + (float)(4 * qx + t);
                const float X = (float)(4 * qx + t) - 7.5f;
                A1[t] = fmaf(X, fmaf(X, c2, c1), c0);
                A2[t] = fmaf(w9, a.w, fmaf(w8, a.z, fmaf(w7, a.y, w6 * a.x)));
            }
#pragma unroll 1
            for (int i0 = 0; i0 < len; i0 += kBucket) {
                const int slot = base + i0 + nl;
                const bool valid = i0 + nl < len;
                float4 e0 = s_slot[3 * slot], e1 = s_slot[3 * slot + 1], e2 = s_slot[3 * slot + 2];
                if (!valid) { e0 = zero4; e1 = make_float4(0.f, 0.f, 0.f, __builtin_inff()); e2 = zero4; }   // alpha = 0
                const int pos = valid ? __float_as_int(e0.z) : 0x7fffffff;
                const float dy = e0.y - pyf;
                f32x4 D1 = {0.f, 0.f, 0.f, 0.f}, D2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float dx = e0.x - pxf[t];
                    const float oG = pair_alpha_unclamped(dx, dy, e1);   // opacity * G: the forward's instruction sequence
                    const bool hit = (oG >= kAlphaMin) && (pos < last[t]);
                    const float oGc = hit ? oG : 0.0f;                   // everyone else: alpha = G = 0, transparent to the scans
                    const float alpha = __builtin_amdgcn_fmed3f(oGc, 0.0f, kAlphaMax);
                    const float ginv = __builtin_amdgcn_rcpf(1.0f - alpha);
                    // transmittance in front of entry n: T_n = (T behind the bucket) * prod_{j <= n} 1 / (1 - alpha_j)
                    const float T = row_scan_mul(shift_in_carry(ST[t], ginv)) * ginv;
                    ST[t] = T;   // lane 15: behind the next bucket
                    const float wgt = alpha * T;
                    float cgv = gA[t];   // the alpha channel's "colour" is 1 for every splat
                    cgv = fmaf(e2.w, gD[t], cgv); cgv = fmaf(e2.z, gB[t], cgv); cgv = fmaf(e2.y, gG[t], cgv); cgv = fmaf(e2.x, gR[t], cgv);
                    const float z = wgt * cgv;
                    // (colour accumulated behind entry n, incl. background) . upstream gradient
                    const float behind = row_scan_add(shift_in_carry(SB[t], z));
                    SB[t] = behind + z;
                    // dL/dalpha_n = T_n (c_n . g) - behind_n / (1 - alpha_n); gradients pass through the 0.99 clamp, as upstream
                    const float dLa = T * cgv - ginv * behind;
                    const float g1 = oGc * dLa;   // the six geometric sums carry the factor `opacity` (k_preprocess_backward)
                    D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[t], g1, D1, 0, 0, 0);
                    D2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[t], wgt, D2, 0, 0, 0);
                }
                // D rows 4k..4k+3 of entry column n live in lane (k, n): rows 0-5 moments, 6-9 colour / depth sums
                if (k < 3) s_slot[3 * slot + k] = make_float4(D1[0] + D2[0], D1[1] + D2[1], D1[2] + D2[2], D1[3] + D2[3]);
            }
            if (nl == 15) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float* st = reinterpret_cast<float*>(&s_pixB[prow + t]);
                    st[2] = ST[t]; st[3] = SB[t];
                }
            }
        }
        __syncthreads();
        // ---------------- (E) combine the quads' sums of every entry, shift the moments to the splat centre ----------------
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const int blk = e * 4 + wave;
            if (e < E && blk < used && pos_of[e] >= 0) {
                float4 s0 = zero4, s1 = zero4, s2 = zero4;
                uint32_t m = qm[e];
                while (m) {   // ascending quad index: fixed summation order
                    const int q = __builtin_ctz(m);
                    m &= m - 1u;
                    const uint32_t slot = (uint32_t)s_bb[q][blk] + (uint32_t)__popcll(s_mask[q][blk] & lt_mask);
                    const float4 a = s_slot[3 * slot], c = s_slot[3 * slot + 1], d = s_slot[3 * slot + 2];
                    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
                    s1.x += c.x; s1.y += c.y; s1.z += c.z; s1.w += c.w;
                    s2.x += d.x; s2.y += d.y;
                }
                // moments about the tile centre (X, Y = pixel - centre) -> sums of g1 dx^a dy^b with dx = cx - pixel x = ox - X
                const float ox = r0_of[e].x - (tx0f + 7.5f), oy = r0_of[e].y - (ty0f + 7.5f);
                const float M0 = s0.x, MX = s0.y, MY = s0.z, MXX = s0.w, MXY = s1.x, MYY = s1.y;
                const float Sx = ox * M0 - MX, Sy = oy * M0 - MY;
                const float Sxx = fmaf(ox, Sx - MX, MXX);               // ox^2 M0 - 2 ox MX + MXX
                const float Syy = fmaf(oy, Sy - MY, MYY);
                const float Sxy = fmaf(ox, Sy, MXY) - oy * MX;          // ox oy M0 - ox MY - oy MX + MXY
                const size_t inst = inst_of[e];
                slot4[inst * 3] = make_float4(M0, Sx, Sy, Sxx);
                slot4[inst * 3 + 1] = make_float4(Sxy, Syy, s1.z, s1.w);
                slot4[inst * 3 + 2] = make_float4(s2.x, s2.y, 0.f, 0.f);
                if (flags) reached[inst] = 1;
            }
        }
        __syncthreads();   // the slots and masks are reused by the next chunk
        hi -= kBlkEntries * used;
        // next chunk: as many blocks as fit at the density of (quad, entry) pairs just seen
        const float per_block = fmaxf((float)pairs / (float)max(used, 1), 1.0f);
        E = min(kMaxE, max(1, (int)(0.9f * (float)kCap / per_block) / 4));
    }
}

void launch_render_backward_mfma(const ViewK& v, const Geom& g, const Binning& b, const Image& im,
                                 const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                                 float* slots, uint8_t* reached, hipStream_t st) {
    const int tiles = v.gx * v.gy;
    if (tiles <= 0) return;
    hipLaunchKernelGGL(k_render_backward_mfma, dim3(tiles), dim3(kBlock), 0, st, v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots, reached);
}

}  // namespace sr
