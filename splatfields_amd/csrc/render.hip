// Alpha compositing (forward) and its replay (backward) for gfx950.
//
// Semantics: SURVEY.md Appendix A "Forward blend" / "Backward blend" -- the renderCUDA kernels of
// the published rasterizer reached from reference gaussian_renderer/__init__.py:94-102, plus the
// depth output of the pinned fork and an alpha (= 1 - final transmittance) output that makes the
// reference's second "mask" pass (gaussian_renderer/__init__.py:104-115) unnecessary.
//
// MI355X mapping: one workgroup (4 wavefronts) per 16x16 binning tile; each 64-lane wavefront owns
// one 8x8 sub-tile.  A batch of the tile's sorted list is staged once in LDS by all 256 threads
// (coalesced id loads + 48-byte record gathers), then every wavefront compacts -- with one ballot per
// 64 entries -- the indices of the splats whose alpha >= 1/255 support box reaches *its* 8x8 pixels
// and blends only those, reading records back with LDS broadcast reads.  Early termination is per
// wavefront.  The backward pass reduces the per-pixel gradient contributions with DPP wave
// reductions, combines the 4 wavefronts through LDS in a fixed order and writes ONE record per
// tile-splat instance to a scratch slot: no floating-point atomics anywhere, bit-reproducible.
#include "kernels.h"
#define SR_QM_DEVICE 1
#include "quadmask.h"

namespace sr {

constexpr int kBwdBatch = 128;  // 64 / 96 / 192 / 256 all measured slower (NOTEBOOK.md, tried and rejected)

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Tile order: workgroup b renders tile b.  Consecutive workgroups land on different XCDs (b % 8), which interleaves
// heavy (image centre) and light tiles across the 8 XCDs.  Dealing each XCD a contiguous band of tiles for L2 locality
// was measured and is WORSE (forward 0.163 -> 0.219 ms, backward 0.528 -> 0.639 ms): the kernels are VALU-bound, so the
// XCD load imbalance costs more than the shared-record L2 hits save.
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

// x, y hold two different quantities per lane.  Returns r with r[l] = x[l] + x[l+32] for l < 32 and
// r[l] = y[l-32] + y[l] for l >= 32 (one v_permlane32_swap + one add): each half-wave now owns one quantity.
// mask ? a : b with the lane mask held in a scalar register pair (v_cndmask_b32 takes it directly)
__device__ __forceinline__ float mask_select(uint64_t mask, float a, float b) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
__device__ __forceinline__ uint32_t mask_select(uint64_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}

__device__ __forceinline__ float swap32_add(float x, float y) {
    const uint2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// Same one level down: rows (16 lanes) 0 and 2 end up with x's row pairs (0+1, 2+3), rows 1 and 3 with y's.
__device__ __forceinline__ float swap16_add(float x, float y) {
    const uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// What the backward needs to know about how far the forward got, written once per tile at the end of the forward blend:
//   tile_qlast[tile][quad] = max over the quad's pixels of `last` (entries in front of the pixel's stop).
// The backward blend replays entries [0, max over the quads) without first reducing the per-pixel values itself.
// D1..D4: the four lane-index bits that enumerate the 16 pixels of a quad in the caller's lane layout; `writer`: one lane
// per quad.
__device__ __forceinline__ void publish_tile_reach(const Geom& g, int tile, uint32_t last, int qx, int qy, int D1, int D2, int D3, int D4,
                                                   bool writer) {
    uint32_t ql = last;
    ql = max(ql, (uint32_t)__shfl_xor((int)ql, D1, 64));
    ql = max(ql, (uint32_t)__shfl_xor((int)ql, D2, 64));
    ql = max(ql, (uint32_t)__shfl_xor((int)ql, D3, 64));
    ql = max(ql, (uint32_t)__shfl_xor((int)ql, D4, 64));
    if (writer) g.tile_qlast[16 * (size_t)tile + 4 * qy + qx] = ql;
}

// ------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_render_forward(const ViewK v, const Geom g, const Binning b, const Image im,
                                                           float* __restrict__ out_color, float* __restrict__ out_depth,
                                                           float* __restrict__ out_alpha) {
    // Staging area of one batch of the tile's list: the three hot quarters of every entry's record and its quad-reach mask.
    // ONE copy of 256 entries (12.5 KB of LDS, eight workgroups per CU).  (Two copies with the next batch in flight during the
    // blending -- 25 KB, six per CU -- and the next batch travelling through registers were both measured and are slower or equal:
    // the blend loop is issue-bound and lives on its occupancy, the CU's other workgroups already cover a batch's gather;
    // NOTEBOOK.md, rejected tables of rounds 3 and 5.)
    constexpr int kB = 256;
    static_assert(kB == 256 || kB == 128, "a batch is staged by four or by two wavefronts");
    constexpr int kStagers = kB / kWave;       // wavefronts that stage one batch
    constexpr uint32_t kStep = 4 / kStagers;   // a wavefront stages every kStep-th batch
    __shared__ float4 s_r0[1][kB];
    __shared__ float4 s_r1[1][kB];
    __shared__ float4 s_r2[1][kB];
    __shared__ uint16_t s_qm[1][kB];   // quad-reach mask of every staged entry (quadmask.h)
    __shared__ uint32_t s_live[2][4];            // per batch parity and wavefront: does it still have accumulating pixels?
    __shared__ uint8_t s_idx[4][4][kB + 4];      // per wavefront and quad of its sub-tile: the batch entries that reach the quad
                                                 // (+ 4: the pipelined walk reads its index two trips ahead)
    // the pipelined walk fetches a record by an index byte it has not masked yet (a stale byte of s_idx): any byte must be a slot
    static_assert(kB == 256, "the pipelined walk indexes the staging area with unmasked bytes: 256 slots");

    // (the three scalar loads are requested together and tested with one wait: written as `a > x || b > y` ahead of the tile
    // lookup they were three dependent memory round trips at the head of every workgroup)
    const uint32_t total_instances = g.total[0], longest_list = g.total[1];
    const int tile = (int)g.tile_order[blockIdx.x];  // longest lists first
    asm volatile("" :: "s"(tile));   // (keeps the compiler from sinking this load below the early exit: it is requested with the two above)
    if ((total_instances > b.capacity) | (longest_list > b.sorted_up_to)) {   // uniform: see sr_forward / sr_forward_async
        // The launch was made on a promise (sr_forward_async: capacity / sort classes of the camera's last render) that did not
        // hold: the lists of this view were never built.  The view has NO result, and it must not be possible to mistake what
        // the output buffers happen to hold for one: every workgroup fills its tile with NaN (by grid position: the launch
        // order table is not needed), so a loss computed from this image is NaN until the caller, told by its ticket
        // (RasterizerOverflow), renders again.
        const int t = (int)blockIdx.x, lx = (int)threadIdx.x & 15, ly = (int)threadIdx.x >> 4;
        const int qx = (t % v.gx) * kTile + lx, qy = (t / v.gx) * kTile + ly;
        if (qx < v.W && qy < v.H) {
            const size_t hw = (size_t)v.H * v.W, pix = (size_t)qy * v.W + qx;
            const float nan = __builtin_nanf("");
            out_color[pix] = nan; out_color[hw + pix] = nan; out_color[2 * hw + pix] = nan;
            out_depth[pix] = nan;
            if (out_alpha) out_alpha[pix] = nan;
        }
        return;
    }
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int wave = __builtin_amdgcn_readfirstlane(wave_id()), lane = lane_id();
    const int sx = tx * kTile + (wave & 1) * kSub, sy = ty * kTile + (wave >> 1) * kSub;
    // row j of 16 lanes = quad j of the sub-tile (4x4 pixels): every row walks ITS quad's entries
    const int qrow = lane >> 4;
    const int px = sx + 4 * (qrow & 1) + (lane & 3), py = sy + 4 * (qrow >> 1) + ((lane >> 2) & 3);
    const bool inside = px < v.W && py < v.H;
    const float tx0f = (float)(tx * kTile), ty0f = (float)(ty * kTile);
    const float Xf = (float)(px - tx * kTile), Yf = (float)(py - ty * kTile);   // this lane's pixel relative to the tile
    const uint32_t start = g.tile_start[tile], end = g.tile_start[tile + 1];
    // the four 4x4 quads of this wavefront's 8x8 sub-tile as bits 4 qy + qx
    const uint32_t my_quads = 0x33u << (2 * (wave & 1) + 8 * (wave >> 1));

    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, D = 0.f;
    // Pixels still accumulating (not yet stopped by T < 1e-4) are tracked as a SCALAR 64-bit mask: all the skip / stop /
    // blend logic of an entry is mask arithmetic on the scalar unit (v_cmp results are scalar masks already), and the only
    // per-lane selects left are the two v_cndmask below.  `last` = number of list entries in front of the one that stopped
    // the pixel (the whole list if it never stopped): the backward replays entries [0, last) -- entries behind a pixel's
    // true last contributor fail the alpha test there exactly as they did here, so the looser bound changes no result.
    uint64_t livem = __builtin_amdgcn_ballot_w64(inside);
    uint32_t last = inside ? end - start : 0u;
    if (start == end) {   // uniform: nothing to blend (the reach table still has to be written)
        goto done;
    }
    {
    // Which wavefronts stage batch k: all four (256 entries), or {0, 1} for even k and {2, 3} for odd k (128 entries).
    auto stages = [&](uint32_t k) { return kStagers == 4 || (uint32_t)(wave >> 1) == (k & 1u); };
    const int slot = kStagers == 4 ? (int)threadIdx.x : (int)(threadIdx.x & 127u);   // this thread's entry of a batch it stages
    const int wslot = kStagers == 4 ? wave : (wave & 1);                              // its wavefront's 64-entry group there
    // LDS byte addresses of this wavefront's 64 slots in each copy (wave-uniform: the copy instruction adds 16 x lane)
    const uint32_t l0 = lds_address(&s_r0[0][wslot * kWave]), l1 = lds_address(&s_r1[0][wslot * kWave]), l2 = lds_address(&s_r2[0][wslot * kWave]);
    constexpr uint32_t kCopyStride = kB * sizeof(float4);
    // request the three record quarters of a list entry (index clamped into the list: threads past its end re-request the
    // last entry, whose mask below is zero) into copy `buf`: LDS-direct loads, no staging registers, no ds_write
    auto request = [&](int buf, uint32_t id) {
        const float4* rec = g.rec + 4 * (size_t)id;
        lds_copy16_async(rec, l0 + buf * kCopyStride);
        lds_copy16_async(rec + 1, l1 + buf * kCopyStride);
        lds_copy16_async(rec + 2, l2 + buf * kCopyStride);
    };
    // after the copies of `buf` have landed: this thread's entry -> its quad-reach mask (exact support of alpha >= 1/255
    // against the sixteen 4x4 quads, quadmask.h): LDS for this kernel's sub-tile culling, memory for the backward's bucketing
    auto publish_mask = [&](int buf, uint32_t i) {
        uint32_t qm = 0u;
        if (i < end) {
            const float4 r0 = s_r0[buf][slot], r1 = s_r1[buf][slot];
            qm = sr_quad_mask(r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, tx0f, ty0f);
            b.qmask[i] = qm;
            // the per-(tile, entry) part of the exponent (common.h: exponent_terms) replaces the centre and the factor s in
            // the staged record: the blend loop needs neither any more
            float E0, F0, ps;
            exponent_terms(r0.x, r0.y, r1, tx0f, ty0f, E0, F0, ps);
            *reinterpret_cast<float2*>(&s_r0[buf][slot]) = make_float2(E0, F0);
            s_r1[buf][slot].y = ps;
        }
        s_qm[buf][slot] = (uint16_t)qm;
    };
    const uint32_t lastpos = end - 1u;
    const uint32_t i0 = start + (uint32_t)slot;   // this thread's entry of batch 0
    // Splat indices are plain loads with the index clamped into the list (unconditional: no branch).  id_nxt = index of this
    // thread's entry in the NEXT batch it stages; it is (re)loaded right after the thread's copies have landed and before
    // its next copies go out: while LDS-direct copies are in flight no ordinary load may be pending, because the wait the
    // compiler places in front of its first use counts (and would drain) the copies as well.
    const uint32_t k_first = kStagers == 4 ? 1u : ((wave >> 1) == 1 ? 1u : 2u);
    uint32_t id_nxt = b.sorted_id[min(i0 + k_first * kB, lastpos)];
    if (stages(0u)) {
        const uint32_t id0 = b.sorted_id[min(i0, lastpos)];
        request(0, id0);
        lds_copy_wait();
        publish_mask(0, i0);
    }
    asm volatile("" :: "v"(id_nxt));
    lds_barrier();

    uint32_t k = 0;
    for (uint32_t base = start; base < end; base += kB, ++k) {
        const bool more = base + kB < end;   // uniform
        constexpr int buf = 0, nbuf = 0;   // one staging copy
        // (The empty statement makes the compiler wait for the index load on EVERY path into the blend loop: it must not
        // believe a load is still pending there, or it protects the reuse of that register with a wait inside the loop.)
        asm volatile("" :: "v"(id_nxt));
        const int cnt = (int)min((uint32_t)kB, end - base);
        if (livem != 0ull) {
            // Quad-granular walk: a splat's support is ~7 pixels wide, so of the entries that reach an 8x8 sub-tile most reach
            // one or two of its four quads -- walked by all 64 lanes, three quarters of them evaluate pixels the splat cannot
            // reach.  Here every row of 16 lanes (one quad) walks only the entries whose quad-reach mask has ITS quad: the
            // wavefront first compacts, per quad, the batch indices of those entries (a ballot + a lane rank per quad and 64
            // entries, one byte store each), then runs max(list lengths) trips in which the four rows read four different
            // entries.  The skip / stop / blend logic stays scalar mask arithmetic (a lane's bit is a lane's bit).
            uint32_t qlen[4] = {0u, 0u, 0u, 0u};
            const int qbit0 = 4 * (2 * (wave >> 1)) + 2 * (wave & 1);   // tile quad index of sub-tile quad 0; +1, +4, +5 for 1, 2, 3
            for (int c = 0; c < cnt; c += kWave) {
                const int el = c + lane;
                const uint32_t qm = el < cnt ? (uint32_t)s_qm[buf][el] : 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = ((qm >> (qbit0 + (j & 1) + 4 * (j >> 1))) & 1u) != 0u;
                    const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (ok) s_idx[wave][j][qlen[j] + rank] = (uint8_t)el;
                    qlen[j] += (uint32_t)__popcll(m);
                }
            }
            wave_lds_fence();
            const uint32_t my_len = qrow == 0 ? qlen[0] : qrow == 1 ? qlen[1] : qrow == 2 ? qlen[2] : qlen[3];
            const uint32_t n_trip = (uint32_t)__builtin_amdgcn_readfirstlane((int)max(max(qlen[0], qlen[1]), max(qlen[2], qlen[3])));
            const uint8_t* my_idx = &s_idx[wave][qrow][0];
            const uint32_t posb = base - start;
            // One trip = four entries (one per row of 16 lanes): index (1 byte) -> record (40 bytes) -> ~20 dependent VALU
            // instructions.  Read serially that is two LDS round trips in front of every trip's arithmetic; here the index of
            // trip i + 2 and the record of trip i + 1 are requested before trip i's arithmetic starts (LDS returns in order, so
            // the waits the compiler places count exactly what is still outstanding).  Two copies of the body alternate two
            // register sets: nothing is copied.  Indices past a row's list are stale bytes (< 256: a valid slot of the staging
            // area) and are masked by `i < my_len` as before.
            struct Rec { float2 ef; float4 r1, r2; };
            auto fetch = [&](uint32_t e) { Rec r; r.ef = *reinterpret_cast<const float2*>(&s_r0[buf][e]); r.r1 = s_r1[buf][e]; r.r2 = s_r2[buf][e]; return r; };
            auto trip = [&](uint32_t i, uint32_t e, const Rec& rc) -> bool {   // true: every pixel of the wavefront has stopped
                float G0, K;
                pair_alpha_row(Yf, rc.ef.x, rc.ef.y, rc.r1.y, rc.r1.z, rc.r1.w, G0, K);
                const float alpha = fminf(kAlphaMax, pair_alpha_px(Xf, rc.r1.x, G0, K));
                const float test_T = T * (1.0f - alpha);
                const uint64_t hitm = __builtin_amdgcn_ballot_w64(alpha >= kAlphaMin) & __builtin_amdgcn_ballot_w64(i < my_len) & livem;
                const uint64_t stopm = __builtin_amdgcn_ballot_w64(test_T < kTStop) & hitm;
                const uint64_t blendm = hitm ^ stopm;  // stop implies hit
                livem &= ~stopm;
                const float w = mask_select(blendm, alpha * T, 0.0f);
                Cr = fmaf(rc.r2.x, w, Cr); Cg = fmaf(rc.r2.y, w, Cg); Cb = fmaf(rc.r2.z, w, Cb); D = fmaf(rc.r2.w, w, D);
                T = mask_select(blendm, test_T, T);
                if (stopm != 0ull) {   // rare: once per pixel at most
                    last = mask_select(stopm, posb + e, last);
                    return livem == 0ull;
                }
                return false;
            };
            if (n_trip != 0u) {
                uint32_t ea = my_idx[0], eb = my_idx[1];
                Rec ra = fetch(ea);
#pragma unroll 1
                for (uint32_t i = 0;;) {
                    const Rec rb = fetch(eb);                 // trip i + 1
                    const uint32_t ea_n = my_idx[i + 2];
                    if (trip(i, ea, ra)) break;
                    if (++i >= n_trip) break;
                    ra = fetch(ea_n);                         // trip i + 1 (after the increment): i + 2 of the loop top
                    const uint32_t eb_n = my_idx[i + 2];
                    if (trip(i, eb, rb)) break;
                    if (++i >= n_trip) break;
                    ea = ea_n; eb = eb_n;
                }
            }
        }
        if (!more) break;
        if (lane == 0) s_live[k & 1u][wave] = livem != 0ull ? 1u : 0u;
        lds_barrier();   // everybody is done with the batch: the one copy may be overwritten
        if ((s_live[k & 1u][0] | s_live[k & 1u][1] | s_live[k & 1u][2] | s_live[k & 1u][3]) == 0u) break;   // every pixel has stopped
        if (stages(k + 1u)) request(0, id_nxt);
        if (stages(k + 1u)) {
            lds_copy_wait();
            id_nxt = b.sorted_id[min(base + (1u + kStep) * kB + (uint32_t)slot, lastpos)];
            publish_mask(nbuf, base + kB + (uint32_t)slot);
        }
        lds_barrier();
    }
    }
done:
    if (inside) {
        const size_t hw = (size_t)v.H * v.W, pix = (size_t)py * v.W + px;
        out_color[pix] = Cr + T * v.bg[0];
        out_color[hw + pix] = Cg + T * v.bg[1];
        out_color[2 * hw + pix] = Cb + T * v.bg[2];
        out_depth[pix] = D;
        if (out_alpha) out_alpha[pix] = 1.0f - T;
        im.final_T[pix] = T;
        im.n_contrib[pix] = last;
    }
    publish_tile_reach(g, tile, last, /*quad of this lane: */ 2 * (wave & 1) + (qrow & 1), 2 * (wave >> 1) + (qrow >> 1),
                       /*lane bits spanning a quad: */ 1, 2, 4, 8, (lane & 15) == 0);
}

void launch_render_forward(const ViewK& v, const Geom& g, const Binning& b, const Image& im,
                           float* out_color, float* out_depth, float* out_alpha, hipStream_t st) {
    const int tiles = v.gx * v.gy;
    // (Occupancy: eight workgroups per CU.  Capping it with LDS padding was measured in round 4 -- 7 / 6 / 5 per CU: 0.1235 /
    // 0.130 / 0.1525 ms against 0.114 -- the kernel lives on its eight wavefronts per SIMD; the 1.22-round tail is not the cost.)
    if (tiles > 0) hipLaunchKernelGGL(k_render_forward, dim3(tiles), dim3(kBlock), 0, st, v, g, b, im, out_color, out_depth, out_alpha);
}

// ------------------------------------------------------------------------------------------
// Backward
// ------------------------------------------------------------------------------------------
// HAS_D / HAS_A: the caller supplied dL/ddepth / dL/dalpha.  SplatFields' default losses leave the depth gradient
// empty (reference arguments/__init__.py:166,168), so the depth channel's replay and its wave reduction are compiled out.
template <bool HAS_D, bool HAS_A>
__global__ void __launch_bounds__(kBlock) k_render_backward(const ViewK v, const Geom g, const Binning b, const Image im,
                                                            const float* __restrict__ dL_dcolor,
                                                            const float* __restrict__ dL_ddepth,
                                                            const float* __restrict__ dL_dalpha,
                                                            float* __restrict__ slots) {
    // slot kBwdBatch is a sentinel record whose exponent offset is +inf (alpha = 0): the dummy entries of an incomplete
    // group of 4 point at it and contribute nothing without any special-casing in the loop
    __shared__ float4 s_r0[kBwdBatch + 1];
    __shared__ float4 s_r1[kBwdBatch + 1];
    __shared__ float4 s_r2[kBwdBatch + 1];
    __shared__ float4 s_acc[4][kBwdBatch][3];
    __shared__ float2 s_ef[kBwdBatch + 1];   // (E0, F0): the per-(tile, entry) part of the exponent (common.h: exponent_terms);
                                             // p s rides in the unused fourth component of s_r0
    __shared__ uint32_t s_max[4];

    const int tile = (int)g.tile_order[blockIdx.x];  // longest lists first
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int wave = wave_id(), lane = lane_id();
    const int sx = tx * kTile + (wave & 1) * kSub, sy = ty * kTile + (wave >> 1) * kSub;
    const int px = sx + (lane & 7), py = sy + (lane >> 3);
    const bool inside = px < v.W && py < v.H;
    const float pxf = (float)px, pyf = (float)py, sxf = (float)sx, syf = (float)sy;
    const float tx0f = (float)(tx * kTile), ty0f = (float)(ty * kTile);
    const float Xf = (float)(px - tx * kTile), Yf = (float)(py - ty * kTile);
    const uint32_t start = g.tile_start[tile], end = g.tile_start[tile + 1];
    const int n = (int)(end - start);
    const size_t hw = (size_t)v.H * v.W, pix = (size_t)py * v.W + px;

    int my_last = 0;
    float T_final = 0.f, gR = 0.f, gG = 0.f, gB = 0.f, gD = 0.f, gA = 0.f;
    if (inside) {
        my_last = (int)im.n_contrib[pix];
        T_final = im.final_T[pix];
        gR = dL_dcolor[pix]; gG = dL_dcolor[hw + pix]; gB = dL_dcolor[2 * hw + pix];
        if (HAS_D) gD = dL_ddepth[pix];
        if (HAS_A) gA = dL_dalpha[pix];
    }
    const float bg_dot = v.bg[0] * gR + v.bg[1] * gG + v.bg[2] * gB;

    int wmax = my_last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    if (lane == 0) s_max[wave] = (uint32_t)wmax;
    __syncthreads();
    const int bmax = (int)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));

    float4* slot4 = reinterpret_cast<float4*>(slots);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // List entries at positions >= bmax (behind every pixel's stop) receive no gradient and no slot: their `reached` byte
    // stays 0 (cleared by the scatter) and k_preprocess_backward skips them.

    if (threadIdx.x == 0) {
        s_r0[kBwdBatch] = zero4;
        s_r1[kBwdBatch] = make_float4(0.f, 0.f, 0.f, __builtin_inff());
        s_r2[kBwdBatch] = zero4;
        s_ef[kBwdBatch] = make_float2(0.f, 0.f);
    }
    float T = T_final;
    float behind_g = T_final * bg_dot;  // (colour accumulated behind the current splat, incl. background) . upstream gradient

    for (int top = bmax; top > 0; top -= kBwdBatch) {
        const int cnt = min(kBwdBatch, top);
        // instance index (slot of the gradient scratch) of this thread's list entry = the splat's first instance + row-major
        // position of this tile inside the splat's tile rect (which rides in the record's fourth quarter).  The offsets
        // gather is issued last and consumed only after the blend loop, so its latency is never waited for.
        uint32_t inst_v = 0u;
        if ((int)threadIdx.x < cnt) {
            const uint32_t pos = start + (uint32_t)(top - 1 - (int)threadIdx.x);
            const uint32_t id = b.sorted_id[pos];
            const float4* rec = g.rec + 4 * (size_t)id;
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
            const uint32_t inst_first = g.block_offsets[id >> 8] + __float_as_uint(r3.z);   // see load_entry (blend_bwd.hip)
            float E0, F0, ps;
            exponent_terms(r0.x, r0.y, r1, tx0f, ty0f, E0, F0, ps);
            s_r0[threadIdx.x] = make_float4(r0.x, r0.y, r0.z, ps);
            s_r1[threadIdx.x] = r1;
            s_r2[threadIdx.x] = r2;
            s_ef[threadIdx.x] = make_float2(E0, F0);
            const uint32_t xy = __float_as_uint(r3.x), rw = __float_as_uint(r3.y);
            inst_v = inst_first + ((uint32_t)ty - (xy >> 16)) * rw + ((uint32_t)tx - (xy & 0xffffu));
        }
        {
            float4* acc = &s_acc[0][0][0];
            for (int i = threadIdx.x; i < 4 * kBwdBatch * 3; i += kBlock) acc[i] = zero4;
        }
        __syncthreads();
        if (wmax > top - cnt) {
            for (int c = 0; c < cnt; c += kWave) {
                const int el = c + lane;
                const int ec = el < cnt ? el : cnt - 1;
                const bool ok = el < cnt && (top - 1 - el) < wmax && subtile_overlap(s_r0[ec], s_r1[ec], sxf, syf);
                uint64_t m = __builtin_amdgcn_ballot_w64(ok);
                // 4 list entries per step (taken from the scalar bit mask): their 4 x 10 per-lane partial sums
                // are reduced together by a butterfly (v_permlane32_swap, v_permlane16_swap, then 4 DPP steps
                // inside each row of 16 lanes): 100 cross-lane adds per 4 entries instead of 4 x 60.
                while (m) {
                    float pv[4][10];
                    int ent[4];
                    uint64_t any = 0ull;  // wave-level: kept as a scalar mask, no per-lane flag
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // fewer than 4 entries left in the mask: the sentinel record (contributes nothing, stores nothing)
                        const int e = m != 0ull ? c + (int)__builtin_ctzll(m) : kBwdBatch;
                        m &= (m - 1);  // stays 0 once empty
                        ent[q] = e;
                        const float4 r0 = s_r0[e], r1 = s_r1[e], r2 = s_r2[e];
                        const float dx = r0.x - pxf, dy = r0.y - pyf;
                        const float2 ef = s_ef[e];
                        float G0, K;
                        pair_alpha_row(Yf, ef.x, ef.y, r0.w, r1.z, r1.w, G0, K);
                        const float oG = pair_alpha_px(Xf, r1.x, G0, K);   // opacity * G, the forward's expression
                        // Who contributes is decided on the scalar unit (oG >= 1/255 is the forward's alpha >= 1/255: the
                        // 0.99 clamp does not move that threshold).  Everyone else gets alpha = G = 0, which makes every
                        // product below 0 and leaves T and behind_g unchanged (1/(1-0) = 1 exactly): no divergent branch, so
                        // the four entries of a step form one basic block whose LDS reads and arithmetic interleave freely.
                        const uint64_t cm = __builtin_amdgcn_ballot_w64(oG >= kAlphaMin) & __builtin_amdgcn_ballot_w64((top - 1 - e) < my_last);
                        any |= cm;
                        const float oGc = mask_select(cm, oG, 0.0f);
                        const float alpha = __builtin_amdgcn_fmed3f(oGc, 0.0f, kAlphaMax);   // min(0.99, o G), o G >= 0
                        const float inv_keep = __builtin_amdgcn_rcpf(1.0f - alpha);  // v_rcp_f32, 1 ulp
                        T = T * inv_keep;   // transmittance in front of this splat
                        const float wgt = alpha * T;
                        // dL/dalpha_i = T_i (c_i . g) - (sum_{j behind i} w_j (c_j . g) + T_final (bg . g)) / (1 - alpha_i).
                        // The upstream gradient g is constant along the list, so the "colour behind" enters only through
                        // its dot product with g: ONE scalar recurrence (behind_g) instead of one per channel.
                        float cg = HAS_A ? gA : 0.0f;  // the alpha channel's "colour" is 1 for every splat
                        if (HAS_D) cg = fmaf(r2.w, gD, cg);
                        cg = fmaf(r2.z, gB, cg); cg = fmaf(r2.y, gG, cg); cg = fmaf(r2.x, gR, cg);
                        const float dLa = T * cg - inv_keep * behind_g;
                        behind_g = fmaf(wgt, cg, behind_g);
                        // opacity * G * dL/dalpha: the six geometric sums carry the factor `opacity` (k_preprocess_backward
                        // divides it out of dL/dopacity instead of multiplying it into the other five);
                        // gradients pass through the min(0.99, .) clamp, as upstream
                        const float g1 = oGc * dLa;
                        const float sxv = g1 * dx, syv = g1 * dy;
                        pv[q][0] = g1; pv[q][1] = sxv; pv[q][2] = syv; pv[q][3] = sxv * dx; pv[q][4] = sxv * dy; pv[q][5] = syv * dy;
                        pv[q][6] = wgt * gR; pv[q][7] = wgt * gG; pv[q][8] = wgt * gB; pv[q][9] = HAS_D ? wgt * gD : 0.f;
                    }
                    if (any == 0ull) continue;
                    float red[10];
                    red[9] = 0.f;
#pragma unroll
                    for (int k = 0; k < (HAS_D ? 10 : 9); ++k) {
                        const float z01 = swap32_add(pv[0][k], pv[1][k]);   // lanes 0-31: entry 0, lanes 32-63: entry 1
                        const float z23 = swap32_add(pv[2][k], pv[3][k]);   // lanes 0-31: entry 2, lanes 32-63: entry 3
                        float w = swap16_add(z01, z23);                     // rows: 0 -> entry 0, 1 -> entry 2, 2 -> entry 1, 3 -> entry 3
                        w = dpp_add<0xB1>(w); w = dpp_add<0x4E>(w); w = dpp_add<0x124>(w); w = dpp_add<0x128>(w);
                        // keep the last add next to its DPP move (one fused v_add_f32_dpp) instead of letting it sink into the
                        // lane-0-of-each-row store below as v_mov 0 + v_mov_dpp + v_add
                        asm volatile("" : "+v"(w));
                        red[k] = w;
                    }
                    const int row = lane >> 4;
                    const int q_of_row = ((row & 1) << 1) | (row >> 1);
                    const int e_mine = q_of_row == 0 ? ent[0] : (q_of_row == 1 ? ent[1] : (q_of_row == 2 ? ent[2] : ent[3]));
                    if ((lane & 15) == 0 && e_mine < kBwdBatch) {
                        s_acc[wave][e_mine][0] = make_float4(red[0], red[1], red[2], red[3]);
                        s_acc[wave][e_mine][1] = make_float4(red[4], red[5], red[6], red[7]);
                        s_acc[wave][e_mine][2] = make_float4(red[8], red[9], 0.f, 0.f);
                    }
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const int e = threadIdx.x;
            const size_t inst = (size_t)inst_v;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 a0 = s_acc[0][e][q], a1 = s_acc[1][e][q], a2 = s_acc[2][e][q], a3 = s_acc[3][e][q];
                slot4[inst * kSlotF4 + q] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                                  (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
            }
            b.reached[inst] = 1;
        }
        __syncthreads();
    }
}

void launch_render_backward(const ViewK& v, const Geom& g, const Binning& b, const Image& im,
                            const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                            float* slots, hipStream_t st) {
    const int tiles = v.gx * v.gy;
    if (tiles <= 0) return;
    const bool d = dL_ddepth != nullptr, a = dL_dalpha != nullptr;
    if (d && a) hipLaunchKernelGGL((k_render_backward<true, true>), dim3(tiles), dim3(kBlock), 0, st, v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots);
    else if (d) hipLaunchKernelGGL((k_render_backward<true, false>), dim3(tiles), dim3(kBlock), 0, st, v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots);
    else if (a) hipLaunchKernelGGL((k_render_backward<false, true>), dim3(tiles), dim3(kBlock), 0, st, v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots);
    else hipLaunchKernelGGL((k_render_backward<false, false>), dim3(tiles), dim3(kBlock), 0, st, v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots);
}

}  // namespace sr
