// Backward of the alpha compositing for gfx950 -- "entry-per-lane" formulation over 4x4-pixel quads.
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
//     shifts, carry between buckets in lane 15) -- no per-pixel serial loop, no early-exit divergence.  The four columns are
//     computed stage by stage side by side (replay_bucket): four independent chains fill each other's dependent-issue gaps.
//   * the ten sums over pixels: seven per-lane accumulators over the four columns (moments of g1 in X, colour / depth sums of
//     the weights), the row's Y applied per lane, then eight swap-adds (v_permlane32_swap / v_permlane16_swap) over the four
//     rows.  Moments are taken about the TILE centre and shifted to the splat centre once per (tile, entry) when the quads'
//     partial sums are combined.  (Rounds 2-3 contracted them with two v_mfma_f32_16x16x4_f32 per column.  Round 4 measured
//     -- tools/mfma_valu_coissue.hip -- that on gfx950 the fp32 MFMA and the SIMD's VALU instructions do not overlap: their
//     times ADD, an fp32 MFMA is 32 cycles of the vector ALUs; eight per bucket cost more than the VALU form of the sums.)
//   * the quads' entry lists are built per chunk of the tile list from the quad-reach masks the forward left per list entry
//     (exact-support test, quadmask.h), bucketed with ballots + prefix counts; a slot of the LDS scratch first carries the
//     entry's record to the wavefront that owns the quad and then carries the 10 sums back.  Everything is summed in a fixed
//     order: bit-reproducible, no floating-point atomics (as before).
#include "kernels.h"

namespace sr {

namespace {

constexpr int kBucket = 16;                 // list entries per bucket = N of the MFMA
constexpr int kBlkEntries = 32;             // granularity of a chunk of the tile list ("block" = half a wavefront's entries)
constexpr int kBlocks = 8;                  // a chunk = up to 8 blocks = one list entry per thread
// (quad, entry) slots per pass over a chunk.  A slot holds 10 floats (12 when the caller supplies a depth gradient, so that
// the record's depth rides along).  LDS per workgroup: slots + 10 KB of tables = 40.7 KB, and 128 registers per lane: FOUR
// workgroups per CU, one wavefront of each per SIMD.
// (History: with the MFMA replay of round 3 four per CU -- registers capped at 128 -- spilled and was slower, 0.290 vs 0.277
// ms, so rounds 3-4 ran three per CU with 53 KB of slots.  The stage-major replay of round 4 fits 128 registers without a
// spill; measured then, same box: 0.1962 vs 0.2094 ms at the headline for four vs three per CU, dense workloads equal.  The
// kernel issues VALU instructions in 39 % of its cycles -- the fourth wavefront per SIMD fills some of the rest.  Five per CU
// would need 96 registers: 35 spilled.)
#ifndef SR_BWD_CAP
#define SR_BWD_CAP 752
#endif
#ifndef SR_BWD_CAP_DEPTH
#define SR_BWD_CAP_DEPTH 624
#endif
// slots a pass may assign (40-byte / 48-byte slots: 40.7 KB of LDS per workgroup with the tables); a quad's run can
// need 256, and reads run up to 15 slots past a run
template <bool HAS_D> struct SlotCap { static constexpr int kCap = HAS_D ? SR_BWD_CAP_DEPTH : SR_BWD_CAP; };
constexpr int kCapSlack = 16;
static_assert(SR_BWD_CAP >= 256 && SR_BWD_CAP_DEPTH >= 256, "one quad's entries of a chunk must always fit");
#ifndef SR_BWD_WAVES_PER_SIMD
#define SR_BWD_WAVES_PER_SIMD 4   // register budget: 512 / 4 = 128 per lane
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef SR_BWD_STATS
// diagnostic build only (bench.py's pairs_evaluated / pairs_blended figures): [0] list entries tested, [1] (quad, entry)
// pairs, [2] buckets, [3] (pixel, entry) pairs evaluated, [4] pairs blended (alpha >= 1/255 and not behind the pixel's
// last contributor), [5] chunks
__device__ unsigned long long g_bwd_stats[16];   // [8..15]: shader-clock cycles per phase, summed over wavefronts
#define SR_STAT_ADD(i, x) atomicAdd(&g_bwd_stats[i], (unsigned long long)(x))
#define SR_PHASE(i) do { const long long now_ = clock64(); ph_[i] += now_ - t_; t_ = now_; } while (0)
#else
#define SR_STAT_ADD(i, x) do {} while (0)
#define SR_PHASE(i) do {} while (0)
#endif

// DPP controls: row_shr:n = 0x110 + n (lane l reads lane l - n of its row of 16), row_shl:n = 0x100 + n, row_ror:n = 0x120 + n
constexpr int kShr1 = 0x111, kShr2 = 0x112, kShr4 = 0x114, kShr8 = 0x118;

// value of lane (l - shift) of the row; lanes without a source (and nothing else) keep `old`
template <int CTRL> __device__ __forceinline__ float dpp_f(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, 0xf, false);
}

// Row scans in both directions.  UP: lane l accumulates lanes <= l (row_shr), the carry of the previous bucket sits in
// lane 0's `old` slot; DOWN: lane l accumulates lanes >= l (row_shl), carry in lane 15.  Buckets alternate direction, so
// the total a bucket leaves in its last lane is already where the next bucket's first entry picks it up: no lane
// broadcast, no rotate.
constexpr int kShl1 = 0x101, kShl2 = 0x102, kShl4 = 0x104, kShl8 = 0x108;
// inclusive prefix over each row of 16 lanes (Kogge-Stone, 4 DPP steps)
template <bool UP> __device__ __forceinline__ float row_scan_add(float x) {
    if (UP) { x += dpp_f<kShr1>(0.f, x); x += dpp_f<kShr2>(0.f, x); x += dpp_f<kShr4>(0.f, x); x += dpp_f<kShr8>(0.f, x); }
    else { x += dpp_f<kShl1>(0.f, x); x += dpp_f<kShl2>(0.f, x); x += dpp_f<kShl4>(0.f, x); x += dpp_f<kShl8>(0.f, x); }
    return x;
}
// The product scan needs "lanes without a source keep their own value", i.e. v_mul_f32_dpp with the destination as
// `old` operand; the compiler only fuses DPP moves whose fill value is 0 (it emits v_mov 1.0 + v_mov_dpp + v_mul, three
// issue slots per level), so the four levels are written out.  s_nop 1 = the two wait states a DPP read needs after the
// VALU write of its source (the hazard recogniser does not look inside inline assembly).
template <bool UP> __device__ __forceinline__ float row_scan_mul(float x) {
    if (UP) {
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    } else {
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    }
    return x;
}
// the neighbour's value; the first lane of the scan direction (no source) keeps its own `carry`
template <bool UP> __device__ __forceinline__ float shift_in_carry(float carry, float v) {
    return UP ? dpp_f<kShr1>(carry, v) : dpp_f<kShl1>(carry, v);
}

// x, y hold two different quantities per lane.  swap32: r[l] = x[l] + x[l + 32] for l < 32, y[l - 32] + y[l] above (one
// v_permlane32_swap + one add: each half-wave now owns one quantity); swap16 the same one level down (rows 0, 2 own x, rows
// 1, 3 own y).  Two levels sum FOUR per-lane quantities over the four rows of a wavefront with three swap-adds:
//   w = swap16(swap32(a, b), swap32(c, d)):   row 0: sum a,  row 1: sum c,  row 2: sum b,  row 3: sum d     (per lane n = lane & 15)
typedef unsigned int uint2q __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float swap32_sum(float x, float y) {
    const uint2q r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float swap16_sum(float x, float y) {
    const uint2q r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__device__ __forceinline__ uint32_t row_allsum_u32(uint32_t v) {   // every lane of a row: the row's sum
    v += dpp_u<0xB1>(0u, v); v += dpp_u<0x4E>(0u, v); v += dpp_u<0x124>(0u, v); v += dpp_u<0x128>(0u, v);
    return v;
}
__device__ __forceinline__ uint32_t row_scan_add_u32(uint32_t x) {
    x += dpp_u<kShr1>(0u, x); x += dpp_u<kShr2>(0u, x); x += dpp_u<kShr4>(0u, x); x += dpp_u<kShr8>(0u, x);
    return x;
}

}  // namespace

// One list entry as the build phase holds it (one per thread and chunk).
struct BwdEntry {
    int pos;            // list position (descending with the thread index), < 0: none
    float2 c;           // splat centre in pixels
    float2 ef;          // (E0, F0): the per-(tile, entry) part of the exponent (common.h: exponent_terms); set by finish_entry
    float4 r1, r2;      // record quarters (p, s -> p s after finish_entry, q, -log2 o) (r, g, b, depth)
    uint32_t qm;        // quad-reach mask the forward computed for this (tile, entry)
    uint32_t inst;      // instance index of the (splat, tile) pair: slot of the gradient scratch (after finish_entry)
    uint32_t rect_xy, rect_w, first_rel;   // raw record quarter 3 until finish_entry
};

// Positions are clamped to 0 so that every load is unconditional (a thread without an entry loads entry 0 and drops it
// through pos < 0): no branch, no merge with zeros behind the loads.
// The gather is split in two so that nothing WAITS where it is issued: load_entry only requests -- the splat index `id` comes
// from a load issued a whole chunk earlier (the record address depends on it: requested here, it was a full memory round trip
// of ~1-2 us per chunk in which the wavefront did nothing) -- and finish_entry does the arithmetic on the loaded values where
// they are first needed (the scatter, two phases later).
__device__ __forceinline__ BwdEntry load_entry(const Geom& g, uint32_t id, const uint32_t* qms, int pos) {
    BwdEntry e;
    const int pc = pos > 0 ? pos : 0;
    const float4* rec = g.rec + 4 * (size_t)id;
    e.pos = pos;
    e.c = *reinterpret_cast<const float2*>(rec);
    e.r1 = rec[1]; e.r2 = rec[2];
    const float4 r3 = rec[3];   // (tile rect origin, rect width, first instance relative to the splat's 256-splat sub-batch)
    e.rect_xy = __float_as_uint(r3.x); e.rect_w = __float_as_uint(r3.y); e.first_rel = __float_as_uint(r3.z);
    // block_offsets is a 4 B x N / 256 table (16 KB at 1 M splats: cache resident); the per-splat `offsets` array would cost a
    // line of memory traffic per list entry for 4 useful bytes
    e.inst = g.block_offsets[id >> 8];
    e.ef = make_float2(0.f, 0.f);
    e.qm = qms[pc];
    return e;
}
__device__ __forceinline__ void finish_entry(BwdEntry& e, uint32_t tx, uint32_t ty) {
    // what the forward's staging thread did with the same record (render.hip): same function, same inputs, same bits
    float E0, F0, ps;
    exponent_terms(e.c.x, e.c.y, e.r1, (float)(tx * kTile), (float)(ty * kTile), E0, F0, ps);
    e.ef = make_float2(E0, F0);
    e.r1.y = ps;
    // the instance index = the splat's first instance + the tile's row-major position in its rect
    e.inst += e.first_rel + (ty - (e.rect_xy >> 16)) * e.rect_w + (tx - (e.rect_xy & 0xffffu));
}

// ---- the LDS slot of one (quad, entry) pair -------------------------------------------------------------------------
// It first carries the entry's record to the wavefront that replays the quad, then the ten sums back to the entry's thread:
//   in : [0] E0  [1] F0  [2] list position (int bits)  [3] p | [4] p s  [5] q  [6] -log2 o  [7] r | [8] g  [9] b | ([10] depth)
//   out: [0..3] M0 MX MY MXX | [4..7] MXY MYY dr dg | [8] db  [9] d(depth)
// 10 floats (8-byte aligned, moved as 64-bit pieces) unless the depth gradient is live: then 12 floats, 128-bit pieces.
template <bool HAS_D> struct SlotFmt { static constexpr int kF = HAS_D ? 12 : 10; };

struct SlotIn { float E0, F0, p, ps, q, nlo, r, g, b, depth; int pos; };

template <bool HAS_D>
__device__ __forceinline__ void slot_put_record(float* sl, const BwdEntry& e) {
    if constexpr (HAS_D) {
        float4* d = reinterpret_cast<float4*>(sl);
        d[0] = make_float4(e.ef.x, e.ef.y, __int_as_float(e.pos), e.r1.x);
        d[1] = make_float4(e.r1.y, e.r1.z, e.r1.w, e.r2.x);
        d[2] = make_float4(e.r2.y, e.r2.z, e.r2.w, 0.f);
    } else {
        float2* d = reinterpret_cast<float2*>(sl);
        d[0] = make_float2(e.ef.x, e.ef.y);
        d[1] = make_float2(__int_as_float(e.pos), e.r1.x);
        d[2] = make_float2(e.r1.y, e.r1.z);
        d[3] = make_float2(e.r1.w, e.r2.x);
        d[4] = make_float2(e.r2.y, e.r2.z);
    }
}
template <bool HAS_D>
__device__ __forceinline__ SlotIn slot_get_record(const float* sl) {
    SlotIn r;
    if constexpr (HAS_D) {
        const float4* d = reinterpret_cast<const float4*>(sl);
        const float4 a = d[0], b = d[1], c = d[2];
        r.E0 = a.x; r.F0 = a.y; r.pos = __float_as_int(a.z); r.p = a.w;
        r.ps = b.x; r.q = b.y; r.nlo = b.z; r.r = b.w;
        r.g = c.x; r.b = c.y; r.depth = c.z;
    } else {
        const float2* d = reinterpret_cast<const float2*>(sl);
        const float2 a = d[0], b = d[1], c = d[2], e = d[3], f = d[4];
        r.E0 = a.x; r.F0 = a.y; r.pos = __float_as_int(b.x); r.p = b.y;
        r.ps = c.x; r.q = c.y; r.nlo = e.x; r.r = e.y;
        r.g = f.x; r.b = f.y; r.depth = 0.f;
    }
    return r;
}
// A lane without an entry (it sees the next run's slot, or the slack behind the buffer): its exponent offset becomes +inf, so
// opacity * G is 0 or NaN whatever the other exponent terms hold, the `hit` select turns either into alpha = G = 0, and the
// lane is transparent to the scans; the colour (and depth) must be finite, because 0 * inf would poison the sum scan.
// (Five selects per bucket instead of the eleven that cleaned every field: 27 of a bucket's ~750 VALU cycles.)
__device__ __forceinline__ void slot_mask_invalid(SlotIn& r, bool valid) {
    r.nlo = valid ? r.nlo : __builtin_inff();
    r.r = valid ? r.r : 0.f; r.g = valid ? r.g : 0.f; r.b = valid ? r.b : 0.f; r.depth = valid ? r.depth : 0.f;
}
// The ten sums of a (quad, entry) pair leave the replay spread over the four rows of the wavefront (replay_bucket: three
// registers per lane), and are stored where their row finds them with one 8-byte and at most one 4-byte store:
//   floats  [0] M0  [1] MXY  [2] db  |  [3] MX  [4] MYY  [5] d(depth)  |  [6] MY  [7] dr  |  [8] MXX  [9] dg
//            row 0 (w0, w1, w2)         row 2 (w0, w1, w2)                 row 1 (w0, w1)     row 3 (w0, w1)
struct BucketSums { float w0, w1, w2; };
__device__ __forceinline__ void slot_put_sums(float* sl, int row, const BucketSums d) {
    const int o64 = row == 0 ? 0 : row == 1 ? 6 : row == 2 ? 4 : 8;
    *reinterpret_cast<float2*>(sl + o64) = row == 2 ? make_float2(d.w1, d.w2) : make_float2(d.w0, d.w1);
    if ((row & 1) == 0) sl[row == 0 ? 2 : 3] = row == 0 ? d.w2 : d.w0;
}
// -> a = (M0, MX, MY, MXX), b = (MXY, MYY, dr, dg), c = (db, d depth)
template <bool HAS_D>
__device__ __forceinline__ void slot_get_sums(const float* sl, float4& a, float4& b, float2& c) {
    if constexpr (HAS_D) {
        const float4* d = reinterpret_cast<const float4*>(sl);
        const float4 x = d[0], y = d[1], z = d[2];
        a = make_float4(x.x, x.w, y.z, z.x); b = make_float4(x.y, y.x, y.w, z.y); c = make_float2(x.z, y.y);
    } else {
        const float2* d = reinterpret_cast<const float2*>(sl);
        const float2 x0 = d[0], x1 = d[1], x2 = d[2], x3 = d[3], x4 = d[4];
        a = make_float4(x0.x, x1.y, x3.x, x4.x); b = make_float4(x0.y, x2.x, x3.y, x4.y); c = make_float2(x1.x, x2.y);
    }
}

// tile-relative column x = 0..15: (x, x - 7.5, (x - 7.5)^2) -- read with scalar loads by the (wave-uniform) quad column
__constant__ float kColumn[3][16] = {
    {0.f, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f, 9.f, 10.f, 11.f, 12.f, 13.f, 14.f, 15.f},
    {-7.5f, -6.5f, -5.5f, -4.5f, -3.5f, -2.5f, -1.5f, -0.5f, 0.5f, 1.5f, 2.5f, 3.5f, 4.5f, 5.5f, 6.5f, 7.5f},
    {56.25f, 42.25f, 30.25f, 20.25f, 12.25f, 6.25f, 2.25f, 0.25f, 0.25f, 2.25f, 6.25f, 12.25f, 20.25f, 30.25f, 42.25f, 56.25f}};

// Per-quad constants of the replay (pixel row k of the quad, pixel columns t = 0..3).
struct QuadCtx {
    float gR[4], gG[4], gB[4], gD[4], gA[4];
    int last[4];
    float Y;                    // tile-relative row of this lane's pixels
    float Xa[4], Xc[4], Xc2[4];   // per column, wave-uniform (scalar registers): tile-relative X, X about the tile centre, its square
};

// One bucket: 16 entries (lanes n) x 4 pixel rows (k) x 4 pixel columns (steps).  ST / SB carry the transmittance and
// the "colour behind . g" of every pixel from bucket to bucket: an UP bucket (entries back to front in lanes 0..15) takes
// them from lane 0 and leaves them in lane 15, a DOWN bucket (entries in lanes 15..0) the other way round.
//
// Written STAGE-major, not column-major: every stage is applied to the four pixel columns before the next stage starts.  One
// column's chain is ~35 DEPENDENT instructions (exponent -> v_exp -> select -> v_rcp -> a 5-deep DPP product scan -> a 5-deep
// DPP sum scan -> MFMA), and a wavefront issues in order: written column by column (round 3) the compiler kept that order,
// the chain's latency (~8 cycles per dependent VALU instruction, two wait states in front of every DPP read of a fresh
// result) was exposed on every instruction, and the replay ran at the speed of its dependency chain, not of the VALU
// (measured: 16 % fewer instructions gave no time back).  Four independent chains side by side fill those gaps.
// The product scan is inline assembly (the compiler only fuses DPP moves whose fill value is 0): its four columns are
// interleaved inside ONE statement, so consecutive levels of a column are three instructions apart -- the two wait states a
// DPP read needs after the VALU write of its source come for free, one s_nop per statement covers the statement's first read.
#define SR_DPP_MUL4(ctrl) \
    "v_mul_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mul_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mul_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_mul_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
template <bool UP> __device__ __forceinline__ void row_scan_mul4(float (&x)[4]) {
    if (UP) asm("s_nop 1\n\t" SR_DPP_MUL4("row_shr:1") SR_DPP_MUL4("row_shr:2") SR_DPP_MUL4("row_shr:4") SR_DPP_MUL4("row_shr:8")
                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    else asm("s_nop 1\n\t" SR_DPP_MUL4("row_shl:1") SR_DPP_MUL4("row_shl:2") SR_DPP_MUL4("row_shl:4") SR_DPP_MUL4("row_shl:8")
             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}
#define SR_DPP_ADD4(ctrl) \
    "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// the sum scan the same way (left to the compiler, its scheduler puts the four columns' scans one after the other again to
// save registers, with wait states between the levels)
template <bool UP> __device__ __forceinline__ void row_scan_add4(float (&x)[4]) {
    if (UP) asm("s_nop 1\n\t" SR_DPP_ADD4("row_shr:1") SR_DPP_ADD4("row_shr:2") SR_DPP_ADD4("row_shr:4") SR_DPP_ADD4("row_shr:8")
                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    else asm("s_nop 1\n\t" SR_DPP_ADD4("row_shl:1") SR_DPP_ADD4("row_shl:2") SR_DPP_ADD4("row_shl:4") SR_DPP_ADD4("row_shl:8")
             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}

template <bool UP, bool HAS_D>
__device__ __forceinline__ BucketSums replay_bucket(const QuadCtx& c, const SlotIn& e, float ST[4], float SB[4], int lane) {
    float oG[4], oGc[4], alpha[4], ginv[4], x[4], T[4], wgt[4], cgv[4], z[4], y[4];
    // opacity * G of the four pixels of this row: the forward's expression (common.h: pair_alpha_row / pair_alpha_px), so both
    // passes make the same alpha >= 1/255 decisions; the row part is shared by the four columns
    float G0, K;
    pair_alpha_row(c.Y, e.E0, e.F0, e.ps, e.q, e.nlo, G0, K);
#pragma unroll
    for (int t = 0; t < 4; ++t) oG[t] = pair_alpha_px(c.Xa[t], e.p, G0, K);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool hit = (oG[t] >= kAlphaMin) && (e.pos < c.last[t]);
        oGc[t] = hit ? oG[t] : 0.0f;                     // everyone else: alpha = G = 0, transparent to the scans
#ifdef SR_BWD_STATS
        { const int nh = __popcll(__builtin_amdgcn_ballot_w64(hit)); if (lane == 0) SR_STAT_ADD(4, nh); }
#endif
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) alpha[t] = fminf(oGc[t], kAlphaMax);   // oGc >= 0
#pragma unroll
    for (int t = 0; t < 4; ++t) ginv[t] = __builtin_amdgcn_rcpf(1.0f - alpha[t]);
    // (colour . upstream gradient) of every pixel: independent of the chain above, fills its gaps
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float v = c.gA[t];   // the alpha channel's "colour" is 1 for every splat
        if (HAS_D) v = fmaf(e.depth, c.gD[t], v);
        v = fmaf(e.b, c.gB[t], v); v = fmaf(e.g, c.gG[t], v); v = fmaf(e.r, c.gR[t], v);
        cgv[t] = v;
    }
    // transmittance in front of entry n: T_n = (T behind the bucket) * prod_{j <= n} 1 / (1 - alpha_j)
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = shift_in_carry<UP>(ST[t], ginv[t]);
    row_scan_mul4<UP>(x);
#pragma unroll
    for (int t = 0; t < 4; ++t) { T[t] = x[t] * ginv[t]; ST[t] = T[t]; }   // last lane of the scan: behind the next bucket
#pragma unroll
    for (int t = 0; t < 4; ++t) { wgt[t] = alpha[t] * T[t]; z[t] = wgt[t] * cgv[t]; }
    // (colour accumulated behind entry n, incl. background) . upstream gradient
#pragma unroll
    for (int t = 0; t < 4; ++t) y[t] = shift_in_carry<UP>(SB[t], z[t]);
    row_scan_add4<UP>(y);
    // The ten sums over the quad's pixels.  Per lane (one pixel row, four columns): moments of g1 in X about the tile centre
    // and the colour / depth sums of the weights; the row's Y is a per-lane constant applied afterwards; then the four rows of
    // the wavefront are added with eight swap-adds (swap32_sum / swap16_sum).
    // (Rounds 2-3 did this with two v_mfma_f32_16x16x4_f32 per column -- "the reduction on the matrix cores".  Measured in
    // round 4 (tools/mfma_valu_coissue.hip): on gfx950 the fp32 MFMA does NOT run beside the SIMD's VALU instructions, whichever
    // wavefront issues them -- 16 v_fma + 2 MFMA per group take 445 cycles at four wavefronts per SIMD, 154 + 257 apart: it
    // occupies the vector ALUs for its 32 cycles.  Eight of them per bucket were 256 of the bucket's ~880 cycles for 640 useful
    // multiply-adds per lane-row; the same sums cost ~31 VALU instructions + 8 swap-adds, ~170 cycles.)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, br = 0.f, bg = 0.f, bb = 0.f, bd = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float behind = y[t];
        SB[t] = behind + z[t];
        // dL/dalpha_n = T_n (c_n . g) - behind_n / (1 - alpha_n); gradients pass through the 0.99 clamp, as upstream
        const float dLa = T[t] * cgv[t] - ginv[t] * behind;
        const float g1 = oGc[t] * dLa;   // the six geometric sums carry the factor `opacity` (k_preprocess_backward)
        a0 += g1; a1 = fmaf(g1, c.Xc[t], a1); a2 = fmaf(g1, c.Xc2[t], a2);
        br = fmaf(wgt[t], c.gR[t], br); bg = fmaf(wgt[t], c.gG[t], bg); bb = fmaf(wgt[t], c.gB[t], bb);
        if (HAS_D) bd = fmaf(wgt[t], c.gD[t], bd);
    }
    const float Yc = c.Y - 7.5f;
    const float mY = Yc * a0, mXY = Yc * a1, mYY = Yc * mY;
    BucketSums r;
    r.w0 = swap16_sum(swap32_sum(a0, a1), swap32_sum(mY, a2));     // rows 0..3: M0, MY, MX, MXX
    r.w1 = swap16_sum(swap32_sum(mXY, mYY), swap32_sum(br, bg));   //            MXY, dr, MYY, dg
    const float zz = swap32_sum(bb, bd);
    r.w2 = swap16_sum(zz, zz);                                      //            db, -, d(depth), -
    (void)lane;
    return r;
}

// HAS_D: the caller supplied dL/ddepth.  SplatFields' default losses leave the depth gradient empty (reference
// arguments/__init__.py:166,168): the depth channel is then compiled out of the replay and of the LDS slots.
template <bool HAS_D>
__global__ void __launch_bounds__(kBlock, SR_BWD_WAVES_PER_SIMD)
k_render_backward_quads(const ViewK v, const Geom g, const Binning b, const Image im, const float* __restrict__ dL_dcolor,
                       const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha, float* __restrict__ slots) {
    constexpr int kF = SlotFmt<HAS_D>::kF;
    constexpr int kCap = SlotCap<HAS_D>::kCap;
    __shared__ __attribute__((aligned(16))) float s_slot[(kCap + kCapSlack) * kF];
    __shared__ float4 s_pixA[256];                   // per pixel of the tile: dL/d(r, g, b, depth)
    __shared__ float4 s_pixB[256];                   // (dL/dalpha, last contributor + 1 [int bits], T state, "behind . g" state)
    // the small tables are double-buffered by chunk parity: a wavefront may start testing the next chunk while others
    // still combine the previous one
    __shared__ uint32_t s_mask[2][16][kBlocks];      // [quad][block]: entries of the block that reach the quad
    __shared__ uint16_t s_bb[2][16][kBlocks];        // slot of the first entry of (quad, block)
    __shared__ uint32_t s_qlen[2][16], s_qbase[2][16];
    __shared__ uint32_t s_ticket[2];                 // next quad to replay in this chunk (wavefronts take quads as they get free)

    const int tile = (int)g.tile_order[blockIdx.x];  // longest lists first
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int wave = wave_id(), lane = lane_id();
    const uint32_t start = g.tile_start[tile];
    const float tx0f = (float)(tx * kTile), ty0f = (float)(ty * kTile);
#ifdef SR_BWD_STATS
    long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_ = clock64();
#endif

    const int tid = (int)threadIdx.x;
    const uint32_t* ids = b.sorted_id + start;

    // How far the forward got, per 4x4 quad (Geom::tile_qlast, written at the end of the forward blend): list entries at
    // positions >= qlast[q] are behind the stop of every pixel of quad q.  The tile is uniform, so these are scalar loads;
    // nothing of the per-pixel data is needed to start fetching the list.
    uint32_t qlast[16];
    {
        const uint32_t* ql = g.tile_qlast + 16 * (size_t)tile;
#pragma unroll
        for (int q = 0; q < 16; ++q) qlast[q] = ql[q];
    }
    uint32_t bmax_u = 0u, qlast_min = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 16; ++q) { bmax_u = max(bmax_u, qlast[q]); qlast_min = min(qlast_min, qlast[q]); }
    const int bmax = __builtin_amdgcn_readfirstlane((int)bmax_u);   // uniform by construction; tell the compiler
    qlast_min = (uint32_t)__builtin_amdgcn_readfirstlane((int)qlast_min);

    // The list is replayed back to front in chunks of kChunk = one entry per thread; a chunk is always the next kChunk entries
    // (when its (quad, entry) pairs exceed the slot buffer, the quads are served in several passes instead of shortening it).
    // A chunk's entries are loaded at the end of the previous chunk, synchronously: measured on MI355X, every form of
    // prefetch tried here -- a second register set swapped by unrolling the loop twice, LDS-direct loads into a staging area
    // (0.337 ms: seven small copies per entry saturate the CU's copy path) -- was no faster than this (0.273 ms); the other
    // workgroups of the CU (three then, four since round 4) already cover a chunk's gather with their replay.
    constexpr int kChunk = kBlocks * kBlkEntries;
    static_assert(kChunk == kBlock, "one list entry per thread");
    if (bmax == 0) return;   // uniform: empty list, or no pixel of the tile blended anything
    int hi = bmax;   // list entries [0, hi) are still to be replayed (back to front)
    const uint32_t* qms = b.qmask + start;
    // (the per-pixel inputs are requested BEHIND the record gather: in front of it -- one memory round trip less in a tile's
    // preamble on paper -- they delay the gather the first chunk waits for: 0.1977 vs 0.1933 ms, round 5)
    const uint32_t id_first = ids[max(hi - 1 - tid, 0)];
    uint32_t id_next = ids[max(hi - kChunk - 1 - tid, 0)];   // its splat index: loaded another chunk earlier
    BwdEntry cur = load_entry(g, id_first, qms, hi - 1 - tid);

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
            if (HAS_D) gD = dL_ddepth[pix];
            if (dL_dalpha) gA = dL_dalpha[pix];
        }
        const float bg_dot = v.bg[0] * gR + v.bg[1] * gG + v.bg[2] * gB;
        s_pixA[threadIdx.x] = make_float4(gR, gG, gB, gD);
        s_pixB[threadIdx.x] = make_float4(gA, __uint_as_float(my_last), T_final, T_final * bg_dot);
        if (threadIdx.x < 2) s_ticket[threadIdx.x] = 0u;
        // no barrier here: the first readers of these tables (replay) sit behind the two barriers of the first chunk
    }

    const int k = lane >> 4, nl = lane & 15;
    const uint32_t lt_mask = (1u << (lane & 31)) - 1u;   // earlier entries of this thread's block
    const int myblk = (int)threadIdx.x >> 5;

    float4* slot4 = reinterpret_cast<float4*>(slots);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // List entries at positions >= bmax (behind every pixel's stop) receive no gradient and no slot: their `reached` byte
    // stays 0 (cleared by the scatter) and k_preprocess_backward skips them.

    int par = 0;   // chunk parity: which copy of the small tables
    SR_PHASE(0);   // preamble

    while (hi > 0) {
        // ---------------- (A) which quads does this thread's entry reach?  (mask from the forward, minus finished quads) ----
        uint32_t qm = 0u;
        if (cur.pos >= 0) {
            qm = cur.qm & 0xffffu;
            if ((uint32_t)cur.pos >= qlast_min) {   // behind the last contributor of every pixel of some quad
#pragma unroll
                for (int q = 0; q < 16; ++q) if ((uint32_t)cur.pos >= qlast[q]) qm &= ~(1u << q);
            }
        }
        {   // one ballot per quad (its halves are the wavefront's two blocks); lane q (< 16) collects quad q's mask
            uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
            for (int q = 0; q < 16; ++q) {   // v_writelane: the scalar ballot halves go straight into lane q (one instruction each)
                const uint64_t bal = __builtin_amdgcn_ballot_w64(((qm >> q) & 1u) != 0u);
                // (inline assembly: no builtin; s_nop 3 = the wait states a v_writelane needs behind the VALU write of the scalar
                // it reads -- the ballot's v_cmp --, which the hazard recogniser does not see inside an asm statement)
                asm("s_nop 3\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                    : "+v"(mlo), "+v"(mhi) : "s"((uint32_t)bal), "s"((uint32_t)(bal >> 32)), "n"(q));
            }
            if (lane < 16) { s_mask[par][lane][2 * wave] = mlo; s_mask[par][lane][2 * wave + 1] = mhi; }
        }
        SR_PHASE(1);   // (A) test + ballots
        lds_barrier();
        SR_PHASE(2);   // barrier after (A)
        // ---------------- (B) slot assignment (every wavefront computes the same) --------
        // Quad q's entries of the chunk occupy a run of slots (no padding between runs: a bucket that reads past the end of its
        // run sees the next run's slots, or the slack behind the buffer, and masks them).  Usually all sixteen runs fit the
        // slot buffer at once; when they do not (dense footprints), the quads are served in several PASSES of consecutive
        // quads, each pass packing its runs from slot 0.  The chunk itself is always the next kChunk entries, which is what
        // lets the next chunk be fetched before this one has been looked at.
        uint32_t pass_starts = 1u;   // uniform: bit q set = a pass starts at quad q
        {
            const int q = nl;   // the four rows of a wavefront compute the same thing, too
            uint32_t len = 0u;
            uint32_t first_of[kBlocks];
#pragma unroll
            for (int blk = 0; blk < kBlocks; ++blk) {
                first_of[blk] = len;
                len += (uint32_t)__popc(s_mask[par][q][blk]);
            }
            // run bases = exclusive prefix of the run lengths over the sixteen quads (a row scan).  When the total fits the
            // slot buffer -- the usual case -- that is all; otherwise the quads are partitioned greedily into passes
            // (sixteen scalar steps: ~160 instructions that every chunk used to pay)
            const uint32_t incl = row_scan_add_u32(len);
            uint32_t my_base = incl - len;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
            if (total > (uint32_t)kCap) {   // uniform
                uint32_t run = 0u;
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {   // values are wave-uniform (readlane)
                    const uint32_t pq = (uint32_t)__builtin_amdgcn_readlane((int)len, qq);
                    if (run + pq > (uint32_t)kCap) { pass_starts |= 1u << qq; run = 0u; }
                    my_base = q == qq ? run : my_base;
                    run += pq;
                }
            }
            if (lane < 16) {   // all four wavefronts write the same values; each reads back only its own writes
                s_qlen[par][q] = len; s_qbase[par][q] = my_base;
#pragma unroll
                for (int blk = 0; blk < kBlocks; ++blk) s_bb[par][q][blk] = (uint16_t)(my_base + first_of[blk]);
            }
            if (threadIdx.x == 0) s_ticket[par ^ 1] = 0u;   // every wavefront has left the previous chunk's replay
#ifdef SR_BWD_STATS
            { const uint32_t pairs = row_allsum_u32(len);
              if (threadIdx.x == 0) { SR_STAT_ADD(0, min(hi, kChunk)); SR_STAT_ADD(1, pairs); SR_STAT_ADD(5, 1); } }
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        float4 s0 = zero4, s1 = zero4;
        float2 s2 = make_float2(0.f, 0.f);
        pass_starts = (uint32_t)__builtin_amdgcn_readfirstlane((int)pass_starts) | 0x10000u;   // sentinel: "pass" 17 starts at 16
        uint32_t last_pmask = 0u;
        auto combine = [&](uint32_t pmask) {
            if (cur.pos >= 0) {
                uint32_t m = qm & pmask;
                while (m) {
                    const int q = __builtin_ctz(m);
                    m &= m - 1u;
                    const uint32_t slot = (uint32_t)s_bb[par][q][myblk] + (uint32_t)__popc(s_mask[par][q][myblk] & lt_mask);
                    float4 a, cc; float2 d;
                    slot_get_sums<HAS_D>(s_slot + kF * slot, a, cc, d);
                    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
                    s1.x += cc.x; s1.y += cc.y; s1.z += cc.z; s1.w += cc.w;
                    s2.x += d.x; if (HAS_D) s2.y += d.y;
                }
            }
        };
#pragma unroll 1
        for (int q_lo = 0; q_lo < 16;) {
            const int q_hi = __builtin_ctz(pass_starts & ~((2u << q_lo) - 1u));   // next pass start behind q_lo
            const uint32_t pmask = ((1u << q_hi) - 1u) & ~((1u << q_lo) - 1u);
            // ---------------- (C) scatter the records into the quads' slot runs ----------------
            if (q_lo == 0) finish_entry(cur, (uint32_t)tx, (uint32_t)ty);   // first use of the gathered record
            {
                uint32_t m = qm & pmask;
                while (m) {
                    const int q = __builtin_ctz(m);
                    m &= m - 1u;
                    const uint32_t slot = (uint32_t)s_bb[par][q][myblk] + (uint32_t)__popc(s_mask[par][q][myblk] & lt_mask);
                    slot_put_record<HAS_D>(s_slot + kF * slot, cur);
                }
            }
            if (threadIdx.x == 0) s_ticket[par] = (uint32_t)q_lo;   // nobody draws tickets between the barrier behind the last
                                                                     // replay and the one below
            SR_PHASE(3);   // (B) + (C)
            lds_barrier();
            SR_PHASE(4);   // barrier after (C)
            // ---------------- (D) replay: each wavefront walks the buckets of its quads ----------------
            // Quads are handed out through an LDS ticket (which wavefront replays which quad does not reach the results: every
            // quad's sums go to its own slots and are combined in a fixed order).  The next ticket is drawn one quad ahead.
            // (Tickets in longest-run-first order -- a rank per quad from sixteen readlane compares in phase B -- were measured
            // in round 4: 0.2123 vs 0.2098 ms, not better: the ranking costs what the shorter wait at the barrier gives back.)
            uint32_t ticket = 0u;
            if (lane == 0) ticket = atomicAdd(&s_ticket[par], 1u);
#pragma unroll 1
            for (;;) {
                const int q = __builtin_amdgcn_readfirstlane((int)ticket);
                if (q >= q_hi) break;
                if (lane == 0) ticket = atomicAdd(&s_ticket[par], 1u);
                const int qy = q >> 2, qx = q & 3;
                const int len = __builtin_amdgcn_readfirstlane((int)s_qlen[par][q]);
                if (len == 0) continue;
                const int base = __builtin_amdgcn_readfirstlane((int)s_qbase[par][q]);
                const int prow = (4 * qy + k) * 16 + 4 * qx;   // tile-local index of pixel (t = 0, row k) of the quad
                QuadCtx c;
                float ST[4], SB[4];
                c.Y = (float)(4 * qy + k);
#pragma unroll
                for (int t = 0; t < 4; ++t) {   // gfx950 has no scalar float arithmetic: three scalar loads from a constant table
                    c.Xa[t] = kColumn[0][4 * qx + t]; c.Xc[t] = kColumn[1][4 * qx + t]; c.Xc2[t] = kColumn[2][4 * qx + t];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 a = s_pixA[prow + t], cb = s_pixB[prow + t];
                    c.gR[t] = a.x; c.gG[t] = a.y; c.gB[t] = a.z; c.gD[t] = a.w;
                    c.gA[t] = cb.x; c.last[t] = __float_as_int(cb.y); ST[t] = cb.z; SB[t] = cb.w;
                }
                // Two buckets per iteration, UP then DOWN (their carries meet in the lanes where they are produced).  The records
                // are read where they are used: reading the next iteration's a round ahead (rounds 2-3) cost 24 register
                // copies per iteration for the hand-over -- 65 VALU cycles of an issue-bound loop -- to hide an LDS latency that
                // the SIMD's other wavefronts cover anyway.
                const int last_bucket = ((len - 1) >> 4) << 4;   // first entry of the quad's last bucket
                auto slot_of = [&](int i, bool up) { return s_slot + kF * (base + min(i, last_bucket) + (up ? nl : 15 - nl)); };
                int i0 = 0;
#pragma unroll 1
                for (; i0 + kBucket < len; i0 += 2 * kBucket) {
                    float* sa = slot_of(i0, true);
                    float* sb = slot_of(i0 + kBucket, false);
                    const SlotIn ea = slot_get_record<HAS_D>(sa);
                    SlotIn eb = slot_get_record<HAS_D>(sb);
                    const bool vb = i0 + kBucket + (15 - nl) < len;   // bucket A is full
                    slot_mask_invalid(eb, vb);
#ifdef SR_BWD_STATS
                    if (lane == 0) { SR_STAT_ADD(2, 2); SR_STAT_ADD(3, 16 * (kBucket + min(kBucket, len - i0 - kBucket))); }
#endif
                    const BucketSums Da = replay_bucket<true, HAS_D>(c, ea, ST, SB, lane);
                    const BucketSums Db = replay_bucket<false, HAS_D>(c, eb, ST, SB, lane);
                    slot_put_sums(sa, k, Da);
                    if (vb) slot_put_sums(sb, k, Db);
                }
                const bool tail = i0 < len;   // a last single bucket (UP): its carries end in lane 15, otherwise they are in lane 0
                if (tail) {
                    float* sa = slot_of(i0, true);
                    SlotIn ea = slot_get_record<HAS_D>(sa);
                    const bool va = i0 + nl < len;
                    slot_mask_invalid(ea, va);
#ifdef SR_BWD_STATS
                    if (lane == 0) { SR_STAT_ADD(2, 1); SR_STAT_ADD(3, 16 * min(kBucket, len - i0)); }
#endif
                    const BucketSums Da = replay_bucket<true, HAS_D>(c, ea, ST, SB, lane);
                    if (va) slot_put_sums(sa, k, Da);
                }
                if (nl == (tail ? 15 : 0)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float* st = reinterpret_cast<float*>(&s_pixB[prow + t]);
                        st[2] = ST[t]; st[3] = SB[t];
                    }
                }
            }
            SR_PHASE(5);   // (D) replay
            lds_barrier();
            SR_PHASE(6);   // barrier after (D)
            // ---------------- (E) add up the quads' sums of every entry (ascending quad index: fixed summation order) ----------
            if (q_hi == 16) { last_pmask = pmask; break; }   // the last pass: combined below, behind the next chunk's requests
            combine(pmask);
            lds_barrier();   // the next pass overwrites the slots
            q_lo = q_hi;
        }
        combine(last_pmask);
        // no barrier here: the next chunk's (A) touches only the other copy of the small tables, and its scatter (C) comes
        // after the barrier that follows (A)
        hi -= kChunk;
        par ^= 1;
        // `cur` is dead except for the few values the final store needs: the next chunk's entries are requested before this
        // chunk's stores go out
        const float2 centre = cur.c;
        const size_t inst = cur.inst;
        const bool has_entry = cur.pos >= 0;
        // shift the moments to the splat centre and store the instance's gradient slot
        if (has_entry) {
            // moments about the tile centre (X, Y = pixel - centre) -> sums of g1 dx^a dy^b with dx = cx - pixel x = ox - X
            const float ox = centre.x - (tx0f + 7.5f), oy = centre.y - (ty0f + 7.5f);
            const float M0 = s0.x, MX = s0.y, MY = s0.z, MXX = s0.w, MXY = s1.x, MYY = s1.y;
            const float Sx = ox * M0 - MX, Sy = oy * M0 - MY;
            const float Sxx = fmaf(ox, Sx - MX, MXX);               // ox^2 M0 - 2 ox MX + MXX
            const float Syy = fmaf(oy, Sy - MY, MYY);
            const float Sxy = fmaf(ox, Sy, MXY) - oy * MX;          // ox oy M0 - ox MY - oy MX + MXY
            slot4[inst * kSlotF4] = make_float4(M0, Sx, Sy, Sxx);
            slot4[inst * kSlotF4 + 1] = make_float4(Sxy, Syy, s1.z, s1.w);
            slot4[inst * kSlotF4 + 2] = make_float4(s2.x, s2.y, 0.f, 0.f);
            b.reached[inst] = 1;
        }
        if (hi > 0) {   // uniform condition
            cur = load_entry(g, id_next, qms, hi - 1 - tid);
            id_next = ids[max(hi - kChunk - 1 - tid, 0)];
        }
        SR_PHASE(7);   // (E) combine
    }
#ifdef SR_BWD_STATS
    if (lane == 0) for (int i = 0; i < 8; ++i) SR_STAT_ADD(8 + i, ph_[i]);
#endif
}

// diagnostic counters of the SR_BWD_STATS build (zeros otherwise); reset != 0 clears them after reading
int backward_stats(unsigned long long* out8, int reset) {   // out8: 16 entries
#ifdef SR_BWD_STATS
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bwd_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stats), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
#else
    for (int i = 0; i < 16; ++i) out8[i] = 0ull;
    (void)reset;
    return 0;
#endif
}

void launch_render_backward_quads(const ViewK& v, const Geom& g, const Binning& b, const Image& im,
                                 const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                                 float* slots, hipStream_t st) {
    const int tiles = v.gx * v.gy;
    if (tiles <= 0) return;
    if (dL_ddepth) hipLaunchKernelGGL(k_render_backward_quads<true>, dim3(tiles), dim3(kBlock), 0, st, v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots);
    else hipLaunchKernelGGL(k_render_backward_quads<false>, dim3(tiles), dim3(kBlock), 0, st, v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots);
}

}  // namespace sr
