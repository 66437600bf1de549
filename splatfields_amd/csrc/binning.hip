// Per-tile bucketing and sort for gfx950 (replaces the published pipeline's
// InclusiveSum -> duplicateWithKeys -> 64-bit global radix sort -> identifyTileRanges).
//
// MI355X-first design: the sort key is (tile, depth, splat index).
//  * The tile digit is resolved by a bucket scatter.  Per-tile counts come from a count matrix [chunk][tile] built with
//    LDS histograms (no global atomics), prefix-summed along chunks and tiles, which also yields the tile ranges (and the
//    longest-list-first launch order of the per-tile kernels).
//  * Each tile's list -- ~900 entries at 1 M splats / 800x800, thousands in dense scenes -- is sorted by one workgroup on
//    the 64-bit key (depth bits << 32 | splat index), whose low half is also the payload: a key-only bitonic network held
//    in registers (lane exchanges by DPP / v_permlane swaps, 3 LDS stages) for runs of <= 1024 entries, and pairwise
//    merge-path passes in LDS for longer lists.  A splat appears once per tile, so the key is unique and the order is exactly
//    the published "stable sort by depth, ties by splat index": a total order, independent of the arrival order of the
//    scatter (bit-reproducible).
// HBM traffic per instance: 8 B scatter write + 8 B sort read + 4 B sorted write, versus 6 radix passes x 24 B for a
// global 44-bit LSD sort.
#include <atomic>
#include "kernels.h"
#include "expand.h"

namespace sr {

// Kernels whose dynamic LDS can exceed the 64 KiB default limit need the attribute raised once per device (it belongs to
// the device's copy of the kernel).  `slot` = a small fixed id per kernel.
static void allow_dynamic_lds(const void* kernel, int slot, int bytes) {
    static std::atomic<bool> done[8][64];   // zero-initialised; host threads may race here, the attribute call is idempotent
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    if (dev >= 0 && dev < 64 && done[slot][dev].load(std::memory_order_acquire)) return;
    // a failure is not cached: the launch that needs the larger limit then fails with its own error, and the next call retries
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
    if (dev >= 0 && dev < 64) done[slot][dev].store(true, std::memory_order_release);
}

// ---- atomic-free per-tile counts: count matrix [chunk][tile] built with LDS atomics only ----------
// A chunk is a run of consecutive 256-splat sub-batches owned by one workgroup.  The workgroup keeps a
// histogram over ALL tiles in LDS (800x800 -> 2500 counters = 10 KiB of the CU's 160 KiB), walks its
// sub-batches with the block-cooperative expansion, and writes its row of the matrix once, coalesced.
// (Measured on MI355X: 2.3 M device-scope atomics on 2500 counters cost ~0.17 ms without return value and
// ~0.24 ms with; LDS atomics + a 10 MB matrix cost a few microseconds.)
__global__ void __launch_bounds__(kBlock) k_count_tiles(const ViewK v, int N, const Geom g, const Chunking ch) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];  // [tiles_padded]
    __shared__ uint32_t s_off[kBlock + 1];
    __shared__ ushort4 s_rect[kBlock];
    __shared__ float4 s_r0[kBlock], s_r1[kBlock];   // ellipse of each splat, prepared for the tile_reached test
    __shared__ uint32_t s_scan[8];
    for (int t = 4 * threadIdx.x; t < ch.tiles_padded; t += 4 * kBlock) *reinterpret_cast<uint4*>(&s_hist[t]) = make_uint4(0u, 0u, 0u, 0u);   // tiles_padded % 64 == 0
    // the per-thread path below adds into s_hist before the first workgroup barrier of the sub-batch loop: without this one a
    // wavefront still clearing its share of the table could overwrite another wavefront's increment (a tile count comes out
    // low, and k_emit's piece of that tile segment then overflows into the neighbouring chunk's)
    __syncthreads();
    const int base_chunk = (int)blockIdx.x / ch.slices, slice = (int)blockIdx.x % ch.slices;
    const int sb0 = base_chunk * ch.sub_per_chunk, sb1 = min(sb0 + ch.sub_per_chunk, ch.n_sub);
    // (Requesting the inputs of four sub-batches at a time, as k_emit does, was measured here in round 4: the scan stage went
    // from 0.0489 to 0.0516 ms -- 83 instead of 56 registers -- where the scatter gained 2 us.)
    // Rectangles of up to kDirectTiles tiles -- most of them when a splat's support is a few pixels -- are counted by their own
    // thread (a few LDS atomics); only larger ones go through the workgroup's scan and the cooperative expansion (a
    // binary search per instance), and a sub-batch without any skips both.
    for (int sb = sb0; sb < sb1; ++sb) {
        const int idx = sb * kBlock + threadIdx.x;
        uint32_t touched = 0;
        ushort4 rect = make_ushort4(0, 0, 0, 0);
        float4 r0 = make_float4(0.f, 0.f, -1.f, 0.f), r1 = r0;
        if (idx < N) {
            touched = g.touched[idx]; rect = g.rect[idx];
            if (touched > kMaskTiles) { r0 = g.rec[4 * (size_t)idx]; r1 = g.rec[4 * (size_t)idx + 1]; }   // larger rectangles are tested here
        }
        const uint32_t mask = rect_mask16(rect);   // bit k of a small rectangle's mask (bit 15: mask present)
        const ushort4 packed = rect;
        rect = rect_clean(rect);
        const bool direct = ch.slices == 1 && touched <= kDirectTiles;
        if (direct && touched > 0u) {
            uint32_t k = 0;
            for (uint32_t ty = rect.y; ty < rect.w; ++ty)
                for (uint32_t tx = rect.x; tx < rect.z; ++tx, ++k)
                    if ((mask >> k) & 1u) atomicAdd(&s_hist[ty * (uint32_t)v.gx + tx], 1u);  // LDS atomic
        }
        const uint32_t coop = direct ? 0u : touched;
        if (!__syncthreads_or(coop != 0u)) continue;   // uniform: no large rectangle in this sub-batch (also protects s_off reuse)
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(coop, s_scan, total);
        s_off[threadIdx.x] = excl;
        s_rect[threadIdx.x] = packed;
        tile_test_prepare(r0, r1, s_r0[threadIdx.x], s_r1[threadIdx.x]);
        if (threadIdx.x == 0) s_off[kBlock] = total;
        __syncthreads();
        for_each_block_instance(s_off, s_rect, v.gx, [&](int e, uint32_t k, uint32_t tile, uint32_t, uint32_t tile_x, uint32_t tile_y, uint32_t w) {
            if ((w & kRectMasked) ? (w >> k) & 1u : tile_reached(s_r0[e], s_r1[e], tile_x, tile_y)) atomicAdd(&s_hist[tile], 1u);  // LDS atomic
        }, (uint32_t)slice, (uint32_t)ch.slices);
    }
    __syncthreads();
    uint32_t* row = g.cnt + (size_t)blockIdx.x * ch.tiles_padded;
    for (int t = 4 * threadIdx.x; t < ch.tiles_padded; t += 4 * kBlock) *reinterpret_cast<uint4*>(&row[t]) = *reinterpret_cast<const uint4*>(&s_hist[t]);
}

// Exclusive prefix along the chunk axis of the whole count matrix, per tile column, in ONE pass: a workgroup owns a strip of
// kStripCols columns over all (<= kMaxChunks) rows and holds it in registers -- thread (row group, column) loads its
// kStripRows consecutive rows (all loads in flight: the strip is 125 KB at 1024 chunks), prefixes them serially, the row
// groups' totals meet in LDS, and the strip is written back once.  The column totals (a tile's list length) go to `coltot`.
// (Rounds 2-4 scanned segments of 128 rows through LDS -- grid (tiles / 64, segments) -- and left the prefix over the
// segments to k_scan_small and a third table to k_emit: 15 us + part of k_scan_small's 10.7 us for a 10 MB matrix.)
#ifndef SR_STRIP_COLS
#define SR_STRIP_COLS 32
#endif
constexpr int kStripCols = SR_STRIP_COLS, kStripGroups = 1024 / kStripCols, kStripRows = (kMaxChunks + kStripGroups - 1) / kStripGroups;
__global__ void __launch_bounds__(1024) k_colscan(const Geom g, const Chunking ch) {
    __shared__ uint32_t s_tot[kStripGroups][kStripCols + 1];
    const int c = threadIdx.x % kStripCols, rg = threadIdx.x / kStripCols;
    const int col = blockIdx.x * kStripCols + c, row0 = rg * kStripRows;
    uint32_t* p = g.cnt + (size_t)row0 * ch.tiles_padded + col;
    uint32_t x[kStripRows];
#pragma unroll
    for (int r = 0; r < kStripRows; ++r) x[r] = row0 + r < ch.chunks ? p[(size_t)r * ch.tiles_padded] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int r = 0; r < kStripRows; ++r) { const uint32_t t = x[r]; x[r] = run; run += t; }
    s_tot[rg][c] = run;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll 8
    for (int k = 0; k < kStripGroups; ++k) { const uint32_t t = s_tot[k][c]; base += k < rg ? t : 0u; all += t; }
    if (rg == 0) g.segtot[col] = all;   // column total
#pragma unroll
    for (int r = 0; r < kStripRows; ++r) if (row0 + r < ch.chunks) p[(size_t)r * ch.tiles_padded] = x[r] + base;
}

void launch_count_tiles(const ViewK& v, int N, const Geom& g, hipStream_t st) {
    if (!use_count_matrix(v) || N <= 0) return;
    const Chunking ch = make_chunking(N, v.gx * v.gy);
    // images close to kMaxMatrixTiles tiles: 64 KiB of dynamic histogram + 3 KiB static exceeds the 64 KiB default limit
    allow_dynamic_lds(reinterpret_cast<const void*>(&k_count_tiles), 0, (int)(sizeof(uint32_t) * kMaxMatrixTiles));
    hipLaunchKernelGGL(k_count_tiles, dim3(ch.chunks), dim3(kBlock), sizeof(uint32_t) * ch.tiles_padded, st, v, N, g, ch);
    hipLaunchKernelGGL(k_colscan, dim3(ch.tiles_padded / kStripCols), dim3(1024), 0, st, g, ch);
}

// ---- two small prefix sums in one launch ---------------------------------------------------
// block 0: block_sums[n_sub] -> block_offsets, grand total -> total[0]
// block 1: per-tile totals (count-matrix path: the column totals left by k_colscan; fallback path: the atomically
//          accumulated tile_count) -> tile_start[tiles+1];
//          clears tile_cursor
__global__ void __launch_bounds__(1024) k_scan_small(const Geom g, int n_sub, int n_tiles, const Chunking ch, int use_matrix,
                                                      uint32_t* __restrict__ host_out, uint32_t host_seq) {
    __shared__ uint32_t s_wave[16];
    const bool tiles = blockIdx.x == 1;
    uint32_t* dst = tiles ? g.tile_start : g.block_offsets;
    const int n = tiles ? n_tiles : n_sub;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // kPer consecutive elements per thread and round: 2500 tiles / 3907 sub-batches are ONE round (a round is a wavefront scan,
    // a barrier, 16 LDS reads and a barrier: the kernel is two workgroups' latency chain, 10.7 us with one element per thread)
    constexpr int kPer = 4;
    uint32_t carry = 0, vmax = 0;
    uint32_t v[kPer];   // the last round's elements: the lengths themselves when n <= 1024 * kPer
    for (int base = 0; base < n; base += 1024 * kPer) {
        const int i0 = base + tid * kPer;
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int i = i0 + k;
            v[k] = 0;
            if (i < n) {
                if (!tiles) v[k] = g.block_sums[i];
                else if (!use_matrix) v[k] = g.tile_count[i];
                else v[k] = g.segtot[i];   // column total of the count matrix (k_colscan)
            }
            vmax = max(vmax, v[k]);
            sum += v[k];
        }
        const uint32_t inc = wave_inclusive_scan(sum);
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        uint32_t wave_base = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t t = s_wave[k]; if (k < w) wave_base += t; all += t; }
        uint32_t run = carry + wave_base + inc - sum;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int i = i0 + k;
            if (i < n) { dst[i] = run; if (tiles) g.tile_cursor[i] = 0u; }
            run += v[k];
        }
        carry += all;
        __syncthreads();
    }
    const bool in_regs = n <= 1024 * kPer;   // uniform
    // host_out: pinned, coherent host memory: [0] instance count, [1] longest list, [2] / [3] = host_seq once [0] / [1] are
    // there (system-scope release stores: the host POLLS the two flags -- no event, no copy command in the stream; round 5: an
    // event record between this kernel and the scatter cost the stream 8 us).  Earlier text:
    // host_out: two words of pinned host memory (sr_forward's instance count / longest list read-back): stored from here,
    // the host reads them after the event that follows this kernel -- no copy command in the stream
    if (tid == 0) {
        if (tiles) dst[n] = carry;
        else {
            g.total[0] = carry;
            if (host_out) { host_out[0] = carry; __hip_atomic_store(host_out + 2, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
    if (tiles) {  // longest tile list: lets the host skip the launches of the rare long-list sort classes
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, d, 64));
        __syncthreads();
        if (lane == 0) s_wave[w] = vmax;
        __syncthreads();
        if (tid == 0) {
            uint32_t m = 0;
            for (int k = 0; k < 16; ++k) m = max(m, s_wave[k]);
            g.total[1] = m; s_wave[0] = m;
            if (host_out) { host_out[1] = m; __hip_atomic_store(host_out + 3, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        __syncthreads();
        // Launch order of the blend kernels: longest lists first (longest-processing-time-first keeps the tail of the
        // launch short: 2500 tiles are only ~1.6 rounds of resident workgroups).  Counting sort into 256 linear length
        // classes; the order inside a class is arbitrary (it only affects scheduling, never results).
        __shared__ uint32_t s_hist[256];
        const uint32_t longest = max(s_wave[0], 1u);
        const float to_class = 255.0f / (float)longest;   // class = 255 - floor(255 len / longest), in float: no 64-bit division
        if (tid < 256) s_hist[tid] = 0u;
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < kPer; ++k)
                if (tid * kPer + k < n) atomicAdd(&s_hist[255u - min(255u, (uint32_t)((float)v[k] * to_class))], 1u);
        } else {
            for (int i = tid; i < n; i += 1024) {
                const uint32_t len = dst[i + 1] - dst[i];   // written by this workgroup above (barriers in between)
                atomicAdd(&s_hist[255u - min(255u, (uint32_t)((float)len * to_class))], 1u);
            }
        }
        __syncthreads();
        if (w == 0) {  // exclusive prefix over the 256 classes: 4 per lane
            uint32_t c0 = s_hist[4 * lane], c1 = s_hist[4 * lane + 1], c2 = s_hist[4 * lane + 2], c3 = s_hist[4 * lane + 3];
            const uint32_t sum = c0 + c1 + c2 + c3;
            const uint32_t excl = wave_inclusive_scan(sum) - sum;
            s_hist[4 * lane] = excl; s_hist[4 * lane + 1] = excl + c0; s_hist[4 * lane + 2] = excl + c0 + c1;
            s_hist[4 * lane + 3] = excl + c0 + c1 + c2;
        }
        __syncthreads();
        // How many slots at the head of tile_order can hold a list longer than 2048 / 4096 / 8192 entries (all tiles of the
        // classes down to the one such a length falls into): the long-list sort classes walk only those slots instead of
        // spinning up a workgroup per tile to find out that the tile is not theirs (13.8 us per launch at the headline, on the
        // views that have a single list beyond 2048 entries).
        if (tid < 3) {
            const uint32_t lo = 2048u << tid;   // lists longer than lo
            uint32_t cand = 0u;
            if (longest > lo) {
                const uint32_t cls = 255u - min(255u, (uint32_t)((float)(lo + 1u) * to_class));   // class of length lo + 1: longer lists have classes <= cls
                cand = cls >= 255u ? (uint32_t)n : s_hist[cls + 1u];   // exclusive prefix = tiles in classes 0 .. cls
            }
            g.total[4 + tid] = cand;
        }
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < kPer; ++k)
                if (tid * kPer + k < n)
                    g.tile_order[atomicAdd(&s_hist[255u - min(255u, (uint32_t)((float)v[k] * to_class))], 1u)] = (uint32_t)(tid * kPer + k);
        } else {
            for (int i = tid; i < n; i += 1024) {
                const uint32_t len = dst[i + 1] - dst[i];
                g.tile_order[atomicAdd(&s_hist[255u - min(255u, (uint32_t)((float)len * to_class))], 1u)] = (uint32_t)i;
            }
        }
    }
}

void launch_scan_small(const ViewK& v, int N, const Geom& g, uint32_t* host_out, uint32_t host_seq, hipStream_t st) {
    const Chunking ch = make_chunking(N, v.gx * v.gy);
    hipLaunchKernelGGL(k_scan_small, dim3(2), dim3(1024), 0, st, g, (N + kBlock - 1) / kBlock, v.gx * v.gy, ch,
                       use_count_matrix(v) ? 1 : 0, host_out, host_seq);
}

// ---- emit instances straight into their tile's segment -------------------------------------
// MATRIX: the workgroup owns the same chunk as in k_count_tiles; its write cursors
//   tile_start[t] + (prefix over earlier chunks: the count matrix after k_colscan)
// live in LDS, so slot allocation is an LDS atomic.  The order inside a (chunk, tile) group is
// arbitrary; the per-tile sort below imposes a total order, so the final lists are deterministic.
template <bool MATRIX>
__global__ void __launch_bounds__(kBlock) k_emit(const ViewK v, int N, const Geom g, const Binning b, const Chunking ch) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cur[];  // [tiles_padded] (MATRIX only)
    __shared__ uint32_t s_off[kBlock + 1];
    __shared__ ushort4 s_rect[kBlock];
    __shared__ uint32_t s_depth[kBlock];
    __shared__ float4 s_r0[kBlock], s_r1[kBlock];
    __shared__ uint32_t s_scan[8];
    if (g.total[0] > b.capacity) return;  // binning buffer too small: the host re-runs stage 2 with a larger one
    int sb0 = blockIdx.x, sb1 = blockIdx.x + 1;
    uint32_t slice = 0, slices = 1;
    if constexpr (MATRIX) {
        // Which chunk this workgroup scatters.  Inside a tile's segment of `ent` the chunks' pieces (~1 entry each at 1 M
        // splats) follow each other in chunk order, and workgroup b runs on XCD b % 8 with its own L2: dealt round-robin,
        // every 128-byte line of the segment would collect 8-byte pieces from all eight L2s and leave each of them as a
        // masked partial write (measured: 2.97x the algorithmic bytes).  Giving every XCD a contiguous band of chunks lets one
        // L2 assemble whole lines.
        const int per = (ch.chunks + 7) >> 3;
        const int chunk = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= per || chunk >= ch.chunks) return;
        slice = (uint32_t)(chunk % ch.slices); slices = (uint32_t)ch.slices;
        sb0 = (chunk / ch.slices) * ch.sub_per_chunk; sb1 = min(sb0 + ch.sub_per_chunk, ch.n_sub);
        const uint32_t* row = g.cnt + (size_t)chunk * ch.tiles_padded;
        const int n_tiles = v.gx * v.gy;
        // four tiles per thread and step (rows and tables are 256-byte aligned; the last step may read up to 3 words behind
        // tile_start[n_tiles], inside the table's alignment padding, for cursors nobody uses)
        for (int t = 4 * threadIdx.x; t < n_tiles; t += 4 * kBlock) {
            const uint4 a = *reinterpret_cast<const uint4*>(&g.tile_start[t]), r = *reinterpret_cast<const uint4*>(&row[t]);
            *reinterpret_cast<uint4*>(&s_cur[t]) = make_uint4(a.x + r.x, a.y + r.y, a.z + r.z, a.w + r.w);
        }
    }
    // The chunk's sub-batches are walked one after the other (each needs the workgroup's scan and its LDS tables), but their
    // per-splat inputs are requested kAhead sub-batches at a time: a workgroup is a serial chain of sub-batches with a memory
    // round trip per link otherwise (0.0317 -> 0.0298 ms at the headline)
    constexpr int kAhead = 4;
    for (int sbg = sb0; sbg < sb1; sbg += kAhead) {
        uint32_t touched_a[kAhead], word_a[kAhead], dbits_a[kAhead], base_a[kAhead];
        ushort4 rect_a[kAhead];
        float4 r0_a[kAhead], r1_a[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int idx = (sbg + u) * kBlock + threadIdx.x;
            touched_a[u] = 0; dbits_a[u] = 0; base_a[u] = 0; rect_a[u] = make_ushort4(0, 0, 0, 0);
            if (sbg + u < sb1) {
                base_a[u] = g.block_offsets[sbg + u];
                if (idx < N) {
                    touched_a[u] = g.touched[idx];
                    rect_a[u] = g.rect[idx];
                    dbits_a[u] = g.depth_bits[idx];  // view depth > 0.2: float bits sort as integers
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int idx = (sbg + u) * kBlock + threadIdx.x;
            r0_a[u] = make_float4(0.f, 0.f, -1.f, 0.f); r1_a[u] = r0_a[u];
            if (touched_a[u] > kMaskTiles) { r0_a[u] = g.rec[4 * (size_t)idx]; r1_a[u] = g.rec[4 * (size_t)idx + 1]; }   // as in the count pass
            word_a[u] = rect_mask16(rect_a[u]);   // bit k of a small rectangle's mask (bit 15: mask present); rect_a stays packed
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int sb = sbg + u;
            if (sb >= sb1) break;   // uniform
            const int idx = sb * kBlock + threadIdx.x;
            const uint32_t touched = touched_a[u], mask = word_a[u];
            // two prefixes over the sub-batch: all instances (their indices; `offsets`) and those of the rectangles that take the
            // cooperative expansion -- the ones above kDirectTiles tiles (k_count_tiles makes the same split)
            const bool direct = slices == 1u && touched <= kDirectTiles;
            const uint32_t coop = direct ? 0u : touched;
            uint32_t total, coop_total;
            uint32_t coop_excl;
            const uint32_t excl = block_exclusive_scan2(touched, coop, s_scan, total, coop_excl, coop_total);  // leading barrier protects LDS reuse
            const uint32_t base = base_a[u];
            if (idx < N) g.offsets[idx] = base + excl;
            if (slice == 0u)
                for (uint32_t i = threadIdx.x; i < total; i += kBlock) b.reached[base + i] = 0;   // consecutive threads, consecutive bytes
            const uint32_t first_splat = (uint32_t)sb * kBlock;
            if (direct && touched > 0u) {
                const uint64_t entry = ((uint64_t)dbits_a[u] << 32) | (uint64_t)(first_splat + threadIdx.x);
                uint32_t k = 0;
                const ushort4 rc = rect_clean(rect_a[u]);
                for (uint32_t ty = rc.y; ty < rc.w; ++ty)
                    for (uint32_t tx = rc.x; tx < rc.z; ++tx, ++k) {
                        if (!((mask >> k) & 1u)) continue;   // as in the count pass
                        const uint32_t tile = ty * (uint32_t)v.gx + tx;
                        uint32_t slot;
                        if constexpr (MATRIX) slot = atomicAdd(&s_cur[tile], 1u);  // LDS
                        else slot = g.tile_start[tile] + atomicAdd(&g.tile_cursor[tile], 1u);
                        b.ent[slot] = entry;
                    }
            }
            if (coop_total == 0u) continue;   // uniform
            s_off[threadIdx.x] = coop_excl;
            s_rect[threadIdx.x] = rect_a[u];
            s_depth[threadIdx.x] = dbits_a[u];
            tile_test_prepare(r0_a[u], r1_a[u], s_r0[threadIdx.x], s_r1[threadIdx.x]);
            if (threadIdx.x == 0) s_off[kBlock] = coop_total;
            __syncthreads();
            for_each_block_instance(s_off, s_rect, v.gx, [&](int e, uint32_t k, uint32_t tile, uint32_t, uint32_t tile_x, uint32_t tile_y, uint32_t w) {   // w: as in the count pass
                if (!((w & kRectMasked) ? (w >> k) & 1u : tile_reached(s_r0[e], s_r1[e], tile_x, tile_y))) return;
                uint32_t slot;
                if constexpr (MATRIX) slot = atomicAdd(&s_cur[tile], 1u);  // LDS
                else slot = g.tile_start[tile] + atomicAdd(&g.tile_cursor[tile], 1u);
                b.ent[slot] = ((uint64_t)s_depth[e] << 32) | (uint64_t)(first_splat + (uint32_t)e);
            }, slice, slices);
        }
    }
}

void launch_emit(const ViewK& v, int N, const Geom& g, const Binning& b, hipStream_t st) {
    if (N <= 0) return;
    const Chunking ch = make_chunking(N, v.gx * v.gy);
    allow_dynamic_lds(reinterpret_cast<const void*>(&k_emit<true>), 1, (int)(sizeof(uint32_t) * kMaxMatrixTiles));   // see launch_count_tiles
    if (use_count_matrix(v))
        hipLaunchKernelGGL(k_emit<true>, dim3((ch.chunks + 7) / 8 * 8), dim3(kBlock), sizeof(uint32_t) * ch.tiles_padded, st, v, N, g, b, ch);
    else
        hipLaunchKernelGGL(k_emit<false>, dim3(ch.n_sub), dim3(kBlock), 0, st, v, N, g, b, ch);
}

// ---- per-tile sort: register-resident bitonic network ---------------------------------------------------------
// 256 threads, E consecutive elements per thread (N = 256*E).  A compare-exchange at distance j is
//   j < E        : inside one thread's registers,
//   j < 64*E     : with lane (lane ^ j/E) of the same wavefront through a cross-lane shuffle,
//   otherwise    : with another wavefront through LDS -- exactly 3 such stages for every N.
// (A network living entirely in LDS moves all 12 bytes/entry through LDS log2(N)(log2(N)+1)/2 times -- 55 for N = 1024 --
// and was LDS-bandwidth bound in the first version; this one touches LDS 3 times.)
// Value of lane (lane ^ D) without touching LDS (ds_bpermute_b32 measured at ~24 cycles per wave-instruction
// per SIMD on MI355X, DPP adds/moves at ~4, v_permlane*_swap at ~8):
//   D = 1, 2 : DPP quad_perm          D = 8 : DPP row_ror:8 (rotation by 8 in a row of 16 is xor 8)
//   D = 4    : row_ror:4 for lanes with bit 2 set, row_ror:12 for the others
//   D = 16/32: v_permlane16_swap / v_permlane32_swap of the value with itself, then pick the swapped half
template <int D>
__device__ __forceinline__ uint32_t lane_xor(uint32_t x) {
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false);
    else if constexpr (D == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false);
    else if constexpr (D == 4) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false);  // from lane-4 (mod 16)
        const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x12C, 0xf, 0xf, false);  // from lane+4 (mod 16)
        return (threadIdx.x & 4u) ? a : b;
    } else if constexpr (D == 16) {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const u2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // r.x rows = (x0,x0,x2,x2), r.y rows = (x1,x1,x3,x3)
        return (threadIdx.x & 16u) ? r.x : r.y;
    } else {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const u2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // r.x = (lo,lo), r.y = (hi,hi)
        return (threadIdx.x & 32u) ? r.x : r.y;
    }
}

// d is a compile-time constant after the network loops are unrolled; the dead branches fold away
__device__ __forceinline__ uint32_t lane_xor_d(uint32_t x, int d) {
    switch (d) {
        case 1: return lane_xor<1>(x);
        case 2: return lane_xor<2>(x);
        case 4: return lane_xor<4>(x);
        case 8: return lane_xor<8>(x);
        case 16: return lane_xor<16>(x);
        default: return lane_xor<32>(x);
    }
}

// The sort key is (view-depth bits << 32) | splat index: unique inside a tile (a splat appears once per tile), equal
// depths are ordered by splat index like the stable radix sort of the published pipeline.  The payload IS the low half
// of the key, so the network moves 8 bytes per entry and nothing else.
__device__ __forceinline__ void exchange_select(uint64_t& k, uint64_t pk, bool keep_min) {
    // keys are unique except for the +inf padding, where either choice is fine: one compare, mask xnor, 2 selects
    const bool take = (pk < k) == keep_min;
    k = take ? pk : k;
}

// `tid` is the thread's index inside its 256-thread group (several groups of one workgroup may run the same network
// side by side on different data; the barriers inside are workgroup-wide, so all groups must call it together).
template <int N, int E>
__device__ __forceinline__ void bitonic_regs(uint64_t (&k)[E], uint64_t* skey, uint32_t tid) {
#pragma unroll
    for (int kk = 2; kk <= N; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j < E) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & j) == 0) {
                        const uint32_t i = tid * E + e;
                        const bool asc = (i & kk) == 0;
                        const bool swap = (k[e] > k[e | j]) == asc;
                        const uint64_t ka = k[e], kb = k[e | j];
                        k[e] = swap ? kb : ka; k[e | j] = swap ? ka : kb;
                    }
                }
            } else if (j < 64 * E) {
                const int d = j / E;
                const bool lower = (tid & d) == 0;
                // kk >= 2j > E here, so the direction bit (i & kk) depends on the thread only
                const bool keep_min = lower == (((tid * E) & kk) == 0);
                uint32_t plo[E], phi[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    plo[e] = lane_xor_d((uint32_t)k[e], d);
                    phi[e] = lane_xor_d((uint32_t)(k[e] >> 32), d);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) exchange_select(k[e], ((uint64_t)phi[e] << 32) | plo[e], keep_min);
            } else {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) skey[tid * E + e] = k[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t i = tid * E + e;
                    const bool asc = (i & kk) == 0;
                    const bool lower = (i & j) == 0;
                    exchange_select(k[e], skey[i ^ j], lower == asc);
                }
            }
        }
    }
}

// Loads up to N = 256*E consecutive entries (padding with +inf keys), sorts them in registers; afterwards
// thread t holds the entries of rank t*E .. t*E+E-1.
template <int N, int E>
__device__ __forceinline__ void load_sort_chunk(const Binning& b, uint32_t first, uint32_t m, uint64_t (&k)[E], uint64_t* scratch_key, uint32_t tid) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = tid * E + e;
        k[e] = i < m ? b.ent[first + i] : ~0ull;
    }
    bitonic_regs<N, E>(k, scratch_key, tid);
}

template <int N, int E>
__device__ __forceinline__ void sort_tile_regs(const Binning& b, uint32_t start, uint32_t n, uint64_t* skey) {
    uint64_t k[E];
    load_sort_chunk<N, E>(b, start, n, k, skey, threadIdx.x);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = threadIdx.x * E + e;
        if (i < n) b.sorted_id[start + i] = (uint32_t)k[e];
    }
}

// forward declaration (defined below): runs of 1024 sorted by the register network + pairwise merge passes in LDS
template <int CAP, int THREADS, typename Emit>
__device__ __forceinline__ void sort_block_lds(const Binning& b, uint32_t first, uint32_t n, uint64_t* run_key, Emit&& emit);

// Lists of 1..2048 entries, one workgroup (256 threads) per tile, LDS 16 KiB: up to 1024 entries one register network;
// 1025..2048 two runs sorted one after the other + the merge pass.  One launch for what used to be two (a launch per length
// class spins up one workgroup per tile just to find out that most lists belong to the other class: at the headline
// workload 2500 workgroups each, plus the launch gap, for ~40 % of the tiles).
__global__ void __launch_bounds__(256) k_sort_tiles_regs(const Geom g, const Binning b) {
    __shared__ uint64_t skey[2048];
    const uint32_t total_instances = g.total[0];
    const uint32_t tile = g.tile_order[blockIdx.x];  // longest lists first
    asm volatile("" :: "s"(tile));   // requested together with the count above, not behind the early exit (one round trip less per workgroup)
    if (total_instances > b.capacity) return;
    const uint32_t start = g.tile_start[tile];
    const uint32_t n = g.tile_start[tile + 1] - start;
    if (n == 0u || n > 2048u) return;
    if (n <= 256u) sort_tile_regs<256, 1>(b, start, n, skey);
    else if (n <= 512u) sort_tile_regs<512, 2>(b, start, n, skey);
    else if (n <= 1024u) sort_tile_regs<1024, 4>(b, start, n, skey);
    else sort_block_lds<2048, 256>(b, start, n, skey, [&](uint32_t rank, uint64_t key) { b.sorted_id[start + rank] = (uint32_t)key; });
}

// One merge pass in LDS: src[0, n) holds sorted runs of `width` entries (the last one may be short); adjacent pairs are
// merged, out(position, key) receives every entry with its position after the pass.  Merge path: a thread produces E
// consecutive outputs of its pair -- one binary search along the diagonal for where they start in the two runs
// (log2(width) probes per THREAD), then E sequential compare-and-advance steps with one LDS read each.  (Ranking every
// entry by its own binary search in the other run costs log2(width) dependent probes per ENTRY: a quarter to a third of the
// whole sort at lists of 2000-3000 entries.)  Keys are unique.  E divides 2 * width.
template <int E, typename Out>
__device__ __forceinline__ void merge_run_pairs(const uint64_t* src, uint32_t n, uint32_t width, Out&& out) {
    const uint32_t o = threadIdx.x * (uint32_t)E;  // first output of this thread
    if (o >= n) return;
    const uint32_t base = o & ~(2u * width - 1u);  // start of the pair
    const uint64_t* A = src + base;                 // run B follows run A at A + width
    const uint32_t a = min(width, n - base);
    const uint32_t b = n - base - a < width ? n - base - a : width;
    const uint32_t d = o - base;                    // outputs of the pair before this thread's
    uint32_t lo = d > b ? d - b : 0u, hi = min(d, a);
    while (lo < hi) {                               // smallest ai with A[ai] > B[d - 1 - ai]
        const uint32_t mid = (lo + hi) >> 1;
        if (A[mid] < A[width + d - 1u - mid]) lo = mid + 1u; else hi = mid;
    }
    uint32_t ai = lo, bi = d - lo;
    constexpr uint64_t kInf = ~0ull;
    uint64_t ka = ai < a ? A[ai] : kInf;
    uint64_t kb = bi < b ? A[width + bi] : kInf;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool take_a = ka < kb;
        const uint64_t key = take_a ? ka : kb;
        if (d + (uint32_t)e < a + b) out(o + (uint32_t)e, key);
        ai += take_a ? 1u : 0u;
        bi += take_a ? 0u : 1u;
        const bool more = take_a ? ai < a : bi < b;
        const uint64_t nxt = A[take_a ? min(ai, a - 1u) : width + min(bi, max(b, 1u) - 1u)];
        if (take_a) ka = more ? nxt : kInf; else kb = more ? nxt : kInf;
    }
}

// Sorts n <= CAP entries starting at b.ent[first]: runs of 1024 are sorted by the register network into LDS (one
// 256-thread group per run, side by side), then merged pairwise (merge_run_pairs) until one run is left.  A list of 1100
// entries costs a 1024- and a 256-network instead of the 2048-network of a power-of-two bitonic sort.
// emit(rank, key) receives every entry with its final rank.  run_key: CAP entries, + CAP more when n > 2048.
// Ends with a workgroup barrier.
template <int CAP, int THREADS, typename Emit>
__device__ __forceinline__ void sort_block_lds(const Binning& b, uint32_t first, uint32_t n, uint64_t* run_key, Emit&& emit) {
    constexpr int GROUPS = THREADS / 256;
    const uint32_t group = threadIdx.x >> 8, tid = threadIdx.x & 255u;
    const uint32_t n_runs = (n + 1023u) / 1024u;
    for (uint32_t r0 = 0; r0 < n_runs; r0 += GROUPS) {  // uniform trip count: the barriers below are workgroup-wide
        const uint32_t r = r0 + group;
        const uint32_t m = r < n_runs ? min(1024u, n - r * 1024u) : 0u;
        __syncthreads();  // scratch reuse
        // Every network size has exactly 3 cross-wavefront (LDS) stages = 6 workgroup barriers, so groups may run
        // differently sized networks side by side: a short last run does not pay for a 1024-network.  A group uses the
        // 1024-entry area of its run as the scratch of those stages and then leaves the sorted run there.
        // r < CAP/1024 always (CAP/1024 is a multiple of GROUPS); an idle group (m = 0) sorts padding in its own free slot.
        uint64_t* rk = run_key + r * 1024u;
        uint64_t k[4];  // a 256- / 512-network uses the first 1 / 2 of them
        if (m <= 256u) {
            uint64_t k1[1];
            load_sort_chunk<256, 1>(b, first + r * 1024u, m, k1, rk, tid);
            k[0] = k1[0];
        } else if (m <= 512u) {
            uint64_t k2[2];
            load_sort_chunk<512, 2>(b, first + r * 1024u, m, k2, rk, tid);
            k[0] = k2[0]; k[1] = k2[1];
        } else {
            load_sort_chunk<1024, 4>(b, first + r * 1024u, m, k, rk, tid);
        }
        __syncthreads();  // every wavefront is past the last LDS stage of its network before the area is overwritten
        const uint32_t per = m <= 256u ? 1u : (m <= 512u ? 2u : 4u);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = tid * per + e;
            if ((uint32_t)e < per && i < m) rk[i] = k[e];
        }
    }
    __syncthreads();
    // Pairwise merge passes over the runs (1024 -> 2048 -> ...); the last one hands the entries to emit().  Lists of up
    // to 2048 entries (two runs) need one pass and no second buffer; longer ones ping-pong with run_key + CAP.
    uint64_t* src = run_key;
    uint64_t* dst = run_key + CAP;
    for (uint32_t width = 1024u;; width <<= 1) {
        if (2u * width >= n) {
            merge_run_pairs<CAP / THREADS>(src, n, width, emit);
            break;
        }
        merge_run_pairs<CAP / THREADS>(src, n, width, [&](uint32_t pos, uint64_t key) { dst[pos] = key; });
        __syncthreads();
        uint64_t* t = src; src = dst; dst = t;
    }
    __syncthreads();
}

// lists of (LO, CAP] entries, one workgroup per tile; LDS = CAP * 16 bytes (runs + merge ping-pong buffer)
template <int LO, int CAP, int THREADS>
__global__ void __launch_bounds__(THREADS) k_sort_tiles_merge(const Geom g, const Binning b, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint64_t* run_key = reinterpret_cast<uint64_t*>(s_dyn);
    if (g.total[0] > b.capacity) return;
    const uint32_t limit = min((uint32_t)n_tiles, g.total[LO == 2048 ? 4 : 5]);   // slots that can hold a list longer than LO
    for (uint32_t slot = blockIdx.x; slot < limit; slot += gridDim.x) {
        const uint32_t tile = g.tile_order[slot];  // longest lists first
        const uint32_t start = g.tile_start[tile];
        const uint32_t n = g.tile_start[tile + 1] - start;
        if (n <= (uint32_t)LO || n > (uint32_t)CAP) continue;
        sort_block_lds<CAP, THREADS>(b, start, n, run_key, [&](uint32_t rank, uint64_t key) { b.sorted_id[start + rank] = (uint32_t)key; });
    }
}

// Lists longer than 8192 entries: chunks of 8192 are sorted as above (into the global ping-pong buffer A), then the
// sorted chunks are merged pairwise through the global buffers (rank by binary search; keys are unique).
// One workgroup per long tile.
template <typename T>
__device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int CAP, int THREADS>
__global__ void __launch_bounds__(THREADS) k_sort_tiles_long(const Geom g, const Binning b, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint64_t* run_key = reinterpret_cast<uint64_t*>(s_dyn);
    if (g.total[0] > b.capacity) return;
    const uint32_t limit = min((uint32_t)n_tiles, g.total[6]);   // slots that can hold a list longer than 8192 entries
    for (uint32_t slot = blockIdx.x; slot < limit; slot += gridDim.x) {
        const uint32_t tile = g.tile_order[slot];  // longest lists first
        const uint32_t start = g.tile_start[tile];
        const uint32_t n = g.tile_start[tile + 1] - start;
        if (n <= (uint32_t)CAP) continue;
        uint64_t* k0 = b.keys + start;
        uint64_t* k1 = b.keys_tmp + start;
        for (uint32_t c0 = 0; c0 < n; c0 += CAP) {
            const uint32_t m = min((uint32_t)CAP, n - c0);
            sort_block_lds<CAP, THREADS>(b, start + c0, m, run_key, [&](uint32_t rank, uint64_t key) { k0[c0 + rank] = key; });
        }
        for (uint32_t width = CAP; width < n; width <<= 1) {
            __threadfence();
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
                const uint32_t pair_base = (i / (2 * width)) * (2 * width);
                const uint32_t mid = min(pair_base + width, n), end = min(pair_base + 2 * width, n);
                const bool left = i < mid;
                const uint64_t key = ld_agent(k0 + i);
                uint32_t lo = left ? mid : pair_base, hi = left ? end : mid;  // search the other run
                while (lo < hi) {
                    const uint32_t mm = (lo + hi) >> 1;
                    if (ld_agent(k0 + mm) < key) lo = mm + 1; else hi = mm;
                }
                const uint32_t rank = lo - (left ? mid : pair_base);
                const uint32_t pos = pair_base + (i - (left ? pair_base : mid)) + rank;
                k1[pos] = key;
            }
            uint64_t* tk = k0; k0 = k1; k1 = tk;
        }
        __threadfence();
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += THREADS) b.sorted_id[start + i] = (uint32_t)ld_agent(k0 + i);
        __syncthreads();
    }
}

// max_len: longest tile list if the host knows it (< 0: unknown, launch every class).
// expected_len (sr_forward_async; < 0: none): the longest list the caller really EXPECTS, max_len then being that figure with
// headroom: a long-list class that is launched only because of the headroom -- no list of its length is expected -- gets a
// grid of a few workgroups (they find Geom::total[4..6] = 0 slots and leave; should a list have grown into the class after
// all, they sort it) instead of 512 x 1024 threads that start up to find nothing.
void launch_sort_tiles(const ViewK& v, const Geom& g, const Binning& b, long long max_len, long long expected_len, hipStream_t st) {
    const int tiles = v.gx * v.gy;
    if (tiles <= 0) return;
    auto lds = [](int cap, int) { return (size_t)cap * 16; };   // the runs + the ping-pong buffer of the merge passes
    hipLaunchKernelGGL(k_sort_tiles_regs, dim3(tiles), dim3(256), 0, st, g, b);   // every list of up to 2048 entries
    if (max_len >= 0 && max_len <= 2048) return;
    allow_dynamic_lds(reinterpret_cast<const void*>(&k_sort_tiles_merge<2048, 4096, 1024>), 2, (int)lds(4096, 1024));
    allow_dynamic_lds(reinterpret_cast<const void*>(&k_sort_tiles_merge<4096, 8192, 1024>), 3, (int)lds(8192, 1024));
    allow_dynamic_lds(reinterpret_cast<const void*>(&k_sort_tiles_long<8192, 1024>), 4, (int)lds(8192, 1024));
    // the long classes walk the head of tile_order (Geom::total[4..6] slots) with a grid of at most two workgroups per CU
    const int long_grid = tiles < 512 ? tiles : 512;
    auto grid_for = [&](long long lo) { return (expected_len >= 0 && expected_len <= lo) ? (long_grid < 8 ? long_grid : 8) : long_grid; };
    hipLaunchKernelGGL((k_sort_tiles_merge<2048, 4096, 1024>), dim3(grid_for(2048)), dim3(1024), lds(4096, 1024), st, g, b, tiles);
    if (max_len >= 0 && max_len <= 4096) return;
    hipLaunchKernelGGL((k_sort_tiles_merge<4096, 8192, 1024>), dim3(grid_for(4096)), dim3(1024), lds(8192, 1024), st, g, b, tiles);
    if (max_len >= 0 && max_len <= 8192) return;
    hipLaunchKernelGGL((k_sort_tiles_long<8192, 1024>), dim3(grid_for(8192)), dim3(1024), lds(8192, 1024), st, g, b, tiles);
}

}  // namespace sr
