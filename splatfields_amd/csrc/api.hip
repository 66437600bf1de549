// C ABI of libsplatraster.so (see include/splatraster.h for what each entry replaces).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <string>
#include <vector>
#include "kernels.h"

namespace {

thread_local std::string g_last_error;
thread_local long long g_last_longest = -1;   // longest tile list of this thread's most recent sr_forward (sr_last_longest_list)

int fail(const std::string& msg) { g_last_error = msg; return 1; }

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    return fail(std::string(what) + ": " + hipGetErrorString(e));
}

#define SR_TRY(expr) do { if (int rc_ = (expr)) return rc_; } while (0)

// ---- optional per-stage timing (HIP events on the launch stream) ----
struct ProfRec { int stage; hipEvent_t a, b; bool ended; unsigned long long serial; };
unsigned long long g_prof_serial = 0;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::mutex g_prof_mutex;   // host threads rendering on their own streams may record concurrently
const char* kStageNames[SR_PROFILE_STAGES] = {"preprocess", "scan", "emit", "sort_tiles", "render_forward",
                                              "render_backward", "preprocess_backward"};
struct StageTimer {
    int idx = -1; hipStream_t st;
    StageTimer(int stage, hipStream_t s) : st(s) {
        if (!g_prof_on) return;
        ProfRec r; r.stage = stage; r.ended = false;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        hipEventRecord(r.a, st);
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        serial = ++g_prof_serial;
        r.serial = serial;
        g_prof.push_back(r); idx = 0;
        end = r.b;
    }
    hipEvent_t end = nullptr;
    unsigned long long serial = 0;
    // the end event is recorded under the same mutex that sr_profile_collect holds while it walks (and destroys) the list:
    // a record whose timer is still open is left alone by the collector
    ~StageTimer() {
        if (idx < 0) return;
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        for (auto& r : g_prof)
            if (r.serial == serial) { hipEventRecord(end, st); r.ended = true; break; }
    }
};

// ---- status blocks: how the instance count of a forward reaches the host ------------------------------------------------
// A pinned, COHERENT 64-byte block of host memory: k_scan_small stores the instance count and the longest list into words 0 / 1
// and then -- system-scope release stores -- the block's current sequence number into words 2 / 3.  The host polls those two
// words: nothing is enqueued in the stream for the read-back (rounds 1-3: a copy command; round 4: an event record, which still
// cost the stream ~8 us between k_scan_small and the scatter -- rocprofv3 kernel trace, profiles/r05_v1_*).
// sr_forward keeps one block per (host thread, device); sr_forward_async hands one out per forward in flight (a "ticket",
// pooled per device under a mutex).  A few bytes each, created lazily.
struct StatusBlock { uint32_t* pinned = nullptr; uint32_t* pinned_dev = nullptr; uint32_t seq = 0; hipStream_t stream = nullptr; int dev = 0; };
typedef StatusBlock HostSync;
typedef StatusBlock Ticket;
thread_local StatusBlock g_sync[64];
std::mutex g_ticket_mutex;
std::vector<StatusBlock*> g_ticket_free[64];
// diagnostics (sr_debug_counters): [0] host waits inside sr_forward, [1] sr_forward_async calls, [2] ticket redemptions that
// found stage 1 still running (the host had to wait), [3] tickets ever created
std::atomic<long long> g_counters[4];

int status_block_init(StatusBlock& b, int dev) {
    // coherent (fine-grained) host memory: the kernel's stores become visible to the host while the stream keeps running
    if (hipHostMalloc(reinterpret_cast<void**>(&b.pinned), 64, hipHostMallocCoherent) != hipSuccess &&
        hipHostMalloc(reinterpret_cast<void**>(&b.pinned), 64, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc failed");
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, b.pinned, 0) != hipSuccess) { hipHostFree(b.pinned); b.pinned = nullptr; return fail("hipHostGetDevicePointer failed"); }
    b.pinned_dev = static_cast<uint32_t*>(dp);
    for (int i = 0; i < 16; ++i) b.pinned[i] = 0u;
    b.dev = dev; b.seq = 0;
    return 0;
}
// a new use of the block: the sequence number the kernel will publish (never 0)
uint32_t status_block_arm(StatusBlock& b, hipStream_t st) {
    b.seq = b.seq + 1u == 0u ? 1u : b.seq + 1u;
    b.stream = st;
    return b.seq;
}
// Waits until both words of the armed use have arrived.  *waited = the first look found them missing.  Polls (the data arrives
// while the stream runs); after two seconds without it falls back to draining the stream, which also surfaces a failed launch.
int status_block_wait(StatusBlock& b, bool* waited) {
    volatile uint32_t* p = b.pinned;
    auto ready = [&] { return p[2] == b.seq && p[3] == b.seq; };
    if (waited) *waited = false;
    if (!ready()) {
        if (waited) *waited = true;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; !ready(); ++spins) {
            if (spins > 4096u) std::this_thread::yield();
            if ((spins & 0xfffu) == 0xfffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                SR_TRY(check_hip(hipStreamSynchronize(b.stream), "wait for the instance count"));
                if (!ready()) return fail("the instance count of the forward never reached the host (k_scan_small did not run?)");
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}

int get_host_sync(HostSync** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail("hipGetDevice failed");
    StatusBlock& h = g_sync[dev];
    if (!h.pinned) SR_TRY(status_block_init(h, dev));
    *out = &h;
    return 0;
}

int ticket_acquire(Ticket** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail("hipGetDevice failed");
    {
        std::lock_guard<std::mutex> lock(g_ticket_mutex);
        if (!g_ticket_free[dev].empty()) { *out = g_ticket_free[dev].back(); g_ticket_free[dev].pop_back(); return 0; }
    }
    StatusBlock* t = new StatusBlock;
    if (int rc = status_block_init(*t, dev)) { delete t; return rc; }
    g_counters[3].fetch_add(1, std::memory_order_relaxed);
    *out = t;
    return 0;
}
void ticket_recycle(Ticket* t) {
    std::lock_guard<std::mutex> lock(g_ticket_mutex);
    g_ticket_free[t->dev].push_back(t);
}

// 0 = choose the backward blend kernel by footprint, 1 = pixel-per-lane, 2 = entry-per-lane (sr_set_backward_kernel)
std::atomic<int> g_bwd_kernel{[] {
    const char* sel = getenv("SPLATRASTER_BWD");
    return !sel ? 0 : (std::string(sel) == "wave" ? 1 : ((std::string(sel) == "quads" || std::string(sel) == "mfma") ? 2 : 0));
}()};


// ---- debug = true: input snapshot on a failed launch ------------------------------------------------------------------
// [EXT] with `debug` set synchronises after every kernel and, when one fails, dumps the call's arguments to
// snapshot_fw.dump / snapshot_bw.dump in the working directory before it re-raises (SURVEY.md section 8b "Error
// conventions").  Same here.  File: "SRSNAP1\0", the failing stage (32 bytes), then records {name[16], uint64 bytes, payload}:
// "view" (SrView scalars: height, width, tanfovx, tanfovy, scale_modifier, sh_degree, sh_coeffs, prefiltered as 8 x 4 bytes),
// "count" (int32), the four camera arrays and every per-splat input that was passed (copied back from the device; a record
// whose copy fails -- the device may be gone after a fault -- has bytes = 0), for the backward also the upstream gradients.
struct CallContext { const SrView* view = nullptr; const SrSplats* splats = nullptr; const char* file = nullptr;
                     const float* dL_dcolor = nullptr; const float* dL_ddepth = nullptr; const float* dL_dalpha = nullptr; };
thread_local CallContext g_call;

void dump_record(FILE* f, const char* name, const void* dev, size_t bytes, bool on_device) {
    char tag[16] = {0};
    std::strncpy(tag, name, 15);
    std::vector<char> host;
    uint64_t n = 0;
    if (dev && bytes) {
        host.resize(bytes);
        if (!on_device) { std::memcpy(host.data(), dev, bytes); n = bytes; }
        else if (hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost) == hipSuccess) n = bytes;
        else (void)hipGetLastError();
    }
    std::fwrite(tag, 1, 16, f); std::fwrite(&n, 8, 1, f);
    if (n) std::fwrite(host.data(), 1, (size_t)n, f);
}

void dump_snapshot(const char* stage) {
    const CallContext& c = g_call;
    if (!c.view || !c.splats || !c.file) return;
    FILE* f = std::fopen(c.file, "wb");
    if (!f) return;
    char head[8] = {'S', 'R', 'S', 'N', 'A', 'P', '1', 0}, st[32] = {0};
    std::strncpy(st, stage, 31);
    std::fwrite(head, 1, 8, f); std::fwrite(st, 1, 32, f);
    const SrView* v = c.view; const SrSplats* s = c.splats;
    const int32_t vi[4] = {v->image_height, v->image_width, v->sh_degree, v->sh_coeffs};
    const float vf[3] = {v->tanfovx, v->tanfovy, v->scale_modifier};
    char vb[32]; std::memcpy(vb, vi, 16); std::memcpy(vb + 16, vf, 12); const int32_t pf = v->prefiltered; std::memcpy(vb + 28, &pf, 4);
    dump_record(f, "view", vb, 32, false);
    const int32_t n = s->count;
    dump_record(f, "count", &n, 4, false);
    dump_record(f, "viewmatrix", v->viewmatrix, 64, true); dump_record(f, "projmatrix", v->projmatrix, 64, true);
    dump_record(f, "campos", v->campos, 12, true); dump_record(f, "bg", v->bg, 12, true);
    const size_t N = (size_t)(n > 0 ? n : 0);
    dump_record(f, "means3D", s->means3D, N * 12, true); dump_record(f, "opacities", s->opacities, N * 4, true);
    dump_record(f, "scales", s->scales, N * 12, true); dump_record(f, "rotations", s->rotations, N * 16, true);
    dump_record(f, "cov3D_precomp", s->cov3D_precomp, N * 24, true);
    dump_record(f, "shs", s->shs, s->shs ? N * 12 * (size_t)(s->shs_rest ? 1 : v->sh_coeffs) : 0, true);
    dump_record(f, "shs_rest", s->shs_rest, N * 12 * 15, true);
    dump_record(f, "colors_precomp", s->colors_precomp, N * 12, true);
    const size_t px = (size_t)v->image_height * v->image_width;
    dump_record(f, "dL_dcolor", c.dL_dcolor, px * 12, true); dump_record(f, "dL_ddepth", c.dL_ddepth, px * 4, true);
    dump_record(f, "dL_dalpha", c.dL_dalpha, px * 4, true);
    std::fclose(f);
}

int after_launch(const SrView* view, hipStream_t st, const char* what) {
    int rc = check_hip(hipGetLastError(), what);
    if (!rc && view && view->debug) rc = check_hip(hipStreamSynchronize(st), what);
    if (rc && view && view->debug) {
        const std::string keep = g_last_error;
        dump_snapshot(what);
        g_last_error = keep + (g_call.file ? std::string(" (inputs written to ") + g_call.file + ")" : std::string());
    }
    return rc;
}

int validate(const SrView* view, const SrSplats* s) {
    if (!view || !s) return fail("null view/splats");
    if (s->count < 0) return fail("negative splat count");
    if (view->image_height <= 0 || view->image_width <= 0) return fail("image size must be positive");
    if (!view->viewmatrix || !view->projmatrix || !view->campos || !view->bg) return fail("viewmatrix/projmatrix/campos/bg must be device pointers");
    if (s->count > 0) {
        if (!s->means3D || !s->opacities) return fail("means3D/opacities missing");
        const bool has_sr = s->scales && s->rotations;
        if (has_sr == (s->cov3D_precomp != nullptr)) return fail("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        if ((s->shs != nullptr) == (s->colors_precomp != nullptr)) return fail("Please provide excatly one of either SHs or precomputed colors!");
        if (s->shs) {
            if (view->sh_degree < 0 || view->sh_degree > 3) return fail("sh_degree must be 0..3");
            if (view->sh_coeffs < (view->sh_degree + 1) * (view->sh_degree + 1)) return fail("shs holds fewer coefficients than sh_degree needs");
            if (s->shs_rest) {
                if (view->sh_coeffs != 16) return fail("shs_rest (dc + rest SH tensors) needs sh_coeffs == 16");
                if ((reinterpret_cast<uintptr_t>(s->shs) | reinterpret_cast<uintptr_t>(s->shs_rest)) & 15u) return fail("shs / shs_rest must be 16-byte aligned");
            }
        } else if (s->shs_rest) return fail("shs_rest without shs");
        if (s->raw_params & ~(SR_RAW_SCALES | SR_RAW_OPACITY | SR_RAW_ROTATIONS | SR_FORWARD_ONLY)) return fail("unknown bits in raw_params");
        if (s->cov3D_precomp && (s->raw_params & (SR_RAW_SCALES | SR_RAW_ROTATIONS))) return fail("raw scales/rotations with cov3D_precomp");
    }
    // tile coordinates are stored in 12 bits (Geom::rect carries the small rectangles' reach mask in the top nibbles, common.h)
    if ((sr::tiles_x(view->image_width) > sr::kMaxTilesPerSide) || (sr::tiles_y(view->image_height) > sr::kMaxTilesPerSide)) return fail("image too large (more than 65520 pixels a side)");
    return 0;
}

sr::ViewK make_view(const SrView* v) {
    sr::ViewK k;
    k.H = v->image_height; k.W = v->image_width;
    k.gx = sr::tiles_x(k.W); k.gy = sr::tiles_y(k.H);
    k.tanfovx = v->tanfovx; k.tanfovy = v->tanfovy;
    k.focal_x = k.W / (2.0f * v->tanfovx); k.focal_y = k.H / (2.0f * v->tanfovy);
    k.scale_modifier = v->scale_modifier;
    k.sh_degree = v->sh_degree; k.sh_coeffs = v->sh_coeffs;
    k.viewmatrix = v->viewmatrix; k.projmatrix = v->projmatrix; k.campos = v->campos; k.bg = v->bg;
    return k;
}

sr::SplatsK make_splats(const SrSplats* s) {
    sr::SplatsK k;
    k.N = s->count; k.means3D = s->means3D; k.opacities = s->opacities; k.scales = s->scales;
    k.rotations = s->rotations; k.cov3D = s->cov3D_precomp; k.shs = s->shs; k.colors = s->colors_precomp;
    k.raw = s->raw_params; k.shs_rest = s->shs_rest;
    return k;
}

}  // namespace

extern "C" {

int sr_version(void) { return SR_VERSION; }
const char* sr_last_error(void) { return g_last_error.c_str(); }

size_t sr_geom_bytes(int n, int h, int w) { return sr::carve_geom(nullptr, n, h, w, nullptr); }
size_t sr_binning_bytes(long long r, int, int) { return sr::carve_binning(nullptr, r, nullptr); }
size_t sr_image_bytes(int h, int w) { return sr::carve_image(nullptr, h, w, nullptr); }
// scratch layout: one 48-byte gradient slot per tile-splat instance (written only for the instances the forward reached)
size_t sr_backward_scratch_bytes(long long r) {
    const size_t n = (size_t)(r > 0 ? r : 1);
    return sr::align_up(n * sr::kSlotFloats * sizeof(float), 256);
}

}  // extern "C"

namespace {

int launch_stage1(const SrView* view, const sr::ViewK& v, const sr::SplatsK& s, const sr::Geom& g, int* radii, uint32_t* host_out,
                  uint32_t host_seq, hipStream_t st) {
    { StageTimer t_(0, st); sr::launch_preprocess(v, s, g, radii, st); }
    SR_TRY(after_launch(view, st, "preprocess"));
    { StageTimer t_(1, st); sr::launch_count_tiles(v, s.N, g, st); sr::launch_scan_small(v, s.N, g, host_out, host_seq, st); }
    SR_TRY(after_launch(view, st, "scan"));
    return 0;
}

// hs != nullptr: k_scan_small stores the instance count / longest list into hs->pinned (status block above); the host waits for
// them after launching the scatter (the GPU keeps working) and then launches only the sort classes that are needed.
// hs == nullptr: nobody waits; `max_len` = the longest list the sort classes must cover (< 0: launch every class).
int launch_stage2(const SrView* view, const sr::ViewK& v, const sr::SplatsK& s, const sr::Geom& g, const sr::Binning& b,
                  const sr::Image& im, float* out_color, float* out_depth, float* out_alpha, HostSync* hs, long long max_len,
                  long long expected_len, hipStream_t st) {
    { StageTimer t_(2, st); sr::launch_emit(v, s.N, g, b, st); }
    SR_TRY(after_launch(view, st, "emit"));
    if (hs) {
        g_counters[0].fetch_add(1, std::memory_order_relaxed);
        SR_TRY(status_block_wait(*hs, nullptr));
        max_len = (long long)hs->pinned[1];
        // the binning buffer is too small: the scatter just launched exits on its own, and nothing else of stage 2 is worth
        // launching -- the caller re-runs it with a buffer that fits (SR_NEED_CAPACITY)
        if ((long long)hs->pinned[0] > (long long)b.capacity) return 0;
    }
    { StageTimer t_(3, st); sr::launch_sort_tiles(v, g, b, max_len, expected_len, st); }
    SR_TRY(after_launch(view, st, "sort_tiles"));
    { StageTimer t_(4, st); sr::launch_render_forward(v, g, b, im, out_color, out_depth, out_alpha, st); }
    SR_TRY(after_launch(view, st, "render_forward"));
    return 0;
}

}  // namespace

extern "C" {

int sr_forward_prepare(const SrView* view, const SrSplats* splats, void* geom, int* radii,
                       long long* instances_out, void* hip_stream) {
    SR_TRY(validate(view, splats));
    g_call = CallContext{view, splats, "snapshot_fw.dump"};
    if (!geom || !instances_out || (splats->count > 0 && !radii)) return fail("null geom/radii/instances_out");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const sr::ViewK v = make_view(view);
    const sr::SplatsK s = make_splats(splats);
    sr::Geom g;
    sr::carve_geom(geom, s.N, v.H, v.W, &g);
    HostSync* hs = nullptr;
    SR_TRY(get_host_sync(&hs));
    SR_TRY(launch_stage1(view, v, s, g, radii, hs->pinned_dev, status_block_arm(*hs, st), st));
    SR_TRY(check_hip(hipStreamSynchronize(st), "sync after prepare"));
    *instances_out = (long long)hs->pinned[0];
    return 0;
}

int sr_forward(const SrView* view, const SrSplats* splats, void* geom, int* radii, void* binning,
               long long binning_capacity, void* image, float* out_color, float* out_depth, float* out_alpha,
               long long* instances_out, void* hip_stream) {
    SR_TRY(validate(view, splats));
    g_call = CallContext{view, splats, "snapshot_fw.dump"};
    if (!geom || !binning || !image || !out_color || !out_depth || !instances_out || (splats->count > 0 && !radii)) return fail("null buffer");
    if (binning_capacity < 0 || binning_capacity >= (1ll << 32)) return fail("binning capacity out of range");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const sr::ViewK v = make_view(view);
    const sr::SplatsK s = make_splats(splats);
    sr::Geom g; sr::Binning b; sr::Image im;
    sr::carve_geom(geom, s.N, v.H, v.W, &g);
    sr::carve_binning(binning, binning_capacity, &b);
    sr::carve_image(image, v.H, v.W, &im);
    HostSync* hs = nullptr;
    SR_TRY(get_host_sync(&hs));
    SR_TRY(launch_stage1(view, v, s, g, radii, hs->pinned_dev, status_block_arm(*hs, st), st));   // k_scan_small stores the counters into hs->pinned
    SR_TRY(launch_stage2(view, v, s, g, b, im, out_color, out_depth, out_alpha, hs, -1, -1, st));  // waits inside, GPU busy
    const long long total = (long long)hs->pinned[0];
    *instances_out = total;
    g_last_longest = (long long)hs->pinned[1];
    return total > binning_capacity ? SR_NEED_CAPACITY : 0;
}

int sr_forward_render(const SrView* view, const SrSplats* splats, void* geom, void* binning,
                      long long instances, void* image, float* out_color, float* out_depth,
                      float* out_alpha, void* hip_stream) {
    SR_TRY(validate(view, splats));
    g_call = CallContext{view, splats, "snapshot_fw.dump"};
    if (!geom || !binning || !image || !out_color || !out_depth) return fail("null buffer");
    if (instances < 0 || instances >= (1ll << 32)) return fail("instance count out of range");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const sr::ViewK v = make_view(view);
    const sr::SplatsK s = make_splats(splats);
    sr::Geom g; sr::Binning b; sr::Image im;
    sr::carve_geom(geom, s.N, v.H, v.W, &g);
    sr::carve_binning(binning, instances, &b);
    sr::carve_image(image, v.H, v.W, &im);
    return launch_stage2(view, v, s, g, b, im, out_color, out_depth, out_alpha, nullptr, -1, -1, st);
}

int sr_forward_async(const SrView* view, const SrSplats* splats, void* geom, int* radii, void* binning,
                     long long binning_capacity, long long longest_list_hint, long long longest_list_expected, void* image,
                     float* out_color, float* out_depth, float* out_alpha, void** ticket_out, void* hip_stream) {
    SR_TRY(validate(view, splats));
    g_call = CallContext{view, splats, "snapshot_fw.dump"};
    if (!geom || !binning || !image || !out_color || !out_depth || !ticket_out || (splats->count > 0 && !radii)) return fail("null buffer");
    if (binning_capacity < 0 || binning_capacity >= (1ll << 32)) return fail("binning capacity out of range");
    if (longest_list_hint < 0) return fail("longest_list_hint must be >= 0");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const sr::ViewK v = make_view(view);
    const sr::SplatsK s = make_splats(splats);
    sr::Geom g; sr::Binning b; sr::Image im;
    sr::carve_geom(geom, s.N, v.H, v.W, &g);
    sr::carve_binning(binning, binning_capacity, &b);
    sr::carve_image(image, v.H, v.W, &im);
    // the sort classes the hint asks for: lists up to 2048 / 4096 / 8192 entries, or all of them
    const long long covered = longest_list_hint <= 2048 ? 2048 : longest_list_hint <= 4096 ? 4096 : longest_list_hint <= 8192 ? 8192 : -1;
    b.sorted_up_to = covered < 0 ? 0xffffffffu : (uint32_t)covered;
    Ticket* t = nullptr;
    SR_TRY(ticket_acquire(&t));
    int rc = launch_stage1(view, v, s, g, radii, t->pinned_dev, status_block_arm(*t, st), st);
    if (!rc) rc = launch_stage2(view, v, s, g, b, im, out_color, out_depth, out_alpha, nullptr, covered,
                                longest_list_expected >= 0 && longest_list_expected <= longest_list_hint ? longest_list_expected : -1, st);
    if (rc) {   // nothing of this call may still write the block when it is handed out again
        (void)hipStreamSynchronize(st);
        ticket_recycle(t);
        return rc;
    }
    g_counters[1].fetch_add(1, std::memory_order_relaxed);
    *ticket_out = t;
    return 0;
}

int sr_ticket_wait(void* ticket, long long* instances_out, long long* longest_list_out) {
    if (!ticket) return fail("null ticket");
    Ticket* t = static_cast<Ticket*>(ticket);
    bool waited = false;
    const int rc = status_block_wait(*t, &waited);
    if (waited) g_counters[2].fetch_add(1, std::memory_order_relaxed);
    if (rc) {
        // The figures never arrived (failed launch, lost device): the block is NOT handed out again -- a k_scan_small that runs
        // late after all would write into a block that belongs to another forward by then -- and the outputs stay untouched.
        // 64 bytes of pinned memory are leaked per failed forward; the process is about to report a device error anyway.
        return rc;
    }
    if (instances_out) *instances_out = (long long)t->pinned[0];
    if (longest_list_out) *longest_list_out = (long long)t->pinned[1];
    ticket_recycle(t);
    return 0;
}

long long sr_last_longest_list(void) { return g_last_longest; }

int sr_ticket_release(void* ticket) {
    return sr_ticket_wait(ticket, nullptr, nullptr);
}

int sr_debug_counters(long long* out4, int reset) {
    if (!out4) return fail("null pointer in sr_debug_counters");
    for (int i = 0; i < 4; ++i) out4[i] = reset ? g_counters[i].exchange(0, std::memory_order_relaxed) : g_counters[i].load(std::memory_order_relaxed);
    return 0;
}

namespace {
// what: 1 = backward blend, 2 = per-splat part for splats [first, first + count), 3 = both
int backward_impl(int what, const SrView* view, const SrSplats* splats, const void* geom, void* binning,
                  long long instances, long long instances_rendered, const void* image, const int* radii,
                  const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, void* scratch,
                  const SrGrads* grads, int first, int count, void* hip_stream) {
    SR_TRY(validate(view, splats));
    g_call = CallContext{view, splats, "snapshot_bw.dump", dL_dcolor, dL_ddepth, dL_dalpha};
    if (!geom || !binning || !image || !scratch) return fail("null buffer");
    if ((what & 1) && !dL_dcolor) return fail("null buffer");
    if ((what & 2) && !grads) return fail("null buffer");
    if (what & 2) {
        if (splats->count > 0 && (!radii || !grads->dL_dmeans3D || !grads->dL_dmeans2D || !grads->dL_dopacity)) return fail("null gradient output");
        if (splats->count > 0 && splats->shs && !grads->dL_dshs && !grads->dL_dcolors) return fail("dL_dshs (or dL_dcolors for the colour-gradient mode) missing");
        if (splats->count > 0 && splats->shs_rest && grads->dL_dshs && !grads->dL_dshs_rest) return fail("dL_dshs_rest missing");
        if (splats->count > 0 && splats->colors_precomp && !grads->dL_dcolors) return fail("dL_dcolors missing");
        if (splats->count > 0 && splats->cov3D_precomp && !grads->dL_dcov3D) return fail("dL_dcov3D missing");
        if (splats->count > 0 && !splats->cov3D_precomp && (!grads->dL_dscales || !grads->dL_drotations)) return fail("dL_dscales/dL_drotations missing");
        if (first < 0 || count < 0 || (first % sr::kBlock) != 0) return fail("splat range must start at a multiple of 256");
    }
    if (splats->raw_params & SR_FORWARD_ONLY) return fail("the forward of these buffers was run with SR_FORWARD_ONLY");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const sr::ViewK v = make_view(view);
    const sr::SplatsK s = make_splats(splats);
    sr::Geom g; sr::Binning b; sr::Image im;
    sr::carve_geom(const_cast<void*>(geom), s.N, v.H, v.W, &g);
    sr::carve_binning(binning, instances, &b);   // Binning::reached is written by the backward blend (splatraster.h)
    sr::carve_image(const_cast<void*>(image), v.H, v.W, &im);
    float* slots = static_cast<float*>(scratch);
    if (what & 1) {
        StageTimer t_(5, st);
        // Two kernels, one slot format.  The entry-per-lane kernel over 4x4 quads (blend_bwd.hip) is the product path at EVERY
        // footprint.  (Round 2 measured a crossover -- pixel-per-lane 0.192 vs 0.203 ms at 7.4 instances per splat, 0.182 vs 0.203
        // at 15 -- and rounds 2-5 switched kernels at SR_BWD_SLOT_SPEC_BELOW instances per splat.  Re-measured in round 6, after
        // three rounds of work on the quad kernel only (MI355X, 800x800, ms of the backward blend, quads vs pixel-per-lane): 4.6
        // instances per splat 0.156 vs 0.262, 7.4: 0.136 vs 0.212, 15: 0.129 vs 0.186, 67: 0.157 vs 0.210, 225: 0.182 vs 0.211,
        // 715: 0.163 vs 0.179, 1270: 0.168 vs 0.171.)  The pixel-per-lane kernel (render.hip) stays as an independently written
        // second implementation of the same slots: SPLATRASTER_BWD=wave / sr_set_backward_kernel(1) select it, the tests and
        // tools/fuzz_backward.py compare the two ("mfma", the name of rounds 2-3, is accepted for "quads").
        const int pinned = g_bwd_kernel.load(std::memory_order_relaxed);   // sr_set_backward_kernel / SPLATRASTER_BWD at load time
        const bool wave_kernel = pinned == 1;
        if (wave_kernel) sr::launch_render_backward(v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots, st);
        else sr::launch_render_backward_quads(v, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, slots, st);
    }
    if (what & 1) SR_TRY(after_launch(view, st, "render_backward"));
    if (what & 2) {
        sr::GradsK gr;
        gr.means3D = grads->dL_dmeans3D; gr.means2D = grads->dL_dmeans2D; gr.opacity = grads->dL_dopacity;
        gr.scales = s.cov3D ? nullptr : grads->dL_dscales; gr.rotations = s.cov3D ? nullptr : grads->dL_drotations;
        gr.cov3D = s.cov3D ? grads->dL_dcov3D : nullptr;
        gr.shs = s.shs ? grads->dL_dshs : nullptr;
        gr.shs_rest = (s.shs_rest && gr.shs) ? grads->dL_dshs_rest : nullptr;
        gr.colors = (s.colors || (s.shs && !grads->dL_dshs)) ? grads->dL_dcolors : nullptr;  // SH input + colours only: colour-gradient mode
        // small footprints (fewer than SR_BWD_SLOT_SPEC_BELOW instances per splat on average; unknown counts as small): the
        // variant that requests a splat's first gradient slots together with their `reached` bytes
        const bool small_fp = !(instances_rendered >= 0 && instances_rendered > (long long)SR_BWD_SLOT_SPEC_BELOW * (long long)s.N);
        { StageTimer t_(6, st); sr::launch_preprocess_backward(v, s, g, radii, slots, b.reached, gr, first, count, small_fp, st); }
        SR_TRY(after_launch(view, st, "preprocess_backward"));
    }
    return 0;
}
}  // namespace

int sr_backward(const SrView* view, const SrSplats* splats, const void* geom, void* binning,
                long long instances, long long instances_rendered, const void* image, const int* radii,
                const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, void* scratch,
                const SrGrads* grads, void* hip_stream) {
    return backward_impl(3, view, splats, geom, binning, instances, instances_rendered, image, radii, dL_dcolor, dL_ddepth, dL_dalpha,
                         scratch, grads, 0, splats ? splats->count : 0, hip_stream);
}

int sr_backward_blend(const SrView* view, const SrSplats* splats, const void* geom, void* binning,
                      long long instances, long long instances_rendered, const void* image,
                      const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, void* scratch, void* hip_stream) {
    return backward_impl(1, view, splats, geom, binning, instances, instances_rendered, image, nullptr, dL_dcolor, dL_ddepth, dL_dalpha,
                         scratch, nullptr, 0, 0, hip_stream);
}

int sr_backward_splats(const SrView* view, const SrSplats* splats, const void* geom, void* binning,
                       long long instances, long long instances_rendered, const void* image, const int* radii, void* scratch,
                       const SrGrads* grads, int first_splat, int n_splats, void* hip_stream) {
    return backward_impl(2, view, splats, geom, binning, instances, instances_rendered, image, radii, nullptr, nullptr, nullptr, scratch,
                         grads, first_splat, n_splats, hip_stream);
}

int sr_set_backward_kernel(int which) {
    if (which < 0 || which > 2) return -1;
    return g_bwd_kernel.exchange(which, std::memory_order_relaxed);
}

int sr_mark_visible(int n, const float* means3D, const float* viewmatrix, const float*, unsigned char* present, void* hip_stream) {
    if (n < 0 || (n > 0 && (!means3D || !viewmatrix || !present))) return fail("bad arguments to sr_mark_visible");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    sr::launch_mark_visible(n, means3D, viewmatrix, present, st);
    return check_hip(hipGetLastError(), "mark_visible");
}

int sr_densification_stats(int n, const float* dL_dmeans2D, const int* radii, float* grad_accum, float* denom, float* max_radii2D,
                           void* hip_stream) {
    if (n < 0) return fail("bad arguments to sr_densification_stats");
    if (n > 0 && (!dL_dmeans2D || !radii)) return fail("null pointer in sr_densification_stats");
    sr::launch_densification_stats(n, dL_dmeans2D, radii, grad_accum, denom, max_radii2D, static_cast<hipStream_t>(hip_stream));
    return check_hip(hipGetLastError(), "densification_stats");
}

int sr_sh_forward(int n, int sh_coeffs, int sh_degree, const float* means3D, const float* shs, const float* campos,
                  float* colors, unsigned char* clamped, void* hip_stream) {
    if (n < 0 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < (sh_degree + 1) * (sh_degree + 1)) return fail("bad arguments to sr_sh_forward");
    if (n > 0 && (!means3D || !shs || !campos || !colors || !clamped)) return fail("null pointer in sr_sh_forward");
    sr::launch_sh_forward(n, sh_coeffs, sh_degree, means3D, shs, campos, colors, clamped, static_cast<hipStream_t>(hip_stream));
    return check_hip(hipGetLastError(), "sh_forward");
}

int sr_sh_forward_views(int n, int sh_coeffs, int sh_degree, int n_views, const float* means3D, const float* shs, const float* campos,
                        float* colors, float* keep, void* hip_stream) {
    if (n < 0 || n_views < 0 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < (sh_degree + 1) * (sh_degree + 1)) return fail("bad arguments to sr_sh_forward_views");
    if (n > 0 && n_views > 0 && (!means3D || !shs || !campos || !colors)) return fail("null pointer in sr_sh_forward_views");
    sr::launch_sh_forward_views(n, sh_coeffs, sh_degree, n_views, means3D, shs, campos, colors, keep, static_cast<hipStream_t>(hip_stream));
    return check_hip(hipGetLastError(), "sh_forward_views");
}

int sr_sh_backward(int n, int sh_coeffs, int sh_degree, int n_views, const float* means3D, const float* shs, const float* campos,
                   const float* dL_dcolors, float scale, float* dL_dshs, float* dL_dmeans3D, int accumulate_means, void* hip_stream) {
    if (n < 0 || n_views < 0 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < (sh_degree + 1) * (sh_degree + 1)) return fail("bad arguments to sr_sh_backward");
    if (n > 0 && n_views > 0 && (!means3D || !shs || !campos || !dL_dcolors)) return fail("null pointer in sr_sh_backward");
    sr::launch_sh_backward(n, sh_coeffs, sh_degree, n_views, means3D, shs, campos, dL_dcolors, scale, dL_dshs, dL_dmeans3D, accumulate_means,
                           static_cast<hipStream_t>(hip_stream));
    return check_hip(hipGetLastError(), "sh_backward");
}

size_t sr_knn_workspace_bytes(int n) { return sr::knn_workspace_bytes(n); }

int sr_knn3_mean_dist2(int n, const float* points, float* mean_dist2, void* workspace, void* hip_stream) {
    if (n < 0 || (n > 0 && (!points || !mean_dist2 || !workspace))) return fail("bad arguments to sr_knn3_mean_dist2");
    sr::launch_knn3(n, points, mean_dist2, workspace, static_cast<hipStream_t>(hip_stream));
    return check_hip(hipGetLastError(), "knn3");
}

size_t sr_densify_workspace_bytes(int n) { return sr::densify_workspace_bytes(n); }

int sr_densify_plan(int n, const float* log_scales, int scale_cols, const float* opacity_logits, const float* grad_accum,
                    const float* denom, const float* max_radii2D, float grad_threshold, float min_opacity, float extent,
                    float percent_dense, float max_screen_size, void* workspace, int* dest, long long* counts5, void* hip_stream) {
    if (n < 0 || (scale_cols != 1 && scale_cols != 3)) return fail("bad arguments to sr_densify_plan");
    if (!counts5) return fail("null counts in sr_densify_plan");
    for (int k = 0; k < 5; ++k) counts5[k] = 0;
    if (n == 0) return 0;
    if (!log_scales || !opacity_logits || !grad_accum || !denom || !workspace || !dest) return fail("null pointer in sr_densify_plan");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const sr::DensifyArgs a{grad_threshold, min_opacity, extent, percent_dense, max_screen_size};
    uint32_t* totals = nullptr;
    sr::launch_densify_plan(n, log_scales, scale_cols, opacity_logits, grad_accum, denom, max_radii2D, a, workspace, dest, &totals, st);
    SR_TRY(check_hip(hipGetLastError(), "densify_plan"));
    uint32_t host[5];
    SR_TRY(check_hip(hipMemcpyAsync(host, totals, sizeof(host), hipMemcpyDeviceToHost, st), "read densify counts"));
    SR_TRY(check_hip(hipStreamSynchronize(st), "sync after densify_plan"));
    for (int k = 0; k < 5; ++k) counts5[k] = (long long)host[k];
    return 0;
}

int sr_densify_gather(int n, int row_floats, const float* src, float* dst, const int* dest, int mode, const float* log_scales,
                      int scale_cols, const float* rotations, const float* unit_normals, void* hip_stream) {
    if (n < 0 || row_floats <= 0 || mode < 0 || mode > 3) return fail("bad arguments to sr_densify_gather");
    if (n == 0) return 0;
    if (!src || !dst || !dest) return fail("null pointer in sr_densify_gather");
    if (mode == 2 && (row_floats != 3 || !log_scales || !rotations || !unit_normals || (scale_cols != 1 && scale_cols != 3)))
        return fail("sr_densify_gather mode 2 (positions) needs [N,3] rows, log-scales, rotations and unit normals");
    sr::launch_densify_gather(n, row_floats, src, dst, dest, mode, log_scales, scale_cols, rotations, unit_normals,
                              static_cast<hipStream_t>(hip_stream));
    return check_hip(hipGetLastError(), "densify_gather");
}

int sr_mlp_chain(int n_points, int hidden_tiles, int n_ops, const SrMlpOp* ops, float negative_slope, void* hip_stream) {
    if (n_points < 0 || !ops) return fail("bad arguments to sr_mlp_chain");
    if (!(negative_slope >= 0.0f && negative_slope < 1.0f)) return fail("sr_mlp_chain: negative_slope must be in [0, 1)");
    if (sr::launch_mlp_chain(n_points, hidden_tiles, n_ops, ops, negative_slope, static_cast<hipStream_t>(hip_stream)))
        return fail("sr_mlp_chain: unsupported op list (hidden_tiles 4 or 8, <= SR_MLP_MAX_OPS ops, even input tile counts, "
                    "16-byte aligned rows wide enough for the tiles read)");
    return check_hip(hipGetLastError(), "mlp_chain");
}

int sr_mlp_pack(int n_jobs, const SrMlpPackJob* jobs, void* hip_stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail("bad arguments to sr_mlp_pack");
    if (sr::launch_mlp_pack(n_jobs, jobs, static_cast<hipStream_t>(hip_stream)))
        return fail("sr_mlp_pack: unsupported job (<= SR_MLP_MAX_PACK_JOBS jobs; mem_pad multiple of 32, reg_width of 16, even tile "
                    "count; counts within their padded sizes; 16-byte aligned destination)");
    return check_hip(hipGetLastError(), "mlp_pack");
}

size_t sr_mlp_weight_grad_workspace(int n_points, int n_jobs, const SrMlpGradJob* jobs) {
    if (!jobs) return 0;
    return sr::mlp_weight_grad_workspace(n_points, n_jobs, jobs);
}

int sr_mlp_weight_grad(int n_points, int n_jobs, const SrMlpGradJob* jobs, void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (!jobs || n_points < 1) return fail("bad arguments to sr_mlp_weight_grad");
    if (sr::launch_mlp_weight_grad(n_points, n_jobs, jobs, workspace, workspace_bytes, static_cast<hipStream_t>(hip_stream)))
        return fail("sr_mlp_weight_grad: unsupported job list (<= SR_MLP_MAX_GRAD_JOBS jobs, <= SR_MLP_MAX_GRAD_TASKS 64x64 blocks, "
                    "16-byte aligned rows with strides that are multiples of 4) or workspace too small");
    return check_hip(hipGetLastError(), "mlp_weight_grad");
}

int sr_mlp_input_forward(int n_points, int multires, int n_features, int time_multires, int row, const float* xyz, const float* features,
                         const float* time, float* x0, void* hip_stream) {
    if (sr::launch_mlp_input_forward(n_points, multires, n_features, time_multires, row, xyz, features, time, x0, static_cast<hipStream_t>(hip_stream)))
        return fail("bad arguments to sr_mlp_input_forward (row a multiple of 4 and >= 3 + 6 multires + n_features [+ 1 + 2 time_multires "
                    "with a time], both multires <= 16)");
    return check_hip(hipGetLastError(), "mlp_input_forward");
}

int sr_mlp_top_gradient(int n_points, int out_features, int row, const float* y, const float* dL_dy, float negative_slope, float* g,
                        void* hip_stream) {
    if (!(negative_slope >= 0.0f && negative_slope < 1.0f)) return fail("sr_mlp_top_gradient: negative_slope must be in [0, 1)");
    if (sr::launch_mlp_top_gradient(n_points, out_features, row, y, dL_dy, negative_slope, g, static_cast<hipStream_t>(hip_stream)))
        return fail("bad arguments to sr_mlp_top_gradient (row a multiple of 4 and >= out_features)");
    return check_hip(hipGetLastError(), "mlp_top_gradient");
}

int sr_mlp_input_backward(int n_points, int multires, int n_features, int row, const float* xyz, const float* dL_dx0, float* dL_dxyz,
                          float* dL_dfeatures, void* hip_stream) {
    if (sr::launch_mlp_input_backward(n_points, multires, n_features, row, xyz, dL_dx0, dL_dxyz, dL_dfeatures, static_cast<hipStream_t>(hip_stream)))
        return fail("bad arguments to sr_mlp_input_backward");
    return check_hip(hipGetLastError(), "mlp_input_backward");
}

int sr_resfield_compose(int n_jobs, const SrResFieldJob* jobs, const long long* frame, void* hip_stream) {
    if (sr::launch_resfield_compose(n_jobs, jobs, frame, static_cast<hipStream_t>(hip_stream)))
        return fail("sr_resfield_compose: unsupported job list (<= SR_RESFIELD_MAX_JOBS jobs, count a multiple of 4, 16-byte aligned "
                    "arrays, 1 <= rank <= SR_RESFIELD_MAX_RANK) or null frame pointer");
    return check_hip(hipGetLastError(), "resfield_compose");
}

size_t sr_resfield_backward_workspace(int n_jobs, const SrResFieldJob* jobs) { return sr::resfield_backward_workspace(n_jobs, jobs); }

int sr_resfield_backward(int n_jobs, const SrResFieldJob* jobs, const long long* frame, void* workspace, size_t workspace_bytes,
                         void* hip_stream) {
    if (sr::launch_resfield_backward(n_jobs, jobs, frame, workspace, workspace_bytes, static_cast<hipStream_t>(hip_stream)))
        return fail("sr_resfield_backward: unsupported job list, null frame pointer or workspace too small");
    return check_hip(hipGetLastError(), "resfield_backward");
}

size_t sr_triplane_backward_workspace(int n_points, int channels, int height, int width) {
    return sr::triplane_backward_workspace(n_points, channels, height, width);
}

int sr_triplane_forward(int n_points, int channels, int height, int width, const float* planes, float* planes_texel_major,
                        const float* points, float* out, void* hip_stream) {
    if (n_points < 0 || !planes || !planes_texel_major || (n_points > 0 && (!points || !out))) return fail("bad arguments to sr_triplane_forward");
    if (sr::launch_triplane_forward(n_points, channels, height, width, planes, planes_texel_major, points, out, static_cast<hipStream_t>(hip_stream)))
        return fail("sr_triplane_forward: channels must be a positive multiple of 4, height * width <= 2^30");
    return check_hip(hipGetLastError(), "triplane_forward");
}

int sr_triplane_backward(int n_points, int channels, int height, int width, const float* planes_texel_major, const float* points,
                         const float* dL_dout, float* dL_dplanes, float* dL_dpoints, void* workspace, void* hip_stream) {
    if (n_points < 0 || !planes_texel_major || (n_points > 0 && (!points || !dL_dout)) || (dL_dplanes && !workspace))
        return fail("bad arguments to sr_triplane_backward");
    const int rc = sr::launch_triplane_backward(n_points, channels, height, width, planes_texel_major, points, dL_dout, dL_dplanes, dL_dpoints,
                                                workspace, static_cast<hipStream_t>(hip_stream));
    if (rc == 1) return fail("sr_triplane_backward: channels must be a multiple of 4 in 4..128, height * width <= 2^30, 12 n_points < 2^32");
    if (rc) return fail("sr_triplane_backward: clearing the tile counters failed");
    return check_hip(hipGetLastError(), "triplane_backward");
}

int sr_debug_layout(int n, int h, int w, long long instances, size_t* out4) {
    if (!out4 || n < 0 || h <= 0 || w <= 0 || instances < 0) return fail("bad arguments to sr_debug_layout");
    sr::Geom g; sr::Binning b;
    char* const origin = reinterpret_cast<char*>(4096);   // carve relative to a non-null base: only differences are used
    sr::carve_geom(origin, n, h, w, &g);
    sr::carve_binning(origin, instances, &b);
    out4[0] = (size_t)(reinterpret_cast<char*>(g.tile_start) - origin);
    out4[1] = (size_t)(reinterpret_cast<char*>(b.sorted_id) - origin);
    out4[2] = (size_t)(reinterpret_cast<char*>(g.total) - origin);
    out4[3] = (size_t)(reinterpret_cast<char*>(g.offsets) - origin);
    return 0;
}

int sr_debug_backward_stats(unsigned long long* out8, int reset) {
    if (!out8) return fail("null pointer in sr_debug_backward_stats");
    if (hipDeviceSynchronize() != hipSuccess) return fail("sr_debug_backward_stats: device synchronisation failed");
    return sr::backward_stats(out8, reset) ? fail("sr_debug_backward_stats: counter copy failed") : 0;
}

/* Test hook: runs the debug path of a failed launch for the given call arguments without breaking the device (writes
 * snapshot_fw.dump / snapshot_bw.dump exactly as a failing stage with view->debug set would).  Returns 0. */
int sr_debug_snapshot(const SrView* view, const SrSplats* splats, const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                      int backward) {
    if (!view || !splats) return fail("null view/splats");
    g_call = CallContext{view, splats, backward ? "snapshot_bw.dump" : "snapshot_fw.dump", dL_dcolor, dL_ddepth, dL_dalpha};
    dump_snapshot(backward ? "render_backward (test hook)" : "preprocess (test hook)");
    return 0;
}

int sr_profile_enable(int on) { g_prof_on = on != 0; return 0; }

int sr_profile_collect(double* ms_sum, long long* launches) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    std::vector<ProfRec> open;
    for (auto& r : g_prof) {
        if (!r.ended) { open.push_back(r); continue; }   // another host thread is inside this stage right now
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            if (ms_sum) ms_sum[r.stage] += ms;
            if (launches) launches[r.stage] += 1;
        }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    g_prof.swap(open);
    return 0;
}

const char* sr_profile_stage_name(int stage) { return (stage >= 0 && stage < SR_PROFILE_STAGES) ? kStageNames[stage] : ""; }

}  // extern "C"
