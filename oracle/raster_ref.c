/*
 * CPU oracle #2: plain-C restatement of the tile rasterizer, forward AND the explicit (hand-derived)
 * backward.  TEST INFRASTRUCTURE ONLY -- never linked, imported or called by the product
 * (splatfields_amd/, diff_gaussian_rasterization/); used by tests/, by __graft_entry__.smoke() and as
 * bench.py's timed "cpu_baseline" (kind "port").
 *
 * PARITY UNPINNED: the algorithm lives in the un-vendored dependency
 * ingra14m/depth-diff-gaussian-rasterization@f2d8fa9 (reference README.md:28, imported at
 * gaussian_renderer/__init__.py:14); the reference holds no golden vectors for it.
 * This file restates the published algorithm, step by step in the order of
 * the published CUDA pipeline: preprocess -> per-tile lists (stable by depth, ties by splat index)
 * -> per-pixel front-to-back blend -> per-pixel back-to-front gradient replay -> per-splat chain
 * rule.  Unlike oracle/torch_oracle.py (autograd), the backward here is written out explicitly,
 * including upstream's 1e-7 regulariser on det^2, its pass-through of the 0.99 alpha clamp and its
 * treatment of the +-1.3 tanfov clamp -- so the two oracles check each other.
 * It is pinned against torch_oracle.py and the in-tree reference pieces
 * (utils/sh_utils.py:57-112, utils/general_utils.py:138-171, utils/graphics_utils.py:42-76)
 * through tests/test_oracle_*.py.
 *
 * Working precision: every intermediate is `real`.  float (default) = the published pipeline's
 * arithmetic -- libraster_ref.so, also the timed cpu_baseline; -DREF_REAL=double = the same
 * statements in double -- libraster_ref64.so: the rounding-free reference the <= 1e-4 image bound is
 * asserted against at the full sizes, as torch_oracle.py's fp64 mode is on the small scenes.  Inputs
 * and outputs are float arrays in both; the lists are ordered by the FLOAT depth in both.
 *
 * Threshold-fragile pixels (ref_rasterize_ex): the blend takes three discrete decisions per
 * (pixel, splat) pair -- power <= 0, alpha >= 1/255, T >= 1e-4 -- and the order of two splats of
 * (nearly) equal depth.  Two correct fp32 evaluations of the same pair may decide differently when
 * the quantity sits within rounding of its threshold, and the pixel then differs by one blended pair.
 * The oracle reports such pixels (`fragile`, same margins as oracle/torch_oracle.py:288-293) and
 * every splat blended into one of them (`splat_flag`: its gradient sums contain that pixel).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/raster_ref.c -o oracle/_build/libraster_ref.so -lm
 *        gcc -O2 -fopenmp -shared -fPIC -DREF_REAL=double oracle/raster_ref.c -o oracle/_build/libraster_ref64.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#ifndef REF_REAL
#define REF_REAL float
#endif
typedef REF_REAL real;
#define RC(x) ((real)(x))
static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_ceil(real x) { return sizeof(real) == 4 ? (real)ceilf((float)x) : (real)ceil((double)x); }
static inline real r_min(real a, real b) { return a < b ? a : b; }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_abs(real a) { return a < 0 ? -a : a; }

static const real SH_C0 = RC(0.28209479177387814), SH_C1 = RC(0.4886025119029199);
static const real SH_C2[5] = {RC(1.0925484305920792), RC(-1.0925484305920792), RC(0.31539156525252005), RC(-1.0925484305920792), RC(0.5462742152960396)};
static const real SH_C3[7] = {RC(-0.5900435899266435), RC(2.890611442640554), RC(-0.4570457994644658), RC(0.3731763325901154),
                              RC(-0.4570457994644658), RC(1.445305721320277), RC(-0.5900435899266435)};
static const real ALPHA_MIN = RC(1.0) / RC(255.0), ALPHA_MAX = RC(0.99), T_STOP = RC(0.0001);

typedef struct RefView {
    int H, W;
    float tanfovx, tanfovy, scale_modifier;
    int sh_degree, sh_coeffs;
    float viewmatrix[16], projmatrix[16], campos[3], bg[3];
} RefView;

typedef struct { float depth; int id; } ListEntry;

static uint32_t float_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static int cmp_entry(const void* a, const void* b) {
    const ListEntry* x = (const ListEntry*)a; const ListEntry* y = (const ListEntry*)b;
    const uint32_t dx = float_bits(x->depth), dy = float_bits(y->depth);  /* depth > 0: bits order = value order */
    if (dx != dy) return dx < dy ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

static void sh_basis(int deg, const real d[3], real B[16]) {
    const real x = d[0], y = d[1], z = d[2];
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            const real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (RC(2) * zz - xx - yy); B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = SH_C3[0] * y * (RC(3) * xx - yy); B[10] = SH_C3[1] * xy * z; B[11] = SH_C3[2] * y * (RC(4) * zz - xx - yy);
                B[12] = SH_C3[3] * z * (RC(2) * zz - RC(3) * xx - RC(3) * yy); B[13] = SH_C3[4] * x * (RC(4) * zz - xx - yy);
                B[14] = SH_C3[5] * z * (xx - yy); B[15] = SH_C3[6] * x * (xx - RC(3) * yy);
            }
        }
    }
}

/* d(basis_k)/d(direction) for k < (deg+1)^2 */
static void sh_basis_grad(int deg, const real d[3], real dB[16][3]) {
    const real x = d[0], y = d[1], z = d[2];
    memset(dB, 0, sizeof(real) * 48);
    if (deg > 0) {
        dB[1][1] = -SH_C1; dB[2][2] = SH_C1; dB[3][0] = -SH_C1;
        if (deg > 1) {
            dB[4][0] = SH_C2[0] * y; dB[4][1] = SH_C2[0] * x;
            dB[5][1] = SH_C2[1] * z; dB[5][2] = SH_C2[1] * y;
            dB[6][0] = SH_C2[2] * RC(-2) * x; dB[6][1] = SH_C2[2] * RC(-2) * y; dB[6][2] = SH_C2[2] * RC(4) * z;
            dB[7][0] = SH_C2[3] * z; dB[7][2] = SH_C2[3] * x;
            dB[8][0] = SH_C2[4] * RC(2) * x; dB[8][1] = SH_C2[4] * RC(-2) * y;
            if (deg > 2) {
                const real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dB[9][0] = SH_C3[0] * RC(6) * xy; dB[9][1] = SH_C3[0] * (RC(3) * xx - RC(3) * yy);
                dB[10][0] = SH_C3[1] * yz; dB[10][1] = SH_C3[1] * xz; dB[10][2] = SH_C3[1] * xy;
                dB[11][0] = SH_C3[2] * RC(-2) * xy; dB[11][1] = SH_C3[2] * (RC(4) * zz - xx - RC(3) * yy); dB[11][2] = SH_C3[2] * RC(8) * yz;
                dB[12][0] = SH_C3[3] * RC(-6) * xz; dB[12][1] = SH_C3[3] * RC(-6) * yz; dB[12][2] = SH_C3[3] * (RC(6) * zz - RC(3) * xx - RC(3) * yy);
                dB[13][0] = SH_C3[4] * (RC(4) * zz - RC(3) * xx - yy); dB[13][1] = SH_C3[4] * RC(-2) * xy; dB[13][2] = SH_C3[4] * RC(8) * xz;
                dB[14][0] = SH_C3[5] * RC(2) * xz; dB[14][1] = SH_C3[5] * RC(-2) * yz; dB[14][2] = SH_C3[5] * (xx - yy);
                dB[15][0] = SH_C3[6] * (RC(3) * xx - RC(3) * yy); dB[15][1] = SH_C3[6] * RC(-6) * xy;
            }
        }
    }
}

static void quat_rot(const float* q, real R[9]) {
    const real r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = RC(1) - RC(2) * (y * y + z * z); R[1] = RC(2) * (x * y - r * z); R[2] = RC(2) * (x * z + r * y);
    R[3] = RC(2) * (x * y + r * z); R[4] = RC(1) - RC(2) * (x * x + z * z); R[5] = RC(2) * (y * z - r * x);
    R[6] = RC(2) * (x * z - r * y); R[7] = RC(2) * (y * z + r * x); R[8] = RC(1) - RC(2) * (x * x + y * y);
}

typedef struct {
    real m0[3], m1[3], tx, ty, tz; int cx, cy;
} Ewa;

static void ewa_setup(const RefView* v, const real pv[3], Ewa* e) {
    const float* vm = v->viewmatrix;
    const real fx = v->W / (RC(2) * v->tanfovx), fy = v->H / (RC(2) * v->tanfovy);
    const real limx = RC(1.3) * v->tanfovx, limy = RC(1.3) * v->tanfovy;
    const real txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    e->cx = (txtz < -limx) || (txtz > limx); e->cy = (tytz < -limy) || (tytz > limy);
    e->tx = r_min(limx, r_max(-limx, txtz)) * pv[2]; e->ty = r_min(limy, r_max(-limy, tytz)) * pv[2]; e->tz = pv[2];
    const real j00 = fx / e->tz, j02 = -(fx * e->tx) / (e->tz * e->tz), j11 = fy / e->tz, j12 = -(fy * e->ty) / (e->tz * e->tz);
    for (int c = 0; c < 3; ++c) {  /* R_view[r][c] = vm[4c + r] */
        e->m0[c] = j00 * vm[4 * c + 0] + j02 * vm[4 * c + 2];
        e->m1[c] = j11 * vm[4 * c + 1] + j12 * vm[4 * c + 2];
    }
}

static void sym_mul(const real S[6], const real v[3], real out[3]) {
    out[0] = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
    out[1] = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
    out[2] = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
}

static void cov3d(const float* scale, real mod, const float* q, real S[6], real R[9], real sv[3]) {
    quat_rot(q, R);
    for (int k = 0; k < 3; ++k) sv[k] = mod * scale[k];
    const real a = sv[0] * sv[0], b = sv[1] * sv[1], c = sv[2] * sv[2];
    S[0] = R[0] * R[0] * a + R[1] * R[1] * b + R[2] * R[2] * c;
    S[1] = R[0] * R[3] * a + R[1] * R[4] * b + R[2] * R[5] * c;
    S[2] = R[0] * R[6] * a + R[1] * R[7] * b + R[2] * R[8] * c;
    S[3] = R[3] * R[3] * a + R[4] * R[4] * b + R[5] * R[5] * c;
    S[4] = R[3] * R[6] * a + R[4] * R[7] * b + R[5] * R[8] * c;
    S[5] = R[6] * R[6] * a + R[7] * R[7] * b + R[8] * R[8] * c;
}

static void view_point(const float* vm, const float* p, real pv[3]) {
    pv[0] = RC(vm[0]) * p[0] + RC(vm[4]) * p[1] + RC(vm[8]) * p[2] + vm[12];
    pv[1] = RC(vm[1]) * p[0] + RC(vm[5]) * p[1] + RC(vm[9]) * p[2] + vm[13];
    pv[2] = RC(vm[2]) * p[0] + RC(vm[6]) * p[1] + RC(vm[10]) * p[2] + vm[14];
}

/* opacity * exp(power) of one (pixel, splat) pair: returns 0 and *power_out > 0 for the published "power > 0" skip */
static inline real pair_alpha_raw(const real* xy, const real* con_o, int id, int px, int py, real* power_out, real* G_out,
                                  real* dx_out, real* dy_out) {
    const real dx = xy[2 * id] - (real)px, dy = xy[2 * id + 1] - (real)py;
    const real* co = con_o + 4 * (size_t)id;
    const real power = RC(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
    *power_out = power; *dx_out = dx; *dy_out = dy;
    if (power > 0) { *G_out = 0; return 0; }
    const real G = r_exp(power);
    *G_out = G;
    return co[3] * G;
}

/*
 * Forward (+ backward when dL_dcolor != NULL).  All pointers are host memory.
 * Outputs: out_color[3*H*W], out_depth[H*W], out_alpha[H*W], radii[N], *num_rendered.
 * Gradients (each may be NULL to skip): d_means3D[N*3], d_means2D[N*3], d_opacity[N], d_scales[N*3],
 * d_rotations[N*4], d_shs[N*K*3], d_colors[N*3].  dL_ddepth / dL_dalpha may be NULL.
 * Optional tile window [tile_y0, tile_y1) x [tile_x0, tile_x1) restricts the blend to a crop (bounded
 * CPU-baseline samples); pass all zeros for the whole image.
 * fragile[H*W] / splat_flag[N] (each may be NULL): see the header; `margin` = relative fp32 margin of the
 * alpha threshold (torch_oracle.py uses 2e-4; the T threshold gets 50 x that, a product of many factors).
 * cond_bound[5*H*W] (may be NULL; channels r, g, b, depth, alpha): first-order bound on what `xy_ulps` float ulps of
 * rounding in every splat's screen-space centre (and a few ulps in its exponent) do to the pixel -- the noise ANY fp32
 * pipeline carries, the published one included: the centre is stored as a float in absolute pixel coordinates
 * (ulp(700) = 6e-5 px), and a splat seen at the edge of its support (|power| ~ 5, |grad power| ~ 2.7 / px) moves its
 * alpha by ~2e-4 relative per ulp.  A pixel that a hundred splats cover averages this out; a pixel that shows the rim of
 * one or two faint splats carries it whole.  radius_raw[N] (may be NULL): 3 sqrt(lambda_max) before the ceil.
 */
int ref_rasterize_ex(const RefView* v, int N, const float* means3D, const float* opacities, const float* scales,
                     const float* rotations, const float* shs, const float* colors,
                     float* out_color, float* out_depth, float* out_alpha, int* radii, long long* num_rendered,
                     const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                     float* d_means3D, float* d_means2D, float* d_opacity, float* d_scales, float* d_rotations,
                     float* d_shs, float* d_colors,
                     int tile_x0, int tile_y0, int tile_x1, int tile_y1, int threads,
                     unsigned char* fragile, unsigned char* splat_flag, float margin,
                     float* cond_bound, float xy_ulps, float* radius_raw) {
    const int H = v->H, W = v->W, K = v->sh_coeffs;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const float* vm = v->viewmatrix; const float* pm = v->projmatrix;
    if (tile_x1 <= tile_x0 || tile_y1 <= tile_y0) { tile_x0 = 0; tile_y0 = 0; tile_x1 = gx; tile_y1 = gy; }
#ifdef _OPENMP
    /* the caller's process shares this OpenMP runtime (PyTorch's CPU kernels): the team size is restored on the way out,
     * otherwise every later small tensor op runs on `threads` threads */
    const int omp_prev_threads = omp_get_max_threads();
    if (threads > 0) omp_set_num_threads(threads);
#endif
    real* xy = (real*)malloc(sizeof(real) * 2 * (size_t)(N + 1));
    real* con_o = (real*)malloc(sizeof(real) * 4 * (size_t)(N + 1));
    real* rgb = (real*)malloc(sizeof(real) * 3 * (size_t)(N + 1));
    real* depth = (real*)malloc(sizeof(real) * (size_t)(N + 1));
    int* rect = (int*)malloc(sizeof(int) * 4 * (size_t)(N + 1));
    unsigned char* clamped = (unsigned char*)calloc((size_t)(N + 1) * 3, 1);
    uint32_t* tile_cnt = (uint32_t*)calloc((size_t)gx * gy + 1, sizeof(uint32_t));
    if (fragile) memset(fragile, 0, (size_t)H * W);
    if (splat_flag) memset(splat_flag, 0, (size_t)(N > 0 ? N : 0));
    const int want_fragile = fragile != NULL || splat_flag != NULL;
    real* rraw = (real*)calloc((size_t)(N + 1), sizeof(real));
    unsigned char* pre_fragile = want_fragile ? (unsigned char*)calloc((size_t)H * W, 1) : NULL;

    /* ---- preprocess ---- */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        radii[i] = 0; rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
        const float* p = means3D + 3 * (size_t)i;
        real pv[3]; view_point(vm, p, pv);
        if (pv[2] <= RC(0.2)) continue;
        const real hx = RC(pm[0]) * p[0] + RC(pm[4]) * p[1] + RC(pm[8]) * p[2] + pm[12];
        const real hy = RC(pm[1]) * p[0] + RC(pm[5]) * p[1] + RC(pm[9]) * p[2] + pm[13];
        const real hw = RC(pm[3]) * p[0] + RC(pm[7]) * p[1] + RC(pm[11]) * p[2] + pm[15];
        const real pw = RC(1) / (hw + RC(0.0000001));
        real S[6], R[9], sv[3];
        cov3d(scales + 3 * (size_t)i, v->scale_modifier, rotations + 4 * (size_t)i, S, R, sv);
        Ewa e; ewa_setup(v, pv, &e);
        real Sm0[3], Sm1[3]; sym_mul(S, e.m0, Sm0); sym_mul(S, e.m1, Sm1);
        const real a = e.m0[0] * Sm0[0] + e.m0[1] * Sm0[1] + e.m0[2] * Sm0[2] + RC(0.3);
        const real b = e.m0[0] * Sm1[0] + e.m0[1] * Sm1[1] + e.m0[2] * Sm1[2];
        const real c = e.m1[0] * Sm1[0] + e.m1[1] * Sm1[1] + e.m1[2] * Sm1[2] + RC(0.3);
        const real det = a * c - b * b;
        if (det == 0) continue;
        const real det_inv = RC(1) / det;
        const real mid = RC(0.5) * (a + c);
        const real lam1 = mid + r_sqrt(r_max(RC(0.1), mid * mid - det));
        const real my_radius = r_ceil(RC(3) * r_sqrt(lam1));
        rraw[i] = RC(3) * r_sqrt(lam1);
        const real px = ((hx * pw + RC(1)) * W - RC(1)) * RC(0.5), py = ((hy * pw + RC(1)) * H - RC(1)) * RC(0.5);
        int x0 = (int)((px - my_radius) / TILE), y0 = (int)((py - my_radius) / TILE);
        int x1 = (int)((px + my_radius + TILE - 1) / TILE), y1 = (int)((py + my_radius + TILE - 1) / TILE);
        x0 = x0 < 0 ? 0 : (x0 > gx ? gx : x0); x1 = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
        y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0); y1 = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        if (colors) { rgb[3 * i] = colors[3 * (size_t)i]; rgb[3 * i + 1] = colors[3 * (size_t)i + 1]; rgb[3 * i + 2] = colors[3 * (size_t)i + 2]; }
        else {
            real d[3] = {RC(p[0]) - v->campos[0], RC(p[1]) - v->campos[1], RC(p[2]) - v->campos[2]};
            const real il = RC(1) / r_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] *= il; d[1] *= il; d[2] *= il;
            real B[16]; sh_basis(v->sh_degree, d, B);
            const int nb = (v->sh_degree + 1) * (v->sh_degree + 1);
            const float* sh = shs + (size_t)i * K * 3;
            for (int ch = 0; ch < 3; ++ch) {
                real r = 0;
                for (int k = 0; k < nb; ++k) r += B[k] * sh[3 * k + ch];
                r += RC(0.5);
                clamped[3 * i + ch] = r < 0;
                rgb[3 * i + ch] = r < 0 ? 0 : r;
            }
        }
        depth[i] = pv[2]; radii[i] = (int)my_radius; xy[2 * i] = px; xy[2 * i + 1] = py;
        con_o[4 * i] = c * det_inv; con_o[4 * i + 1] = -b * det_inv; con_o[4 * i + 2] = a * det_inv; con_o[4 * i + 3] = opacities[i];
        rect[4 * i] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
    }

    if (radius_raw) for (int i = 0; i < N; ++i) radius_raw[i] = (float)rraw[i];

    /* ---- splats whose TILE RECTANGLE is a rounding decision: 3 sqrt(lambda) within fp32 rounding of an integer (the ceil
     * flips) or a rectangle bound (centre -+ radius) / 16 within rounding of an integer (the cast flips).  Another fp32
     * evaluation may give such a splat one more / one fewer row or column of tiles; the pixels of those tiles that the
     * splat can reach with alpha >= 1/255 are fragile. ---- */
    if (want_fragile) {
        const real m_a = RC(margin) * ALPHA_MIN;
#pragma omp parallel for schedule(dynamic, 256)
        for (int i = 0; i < N; ++i) {
            if (radii[i] <= 0) continue;
            const real raw = rraw[i], rr = (real)radii[i];
            const real near = raw - (real)(long long)(raw + RC(0.5));      /* distance to the nearest integer, signed */
            const real tol_r = RC(2e-4) * r_max(RC(1), raw), tol_p = RC(1e-3);
            real r_alt[2] = {rr, rr};
            if (r_abs(near) < tol_r) r_alt[1] = near > 0 ? rr - RC(1) : rr + RC(1);
            int lo[4] = {1 << 30, 1 << 30, 1 << 30, 1 << 30}, hi[4] = {-1, -1, -1, -1};   /* x0, y0, x1, y1 over the alternatives */
            for (int a = 0; a < 2; ++a)
                for (int sgn = -1; sgn <= 1; sgn += 2) {
                    const real cx = xy[2 * i] + sgn * tol_p, cy = xy[2 * i + 1] + sgn * tol_p, r = r_alt[a];
                    int b[4] = {(int)((cx - r) / TILE), (int)((cy - r) / TILE), (int)((cx + r + TILE - 1) / TILE), (int)((cy + r + TILE - 1) / TILE)};
                    for (int k = 0; k < 4; ++k) {
                        const int g = (k & 1) ? gy : gx;
                        b[k] = b[k] < 0 ? 0 : (b[k] > g ? g : b[k]);
                        if (b[k] < lo[k]) lo[k] = b[k];
                        if (b[k] > hi[k]) hi[k] = b[k];
                    }
                }
            if (lo[0] == hi[0] && lo[1] == hi[1] && lo[2] == hi[2] && lo[3] == hi[3]) continue;   /* every alternative gives the same rectangle */
            /* tiles in the union [lo0, hi2) x [lo1, hi3) that are not in the intersection [hi0, lo2) x [hi1, lo3) */
            for (int ty = lo[1]; ty < hi[3]; ++ty)
                for (int tx = lo[0]; tx < hi[2]; ++tx) {
                    if (tx >= hi[0] && tx < lo[2] && ty >= hi[1] && ty < lo[3]) continue;
                    for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; ++py)
                        for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; ++px) {
                            real power, G, dx, dy;
                            const real raw_a = pair_alpha_raw(xy, con_o, i, px, py, &power, &G, &dx, &dy);
                            if (power <= RC(1e-6) && r_min(ALPHA_MAX, raw_a) >= ALPHA_MIN - m_a) {
#pragma omp atomic write
                                pre_fragile[(size_t)py * W + px] = 1;
                            }
                        }
                }
        }
    }

    /* ---- per-tile lists: count, prefix, fill (splat order), sort by (float depth, id) ---- */
    long long total = 0;
    for (int i = 0; i < N; ++i) {
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) tile_cnt[y * gx + x]++;
        total += (long long)(rect[4 * i + 2] - rect[4 * i]) * (rect[4 * i + 3] - rect[4 * i + 1]);
    }
    if (num_rendered) *num_rendered = total;
    uint64_t* tile_start = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)gx * gy + 1));
    tile_start[0] = 0;
    for (int t = 0; t < gx * gy; ++t) tile_start[t + 1] = tile_start[t] + tile_cnt[t];
    ListEntry* list = (ListEntry*)malloc(sizeof(ListEntry) * (size_t)(total + 1));
    memset(tile_cnt, 0, sizeof(uint32_t) * (size_t)gx * gy);
    for (int i = 0; i < N; ++i)
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) {
                const int t = y * gx + x;
                ListEntry le; le.depth = (float)depth[i]; le.id = i;
                list[tile_start[t] + tile_cnt[t]++] = le;
            }
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < gx * gy; ++t) qsort(list + tile_start[t], tile_cnt[t], sizeof(ListEntry), cmp_entry);

    /* ---- forward blend ---- */
    real* final_T = (real*)malloc(sizeof(real) * (size_t)H * W);
    uint32_t* n_contrib = (uint32_t*)calloc((size_t)H * W, sizeof(uint32_t));
    const size_t hw = (size_t)H * W;
    const real m_alpha = RC(margin) * ALPHA_MIN, m_T = RC(50) * RC(margin) * T_STOP, reach_T = T_STOP * (RC(1) - RC(50) * RC(margin));
    uint32_t longest = 1;
    for (int t = 0; t < gx * gy; ++t) if (tile_cnt[t] > longest) longest = tile_cnt[t];
    if (cond_bound) memset(cond_bound, 0, sizeof(float) * 5 * hw);
#pragma omp parallel
    {
    /* per-thread record of the blended entries of one pixel (cond_bound): alpha, T in front, list position */
    real* rec_a = cond_bound ? (real*)malloc(sizeof(real) * 2 * (size_t)longest) : NULL;
    uint32_t* rec_j = cond_bound ? (uint32_t*)malloc(sizeof(uint32_t) * (size_t)longest) : NULL;
#pragma omp for schedule(dynamic, 1) collapse(2)
    for (int ty = tile_y0; ty < tile_y1; ++ty)
        for (int tx = tile_x0; tx < tile_x1; ++tx) {
            const int t = ty * gx + tx;
            const ListEntry* L = list + tile_start[t]; const uint32_t n = tile_cnt[t];
            for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; ++py)
                for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; ++px) {
                    real T = 1, C[3] = {0, 0, 0}, D = 0; uint32_t last = 0, n_rec = 0;
                    /* the fragile analysis follows the list a little beyond the stop: T_incl = product over ALL valid entries */
                    real T_incl = 1; int stopped = 0, frag = 0; uint32_t prev_valid_bits = 0; uint32_t j_end = 0;
                    for (uint32_t j = 0; j < n; ++j) {
                        if (stopped && !(want_fragile && T_incl >= reach_T)) break;
                        j_end = j + 1;
                        const int id = L[j].id;
                        real power, G, dx, dy;
                        const real raw = pair_alpha_raw(xy, con_o, id, px, py, &power, &G, &dx, &dy);
                        const real alpha = r_min(ALPHA_MAX, raw);
                        const int valid = power <= 0 && alpha >= ALPHA_MIN;
                        if (want_fragile) {   /* this entry is `reached`: T_incl of the entries in front of it >= reach_T */
                            if (power <= 0 && r_abs(alpha - ALPHA_MIN) < m_alpha) frag = 1;
                            if (r_abs(power) < RC(1e-6) && con_o[4 * (size_t)id + 3] >= ALPHA_MIN) frag = 1;   /* power ~ 0: exp ~ 1 */
                            if (valid) {
                                const real T_next = T_incl * (RC(1) - alpha);
                                if (r_abs(T_next - T_STOP) < m_T) frag = 1;
                                /* two valid entries whose float depths are within 2 ulp: another fp32 evaluation of the depth may order them the other way */
                                const uint32_t bits = float_bits(L[j].depth);
                                if (prev_valid_bits && bits - prev_valid_bits <= 2u) frag = 1;
                                prev_valid_bits = bits;
                                T_incl = T_next;
                            }
                        }
                        if (stopped || !valid) continue;
                        const real test_T = T * (RC(1) - alpha);
                        if (test_T < T_STOP) { stopped = 1; continue; }
                        for (int ch = 0; ch < 3; ++ch) C[ch] += rgb[3 * id + ch] * alpha * T;
                        D += depth[id] * alpha * T;
                        if (cond_bound) { rec_a[2 * n_rec] = alpha; rec_a[2 * n_rec + 1] = T; rec_j[n_rec++] = j; }
                        T = test_T; last = j + 1;
                    }
                    const size_t pix = (size_t)py * W + px;
                    if (want_fragile && pre_fragile[pix]) frag = 1;
                    if (cond_bound && n_rec) {
                        /* back to front: d out / d alpha_i = T_i c_i - (colour behind i + T_final bg) / (1 - alpha_i)  (the backward's dL/dalpha per output);
                         * d alpha_i = alpha_i * (|grad power| . xy_ulps ulp(centre) + 4 ulp |power|), nothing through the 0.99 clamp */
                        real behind[4] = {T * v->bg[0], T * v->bg[1], T * v->bg[2], 0}, bnd[5] = {0, 0, 0, 0, 0};
                        for (int r = (int)n_rec - 1; r >= 0; --r) {
                            const int id = L[rec_j[r]].id;
                            const real al = rec_a[2 * r], Ti = rec_a[2 * r + 1], inv = RC(1) / (RC(1) - al);
                            real power, G, dx, dy;
                            const real raw = pair_alpha_raw(xy, con_o, id, px, py, &power, &G, &dx, &dy);
                            const real* co = con_o + 4 * (size_t)id;
                            const float cxf = (float)xy[2 * id], cyf = (float)xy[2 * id + 1];
                            const real ux = (real)(nextafterf(fabsf(cxf), INFINITY) - fabsf(cxf)), uy = (real)(nextafterf(fabsf(cyf), INFINITY) - fabsf(cyf));
                            const real dpow = RC(xy_ulps) * (r_abs(co[0] * dx + co[1] * dy) * ux + r_abs(co[1] * dx + co[2] * dy) * uy) + RC(4) * RC(5.9604645e-8) * r_abs(power);
                            const real dal = raw < ALPHA_MAX ? al * dpow : 0;
                            const real cc[4] = {rgb[3 * id], rgb[3 * id + 1], rgb[3 * id + 2], depth[id]};
                            for (int ch = 0; ch < 4; ++ch) {
                                bnd[ch] += r_abs(Ti * cc[ch] - behind[ch] * inv) * dal;
                                behind[ch] += al * Ti * cc[ch];
                            }
                            bnd[4] += T * inv * dal;
                        }
                        for (int ch = 0; ch < 5; ++ch) cond_bound[ch * hw + pix] = (float)bnd[ch];
                    }
                    final_T[pix] = T; n_contrib[pix] = last;
                    for (int ch = 0; ch < 3; ++ch) out_color[ch * hw + pix] = (float)(C[ch] + T * v->bg[ch]);
                    out_depth[pix] = (float)D;
                    if (out_alpha) out_alpha[pix] = (float)(RC(1) - T);
                    if (frag) {
                        if (fragile) fragile[pix] = 1;
                        if (splat_flag)   /* every splat this pixel blends, or nearly blends, carries the pixel in its sums */
                            for (uint32_t j = 0; j < j_end; ++j) {
                                const int id = L[j].id;
                                real power, G, dx, dy;
                                const real raw = pair_alpha_raw(xy, con_o, id, px, py, &power, &G, &dx, &dy);
                                if (power <= RC(1e-6) && r_min(ALPHA_MAX, raw) >= ALPHA_MIN - m_alpha) {
#pragma omp atomic write
                                    splat_flag[id] = 1;
                                }
                            }
                    }
                }
        }
    free(rec_a); free(rec_j);
    }

    /* ---- backward ---- */
    if (dL_dcolor) {
        /* screen-space accumulators: mean2D(2), conic(3: x, y(half), w), opacity, colour(3), depth */
        real* acc = (real*)calloc((size_t)(N + 1) * 10, sizeof(real));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
        for (int ty = tile_y0; ty < tile_y1; ++ty)
            for (int tx = tile_x0; tx < tile_x1; ++tx) {
                const int t = ty * gx + tx;
                const ListEntry* L = list + tile_start[t];
                for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; ++py)
                    for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; ++px) {
                        const size_t pix = (size_t)py * W + px;
                        const real T_final = final_T[pix];
                        real T = T_final;
                        const real g[3] = {dL_dcolor[pix], dL_dcolor[hw + pix], dL_dcolor[2 * hw + pix]};
                        const real gD = dL_ddepth ? dL_ddepth[pix] : 0, gA = dL_dalpha ? dL_dalpha[pix] : 0;
                        const real bg_dot = v->bg[0] * g[0] + v->bg[1] * g[1] + v->bg[2] * g[2];
                        real accum[3] = {0, 0, 0}, accum_d = 0, accum_a = 0, last_alpha = 0, last_c[3] = {0, 0, 0}, last_d = 0;
                        for (int j = (int)n_contrib[pix] - 1; j >= 0; --j) {
                            const int id = L[j].id;
                            real power, G, dx, dy;
                            const real raw = pair_alpha_raw(xy, con_o, id, px, py, &power, &G, &dx, &dy);
                            if (power > 0) continue;
                            const real* co = con_o + 4 * (size_t)id;
                            const real alpha = r_min(ALPHA_MAX, raw);
                            if (alpha < ALPHA_MIN) continue;
                            T = T / (RC(1) - alpha);
                            const real w = alpha * T;
                            real dL_dalpha_ = 0;
                            real* a = acc + 10 * (size_t)id;
                            for (int ch = 0; ch < 3; ++ch) {
                                const real c = rgb[3 * id + ch];
                                accum[ch] = last_alpha * last_c[ch] + (RC(1) - last_alpha) * accum[ch]; last_c[ch] = c;
                                dL_dalpha_ += (c - accum[ch]) * g[ch];
#pragma omp atomic
                                a[6 + ch] += w * g[ch];
                            }
                            accum_d = last_alpha * last_d + (RC(1) - last_alpha) * accum_d; last_d = depth[id];
                            dL_dalpha_ += (depth[id] - accum_d) * gD;
                            accum_a = last_alpha + (RC(1) - last_alpha) * accum_a;
                            dL_dalpha_ += (RC(1) - accum_a) * gA;
#pragma omp atomic
                            a[9] += w * gD;
                            dL_dalpha_ *= T;
                            last_alpha = alpha;
                            dL_dalpha_ += (-T_final / (RC(1) - alpha)) * bg_dot;
                            const real dL_dG = co[3] * dL_dalpha_;
                            const real gdx = G * dx, gdy = G * dy;
                            const real dG_ddelx = -gdx * co[0] - gdy * co[1], dG_ddely = -gdy * co[2] - gdx * co[1];
#pragma omp atomic
                            a[0] += dL_dG * dG_ddelx * (RC(0.5) * W);
#pragma omp atomic
                            a[1] += dL_dG * dG_ddely * (RC(0.5) * H);
#pragma omp atomic
                            a[2] += RC(-0.5) * gdx * dx * dL_dG;
#pragma omp atomic
                            a[3] += RC(-0.5) * gdx * dy * dL_dG;
#pragma omp atomic
                            a[4] += RC(-0.5) * gdy * dy * dL_dG;
#pragma omp atomic
                            a[5] += G * dL_dalpha_;
                        }
                    }
            }

        /* per-splat chain rule */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) {
            real dm[3] = {0, 0, 0}, ds[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0};
            const real* a = acc + 10 * (size_t)i;
            const int nbK = K;
            if (d_shs) for (int k = 0; k < nbK * 3; ++k) d_shs[(size_t)i * K * 3 + k] = 0.f;
            if (radii[i] > 0) {
                const float* p = means3D + 3 * (size_t)i;
                real pv[3]; view_point(vm, p, pv);
                real S[6], R[9], sv[3];
                cov3d(scales + 3 * (size_t)i, v->scale_modifier, rotations + 4 * (size_t)i, S, R, sv);
                Ewa e; ewa_setup(v, pv, &e);
                real Sm0[3], Sm1[3]; sym_mul(S, e.m0, Sm0); sym_mul(S, e.m1, Sm1);
                const real ca = e.m0[0] * Sm0[0] + e.m0[1] * Sm0[1] + e.m0[2] * Sm0[2] + RC(0.3);
                const real cb = e.m0[0] * Sm1[0] + e.m0[1] * Sm1[1] + e.m0[2] * Sm1[2];
                const real cc = e.m1[0] * Sm1[0] + e.m1[1] * Sm1[1] + e.m1[2] * Sm1[2] + RC(0.3);
                const real denom = ca * cc - cb * cb;
                const real d2 = RC(1) / (denom * denom + RC(0.0000001));
                real dL_da = 0, dL_db = 0, dL_dc = 0;
                if (d2 != 0) {
                    dL_da = d2 * (-cc * cc * a[2] + RC(2) * cb * cc * a[3] + (denom - ca * cc) * a[4]);
                    dL_dc = d2 * (-ca * ca * a[4] + RC(2) * ca * cb * a[3] + (denom - ca * cc) * a[2]);
                    dL_db = d2 * RC(2) * (cb * cc * a[2] - (denom + RC(2) * cb * cb) * a[3] + ca * cb * a[4]);
                }
                real dcov[6];
                const real* m0 = e.m0; const real* m1 = e.m1;
                dcov[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
                dcov[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
                dcov[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
                dcov[1] = RC(2) * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + RC(2) * m1[0] * m1[1] * dL_dc;
                dcov[2] = RC(2) * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + RC(2) * m1[0] * m1[2] * dL_dc;
                dcov[4] = RC(2) * m0[1] * m0[2] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + RC(2) * m1[1] * m1[2] * dL_dc;
                real dm0[3], dm1[3];
                for (int c = 0; c < 3; ++c) { dm0[c] = RC(2) * dL_da * Sm0[c] + dL_db * Sm1[c]; dm1[c] = RC(2) * dL_dc * Sm1[c] + dL_db * Sm0[c]; }
                real dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
                for (int c = 0; c < 3; ++c) { dJ00 += dm0[c] * vm[4 * c]; dJ02 += dm0[c] * vm[4 * c + 2]; dJ11 += dm1[c] * vm[4 * c + 1]; dJ12 += dm1[c] * vm[4 * c + 2]; }
                const real fx = v->W / (RC(2) * v->tanfovx), fy = v->H / (RC(2) * v->tanfovy);
                const real tz = RC(1) / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
                const real dtx = (e.cx ? RC(0) : RC(1)) * -fx * tz2 * dJ02, dty = (e.cy ? RC(0) : RC(1)) * -fy * tz2 * dJ12;
                const real dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (RC(2) * fx * e.tx) * tz3 * dJ02 + (RC(2) * fy * e.ty) * tz3 * dJ12 + a[9];
                dm[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
                dm[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
                dm[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
                const real hx = RC(pm[0]) * p[0] + RC(pm[4]) * p[1] + RC(pm[8]) * p[2] + pm[12];
                const real hy = RC(pm[1]) * p[0] + RC(pm[5]) * p[1] + RC(pm[9]) * p[2] + pm[13];
                const real hw_ = RC(pm[3]) * p[0] + RC(pm[7]) * p[1] + RC(pm[11]) * p[2] + pm[15];
                const real mw = RC(1) / (hw_ + RC(0.0000001)), mul1 = hx * mw * mw, mul2 = hy * mw * mw;
                dm[0] += (pm[0] * mw - pm[3] * mul1) * a[0] + (pm[1] * mw - pm[3] * mul2) * a[1];
                dm[1] += (pm[4] * mw - pm[7] * mul1) * a[0] + (pm[5] * mw - pm[7] * mul2) * a[1];
                dm[2] += (pm[8] * mw - pm[11] * mul1) * a[0] + (pm[9] * mw - pm[11] * mul2) * a[1];
                /* Sigma -> scale, quaternion */
                const real G6[6] = {dcov[0], RC(0.5) * dcov[1], RC(0.5) * dcov[2], dcov[3], RC(0.5) * dcov[4], dcov[5]};
                real dLm[9], Dm[9];
                for (int k = 0; k < 3; ++k) {
                    const real l[3] = {R[k] * sv[k], R[3 + k] * sv[k], R[6 + k] * sv[k]};
                    real gl[3]; sym_mul(G6, l, gl);
                    dLm[k] = RC(2) * gl[0]; dLm[3 + k] = RC(2) * gl[1]; dLm[6 + k] = RC(2) * gl[2];
                }
                for (int k = 0; k < 3; ++k) ds[k] = v->scale_modifier * (R[k] * dLm[k] + R[3 + k] * dLm[3 + k] + R[6 + k] * dLm[6 + k]);
                for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Dm[3 * r + k] = dLm[3 * r + k] * sv[k];
                const float* q = rotations + 4 * (size_t)i; const real r = q[0], x = q[1], y = q[2], z = q[3];
                dq[0] = RC(2) * (-z * Dm[1] + y * Dm[2] + z * Dm[3] - x * Dm[5] - y * Dm[6] + x * Dm[7]);
                dq[1] = RC(2) * (y * Dm[1] + z * Dm[2] + y * Dm[3] - RC(2) * x * Dm[4] - r * Dm[5] + z * Dm[6] + r * Dm[7] - RC(2) * x * Dm[8]);
                dq[2] = RC(2) * (RC(-2) * y * Dm[0] + x * Dm[1] + r * Dm[2] + x * Dm[3] + z * Dm[5] - r * Dm[6] + z * Dm[7] - RC(2) * y * Dm[8]);
                dq[3] = RC(2) * (RC(-2) * z * Dm[0] - r * Dm[1] + x * Dm[2] + r * Dm[3] - RC(2) * z * Dm[4] + y * Dm[5] + x * Dm[6] + y * Dm[7]);
                if (d_shs) {
                    real dc[3] = {clamped[3 * i] ? 0 : a[6], clamped[3 * i + 1] ? 0 : a[7], clamped[3 * i + 2] ? 0 : a[8]};
                    real dv[3] = {RC(p[0]) - v->campos[0], RC(p[1]) - v->campos[1], RC(p[2]) - v->campos[2]};
                    const real il = RC(1) / r_sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
                    const real d[3] = {dv[0] * il, dv[1] * il, dv[2] * il};
                    real B[16], dB[16][3]; sh_basis(v->sh_degree, d, B); sh_basis_grad(v->sh_degree, d, dB);
                    const int nb = (v->sh_degree + 1) * (v->sh_degree + 1);
                    const float* sh = shs + (size_t)i * K * 3;
                    real dd[3] = {0, 0, 0};
                    for (int k = 0; k < nb; ++k) {
                        real gk = 0;
                        for (int ch = 0; ch < 3; ++ch) { d_shs[((size_t)i * K + k) * 3 + ch] = (float)(B[k] * dc[ch]); gk += sh[3 * k + ch] * dc[ch]; }
                        for (int c = 0; c < 3; ++c) dd[c] += dB[k][c] * gk;
                    }
                    const real pr = d[0] * dd[0] + d[1] * dd[1] + d[2] * dd[2];
                    for (int c = 0; c < 3; ++c) dm[c] += (dd[c] - d[c] * pr) * il;
                }
            }
            const int vis = radii[i] > 0;
            if (d_means3D) for (int c = 0; c < 3; ++c) d_means3D[3 * (size_t)i + c] = (float)dm[c];
            if (d_means2D) { d_means2D[3 * (size_t)i] = vis ? (float)a[0] : 0.f; d_means2D[3 * (size_t)i + 1] = vis ? (float)a[1] : 0.f; d_means2D[3 * (size_t)i + 2] = 0.f; }
            if (d_opacity) d_opacity[i] = vis ? (float)a[5] : 0.f;
            if (d_scales) for (int c = 0; c < 3; ++c) d_scales[3 * (size_t)i + c] = (float)ds[c];
            if (d_rotations) for (int c = 0; c < 4; ++c) d_rotations[4 * (size_t)i + c] = (float)dq[c];
            if (d_colors) for (int c = 0; c < 3; ++c) d_colors[3 * (size_t)i + c] = vis ? (float)a[6 + c] : 0.f;
        }
        free(acc);
    }
    free(xy); free(con_o); free(rgb); free(depth); free(rect); free(clamped); free(tile_cnt); free(tile_start); free(list);
    free(final_T); free(n_contrib); free(rraw); free(pre_fragile);
#ifdef _OPENMP
    omp_set_num_threads(omp_prev_threads);
#endif
    return 0;
}

/* the entry point of rounds 1-5 (no fragile analysis): what bench.py times as cpu_baseline */
int ref_rasterize(const RefView* v, int N, const float* means3D, const float* opacities, const float* scales,
                  const float* rotations, const float* shs, const float* colors,
                  float* out_color, float* out_depth, float* out_alpha, int* radii, long long* num_rendered,
                  const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                  float* d_means3D, float* d_means2D, float* d_opacity, float* d_scales, float* d_rotations,
                  float* d_shs, float* d_colors,
                  int tile_x0, int tile_y0, int tile_x1, int tile_y1, int threads) {
    return ref_rasterize_ex(v, N, means3D, opacities, scales, rotations, shs, colors, out_color, out_depth, out_alpha, radii,
                            num_rendered, dL_dcolor, dL_ddepth, dL_dalpha, d_means3D, d_means2D, d_opacity, d_scales,
                            d_rotations, d_shs, d_colors, tile_x0, tile_y0, tile_x1, tile_y1, threads, NULL, NULL, 0.0f, NULL, 0.0f, NULL);
}
