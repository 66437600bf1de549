/*
 * CPU oracle #2: plain-C, single-precision restatement of the tile rasterizer, forward AND the
 * explicit (hand-derived) backward.  TEST INFRASTRUCTURE ONLY -- never linked, imported or called by
 * the product (splatfields_amd/, diff_gaussian_rasterization/); used by tests/, by
 * __graft_entry__.smoke() and as bench.py's timed "cpu_baseline" (kind "port").
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
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/raster_ref.c -o oracle/_build/libraster_ref.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
static const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct RefView {
    int H, W;
    float tanfovx, tanfovy, scale_modifier;
    int sh_degree, sh_coeffs;
    float viewmatrix[16], projmatrix[16], campos[3], bg[3];
} RefView;

typedef struct { float depth; int id; } ListEntry;

static int cmp_entry(const void* a, const void* b) {
    const ListEntry* x = (const ListEntry*)a; const ListEntry* y = (const ListEntry*)b;
    uint32_t dx, dy; memcpy(&dx, &x->depth, 4); memcpy(&dy, &y->depth, 4);  /* depth > 0: bits order = value order */
    if (dx != dy) return dx < dy ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

static void sh_basis(int deg, const float d[3], float B[16]) {
    const float x = d[0], y = d[1], z = d[2];
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy); B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = SH_C3[0] * y * (3.f * xx - yy); B[10] = SH_C3[1] * xy * z; B[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); B[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                B[14] = SH_C3[5] * z * (xx - yy); B[15] = SH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

/* d(basis_k)/d(direction) for k < (deg+1)^2 */
static void sh_basis_grad(int deg, const float d[3], float dB[16][3]) {
    const float x = d[0], y = d[1], z = d[2];
    memset(dB, 0, sizeof(float) * 48);
    if (deg > 0) {
        dB[1][1] = -SH_C1; dB[2][2] = SH_C1; dB[3][0] = -SH_C1;
        if (deg > 1) {
            dB[4][0] = SH_C2[0] * y; dB[4][1] = SH_C2[0] * x;
            dB[5][1] = SH_C2[1] * z; dB[5][2] = SH_C2[1] * y;
            dB[6][0] = SH_C2[2] * -2.f * x; dB[6][1] = SH_C2[2] * -2.f * y; dB[6][2] = SH_C2[2] * 4.f * z;
            dB[7][0] = SH_C2[3] * z; dB[7][2] = SH_C2[3] * x;
            dB[8][0] = SH_C2[4] * 2.f * x; dB[8][1] = SH_C2[4] * -2.f * y;
            if (deg > 2) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dB[9][0] = SH_C3[0] * 6.f * xy; dB[9][1] = SH_C3[0] * (3.f * xx - 3.f * yy);
                dB[10][0] = SH_C3[1] * yz; dB[10][1] = SH_C3[1] * xz; dB[10][2] = SH_C3[1] * xy;
                dB[11][0] = SH_C3[2] * -2.f * xy; dB[11][1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); dB[11][2] = SH_C3[2] * 8.f * yz;
                dB[12][0] = SH_C3[3] * -6.f * xz; dB[12][1] = SH_C3[3] * -6.f * yz; dB[12][2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                dB[13][0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); dB[13][1] = SH_C3[4] * -2.f * xy; dB[13][2] = SH_C3[4] * 8.f * xz;
                dB[14][0] = SH_C3[5] * 2.f * xz; dB[14][1] = SH_C3[5] * -2.f * yz; dB[14][2] = SH_C3[5] * (xx - yy);
                dB[15][0] = SH_C3[6] * (3.f * xx - 3.f * yy); dB[15][1] = SH_C3[6] * -6.f * xy;
            }
        }
    }
}

static void quat_rot(const float* q, float R[9]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

typedef struct {
    float m0[3], m1[3], tx, ty, tz; int cx, cy;
} Ewa;

static void ewa_setup(const RefView* v, const float pv[3], Ewa* e) {
    const float* vm = v->viewmatrix;
    const float fx = v->W / (2.f * v->tanfovx), fy = v->H / (2.f * v->tanfovy);
    const float limx = 1.3f * v->tanfovx, limy = 1.3f * v->tanfovy;
    const float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    e->cx = (txtz < -limx) || (txtz > limx); e->cy = (tytz < -limy) || (tytz > limy);
    e->tx = fminf(limx, fmaxf(-limx, txtz)) * pv[2]; e->ty = fminf(limy, fmaxf(-limy, tytz)) * pv[2]; e->tz = pv[2];
    const float j00 = fx / e->tz, j02 = -(fx * e->tx) / (e->tz * e->tz), j11 = fy / e->tz, j12 = -(fy * e->ty) / (e->tz * e->tz);
    for (int c = 0; c < 3; ++c) {  /* R_view[r][c] = vm[4c + r] */
        e->m0[c] = j00 * vm[4 * c + 0] + j02 * vm[4 * c + 2];
        e->m1[c] = j11 * vm[4 * c + 1] + j12 * vm[4 * c + 2];
    }
}

static void sym_mul(const float S[6], const float v[3], float out[3]) {
    out[0] = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
    out[1] = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
    out[2] = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
}

static void cov3d(const float* scale, float mod, const float* q, float S[6], float R[9], float sv[3]) {
    quat_rot(q, R);
    for (int k = 0; k < 3; ++k) sv[k] = mod * scale[k];
    const float a = sv[0] * sv[0], b = sv[1] * sv[1], c = sv[2] * sv[2];
    S[0] = R[0] * R[0] * a + R[1] * R[1] * b + R[2] * R[2] * c;
    S[1] = R[0] * R[3] * a + R[1] * R[4] * b + R[2] * R[5] * c;
    S[2] = R[0] * R[6] * a + R[1] * R[7] * b + R[2] * R[8] * c;
    S[3] = R[3] * R[3] * a + R[4] * R[4] * b + R[5] * R[5] * c;
    S[4] = R[3] * R[6] * a + R[4] * R[7] * b + R[5] * R[8] * c;
    S[5] = R[6] * R[6] * a + R[7] * R[7] * b + R[8] * R[8] * c;
}

/*
 * Forward (+ backward when dL_dcolor != NULL).  All pointers are host memory.
 * Outputs: out_color[3*H*W], out_depth[H*W], out_alpha[H*W], radii[N], *num_rendered.
 * Gradients (each may be NULL to skip): d_means3D[N*3], d_means2D[N*3], d_opacity[N], d_scales[N*3],
 * d_rotations[N*4], d_shs[N*K*3], d_colors[N*3].  dL_ddepth / dL_dalpha may be NULL.
 * Optional tile window [tile_y0, tile_y1) x [tile_x0, tile_x1) restricts the blend to a crop (bounded
 * CPU-baseline samples); pass all zeros for the whole image.
 */
int ref_rasterize(const RefView* v, int N, const float* means3D, const float* opacities, const float* scales,
                  const float* rotations, const float* shs, const float* colors,
                  float* out_color, float* out_depth, float* out_alpha, int* radii, long long* num_rendered,
                  const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                  float* d_means3D, float* d_means2D, float* d_opacity, float* d_scales, float* d_rotations,
                  float* d_shs, float* d_colors,
                  int tile_x0, int tile_y0, int tile_x1, int tile_y1, int threads) {
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
    float* xy = (float*)malloc(sizeof(float) * 2 * (size_t)(N + 1));
    float* con_o = (float*)malloc(sizeof(float) * 4 * (size_t)(N + 1));
    float* rgb = (float*)malloc(sizeof(float) * 3 * (size_t)(N + 1));
    float* depth = (float*)malloc(sizeof(float) * (size_t)(N + 1));
    int* rect = (int*)malloc(sizeof(int) * 4 * (size_t)(N + 1));
    unsigned char* clamped = (unsigned char*)calloc((size_t)(N + 1) * 3, 1);
    uint32_t* tile_cnt = (uint32_t*)calloc((size_t)gx * gy + 1, sizeof(uint32_t));

    /* ---- preprocess ---- */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        radii[i] = 0; rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
        const float* p = means3D + 3 * (size_t)i;
        const float pv[3] = {vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12], vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13],
                             vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14]};
        if (pv[2] <= 0.2f) continue;
        const float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
        const float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
        const float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        float S[6], R[9], sv[3];
        cov3d(scales + 3 * (size_t)i, v->scale_modifier, rotations + 4 * (size_t)i, S, R, sv);
        Ewa e; ewa_setup(v, pv, &e);
        float Sm0[3], Sm1[3]; sym_mul(S, e.m0, Sm0); sym_mul(S, e.m1, Sm1);
        const float a = e.m0[0] * Sm0[0] + e.m0[1] * Sm0[1] + e.m0[2] * Sm0[2] + 0.3f;
        const float b = e.m0[0] * Sm1[0] + e.m0[1] * Sm1[1] + e.m0[2] * Sm1[2];
        const float c = e.m1[0] * Sm1[0] + e.m1[1] * Sm1[1] + e.m1[2] * Sm1[2] + 0.3f;
        const float det = a * c - b * b;
        if (det == 0.0f) continue;
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (a + c);
        const float lam1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(lam1));
        const float px = ((hx * pw + 1.0f) * W - 1.0f) * 0.5f, py = ((hy * pw + 1.0f) * H - 1.0f) * 0.5f;
        int x0 = (int)((px - my_radius) / TILE), y0 = (int)((py - my_radius) / TILE);
        int x1 = (int)((px + my_radius + TILE - 1) / TILE), y1 = (int)((py + my_radius + TILE - 1) / TILE);
        x0 = x0 < 0 ? 0 : (x0 > gx ? gx : x0); x1 = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
        y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0); y1 = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        if (colors) { rgb[3 * i] = colors[3 * (size_t)i]; rgb[3 * i + 1] = colors[3 * (size_t)i + 1]; rgb[3 * i + 2] = colors[3 * (size_t)i + 2]; }
        else {
            float d[3] = {p[0] - v->campos[0], p[1] - v->campos[1], p[2] - v->campos[2]};
            const float il = 1.f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] *= il; d[1] *= il; d[2] *= il;
            float B[16]; sh_basis(v->sh_degree, d, B);
            const int nb = (v->sh_degree + 1) * (v->sh_degree + 1);
            const float* sh = shs + (size_t)i * K * 3;
            for (int ch = 0; ch < 3; ++ch) {
                float r = 0.f;
                for (int k = 0; k < nb; ++k) r += B[k] * sh[3 * k + ch];
                r += 0.5f;
                clamped[3 * i + ch] = r < 0.f;
                rgb[3 * i + ch] = r < 0.f ? 0.f : r;
            }
        }
        depth[i] = pv[2]; radii[i] = (int)my_radius; xy[2 * i] = px; xy[2 * i + 1] = py;
        con_o[4 * i] = c * det_inv; con_o[4 * i + 1] = -b * det_inv; con_o[4 * i + 2] = a * det_inv; con_o[4 * i + 3] = opacities[i];
        rect[4 * i] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
    }

    /* ---- per-tile lists: count, prefix, fill (splat order), sort by (depth, id) ---- */
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
                ListEntry le; le.depth = depth[i]; le.id = i;
                list[tile_start[t] + tile_cnt[t]++] = le;
            }
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < gx * gy; ++t) qsort(list + tile_start[t], tile_cnt[t], sizeof(ListEntry), cmp_entry);

    /* ---- forward blend ---- */
    float* final_T = (float*)malloc(sizeof(float) * (size_t)H * W);
    uint32_t* n_contrib = (uint32_t*)calloc((size_t)H * W, sizeof(uint32_t));
    const size_t hw = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = tile_y0; ty < tile_y1; ++ty)
        for (int tx = tile_x0; tx < tile_x1; ++tx) {
            const int t = ty * gx + tx;
            const ListEntry* L = list + tile_start[t]; const uint32_t n = tile_cnt[t];
            for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; ++py)
                for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; ++px) {
                    float T = 1.f, C[3] = {0.f, 0.f, 0.f}, D = 0.f; uint32_t last = 0;
                    for (uint32_t j = 0; j < n; ++j) {
                        const int id = L[j].id;
                        const float dx = xy[2 * id] - (float)px, dy = xy[2 * id + 1] - (float)py;
                        const float* co = con_o + 4 * (size_t)id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.f) continue;
                        const float alpha = fminf(0.99f, co[3] * expf(power));
                        if (alpha < 1.f / 255.f) continue;
                        const float test_T = T * (1.f - alpha);
                        if (test_T < 0.0001f) break;
                        for (int ch = 0; ch < 3; ++ch) C[ch] += rgb[3 * id + ch] * alpha * T;
                        D += depth[id] * alpha * T;
                        T = test_T; last = j + 1;
                    }
                    const size_t pix = (size_t)py * W + px;
                    final_T[pix] = T; n_contrib[pix] = last;
                    for (int ch = 0; ch < 3; ++ch) out_color[ch * hw + pix] = C[ch] + T * v->bg[ch];
                    out_depth[pix] = D;
                    if (out_alpha) out_alpha[pix] = 1.f - T;
                }
        }

    /* ---- backward ---- */
    if (dL_dcolor) {
        /* screen-space accumulators: mean2D(2), conic(3: x, y(half), w), opacity, colour(3), depth */
        float* acc = (float*)calloc((size_t)(N + 1) * 10, sizeof(float));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
        for (int ty = tile_y0; ty < tile_y1; ++ty)
            for (int tx = tile_x0; tx < tile_x1; ++tx) {
                const int t = ty * gx + tx;
                const ListEntry* L = list + tile_start[t];
                for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; ++py)
                    for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; ++px) {
                        const size_t pix = (size_t)py * W + px;
                        const float T_final = final_T[pix];
                        float T = T_final;
                        const float g[3] = {dL_dcolor[pix], dL_dcolor[hw + pix], dL_dcolor[2 * hw + pix]};
                        const float gD = dL_ddepth ? dL_ddepth[pix] : 0.f, gA = dL_dalpha ? dL_dalpha[pix] : 0.f;
                        const float bg_dot = v->bg[0] * g[0] + v->bg[1] * g[1] + v->bg[2] * g[2];
                        float accum[3] = {0.f, 0.f, 0.f}, accum_d = 0.f, accum_a = 0.f, last_alpha = 0.f, last_c[3] = {0.f, 0.f, 0.f}, last_d = 0.f;
                        for (int j = (int)n_contrib[pix] - 1; j >= 0; --j) {
                            const int id = L[j].id;
                            const float dx = xy[2 * id] - (float)px, dy = xy[2 * id + 1] - (float)py;
                            const float* co = con_o + 4 * (size_t)id;
                            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                            if (power > 0.f) continue;
                            const float G = expf(power);
                            const float alpha = fminf(0.99f, co[3] * G);
                            if (alpha < 1.f / 255.f) continue;
                            T = T / (1.f - alpha);
                            const float w = alpha * T;
                            float dL_dalpha_ = 0.f;
                            float* a = acc + 10 * (size_t)id;
                            for (int ch = 0; ch < 3; ++ch) {
                                const float c = rgb[3 * id + ch];
                                accum[ch] = last_alpha * last_c[ch] + (1.f - last_alpha) * accum[ch]; last_c[ch] = c;
                                dL_dalpha_ += (c - accum[ch]) * g[ch];
#pragma omp atomic
                                a[6 + ch] += w * g[ch];
                            }
                            accum_d = last_alpha * last_d + (1.f - last_alpha) * accum_d; last_d = depth[id];
                            dL_dalpha_ += (depth[id] - accum_d) * gD;
                            accum_a = last_alpha + (1.f - last_alpha) * accum_a;
                            dL_dalpha_ += (1.f - accum_a) * gA;
#pragma omp atomic
                            a[9] += w * gD;
                            dL_dalpha_ *= T;
                            last_alpha = alpha;
                            dL_dalpha_ += (-T_final / (1.f - alpha)) * bg_dot;
                            const float dL_dG = co[3] * dL_dalpha_;
                            const float gdx = G * dx, gdy = G * dy;
                            const float dG_ddelx = -gdx * co[0] - gdy * co[1], dG_ddely = -gdy * co[2] - gdx * co[1];
#pragma omp atomic
                            a[0] += dL_dG * dG_ddelx * (0.5f * W);
#pragma omp atomic
                            a[1] += dL_dG * dG_ddely * (0.5f * H);
#pragma omp atomic
                            a[2] += -0.5f * gdx * dx * dL_dG;
#pragma omp atomic
                            a[3] += -0.5f * gdx * dy * dL_dG;
#pragma omp atomic
                            a[4] += -0.5f * gdy * dy * dL_dG;
#pragma omp atomic
                            a[5] += G * dL_dalpha_;
                        }
                    }
            }

        /* per-splat chain rule */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) {
            float dm[3] = {0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
            const float* a = acc + 10 * (size_t)i;
            const int nbK = K;
            if (d_shs) for (int k = 0; k < nbK * 3; ++k) d_shs[(size_t)i * K * 3 + k] = 0.f;
            if (radii[i] > 0) {
                const float* p = means3D + 3 * (size_t)i;
                const float pv[3] = {vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12], vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13],
                                     vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14]};
                float S[6], R[9], sv[3];
                cov3d(scales + 3 * (size_t)i, v->scale_modifier, rotations + 4 * (size_t)i, S, R, sv);
                Ewa e; ewa_setup(v, pv, &e);
                float Sm0[3], Sm1[3]; sym_mul(S, e.m0, Sm0); sym_mul(S, e.m1, Sm1);
                const float ca = e.m0[0] * Sm0[0] + e.m0[1] * Sm0[1] + e.m0[2] * Sm0[2] + 0.3f;
                const float cb = e.m0[0] * Sm1[0] + e.m0[1] * Sm1[1] + e.m0[2] * Sm1[2];
                const float cc = e.m1[0] * Sm1[0] + e.m1[1] * Sm1[1] + e.m1[2] * Sm1[2] + 0.3f;
                const float denom = ca * cc - cb * cb;
                const float d2 = 1.f / (denom * denom + 0.0000001f);
                float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
                if (d2 != 0.f) {
                    dL_da = d2 * (-cc * cc * a[2] + 2.f * cb * cc * a[3] + (denom - ca * cc) * a[4]);
                    dL_dc = d2 * (-ca * ca * a[4] + 2.f * ca * cb * a[3] + (denom - ca * cc) * a[2]);
                    dL_db = d2 * 2.f * (cb * cc * a[2] - (denom + 2.f * cb * cb) * a[3] + ca * cb * a[4]);
                }
                float dcov[6];
                const float* m0 = e.m0; const float* m1 = e.m1;
                dcov[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
                dcov[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
                dcov[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
                dcov[1] = 2.f * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2.f * m1[0] * m1[1] * dL_dc;
                dcov[2] = 2.f * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2.f * m1[0] * m1[2] * dL_dc;
                dcov[4] = 2.f * m0[1] * m0[2] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2.f * m1[1] * m1[2] * dL_dc;
                float dm0[3], dm1[3];
                for (int c = 0; c < 3; ++c) { dm0[c] = 2.f * dL_da * Sm0[c] + dL_db * Sm1[c]; dm1[c] = 2.f * dL_dc * Sm1[c] + dL_db * Sm0[c]; }
                float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
                for (int c = 0; c < 3; ++c) { dJ00 += dm0[c] * vm[4 * c]; dJ02 += dm0[c] * vm[4 * c + 2]; dJ11 += dm1[c] * vm[4 * c + 1]; dJ12 += dm1[c] * vm[4 * c + 2]; }
                const float fx = v->W / (2.f * v->tanfovx), fy = v->H / (2.f * v->tanfovy);
                const float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
                const float dtx = (e.cx ? 0.f : 1.f) * -fx * tz2 * dJ02, dty = (e.cy ? 0.f : 1.f) * -fy * tz2 * dJ12;
                const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * e.tx) * tz3 * dJ02 + (2.f * fy * e.ty) * tz3 * dJ12 + a[9];
                dm[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
                dm[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
                dm[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
                const float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
                const float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
                const float hw_ = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
                const float mw = 1.f / (hw_ + 0.0000001f), mul1 = hx * mw * mw, mul2 = hy * mw * mw;
                dm[0] += (pm[0] * mw - pm[3] * mul1) * a[0] + (pm[1] * mw - pm[3] * mul2) * a[1];
                dm[1] += (pm[4] * mw - pm[7] * mul1) * a[0] + (pm[5] * mw - pm[7] * mul2) * a[1];
                dm[2] += (pm[8] * mw - pm[11] * mul1) * a[0] + (pm[9] * mw - pm[11] * mul2) * a[1];
                /* Sigma -> scale, quaternion */
                const float G6[6] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], dcov[3], 0.5f * dcov[4], dcov[5]};
                float dLm[9], Dm[9];
                for (int k = 0; k < 3; ++k) {
                    const float l[3] = {R[k] * sv[k], R[3 + k] * sv[k], R[6 + k] * sv[k]};
                    float gl[3]; sym_mul(G6, l, gl);
                    dLm[k] = 2.f * gl[0]; dLm[3 + k] = 2.f * gl[1]; dLm[6 + k] = 2.f * gl[2];
                }
                for (int k = 0; k < 3; ++k) ds[k] = v->scale_modifier * (R[k] * dLm[k] + R[3 + k] * dLm[3 + k] + R[6 + k] * dLm[6 + k]);
                for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Dm[3 * r + k] = dLm[3 * r + k] * sv[k];
                const float* q = rotations + 4 * (size_t)i; const float r = q[0], x = q[1], y = q[2], z = q[3];
                dq[0] = 2.f * (-z * Dm[1] + y * Dm[2] + z * Dm[3] - x * Dm[5] - y * Dm[6] + x * Dm[7]);
                dq[1] = 2.f * (y * Dm[1] + z * Dm[2] + y * Dm[3] - 2.f * x * Dm[4] - r * Dm[5] + z * Dm[6] + r * Dm[7] - 2.f * x * Dm[8]);
                dq[2] = 2.f * (-2.f * y * Dm[0] + x * Dm[1] + r * Dm[2] + x * Dm[3] + z * Dm[5] - r * Dm[6] + z * Dm[7] - 2.f * y * Dm[8]);
                dq[3] = 2.f * (-2.f * z * Dm[0] - r * Dm[1] + x * Dm[2] + r * Dm[3] - 2.f * z * Dm[4] + y * Dm[5] + x * Dm[6] + y * Dm[7]);
                if (d_shs) {
                    float dc[3] = {clamped[3 * i] ? 0.f : a[6], clamped[3 * i + 1] ? 0.f : a[7], clamped[3 * i + 2] ? 0.f : a[8]};
                    float dv[3] = {p[0] - v->campos[0], p[1] - v->campos[1], p[2] - v->campos[2]};
                    const float il = 1.f / sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
                    const float d[3] = {dv[0] * il, dv[1] * il, dv[2] * il};
                    float B[16], dB[16][3]; sh_basis(v->sh_degree, d, B); sh_basis_grad(v->sh_degree, d, dB);
                    const int nb = (v->sh_degree + 1) * (v->sh_degree + 1);
                    const float* sh = shs + (size_t)i * K * 3;
                    float dd[3] = {0.f, 0.f, 0.f};
                    for (int k = 0; k < nb; ++k) {
                        float gk = 0.f;
                        for (int ch = 0; ch < 3; ++ch) { d_shs[((size_t)i * K + k) * 3 + ch] = B[k] * dc[ch]; gk += sh[3 * k + ch] * dc[ch]; }
                        for (int c = 0; c < 3; ++c) dd[c] += dB[k][c] * gk;
                    }
                    const float pr = d[0] * dd[0] + d[1] * dd[1] + d[2] * dd[2];
                    for (int c = 0; c < 3; ++c) dm[c] += (dd[c] - d[c] * pr) * il;
                }
            }
            const int vis = radii[i] > 0;
            if (d_means3D) for (int c = 0; c < 3; ++c) d_means3D[3 * (size_t)i + c] = dm[c];
            if (d_means2D) { d_means2D[3 * (size_t)i] = vis ? a[0] : 0.f; d_means2D[3 * (size_t)i + 1] = vis ? a[1] : 0.f; d_means2D[3 * (size_t)i + 2] = 0.f; }
            if (d_opacity) d_opacity[i] = vis ? a[5] : 0.f;
            if (d_scales) for (int c = 0; c < 3; ++c) d_scales[3 * (size_t)i + c] = ds[c];
            if (d_rotations) for (int c = 0; c < 4; ++c) d_rotations[4 * (size_t)i + c] = dq[c];
            if (d_colors) for (int c = 0; c < 3; ++c) d_colors[3 * (size_t)i + c] = vis ? a[6 + c] : 0.f;
        }
        free(acc);
    }
    free(xy); free(con_o); free(rgb); free(depth); free(rect); free(clamped); free(tile_cnt); free(tile_start); free(list);
    free(final_T); free(n_contrib);
#ifdef _OPENMP
    omp_set_num_threads(omp_prev_threads);
#endif
    return 0;
}
