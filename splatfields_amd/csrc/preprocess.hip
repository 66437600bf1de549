// Per-splat stages of the rasterizer for gfx950: forward preprocess (cull, 3D->2D covariance
// projection, conic, radius, tile rectangle, SH->RGB, per-tile instance counting), the
// near-plane visibility mark, and the backward preprocess (segmented reduction of the per-instance
// gradient slots + chain rule to means3D / scales / rotations / SH / opacity).
//
// Semantics: SURVEY.md Appendix A ("Forward preprocess", "Backward preprocess"), i.e. the
// published 3DGS rasterizer that reference gaussian_renderer/__init__.py:94-102 calls.
// One thread per splat, 256-thread workgroups; all traffic is per-splat streaming (HBM-bound).
#include "kernels.h"
#include "expand.h"
#include "sh_stage.h"

namespace sr {

struct Sym3 { float xx, xy, xz, yy, yz, zz; };

__device__ __forceinline__ void quat_to_rot(const float4 q, float R[9]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;  // (r,x,y,z) used as given
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R diag(s^2) R^T
__device__ __forceinline__ Sym3 cov3d_from(const float3 s, const float R[9]) {
    const float a = s.x * s.x, b = s.y * s.y, c = s.z * s.z;
    Sym3 S;
    S.xx = R[0] * R[0] * a + R[1] * R[1] * b + R[2] * R[2] * c;
    S.xy = R[0] * R[3] * a + R[1] * R[4] * b + R[2] * R[5] * c;
    S.xz = R[0] * R[6] * a + R[1] * R[7] * b + R[2] * R[8] * c;
    S.yy = R[3] * R[3] * a + R[4] * R[4] * b + R[5] * R[5] * c;
    S.yz = R[3] * R[6] * a + R[4] * R[7] * b + R[5] * R[8] * c;
    S.zz = R[6] * R[6] * a + R[7] * R[7] * b + R[8] * R[8] * c;
    return S;
}

__device__ __forceinline__ float3 sym_mul(const Sym3& S, const float3 v) {
    return make_float3(S.xx * v.x + S.xy * v.y + S.xz * v.z,
                       S.xy * v.x + S.yy * v.y + S.yz * v.z,
                       S.xz * v.x + S.yz * v.y + S.zz * v.z);
}
__device__ __forceinline__ float dot3(const float3 a, const float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Shared by forward and backward: the two rows of M = J * R_view, and the (clamped) t used in J.
struct Ewa {
    float3 m0, m1;     // rows of the 2x3 matrix mapping world covariance to screen covariance
    float tx, ty, tz;  // view-space position with the +-1.3 tanfov clamp applied to x/z, y/z
    bool clamp_x, clamp_y;
};

__device__ __forceinline__ Ewa ewa_setup(const float3 pv, const ViewK& v, const float* vm) {
    Ewa e;
    const float limx = kClampFov * v.tanfovx, limy = kClampFov * v.tanfovy;
    const float txtz = pv.x / pv.z, tytz = pv.y / pv.z;
    e.clamp_x = (txtz < -limx) || (txtz > limx);
    e.clamp_y = (tytz < -limy) || (tytz > limy);
    e.tx = fminf(limx, fmaxf(-limx, txtz)) * pv.z;
    e.ty = fminf(limy, fmaxf(-limy, tytz)) * pv.z;
    e.tz = pv.z;
    const float j00 = v.focal_x / e.tz, j02 = -(v.focal_x * e.tx) / (e.tz * e.tz);
    const float j11 = v.focal_y / e.tz, j12 = -(v.focal_y * e.ty) / (e.tz * e.tz);
    // R_view[r][c] = vm[4c + r]
    const float3 r0 = make_float3(vm[0], vm[4], vm[8]);
    const float3 r1 = make_float3(vm[1], vm[5], vm[9]);
    const float3 r2 = make_float3(vm[2], vm[6], vm[10]);
    e.m0 = make_float3(j00 * r0.x + j02 * r2.x, j00 * r0.y + j02 * r2.y, j00 * r0.z + j02 * r2.z);
    e.m1 = make_float3(j11 * r1.x + j12 * r2.x, j11 * r1.y + j12 * r2.y, j11 * r1.z + j12 * r2.z);
    return e;
}

__device__ __forceinline__ void sh_basis(int deg, const float3 d, float B[16]) {
    const float x = d.x, y = d.y, z = d.z;
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = SH_C2_0 * xy; B[5] = SH_C2_1 * yz; B[6] = SH_C2_2 * (2.f * zz - xx - yy);
            B[7] = SH_C2_3 * xz; B[8] = SH_C2_4 * (xx - yy);
            if (deg > 2) {
                B[9] = SH_C3_0 * y * (3.f * xx - yy); B[10] = SH_C3_1 * xy * z;
                B[11] = SH_C3_2 * y * (4.f * zz - xx - yy); B[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                B[13] = SH_C3_4 * x * (4.f * zz - xx - yy); B[14] = SH_C3_5 * z * (xx - yy);
                B[15] = SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

// sum_k gk[k] * d basis_k / d(unit direction): with gk[k] = sh[k][c] this is d colour_c / d direction (forward), with
// gk[k] = sh[k] . dL/dcolour it is dL/d direction (sr_sh_backward keeps its own copy for the multi-view rebuild).
__device__ __forceinline__ float3 sh_dir_gradient(int deg, const float3 d, const float gk[16]) {
    float3 dd_ = make_float3(0.f, 0.f, 0.f);
    if (deg > 0) {
        const float x = d.x, y = d.y, z = d.z;
        dd_.x += -SH_C1 * gk[3]; dd_.y += -SH_C1 * gk[1]; dd_.z += SH_C1 * gk[2];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dd_.x += SH_C2_0 * y * gk[4] + SH_C2_2 * -2.f * x * gk[6] + SH_C2_3 * z * gk[7] + SH_C2_4 * 2.f * x * gk[8];
            dd_.y += SH_C2_0 * x * gk[4] + SH_C2_1 * z * gk[5] + SH_C2_2 * -2.f * y * gk[6] + SH_C2_4 * -2.f * y * gk[8];
            dd_.z += SH_C2_1 * y * gk[5] + SH_C2_2 * 4.f * z * gk[6] + SH_C2_3 * x * gk[7];
            if (deg > 2) {
                dd_.x += SH_C3_0 * 6.f * xy * gk[9] + SH_C3_1 * yz * gk[10] + SH_C3_2 * -2.f * xy * gk[11] +
                         SH_C3_3 * -6.f * xz * gk[12] + SH_C3_4 * (4.f * zz - 3.f * xx - yy) * gk[13] +
                         SH_C3_5 * 2.f * xz * gk[14] + SH_C3_6 * (3.f * xx - 3.f * yy) * gk[15];
                dd_.y += SH_C3_0 * (3.f * xx - 3.f * yy) * gk[9] + SH_C3_1 * xz * gk[10] +
                         SH_C3_2 * (4.f * zz - xx - 3.f * yy) * gk[11] + SH_C3_3 * -6.f * yz * gk[12] +
                         SH_C3_4 * -2.f * xy * gk[13] + SH_C3_5 * -2.f * yz * gk[14] + SH_C3_6 * -6.f * xy * gk[15];
                dd_.z += SH_C3_1 * xy * gk[10] + SH_C3_2 * 8.f * yz * gk[11] +
                         SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy) * gk[12] + SH_C3_4 * 8.f * xz * gk[13] +
                         SH_C3_5 * (xx - yy) * gk[14];
            }
        }
    }
    return dd_;
}

// The optimiser's raw parameters -> what the rasterizer consumes (reference scene/gaussian_model.py:64-86), applied on load
// in both preprocess kernels when the SR_RAW_* bit is set.  `q_norm` returns |q| of the raw quaternion (1 if not raw).
__device__ __forceinline__ void activate_inputs(int raw, float3& sc, float& opac, float4& q, float& q_norm) {
    if (raw & SR_RAW_SCALES) sc = make_float3(expf(sc.x), expf(sc.y), expf(sc.z));
    if (raw & SR_RAW_OPACITY) opac = 1.0f / (1.0f + expf(-opac));
    q_norm = 1.0f;
    if (raw & SR_RAW_ROTATIONS) {
        q_norm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);  // F.normalize's eps
        const float inv = 1.0f / q_norm;
        q = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    }
}

// ------------------------------------------------------------------------------------------
// Forward preprocess
// ------------------------------------------------------------------------------------------
template <bool STAGE_SH, bool COUNT_ATOMIC>
__global__ void __launch_bounds__(kBlock) k_preprocess(const ViewK v, const SplatsK s, const Geom g, int* __restrict__ radii) {
    __shared__ uint32_t s_off[COUNT_ATOMIC ? kBlock + 1 : 1];
    __shared__ ushort4 s_rect[COUNT_ATOMIC ? kBlock : 1];
    __shared__ float4 s_r0[COUNT_ATOMIC ? kBlock : 1], s_r1[COUNT_ATOMIC ? kBlock : 1];   // ellipses for the tile_reached test
    __shared__ uint32_t s_scan[8];
    // The 64-byte records leave through LDS, quarter-major with a padded row: consecutive lanes store consecutive 16-byte pieces
    // of the workgroup's contiguous 16 KB -- whole lines per wave-instruction.  (Rounds 1-4: every thread stored its own four
    // quarters, 16-byte pieces at a 64-byte stride, and the L2 had to assemble every line from four partial writes.)
    // float4 per quarter row.  Lane l of the transposed read takes quarter l & 3 of record l >> 2: eight consecutive lanes -- one
    // 128-byte LDS pass -- hit the 16-byte bank groups (q R + r) mod 8, q = 0..3, r = 0..1, which are all different iff
    // R = 2 (mod 8).  (Round 5 padded by 1: lanes 1..3 of a record shared their groups with lanes 4..6 of the next one.)
    constexpr int kRecRow = kBlock + 2;
    constexpr int kRecF4 = 4 * kRecRow;
    __shared__ float4 s_sh[STAGE_SH ? kBlock * kShRowF4 : kRecF4];   // SH staging first; the records' transpose afterwards
    static_assert(!STAGE_SH || kBlock * kShRowF4 >= kRecF4, "the record transpose reuses the SH staging area");

    const int idx = blockIdx.x * kBlock + threadIdx.x;
    const float* vm = v.viewmatrix;
    const float* pm = v.projmatrix;
    uint32_t touched = 0, depth_bits = 0, mask16 = 0;
    ushort4 rect = make_ushort4(0, 0, 0, 0);
    float4 ell0 = make_float4(0.f, 0.f, -1.f, 0.f), ell1 = ell0, rec2 = ell0;
    uint32_t rect_bits = 0;
    bool write_rec = false;

    // issue this splat's own loads first, then the staged SH block: everything is in flight together
    float3 p = make_float3(0.f, 0.f, 0.f), sc_in = make_float3(0.f, 0.f, 0.f);
    float4 q_in = make_float4(1.f, 0.f, 0.f, 0.f);
    float opac_in = 0.f;
    float3 col_in = make_float3(0.f, 0.f, 0.f);
    if (idx < s.N) {
        p = make_float3(s.means3D[3 * idx], s.means3D[3 * idx + 1], s.means3D[3 * idx + 2]);
        opac_in = s.opacities[idx];
        if (!s.cov3D) {
            q_in = reinterpret_cast<const float4*>(s.rotations)[idx];
            sc_in = make_float3(s.scales[3 * idx], s.scales[3 * idx + 1], s.scales[3 * idx + 2]);
        }
        // (round 5) precomputed colours too: requested where they were used -- behind the projection -- they were a second memory
        // round trip in the middle of every workgroup's life (-0.6 us).  Only in the variants without SH staging: the same three
        // registers held across the staging of the SH path cost that kernel 3 us (measured).
        if constexpr (!STAGE_SH) { if (s.colors) col_in = make_float3(s.colors[3 * idx], s.colors[3 * idx + 1], s.colors[3 * idx + 2]); }
    }
    if constexpr (STAGE_SH) {
        const size_t first = (size_t)blockIdx.x * kBlock;
        if (s.shs_rest) stage_sh_in_split(s_sh, s.shs, s.shs_rest, first, min(kBlock, s.N - (int)first));
        else stage_sh_in(s_sh, s.shs, first, min(kBlock, s.N - (int)first));
        __syncthreads();
    }
    if (s.raw) { float qn; activate_inputs(s.raw, sc_in, opac_in, q_in, qn); }

    if (idx < s.N) {
        int out_radius = 0;
        uint8_t flags = 0;
        const float3 pv = make_float3(vm[0] * p.x + vm[4] * p.y + vm[8] * p.z + vm[12],
                                      vm[1] * p.x + vm[5] * p.y + vm[9] * p.z + vm[13],
                                      vm[2] * p.x + vm[6] * p.y + vm[10] * p.z + vm[14]);
        if (pv.z > kNearCullZ) {
            const float hx_ = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
            const float hy_ = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
            const float hw_ = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
            const float inv_w = 1.0f / (hw_ + kWEps);
            const float ndc_x = hx_ * inv_w, ndc_y = hy_ * inv_w;

            Sym3 S;
            if (s.cov3D) {
                const float* c = s.cov3D + 6 * (size_t)idx;
                S.xx = c[0]; S.xy = c[1]; S.xz = c[2]; S.yy = c[3]; S.yz = c[4]; S.zz = c[5];
            } else {
                float R[9];
                quat_to_rot(q_in, R);
                const float m = v.scale_modifier;
                S = cov3d_from(make_float3(m * sc_in.x, m * sc_in.y, m * sc_in.z), R);
            }
            const Ewa e = ewa_setup(pv, v, vm);
            if (e.clamp_x) flags |= kFlagClampTx;
            if (e.clamp_y) flags |= kFlagClampTy;
            const float3 Sm0 = sym_mul(S, e.m0), Sm1 = sym_mul(S, e.m1);
            const float a = dot3(e.m0, Sm0) + kDilation;
            const float b = dot3(e.m0, Sm1);
            const float c = dot3(e.m1, Sm1) + kDilation;
            const float det = a * c - b * b;
            if (det != 0.0f) {
                const float det_inv = 1.0f / det;
                const float mid = 0.5f * (a + c);
                const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
                const float my_radius = ceilf(3.0f * sqrtf(lam));
                const float px = ((ndc_x + 1.0f) * v.W - 1.0f) * 0.5f;
                const float py = ((ndc_y + 1.0f) * v.H - 1.0f) * 0.5f;
                // upstream tile rectangle (3-sigma bounding square), max exclusive
                const int uxmin = min(v.gx, max(0, (int)((px - my_radius) / kTile)));
                const int uymin = min(v.gy, max(0, (int)((py - my_radius) / kTile)));
                const int uxmax = min(v.gx, max(0, (int)((px + my_radius + kTile - 1) / kTile)));
                const int uymax = min(v.gy, max(0, (int)((py + my_radius + kTile - 1) / kTile)));
                if ((uxmax - uxmin) * (uymax - uymin) > 0) {
                    out_radius = (int)my_radius;
                    const float opac = opac_in;
                    // exact support of alpha >= 1/255:  d^T Q d <= 2 ln(255 o)  ->  |dx| <= sqrt(tau * a)
                    float ex = -1.0f, ey = -1.0f;
                    const float tau = 2.0f * __logf(255.0f * opac);
                    if (tau > 0.0f) {
                        ex = sqrtf(tau * a) * 1.002f + 0.02f;
                        ey = sqrtf(tau * c) * 1.002f + 0.02f;
                    }
                    int xmin = 0, ymin = 0, xmax = 0, ymax = 0;
                    if (ex >= 0.0f) {
                        // tiles containing a pixel centre within the support box, intersected with upstream's rect
                        xmin = max(uxmin, (int)floorf(fmaxf(px - ex, 0.0f) / kTile));
                        ymin = max(uymin, (int)floorf(fmaxf(py - ey, 0.0f) / kTile));
                        xmax = (px + ex < 0.0f) ? 0 : min(uxmax, (int)floorf((px + ex) / kTile) + 1);
                        ymax = (py + ey < 0.0f) ? 0 : min(uymax, (int)floorf((py + ey) / kTile) + 1);
                        if (xmax <= xmin || ymax <= ymin) { xmin = ymin = xmax = ymax = 0; }
                    }
                    rect = make_ushort4((unsigned short)xmin, (unsigned short)ymin, (unsigned short)xmax, (unsigned short)ymax);
                    touched = (uint32_t)((xmax - xmin) * (ymax - ymin));

                    float3 rgb;
                    if (s.colors) {
                        if constexpr (STAGE_SH) rgb = make_float3(s.colors[3 * idx], s.colors[3 * idx + 1], s.colors[3 * idx + 2]);
                        else rgb = col_in;
                    } else {
                        const float* cp = v.campos;
                        float3 d = make_float3(p.x - cp[0], p.y - cp[1], p.z - cp[2]);
                        const float inv_len = 1.0f / sqrtf(dot3(d, d));
                        d.x *= inv_len; d.y *= inv_len; d.z *= inv_len;
                        float B[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) B[k] = 0.f;   // bands above the active degree contribute nothing
                        sh_basis(v.sh_degree, d, B);
                        const int nb = (v.sh_degree + 1) * (v.sh_degree + 1);
                        // This splat's coefficients into registers ONCE (float j = 3k + c; zero beyond the active bands).
                        // Staged, one source tensor: the thread's LDS row is 13 float4, read as twelve 16-byte pieces -- the odd
                        // row length makes those conflict-free.  (Rounds 1-5 read the row float by float, ~90 ds_read_b32 per
                        // thread at a stride of 52 floats = 20 banks: four lanes on every bank, 9.7 M bank-conflict cycles per
                        // launch -- SQ_LDS_BANK_CONFLICT, profiles/r05_v3_pmc.json -- for 1.6 M LDS instructions.)
                        float shc[48];
                        if (STAGE_SH && !s.shs_rest) {
                            const float4* row = s_sh + kShRowF4 * (int)threadIdx.x;
#pragma unroll
                            for (int it = 0; it < 12; ++it) {
                                const float4 t4 = row[it];
                                shc[4 * it] = t4.x; shc[4 * it + 1] = t4.y; shc[4 * it + 2] = t4.z; shc[4 * it + 3] = t4.w;
                            }
#pragma unroll
                            for (int k = 0; k < 16; ++k) if (k >= nb) { shc[3 * k] = 0.f; shc[3 * k + 1] = 0.f; shc[3 * k + 2] = 0.f; }
                        } else {
                            // two-tensor staging (odd strides of 3 and 45 floats: conflict-free 4-byte reads) or straight from memory:
                            // sh_lo[j] for the dc triple, sh_hi[j] for the rest
                            float *sh_lo, *sh_hi;
                            if (STAGE_SH) sh_row_pointers(s_sh, true, threadIdx.x, sh_lo, sh_hi);
                            else sh_lo = sh_hi = const_cast<float*>(s.shs) + (size_t)idx * v.sh_coeffs * 3;
#pragma unroll
                            for (int k = 0; k < 16; ++k) {
                                const float* src = k == 0 ? sh_lo : sh_hi + 3 * k;
                                const bool on = k < nb;
                                shc[3 * k] = on ? src[0] : 0.f; shc[3 * k + 1] = on ? src[1] : 0.f; shc[3 * k + 2] = on ? src[2] : 0.f;
                            }
                        }
                        rgb = make_float3(0.f, 0.f, 0.f);
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            rgb.x = fmaf(B[k], shc[3 * k], rgb.x); rgb.y = fmaf(B[k], shc[3 * k + 1], rgb.y); rgb.z = fmaf(B[k], shc[3 * k + 2], rgb.z);
                        }
                        if (v.sh_degree > 0 && !(s.raw & SR_FORWARD_ONLY)) {
                            // d colour / d direction while the coefficients are at hand (36 bytes per splat instead of the
                            // backward re-reading 192)
                            float jac[9];
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) {
                                float gk[16];
#pragma unroll
                                for (int k = 0; k < 16; ++k) gk[k] = shc[3 * k + ch];
                                const float3 j = sh_dir_gradient(v.sh_degree, d, gk);
                                jac[3 * ch] = j.x; jac[3 * ch + 1] = j.y; jac[3 * ch + 2] = j.z;
                            }
                            // two 16-byte stores + one 4-byte store per splat (three coalesced streams)
                            float4* jq = reinterpret_cast<float4*>(g.dcol_ddir);
                            store_stream(jq + idx, make_float4(jac[0], jac[1], jac[2], jac[3]));   // whole lines, read next in the backward
                            store_stream(jq + (size_t)s.N + idx, make_float4(jac[4], jac[5], jac[6], jac[7]));
                            g.dcol_ddir[(size_t)8 * s.N + idx] = jac[8];
                        }
                        rgb.x += 0.5f; rgb.y += 0.5f; rgb.z += 0.5f;
                        if (rgb.x < 0.f) { flags |= kFlagClampR; rgb.x = 0.f; }
                        if (rgb.y < 0.f) { flags |= kFlagClampG; rgb.y = 0.f; }
                        if (rgb.z < 0.f) { flags |= kFlagClampB; rgb.z = 0.f; }
                    }
                    ell0 = make_float4(px, py, tau > 0.0f ? tau * kLog2e : -1.0f, pv.z);
                    depth_bits = __float_as_uint(pv.z);
                    // exponent factors (common.h, pair_alpha_unclamped): all well conditioned, c >= 0.3 by the dilation
                    const float inv_c = 1.0f / c;
                    ell1 = make_float4(sqrtf(0.5f * kLog2e * c * det_inv), -b * inv_c, sqrtf(0.5f * kLog2e * inv_c),
                                       opac > 0.0f ? -__log2f(opac) : 0.0f);
                    rec2 = make_float4(rgb.x, rgb.y, rgb.z, pv.z);  // the blend kernels read the depth with the colour: one 16-byte LDS read
                    rect_bits = (uint32_t)xmin | ((uint32_t)ymin << 16);
                    write_rec = true;   // the record is stored behind the sub-batch scan below (its fourth quarter needs the scan)
                }
            }
        }
        radii[idx] = out_radius;
        // small rectangles: which of their tiles the ellipse reaches, decided here once (common.h, kMaskTiles)
        if (touched > 0u && touched <= kMaskTiles) {
            uint32_t mask = (1u << touched) - 1u;
            if (touched >= kMaskMinTiles) {
                float4 e0, e1;
                tile_test_prepare(ell0, ell1, e0, e1);
                mask = 0u;
                uint32_t k = 0;
                for (uint32_t ty = rect.y; ty < rect.w; ++ty)
                    for (uint32_t tx = rect.x; tx < rect.z; ++tx, ++k)
                        if (tile_reached(e0, e1, tx, ty)) mask |= 1u << k;
            }
            mask16 = kRectMasked | mask;
        }
        g.rect[idx] = rect_pack(rect, mask16);
        g.touched[idx] = touched;
        g.depth_bits[idx] = depth_bits;
        g.flags[idx] = flags;
    }

    // instances emitted by this 256-splat sub-batch (input of the global exclusive scan)
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(touched, s_scan, total);
    if (threadIdx.x == 0) g.block_sums[blockIdx.x] = total;
    {
        // (block_exclusive_scan's barriers lie behind every thread's last read of the SH rows: the area is free)
        // The record is always written whole (a line with a 16-byte hole leaves the L2 as a masked partial write, which costs
        // more than the 16 bytes).  q3 = (tile rect origin, rect width, first instance of the splat RELATIVE to its 256-splat
        // sub-batch, 0): the backward blend derives a (splat, tile) pair's instance index from it as block_offsets[splat >> 8] +
        // q3.z + position of the tile in the rect -- a 16 KB table that lives in the caches instead of a 4-byte gather per list
        // entry into the 4 MB `offsets` array (one line of HBM traffic each).
        float4* s_rec = s_sh;
        const float4 q3 = make_float4(__uint_as_float(rect_bits), __uint_as_float((uint32_t)(rect.z - rect.x)), __uint_as_float(excl), 0.f);
        // a splat without a record (culled, or no tile) stores the "never visible" defaults: nothing ever gathers its record
        s_rec[threadIdx.x] = ell0; s_rec[kRecRow + threadIdx.x] = ell1; s_rec[2 * kRecRow + threadIdx.x] = rec2; s_rec[3 * kRecRow + threadIdx.x] = q3;
        __syncthreads();
        const int n_here = min(kBlock, s.N - (int)blockIdx.x * kBlock);
        float4* out = g.rec + 4 * (size_t)blockIdx.x * kBlock;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int G = j * kBlock + (int)threadIdx.x;       // float4 index inside the workgroup's block: record G / 4, quarter G % 4
            if (G < 4 * n_here) out[G] = s_rec[(G & 3) * kRecRow + (G >> 2)];
        }
    }
    if constexpr (COUNT_ATOMIC) {
        // fallback for very large images: per-tile counts with global atomics
        // (block-cooperative expansion keeps large splats from serialising a lane)
        s_off[threadIdx.x] = excl;
        s_rect[threadIdx.x] = rect_pack(rect, mask16);
        tile_test_prepare(ell0, ell1, s_r0[threadIdx.x], s_r1[threadIdx.x]);
        if (threadIdx.x == 0) s_off[kBlock] = total;
        __syncthreads();
        for_each_block_instance(s_off, s_rect, v.gx, [&](int e, uint32_t k, uint32_t tile, uint32_t, uint32_t tile_x, uint32_t tile_y, uint32_t w) {
            // the same decision as k_emit<false>: the mask where there is one
            if ((w & kRectMasked) ? (w >> k) & 1u : tile_reached(s_r0[e], s_r1[e], tile_x, tile_y)) atomicAdd(&g.tile_count[tile], 1u);
        });
    }
}

void launch_preprocess(const ViewK& v, const SplatsK& s, const Geom& g, int* radii, hipStream_t st) {
    const int nb = (s.N + kBlock - 1) / kBlock;
    const bool atomic = !use_count_matrix(v);
    if (atomic) hipMemsetAsync(g.tile_count, 0, sizeof(uint32_t) * (size_t)v.gx * v.gy, st);
    if (nb <= 0) return;
    // below degree 2 only <= 48 of the 192 bytes are needed; the two-tensor SH input always goes through the staging
    const bool stage = s.shs && v.sh_coeffs == 16 && (v.sh_degree >= 2 || s.shs_rest);
    if (stage && atomic) hipLaunchKernelGGL((k_preprocess<true, true>), dim3(nb), dim3(kBlock), 0, st, v, s, g, radii);
    else if (stage) hipLaunchKernelGGL((k_preprocess<true, false>), dim3(nb), dim3(kBlock), 0, st, v, s, g, radii);
    else if (atomic) hipLaunchKernelGGL((k_preprocess<false, true>), dim3(nb), dim3(kBlock), 0, st, v, s, g, radii);
    else hipLaunchKernelGGL((k_preprocess<false, false>), dim3(nb), dim3(kBlock), 0, st, v, s, g, radii);
}

// ------------------------------------------------------------------------------------------
// markVisible
// ------------------------------------------------------------------------------------------
__global__ void k_mark_visible(int N, const float* __restrict__ means3D, const float* __restrict__ vm,
                               unsigned char* __restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N) return;
    const float z = vm[2] * means3D[3 * idx] + vm[6] * means3D[3 * idx + 1] + vm[10] * means3D[3 * idx + 2] + vm[14];
    present[idx] = z > kNearCullZ ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// Densification statistics of one rendered view (the consumers of the rasterizer's means2D gradient and radii)
// ------------------------------------------------------------------------------------------
__global__ void k_densification_stats(int N, const float* __restrict__ dL_dmeans2D, const int* __restrict__ radii,
                                      float* __restrict__ grad_accum, float* __restrict__ denom, float* __restrict__ max_radii2D) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N) return;
    const int r = radii[idx];
    if (r <= 0) return;  // visibility_filter = radii > 0
    const float gx = dL_dmeans2D[3 * (size_t)idx], gy = dL_dmeans2D[3 * (size_t)idx + 1];
    if (grad_accum) grad_accum[idx] += sqrtf(gx * gx + gy * gy);
    if (denom) denom[idx] += 1.0f;
    if (max_radii2D) max_radii2D[idx] = fmaxf(max_radii2D[idx], (float)r);
}

void launch_densification_stats(int N, const float* dL_dmeans2D, const int* radii, float* grad_accum, float* denom, float* max_radii2D,
                                hipStream_t st) {
    if (N > 0) hipLaunchKernelGGL(k_densification_stats, dim3((N + 255) / 256), dim3(256), 0, st, N, dL_dmeans2D, radii, grad_accum, denom, max_radii2D);
}

void launch_mark_visible(int N, const float* means3D, const float* viewmatrix, unsigned char* present, hipStream_t st) {
    if (N <= 0) return;
    hipLaunchKernelGGL(k_mark_visible, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, means3D, viewmatrix, present);
}

// ------------------------------------------------------------------------------------------
// Backward preprocess
// ------------------------------------------------------------------------------------------
// SH_TO_COLORS: the splats carry SH, but instead of dL/dsh [N,K,3] the kernel writes the (clamp-masked) colour gradient
// dL/dcolour [N,3] -- the view-parallel step all-gathers those 12 bytes and rebuilds the SH gradient of all views
// locally (sh.hip); the gradient through the view direction still goes into dL/dmeans3D here.
template <bool STAGE_SH, bool SH_TO_COLORS, bool SPEC>
__global__ void __launch_bounds__(kBlock) k_preprocess_backward(const ViewK v, const SplatsK s, const Geom g,
                                                                const int* __restrict__ radii,
                                                                const float* __restrict__ slots,
                                                                const uint8_t* __restrict__ reached, const GradsK gr,
                                                                const int first_splat, const int end_splat) {
    // splats [first_splat, end_splat) of the cloud (first_splat a multiple of 256): the whole cloud in one launch, or one
    // slice of it per launch when the caller overlaps an exchange of the finished slices with the rest (sr_backward_splats)
    // (The three [N,3] gradient tensors leave as three 4-byte stores at a 12-byte stride per tensor and thread.  Sending them
    // through LDS as whole 16-byte pieces -- the transpose that gives k_preprocess 3 us -- was measured in round 5: SLOWER, 0.1038 vs
    // 0.0955 ms: it has to wait for the SH gradient to leave the staging area, which moves the small stores from the middle of
    // the workgroup's life to its very end.)
    __shared__ __attribute__((aligned(16))) float4 s_sh[STAGE_SH ? kBlock * kShRowF4 : 1];
    const int idx = first_splat + blockIdx.x * kBlock + threadIdx.x;
    const bool valid = idx < end_splat;
    int radius_in = 0;
    uint32_t first_in = 0, cnt_in = 0;
    uint8_t flags_in = 0;
    float4 q_in = make_float4(1.f, 0.f, 0.f, 0.f);
    float opac_in = 0.f;
    float3 p_in = make_float3(0.f, 0.f, 0.f), sc_in = make_float3(0.f, 0.f, 0.f);
    if (valid) {
        radius_in = radii[idx];
        first_in = g.offsets[idx]; cnt_in = g.touched[idx];
        flags_in = g.flags[idx];
        opac_in = s.opacities[idx];
        p_in = make_float3(s.means3D[3 * idx], s.means3D[3 * idx + 1], s.means3D[3 * idx + 2]);
        if (!s.cov3D) {
            q_in = reinterpret_cast<const float4*>(s.rotations)[idx];
            sc_in = make_float3(s.scales[3 * idx], s.scales[3 * idx + 1], s.scales[3 * idx + 2]);
        }
    }
    // the colour / direction Jacobian of the forward (SH path): requested with the other per-splat inputs, not where it is used --
    // behind the slot reduction and the covariance chain it was one more memory round trip at the end of the kernel
    float4 jac0 = make_float4(0.f, 0.f, 0.f, 0.f), jac1 = jac0;
    float jac22 = 0.f;
    if (valid && (gr.shs || SH_TO_COLORS) && v.sh_degree > 0) {
        const float4* jq = reinterpret_cast<const float4*>(g.dcol_ddir);
        jac0 = jq[idx]; jac1 = jq[(size_t)s.N + idx];
        jac22 = g.dcol_ddir[(size_t)8 * s.N + idx];
    }
    float q_norm = 1.0f;
    if (s.raw) activate_inputs(s.raw, sc_in, opac_in, q_in, q_norm);
    // Which of the splat's instances hold a gradient slot: the backward blend wrote one (and set the instance's `reached` byte)
    // for every list entry in front of the stop of its tile's last pixel; the others are neither written nor read.  The bytes
    // of the first 4 instances are requested now, together with the splat's own loads (most splats have <= 4 instances).
    const bool vis_in = valid && radius_in > 0;
    // The gradient slots of a splat's first four instances are requested TOGETHER with their `reached` bytes, before anything looks
    // at the bytes (one memory round trip instead of two: offsets -> {reached, slots} instead of offsets -> reached -> slots); a
    // slot the backward blend did not write holds stale bytes and is dropped by its byte.  Round 5, same box, alternated: 0.0941 ->
    // 0.0857 ms -- the kernel is a chain of dependent memory round trips per workgroup (timing builds without the slot reads,
    // without the SH stage-out and without the small stores: 0.067 / 0.068 / 0.088 ms; the parts ADD), and every round trip taken
    // out of the chain is time.
    uint32_t reached4 = 0u;
    // Only for small footprints -- a template parameter, chosen by the host with the rule that picks the backward blend kernel (at
    // most SR_BWD_SLOT_SPEC_BELOW instances per splat on average): with 15 instances per splat (300 k x 0.02) the speculative
    // form is SLOWER, 0.0787 vs 0.0680 ms (most of a splat's slots are then fetched by the loop below anyway), and a run-time
    // switch inside one kernel costs the dense case 7 us of its own (0.0749: the speculative registers stay allocated).
    constexpr bool use_spec = SPEC;
    float4 spec[4][3];
    {
        // branch-free inside the guard: positions beyond the splat's last instance re-request the last one (its line is there)
        uint32_t rb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) spec[i][0] = spec[i][1] = spec[i][2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vis_in && cnt_in > 0u) {
            const float4* sl0 = reinterpret_cast<const float4*>(slots) + (size_t)first_in * kSlotF4;
            const uint32_t lasti = min(cnt_in, 4u) - 1u;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) rb[i] = reached[first_in + min(i, lasti)];
            if (use_spec && cnt_in < 32u) {
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) {
                    const uint32_t j = min(i, lasti);
                    spec[i][0] = sl0[kSlotF4 * j]; spec[i][1] = sl0[kSlotF4 * j + 1]; spec[i][2] = sl0[kSlotF4 * j + 2];
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) if (i <= lasti) reached4 |= rb[i] << (8 * i);
        }
    }
    // ---- segmented reduction of every splat's instance slots (fixed order -> deterministic) ----
    // Splats with many instances (large footprints; dense real scenes) are reduced by the whole wavefront, 64
    // instances per step + one DPP reduction, instead of serialising hundreds of iterations in one lane.
    float sum[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) sum[k] = 0.f;
    {
        const float4* sl_all = reinterpret_cast<const float4*>(slots);
        const bool big = vis_in && cnt_in >= 32u;
        const int lane = lane_id();
        uint64_t todo = __ballot(big);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= (todo - 1);
            const uint32_t f = (uint32_t)__shfl((int)first_in, src, 64), c = (uint32_t)__shfl((int)cnt_in, src, 64);
            float part[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) part[k] = 0.f;
            for (uint32_t i = lane; i < c; i += 64u) {
                if (!reached[f + i]) continue;  // never written by the backward blend: contributes nothing
                const float4* sl = sl_all + (size_t)(f + i) * kSlotF4;
                const float4 a = sl[0], b4 = sl[1], c4 = sl[2];
                part[0] += a.x; part[1] += a.y; part[2] += a.z; part[3] += a.w;
                part[4] += b4.x; part[5] += b4.y; part[6] += b4.z; part[7] += b4.w;
                part[8] += c4.x; part[9] += c4.y;
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const float tot = __shfl(wave_sum_to_lane63(part[k]), 63, 64);
                if (lane == src) sum[k] = tot;
            }
        }
        if (vis_in && !big) {
            const float4* sl = sl_all + (size_t)first_in * kSlotF4;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {   // same order of additions as the loop below: bit-identical sums
                if (use_spec && i < cnt_in && ((reached4 >> (8 * i)) & 0xffu) != 0u) {
                    const float4 a = spec[i][0], b4 = spec[i][1], c4 = spec[i][2];
                    sum[0] += a.x; sum[1] += a.y; sum[2] += a.z; sum[3] += a.w;
                    sum[4] += b4.x; sum[5] += b4.y; sum[6] += b4.z; sum[7] += b4.w;
                    sum[8] += c4.x; sum[9] += c4.y;
                }
            }
            for (uint32_t i = use_spec ? 4u : 0u; i < cnt_in; ++i) {
                const bool hit = i < 4u ? ((reached4 >> (8 * i)) & 0xffu) != 0u : reached[first_in + i] != 0;
                if (!hit) continue;
                const float4 a = sl[kSlotF4 * i], b4 = sl[kSlotF4 * i + 1], c4 = sl[kSlotF4 * i + 2];
                sum[0] += a.x; sum[1] += a.y; sum[2] += a.z; sum[3] += a.w;
                sum[4] += b4.x; sum[5] += b4.y; sum[6] += b4.z; sum[7] += b4.w;
                sum[8] += c4.x; sum[9] += c4.y;
            }
        }
    }
    float3 d_mean = make_float3(0.f, 0.f, 0.f);
    float3 d_scale = make_float3(0.f, 0.f, 0.f);
    float2 d_m2d = make_float2(0.f, 0.f);
    const bool visible = valid && radius_in > 0;
    const int K = v.sh_coeffs;
    const float3 p = visible ? p_in : make_float3(0.f, 0.f, 0.f);
    const uint8_t flags = visible ? flags_in : (uint8_t)0;
    float3 d_rgb = visible ? make_float3(sum[6], sum[7], sum[8]) : make_float3(0.f, 0.f, 0.f);
    float3 dm_dir = make_float3(0.f, 0.f, 0.f);   // dL/dmean through the view direction of the SH colour (added behind the other terms)
    // ---- colour: SH coefficients and view direction, or precomputed colours ----
    auto colour_part = [&]() {
    if (gr.shs || SH_TO_COLORS) {
        // staged: this thread's LDS row first supplies its SH coefficients, then receives its gradients
        float *out_lo, *out_hi;   // float j = 3k + c of this splat's SH gradient: out_lo[j] for j < 3, out_hi[j] beyond
        if (STAGE_SH) sh_row_pointers(s_sh, gr.shs_rest != nullptr, threadIdx.x, out_lo, out_hi);
        else out_lo = out_hi = gr.shs + (size_t)idx * K * 3;
        int nb = 0;
        if (visible) {
            nb = (v.sh_degree + 1) * (v.sh_degree + 1);
            float3 dc = d_rgb;
            if (flags & kFlagClampR) dc.x = 0.f;
            if (flags & kFlagClampG) dc.y = 0.f;
            if (flags & kFlagClampB) dc.z = 0.f;
            const float* cp = v.campos;
            const float3 dv = make_float3(p.x - cp[0], p.y - cp[1], p.z - cp[2]);
            const float inv_len = 1.0f / sqrtf(dot3(dv, dv));
            const float3 d = make_float3(dv.x * inv_len, dv.y * inv_len, dv.z * inv_len);
            float B[16];
            sh_basis(v.sh_degree, d, B);
            if (!SH_TO_COLORS && !(STAGE_SH && !gr.shs_rest))
                for (int k = 0; k < nb; ++k) {
                    float* o = k == 0 ? out_lo : out_hi + 3 * k;
                    o[0] = B[k] * dc.x; o[1] = B[k] * dc.y; o[2] = B[k] * dc.z;
                }
            if (!SH_TO_COLORS && STAGE_SH && !gr.shs_rest) {
                // the thread's padded LDS row (13 float4) written as twelve 16-byte pieces: conflict-free, where the 48 float
                // stores of rounds 1-5 (stride 52 floats = 20 banks) put four lanes on every bank
                float g48[48];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float bk = k < nb ? B[k] : 0.f;
                    g48[3 * k] = bk * dc.x; g48[3 * k + 1] = bk * dc.y; g48[3 * k + 2] = bk * dc.z;
                }
                float4* row = s_sh + kShRowF4 * (int)threadIdx.x;
#pragma unroll
                for (int it = 0; it < 12; ++it) row[it] = make_float4(g48[4 * it], g48[4 * it + 1], g48[4 * it + 2], g48[4 * it + 3]);
            }
            if (SH_TO_COLORS) d_rgb = dc;  // what leaves is the masked colour gradient
            if (v.sh_degree > 0) {
                // dL/d(unit direction) = sum_c dL/dcolour_c * d colour_c / d direction (Jacobian stored by the forward)
                const float4 j0 = jac0, j1 = jac1;
                const float j22 = jac22;
                const float3 dd_ = make_float3(dc.x * j0.x + dc.y * j0.w + dc.z * j1.z,
                                               dc.x * j0.y + dc.y * j1.x + dc.z * j1.w,
                                               dc.x * j0.z + dc.y * j1.y + dc.z * j22);
                // through the normalisation d = dv / |dv|
                const float proj = dot3(d, dd_);
                dm_dir = make_float3((dd_.x - d.x * proj) * inv_len, (dd_.y - d.y * proj) * inv_len, (dd_.z - d.z * proj) * inv_len);
            }
        }
        if (!SH_TO_COLORS) {
            if (STAGE_SH && !gr.shs_rest) {
                if (!visible) {   // (a visible splat's row was written whole above)
                    float4* row = s_sh + kShRowF4 * (int)threadIdx.x;
#pragma unroll
                    for (int it = 0; it < 12; ++it) row[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else {
                for (int k = nb; k < K; ++k) { float* o = k == 0 ? out_lo : out_hi + 3 * k; o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; }
            }
        }
    }
    };
    auto stage_out_part = [&]() {
        __syncthreads();
        const size_t first = (size_t)first_splat + (size_t)blockIdx.x * kBlock;
        {
        if (gr.shs_rest) stage_sh_out_split(s_sh, gr.shs, gr.shs_rest, first, min(kBlock, end_splat - (int)first));
        else stage_sh_out(s_sh, gr.shs, first, min(kBlock, end_splat - (int)first));
        }
    };
    // (Computing, staging and storing the SH gradient FIRST, with the covariance chain rule running while those stores drain, was
    // measured in round 5: 0.0837 vs 0.0825 ms, nothing -- a wavefront does not wait for its stores anyway.)
    if (valid) {
    const float* vm = v.viewmatrix;
    const float* pm = v.projmatrix;

    float4 d_rot = make_float4(0.f, 0.f, 0.f, 0.f);
    float d_opac = 0.f;
    float d_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (visible) {
        const float S0 = sum[0], Sx = sum[1], Sy = sum[2], Sxx = sum[3], Sxy = sum[4], Syy = sum[5];
        const float dd = sum[9];
        const float o = opac_in;
        // the six geometric sums arrive multiplied by the opacity (k_render_backward accumulates o G dL/dalpha)
        d_opac = o > 0.0f ? S0 / o : 0.0f;
        // conic gradients; .y is half of the true dL/dB (off-diagonal counted once, used twice below)
        const float dcon_x = -0.5f * Sxx, dcon_y = -0.5f * Sxy, dcon_z = -0.5f * Syy;

        const float3 pv = make_float3(vm[0] * p.x + vm[4] * p.y + vm[8] * p.z + vm[12],
                                      vm[1] * p.x + vm[5] * p.y + vm[9] * p.z + vm[13],
                                      vm[2] * p.x + vm[6] * p.y + vm[10] * p.z + vm[14]);
        Sym3 S;
        float R[9];
        float3 sc = make_float3(0.f, 0.f, 0.f);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s.cov3D) {
            const float* c = s.cov3D + 6 * (size_t)idx;
            S.xx = c[0]; S.xy = c[1]; S.xz = c[2]; S.yy = c[3]; S.yz = c[4]; S.zz = c[5];
        } else {
            q = q_in;
            quat_to_rot(q, R);
            const float m = v.scale_modifier;
            sc = make_float3(m * sc_in.x, m * sc_in.y, m * sc_in.z);
            S = cov3d_from(sc, R);
        }
        const Ewa e = ewa_setup(pv, v, vm);
        const float3 Sm0 = sym_mul(S, e.m0), Sm1 = sym_mul(S, e.m1);
        const float a = dot3(e.m0, Sm0) + kDilation;
        const float b = dot3(e.m0, Sm1);
        const float c = dot3(e.m1, Sm1) + kDilation;
        const float denom = a * c - b * b;
        {   // the conic is recomputed from the inputs (as in the forward) instead of re-reading the 64-byte record
            const float det_inv = 1.0f / denom;
            const float A = c * det_inv, B = -b * det_inv, C = a * det_inv;
            // dL/d(NDC mean): (0.5 W, 0.5 H) scaled, the convention scene/gaussian_model.py:427-438 consumes
            d_m2d.x = -(A * Sx + B * Sy) * (0.5f * v.W);
            d_m2d.y = -(B * Sx + C * Sy) * (0.5f * v.H);
        }
        const float denom2inv = 1.0f / (denom * denom + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * dcon_x + 2.f * b * c * dcon_y + (denom - a * c) * dcon_z);
            dL_dc = denom2inv * (-a * a * dcon_z + 2.f * a * b * dcon_y + (denom - a * c) * dcon_x);
            dL_db = denom2inv * 2.f * (b * c * dcon_x - (denom + 2.f * b * b) * dcon_y + a * b * dcon_z);
        }
        // gradient w.r.t. the 6 unique entries of Sigma (off-diagonals appear twice in the quadratic form)
        const float3 m0 = e.m0, m1 = e.m1;
        d_cov[0] = m0.x * m0.x * dL_da + m0.x * m1.x * dL_db + m1.x * m1.x * dL_dc;
        d_cov[3] = m0.y * m0.y * dL_da + m0.y * m1.y * dL_db + m1.y * m1.y * dL_dc;
        d_cov[5] = m0.z * m0.z * dL_da + m0.z * m1.z * dL_db + m1.z * m1.z * dL_dc;
        d_cov[1] = 2.f * m0.x * m0.y * dL_da + (m0.x * m1.y + m0.y * m1.x) * dL_db + 2.f * m1.x * m1.y * dL_dc;
        d_cov[2] = 2.f * m0.x * m0.z * dL_da + (m0.x * m1.z + m0.z * m1.x) * dL_db + 2.f * m1.x * m1.z * dL_dc;
        d_cov[4] = 2.f * m0.y * m0.z * dL_da + (m0.y * m1.z + m0.z * m1.y) * dL_db + 2.f * m1.y * m1.z * dL_dc;

        // gradient w.r.t. the rows of M, then J, then t
        const float3 dm0 = make_float3(2.f * dL_da * Sm0.x + dL_db * Sm1.x, 2.f * dL_da * Sm0.y + dL_db * Sm1.y, 2.f * dL_da * Sm0.z + dL_db * Sm1.z);
        const float3 dm1 = make_float3(2.f * dL_dc * Sm1.x + dL_db * Sm0.x, 2.f * dL_dc * Sm1.y + dL_db * Sm0.y, 2.f * dL_dc * Sm1.z + dL_db * Sm0.z);
        const float3 rv0 = make_float3(vm[0], vm[4], vm[8]);
        const float3 rv1 = make_float3(vm[1], vm[5], vm[9]);
        const float3 rv2 = make_float3(vm[2], vm[6], vm[10]);
        const float dJ00 = dot3(dm0, rv0), dJ02 = dot3(dm0, rv2), dJ11 = dot3(dm1, rv1), dJ12 = dot3(dm1, rv2);
        const float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = (e.clamp_x ? 0.f : 1.f) * -v.focal_x * tz2 * dJ02;
        const float dty = (e.clamp_y ? 0.f : 1.f) * -v.focal_y * tz2 * dJ12;
        const float dtz = -v.focal_x * tz2 * dJ00 - v.focal_y * tz2 * dJ11 +
                          (2.f * v.focal_x * e.tx) * tz3 * dJ02 + (2.f * v.focal_y * e.ty) * tz3 * dJ12;
        // view rotation transposed; plus the depth output: depth = (row 2 of R_view) . p + const
        const float dz_total = dtz + dd;
        d_mean.x = vm[0] * dtx + vm[1] * dty + vm[2] * dz_total;
        d_mean.y = vm[4] * dtx + vm[5] * dty + vm[6] * dz_total;
        d_mean.z = vm[8] * dtx + vm[9] * dty + vm[10] * dz_total;

        // screen-space mean -> 3D mean through the perspective divide (with the +1e-7)
        const float hx_ = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
        const float hy_ = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
        const float hw_ = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
        const float m_w = 1.0f / (hw_ + kWEps);
        const float mul1 = hx_ * m_w * m_w, mul2 = hy_ * m_w * m_w;
        d_mean.x += (pm[0] * m_w - pm[3] * mul1) * d_m2d.x + (pm[1] * m_w - pm[3] * mul2) * d_m2d.y;
        d_mean.y += (pm[4] * m_w - pm[7] * mul1) * d_m2d.x + (pm[5] * m_w - pm[7] * mul2) * d_m2d.y;
        d_mean.z += (pm[8] * m_w - pm[11] * mul1) * d_m2d.x + (pm[9] * m_w - pm[11] * mul2) * d_m2d.y;

        // Sigma -> scales, quaternion
        if (!s.cov3D) {
            // G = dL/dSigma as a full symmetric matrix (off-diagonals halved)
            const float Gxx = d_cov[0], Gxy = 0.5f * d_cov[1], Gxz = 0.5f * d_cov[2], Gyy = d_cov[3], Gyz = 0.5f * d_cov[4], Gzz = d_cov[5];
            // Sigma = L L^T, L = R diag(s):  dL/dL = 2 G L ;  L[i][k] = R[i][k] s_k
            const float sv[3] = {sc.x, sc.y, sc.z};
            float dLm[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float l0 = R[k] * sv[k], l1 = R[3 + k] * sv[k], l2 = R[6 + k] * sv[k];
                dLm[k] = 2.f * (Gxx * l0 + Gxy * l1 + Gxz * l2);
                dLm[3 + k] = 2.f * (Gxy * l0 + Gyy * l1 + Gyz * l2);
                dLm[6 + k] = 2.f * (Gxz * l0 + Gyz * l1 + Gzz * l2);
            }
            const float m = v.scale_modifier;
            d_scale.x = m * (R[0] * dLm[0] + R[3] * dLm[3] + R[6] * dLm[6]);
            d_scale.y = m * (R[1] * dLm[1] + R[4] * dLm[4] + R[7] * dLm[7]);
            d_scale.z = m * (R[2] * dLm[2] + R[5] * dLm[5] + R[8] * dLm[8]);
            float D[9];  // dL/dR[i][k] = dL/dL[i][k] * s_k
#pragma unroll
            for (int i = 0; i < 3; ++i) { D[3 * i] = dLm[3 * i] * sv[0]; D[3 * i + 1] = dLm[3 * i + 1] * sv[1]; D[3 * i + 2] = dLm[3 * i + 2] * sv[2]; }
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            d_rot.x = 2.f * (-z * D[1] + y * D[2] + z * D[3] - x * D[5] - y * D[6] + x * D[7]);
            d_rot.y = 2.f * (y * D[1] + z * D[2] + y * D[3] - 2.f * x * D[4] - r * D[5] + z * D[6] + r * D[7] - 2.f * x * D[8]);
            d_rot.z = 2.f * (-2.f * y * D[0] + x * D[1] + r * D[2] + x * D[3] + z * D[5] - r * D[6] + z * D[7] - 2.f * y * D[8]);
            d_rot.w = 2.f * (-2.f * z * D[0] - r * D[1] + x * D[2] + r * D[3] - 2.f * z * D[4] + y * D[5] + x * D[6] + y * D[7]);
        }
    }

    colour_part();
    d_mean.x += dm_dir.x; d_mean.y += dm_dir.y; d_mean.z += dm_dir.z;
    if (s.raw) {  // derivatives of the activations: gradients w.r.t. the raw parameters
        if (s.raw & SR_RAW_SCALES) { d_scale.x *= sc_in.x; d_scale.y *= sc_in.y; d_scale.z *= sc_in.z; }   // d exp = exp
        if (s.raw & SR_RAW_OPACITY) d_opac *= opac_in * (1.0f - opac_in);                                   // d sigmoid
        if (s.raw & SR_RAW_ROTATIONS) {                                                                     // d (q / |q|)
            const float dot = q_in.x * d_rot.x + q_in.y * d_rot.y + q_in.z * d_rot.z + q_in.w * d_rot.w;
            const float inv = 1.0f / q_norm;
            d_rot = make_float4((d_rot.x - q_in.x * dot) * inv, (d_rot.y - q_in.y * dot) * inv, (d_rot.z - q_in.z * dot) * inv,
                                (d_rot.w - q_in.w * dot) * inv);
        }
    }
    if (gr.colors) { gr.colors[3 * idx] = d_rgb.x; gr.colors[3 * idx + 1] = d_rgb.y; gr.colors[3 * idx + 2] = d_rgb.z; }

    {
    gr.means3D[3 * idx] = d_mean.x; gr.means3D[3 * idx + 1] = d_mean.y; gr.means3D[3 * idx + 2] = d_mean.z;
    gr.means2D[3 * idx] = d_m2d.x; gr.means2D[3 * idx + 1] = d_m2d.y; gr.means2D[3 * idx + 2] = 0.f;
    if (gr.scales) { gr.scales[3 * idx] = d_scale.x; gr.scales[3 * idx + 1] = d_scale.y; gr.scales[3 * idx + 2] = d_scale.z; }
    gr.opacity[idx] = d_opac;
    if (gr.rotations) reinterpret_cast<float4*>(gr.rotations)[idx] = d_rot;
    if (gr.cov3D) { for (int k = 0; k < 6; ++k) gr.cov3D[6 * (size_t)idx + k] = d_cov[k]; }
    }
    }  // valid
    if constexpr (STAGE_SH && !SH_TO_COLORS) stage_out_part();
}

void launch_preprocess_backward(const ViewK& v, const SplatsK& s, const Geom& g, const int* radii,
                                const float* slots, const uint8_t* reached, const GradsK& gr, int first, int count, bool small_footprint,
                                hipStream_t st) {
    const int end = first + count < s.N ? first + count : s.N;
    const int nb = (end - first + kBlock - 1) / kBlock;
    if (nb <= 0) return;
    const bool to_colors = s.shs && !gr.shs && gr.colors;
    const bool stage = s.shs && v.sh_coeffs == 16 && gr.shs;  // LDS rows only carry the SH gradient out (coalesced 16-byte stores)
    const bool spec = small_footprint && 1;
#define SR_LAUNCH_PB(A, B, C) hipLaunchKernelGGL((k_preprocess_backward<A, B, C>), dim3(nb), dim3(kBlock), 0, st, v, s, g, radii, slots, reached, gr, first, end)
    if (to_colors && stage) { if (spec) SR_LAUNCH_PB(true, true, true); else SR_LAUNCH_PB(true, true, false); }
    else if (to_colors) { if (spec) SR_LAUNCH_PB(false, true, true); else SR_LAUNCH_PB(false, true, false); }
    else if (stage) { if (spec) SR_LAUNCH_PB(true, false, true); else SR_LAUNCH_PB(true, false, false); }
    else { if (spec) SR_LAUNCH_PB(false, false, true); else SR_LAUNCH_PB(false, false, false); }
#undef SR_LAUNCH_PB
}

}  // namespace sr
