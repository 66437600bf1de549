// Host-side launchers of the gfx950 kernels (one translation unit per pipeline stage).
#pragma once
#include "common.h"

namespace sr {

struct SplatsK {
    int N;
    const float* means3D; const float* opacities; const float* scales; const float* rotations;
    const float* cov3D; const float* shs; const float* colors;
    int raw;  // SR_RAW_* bits: activations applied on load, their derivatives on the way out
    const float* shs_rest;  // non-NULL: shs = dc [N,1,3], shs_rest = [N,15,3]
};

struct GradsK {
    float* means3D; float* means2D; float* opacity; float* scales; float* rotations; float* cov3D;
    float* shs; float* colors; float* shs_rest;
};

// preprocess.hip
void launch_preprocess(const ViewK& v, const SplatsK& s, const Geom& g, int* radii, hipStream_t st);
void launch_mark_visible(int N, const float* means3D, const float* viewmatrix, unsigned char* present, hipStream_t st);
void launch_densification_stats(int N, const float* dL_dmeans2D, const int* radii, float* grad_accum, float* denom, float* max_radii2D,
                                hipStream_t st);
void launch_preprocess_backward(const ViewK& v, const SplatsK& s, const Geom& g, const int* radii,
                                const float* slots, const uint8_t* reached, const GradsK& gr, int first, int count, bool small_footprint,
                                hipStream_t st);
// binning.hip
// true when the image is small enough for the atomic-free count-matrix bucketing
inline bool use_count_matrix(const ViewK& v) { return v.gx * v.gy <= kMaxMatrixTiles; }
void launch_count_tiles(const ViewK& v, int N, const Geom& g, hipStream_t st);
void launch_scan_small(const ViewK& v, int N, const Geom& g, uint32_t* host_out, uint32_t host_seq, hipStream_t st);
void launch_emit(const ViewK& v, int N, const Geom& g, const Binning& b, hipStream_t st);
void launch_sort_tiles(const ViewK& v, const Geom& g, const Binning& b, long long max_len, long long expected_len, hipStream_t st);
// render.hip
void launch_render_forward(const ViewK& v, const Geom& g, const Binning& b, const Image& im,
                           float* out_color, float* out_depth, float* out_alpha, hipStream_t st);
void launch_render_backward(const ViewK& v, const Geom& g, const Binning& b, const Image& im,
                            const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                            float* slots, hipStream_t st);
// blend_bwd.hip
void launch_render_backward_quads(const ViewK& v, const Geom& g, const Binning& b, const Image& im,
                                 const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                                 float* slots, hipStream_t st);

int backward_stats(unsigned long long* out8, int reset);

// densify.hip
struct DensifyArgs { float grad_threshold, min_opacity, extent, percent_dense, max_screen_size; };
size_t densify_workspace_bytes(int n);
void launch_densify_plan(int n, const float* log_scales, int scale_cols, const float* opacity_logit, const float* grad_accum,
                         const float* denom, const float* max_radii2D, const DensifyArgs& a, void* workspace, int32_t* dest,
                         uint32_t** totals_out, hipStream_t st);
void launch_densify_gather(int n, int row, const float* src, float* dst, const int32_t* dest, int mode, const float* log_scales,
                           int scale_cols, const float* rotations, const float* unit, hipStream_t st);

// mlp.hip
size_t mlp_weight_grad_workspace(int n_points, int n_jobs, const SrMlpGradJob* jobs);
int launch_mlp_weight_grad(int n_points, int n_jobs, const SrMlpGradJob* jobs, void* workspace, size_t workspace_bytes, hipStream_t st);
int launch_mlp_pack(int n_jobs, const SrMlpPackJob* jobs, hipStream_t st);
int launch_mlp_chain(int n_points, int hidden_tiles, int n_ops, const SrMlpOp* ops, float slope, hipStream_t st);

int launch_mlp_input_forward(int N, int L, int F, int TL, int row, const float* xyz, const float* feat, const float* time, float* x0, hipStream_t st);
int launch_mlp_top_gradient(int N, int out, int row, const float* y, const float* dy, float slope, float* G, hipStream_t st);
int launch_mlp_input_backward(int N, int L, int F, int row, const float* xyz, const float* g, float* d_xyz, float* d_feat, hipStream_t st);
int launch_resfield_compose(int n_jobs, const SrResFieldJob* jobs, const long long* frame, hipStream_t st);
size_t resfield_backward_workspace(int n_jobs, const SrResFieldJob* jobs);
int launch_resfield_backward(int n_jobs, const SrResFieldJob* jobs, const long long* frame, void* workspace, size_t workspace_bytes, hipStream_t st);

// triplane.hip
int launch_triplane_forward(int N, int C, int H, int W, const float* planes_chw, float* planes_hwc, const float* pts, float* out, hipStream_t st);
int launch_triplane_backward(int N, int C, int H, int W, const float* planes_hwc, const float* pts, const float* g, float* d_planes_chw,
                             float* d_pts, void* workspace, hipStream_t st);
size_t triplane_backward_workspace(int N, int C, int H, int W);

// knn.hip
size_t knn_workspace_bytes(int n);
void launch_knn3(int n, const float* pts, float* out, void* workspace, hipStream_t st);

// sh.hip
void launch_sh_forward(int N, int K, int deg, const float* means3D, const float* shs, const float* campos, float* colors,
                       unsigned char* clamped, hipStream_t st);
void launch_sh_forward_views(int N, int K, int deg, int V, const float* means3D, const float* shs, const float* campos, float* colors,
                             float* keep, hipStream_t st);
void launch_sh_backward(int N, int K, int deg, int V, const float* means3D, const float* shs, const float* campos, const float* dcol,
                        float scale, float* d_shs, float* d_means, int accumulate_means, hipStream_t st);

}  // namespace sr
