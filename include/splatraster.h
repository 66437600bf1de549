/*
 * splatraster.h -- C ABI of libsplatraster.so, the MI355X (gfx950) differentiable
 * Gaussian-splat rasterizer that replaces the `diff_gaussian_rasterization._C` extension
 * SplatFields imports at reference gaussian_renderer/__init__.py:14.
 *
 * Every entry point takes plain pointers and sizes; no torch / C++ types cross this boundary.
 * All device memory (inputs, outputs, gradients, the three opaque state buffers) is owned by
 * the caller (PyTorch-ROCm tensors, `tensor.data_ptr()`); the library owns nothing persistent (its only state: a pinned
 * read-back word + event per host thread and device, the mutex-guarded event list of the optional stage profiling, and
 * per-device "attribute set" flags of three kernels).
 * Every kernel is launched on the caller-supplied HIP stream.  Functions return 0 on success,
 * a non-zero status otherwise; `sr_last_error()` returns a thread-local message.
 *
 * Which upstream interface each entry replaces ([EXT] = the pip dependency
 * ingra14m/depth-diff-gaussian-rasterization@f2d8fa9, reference README.md:28, not vendored):
 *
 *   sr_forward_prepare + sr_forward_render
 *        <- [EXT] `_C.rasterize_gaussians(...)`, reached from `GaussianRasterizer.forward`
 *           called at reference gaussian_renderer/__init__.py:94-102 and :106-114.
 *   sr_backward
 *        <- [EXT] `_C.rasterize_gaussians_backward(...)`, reached from autograd when
 *           reference train.py:252 runs `loss.backward()`.
 *   sr_mark_visible
 *        <- [EXT] `_C.mark_visible(...)` (`GaussianRasterizer.markVisible`; never called by
 *           SplatFields, exported for completeness).
 *   sr_densification_stats
 *        <- reference train.py:280-286 and scene/gaussian_model.py:427-438 (`add_densification_stats`): the
 *           consumers of `radii` and `viewspace_points.grad`, fused into one kernel.
 *   SrView
 *        <- the 12-field `GaussianRasterizationSettings` built at reference
 *           gaussian_renderer/__init__.py:59-72 (and :76-89 for the alpha pass).
 *   SrSplats
 *        <- the keyword arguments of the call at reference gaussian_renderer/__init__.py:94-102.
 *   geom / binning / image buffers
 *        <- [EXT] geomBuffer / binningBuffer / imgBuffer saved by the autograd ctx.
 *
 * Conventions: viewmatrix / projmatrix are the *transposed* matrices of
 * reference scene/cameras.py:68-73 stored contiguously, i.e. element [i][j] at ptr[4*i+j] and
 * p_hom = [x y z 1] @ M; quaternions are (r,x,y,z) used as given; scales and opacities are
 * already activated; SH is [N,K,3] coefficient-major.
 */
#ifndef SPLATRASTER_H
#define SPLATRASTER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this ABI; sr_version() returns the library's.  A caller built against another version must not use the library
 * (the binding checks at load).  History: 3 = rounds 2-3; 4 = round 4's additions made official (sr_backward's `binning` is
 * written; sr_backward_blend / sr_backward_splats / sr_debug_snapshot; Geom's eight count words; images of at most 4095 tiles a
 * side; record quarter 3 / block_offsets carry the instance index) + round 5: sr_forward_async / sr_ticket_* / sr_debug_counters. */
#define SR_VERSION 4
#define SR_TILE 16 /* binning tile edge in pixels (upstream BLOCK_X = BLOCK_Y) */

typedef struct SrView {
    int image_height;         /* 1..65520: at most 4095 tiles a side (tile coordinates are stored in 12 bits) */
    int image_width;          /* 1..65520 */
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int sh_degree;            /* active SH degree, 0..3 */
    int sh_coeffs;            /* K = coefficients stored per splat (shs.shape[1]); 0 with precomputed colours */
    int prefiltered;
    int debug;                /* non-zero: synchronise + check after every launch */
    const float* viewmatrix;  /* device, 16 floats */
    const float* projmatrix;  /* device, 16 floats */
    const float* campos;      /* device, 3 floats */
    const float* bg;          /* device, 3 floats */
} SrView;

typedef struct SrSplats {
    int count;                   /* N */
    const float* means3D;        /* device [N,3] */
    const float* opacities;      /* device [N] (the [N,1] tensor, contiguous) */
    const float* scales;         /* device [N,3]; NULL iff cov3D_precomp is given */
    const float* rotations;      /* device [N,4]; NULL iff cov3D_precomp is given */
    const float* cov3D_precomp;  /* device [N,6] (xx,xy,xz,yy,yz,zz) or NULL */
    const float* shs;            /* device [N,K,3] or NULL */
    const float* colors_precomp; /* device [N,3] or NULL (exactly one of shs / colors_precomp) */
    int raw_params;              /* bit mask of SR_RAW_*: inputs that arrive as the optimiser's raw parameters; the activation of
                                  * reference scene/gaussian_model.py:64-86 is applied inside the preprocess kernels and its
                                  * derivative inside sr_backward (the gradients are then w.r.t. the raw parameters) */
    const float* shs_rest;       /* NULL, or device [N,15,3]: the SH coefficients arrive as the reference stores them, `shs` =
                                  * `_features_dc` [N,1,3] and `shs_rest` = `_features_rest` [N,15,3] (scene/gaussian_model.py:40-41),
                                  * instead of their per-iteration concatenation (:79-82); needs sh_coeffs == 16, both 16-byte
                                  * aligned.  The gradients then go to SrGrads.dL_dshs [N,1,3] and dL_dshs_rest [N,15,3]. */
} SrSplats;

#define SR_RAW_SCALES 1    /* scales = log-scales:        get_scaling  = exp(_scaling)          (gaussian_model.py:64-68) */
#define SR_RAW_OPACITY 2   /* opacities = logits:         get_opacity  = sigmoid(_opacity)      (gaussian_model.py:84-86) */
#define SR_RAW_ROTATIONS 4 /* rotations = unnormalised:   get_rotation = normalize(_rotation)   (gaussian_model.py:70-72) */
#define SR_FORWARD_ONLY 8  /* no sr_backward will follow this forward (rendering / evaluation): state only the backward reads is not
                            * written (the 36 B/splat colour-direction Jacobian of the SH path) */

/* Dense per-splat gradient outputs, all written in full by sr_backward (culled splats get zeros). */
typedef struct SrGrads {
    float* dL_dmeans3D;   /* [N,3] */
    float* dL_dmeans2D;   /* [N,3]; [:, :2] in the upstream NDC-scaled convention (x 0.5*W, 0.5*H), [:,2] = 0 */
    float* dL_dopacity;   /* [N]   */
    float* dL_dscales;    /* [N,3] or NULL with cov3D_precomp */
    float* dL_drotations; /* [N,4] or NULL with cov3D_precomp */
    float* dL_dcov3D;     /* [N,6] or NULL unless cov3D_precomp */
    float* dL_dshs;       /* [N,K,3] or NULL */
    float* dL_dcolors;    /* [N,3] or NULL.  With SH input, dL_dshs == NULL and dL_dcolors != NULL selects the colour-gradient
                           * mode: the clamp-masked dL/dcolour is written instead of dL/dsh (view-parallel exchange, sr_sh_backward);
                           * the gradient through the view direction is still added to dL_dmeans3D. */
    float* dL_dshs_rest;  /* [N,15,3] with SrSplats.shs_rest, else NULL */
} SrGrads;

int sr_version(void);
const char* sr_last_error(void);

/* Sizes (bytes) of the caller-allocated opaque buffers.  `instances` is the value
 * sr_forward_prepare returned for this view. */
size_t sr_geom_bytes(int n_splats, int height, int width);
size_t sr_binning_bytes(long long instances, int height, int width);
size_t sr_image_bytes(int height, int width);
size_t sr_backward_scratch_bytes(long long instances);

/* Stage 1: cull, 3D->2D covariance projection (EWA), conic, radius, tile rectangle, SH->RGB,
 * per-tile instance counts and their prefix sums.  Writes `radii` [N] (int32) and the geom buffer.
 * Blocks on the stream once to read back the number of tile-splat instances. */
int sr_forward_prepare(const SrView* view, const SrSplats* splats, void* geom, int* radii,
                       long long* instances_out, void* hip_stream);

/* Stage 2: bucket instances per tile, sort each tile's list front-to-back (depth, then splat
 * index), alpha-composite.  Writes out_color [3,H,W], out_depth [1,H,W], out_alpha [1,H,W]
 * (= 1 - final transmittance; may be NULL) plus the binning and image state buffers. */
int sr_forward_render(const SrView* view, const SrSplats* splats, void* geom, void* binning,
                      long long instances, void* image, float* out_color, float* out_depth,
                      float* out_alpha, void* hip_stream);

/* Both forward stages in one call WITHOUT draining the pipeline: stage 1 is launched, the instance count is
 * copied to pinned host memory asynchronously, stage 2 is launched right behind it for a binning buffer sized
 * for `binning_capacity` instances (every stage-2 kernel exits immediately if the count exceeds it), and only
 * then does the host wait -- for the early copy, while the GPU keeps running stage 2.  (So the host still blocks once per
 * forward, as upstream's num_rendered read-back does, but only until stage 1 has finished: the GPU never idles.)
 * Returns 0 when the count fits (outputs valid), SR_NEED_CAPACITY when it does not: the caller then allocates
 * sr_binning_bytes(*instances_out) and calls sr_forward_render with instances = *instances_out.
 * The binning buffer is carved for the capacity it was rendered with; pass that same number as `instances`
 * to sr_backward.
 * Threading: the asynchronous read-back uses one pinned word + event per (host thread, device); host threads may call
 * concurrently, each on its own stream and buffers. */
#define SR_NEED_CAPACITY 2
int sr_forward(const SrView* view, const SrSplats* splats, void* geom, int* radii, void* binning,
               long long binning_capacity, void* image, float* out_color, float* out_depth, float* out_alpha,
               long long* instances_out, void* hip_stream);

/* sr_forward WITHOUT any host wait (the capacity-bounded form of SURVEY.md section 8b: "... or is avoided with a capacity-bounded
 * workspace"; [EXT] reads num_rendered back in the middle of every forward).  The caller supplies what sr_forward learns by
 * waiting -- a binning capacity and the longest tile list to expect (`longest_list_hint`: which sort classes to launch: lists up
 * to 2048 / 4096 / 8192 entries, or all) -- typically what an earlier forward of the same view reported, with headroom.  Both
 * stages are launched back to back and the call returns; every stage-2 kernel exits on the device if the instance count
 * exceeds the capacity, and the blend exits if a list is longer than the launched sort classes cover, so a wrong guess never
 * reads or writes out of bounds -- it leaves the OUTPUTS UNDEFINED.  `*ticket_out` receives a ticket (a pinned status block +
 * an event recorded behind stage 1, pooled inside the library); the caller MUST redeem it with sr_ticket_wait before it uses
 * the outputs' gradients (before sr_backward at the latest): it waits for stage 1 of that forward only -- usually long
 * finished -- and returns the instance count and the longest list; the outputs are valid iff
 *     instances <= binning_capacity  and  longest_list <= max(2048, the class bound the hint selected).
 * When they are not, re-run the forward (sr_forward, or sr_forward_async with the reported figures).
 * `longest_list_expected` (<= the hint, or -1): the figure without its headroom.  A sort class that is launched only for the
 * headroom's sake (no list of its length is expected) gets a grid of a few workgroups instead of 512: it finds nothing and
 * leaves, or sorts the one list that did grow into it.
 * sr_ticket_release = sr_ticket_wait without results (a forward whose outputs were dropped).  A ticket is redeemed once. */
int sr_forward_async(const SrView* view, const SrSplats* splats, void* geom, int* radii, void* binning,
                     long long binning_capacity, long long longest_list_hint, long long longest_list_expected, void* image,
                     float* out_color, float* out_depth, float* out_alpha, void** ticket_out, void* hip_stream);
int sr_ticket_wait(void* ticket, long long* instances_out, long long* longest_list_out);
/* the longest tile list of the calling thread's most recent sr_forward (-1 before the first): the hint of a later
 * sr_forward_async of the same view */
long long sr_last_longest_list(void);
int sr_ticket_release(void* ticket);

/* Backward of both stages.  dL_ddepth / dL_dalpha may be NULL (treated as zero).  `instances` is the capacity the binning
 * buffer was rendered with; `scratch` holds sr_backward_scratch_bytes(instances) bytes.
 * `instances_rendered`: the instance count the forward reported for these buffers (*instances_out), or -1 if the caller did
 * not keep it.  It only selects a variant of the per-splat reduction (fewer than SR_BWD_SLOT_SPEC_BELOW instances per splat on
 * average, or unknown: a splat's first gradient slots are requested together with their `reached` bytes); the results are the
 * same.  (Rounds 2-5 also switched the backward BLEND kernel on it; since round 6 the entry-per-lane kernel -- quad buckets,
 * DPP scans: blend_bwd.hip -- runs at every footprint, and the pixel-per-lane kernel of render.hip is kept as a second
 * implementation of the same gradient slots: sr_set_backward_kernel.)
 * `binning` is NOT const: the backward blend sets, inside it, one `reached` byte per tile-splat instance it wrote a gradient
 * slot for (the forward's scatter cleared them), and the per-splat reduction reads them back.  Consequences for callers:
 * ONE backward at a time per set of forward buffers (two concurrent backwards on the same buffers -- different streams or
 * host threads -- race on those bytes), a backward may be REPEATED on the same buffers (it sets the same bytes again:
 * `retain_graph=True` works), and a copy of the binning buffer taken before a backward is a valid input of another one. */
#define SR_BWD_SLOT_SPEC_BELOW 6
int sr_backward(const SrView* view, const SrSplats* splats, const void* geom, void* binning,
                long long instances, long long instances_rendered, const void* image, const int* radii,
                const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, void* scratch,
                const SrGrads* grads, void* hip_stream);

/* sr_backward in two calls, for callers that overlap an exchange of finished gradients with the rest of the backward (the
 * view-parallel training step, splatfields_amd/view_parallel.py; no counterpart in [EXT], which is single-GPU):
 *   sr_backward_blend   -- the backward blend only: fills `scratch` (one gradient slot per reached tile-splat instance);
 *   sr_backward_splats  -- the per-splat part (slot reduction + chain rule to means3D / scales / rotations / SH / opacity)
 *                          for splats [first_splat, first_splat + n_splats) only; first_splat must be a multiple of 256;
 *                          the SrGrads pointers are the bases of the FULL gradient tensors (rows outside the range are not
 *                          touched).  Any partition of [0, count) into such ranges, in any order, after one
 *                          sr_backward_blend, gives exactly what sr_backward gives (same kernels, same arithmetic per splat).
 * Same buffers and the same `instances` / `instances_rendered` meaning as sr_backward (the per-splat kernel has a variant for
 * small footprints too: it requests a splat's first gradient slots together with their `reached` bytes). */
int sr_backward_blend(const SrView* view, const SrSplats* splats, const void* geom, void* binning,
                      long long instances, long long instances_rendered, const void* image,
                      const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, void* scratch, void* hip_stream);
int sr_backward_splats(const SrView* view, const SrSplats* splats, const void* geom, void* binning,
                       long long instances, long long instances_rendered, const void* image, const int* radii, void* scratch,
                       const SrGrads* grads, int first_splat, int n_splats, void* hip_stream);

/* Pins the backward blend kernel for A/B measurements and for the test that compares the two: 0 = chosen per launch by the
 * footprint (default), 1 = pixel-per-lane kernel, 2 = entry-per-lane kernel.  The initial value comes from the environment
 * variable SPLATRASTER_BWD ("wave" = 1, "quads" = 2), read once when the library is loaded.  Process-wide; returns the
 * previous setting, or -1 for an unknown value (nothing changes). */
int sr_set_backward_kernel(int which);

/* present[i] = 1 iff splat i passes the near-plane test (view z > 0.2). */
int sr_mark_visible(int n_splats, const float* means3D, const float* viewmatrix,
                    const float* projmatrix, unsigned char* present, void* hip_stream);

/* The per-view densification bookkeeping that consumes the rasterizer's outputs (reference train.py:280-286 and
 * scene/gaussian_model.py:427-438, `add_densification_stats`), for the splats with radii > 0 (`visibility_filter`), fused:
 *   grad_accum[i] += |dL_dmeans2D[i, :2]|;   denom[i] += 1;   max_radii2D[i] = max(max_radii2D[i], radii[i]).
 * The reference does this with boolean-mask indexing (several kernels and a host synchronisation per iteration).
 * Any of the three outputs may be NULL. */
int sr_densification_stats(int n_splats, const float* dL_dmeans2D, const int* radii, float* grad_accum, float* denom,
                           float* max_radii2D, void* hip_stream);

/* SH colour evaluation as a stand-alone stage (the SH part of the forward preprocess; reference utils/sh_utils.py:57-112,
 * extract_geo.py:40-44): colors[N,3] = max(sum_k basis_k(dir) shs[k] + 0.5, 0), clamped[N] bit c set where channel c was
 * clamped.  Feeding `colors` to sr_forward as colors_precomp gives the same image as passing `shs`. */
int sr_sh_forward(int n_splats, int sh_coeffs, int sh_degree, const float* means3D, const float* shs, const float* campos,
                  float* colors, unsigned char* clamped, void* hip_stream);

/* sr_sh_forward for n_views cameras in one pass over the coefficients (SH-sharded view-parallel step):
 * campos [n_views,3]; colors [n_views,N,3]; keep [n_views,N,3] (may be NULL) = 1 where the channel was not clamped, else 0
 * (the factor that channel's colour gradient gets before sr_sh_backward). */
int sr_sh_forward_views(int n_splats, int sh_coeffs, int sh_degree, int n_views, const float* means3D, const float* shs,
                        const float* campos, float* colors, float* keep, void* hip_stream);

/* Backward of sr_sh_forward for n_views cameras at once (view-parallel training).
 * campos [n_views,3]; dL_dcolors [n_views,N,3], zero where the colour was clamped in that view.
 * dL_dshs [N,K,3] (may be NULL) = scale * sum_v basis(dir_v) (x) dL_dcolors[v];
 * dL_dmeans3D [N,3] (may be NULL) = or += scale * the gradient through the view direction. */
int sr_sh_backward(int n_splats, int sh_coeffs, int sh_degree, int n_views, const float* means3D, const float* shs,
                   const float* campos, const float* dL_dcolors, float scale, float* dL_dshs, float* dL_dmeans3D,
                   int accumulate_means, void* hip_stream);

/* mean_dist2[i] = mean squared distance from point i to its 3 nearest other points (exact k-NN).
 * Replaces [EXT] simple_knn._C.distCUDA2 (reference README.md:29), called once at initialisation by reference
 * scene/gaussian_model.py:105.  `workspace` holds sr_knn_workspace_bytes(n) bytes of device memory. */
size_t sr_knn_workspace_bytes(int n_points);
int sr_knn3_mean_dist2(int n_points, const float* points, float* mean_dist2, void* workspace, void* hip_stream);

/* Densification / pruning of the splat set on the device: reference scene/gaussian_model.py:411-425 (`densify_and_prune`) with
 * :394-409 (`densify_and_clone`), :355-380 (`densify_and_split`, N = 2), :306-353 (`densification_postfix`) and :272-304
 * (`prune_points`), planned together.  Inputs are the optimiser's RAW parameters (log-scales [N,1|3], opacity logits [N]) and
 * the densification statistics (`xyz_gradient_accum` [N], `denom` [N], `max_radii2D` [N] or NULL = not tested).
 *
 * sr_densify_plan writes dest [4][N] (int32): the row, in the final tensors, of splat i itself (k = 0), of its clone (k = 1)
 * and of its two split children (k = 2, 3), or -1 -- in the order the reference's cat / mask sequence produces:
 * [surviving originals][clones][first children][second children] -- and returns counts5 = rows of each kind + their sum.
 * It synchronises the stream once (the caller needs the new size to allocate).  `workspace`: sr_densify_workspace_bytes(N).
 *
 * sr_densify_gather builds one final tensor from one source tensor of [N, row_floats]:
 *   mode 0  every surviving row is a copy (features, opacity, rotation);
 *   mode 1  Adam moment: the splat's own row is copied, new rows are zero (cat_tensors_to_optimizer);
 *   mode 2  positions [N,3]: children = xyz + build_rotation(rotation) (unit_normal * exp(log_scale)), unit_normals [2][N][3];
 *   mode 3  log-scales: children = log(exp(log_scale) / (0.8 * 2)). */
size_t sr_densify_workspace_bytes(int n_splats);
int sr_densify_plan(int n_splats, const float* log_scales, int scale_cols, const float* opacity_logits, const float* grad_accum,
                    const float* denom, const float* max_radii2D, float grad_threshold, float min_opacity, float extent,
                    float percent_dense, float max_screen_size, void* workspace, int* dest, long long* counts5, void* hip_stream);
int sr_densify_gather(int n_splats, int row_floats, const float* src, float* dst, const int* dest, int mode,
                      const float* log_scales, int scale_cols, const float* rotations, const float* unit_normals,
                      void* hip_stream);

/* Fused MLP chains of the SplatFields deform network's `GeneralMLP`s (reference utils/time_utils.py:123-191; SURVEY.md
 * section 8f row 4): n_points points are carried through a list of OPS in ONE kernel, the running state (<= 16 * hidden_tiles
 * channels per point) in registers, exact fp32 MFMA.  An op applies one packed matrix to [mem_tiles x 16 channels read from
 * `src` (row stride src_row floats, rows 16-byte aligned) | reg_tiles x 16 channels of the running state], starting from
 * `bias` (or 0), and produces out_tiles x 16 channels, which then go through `epilogue`:
 *   SR_MLP_LEAKY  leaky ReLU with negative_slope         -- a forward layer: y = act(W [x0 | h] + b); the reference
 *                 concatenates the skip input IN FRONT of the hidden state, hence memory channels first;
 *   SR_MLP_MASK   times leaky'(.) read off the sign of `mask` (the saved activation of the layer below, row stride mask_row)
 *                 -- a backward step: dZ_below = (W_hidden^T dZ) * act'(.);
 *   SR_MLP_NONE   nothing.
 * The result becomes the new running state unless keep_state != 0, and, if `store` is set, its first store_channels
 * channels are written to (store_accumulate: added to) store[point * store_row + channel]: the saved activations / output
 * of the forward, the dZ of every layer (for the weight-gradient GEMMs) and dL/dx0 of the backward.  hidden_tiles = 4 or 8
 * bounds out_tiles and reg_tiles; tile counts of the inputs are even.  Matrices arrive packed for the MFMA K order (layout:
 * csrc/mlp.hip header; packer: splatfields_amd/fused_mlp.py), biases padded to 16 * out_tiles floats.  Host side that builds
 * the forward and backward op lists of a GeneralMLP: splatfields_amd/fused_mlp.py. */
#define SR_MLP_MAX_OPS 24
#define SR_MLP_NONE 0
#define SR_MLP_LEAKY 1
#define SR_MLP_MASK 2
typedef struct SrMlpOp {
    const float* w_packed;
    const float* bias;         /* may be null */
    const float* src;          /* memory input channels (mem_tiles > 0) */
    const float* mask;         /* SR_MLP_MASK */
    float* store;              /* may be null */
    unsigned int* sign_store;  /* may be null: word [point * 4 + k], bit 4 t + i = (result channel 16 t + 4 k + i > 0) */
    const unsigned int* mask_bits; /* SR_MLP_MASK: if set, leaky'(.) comes from these bits (the layout sign_store writes) instead of `mask` */
    int out_tiles;
    int mem_tiles;             /* even */
    int reg_tiles;             /* even */
    int src_row;
    int epilogue;
    int mask_row;
    int store_row;
    int store_channels;
    int store_accumulate;
    int keep_state;
} SrMlpOp;
int sr_mlp_chain(int n_points, int hidden_tiles, int n_ops, const SrMlpOp* ops, float negative_slope, void* hip_stream);

/* Packs matrices for sr_mlp_chain on the device (training repacks every step: the weights change, and ResField layers
 * compose W + delta(frame) per step -- reference utils/resfields.py:378-405).  Job j builds the packed form of the matrix
 *   A[r][c'],  r < 16 * out_tiles,  c' < mem_pad + reg_width:
 *     c' <  mem_pad:  column c = c' of the memory block   (zero unless c < n_mem),  source column mem_col0 + c
 *     c' >= mem_pad:  column c = c' - mem_pad of the register block (zero unless c < n_reg), source column reg_col0 + c
 *   A[r][.] = 0 unless r < n_rows;  element = transposed ? W[column][row0 + r] : W[row0 + r][column],  W row stride `ld`
 * (transposed jobs build the backward's W^T blocks straight from the layer's weight), and, when bias_dst is set, copies
 * n_bias floats of bias_src to bias_dst and zero-fills it up to 16 * out_tiles.  mem_pad and reg_width are multiples of 32 / 16
 * with an even total tile count.  dst holds 16 * out_tiles * (mem_pad + reg_width) floats, 16-byte aligned. */
#define SR_MLP_MAX_PACK_JOBS 32
typedef struct SrMlpPackJob {
    const float* w;
    const float* bias_src;     /* may be null */
    float* dst;
    float* bias_dst;           /* may be null */
    int ld;
    int transposed;
    int row0, n_rows;
    int n_mem, mem_pad, mem_col0;
    int n_reg, reg_width, reg_col0;
    int out_tiles;
    int n_bias;
} SrMlpPackJob;
int sr_mlp_pack(int n_jobs, const SrMlpPackJob* jobs, void* hip_stream);

/* Weight and bias gradients of the layers of a fused MLP: for every job
 *   dw[m * dw_row + dw_col0 + c] = sum over points p of dz[p * dz_row + m] * x[p * x_row + c],   m < job.m, c < job.k
 *   db[m]                        = sum over points p of dz[p * dz_row + m]                       (when db is set)
 * -- dW_l = dZ_l^T [h_in | h_{l-1}] as one job per column block, all layers of the network in ONE launch: the contraction
 * runs over the 10^5 points and the result is tiny, which library GEMMs serialise (csrc/mlp.hip).  dz and x rows are 16-byte
 * aligned with row strides that are multiples of 4 floats; entries of a row beyond m / k up to the row stride may hold
 * anything finite or not (they only reach results that are dropped).  Results are written, not accumulated; the sum over the
 * points has a fixed order (deterministic).  `workspace`: sr_mlp_weight_grad_workspace(...) bytes, 16-byte aligned. */
#define SR_MLP_MAX_GRAD_JOBS 16
#define SR_MLP_MAX_GRAD_TASKS 128      /* 64 x 64 blocks over all jobs */
typedef struct SrMlpGradJob {
    const float* dz;
    const float* x;
    float* dw;
    float* db;                 /* may be null */
    int dz_row, m;
    int x_row, k;
    int dw_row, dw_col0;
} SrMlpGradJob;
size_t sr_mlp_weight_grad_workspace(int n_points, int n_jobs, const SrMlpGradJob* jobs);   /* 0 for an unsupported job list */
int sr_mlp_weight_grad(int n_points, int n_jobs, const SrMlpGradJob* jobs, void* workspace, size_t workspace_bytes, void* hip_stream);

/* Input matrix of a GeneralMLP (reference utils/time_utils.py:9-57 `get_embedder`, :178-181): row p of x0 [N, row] =
 * [ x (3) | sin(2^0 x) | cos(2^0 x) | ... | sin(2^(L-1) x) | cos(2^(L-1) x) | features (n_features) | time encoding | zeros up to `row` ],
 * the layout sr_mlp_chain reads; `row` a multiple of 4 (the fused MLP pads to 32).  `time` (may be NULL): one value per point,
 * encoded as [ t | sin(2^0 t) | cos(2^0 t) | ... | sin(2^(TL-1) t) | cos(2^(TL-1) t) ], TL = time_multires -- the time embedding the
 * reference appends to the features of every network (utils/time_utils.py:455-456); it receives no gradient.  Backward: given
 * dL/dx0, d_xyz = g[0:3] + sum_j 2^j (cos(2^j x) g_sin_j - sin(2^j x) g_cos_j) and d_features = the feature columns of g
 * (either may be NULL). */
int sr_mlp_input_forward(int n_points, int multires, int n_features, int time_multires, int row, const float* xyz, const float* features,
                         const float* time, float* x0, void* hip_stream);
int sr_mlp_input_backward(int n_points, int multires, int n_features, int row, const float* xyz, const float* dL_dx0, float* dL_dxyz,
                          float* dL_dfeatures, void* hip_stream);
/* Top of the backward of a GeneralMLP (the activation follows EVERY layer, the last one included: reference
 * utils/time_utils.py:178-191): g[p][c] = dL_dy[p][c] * (y[p][c] > 0 ? 1 : negative_slope) for c < out_features, 0 for
 * out_features <= c < row -- the padded matrix the backward chain's first op and the last layer's weight gradient read. */
int sr_mlp_top_gradient(int n_points, int out_features, int row, const float* y, const float* dL_dy, float negative_slope, float* g,
                        void* hip_stream);

/* ResField weights of the current frame for all layers of one network in one launch (reference utils/resfields.py:185,229,
 * 294-300,378-405 in the configuration GeneralMLP builds -- compression 'vm', mode 'lookup', fuse 'add'):
 *   out[j] = w[j] + sum_k weights_t[frame * rank + k] * matrix_t[k * count + j],   j < count = out_features * in_features.
 * `frame` points at an int64 on the device (the reference derives frame_id on the device: no host synchronisation).
 * Backward, given d_out = dL/d out: d_matrix_t[k * count + j] = weights_t[frame * rank + k] * d_out[j];
 * d_weights_t [capacity, rank] = 0 except row `frame` = matrix_t . d_out (summed in a fixed order); dL/dw = d_out itself.
 * count a multiple of 4, 16-byte aligned arrays, rank <= SR_RESFIELD_MAX_RANK. */
#define SR_RESFIELD_MAX_JOBS 16
#define SR_RESFIELD_MAX_RANK 64
typedef struct SrResFieldJob {
    const float* w; const float* weights_t; const float* matrix_t; float* out;    /* forward */
    const float* d_out; float* d_matrix_t; float* d_weights_t;                     /* backward (either output may be NULL) */
    int count, rank, capacity;
} SrResFieldJob;
int sr_resfield_compose(int n_jobs, const SrResFieldJob* jobs, const long long* frame, void* hip_stream);
size_t sr_resfield_backward_workspace(int n_jobs, const SrResFieldJob* jobs);     /* 0 for an unsupported job list */
int sr_resfield_backward(int n_jobs, const SrResFieldJob* jobs, const long long* frame, void* workspace, size_t workspace_bytes,
                         void* hip_stream);

/* Tri-plane feature lookup of the deform network's encoder -- the per-point half of the reference's VarTriPlaneEncoder.forward
 * (scene/tripFields.py:430-436): F.grid_sample (bilinear, zero padding, align_corners = False) of three planes [3, C, H, W] at
 * the (x, y), (y, z), (z, x) projections of the points, features concatenated plane-major: out [N, 3 C].  C a multiple of 4.
 * The plane generator (the reference's diffusers-based decoder, :176-204) is the caller's; it hands over `planes`.
 * `planes_texel_major` [3, H, W, C] is written by the forward (a transposed copy: a corner becomes C contiguous floats) and
 * read again by the backward.
 * Backward: dL_dpoints [N, 3] (gather) and / or dL_dplanes [3, C, H, W] (either may be NULL).  The plane gradient is
 * accumulated in 64-bit fixed point with integer atomics -- bit-reproducible, no floating-point atomics: the (point, plane)
 * pairs are binned by tile of the plane and every tile is summed in LDS by one workgroup; `workspace` holds
 * sr_triplane_backward_workspace(N, C, H, W) bytes (tile counters and the binned lists; needed only with dL_dplanes; 0 for
 * unsupported sizes: C a multiple of 4, <= 128). */
size_t sr_triplane_backward_workspace(int n_points, int channels, int height, int width);
int sr_triplane_forward(int n_points, int channels, int height, int width, const float* planes, float* planes_texel_major,
                        const float* points, float* out, void* hip_stream);
int sr_triplane_backward(int n_points, int channels, int height, int width, const float* planes_texel_major, const float* points,
                         const float* dL_dout, float* dL_dplanes, float* dL_dpoints, void* workspace, void* hip_stream);

/* Diagnostics for the parity tests: byte offsets of four arrays inside the opaque buffers of a view with these sizes
 * (`instances` = the capacity the binning buffer was carved for):
 *   out[0]  geom:    tile_start  uint32[tiles + 1]   first list entry of every 16x16 tile (row-major tiles)
 *   out[1]  binning: sorted_id   uint32[instances]   splat index of every list entry, per tile front to back
 *   out[2]  geom:    total       uint32[4]           [0] = tile-splat instances
 *   out[3]  geom:    offsets     uint32[N]           first gradient slot of every splat
 * The per-tile lists are the counterpart of [EXT]'s sorted point_list + ranges (binningBuffer / imgBuffer). */
int sr_debug_layout(int n_splats, int height, int width, long long instances, size_t* out4);

/* Work counters of the backward blend, accumulated over the sr_backward calls of a library built with -DSR_BWD_STATS
 * (bench.py's pairs_evaluated / pairs_blended; the product build leaves them out and returns zeros):
 * out16 = { list entries replayed, (4x4 quad, entry) pairs, 16-entry buckets, (pixel, entry) pairs evaluated,
 *           pairs blended, list chunks, 0, 0,  then 8 shader-clock totals (summed over wavefronts) of the kernel's phases:
 *           preamble, test, barrier, slot assignment + scatter, barrier, replay, barrier, combine }.
 * Synchronises the device; reset != 0 clears the counters afterwards. */
int sr_debug_backward_stats(unsigned long long* out16, int reset);

/* Host-synchronisation counters of the forward (process-wide, since load or the last reset):
 *   out4 = { host waits inside sr_forward (one per call: the instance-count read-back),  sr_forward_async calls (none waits),
 *            tickets that were redeemed while stage 1 of their forward was still running (the host waited in sr_ticket_wait),
 *            tickets ever created (the pool's size) }. */
int sr_debug_counters(long long* out4, int reset);

/* debug = true (SrView.debug): every stage is followed by a stream synchronisation + error check, and a FAILED stage writes
 * the call's arguments to snapshot_fw.dump / snapshot_bw.dump in the working directory before the error is returned -- what
 * [EXT] does with `debug` set (its snapshot_fw.dump / snapshot_bw.dump; reference gaussian_renderer/__init__.py:71 passes
 * pipe.debug).  Format: "SRSNAP1\0", failing stage [32], then records {name[16], uint64 bytes, payload}: view scalars,
 * count, camera arrays, every per-splat input that was passed, for the backward the upstream gradients (a record whose
 * device-to-host copy fails has 0 bytes).  sr_debug_snapshot runs that path for given arguments without a failure (tests). */
int sr_debug_snapshot(const SrView* view, const SrSplats* splats, const float* dL_dcolor, const float* dL_ddepth,
                      const float* dL_dalpha, int backward);

/* Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the
 * live roofline figure; off by default, adds two event records per launch when on).
 * sr_profile_collect synchronises the recorded events, ADDS the elapsed milliseconds and launch counts
 * per stage into the caller's arrays of SR_PROFILE_STAGES entries, and clears the recording. */
#define SR_PROFILE_STAGES 7 /* preprocess, scan, emit, sort_tiles, render_forward, render_backward, preprocess_backward */
int sr_profile_enable(int on);
int sr_profile_collect(double* ms_sum, long long* launches);
const char* sr_profile_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* SPLATRASTER_H */
