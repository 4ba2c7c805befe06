/*
 * gsplat.h -- C-ABI of libgsplat_hip.so, the MI355X (gfx950) differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary for the hot path of Ainsley0917/gaussian-splatting:
 * the slot the reference fills with the third-party `diff_gaussian_rasterization` extension
 * (reference gaussian_renderer/__init__.py:19-27 import, :549-668 adapter).  The reference is
 * Python, so the binding a maintainer adds is a ctypes stub (the one shipped here
 * is gaussian-splatting_amd/_lib.py + rasterizer.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (fp32 / int32 / uint32 / uint64, contiguous) unless named
 *     host_*; the library never allocates, frees or keeps caller memory; scratch is passed in.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     except gs_scan_tiles when host_total != NULL (one 4-byte read-back of the pair count D).
 *   - every function returns 0 on success, <0 on error (GS_ERR_*); gs_last_error() gives the
 *     message of the calling thread's last error.  No function allocates device memory.
 *   - matrices are row-major 4x4 applied as out[r] = sum_c M[r][c] * (p,1)[c], exactly as the
 *     reference's built-in rasterizer applies them (gaussian_renderer/__init__.py:122,125).
 *   - semantics (culling, hard 3-sigma integer boxes, far-to-near compositing with alpha_acc, no
 *     alpha clamp, background added un-attenuated) are those of the reference's built-in
 *     PyTorch rasterizer, gaussian_renderer/__init__.py:64-442 (the parity oracle).
 *
 * Re-entrancy: no global mutable state except the thread-local error string and, for
 * gs_sort_pairs_static only, the per-GPU static scratch array documented there.
 */
#ifndef GSPLAT_H
#define GSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_OK 0
#define GS_ERR_ARG (-1)  /* bad argument (null pointer, size, alignment, missing/duplicate input) */
#define GS_ERR_HIP (-2)  /* a HIP call or kernel launch failed */
#define GS_ERR_SIZE (-3) /* size beyond a stated limit */

/* ABI version of this header.  It changes whenever an entry point's argument list, a buffer layout (splat records,
 * gradient rows, sh_jac rows) or a precondition changes: a binding compares gs_version() with the GS_ABI_VERSION it was
 * written against before its first call (gaussian-splatting_amd/_lib.py does) -- a mismatch means shifted arguments.
 * 100: round 1.  200: round 2 (sh_jac arguments, 64-byte gradient rows with six doubles, gs_identify_tile_ranges writes the
 * whole table and wants 16-byte aligned keys).  300: round 3.  303: this header (gs_preprocess_fwd_views and gs_preprocess_bwd_views added, gs_scan_tiles takes the sorted depth keys). */
#define GS_ABI_VERSION 303
int gs_version(void);            /* == GS_ABI_VERSION of the header the library was built from */
const char* gs_last_error(void); /* message of this thread's last failing call */
const char* gs_source_sha(void); /* hash of the sources this library was compiled from (csrc/source_sha.py), "unknown" outside the Makefile */
int gs_tile_size(void);          /* 16: tile edge used by keys and ranges */
int gs_grad_row_floats(void);    /* 16: floats (4-byte units) per per-Gaussian gradient row (see gs_render_bwd) */

/*
 * Per-Gaussian forward stage.  Replaces GaussianRasterizer.forward's vectorised prologue
 * (gaussian_renderer/__init__.py:108-149: compute_colors_from_sh :185, projection :116-133,
 * compute_cov3D :210 / cov_from_upper_triangular :266, project_cov3D_to_2D :291) and the
 * per-Gaussian tests at the top of the splat loop (:368-402).
 * Exactly one of shs (N,K,3) / colors_precomp (N,3); exactly one of (scales (N,3), rotations
 * (N,4)) / cov3D_precomp (N,6).  rotations and splat must be 16-byte aligned.
 * [tile_row_begin, tile_row_end) is the band of 16-pixel tile rows this call owns (whole frame:
 * 0 .. ceil(H/16)); tiles_touched counts only tiles inside the band (tile-row sharded inference,
 * BASELINE.json config 5); radii are independent of the band.
 * Outputs: splat (N x 12 floats, 48-byte records, layout in csrc/gsplat_common.h; rows of culled
 * Gaussians are zero), radii (N) int32 = ceil(3 sigma_max) or 0 when not drawn,
 * tiles_touched (N) = number of 16x16 tiles the integer box overlaps.  depth_keys / ids (N each) /
 * rects (N x 2 uint32: x_min | x_max << 16, y_min | y_max << 16; all zero when not drawn), all three
 * or none NULL: input of the binning stage, depth_keys[i] = 32-bit key whose
 * ascending order is descending depth (0xFFFFFFFF for Gaussians that are not drawn, so they sort last), ids[i] = i.
 * sh_jac (optional, NULL = not wanted; only with shs; 16-byte aligned): side table for gs_preprocess_bwd, one row of
 * gs_sh_jac_row_floats() = 12 floats (48 bytes) per Gaussian (zeros for Gaussians that are not
 * drawn): floats 0..8 = d colour_c / d dir_a = sum_k (d basis_k / d dir_a) sh_kc at index 3 a + c (the
 * view-direction Jacobian autograd forms from utils/sh_utils.py:40-96), word 9 = bit c set when channel c was inside
 * the clamp of gaussian_renderer/__init__.py:206, words 10..11 unused.  With it the backward does not read the
 * N x K x 3 coefficients a second time.
 */
int gs_sh_jac_row_floats(void);
int gs_preprocess_fwd(int N, int K, int sh_degree, int H, int W, const float* viewmatrix,
                      const float* projmatrix, const float* campos, const float* means3D,
                      const float* scales, const float* rotations, const float* cov3D_precomp,
                      const float* opacities, const float* shs, const float* colors_precomp,
                      void* splat, int32_t* radii, uint32_t* tiles_touched, uint32_t* depth_keys,
                      uint32_t* ids, void* rects, int tile_row_begin, int tile_row_end, float* sh_jac,
                      void* stream);

/*
 * Binning stage.  The reference has one global far-to-near order (torch.argsort at
 * gaussian_renderer/__init__.py:153) and loops over Gaussians (:366).  Here that order is the
 * per-Gaussian depth sort: gs_sort_pairs on (depth_keys, ids), all 32 bits -> `order` = Gaussian ids
 * by descending depth, ties by ascending index, Gaussians that are not drawn last.  Pairs (tile id,
 * Gaussian id) are emitted in that order and then STABLY sorted on the tile id alone, which leaves
 * every tile's list far-to-near.
 */
size_t gs_scan_workspace_bytes(long long N);
/* For every depth rank r: rect_in_order[r] (N x 2 uint32) = rects[order[r]], the box of the Gaussian
 * of that rank, and offsets[r] = exclusive scan of the number of tiles of those boxes inside the band
 * of tile rows.  *total_dev = D; host_total (host pointer, may be NULL) receives D after a stream
 * synchronise.  This gather is the only random access of the binning stage.
 * sorted_depth_keys (may be NULL): the depth keys in rank order, as the depth sort left them.  A rank whose key is
 * 0xFFFFFFFF holds a Gaussian that is not drawn (gs_preprocess_fwd gives those that key and an all-zero box): with the
 * keys given, such a rank's box is not fetched (rect_in_order[r] = 0, 0) -- same result, fewer random reads. */
int gs_scan_tiles(long long N, const void* rects, const uint32_t* order, const uint32_t* sorted_depth_keys,
                  void* rect_in_order, uint32_t* offsets, uint32_t* total_dev, void* workspace, uint32_t* host_total,
                  int tile_row_begin, int tile_row_end, void* stream);
/* keys[o] = tile id (row-major), vals[o] = Gaussian index order[r], for o in [offsets[r], offsets[r] +
 * tiles of rect_in_order[r]), tiles of one Gaussian in row-major order.  Pairs with o >= capacity (the
 * number of pairs keys/vals can hold) are dropped, so a caller may size the buffers from an estimate,
 * launch without reading D back first, and check D <= capacity afterwards. */
int gs_duplicate_with_keys(int N, int W, const void* rect_in_order, const uint32_t* order, const uint32_t* offsets,
                           uint32_t* keys, uint32_t* vals, long long capacity, int tile_row_begin,
                           int tile_row_end, void* stream);
size_t gs_sort_workspace_bytes(long long D);
/* Stable LSD radix sort of (uint32 key, uint32 value) pairs on key bits [bit_begin, bit_end), 8 per
 * pass, at most 4 passes.  Input in (keys_a, vals_a); *result_in_b (host int) = 1 when the sorted
 * pairs ended in (keys_b, vals_b).  Used twice per frame: depth keys of the N Gaussians (bits 0..32)
 * and tile ids of the D pairs (bits 0..tile_bits).  count_dev (device pointer, may be NULL): the
 * number of pairs is min(*count_dev, D) and D is only the capacity that sizes grids and workspace
 * (gs_scan_tiles' total_dev can be passed straight in: no host read-back of the pair count). */
int gs_sort_pairs(long long D, const uint32_t* count_dev, int bit_begin, int bit_end, uint32_t* keys_a,
                  uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, void* workspace, size_t workspace_bytes,
                  int* result_in_b, void* stream);
/*
 * gs_sort_pairs for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same sort, same
 * arguments minus the workspace: the histogram tables live in a file-scope device array of the library
 * (4.5 MiB per GPU the library runs on, held until the process ends), so the call allocates nothing --
 * no hipMalloc / hipFree, no allocating library routine.  At most gs_sort_pairs_static_max() = 2^24
 * pairs (capacity D); more returns GS_ERR_SIZE and gs_last_error() says so.  The device that owns
 * `stream` must be the current device.  All calls on one GPU share that GPU's array: calls that overlap
 * on one GPU (two streams, two threads) race -- serialise them or use gs_sort_pairs with workspaces of
 * their own.  Every call rewrites or clears what it reads, so back-to-back calls need no clean-up.
 * First call per device: it resolves that device's address of the array (hipGetSymbolAddress, which may
 * load the library's code object) -- set-up work that does not belong inside a stream capture.  Make one
 * call with D = 0 on every device beforehand (it does the set-up, launches nothing and returns GS_OK);
 * every later call on that device only launches kernels and memsets.
 */
long long gs_sort_pairs_static_max(void);
int gs_sort_pairs_static(long long D, const uint32_t* count_dev, int bit_begin, int bit_end, uint32_t* keys_a,
                         uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, int* result_in_b, void* stream);
/* ranges[2t], ranges[2t+1] = [start, end) of tile t in the sorted list (0,0 when empty); count_dev as
 * in gs_sort_pairs.  keys = tile ids < num_tiles in ascending order (16-byte aligned).  Every entry of
 * `ranges` (8-byte aligned) is written by the call -- the entries of empty tiles too, by whoever sees the
 * keys on either side of them -- so the table needs no clearing.  Keys >= num_tiles (none are emitted by
 * gs_duplicate_with_keys; sorted, they can only end the list) own no entry: the last valid tile ends where
 * they begin and the tiles behind it come out empty. */
int gs_identify_tile_ranges(long long D, const uint32_t* count_dev, int num_tiles, const uint32_t* keys,
                            uint32_t* ranges, void* stream);

/*
 * Per-tile compositing.  Replaces splat_gaussians (gaussian_renderer/__init__.py:336-442).
 * out_image (3,H,W) = bg + sum alpha_i c_i, out_depth (H,W), out_alpha (H,W).  aux_rgb (3,H,W),
 * optional, receives sum alpha_i c_i without bg (needed by gs_render_bwd).  t_eps in [0,1): a pixel
 * stops once 1 - alpha_acc < t_eps; 0 = never (the reference's exact semantics).  Only pixels of
 * tile rows [tile_row_begin, tile_row_end) are computed and written.
 * zero_buf / zero_bytes (optional, 16-byte aligned, multiple of 16): a buffer to clear on the side --
 * meant for gs_render_bwd's grad_rows: this kernel is VALU-bound and leaves HBM idle, so the clear
 * is free here while a separate memset costs its full HBM time.
 */
int gs_render_fwd(int H, int W, const float* bg, const void* splat, const uint32_t* ranges,
                  const uint32_t* point_list, float* out_image, float* out_depth, float* out_alpha,
                  float* aux_rgb, float t_eps, int tile_row_begin, int tile_row_end, void* zero_buf,
                  size_t zero_bytes, void* stream);

/*
 * Backward of gs_render_fwd (the reference gets it from torch autograd).  g_image (3,H,W),
 * g_depth / g_alpha (H,W) optional.  grad_rows (N rows of gs_grad_row_floats() = 16 floats = 64 bytes, 16-byte
 * aligned) must be zero on entry and receives, per Gaussian, six DOUBLES (bytes 0..47): the sums over its pixels
 * of pv {dx, dy, dx^2, dx dy, dy^2, 1} with pv = exp(power) dL/dw and dx, dy measured from the centre of its
 * integer box (csrc/gsplat_common.h) -- fp64 atomics, so they do not depend on the order in which the tiles
 * arrive -- and four floats (bytes 48..63): d/d rgb, d/d depth.  The rows are an opaque hand-over to
 * gs_preprocess_bwd, which turns the moments into d/d screen mean, d/d conic and d/d opacity.
 * t_eps must equal the forward's.  Gaussian indices (point_list) must be below 2^30 (gs_preprocess_fwd rejects more).
 * What it differentiates: the forward as gs_render_fwd ran it.  A pixel that stopped (1 - alpha_acc <= t_eps; at
 * t_eps = 0: alpha_acc == 1 exactly) composites nothing behind the stop, and the part of a pixel's sum behind the
 * Gaussian that stops it is taken as 0 -- the exact gradient of the truncated forward (bounds against the reference's
 * never-stopping compositing: tests/test_stop_threshold_bound.py).
 * Documented deviation from the reference's autograd, the w == 1 corner: a Gaussian whose weight at a pixel is exactly
 * 1.0 (opacity 1.0 AND the screen mean on the pixel centre; the reference's sigmoid opacities cannot reach it) drives the
 * transmittance to exactly 0.  Autograd gives dL/dw_i = T_i (u_i - S_i) there, S_i = what the Gaussians behind it would
 * show through it (a one-sided derivative that needs the hidden part of the list); this backward replays front to back
 * and returns T_i u_i.  Only dL/d opacity_i is affected (d power / d mean and d conic vanish at dx = dy = 0).  Pinned by
 * tests/test_w_equals_one.py: the CPU oracle equals autograd, the HIP path equals oracle + T_i S_i.
 * Conditioning: for a pixel that does not stop, the part behind Gaussian i enters as (total - prefix) / (1 - w_i); with
 * 1 - w_i below ~1e-6 (opacity within 1e-6 of 1 and the pixel within ~1e-3 px of the mean) at t_eps = 0 the fp32 residue
 * of that difference is amplified beyond the stated tolerance -- at any t_eps >= 1e-6 such a visit stops the pixel.
 */
int gs_render_bwd(int H, int W, const void* splat, const uint32_t* ranges, const uint32_t* point_list,
                  const float* aux_rgb, const float* out_depth, const float* out_alpha,
                  const float* g_image, const float* g_depth, const float* g_alpha, float* grad_rows,
                  float t_eps, int tile_row_begin, int tile_row_end, void* stream);

/*
 * gs_preprocess_fwd for k views of the SAME Gaussians: equal to k calls of gs_preprocess_fwd, one per descriptor, with the
 * per-Gaussian inputs -- above all the 12 K bytes of SH coefficients -- read once for all views.  No counterpart in the
 * reference (one view per iteration, train.py:129-135); it is the forward of the multi-view training step of SURVEY.md 8e
 * (config 4: 8 views per step, parallel.ViewShardedStep).
 * views: HOST array of n_views descriptors (device pointers inside; read during the call, not kept): per view the camera
 * and every output of gs_preprocess_fwd, with that function's meaning, alignment and NULL rules.
 * One launch per GS_MAX_FWD_VIEWS views when shs is given with 16-byte aligned rows and either every view or no view has
 * an sh_jac; any other combination of inputs runs view by view inside the call (same results).
 */
#define GS_MAX_FWD_VIEWS 8
typedef struct GsFwdView {
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    void* splat;
    int32_t* radii;
    uint32_t* tiles_touched;
    uint32_t* depth_keys;
    uint32_t* ids;
    void* rects;
    float* sh_jac;
    int32_t H, W, tile_row_begin, tile_row_end;
} GsFwdView;
int gs_preprocess_fwd_views(int n_views, const GsFwdView* views, int N, int K, int sh_degree, const float* means3D,
                            const float* scales, const float* rotations, const float* cov3D_precomp,
                            const float* opacities, const float* shs, const float* colors_precomp, void* stream);

/*
 * Backward of gs_preprocess_fwd: chain rule from grad_rows to the rasterizer's inputs (autograd of
 * gaussian_renderer/__init__.py:116-149,185-334 and utils/sh_utils.py:40-96 in the reference).
 * Any g_* may be NULL.  g_means2D (N,3) receives the densification statistic the reference's
 * trainer reads from viewspace_points.grad (train.py:185-193, scene/gaussian_model.py:701-713):
 * d loss / d screen mean scaled to NDC units (x W/2, y H/2, 0).  Gaussians with radii == 0 get
 * zeros everywhere.
 * accumulate != 0: the gradients of this view are ADDED to what the g_* buffers hold (rows of
 * Gaussians with radii == 0 are not touched) -- several views of one batch summed without a pass of
 * their own, as the multi-view step of parallel.SplatExchange does (no counterpart in the reference,
 * which renders one view per iteration, train.py:129-135).  g_means2D is a per-view statistic and is
 * overwritten either way.
 * sh_jac (optional): the table gs_preprocess_fwd wrote for the same Gaussians, view and shs.
 * Given (and shs rows of at most 48 floats, 16-byte aligned), the whole chain rule runs as ONE launch that
 * never reads shs; NULL = two launches, the second of which re-reads the coefficients.
 */
int gs_preprocess_bwd(int N, int K, int sh_degree, int H, int W, const float* viewmatrix,
                      const float* projmatrix, const float* campos, const float* means3D,
                      const float* scales, const float* rotations, const float* cov3D_precomp,
                      const float* opacities, const float* shs, const float* colors_precomp, const int32_t* radii,
                      const float* grad_rows, float* g_means3D, float* g_means2D, float* g_scales,
                      float* g_rotations, float* g_cov3D_precomp, float* g_opacities, float* g_shs,
                      float* g_colors_precomp, int accumulate, const float* sh_jac, void* stream);

/*
 * gs_preprocess_bwd for k views of the SAME Gaussians, summed: equal to gs_preprocess_bwd(view 0, accumulate) followed by
 * gs_preprocess_bwd(view v, accumulate = 1) for v = 1..k-1 -- the sums are taken in that order -- with every output row
 * written once.  No counterpart in the reference (one view per iteration, train.py:129-135); it is the backward of the
 * multi-view training step of SURVEY.md 8e (config 4: 8 views per step, parallel.ViewShardedStep), where the view-by-view
 * form reads and rewrites 59 floats per Gaussian and view.
 * views: HOST array of n_views descriptors (device pointers inside; read during the call, not kept).  Per view: the
 * camera, the radii and the sh_jac table its gs_preprocess_fwd wrote, the grad_rows its gs_render_bwd wrote, and
 * g_means2D (may be NULL), that view's own (N,3) densification statistic.  Everything else as gs_preprocess_bwd.
 * One launch per GS_MAX_BWD_VIEWS views when shs is given with rows of at most 48 floats (16-byte aligned) and every view
 * has its sh_jac; any other combination of inputs runs the view-by-view form inside the call (same results).
 */
#define GS_MAX_BWD_VIEWS 8
typedef struct GsBwdView {
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    const int32_t* radii;
    const float* grad_rows;
    const float* sh_jac;
    float* g_means2D;
    int32_t H, W;
} GsBwdView;
int gs_preprocess_bwd_views(int n_views, const GsBwdView* views, int N, int K, int sh_degree, const float* means3D,
                            const float* scales, const float* rotations, const float* cov3D_precomp,
                            const float* opacities, const float* shs, const float* colors_precomp, float* g_means3D,
                            float* g_scales, float* g_rotations, float* g_cov3D_precomp, float* g_opacities, float* g_shs,
                            float* g_colors_precomp, int accumulate, void* stream);

/*
 * Training loss, fused (the step right after the rasterizer in every iteration).
 * Replaces l1_loss (utils/loss_utils.py:10-21), ssim / calculate_ssim / ssim_loss (:74-115, :118-140,
 * :193-205; 11x11 window of sigma 1.5, :38-71) and their autograd as train.py:141-146 combines them.
 * img, gt: (C,H,W) fp32.  loss_out (device, 3 floats) = (w_l1 * L1 + w_ssim * mean SSIM, L1, mean SSIM);
 * grad (C,H,W, may be NULL) = d loss_out[0] / d img.  The reference's trainer uses w_l1 = 1 - lambda,
 * w_ssim = +lambda (its "1 - ssim_loss(...)" is SSIM itself); pass w_ssim = -lambda for the D-SSIM of
 * the 3DGS paper (the constant lambda is then missing from loss_out[0]).
 */
size_t gs_l1_ssim_workspace_bytes(int C, int H, int W);
int gs_l1_ssim(int C, int H, int W, const float* img, const float* gt, float w_l1, float w_ssim, float* loss_out,
               float* grad, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Optimizer step + densification statistics, fused (the step right after the backward).
 * Replaces torch.optim.Adam(l, lr=0.0, eps=1e-15).step() over the reference's six parameter groups
 * (scene/gaussian_model.py:264-297, train.py:217), the max_radii2D update (train.py:184-187) and
 * add_densification_stats (scene/gaussian_model.py:701-713) -- one launch instead of ~40.
 * Arithmetic = torch's Adam (no weight decay / amsgrad / maximize) in fp32, operation by operation.
 * `groups` is a HOST array; its pointers are device pointers.  A group is a (rows, row_len) fp32 matrix;
 * param / exp_avg / exp_avg_sq are contiguous, grad may have a row stride (0 = contiguous) so that
 * column ranges of the rasterizer's (N,K,3) SH gradient can be consumed in place (f_dc: row_len 3,
 * stride 3K; f_rest: grad + 3, row_len 3(K-1), stride 3K).
 * step = the 1-based count of this update (torch's state["step"] after incrementing).
 * Statistics (skipped when radii is NULL or num_views == 0), for every Gaussian i and each view v of the
 * step with radii[v][i] > 0, in view order:  xyz_gradient_accum[i] += |means2D_grad[v][i][0:2]|,
 * denom[i] += 1 (both only when means2D_grad is given), max_radii2D[i] = max(., radii[v][i]).
 * The reference has one view per step (num_views = 1); parallel.SplatExchange returns them per view.
 */
typedef struct GsAdamGroup {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long long rows;
    long long grad_row_stride;   /* in floats; 0 = row_len */
    int row_len;
    int reserved;
    double lr;
} GsAdamGroup;
int gs_adam_densify_step(int num_groups, const GsAdamGroup* groups, double beta1, double beta2, double eps,
                         long long step, long long N, int num_views, const float* means2D_grad,
                         const int32_t* radii, float* xyz_gradient_accum, float* denom, float* max_radii2D,
                         void* stream);

/*
 * Trained-model PLY <-> device layout.  Replaces GaussianModel.load_ply / save_ply
 * (scene/gaussian_model.py:377-459, :333-375; attribute order :316-331; plyfile underneath).
 * gs_ply_parse_header (host): reads the ASCII header of a binary_little_endian PLY from the first n_bytes of
 * the file (the header must end inside them; file_size = size of the whole file, checked against the vertex
 * count) and fills the layout: vertex count, byte offset of the first record, record stride and the byte offsets of
 * x y z f_dc_0..2 opacity scale_0..2 rot_0..3 and of the f_rest_* properties in file order (what load_ply
 * keys on).  Other vertex properties (nx ny nz, ...) are skipped by size; elements after the vertices are
 * ignored.  gs_ply_unpack (device): `records` = the vertex records copied to the GPU as they are
 * (file_bytes + data_offset, num_vertices * stride_bytes, 4-byte aligned); writes the parameters exactly
 * as load_ply builds them -- xyz (N,3), features_dc (N,1,3), features_rest (N,K-1,3) (transposed from the
 * file's channel-major order), opacity (N,1), scaling (N,3), rotation (N,4) -- and/or the activated
 * tensors the rasterizer takes (:104-142): shs = cat(dc, rest) (N,K,3), sigmoid(opacity), exp(scaling),
 * normalize(rotation).  Any output may be NULL.  gs_ply_write_header / gs_ply_pack: the inverse, the
 * file save_ply writes (normals zero).
 */
#define GS_PLY_MAX_REST 72
typedef struct GsPlyLayout {
    long long num_vertices;
    long long data_offset;
    int stride_bytes;
    int sh_rest_count;
    int off_xyz[3];
    int off_f_dc[3];
    int off_opacity;
    int off_scale[3];
    int off_rot[4];
    int off_f_rest[GS_PLY_MAX_REST];
} GsPlyLayout;
int gs_ply_parse_header(const void* file_bytes, size_t n_bytes, size_t file_size, GsPlyLayout* layout);
/* returns the header length; writes it when buf is large enough; fills *layout (may be NULL) */
long long gs_ply_write_header(char* buf, size_t cap, long long num_vertices, int sh_rest_count, GsPlyLayout* layout);
int gs_ply_unpack(const GsPlyLayout* layout, const void* records, float* xyz, float* features_dc,
                  float* features_rest, float* opacity, float* scaling, float* rotation, float* shs,
                  float* opacities_act, float* scales_act, float* rotations_act, void* stream);
int gs_ply_pack(const GsPlyLayout* layout, void* records, const float* xyz, const float* features_dc,
                const float* features_rest, const float* opacity, const float* scaling, const float* rotation,
                void* stream);

/*
 * Squared distance of every point to its nearest other point.  Replaces distCUDA2
 * (scene/gaussian_model.py:729-746: N x N torch.cdist, diagonal = inf, row minimum, squared; out of memory
 * at 100 k points), which initialises the scales from the SfM cloud (:212-217).  points (N,3) fp32 device,
 * dist2 (N,) fp32 device; a single point gets +inf, as in the reference.  Uniform grid + the library's radix
 * sort, exact; O(N) memory (workspace: 256-byte aligned device memory).  Points in crowded neighbourhoods
 * (more than 4096 points in their 27 cells) are answered by an adaptive pass over Morton-sorted points, so the
 * cost stays near-linear on clouds of very uneven density.  Synchronises the stream twice (reads the bounding
 * box back to size the grid, and the number of crowded points).  stats (host, 5 ints, may be NULL): grid dims
 * x y z, the number of isolated points settled by the brute-force pass, the number answered by the adaptive pass.
 */
size_t gs_nearest_dist2_workspace_bytes(long long N);
int gs_nearest_dist2(long long N, const float* points, float* dist2, void* workspace, size_t workspace_bytes,
                     int* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_H */
