/*
 * gi2d.h -- C ABI of the MI355X (gfx950) 2D-Gaussian rasterizer that replaces the CUDA
 * extension behind GaussianImage++'s `gsplat` operator surface.
 *
 * Boundary being replaced: the 12 hot-path entries of the reference's pybind table
 *   gsplat/gsplat/cuda/csrc/ext.cpp:16-66 (C++ signatures in csrc/bindings.h).
 * Each function below cites the binding it stands in for.  Conventions:
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in `_host`;
 *   - inputs are borrowed, outputs are caller-allocated and are fully written by the call
 *     (no pre-zeroing needed: the reference's torch::zeros + kernel pair is fused);
 *   - layouts are the reference's: fp32 row-major [N,2]/[N,3]/[H,W,3], int32 ids, int64 keys;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work, they never
 *     synchronise, allocate or free -- safe for hipGraph capture;
 *   - return value: 0 (GI2D_OK) or a negative GI2D_ERR_* / positive hipError_t code.
 *     gi2d_last_error_string() describes the last failure of the calling thread.
 * Tiles are 16x16 pixels (csrc/config.h:1-4); tiles_x = ceil(W/16), tiles_y = ceil(H/16).
 */
#ifndef GI2D_H
#define GI2D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GI2D_OK 0
#define GI2D_ERR_INVALID_ARGUMENT (-1)
#define GI2D_ERR_WORKSPACE_TOO_SMALL (-2)
#define GI2D_ERR_UNSUPPORTED (-3)

#define GI2D_TILE 16
#define GI2D_TILE_LIST_CAP 256 /* forward.cu:553: only the first 256 entries of a tile are consumed */

typedef void *gi2d_stream_t;

const char *gi2d_version(void);
const char *gi2d_last_error_string(void);

/* ------------------------------------------------------------------ projection (a1-a4)
 * bindings.cu:1317-1381  project_gaussians_2d_forward_tensor      (Cholesky, means in NDC)
 * bindings.cu:1449-1513  project_gaussians_2d_covariance_forward_tensor (means in pixels)
 * bindings.cu:1384-1448  project_gaussians_2d_scale_rot_forward_tensor  (means in pixels)
 * Outputs: xys f32[N,2], depths f32[N] (=0), radii i32[N], conics f32[N,3], num_tiles_hit i32[N].
 * clip_thresh is accepted and unused, as in the reference kernels. */
int gi2d_project_gaussians_2d_forward(int num_points, float clip_coe, const float *means2d,
                                      const float *L_elements, unsigned img_height,
                                      unsigned img_width, int tiles_x, int tiles_y,
                                      float clip_thresh, float radius_clip, float *xys,
                                      float *depths, int32_t *radii, float *conics,
                                      int32_t *num_tiles_hit, gi2d_stream_t stream);
int gi2d_project_gaussians_2d_covariance_forward(int num_points, float clip_coe,
                                                 const float *means2d, const float *cov2d,
                                                 unsigned img_height, unsigned img_width,
                                                 int tiles_x, int tiles_y, float clip_thresh,
                                                 float radius_clip, float *xys, float *depths,
                                                 int32_t *radii, float *conics,
                                                 int32_t *num_tiles_hit, gi2d_stream_t stream);
int gi2d_project_gaussians_2d_scale_rot_forward(int num_points, float clip_coe,
                                                const float *means2d, const float *scales2d,
                                                const float *rotation, unsigned img_height,
                                                unsigned img_width, int tiles_x, int tiles_y,
                                                float clip_thresh, float radius_clip, float *xys,
                                                float *depths, int32_t *radii, float *conics,
                                                int32_t *num_tiles_hit, gi2d_stream_t stream);

/* bindings.cu:1517-1564 / :1565-1612 / :1614-1668  *_backward_tensor.
 * Outputs v_cov2d f32[N,3], v_mean2d f32[N,2], then v_L f32[N,3] (Cholesky / covariance) or
 * v_scale f32[N,2] + v_rot f32[N] (scale-rot).  Rows with radii<=0 are written as zeros.
 * The Cholesky and scale-rot variants reproduce the reference's double-counted off-diagonal
 * (backward2d.cu:39-40, :94-96). */
int gi2d_project_gaussians_2d_backward(int num_points, const float *means2d,
                                       const float *L_elements, unsigned img_height,
                                       unsigned img_width, const int32_t *radii,
                                       const float *conics, const float *v_xy,
                                       const float *v_depth, const float *v_conic, float *v_cov2d,
                                       float *v_mean2d, float *v_L_elements, gi2d_stream_t stream);
int gi2d_project_gaussians_2d_covariance_backward(int num_points, const float *means2d,
                                                  const float *cov2d, unsigned img_height,
                                                  unsigned img_width, const int32_t *radii,
                                                  const float *conics, const float *v_xy,
                                                  const float *v_depth, const float *v_conic,
                                                  float *v_cov2d, float *v_mean2d,
                                                  float *v_cov2d_elements, gi2d_stream_t stream);
int gi2d_project_gaussians_2d_scale_rot_backward(int num_points, const float *means2d,
                                                 const float *scales2d, const float *rotation,
                                                 unsigned img_height, unsigned img_width,
                                                 const int32_t *radii, const float *conics,
                                                 const float *v_xy, const float *v_depth,
                                                 const float *v_conic, float *v_cov2d,
                                                 float *v_mean2d, float *v_scale, float *v_rot,
                                                 gi2d_stream_t stream);

/* bindings.cu:44-63 compute_cov2d_bounds_tensor: conics f32[N,3], radii f32[N] (= radius.x). */
int gi2d_compute_cov2d_bounds(int num_pts, float clip_coe, const float *covs2d, float *conics,
                              float *radii, gi2d_stream_t stream);

/* ------------------------------------------------------------------ binning (a5-a8)
 * gsplat/gsplat/utils.py:248-249 (torch.cumsum + .item()): inclusive int32 scan; the total is
 * written to the DEVICE word *total (the caller decides whether/when to read it back). */
int gi2d_cumsum_tiles_hit(int num_points, const int32_t *num_tiles_hit, int32_t *cum_tiles_hit,
                          int32_t *total, gi2d_stream_t stream);

/* bindings.cu:283-365 map_gaussian_to_intersects_tensor.  isect_ids i64[M], gaussian_ids i32[M];
 * every element is written (zeros where the reference leaves its torch::zeros untouched);
 * writes beyond num_intersects are dropped instead of corrupting memory. */
int gi2d_map_gaussian_to_intersects(int num_points, int num_intersects, const float *xys,
                                    const float *depths, const int32_t *radii,
                                    const int32_t *cum_tiles_hit, int tiles_x, int tiles_y,
                                    float radius_clip, int64_t *isect_ids, int32_t *gaussian_ids,
                                    gi2d_stream_t stream);

/* gsplat/gsplat/utils.py:301-302 (torch.sort + torch.gather): STABLE sort of the pairs by key.
 * Keys must have a tile id (bits 63..32) in [0, num_tiles) and non-negative depth bits.
 * Optional outputs (may be NULL): isect_ids_sorted; perm i32[M] (sorted position -> input
 * position); inv_perm i32[M] (input position -> sorted position); tile_bins i32[num_tiles,2]
 * ([start,end) per tile, (0,0) when empty -- same content as gi2d_get_tile_bin_edges on the
 * first num_tiles rows).  workspace: device scratch of at least
 * gi2d_sort_workspace_bytes(M, num_tiles) bytes; after the call its first four int32 words hold
 * status flags {any non-zero depth bits, tile id out of range (pair dropped), tile longer than
 * 1024 entries with non-zero depth bits (unsupported, tile left unsorted), 0}. */
size_t gi2d_sort_workspace_bytes(int num_intersects, int num_tiles);
int gi2d_sort_intersects(int num_intersects, int num_tiles, const int64_t *isect_ids,
                         const int32_t *gaussian_ids, int64_t *isect_ids_sorted,
                         int32_t *gaussian_ids_sorted, int32_t *perm, int32_t *inv_perm,
                         int32_t *tile_bins, void *workspace, size_t workspace_bytes,
                         gi2d_stream_t stream);

/* bindings.cu:368-383 get_tile_bin_edges_tensor.  tile_bins i32[rows,2], indexed by tile id; the
 * reference allocates rows = num_intersects and writes out of bounds for larger tile ids --
 * here such writes are dropped. */
int gi2d_get_tile_bin_edges(int num_intersects, const int64_t *isect_ids_sorted, int rows,
                            int32_t *tile_bins, gi2d_stream_t stream);

/* ------------------------------------------------------------------ rasterizer (a9, a10)
 * bindings.cu:453-526 rasterize_forward_sum_tensor == :529-610 rasterize_sum_plus_forward_tensor.
 * out_img f32[H,W,3], final_Ts f32[H,W] (=1), final_idx i32[H,W].  tile_bins has
 * tile_bins_rows rows; tiles with id >= rows are empty.  `background` (device f32[3]) is only
 * used when num_intersects_dev != NULL and *num_intersects_dev < 1, reproducing
 * rasterize_sum_plus.py:110-118 (image = background) without a host round trip; pass NULL for
 * the plain kernel semantics. */
int gi2d_rasterize_sum_forward(int tiles_x, int tiles_y, unsigned img_width, unsigned img_height,
                               const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                               int tile_bins_rows, const float *xys, const float *conics,
                               const float *colors, const float *opacities,
                               const float *background, const int32_t *num_intersects_dev,
                               float *final_Ts, int32_t *final_idx, float *out_img,
                               gi2d_stream_t stream);
int gi2d_rasterize_sum_plus_forward(int tiles_x, int tiles_y, unsigned img_width,
                                    unsigned img_height, const int32_t *gaussian_ids_sorted,
                                    const int32_t *tile_bins, int tile_bins_rows,
                                    const float *xys, const float *conics, const float *colors,
                                    const float *opacities, const float *background,
                                    const int32_t *num_intersects_dev, float *final_Ts,
                                    int32_t *final_idx, float *out_img, gi2d_stream_t stream);

/* bindings.cu:1166-1240 rasterize_backward_sum_tensor == :1241-1314 rasterize_sum_plus_backward_tensor.
 * v_xy f32[N,2], v_conic f32[N,3], v_rgb f32[N,3], v_opacity f32[N]; every row is written.
 * v_abs_xy (sum form only, may be NULL) f32[N,4] = per-gaussian sums over pixels of
 * (v_x, v_y, |v_x|, |v_y|): the gradient rasterize_sum.py:308,328 hands back for
 * `screenspace_points` (backward.cu:932,959-960, commented out in the shipped kernel).
 * No float atomics: each (tile, gaussian) partial is stored once and summed per gaussian in a
 * fixed order, so gradients are bitwise reproducible.
 * Two ways to provide the gaussian-major index used by the final sum:
 *   - plan form: cum_tiles_hit i32[N] + inv_perm i32[M] (input position -> sorted position,
 *     as written by gi2d_sort_intersects) from the forward binning;
 *   - generic form: pass both NULL; the index is rebuilt from gaussian_ids_sorted.
 * workspace >= gi2d_rasterize_backward_workspace_bytes(N, M). */
size_t gi2d_rasterize_backward_workspace_bytes(int num_points, int num_intersects);
int gi2d_rasterize_sum_backward(int num_points, int num_intersects, unsigned img_height,
                                unsigned img_width, const int32_t *gaussian_ids_sorted,
                                const int32_t *tile_bins, int tile_bins_rows, const float *xys,
                                const float *conics, const float *colors, const float *opacities,
                                const int32_t *final_idx, const float *v_output,
                                const int32_t *cum_tiles_hit, const int32_t *inv_perm,
                                float *v_xy, float *v_conic, float *v_rgb, float *v_opacity,
                                float *v_abs_xy, void *workspace, size_t workspace_bytes,
                                gi2d_stream_t stream);
int gi2d_rasterize_sum_plus_backward(int num_points, int num_intersects, unsigned img_height,
                                     unsigned img_width, const int32_t *gaussian_ids_sorted,
                                     const int32_t *tile_bins, int tile_bins_rows,
                                     const float *xys, const float *conics, const float *colors,
                                     const float *opacities, const int32_t *final_idx,
                                     const float *v_output, const int32_t *cum_tiles_hit,
                                     const int32_t *inv_perm, float *v_xy, float *v_conic,
                                     float *v_rgb, float *v_opacity, void *workspace,
                                     size_t workspace_bytes, gi2d_stream_t stream);

/* ------------------------------------------------------------------ sync-free fast path
 * The same results as the ops above, with the host round trip of the reference orchestration
 * (rasterize_sum_plus.py:108 -> utils.py:249 `.item()`) removed: buffers are sized by a
 * caller-chosen `capacity` (>= the expected number of intersections) and the true count lives on
 * the device.  Every call only enqueues kernels, so project -> bin -> rasterize fwd/bwd can be
 * captured in one hipGraph.
 *
 * gi2d_bin_gaussians replaces compute_cumulative_intersects + map_gaussian_to_intersects +
 * torch.sort/gather + get_tile_bin_edges (utils.py:231-311) for the 2D path (depth == 0):
 *   gaussian_ids_sorted i32[capacity] : per tile, ascending ids of the gaussians with radii > 0,
 *                                       radii >= radius_clip whose tile box covers it
 *   tile_bins i32[tiles_x*tiles_y, 2] : [start, end) per tile, (0,0) when empty
 *   status i32[4]                     : {M = number of intersections, M > capacity (lists were
 *                                       truncated -- enlarge capacity and redo), 0, 0}
 * &status[0] is what gi2d_rasterize_sum_forward takes as num_intersects_dev. */
size_t gi2d_bin_workspace_bytes(int capacity, int num_tiles);
int gi2d_bin_gaussians(int num_points, int capacity, const float *xys, const int32_t *radii,
                       int tiles_x, int tiles_y, float radius_clip, int32_t *gaussian_ids_sorted,
                       int32_t *tile_bins, int32_t *status, void *workspace, size_t workspace_bytes,
                       gi2d_stream_t stream);

/* The two halves of rasterize_sum[_plus]_backward, callable on their own:
 *  _tiles  : the tile kernel; partials f32[capacity,12] receives one row per sorted position:
 *            (v_x, v_y, v_conic[3], v_rgb[3], v_opacity, sum|v_x|, sum|v_y|, 0); rows of positions
 *            past a tile's 256-entry cap are zero; the abs sums are computed iff with_abs != 0.
 *  _reduce : per-gaussian sum of those rows in ascending tile order.  The rows are located by
 *            re-deriving the gaussian's tile box from (xys, radii, radius_clip) as
 *            gi2d_bin_gaussians / map_gaussian_to_intersects do and binary-searching its id in
 *            each tile's ascending list, so no index from the forward pass is needed.
 *            v_abs_xy may be NULL. */
int gi2d_rasterize_backward_tiles(unsigned img_height, unsigned img_width,
                                  const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                                  int tile_bins_rows, const float *xys, const float *conics,
                                  const float *colors, const float *opacities,
                                  const int32_t *final_idx, const float *v_output, int with_abs,
                                  float *partials, gi2d_stream_t stream);
int gi2d_rasterize_backward_reduce(int num_points, const float *xys, const int32_t *radii,
                                   int tiles_x, int tiles_y, float radius_clip,
                                   const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                                   int tile_bins_rows, const float *partials, float *v_xy,
                                   float *v_conic, float *v_rgb, float *v_opacity, float *v_abs_xy,
                                   gi2d_stream_t stream);

/* ------------------------------------------------------------------ fused fast path
 * The whole per-iteration path in 3 launches (project+bin, forward+backward tile pass, reduce+project backward;
 * 4 when forward and backward tiles are issued separately; 2 per step in a loop, where the call that ends a step
 * also projects and bins for the next) on one caller-owned workspace.  Results equal the ops above (bit-identical
 * index work; the same per-pair arithmetic).  Contract:
 *   - workspace: gi2d_fast_workspace_bytes(N, tiles_x, tiles_y) bytes, emptied ONCE with gi2d_fast_workspace_init.
 *     It holds STATE between calls: persistent per-tile id lists, the tile box every gaussian was last binned with,
 *     and one 64-byte record per gaussian (centre, conic, colour, opacity, cull extents, tile box).  A binning call
 *     (gi2d_fast_bin, gi2d_fast_project_bin, the ..._project_bin form of the reduce call) appends a gaussian only to
 *     the tiles it has ENTERED since the previous binning call and refreshes its record; the tile pass drops the
 *     entries that left and keeps every list in ascending id order.  Results are those of a from-scratch binning
 *     for ANY change of the inputs between two calls; what must hold is
 *       * one binning call before every tile pass (the tile pass takes geometry, colour and opacity from the
 *         records of that call, not from the caller's arrays);
 *       * the same num_points in every call between two gi2d_fast_workspace_init (call it again when gaussians are
 *         renumbered, appended or dropped, and after an overflow);
 *       * one workspace per in-flight forward/backward pair (the backward reads what the forward left).
 *   - capacity: at most gi2d_fast_tile_capacity() (= 1024) candidate gaussians per tile, of which the 256 lowest ids
 *     are rasterized (forward.cu:553).  status i32[4] = {1 if any tile is non-empty else 0, 1 if a tile row
 *     overflowed (results invalid: use gi2d_bin_gaussians + the plain ops instead and re-initialise the workspace;
 *     the flag is raised by every tile pass until then), sticky copy of the overflow flag (never reset by the
 *     library: for loops that check once per many steps), 0}; [0], [1] are reset by the binning call and raised by
 *     the tile pass.  The intersection count itself is sum(num_tiles_hit).
 *   - kind: 0 Cholesky (project_gaussians_2d), 1 covariance, 2 scale-rot (p0 = scales, p1 = rot).
 *   - gi2d_fast_rasterize_forward: final_Ts may be NULL (it is the constant 1); `background`
 *     non-NULL adds the "no intersection at all -> image = background" rule of
 *     rasterize_sum_plus.py:110-118 (one extra tiny launch); NULL leaves such an image at 0.
 *   - final_idx (word positions in the workspace's list array, as gi2d_fast_workspace_views reports them) is
 *     optional in both directions: the fused backward does not need it (its forward evaluated every pair with the
 *     same instructions, so "idx <= final_idx" is implied by the alpha test); pass NULL to skip it.
 */
size_t gi2d_fast_workspace_bytes(int num_points, int tiles_x, int tiles_y);
int gi2d_fast_tile_capacity(void);
int gi2d_fast_workspace_init(void *workspace, size_t workspace_bytes, int num_points, int tiles_x,
                             int tiles_y, gi2d_stream_t stream);
/* tile_bins[t] = [start, end) word positions of tile t's ascending id list in gaussian_ids_sorted (valid after a
 * tile pass). */
int gi2d_fast_workspace_views(void *workspace, size_t workspace_bytes, int num_points, int tiles_x,
                              int tiles_y, int32_t **gaussian_ids_sorted, int32_t **tile_bins);
/* Binning step on given projection outputs (replaces compute_cumulative_intersects + bin_and_sort_gaussians,
 * gsplat/gsplat/utils.py:231-311, for depth == 0); conics / colors / opacities go into the records. */
int gi2d_fast_bin(int num_points, const float *xys, const int32_t *radii, const float *conics,
                  const float *colors, const float *opacities, int tiles_x, int tiles_y, float radius_clip,
                  void *workspace, size_t workspace_bytes, int32_t *status, gi2d_stream_t stream);
/* Projection (as gi2d_project_gaussians_2d*_forward) + binning step in one launch. */
int gi2d_fast_project_bin(int kind, int num_points, float clip_coe, const float *means2d,
                          const float *p0, const float *p1, const float *colors, const float *opacities,
                          unsigned img_height, unsigned img_width, int tiles_x, int tiles_y, float radius_clip,
                          float *xys, float *depths, int32_t *radii, float *conics, int32_t *num_tiles_hit,
                          void *workspace, size_t workspace_bytes, int32_t *status, gi2d_stream_t stream);
int gi2d_fast_rasterize_forward(int num_points, int tiles_x, int tiles_y, unsigned img_width,
                                unsigned img_height, const float *background, void *workspace,
                                size_t workspace_bytes, int32_t *status, float *final_Ts, int32_t *final_idx,
                                float *out_img, gi2d_stream_t stream);
/* Forward AND backward tiles in ONE launch (one workgroup per tile validates / gathers / stages its
 * gaussians once and runs both passes on them).  Exactly one of:
 *   v_output f32[H,W,3]  the gradient image is given (it cannot depend on this call's out_img);
 *   target   f32[H,W,3]  the gradient is the L2-loss gradient formed per pixel from the pixel the
 *                        forward has just produced: grad_scale * (clamp(out,0,1) - target), zero
 *                        where out is outside [0,1] (models/gaussianimage_cholesky.py:307-310,
 *                        loss_type "L2": grad_scale = 2/(3*H*W)); tile_sse f32[tiles] receives the
 *                        per-tile sum of squared errors (loss = sum(tile_sse)/(3*H*W)).
 * Leaves the workspace exactly as gi2d_fast_rasterize_forward + _backward_tiles(with_abs=0) would
 * for the reduce calls below, except that the packed record list is not written. */
int gi2d_fast_rasterize_forward_backward(int num_points, int tiles_x, int tiles_y, unsigned img_width,
                                         unsigned img_height, const float *background,
                                         const float *v_output, const float *target, float grad_scale,
                                         float *tile_sse, void *workspace, size_t workspace_bytes,
                                         int32_t *status, float *out_img, gi2d_stream_t stream);
/* Several images per launch.  The reference fits its images one after the other (train.py:294-308); BASELINE config 3
 * is a batch of 24.  One 768x512 image is 1536 tile workgroups -- exactly one residency round of the chip, all in the
 * same phase at the same time -- so K independent images in ONE launch (grid = sum of the images' tiles; a workgroup
 * looks its image up in a table in HBM) overlap each other's load latency and arithmetic.  Every image's results are
 * those of its own single-image call, bit for bit.
 *   images  host array of per-image arguments, meaning as in gi2d_fast_rasterize_forward_backward (one binning call per
 *           image before the pass, one workspace per image); exactly one of v_output / target per image, the same
 *           kind for the whole batch
 *   batch   device scratch of gi2d_batch_bytes(num_images) bytes (16-byte aligned): the table, rewritten by every call
 *           with stream-ordered kernels; num_images <= 64. */
typedef struct gi2d_fast_image {
    int num_points, tiles_x, tiles_y;
    unsigned img_width, img_height;
    float grad_scale;
    const float *v_output, *target;
    float *tile_sse;
    void *workspace;
    size_t workspace_bytes;
    int32_t *status;
    float *out_img;
} gi2d_fast_image;
size_t gi2d_batch_bytes(int num_images);
/* The tile pass of a batch takes one of two forms, with the same results bit for bit: one launch of the general
 * workgroup (256 staged candidates per tile; six workgroups per CU), or two launches -- a small workgroup (128 staged
 * candidates, eight per CU) on every tile it can serve and the general one on the rest -- which is 5 ... 10 % faster
 * while few tiles are that full (at most one in sixteen: the second launch runs at lower occupancy and is as long as
 * one tile's whole dependent chain as soon as it has a tile to serve) and slower otherwise.  The library picks per call
 * from what the PREVIOUS call on the same table reported
 * (the number of fuller tiles travels to pinned memory behind that call's kernels: no call ever waits for it); the
 * first call on a table, a call inside a stream capture and a batch of fewer than 12288 tiles (eight 768x512 images) take
 * the general form.  gi2d_train_steps does the same for a single image of more than 1536 tiles, keyed by its workspace.
 * Returns what the next call on `batch` (a batch table, or such an image's workspace) would take given what has
 * arrived so far: 1 two launches, 0 one.  Environment GI2D_BATCH_TILE_PASS = general | two-phase forces either (tests,
 * measurements). */
int gi2d_batch_tile_pass_form(const void *batch);
int gi2d_fast_rasterize_forward_backward_batched(int num_images, const gi2d_fast_image *images, void *batch,
                                                 size_t batch_bytes, gi2d_stream_t stream);
int gi2d_fast_rasterize_backward_tiles(int num_points, int tiles_x, int tiles_y, unsigned img_width,
                                       unsigned img_height, const int32_t *final_idx,
                                       const float *v_output, int with_abs, void *workspace,
                                       size_t workspace_bytes, gi2d_stream_t stream);
int gi2d_fast_rasterize_backward_reduce(int num_points, int tiles_x, int tiles_y, void *workspace,
                                        size_t workspace_bytes, float *v_xy, float *v_conic,
                                        float *v_rgb, float *v_opacity, float *v_abs_xy,
                                        gi2d_stream_t stream);
int gi2d_fast_reduce_project_backward(int kind, int num_points, const float *p0, const float *p1,
                                      unsigned img_height, unsigned img_width, const float *xys,
                                      const int32_t *radii, const float *conics, int tiles_x,
                                      int tiles_y, float radius_clip, void *workspace,
                                      size_t workspace_bytes, float *v_xy, float *v_conic,
                                      float *v_rgb, float *v_opacity, float *v_abs_xy,
                                      float *v_cov2d, float *v_mean2d, float *v_p0, float *v_p1,
                                      gi2d_stream_t stream);

/* The call that ends step i of a loop can also start step i+1: reduce + projection backward of the
 * gradients just produced, then projection + binning step (gi2d_fast_project_bin) of the same gaussians
 * from means2d / p0 / p1 / colors / opacities as they are NOW, in one launch -- xys / radii / conics /
 * num_tiles_hit are overwritten with the new projection after the backward has consumed the old one, and the
 * next gi2d_fast_rasterize_forward[_backward] finds its lists and records ready.  If the inputs change after this
 * call, simply bin again (gi2d_fast_project_bin) before the next tile pass. */
int gi2d_fast_reduce_project_backward_project_bin(
    int kind, int num_points, float clip_coe, const float *means2d, const float *p0, const float *p1,
    const float *colors, const float *opacities, unsigned img_height, unsigned img_width, float *xys,
    float *depths, int32_t *radii, float *conics, int32_t *num_tiles_hit, int tiles_x, int tiles_y,
    float radius_clip, void *workspace, size_t workspace_bytes, int32_t *status, float *v_xy, float *v_conic,
    float *v_rgb, float *v_opacity, float *v_abs_xy, float *v_cov2d, float *v_mean2d, float *v_p0, float *v_p1,
    gi2d_stream_t stream);

/* Kernel timer for measurement code: an armed timer attaches start/stop events to the NEXT
 * gi2d_fast_rasterize_forward_backward launch issued by the calling thread (hipExtLaunchKernelGGL), so
 * gi2d_timer_elapsed_us returns that kernel's own execution time -- the figure rocprofv3's kernel
 * trace reports -- rather than a span between stream markers.  elapsed_us waits for the kernel. */
int gi2d_timer_create(void **timer);
int gi2d_timer_destroy(void *timer);
int gi2d_timer_arm(void *timer);
/* The same for a loop inside ONE call (gi2d_train_steps, gi2d_train_steps_batched): of the tile-pass launches the
 * calling thread issues from now on, launch i (0, stride, 2 stride, ...) carries timers[i / stride], `count` timers in
 * all; the caller keeps the array alive until the last of them has been used.  count = 0 cancels. */
int gi2d_timer_arm_many(void **timers, int count, int stride);
int gi2d_timer_elapsed_us(void *timer, float *microseconds);

/* ------------------------------------------------------------------ fused fitting iteration
 * SURVEY 8f rank 2 (callers either side of the path): one whole training iteration of the
 * Cholesky (kind 0), covariance (kind 1) or scale-rot (kind 2) model with L2 loss and Adam --
 * (or Adan) -- models/gaussianimage_cholesky.py:302-317 / models/gaussianimage_covariance.py:249-259 --
 * in three launches: activations+projection+fill, one tile pass (rasterize forward, loss
 * gradient formed per pixel in registers, backward), gradient reduce + projection backward + activation
 * backward + torch.optim.Adam update.  All pointers are device pointers owned by the caller.
 *   xyz   f32[N,2]  raw positions (kind 0: pre-tanh; kinds 1, 2: pixels)  updated in place
 *   chol  f32[N,3]  raw Cholesky / covariance triple (bound is added), or, kind 2 (scale-rot model,
 *                   models/gaussianimage_rs.py:166-172): (_scaling.x, _scaling.y, _rotation) with
 *                   scale = |_scaling + bound[0:2]|, rotation = sigmoid(_rotation) * 2 pi   updated in place
 *   feat  f32[N,3]  colours                                              updated in place
 *   opacity f32[N] (not optimised), bound f32[3] (bound_stride 0) or f32[N,3] (bound_stride 3)
 *   m_*, v_*        Adam moments, same shapes as the parameters           updated in place
 *   gt    f32[H,W,3] target image; out_img f32[H,W,3] last render (pre-clamp)
 *   tile_sse f32[tiles] per-tile sum of squared errors of clamp(out_img) vs gt (for PSNR)
 *   xys/conics/radii/num_tiles_hit: projection outputs of the last render
 *   status i32[4], workspace: as for the fast path (gi2d_fast_workspace_bytes/_init)
 *   dbg_grads f32[N,8] or NULL: gradients w.r.t. (xyz, chol, feat) of the last step (tests)
 * gi2d_train_step(state, lr[3], beta1, beta2, eps, step): lr = learning rates of the
 * (xyz, chol, feat) groups for this step, step = 1-based Adam step count. */
typedef struct gi2d_train_state {
    int kind, num_points, img_height, img_width;
    float clip_coe, radius_clip;
    float *xyz, *chol, *feat;
    const float *opacity, *bound;
    int bound_stride, pad0;
    float *m_xyz, *v_xyz, *m_chol, *v_chol, *m_feat, *v_feat;
    const float *gt;
    float *xys, *conics;
    int32_t *radii, *num_tiles_hit;
    float *out_img, *tile_sse;
    int32_t *status;
    void *workspace;
    size_t workspace_bytes;
    float *dbg_grads;
    /* optional best-model snapshot kept on the device (all NULL = off): when the squared error of a step's
     * render is below best_sse, the parameters after that step's update are copied to best_* and best_info =
     * {num_points, step} (train.py:133-139 without the host round trip).  best_sse f32[2], initialised to
     * +inf by the caller, ping-pongs between steps: the current best is best_sse[(last step + 1) & 1]. */
    float *best_xyz, *best_chol, *best_feat, *best_bound, *best_sse;
    int32_t *best_info;
    /* optimizer: 0 = torch.optim.Adam (m_* = exp_avg, v_* = exp_avg_sq; beta1, beta2 of the step call);
     * 1 = Adan (optimizer.py:125-330, what train.py selects for the Cholesky / RS models): v_* = exp_avg_sq n_t
     * with beta3, d_* = exp_avg_diff, pg_* = previous gradient (the reference stores its negative), same shapes
     * as the parameters; weight decay 0, no gradient-norm clipping. */
    int optimizer;
    int pad1;
    double beta3; /* betas and learning rates are doubles, like the Python floats torch's optimizers take: 1 - beta and
                     lr / (1 - beta^t) are formed in double and only then rounded to the fp32 the kernels use */
    float *d_xyz, *d_chol, *d_feat, *pg_xyz, *pg_chol, *pg_feat;
    /* quantisation-aware iterations (NULL = off): see gi2d_train_quant below */
    const struct gi2d_train_quant *quant;
    /* device-resident population (NULL = off: num_points is exact): the number of live gaussians as a word in HBM.
     * When set, num_points is only an UPPER BOUND (launch sizes, workspace layout); every kernel works on the first
     * min(*num_points_dev, num_points) rows, and gi2d_train_prune / gi2d_train_grow change the count without the host
     * having to know it. */
    int32_t *num_points_dev;
    /* the tiles' inboxes (NULL = none): gi2d_train_inbox_bytes(tiles_x, tiles_y) bytes of scratch, uninitialised, that
     * gi2d_train_steps on ONE image uses between the iterations of a call to let a gaussian enter a tile without a
     * returning atomic (128 KB of sparsely touched address space per tile, 192 MiB at 768x512, images of
     * at most 1536 tiles).  Optional: without it the same call appends through the row headers, bit-identical results,
     * the update kernel ~0.7 us slower at N = 50 000.  Nothing else looks at it (batches, the quantised iterations,
     * gi2d_train_render); no call returns with anything left in it. */
    void *inbox;
    size_t inbox_bytes;
} gi2d_train_state;
/* Size of gi2d_train_state::inbox for that tile grid; 0 for a grid that does without (more than 1536 tiles). */
size_t gi2d_train_inbox_bytes(int tiles_x, int tiles_y);

/* Quantisation-aware fitting.  Covariance model (kind 1, Adam): GaussianImage_Covariance.
 * train_iter_quantize / forward_quantize (models/gaussianimage_covariance.py:219-247,384-410) after
 * train_quantize.py's warm-up -- positions through a 2-channel LSQ quantiser (xy_bits), covariance rows through
 * HybirdQuant (log quantiser on the variances, LSQ on the covariance, both cov_bits), colours through a 3-channel LSQ
 * quantiser (color_bits) ahead of the projection, all with straight-through rounding; the twelve learned quantiser
 * values are trained by their own Adam optimizers (lr/eps per quantiser: xy, covariance, colour).  An iteration is
 * four launches: quantise+project+fill, the tile pass, reduce + projection backward + quantiser backward + Adam on
 * the gaussians, and a one-workgroup kernel that closes the whole-array reductions (v_scale / v_beta of the LSQ
 * channels, the gradient that reaches the extremes of the log range, the range of the next iteration).
 *   qparams f32[12]  xy scale[2], xy beta[2], cov scale, cov beta, colour scale[3], colour beta[3]   updated in place
 *   qm, qv  f32[12]  their Adam moments
 *   range   f32[4]   scratch: log range of the variances (maintained by the calls)
 *   qfeat   f32[N,3] dequantised colours of the last render
 *   partial f32[(ceil(N/64) + 4) * 24] scratch (one 96-byte row per wave of the per-gaussian launches, 16-byte aligned);  defer i32[8 + 8*defer_capacity] scratch (32-byte aligned), zero-initialised by the
 *           caller: variances that tie with an extreme of the log range wait here for the global sums (more than
 *           defer_capacity of them in one step sets bit 1 of status[2])
 *   best_qparams f32[12] or NULL: snapshot of qparams taken with the best-model snapshot
 *   dbg_qgrads f32[16] or NULL: gradients of qparams (same order), then the log quantiser's v_scale, v_beta and the
 *           per-element range gradients at the minimum / maximum (tests)
 *   lr, eps (host) per quantiser optimizer; beta1, beta2; first_step = their 1-based Adam step of the call's first
 *           iteration.
 * Rotation-scale model (kind 2, Adam; GaussianImage_RS.forward_quantize / train_iter_quantize,
 * models/gaussianimage_rs.py:131-163,443-485 -- BASELINE config 5): four LSQ quantisers -- positions xy_bits unsigned,
 * the raw `_scaling` cov_bits unsigned, sigmoid(_rotation) * 2 pi rot_bits SIGNED, colours color_bits unsigned -- so
 *   qparams, qm, qv f32[16] (64-byte aligned): xy scale[2], xy beta[2], scaling scale[2], scaling beta[2], rotation
 *           scale, rotation beta, colour scale[3], colour beta[3]; best_qparams f32[16]; dbg_qgrads f32[16];
 *   lr / eps [3] = positions, scaling + rotation (one optimizer in the model file), colours;
 *   range and defer are unused (may be NULL); an iteration is four launches as above. */
typedef struct gi2d_train_quant {
    int xy_bits, cov_bits, color_bits, defer_capacity;
    float *qparams, *qm, *qv, *range, *qfeat, *partial;
    int32_t *defer;
    float *best_qparams, *dbg_qgrads;
    double lr[3];
    float eps[3];
    int pad1;
    double beta1, beta2;
    int first_step;
    int rot_bits; /* kind 2 only: bit depth of the SIGNED rotation quantiser (models/gaussianimage_rs.py:143) */
} gi2d_train_quant;
int gi2d_train_render(const gi2d_train_state *state, gi2d_stream_t stream);
int gi2d_train_step(const gi2d_train_state *state, const double *lr_host, double beta1, double beta2,
                    float eps, int step, gi2d_stream_t stream);
/* `count` iterations with constant learning rates in one call (Adam steps first_step ...): the
 * update kernel of every iteration but the last also activates, projects and bins the updated
 * gaussians for the next one, so the call issues 2*count + 1 launches instead of 3*count. */
int gi2d_train_steps(const gi2d_train_state *state, const double *lr_host, double beta1, double beta2,
                     float eps, int first_step, int count, gi2d_stream_t stream);
/* The same iterations for `num_images` independent images in lockstep, every kernel launched ONCE for the whole batch
 * (2*count + 1 launches plus the table writes, whatever num_images is): states[k] is image k's state (its own
 * parameters, optimizer moments, target, workspace, best-model snapshot, device-resident population); all images share
 * kind, optimizer, learning rates and step count; image sizes and populations may differ.  Results per image are those
 * of gi2d_train_steps on that image alone, bit for bit.  Quantisation-aware batches (every state with a `quant`, same bit
 * depths, learning rates, eps, betas and first_step) run train_iter_quantize for all images: four launches per iteration.
 * batch: device scratch of gi2d_batch_bytes(num_images) bytes, rewritten by every call; num_images <= 64. */
int gi2d_train_steps_batched(int num_images, const gi2d_train_state *const *states, void *batch, size_t batch_bytes,
                             const double *lr_host, double beta1, double beta2, float eps, int first_step, int count,
                             gi2d_stream_t stream);

/* Population changes of a fit on the device (covariance model; state->num_points_dev required):
 *   gi2d_train_prune  non_semi_definite_prune (models/gaussianimage_covariance.py:352-382): rows whose covariance +
 *                     bound is not positive definite are dropped, all per-gaussian arrays of the state (parameters,
 *                     optimizer moments, opacity, per-gaussian bound) compacted in order; *pruned_total (device,
 *                     optional) accumulates the number dropped.  Nothing moves when nothing is pruned.
 *   gi2d_train_grow   add_sample_positions + densification_postfix (train.py:85-118, :307-350 of the model): the
 *                     k = max(0, min(budget_cap, max_points - live count)) pixels of the last render (out_img) with the
 *                     largest summed absolute error, in (error descending, pixel index ascending) order, become
 *                     centres of new gaussians with covariance rand3[r] + (0.5, 0, 0.5), colour 0, opacity 1, zero
 *                     optimizer moments and the low-pass bound of the new population; non-definite draws are skipped.
 *                     rand3 f32[rand_rows,3]: uniform numbers (row r belongs to the r-th selected pixel); budget_cap =
 *                     1000, or max_points at the last growth step of a run (train.py:91-97); *added (device, optional)
 *                     accumulates the number appended.
 * Both need gi2d_densify_scratch_bytes(state, max_points) bytes of scratch and only enqueue kernels.  gi2d_train_prune
 * itself empties state->workspace's persistent tile lists when -- and only when -- rows were renumbered (decided on the
 * device: a check that drops nothing leaves the lists as they are); after a growth the caller raises its own upper
 * bound num_points by budget_cap (clamped to max_points) and re-initialises the fast workspace
 * (gi2d_fast_workspace_init: its layout depends on num_points). */
size_t gi2d_densify_scratch_bytes(const gi2d_train_state *state, int max_points);
int gi2d_train_prune(const gi2d_train_state *state, void *scratch, size_t scratch_bytes, int32_t *pruned_total,
                     gi2d_stream_t stream);
int gi2d_train_grow(const gi2d_train_state *state, int max_points, int budget_cap, const float *rand3, int rand_rows,
                    void *scratch, size_t scratch_bytes, int32_t *added, gi2d_stream_t stream);

/* ------------------------------------------------------------------ quantisation-aware front end
 * the quantisers train_quantize.py puts in front of the projection after its warm-up
 * (models/gaussianimage_covariance.py:384-410 forward_quantize, :412-443 compress_wo_ec), from
 * /root/reference/quantize.py:
 *   GI2D_QUANT_LSQ  UniformQuantizer (LSQ+) :39-156   code = clamp((x - beta)/scale, qmin, qmax), y = round(code)*scale + beta,
 *                                                      learned per-channel scale / beta, straight-through rounding (:23-24)
 *   GI2D_QUANT_LOG  LogQuantizer, learned=False :158-259   t = log(|x| + 1e-6), range [min t, max t] recomputed from the
 *                                                      data on every forward, y = exp(round(code)*scale + beta) (no sign)
 *   HybirdQuant :336-389 on [N,3] rows (a, b, c) is the spec {LOG, LSQ, LOG}; a plain LogQuantizer on [N,2] is {LOG, LOG}.
 * Rows are f32[N, channels], channels <= 4.  `params` is f32[channels][4] = {scale, beta, max t, 0} on the device:
 * read for LSQ channels; WRITTEN by gi2d_quant_forward for log channels (all log channels of a spec share one range
 * there: LogQuantizer.forward reduces over its whole input) and by gi2d_quant_init for every channel (per-channel
 * ranges: _init_data :69-77 / :192-201, which is also what LogQuantizer.compress uses).
 *   gi2d_quant_forward   -> dequant f32[N,C] and/or code f32[N,C] (rounded integer codes as floats; NULL = skip)
 *   gi2d_quant_backward  given v_dequant: v_x f32[N,C] and v_params f32[C][2] = {v_scale, v_beta} (0 for log channels).
 *                        Follows the autograd graph of the forward as written, including the gradient that reaches the
 *                        elements at the extremes of the log range through beta = min() and scale = (max() - min())/Q
 *                        (spread evenly over ties, as torch.min()/max() do).
 *   gi2d_quant_compress  forward with the per-channel ranges in `params` (no range pass): compress() :149-152, :243-255
 *   gi2d_quant_decompress code -> value: decompress() :154-156, :257-259
 *   gi2d_quant_half      FakeQuantizationHalf :27-37 (x.half().float()); its backward is the identity
 * workspace: gi2d_quant_workspace_bytes(N) bytes of scratch; sums are two-stage in a fixed order (bitwise reproducible). */
#define GI2D_QUANT_LSQ 0
#define GI2D_QUANT_LOG 1
#define GI2D_QUANT_MAX_CHANNELS 4
typedef struct gi2d_quant_spec {
    int32_t channels;
    int32_t kind[GI2D_QUANT_MAX_CHANNELS];
    float qmin[GI2D_QUANT_MAX_CHANNELS], qmax[GI2D_QUANT_MAX_CHANNELS];
} gi2d_quant_spec;
size_t gi2d_quant_workspace_bytes(int num_rows);
int gi2d_quant_init(const gi2d_quant_spec *spec_host, int num_rows, const float *x, float *params,
                    void *workspace, size_t workspace_bytes, gi2d_stream_t stream);
int gi2d_quant_forward(const gi2d_quant_spec *spec_host, int num_rows, const float *x, float *params,
                       float *dequant, float *code, void *workspace, size_t workspace_bytes,
                       gi2d_stream_t stream);
int gi2d_quant_backward(const gi2d_quant_spec *spec_host, int num_rows, const float *x, const float *params,
                        const float *v_dequant, float *v_x, float *v_params, void *workspace,
                        size_t workspace_bytes, gi2d_stream_t stream);
int gi2d_quant_compress(const gi2d_quant_spec *spec_host, int num_rows, const float *x, const float *params,
                        float *dequant, float *code, gi2d_stream_t stream);
int gi2d_quant_decompress(const gi2d_quant_spec *spec_host, int num_rows, const float *code,
                          const float *params, float *out, gi2d_stream_t stream);
int gi2d_quant_half(size_t count, const float *x, float *y, gi2d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GI2D_H */
