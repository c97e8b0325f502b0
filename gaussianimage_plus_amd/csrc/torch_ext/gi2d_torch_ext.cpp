// Compiled op table of the drop-in `gsplat` package: the pybind module the reference builds from
// gsplat/gsplat/cuda/csrc/ext.cpp:16-66 (C++ signatures bindings.h:16-19,93-96,115-157,201-216,323-471) -- libtorch
// tensors in, tuples of freshly allocated tensors out, CHECK_INPUT semantics (bindings.h:9-14) -- over the C ABI of
// include/gi2d.h.  Nothing is computed here: every op allocates its outputs with torch, takes the current HIP stream of
// the inputs' device and enqueues the kernels of libgi2d_hip.so (linked, found through $ORIGIN).  The ctypes table in
// gsplat/cuda/__init__.py stays as the fallback OF THIS BINDING where no C++ compiler is at hand; the kernels have none.
//
// Besides the reference's names the module exports the fused fast path the autograd wrappers run on
// (fast_workspace_bytes / fast_workspace_init / fast_forward / fast_backward): one call per direction instead of two
// ctypes calls with a dozen marshalled pointers each.
#include <torch/extension.h>

// ROCm builds of PyTorch present their HIP devices as device type "cuda": guard and stream come from the masquerading
// headers (the plain c10::hip guard refuses a "cuda" device)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <tuple>

#include "gi2d.h"

namespace {

using torch::Tensor;

#define GI2D_CHECK_INPUT(x)                                            \
    TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor");          \
    TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")
#define GI2D_CHECK_F32(x) \
    GI2D_CHECK_INPUT(x);  \
    TORCH_CHECK((x).scalar_type() == torch::kFloat32, #x " must be float32")
#define GI2D_CHECK_I32(x) \
    GI2D_CHECK_INPUT(x);  \
    TORCH_CHECK((x).scalar_type() == torch::kInt32, #x " must be int32")

inline gi2d_stream_t stream_of(const Tensor &t) {
    return (gi2d_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}
inline void check(int rc, const char *what) {
    TORCH_CHECK(rc == 0, what, " failed (status ", rc, "): ", gi2d_last_error_string());
}
inline Tensor f32(const Tensor &like, std::initializer_list<int64_t> shape) {
    return torch::empty(shape, like.options().dtype(torch::kFloat32));
}
inline Tensor i32(const Tensor &like, std::initializer_list<int64_t> shape) {
    return torch::empty(shape, like.options().dtype(torch::kInt32));
}
inline const float *fp(const Tensor &t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline const int32_t *ip(const Tensor &t) { return t.defined() ? t.data_ptr<int32_t>() : nullptr; }
typedef std::tuple<int, int, int> dim3_t;

// ------------------------------------------------------------------------------------------ projection
template <class F>
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> project_fwd(F call, int n, const Tensor &means2d) {
    Tensor xys = f32(means2d, {n, 2}), depths = f32(means2d, {n}), radii = i32(means2d, {n});
    Tensor conics = f32(means2d, {n, 3}), nth = i32(means2d, {n});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means2d.device());
    call(xys.data_ptr<float>(), depths.data_ptr<float>(), radii.data_ptr<int32_t>(), conics.data_ptr<float>(),
         nth.data_ptr<int32_t>(), stream_of(means2d));
    return std::make_tuple(xys, depths, radii, conics, nth);
}

// bindings.cu:1317-1381
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> project_gaussians_2d_forward(
    int num_points, float clip_coe, Tensor &means2d, Tensor &L_elements, unsigned img_height, unsigned img_width,
    dim3_t tile_bounds, float clip_thresh, float radius_clip, bool isprint) {
    GI2D_CHECK_F32(means2d);
    GI2D_CHECK_F32(L_elements);
    return project_fwd(
        [&](float *xys, float *depths, int32_t *radii, float *conics, int32_t *nth, gi2d_stream_t st) {
            check(gi2d_project_gaussians_2d_forward(num_points, clip_coe, fp(means2d), fp(L_elements), img_height,
                                                    img_width, std::get<0>(tile_bounds), std::get<1>(tile_bounds),
                                                    clip_thresh, radius_clip, xys, depths, radii, conics, nth, st),
                  "project_gaussians_2d_forward");
        },
        num_points, means2d);
}
// bindings.cu:1449-1513
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> project_gaussians_2d_covariance_forward(
    int num_points, float clip_coe, Tensor &means2d, Tensor &cov2d, unsigned img_height, unsigned img_width,
    dim3_t tile_bounds, float clip_thresh, float radius_clip, bool isprint) {
    GI2D_CHECK_F32(means2d);
    GI2D_CHECK_F32(cov2d);
    return project_fwd(
        [&](float *xys, float *depths, int32_t *radii, float *conics, int32_t *nth, gi2d_stream_t st) {
            check(gi2d_project_gaussians_2d_covariance_forward(
                      num_points, clip_coe, fp(means2d), fp(cov2d), img_height, img_width, std::get<0>(tile_bounds),
                      std::get<1>(tile_bounds), clip_thresh, radius_clip, xys, depths, radii, conics, nth, st),
                  "project_gaussians_2d_covariance_forward");
        },
        num_points, means2d);
}
// bindings.cu:1384-1448
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> project_gaussians_2d_scale_rot_forward(
    int num_points, float clip_coe, Tensor &means2d, Tensor &scales2d, Tensor &rotation, unsigned img_height,
    unsigned img_width, dim3_t tile_bounds, float clip_thresh, float radius_clip, bool isprint) {
    GI2D_CHECK_F32(means2d);
    GI2D_CHECK_F32(scales2d);
    GI2D_CHECK_F32(rotation);
    return project_fwd(
        [&](float *xys, float *depths, int32_t *radii, float *conics, int32_t *nth, gi2d_stream_t st) {
            check(gi2d_project_gaussians_2d_scale_rot_forward(
                      num_points, clip_coe, fp(means2d), fp(scales2d), fp(rotation), img_height, img_width,
                      std::get<0>(tile_bounds), std::get<1>(tile_bounds), clip_thresh, radius_clip, xys, depths, radii,
                      conics, nth, st),
                  "project_gaussians_2d_scale_rot_forward");
        },
        num_points, means2d);
}

// bindings.cu:1517-1564 -> (v_cov2d, v_mean2d, v_L_elements)
std::tuple<Tensor, Tensor, Tensor> project_gaussians_2d_backward(int num_points, Tensor &means2d, Tensor &L_elements,
                                                                 unsigned img_height, unsigned img_width,
                                                                 Tensor &radii, Tensor &conics, Tensor &v_xy,
                                                                 const c10::optional<Tensor> &v_depth,
                                                                 Tensor &v_conic) {
    GI2D_CHECK_F32(means2d);
    GI2D_CHECK_F32(L_elements);
    GI2D_CHECK_I32(radii);
    GI2D_CHECK_F32(conics);
    GI2D_CHECK_F32(v_xy);
    GI2D_CHECK_F32(v_conic);
    Tensor v_cov2d = f32(means2d, {num_points, 3}), v_mean2d = f32(means2d, {num_points, 2});
    Tensor v_L = f32(means2d, {num_points, 3});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means2d.device());
    check(gi2d_project_gaussians_2d_backward(num_points, fp(means2d), fp(L_elements), img_height, img_width, ip(radii),
                                             fp(conics), fp(v_xy), v_depth ? fp(*v_depth) : nullptr, fp(v_conic),
                                             v_cov2d.data_ptr<float>(), v_mean2d.data_ptr<float>(),
                                             v_L.data_ptr<float>(), stream_of(means2d)),
          "project_gaussians_2d_backward");
    return std::make_tuple(v_cov2d, v_mean2d, v_L);
}
// bindings.cu:1565-1612
std::tuple<Tensor, Tensor, Tensor> project_gaussians_2d_covariance_backward(
    int num_points, Tensor &means2d, Tensor &cov2d, unsigned img_height, unsigned img_width, Tensor &radii,
    Tensor &conics, Tensor &v_xy, const c10::optional<Tensor> &v_depth, Tensor &v_conic) {
    GI2D_CHECK_F32(means2d);
    GI2D_CHECK_F32(cov2d);
    GI2D_CHECK_I32(radii);
    GI2D_CHECK_F32(conics);
    GI2D_CHECK_F32(v_xy);
    GI2D_CHECK_F32(v_conic);
    Tensor v_cov2d = f32(means2d, {num_points, 3}), v_mean2d = f32(means2d, {num_points, 2});
    Tensor v_elem = f32(means2d, {num_points, 3});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means2d.device());
    check(gi2d_project_gaussians_2d_covariance_backward(
              num_points, fp(means2d), fp(cov2d), img_height, img_width, ip(radii), fp(conics), fp(v_xy),
              v_depth ? fp(*v_depth) : nullptr, fp(v_conic), v_cov2d.data_ptr<float>(), v_mean2d.data_ptr<float>(),
              v_elem.data_ptr<float>(), stream_of(means2d)),
          "project_gaussians_2d_covariance_backward");
    return std::make_tuple(v_cov2d, v_mean2d, v_elem);
}
// bindings.cu:1614-1668 -> (v_cov2d, v_mean2d, v_scale[N,2], v_rot[N,1])
std::tuple<Tensor, Tensor, Tensor, Tensor> project_gaussians_2d_scale_rot_backward(
    int num_points, Tensor &means2d, Tensor &scales2d, Tensor &rotation, unsigned img_height, unsigned img_width,
    Tensor &radii, Tensor &conics, Tensor &v_xy, const c10::optional<Tensor> &v_depth, Tensor &v_conic) {
    GI2D_CHECK_F32(means2d);
    GI2D_CHECK_F32(scales2d);
    GI2D_CHECK_F32(rotation);
    GI2D_CHECK_I32(radii);
    GI2D_CHECK_F32(conics);
    GI2D_CHECK_F32(v_xy);
    GI2D_CHECK_F32(v_conic);
    Tensor v_cov2d = f32(means2d, {num_points, 3}), v_mean2d = f32(means2d, {num_points, 2});
    Tensor v_scale = f32(means2d, {num_points, 2}), v_rot = f32(means2d, {num_points, 1});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means2d.device());
    check(gi2d_project_gaussians_2d_scale_rot_backward(
              num_points, fp(means2d), fp(scales2d), fp(rotation), img_height, img_width, ip(radii), fp(conics),
              fp(v_xy), v_depth ? fp(*v_depth) : nullptr, fp(v_conic), v_cov2d.data_ptr<float>(),
              v_mean2d.data_ptr<float>(), v_scale.data_ptr<float>(), v_rot.data_ptr<float>(), stream_of(means2d)),
          "project_gaussians_2d_scale_rot_backward");
    return std::make_tuple(v_cov2d, v_mean2d, v_scale, v_rot);
}

// bindings.cu:44-63 -> (conics[N,3], radii[N,1])
std::tuple<Tensor, Tensor> compute_cov2d_bounds(int num_pts, float clip_coe, Tensor &covs2d) {
    GI2D_CHECK_F32(covs2d);
    Tensor conics = f32(covs2d, {num_pts, 3}), radii = f32(covs2d, {num_pts, 1});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(covs2d.device());
    check(gi2d_compute_cov2d_bounds(num_pts, clip_coe, fp(covs2d), conics.data_ptr<float>(), radii.data_ptr<float>(),
                                    stream_of(covs2d)),
          "compute_cov2d_bounds");
    return std::make_tuple(conics, radii);
}

// ------------------------------------------------------------------------------------------ binning
// bindings.cu:283-365 -> (isect_ids i64[M], gaussian_ids i32[M])
std::tuple<Tensor, Tensor> map_gaussian_to_intersects(int num_points, int num_intersects, Tensor &xys, Tensor &depths,
                                                      Tensor &radii, Tensor &cum_tiles_hit, dim3_t tile_bounds,
                                                      float radius_clip, bool isprint) {
    GI2D_CHECK_F32(xys);
    GI2D_CHECK_F32(depths);
    GI2D_CHECK_I32(radii);
    GI2D_CHECK_I32(cum_tiles_hit);
    Tensor isect = torch::empty({num_intersects}, xys.options().dtype(torch::kInt64));
    Tensor gids = i32(xys, {num_intersects});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(xys.device());
    check(gi2d_map_gaussian_to_intersects(num_points, num_intersects, fp(xys), fp(depths), ip(radii), ip(cum_tiles_hit),
                                          std::get<0>(tile_bounds), std::get<1>(tile_bounds), radius_clip,
                                          isect.data_ptr<int64_t>(), gids.data_ptr<int32_t>(), stream_of(xys)),
          "map_gaussian_to_intersects");
    return std::make_tuple(isect, gids);
}
// bindings.cu:368-383 -> tile_bins i32[rows, 2] (rows = num_intersects, as in the reference, unless given)
Tensor get_tile_bin_edges(int num_intersects, Tensor &isect_ids_sorted, c10::optional<int> rows_opt) {
    GI2D_CHECK_INPUT(isect_ids_sorted);
    TORCH_CHECK(isect_ids_sorted.scalar_type() == torch::kInt64, "isect_ids_sorted must be int64");
    const int rows = rows_opt ? *rows_opt : num_intersects;
    Tensor bins = i32(isect_ids_sorted, {rows, 2});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(isect_ids_sorted.device());
    check(gi2d_get_tile_bin_edges(num_intersects, isect_ids_sorted.data_ptr<int64_t>(), rows, bins.data_ptr<int32_t>(),
                                  stream_of(isect_ids_sorted)),
          "get_tile_bin_edges");
    return bins;
}

// ------------------------------------------------------------------------------------------ rasterizer
void check_block(dim3_t block) {
    TORCH_CHECK(std::get<0>(block) == GI2D_TILE && std::get<1>(block) == GI2D_TILE,
                "only 16x16 tiles are supported (csrc/config.h BLOCK_X/BLOCK_Y)");
}
template <class F>
std::tuple<Tensor, Tensor, Tensor> raster_fwd(F entry, const char *what, dim3_t tile_bounds, dim3_t block,
                                              dim3_t img_size, Tensor &gids, Tensor &tile_bins, Tensor &xys,
                                              Tensor &conics, Tensor &colors, Tensor &opacities, Tensor &background,
                                              const c10::optional<Tensor> &num_intersects_dev) {
    GI2D_CHECK_I32(gids);
    GI2D_CHECK_I32(tile_bins);
    GI2D_CHECK_F32(xys);
    GI2D_CHECK_F32(conics);
    GI2D_CHECK_F32(colors);
    GI2D_CHECK_F32(opacities);
    GI2D_CHECK_F32(background);
    check_block(block);
    TORCH_CHECK(colors.dim() == 2 && colors.size(1) == 3, "colors must have dimensions (num_points, 3)");
    const int w = std::get<0>(img_size), h = std::get<1>(img_size);
    Tensor out_img = f32(xys, {h, w, 3}), final_Ts = f32(xys, {h, w}), final_idx = i32(xys, {h, w});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(xys.device());
    check(entry(std::get<0>(tile_bounds), std::get<1>(tile_bounds), (unsigned)w, (unsigned)h, ip(gids), ip(tile_bins),
                (int)tile_bins.size(0), fp(xys), fp(conics), fp(colors), fp(opacities), fp(background),
                num_intersects_dev ? ip(*num_intersects_dev) : nullptr, final_Ts.data_ptr<float>(),
                final_idx.data_ptr<int32_t>(), out_img.data_ptr<float>(), stream_of(xys)),
          what);
    return std::make_tuple(out_img, final_Ts, final_idx);
}
// bindings.cu:453-526 (+ the 4th result rasterize_sum.py:157 unpacks: cnt_gs_counts, allocated and never filled)
std::tuple<Tensor, Tensor, Tensor, Tensor> rasterize_sum_forward(dim3_t tile_bounds, dim3_t block, dim3_t img_size,
                                                                 Tensor &gids, Tensor &tile_bins, Tensor &xys,
                                                                 Tensor &conics, Tensor &colors, Tensor &opacities,
                                                                 Tensor &background, bool isprint,
                                                                 const c10::optional<Tensor> &num_intersects_dev) {
    auto r = raster_fwd(gi2d_rasterize_sum_forward, "rasterize_sum_forward", tile_bounds, block, img_size, gids,
                        tile_bins, xys, conics, colors, opacities, background, num_intersects_dev);
    return std::make_tuple(std::get<0>(r), std::get<1>(r), std::get<2>(r), torch::zeros_like(std::get<2>(r)));
}
// bindings.cu:529-610
std::tuple<Tensor, Tensor, Tensor> rasterize_sum_plus_forward(dim3_t tile_bounds, dim3_t block, dim3_t img_size,
                                                              Tensor &gids, Tensor &tile_bins, Tensor &xys,
                                                              Tensor &conics, Tensor &colors, Tensor &opacities,
                                                              Tensor &background, bool isprint,
                                                              const c10::optional<Tensor> &num_intersects_dev) {
    return raster_fwd(gi2d_rasterize_sum_plus_forward, "rasterize_sum_plus_forward", tile_bounds, block, img_size, gids,
                      tile_bins, xys, conics, colors, opacities, background, num_intersects_dev);
}

struct BwdOut {
    Tensor v_xy, v_conic, v_colors, v_opacity, v_abs;
};
BwdOut raster_bwd(bool with_abs, unsigned img_height, unsigned img_width, unsigned block_h, unsigned block_w,
                  Tensor &gids, Tensor &tile_bins, Tensor &xys, Tensor &conics, Tensor &colors, Tensor &opacities,
                  Tensor &final_idx, Tensor &v_output, const c10::optional<Tensor> &cum_tiles_hit,
                  const c10::optional<Tensor> &inv_perm) {
    GI2D_CHECK_F32(xys);
    GI2D_CHECK_F32(colors);
    TORCH_CHECK(xys.dim() == 2 && xys.size(1) == 2, "xys must have dimensions (num_points, 2)");  // bindings.cu:1193
    TORCH_CHECK(colors.dim() == 2 && colors.size(1) == 3, "colors must have 2 dimensions");        // bindings.cu:1197
    check_block(dim3_t((int)block_w, (int)block_h, 1));
    GI2D_CHECK_I32(gids);
    GI2D_CHECK_I32(tile_bins);
    GI2D_CHECK_I32(final_idx);
    GI2D_CHECK_F32(conics);
    GI2D_CHECK_F32(opacities);
    GI2D_CHECK_F32(v_output);
    const int n = (int)xys.size(0), m = (int)gids.numel();
    BwdOut o;
    o.v_xy = f32(xys, {n, 2}), o.v_conic = f32(xys, {n, 3}), o.v_colors = f32(xys, {n, 3});
    o.v_opacity = f32(xys, {n, 1});
    if (with_abs) o.v_abs = f32(xys, {n, 4});
    const size_t nbytes = gi2d_rasterize_backward_workspace_bytes(n, m);
    Tensor ws = torch::empty({(int64_t)(nbytes > 256 ? nbytes : 256)}, xys.options().dtype(torch::kUInt8));
    c10::hip::HIPGuardMasqueradingAsCUDA guard(xys.device());
    const int32_t *cum = cum_tiles_hit ? ip(*cum_tiles_hit) : nullptr, *inv = inv_perm ? ip(*inv_perm) : nullptr;
    if (with_abs)
        check(gi2d_rasterize_sum_backward(n, m, img_height, img_width, ip(gids), ip(tile_bins), (int)tile_bins.size(0),
                                          fp(xys), fp(conics), fp(colors), fp(opacities), ip(final_idx), fp(v_output),
                                          cum, inv, o.v_xy.data_ptr<float>(), o.v_conic.data_ptr<float>(),
                                          o.v_colors.data_ptr<float>(), o.v_opacity.data_ptr<float>(),
                                          o.v_abs.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(), stream_of(xys)),
              "rasterize_sum_backward");
    else
        check(gi2d_rasterize_sum_plus_backward(
                  n, m, img_height, img_width, ip(gids), ip(tile_bins), (int)tile_bins.size(0), fp(xys), fp(conics),
                  fp(colors), fp(opacities), ip(final_idx), fp(v_output), cum, inv, o.v_xy.data_ptr<float>(),
                  o.v_conic.data_ptr<float>(), o.v_colors.data_ptr<float>(), o.v_opacity.data_ptr<float>(),
                  ws.data_ptr(), (size_t)ws.numel(), stream_of(xys)),
              "rasterize_sum_plus_backward");
    return o;
}
// bindings.cu:1166-1240 (+ v_abs_xys, rasterize_sum.py:308) -> (v_xy, v_conic, v_colors, v_opacity[N,1], v_abs_xys[N,4])
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_sum_backward(
    unsigned img_height, unsigned img_width, unsigned block_h, unsigned block_w, Tensor &gids, Tensor &tile_bins,
    Tensor &xys, Tensor &conics, Tensor &colors, Tensor &opacities, const c10::optional<Tensor> &background,
    const c10::optional<Tensor> &final_Ts, Tensor &final_idx, Tensor &v_output,
    const c10::optional<Tensor> &v_output_alpha, const c10::optional<Tensor> &cum_tiles_hit,
    const c10::optional<Tensor> &inv_perm) {
    BwdOut o = raster_bwd(true, img_height, img_width, block_h, block_w, gids, tile_bins, xys, conics, colors,
                          opacities, final_idx, v_output, cum_tiles_hit, inv_perm);
    return std::make_tuple(o.v_xy, o.v_conic, o.v_colors, o.v_opacity, o.v_abs);
}
// bindings.cu:1241-1314 -> (v_xy, v_conic, v_colors, v_opacity[N,1])
std::tuple<Tensor, Tensor, Tensor, Tensor> rasterize_sum_plus_backward(
    unsigned img_height, unsigned img_width, unsigned block_h, unsigned block_w, Tensor &gids, Tensor &tile_bins,
    Tensor &xys, Tensor &conics, Tensor &colors, Tensor &opacities, const c10::optional<Tensor> &background,
    const c10::optional<Tensor> &final_Ts, Tensor &final_idx, Tensor &v_output,
    const c10::optional<Tensor> &v_output_alpha, const c10::optional<Tensor> &cum_tiles_hit,
    const c10::optional<Tensor> &inv_perm) {
    BwdOut o = raster_bwd(false, img_height, img_width, block_h, block_w, gids, tile_bins, xys, conics, colors,
                          opacities, final_idx, v_output, cum_tiles_hit, inv_perm);
    return std::make_tuple(o.v_xy, o.v_conic, o.v_colors, o.v_opacity);
}

// ------------------------------------------------------------------------------------------ fused fast path
int64_t fast_workspace_bytes(int num_points, int tiles_x, int tiles_y) {
    return (int64_t)gi2d_fast_workspace_bytes(num_points, tiles_x, tiles_y);
}
void fast_workspace_init(Tensor &ws, int num_points, int tiles_x, int tiles_y) {
    GI2D_CHECK_INPUT(ws);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(ws.device());
    check(gi2d_fast_workspace_init(ws.data_ptr(), (size_t)ws.numel(), num_points, tiles_x, tiles_y, stream_of(ws)),
          "fast_workspace_init");
}
// gi2d_fast_bin (binning step + records on the workspace's persistent lists) + gi2d_fast_rasterize_forward
// -> out_img[H,W,3]; status = {any intersection, overflow, sticky overflow, -}.  `background` (optional): the device
// itself writes the background image when not a single gaussian lands (rasterize_sum_plus.py:110-118).
Tensor fast_forward(Tensor &ws, Tensor &status, int num_points, int tiles_x, int tiles_y, Tensor &xys, Tensor &radii,
                    Tensor &conics, Tensor &colors, Tensor &opacities, unsigned img_height, unsigned img_width,
                    float radius_clip, const c10::optional<Tensor> &background) {
    GI2D_CHECK_F32(xys);
    GI2D_CHECK_I32(radii);
    GI2D_CHECK_F32(conics);
    GI2D_CHECK_F32(colors);
    GI2D_CHECK_F32(opacities);
    GI2D_CHECK_I32(status);
    if (background) {
        GI2D_CHECK_F32(*background);
        TORCH_CHECK(background->numel() >= 3 && background->device() == xys.device(), "background must hold 3 floats on the inputs' device");
    }
    // the kernels trust num_points / tiles against the buffers: say so here, where a mismatch can still raise
    GI2D_CHECK_INPUT(ws);
    TORCH_CHECK(num_points >= 0 && tiles_x >= 0 && tiles_y >= 0, "negative size");
    TORCH_CHECK(xys.dim() == 2 && xys.size(0) == num_points && xys.size(1) == 2, "xys must be [num_points, 2]");
    TORCH_CHECK(radii.numel() == num_points, "radii must hold num_points entries");
    TORCH_CHECK(conics.numel() == 3 * (int64_t)num_points, "conics must be [num_points, 3]");
    TORCH_CHECK(colors.numel() == 3 * (int64_t)num_points, "colors must be [num_points, 3]");
    TORCH_CHECK(opacities.numel() == num_points, "opacities must be [num_points, 1]");
    TORCH_CHECK(status.numel() >= 4, "status must hold 4 words");
    for (const Tensor *t : std::initializer_list<const Tensor *>{&radii, &conics, &colors, &opacities, &status, &ws})
        TORCH_CHECK(t->device() == xys.device(), "every tensor of a fast_forward call must live on xys' device");
    TORCH_CHECK((int64_t)tiles_x * 16 >= (int64_t)img_width && (int64_t)tiles_y * 16 >= (int64_t)img_height,
                "tile grid does not cover the image");
    Tensor out_img = f32(xys, {(int64_t)img_height, (int64_t)img_width, 3});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(xys.device());
    gi2d_stream_t st = stream_of(xys);
    check(gi2d_fast_bin(num_points, fp(xys), ip(radii), fp(conics), fp(colors), fp(opacities), tiles_x, tiles_y,
                        radius_clip, ws.data_ptr(), (size_t)ws.numel(), status.data_ptr<int32_t>(), st),
          "fast_bin");
    check(gi2d_fast_rasterize_forward(num_points, tiles_x, tiles_y, img_width, img_height,
                                      background ? fp(*background) : nullptr, ws.data_ptr(), (size_t)ws.numel(),
                                      status.data_ptr<int32_t>(), nullptr, nullptr, out_img.data_ptr<float>(), st),
          "fast_rasterize_forward");
    return out_img;
}
// gi2d_fast_rasterize_backward_tiles + _reduce on the workspace the forward filled
// -> (v_xy, v_conic, v_colors, v_opacity[N,1], v_abs_xys[N,4] | undefined)
std::tuple<Tensor, Tensor, Tensor, Tensor, c10::optional<Tensor>> fast_backward(Tensor &ws, int num_points, int tiles_x,
                                                                                int tiles_y, Tensor &v_output,
                                                                                unsigned img_height,
                                                                                unsigned img_width, bool with_abs) {
    GI2D_CHECK_F32(v_output);
    GI2D_CHECK_INPUT(ws);
    TORCH_CHECK(num_points >= 0 && tiles_x >= 0 && tiles_y >= 0, "negative size");
    TORCH_CHECK(ws.device() == v_output.device(), "workspace and v_output must live on the same device");
    TORCH_CHECK(v_output.numel() == (int64_t)img_height * img_width * 3, "v_output must be [H, W, 3]");
    TORCH_CHECK((int64_t)tiles_x * 16 >= (int64_t)img_width && (int64_t)tiles_y * 16 >= (int64_t)img_height,
                "tile grid does not cover the image");
    const int n = num_points;
    Tensor v_xy = f32(v_output, {n, 2}), v_conic = f32(v_output, {n, 3}), v_colors = f32(v_output, {n, 3});
    Tensor v_opacity = f32(v_output, {n, 1});
    c10::optional<Tensor> v_abs;
    if (with_abs) v_abs = f32(v_output, {n, 4});
    c10::hip::HIPGuardMasqueradingAsCUDA guard(v_output.device());
    gi2d_stream_t st = stream_of(v_output);
    check(gi2d_fast_rasterize_backward_tiles(n, tiles_x, tiles_y, img_width, img_height, nullptr, fp(v_output),
                                             with_abs ? 1 : 0, ws.data_ptr(), (size_t)ws.numel(), st),
          "fast_rasterize_backward_tiles");
    check(gi2d_fast_rasterize_backward_reduce(n, tiles_x, tiles_y, ws.data_ptr(), (size_t)ws.numel(),
                                              v_xy.data_ptr<float>(), v_conic.data_ptr<float>(),
                                              v_colors.data_ptr<float>(), v_opacity.data_ptr<float>(),
                                              with_abs ? v_abs->data_ptr<float>() : nullptr, st),
          "fast_rasterize_backward_reduce");
    return std::make_tuple(v_xy, v_conic, v_colors, v_opacity, v_abs);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    namespace py = pybind11;
    m.doc() = "gfx950 op table of the drop-in gsplat package (names of gsplat/cuda/csrc/ext.cpp:16-66)";
    m.def("version", [] { return std::string(gi2d_version()); });
    // the 2D path of ext.cpp:16-66
    m.def("project_gaussians_2d_forward", &project_gaussians_2d_forward, py::arg("num_points"), py::arg("clip_coe"),
          py::arg("means2d"), py::arg("L_elements"), py::arg("img_height"), py::arg("img_width"), py::arg("tile_bounds"),
          py::arg("clip_thresh"), py::arg("radius_clip"), py::arg("isprint") = false);
    m.def("project_gaussians_2d_backward", &project_gaussians_2d_backward);
    m.def("project_gaussians_2d_covariance_forward", &project_gaussians_2d_covariance_forward, py::arg("num_points"),
          py::arg("clip_coe"), py::arg("means2d"), py::arg("L_elements"), py::arg("img_height"), py::arg("img_width"),
          py::arg("tile_bounds"), py::arg("clip_thresh"), py::arg("radius_clip"), py::arg("isprint") = false);
    m.def("project_gaussians_2d_covariance_backward", &project_gaussians_2d_covariance_backward);
    m.def("project_gaussians_2d_scale_rot_forward", &project_gaussians_2d_scale_rot_forward, py::arg("num_points"),
          py::arg("clip_coe"), py::arg("means2d"), py::arg("scales2d"), py::arg("rotation"), py::arg("img_height"),
          py::arg("img_width"), py::arg("tile_bounds"), py::arg("clip_thresh"), py::arg("radius_clip"),
          py::arg("isprint") = false);
    m.def("project_gaussians_2d_scale_rot_backward", &project_gaussians_2d_scale_rot_backward);
    m.def("compute_cov2d_bounds", &compute_cov2d_bounds);
    m.def("compute_cov2d_bounds_xy", &compute_cov2d_bounds);  // ext.cpp:55 binds both names to the same function
    m.def("map_gaussian_to_intersects", &map_gaussian_to_intersects, py::arg("num_points"), py::arg("num_intersects"),
          py::arg("xys"), py::arg("depths"), py::arg("radii"), py::arg("cum_tiles_hit"), py::arg("tile_bounds"),
          py::arg("radius_clip") = 1.0f, py::arg("isprint") = false);
    m.def("get_tile_bin_edges", &get_tile_bin_edges, py::arg("num_intersects"), py::arg("isect_ids_sorted"),
          py::arg("rows") = py::none());
    m.def("rasterize_sum_forward", &rasterize_sum_forward, py::arg("tile_bounds"), py::arg("block"), py::arg("img_size"),
          py::arg("gaussian_ids_sorted"), py::arg("tile_bins"), py::arg("xys"), py::arg("conics"), py::arg("colors"),
          py::arg("opacities"), py::arg("background"), py::arg("isprint") = false,
          py::arg("num_intersects_dev") = py::none());
    m.def("rasterize_sum_plus_forward", &rasterize_sum_plus_forward, py::arg("tile_bounds"), py::arg("block"),
          py::arg("img_size"), py::arg("gaussian_ids_sorted"), py::arg("tile_bins"), py::arg("xys"), py::arg("conics"),
          py::arg("colors"), py::arg("opacities"), py::arg("background"), py::arg("isprint") = false,
          py::arg("num_intersects_dev") = py::none());
    m.def("rasterize_sum_backward", &rasterize_sum_backward, py::arg("img_height"), py::arg("img_width"),
          py::arg("BLOCK_H"), py::arg("BLOCK_W"), py::arg("gaussian_ids_sorted"), py::arg("tile_bins"), py::arg("xys"),
          py::arg("conics"), py::arg("colors"), py::arg("opacities"), py::arg("background"), py::arg("final_Ts"),
          py::arg("final_idx"), py::arg("v_output"), py::arg("v_output_alpha") = py::none(),
          py::arg("cum_tiles_hit") = py::none(), py::arg("inv_perm") = py::none());
    m.def("rasterize_sum_plus_backward", &rasterize_sum_plus_backward, py::arg("img_height"), py::arg("img_width"),
          py::arg("BLOCK_H"), py::arg("BLOCK_W"), py::arg("gaussian_ids_sorted"), py::arg("tile_bins"), py::arg("xys"),
          py::arg("conics"), py::arg("colors"), py::arg("opacities"), py::arg("background"), py::arg("final_Ts"),
          py::arg("final_idx"), py::arg("v_output"), py::arg("v_output_alpha") = py::none(),
          py::arg("cum_tiles_hit") = py::none(), py::arg("inv_perm") = py::none());
    // the fused fast path the autograd wrappers run on
    m.def("fast_workspace_bytes", &fast_workspace_bytes);
    m.def("fast_workspace_init", &fast_workspace_init);
    m.def("fast_forward", &fast_forward, py::arg("ws"), py::arg("status"), py::arg("num_points"), py::arg("tiles_x"),
          py::arg("tiles_y"), py::arg("xys"), py::arg("radii"), py::arg("conics"), py::arg("colors"),
          py::arg("opacities"), py::arg("img_height"), py::arg("img_width"), py::arg("radius_clip"),
          py::arg("background") = py::none());
    m.def("fast_backward", &fast_backward);
}
