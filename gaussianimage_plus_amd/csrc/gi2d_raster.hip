// Additive ("sum") tile rasterizer, forward and backward, for gfx950.
//
// Reference: forward.cu:452-567 (== :570-691) and backward.cu:813-991 (== :1168-1350): one
// 16x16 CUDA block per tile, one thread per pixel, a 32-lane shuffle reduction and 9 float
// atomics per (gaussian, warp).  The CDNA4 design differs on purpose:
//
//  forward  : one workgroup (4 waves) per tile, one lane per pixel, wave w owns pixel rows
//             4w..4w+3.  Each wave builds ITS OWN compacted gaussian list in LDS: it walks the
//             tile list 64 entries at a time (one gaussian per lane), drops gaussians whose
//             conservative alpha>=1/255 box misses its 4-row strip, and ballot-compacts the
//             survivors (pre-scaled conic, opacity, colour) into wave-private LDS arrays.  No
//             workgroup barrier; the pixel loop then streams broadcast LDS reads.  The tile's RGB
//             is transposed through LDS and leaves as 16-byte stores of whole 192-byte rows.
//  backward : work items are (gaussian, pixel row) pairs restricted to the rows the gaussian
//             can reach; one lane per item walks the 16 pixels of its row with the row-constant
//             part of the quadratic form hoisted, accumulating 7 sums in registers -- no
//             cross-lane reduction at all.  Row partials go through LDS to the lane that owns
//             the gaussian, which adds them in row order and stores ONE 48-byte partial per
//             (tile, gaussian).  A second kernel sums each gaussian's partials in tile order.
//             No float atomics anywhere: gradients are bitwise reproducible, and the 72 atomics
//             per intersection of the reference become one coalesced 48-byte store.
//
// Roofline: algorithmic HBM bytes are 40*M + 20*H*W (fwd) and 40*M + 16*H*W + 36*N (bwd)
// (SURVEY 8d); every staged gaussian is reused by up to 256 pixels, so the kernels are bound by
// VALU issue (v_exp_f32 + ~25 fp32 ops per pixel-gaussian pair), not by HBM.
#include "gi2d_raster_core.h"

namespace gi2d {

// ------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void raster_fwd_kernel(
    int tiles_x, int tiles_y, int img_w, int img_h, const int32_t *__restrict__ gids_sorted,
    const int2 *__restrict__ tile_bins, int tile_bins_rows, const float2 *__restrict__ xys,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacities, const float *__restrict__ background,
    const int32_t *__restrict__ num_intersects_dev, float *__restrict__ final_Ts,
    int32_t *__restrict__ final_idx, float *__restrict__ out_img) {
    __shared__ FwdLds sm;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x;

    int2 range = make_int2(0, 0);
    if (tile < tile_bins_rows) range = tile_bins[tile];
    int len = range.y - range.x;
    len = len < 0 ? 0 : (len > GI2D_TILE_LIST_CAP ? GI2D_TILE_LIST_CAP : len);  // forward.cu:553

    // phase 1: the workgroup stages the tile list once, one gaussian per lane (a single round of
    // dependent global loads), and records which 4-row strips each gaussian can reach
    if (tid < len) {
        const GaussRec r = load_gaussian(gids_sorted[range.x + tid], xys, conics, colors, opacities);
        const AlphaRule ar = alpha_rule(r.gx, r.gy, r.a, r.b, r.c, r.opac);
        fwd_stage_entry(sm, tid, r, cull_word(r, (float)(tx * GI2D_TILE), (float)(ty * GI2D_TILE), img_h, ar.clamp), ar.lim);
    }
    if (tid == 0) fwd_stage_dummy(sm);
    __syncthreads();
    const bool bg = (num_intersects_dev != nullptr) && (*num_intersects_dev < 1);
    fwd_rasterize_staged(sm, len, range.x, tx, ty, img_w, img_h, bg, background, final_Ts, final_idx, out_img);
}

// ----------------------------------------------------------------------------------- backward
template <bool WITH_ABS>
__global__ __launch_bounds__(256, WITH_ABS ? 4 : GI2D_BWD_OCC) void raster_bwd_kernel(
    int tiles_x, int tiles_y, int img_w, int img_h, const int32_t *__restrict__ gids_sorted,
    const int2 *__restrict__ tile_bins, int tile_bins_rows, const float2 *__restrict__ xys,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacities, const int32_t *__restrict__ final_idx,
    const float *__restrict__ v_output, float4 *__restrict__ partials) {
    __shared__ BwdLds<WITH_ABS> sm;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x;

    int2 range = make_int2(0, 0);
    if (tile < tile_bins_rows) range = tile_bins[tile];
    const int full_len = range.y - range.x;
    if (full_len <= 0) return;
    const int len = full_len > GI2D_TILE_LIST_CAP ? GI2D_TILE_LIST_CAP : full_len;
    // entries beyond the forward cap never contributed (idx > final_idx everywhere)
    for (int k = GI2D_TILE_LIST_CAP + tid; k < full_len; k += 256) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        partials[3 * (size_t)(range.x + k) + 0] = z;
        partials[3 * (size_t)(range.x + k) + 1] = z;
        partials[3 * (size_t)(range.x + k) + 2] = z;
    }
    bwd_stage_pixels(sm, tx, ty, img_w, img_h, final_idx, v_output);
    const float ty0 = (float)(ty * GI2D_TILE), tx0 = (float)(tx * GI2D_TILE);
    unsigned mask = 0;
    if (tid < len) {
        const GaussRec r = load_gaussian(gids_sorted[range.x + tid], xys, conics, colors, opacities);
        const AlphaRule ar = alpha_rule(r.gx, r.gy, r.a, r.b, r.c, r.opac);
        bwd_stage_entry(sm, tid, r, ar.lim);
        mask = cull_word(r, tx0, ty0, img_h, ar.clamp);
    }
    bwd_run_tile<WITH_ABS>(sm, len, mask, range.x, tx0, ty0,
                           tid < len ? partials + 3 * (size_t)(range.x + tid) : nullptr);
}

// Plan form: gaussian g owns input positions [cum[g-1], cum[g]) (ascending tile id);
// inv_perm maps them to sorted positions.  Fixed order -> reproducible sums.
__global__ __launch_bounds__(256) void gather_plan_kernel(int n, int m,
                                                          const int32_t *__restrict__ cum,
                                                          const int32_t *__restrict__ inv_perm,
                                                          const float4 *__restrict__ partials,
                                                          float2 *__restrict__ v_xy,
                                                          float *__restrict__ v_conic,
                                                          float *__restrict__ v_rgb,
                                                          float *__restrict__ v_opacity,
                                                          float4 *__restrict__ v_abs_xy) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float acc[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) acc[q] = 0.f;
    const int s0 = g == 0 ? 0 : cum[g - 1];
    const int s1 = min(cum[g], m);
    for (int s = max(s0, 0); s < s1; ++s) {
        const int pos = inv_perm[s];
        if (pos >= 0 && pos < m) add_partial(acc, partials, pos);
    }
    store_grads(g, acc, v_xy, v_conic, v_rgb, v_opacity, v_abs_xy);
}

// Box form (fast path, pairs with gi2d_bin_gaussians): gaussian g re-derives the tiles it was binned
// into from (xys, radii) exactly as the binning did, and finds its entry in each tile's ascending id
// list by binary search.  Tiles are visited in ascending tile id -> reproducible sums.  Gaussians that
// cover more than GI2D_BIG_TILES tiles are handled by the whole wave (lanes stride over the tiles, fixed
// butterfly reduction) so one huge gaussian does not serialise a lane.
#define GI2D_BIG_TILES 32
__global__ __launch_bounds__(256) void gather_bbox_kernel(
    int n, const float2 *__restrict__ xys, const int32_t *__restrict__ radii, int tiles_x, int tiles_y,
    float radius_clip, const int32_t *__restrict__ gids_sorted, const int2 *__restrict__ tile_bins, int rows,
    const float4 *__restrict__ partials, float2 *__restrict__ v_xy, float *__restrict__ v_conic,
    float *__restrict__ v_rgb, float *__restrict__ v_opacity, float4 *__restrict__ v_abs_xy) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float acc[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) acc[q] = 0.f;
    int mnx = 0, mny = 0, mxx = 0, mxy = 0;
    bool mapped = false;
    if (g < n) {
        const int rad = radii[g];
        if (rad > 0 && !((float)rad < radius_clip)) {
            const float2 c = xys[g];
            tile_bbox(c.x, c.y, (float)rad, tiles_x, tiles_y, mnx, mny, mxx, mxy);
            mapped = mxx > mnx && mxy > mny;
        }
    }
    const int ntiles = mapped ? (mxx - mnx) * (mxy - mny) : 0;
    if (mapped && ntiles <= GI2D_BIG_TILES) {
        for (int i = mny; i < mxy; ++i)
            for (int j = mnx; j < mxx; ++j) {
                const int pos = find_in_tile(gids_sorted, tile_bins, i * tiles_x + j, rows, g);
                if (pos >= 0) add_partial(acc, partials, pos);
            }
    }
    unsigned long long big = __ballot(mapped && ntiles > GI2D_BIG_TILES);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bx0 = __shfl(mnx, src, 64), by0 = __shfl(mny, src, 64);
        const int bw = __shfl(mxx, src, 64) - bx0, bn = __shfl(ntiles, src, 64);
        const int bg = __shfl(g, src, 64);
        float part[11];
#pragma unroll
        for (int q = 0; q < 11; ++q) part[q] = 0.f;
        for (int t = lane; t < bn; t += 64) {
            const int tile = (by0 + t / bw) * tiles_x + bx0 + t % bw;
            const int pos = find_in_tile(gids_sorted, tile_bins, tile, rows, bg);
            if (pos >= 0) add_partial(part, partials, pos);
        }
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            float v = part[q];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == src) acc[q] = v;
        }
    }
    if (g < n) store_grads(g, acc, v_xy, v_conic, v_rgb, v_opacity, v_abs_xy);
}

// Generic form: rebuild the gaussian-major index from gaussian_ids_sorted.
__global__ __launch_bounds__(256) void gidx_count_kernel(int m, int n,
                                                         const int32_t *__restrict__ gids_sorted,
                                                         int32_t *__restrict__ counts) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int g = gids_sorted[p];
    if (g >= 0 && g < n) atomicAdd(&counts[g], 1);
}
__global__ __launch_bounds__(256) void gidx_scatter_kernel(int m, int n,
                                                           const int32_t *__restrict__ gids_sorted,
                                                           const int32_t *__restrict__ start,
                                                           int32_t *__restrict__ cursor,
                                                           int32_t *__restrict__ gslots) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int g = gids_sorted[p];
    if (g < 0 || g >= n) return;
    gslots[start[g] + atomicAdd(&cursor[g], 1)] = p;
}
#define GI2D_ORDERED_MAX 64 /* gaussians on <= 64 tiles are summed in ascending position order */
__global__ __launch_bounds__(256) void gather_generic_kernel(int n, int m,
                                                             const int32_t *__restrict__ start,
                                                             const int32_t *__restrict__ gslots,
                                                             const float4 *__restrict__ partials,
                                                             float2 *__restrict__ v_xy,
                                                             float *__restrict__ v_conic,
                                                             float *__restrict__ v_rgb,
                                                             float *__restrict__ v_opacity,
                                                             float4 *__restrict__ v_abs_xy) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float acc[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) acc[q] = 0.f;
    const int s0 = start[g], s1 = start[g + 1];
    const int cnt = s1 - s0;
    if (cnt <= GI2D_ORDERED_MAX) {
        int last = -1;
        for (int r = 0; r < cnt; ++r) {  // selection by ascending sorted position (cnt is small)
            int best = 0x7fffffff;
            for (int s = s0; s < s1; ++s) {
                const int p = gslots[s];
                best = (p > last && p < best) ? p : best;
            }
            add_partial(acc, partials, best);
            last = best;
        }
    } else {
        for (int s = s0; s < s1; ++s) add_partial(acc, partials, gslots[s]);
    }
    store_grads(g, acc, v_xy, v_conic, v_rgb, v_opacity, v_abs_xy);
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
struct BwdWs {
    float4 *partials;
    int32_t *counts, *start, *cursor, *gslots;
    size_t bytes;
};
static BwdWs carve_bwd_ws(void *base, int n, int m) {
    BwdWs w;
    char *b = (char *)base;
    size_t off = 0;
    const size_t nn = (size_t)(n > 0 ? n : 1), mm = (size_t)(m > 0 ? m : 1);
    w.partials = (float4 *)(b + off);
    off += align_up(mm * 3 * sizeof(float4));
    w.counts = (int32_t *)(b + off);
    off += align_up(nn * sizeof(int32_t));
    w.start = (int32_t *)(b + off);
    off += align_up((nn + 1) * sizeof(int32_t));
    w.cursor = (int32_t *)(b + off);
    off += align_up(nn * sizeof(int32_t));
    w.gslots = (int32_t *)(b + off);
    off += align_up(mm * sizeof(int32_t));
    w.bytes = off;
    return w;
}

int launch_exclusive_scan_with_cursor(int n, const int32_t *counts, int32_t *start, int32_t *cursor,
                                      hipStream_t st);  // gi2d_binning.hip

static int raster_forward(int tiles_x, int tiles_y, unsigned w, unsigned h, const int32_t *gids,
                          const int32_t *bins, int rows, const float *xys, const float *conics,
                          const float *colors, const float *opac, const float *background,
                          const int32_t *m_dev, float *final_Ts, int32_t *final_idx, float *out_img,
                          gi2d_stream_t st) {
    if (tiles_x < 0 || tiles_y < 0 || rows < 0) {
        set_error("rasterize forward: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if ((unsigned)tiles_x * GI2D_TILE < w || (unsigned)tiles_y * GI2D_TILE < h) {
        set_error("rasterize forward: tile grid does not cover the image");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    const long long t = (long long)tiles_x * tiles_y;
    if (t == 0 || w == 0 || h == 0) return GI2D_OK;
    if (!final_Ts || !final_idx || !out_img || (rows > 0 && !bins) || (m_dev && !background)) {
        set_error("rasterize forward: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(raster_fwd_kernel, dim3((unsigned)t), dim3(256), 0, (hipStream_t)st, tiles_x,
                       tiles_y, (int)w, (int)h, gids, (const int2 *)bins, rows, (const float2 *)xys,
                       conics, colors, opac, background, m_dev, final_Ts, final_idx, out_img);
    return check_launch("rasterize forward");
}

static int raster_backward(int n, int m, unsigned h, unsigned w, const int32_t *gids,
                           const int32_t *bins, int rows, const float *xys, const float *conics,
                           const float *colors, const float *opac, const int32_t *final_idx,
                           const float *v_output, const int32_t *cum, const int32_t *inv_perm,
                           float *v_xy, float *v_conic, float *v_rgb, float *v_opacity,
                           float *v_abs_xy, void *ws, size_t ws_bytes, gi2d_stream_t st_) {
    hipStream_t st = (hipStream_t)st_;
    if (n < 0 || m < 0 || rows < 0) {
        set_error("rasterize backward: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return GI2D_OK;
    if (!v_xy || !v_conic || !v_rgb || !v_opacity) {
        set_error("rasterize backward: null output");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if ((cum == nullptr) != (inv_perm == nullptr)) {
        set_error("rasterize backward: cum_tiles_hit and inv_perm must be given together");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (!ws || ws_bytes < gi2d_rasterize_backward_workspace_bytes(n, m)) {
        set_error("rasterize backward: workspace too small");
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    const int tiles_x = (int)((w + GI2D_TILE - 1) / GI2D_TILE), tiles_y = (int)((h + GI2D_TILE - 1) / GI2D_TILE);
    const long long t = (long long)tiles_x * tiles_y;
    BwdWs wsp = carve_bwd_ws(ws, n, m);
    if (m > 0 && t > 0) {
        if (!gids || !bins || !xys || !conics || !colors || !opac || !final_idx || !v_output) {
            set_error("rasterize backward: null input");
            return GI2D_ERR_INVALID_ARGUMENT;
        }
        if ((long long)rows < t) {
            // bins do not cover the grid: positions no tile claims must still read as zero
            hipError_t e = hipMemsetAsync(wsp.partials, 0, sizeof(float4) * 3 * (size_t)m, st);
            if (e != hipSuccess) {
                set_error(hipGetErrorString(e));
                return (int)e;
            }
        }
        if (v_abs_xy)
            hipLaunchKernelGGL(raster_bwd_kernel<true>, dim3((unsigned)t), dim3(256), 0, st, tiles_x,
                               tiles_y, (int)w, (int)h, gids, (const int2 *)bins, rows,
                               (const float2 *)xys, conics, colors, opac, final_idx, v_output,
                               wsp.partials);
        else
            hipLaunchKernelGGL(raster_bwd_kernel<false>, dim3((unsigned)t), dim3(256), 0, st, tiles_x,
                               tiles_y, (int)w, (int)h, gids, (const int2 *)bins, rows,
                               (const float2 *)xys, conics, colors, opac, final_idx, v_output,
                               wsp.partials);
    }
    const int mm = (t > 0) ? m : 0;
    if (cum) {
        hipLaunchKernelGGL(gather_plan_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, mm, cum,
                           inv_perm, wsp.partials, (float2 *)v_xy, v_conic, v_rgb, v_opacity,
                           (float4 *)v_abs_xy);
    } else {
        hipError_t e = hipMemsetAsync(wsp.counts, 0, sizeof(int32_t) * (size_t)n, st);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            return (int)e;
        }
        if (mm > 0)
            hipLaunchKernelGGL(gidx_count_kernel, dim3((mm + 255) / 256), dim3(256), 0, st, mm, n,
                               gids, wsp.counts);
        int rc = launch_exclusive_scan_with_cursor(n, wsp.counts, wsp.start, wsp.cursor, st);
        if (rc != GI2D_OK) return rc;
        if (mm > 0)
            hipLaunchKernelGGL(gidx_scatter_kernel, dim3((mm + 255) / 256), dim3(256), 0, st, mm, n,
                               gids, wsp.start, wsp.cursor, wsp.gslots);
        hipLaunchKernelGGL(gather_generic_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, mm,
                           wsp.start, wsp.gslots, wsp.partials, (float2 *)v_xy, v_conic, v_rgb,
                           v_opacity, (float4 *)v_abs_xy);
    }
    return check_launch("rasterize backward");
}

}  // namespace gi2d

using namespace gi2d;

extern "C" {

size_t gi2d_rasterize_backward_workspace_bytes(int n, int m) {
    return carve_bwd_ws(nullptr, n, m).bytes;
}

int gi2d_rasterize_backward_tiles(unsigned h, unsigned w, const int32_t *gids, const int32_t *bins,
                                  int rows, const float *xys, const float *conics, const float *colors,
                                  const float *opac, const int32_t *final_idx, const float *v_output,
                                  int with_abs, float *partials, gi2d_stream_t st) {
    const int tiles_x = (int)((w + GI2D_TILE - 1) / GI2D_TILE), tiles_y = (int)((h + GI2D_TILE - 1) / GI2D_TILE);
    const long long t = (long long)tiles_x * tiles_y;
    if (t == 0 || rows <= 0) return GI2D_OK;
    if (!gids || !bins || !xys || !conics || !colors || !opac || !final_idx || !v_output || !partials) {
        set_error("rasterize backward tiles: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (with_abs)
        hipLaunchKernelGGL(raster_bwd_kernel<true>, dim3((unsigned)t), dim3(256), 0, (hipStream_t)st, tiles_x,
                           tiles_y, (int)w, (int)h, gids, (const int2 *)bins, rows, (const float2 *)xys, conics,
                           colors, opac, final_idx, v_output, (float4 *)partials);
    else
        hipLaunchKernelGGL(raster_bwd_kernel<false>, dim3((unsigned)t), dim3(256), 0, (hipStream_t)st, tiles_x,
                           tiles_y, (int)w, (int)h, gids, (const int2 *)bins, rows, (const float2 *)xys, conics,
                           colors, opac, final_idx, v_output, (float4 *)partials);
    return check_launch("rasterize backward tiles");
}

int gi2d_rasterize_backward_reduce(int n, const float *xys, const int32_t *radii, int tiles_x, int tiles_y,
                                   float radius_clip, const int32_t *gids, const int32_t *bins, int rows,
                                   const float *partials, float *v_xy, float *v_conic, float *v_rgb,
                                   float *v_opacity, float *v_abs_xy, gi2d_stream_t st) {
    if (n < 0 || rows < 0) {
        set_error("rasterize backward reduce: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return GI2D_OK;
    if (!xys || !radii || !v_xy || !v_conic || !v_rgb || !v_opacity || (rows > 0 && (!gids || !bins || !partials))) {
        set_error("rasterize backward reduce: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(gather_bbox_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, n,
                       (const float2 *)xys, radii, tiles_x, tiles_y, radius_clip, gids, (const int2 *)bins, rows,
                       (const float4 *)partials, (float2 *)v_xy, v_conic, v_rgb, v_opacity, (float4 *)v_abs_xy);
    return check_launch("rasterize backward reduce");
}

int gi2d_rasterize_sum_forward(int tiles_x, int tiles_y, unsigned w, unsigned h,
                               const int32_t *gids, const int32_t *bins, int rows, const float *xys,
                               const float *conics, const float *colors, const float *opac,
                               const float *background, const int32_t *m_dev, float *final_Ts,
                               int32_t *final_idx, float *out_img, gi2d_stream_t st) {
    return raster_forward(tiles_x, tiles_y, w, h, gids, bins, rows, xys, conics, colors, opac,
                          background, m_dev, final_Ts, final_idx, out_img, st);
}
int gi2d_rasterize_sum_plus_forward(int tiles_x, int tiles_y, unsigned w, unsigned h,
                                    const int32_t *gids, const int32_t *bins, int rows,
                                    const float *xys, const float *conics, const float *colors,
                                    const float *opac, const float *background,
                                    const int32_t *m_dev, float *final_Ts, int32_t *final_idx,
                                    float *out_img, gi2d_stream_t st) {
    return raster_forward(tiles_x, tiles_y, w, h, gids, bins, rows, xys, conics, colors, opac,
                          background, m_dev, final_Ts, final_idx, out_img, st);
}
int gi2d_rasterize_sum_backward(int n, int m, unsigned h, unsigned w, const int32_t *gids,
                                const int32_t *bins, int rows, const float *xys,
                                const float *conics, const float *colors, const float *opac,
                                const int32_t *final_idx, const float *v_output, const int32_t *cum,
                                const int32_t *inv_perm, float *v_xy, float *v_conic, float *v_rgb,
                                float *v_opacity, float *v_abs_xy, void *ws, size_t ws_bytes,
                                gi2d_stream_t st) {
    return raster_backward(n, m, h, w, gids, bins, rows, xys, conics, colors, opac, final_idx,
                           v_output, cum, inv_perm, v_xy, v_conic, v_rgb, v_opacity, v_abs_xy, ws,
                           ws_bytes, st);
}
int gi2d_rasterize_sum_plus_backward(int n, int m, unsigned h, unsigned w, const int32_t *gids,
                                     const int32_t *bins, int rows, const float *xys,
                                     const float *conics, const float *colors, const float *opac,
                                     const int32_t *final_idx, const float *v_output,
                                     const int32_t *cum, const int32_t *inv_perm, float *v_xy,
                                     float *v_conic, float *v_rgb, float *v_opacity, void *ws,
                                     size_t ws_bytes, gi2d_stream_t st) {
    return raster_backward(n, m, h, w, gids, bins, rows, xys, conics, colors, opac, final_idx,
                           v_output, cum, inv_perm, v_xy, v_conic, v_rgb, v_opacity, nullptr, ws,
                           ws_bytes, st);
}

}  // extern "C"
