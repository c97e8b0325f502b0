// Tile binning for the 2D-Gaussian rasterizer, hand-written for gfx950.
//
// The reference sorts (tile<<32 | depth_bits) int64 keys with cub's 64-bit radix sort through
// torch.sort (gsplat/gsplat/utils.py:301).  The tile id is the only varying part of the key on
// this path (depth == 0), so the sort here is a STABLE COUNTING SORT BY TILE:
//     tile_hist  -> per-tile population (int atomics, order-free)
//     tile_scan  -> exclusive scan over the T tiles (one workgroup) = tile_bins
//     tile_scatter -> each pair claims a slot inside its tile segment (atomic cursor)
//     tile_order -> one workgroup per tile puts its segment into ascending (depth_bits, input
//                   position) order -- an LDS rank sort for short lists, an LDS bitmap sweep for
//                   long ones -- which is exactly what a stable sort of the whole key array gives.
// All integer work; HBM traffic per pair: 8+4 B read three times, 8+4+4(+4) B written: ~60 B.
#include "gi2d_common.h"

namespace gi2d {

// ------------------------------------------------------------------------------------------
// Inclusive scan of int32[n] by ONE workgroup of 1024 lanes, 4 elements per lane per step.
// N <= a few 100k on this path (n = #gaussians or #tiles), so a single launch with no
// inter-workgroup hand-off beats a multi-kernel scan (boundary cost ~1.5 us each).
// mode 0: out[i] = inclusive prefix, *total = sum.
// mode 1: tile mode -- in = counts[T]; out_start[T+1] = exclusive prefix; bins[t] =
//         (start, start+count) or (0,0) when empty; cursor[t] = 0.
template <int MODE>
__global__ __launch_bounds__(1024) void scan_kernel(int n, const int32_t *__restrict__ in,
                                                    int32_t *__restrict__ out,
                                                    int32_t *__restrict__ total,
                                                    int32_t *__restrict__ bins,
                                                    int32_t *__restrict__ cursor) {
    __shared__ int wave_sums[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 4096) {
        const int i0 = base + tid * 4;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? in[i0 + k] : 0;
        const int local = v[0] + v[1] + v[2] + v[3];
        const int incl = wave_inclusive_scan(local);
        if (lane == 63) wave_sums[wv] = incl;
        __syncthreads();
        int wave_off = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) wave_off += (k < wv) ? wave_sums[k] : 0;
        const int carry = carry_s;
        int run = carry + wave_off + incl - local;  // exclusive prefix of this lane's 4 elements
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) {
                if (MODE == 0) {
                    out[i0 + k] = run + v[k];
                } else {
                    out[i0 + k] = run;
                    if (bins) {
                        bins[2 * (i0 + k)] = v[k] > 0 ? run : 0;
                        bins[2 * (i0 + k) + 1] = v[k] > 0 ? run + v[k] : 0;
                    }
                    cursor[i0 + k] = 0;
                }
            }
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) {
        if (MODE == 0) {
            if (total) *total = carry_s;
        } else {
            out[n] = carry_s;
        }
    }
}

// forward.cu:141-206 map_gaussian_to_intersects (radius_clip overload).  One lane per gaussian;
// gaussians are few-tile objects here (mean 2-7 tiles), so the serial tile loop is short.
__global__ __launch_bounds__(256) void map_kernel(int n, int m, const float2 *__restrict__ xys,
                                                  const float *__restrict__ depths,
                                                  const int32_t *__restrict__ radii,
                                                  const int32_t *__restrict__ cum, int tiles_x,
                                                  int tiles_y, float radius_clip,
                                                  int64_t *__restrict__ isect_ids,
                                                  int32_t *__restrict__ gaussian_ids) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int begin = idx == 0 ? 0 : cum[idx - 1];
    const int end = cum[idx];
    int cur = begin;
    const int rad = radii[idx];
    if (!((float)rad < radius_clip)) {  // forward.cu:161 (int radius vs float clip)
        const float2 c = xys[idx];
        int mnx, mny, mxx, mxy;
        tile_bbox(c.x, c.y, (float)rad, tiles_x, tiles_y, mnx, mny, mxx, mxy);
        const int64_t depth_id = (int64_t)__float_as_int(depths[idx]);
        for (int i = mny; i < mxy; ++i)
            for (int j = mnx; j < mxx; ++j) {
                if (cur >= 0 && cur < m) {
                    isect_ids[cur] = ((int64_t)(i * tiles_x + j) << 32) | depth_id;
                    gaussian_ids[cur] = idx;
                }
                ++cur;
            }
    }
    // slots this gaussian owns but did not fill keep the reference's torch::zeros content
    for (int k = max(cur, 0); k < end && k < m; ++k) {
        isect_ids[k] = 0;
        gaussian_ids[k] = 0;
    }
    if (idx == n - 1)
        for (int k = max(end, 0); k < m; ++k) {  // tail beyond cum[-1] (inconsistent callers only)
            isect_ids[k] = 0;
            gaussian_ids[k] = 0;
        }
}

// flags[0] |= 1 if any key has non-zero depth bits; flags[1] |= 1 if a tile id is out of range;
// flags[2] |= 1 if a long tile with non-zero depths was met (unsupported).
__global__ __launch_bounds__(256) void tile_hist_kernel(int m, int num_tiles,
                                                        const int64_t *__restrict__ isect_ids,
                                                        int32_t *__restrict__ counts,
                                                        int32_t *__restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int64_t key = isect_ids[i];
    const int tile = (int)(key >> 32);
    if ((uint32_t)key != 0u) atomicOr(&flags[0], 1);
    if (tile < 0 || tile >= num_tiles) {
        atomicOr(&flags[1], 1);
        return;
    }
    atomicAdd(&counts[tile], 1);
}

__global__ __launch_bounds__(256) void tile_scatter_kernel(int m, int num_tiles,
                                                           const int64_t *__restrict__ isect_ids,
                                                           const int32_t *__restrict__ start,
                                                           int32_t *__restrict__ cursor,
                                                           int32_t *__restrict__ slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int tile = (int)(isect_ids[i] >> 32);
    if (tile < 0 || tile >= num_tiles) return;
    const int p = start[tile] + atomicAdd(&cursor[tile], 1);
    slots[p] = i;
}

#define GI2D_RANK_MAX 1024   /* longest tile list ordered by the LDS rank sort */
#define GI2D_BITMAP_WORDS 8192 /* 32 KiB LDS bitmap = 262144 input positions per sweep */

// One workgroup per tile: order the tile's segment of `slots` and emit the sorted arrays.
__global__ __launch_bounds__(256) void tile_order_kernel(
    int m, const int64_t *__restrict__ isect_ids, const int32_t *__restrict__ gaussian_ids,
    const int32_t *__restrict__ start, const int32_t *__restrict__ slots,
    int64_t *__restrict__ isect_sorted, int32_t *__restrict__ gids_sorted,
    int32_t *__restrict__ perm, int32_t *__restrict__ inv_perm, int32_t *__restrict__ flags) {
    __shared__ union {
        unsigned long long keys[GI2D_RANK_MAX];
        uint32_t bits[GI2D_BITMAP_WORDS];
    } sm;
    __shared__ int wsum[4];
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s0 = start[tile], len = start[tile + 1] - s0;
    if (len <= 0) return;
    if (len <= GI2D_RANK_MAX) {
        for (int e = tid; e < len; e += 256) {
            const int slot = slots[s0 + e];
            const uint32_t lo = (uint32_t)isect_ids[slot];
            sm.keys[e] = ((unsigned long long)lo << 32) | (uint32_t)slot;
        }
        __syncthreads();
        for (int e = tid; e < len; e += 256) {
            const unsigned long long mine = sm.keys[e];
            int rank = 0;
            for (int j = 0; j < len; ++j) rank += (sm.keys[j] < mine) ? 1 : 0;
            const int slot = (int)(uint32_t)mine;
            const int pos = s0 + rank;
            gids_sorted[pos] = gaussian_ids[slot];
            if (isect_sorted) isect_sorted[pos] = isect_ids[slot];
            if (perm) perm[pos] = slot;
            if (inv_perm) inv_perm[slot] = pos;
        }
        return;
    }
    if (flags[0] != 0) {  // long list with real depth keys: not supported on the 2D path
        if (tid == 0) atomicOr(&flags[2], 1);
        return;
    }
    // Long list, all depth bits zero: order == ascending input position.  Sweep the position
    // space [0, m) in windows of 32*GI2D_BITMAP_WORDS, mark the tile's positions in an LDS bitmap
    // and enumerate the set bits in order.
    int emitted = 0;
    for (int win = 0; win < m; win += 32 * GI2D_BITMAP_WORDS) {
        for (int w = tid; w < GI2D_BITMAP_WORDS; w += 256) sm.bits[w] = 0u;
        __syncthreads();
        for (int e = tid; e < len; e += 256) {
            const int rel = slots[s0 + e] - win;
            if (rel >= 0 && rel < 32 * GI2D_BITMAP_WORDS) atomicOr(&sm.bits[rel >> 5], 1u << (rel & 31));
        }
        __syncthreads();
        const int w0 = tid * (GI2D_BITMAP_WORDS / 256);
        int cnt = 0;
        for (int w = 0; w < GI2D_BITMAP_WORDS / 256; ++w) cnt += __popc(sm.bits[w0 + w]);
        const int incl = wave_inclusive_scan(cnt);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int off = emitted + incl - cnt;
        for (int k = 0; k < wv; ++k) off += wsum[k];
        const int win_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        for (int w = 0; w < GI2D_BITMAP_WORDS / 256; ++w) {
            uint32_t b = sm.bits[w0 + w];
            while (b) {
                const int bit = __ffs(b) - 1;
                b &= b - 1;
                const int slot = win + ((w0 + w) << 5) + bit;
                const int pos = s0 + off++;
                gids_sorted[pos] = gaussian_ids[slot];
                if (isect_sorted) isect_sorted[pos] = isect_ids[slot];
                if (perm) perm[pos] = slot;
                if (inv_perm) inv_perm[slot] = pos;
            }
        }
        emitted += win_total;
        __syncthreads();
    }
}

// forward.cu:211-233 get_tile_bin_edges; rows indexed by tile id, zero-filled first.
__global__ __launch_bounds__(256) void bin_edges_kernel(int m, const int64_t *__restrict__ sorted,
                                                        int rows, int32_t *__restrict__ bins) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m) return;
    const int cur = (int)(sorted[idx] >> 32);
    const bool cur_ok = cur >= 0 && cur < rows;
    if (idx == 0 && cur_ok) bins[2 * cur] = 0;
    if (idx == m - 1 && cur_ok) bins[2 * cur + 1] = m;
    if (idx == 0) return;
    const int prev = (int)(sorted[idx - 1] >> 32);
    if (prev != cur) {
        if (prev >= 0 && prev < rows) bins[2 * prev + 1] = idx;
        if (cur_ok) bins[2 * cur] = idx;
    }
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct SortWs {
    int32_t *counts, *start, *cursor, *slots, *flags;
    size_t bytes;
};
static SortWs carve_sort_ws(void *base, int m, int num_tiles) {
    SortWs w;
    size_t off = 0;
    char *b = (char *)base;
    w.flags = (int32_t *)(b + off);
    off += align_up(4 * sizeof(int32_t));
    w.counts = (int32_t *)(b + off);
    off += align_up(sizeof(int32_t) * (size_t)(num_tiles > 0 ? num_tiles : 1));
    w.start = (int32_t *)(b + off);
    off += align_up(sizeof(int32_t) * ((size_t)(num_tiles > 0 ? num_tiles : 1) + 1));
    w.cursor = (int32_t *)(b + off);
    off += align_up(sizeof(int32_t) * (size_t)(num_tiles > 0 ? num_tiles : 1));
    w.slots = (int32_t *)(b + off);
    off += align_up(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    w.bytes = off;
    return w;
}

// exclusive scan counts[n] -> start[n+1], cursor[n] = 0 (used by the backward's index build)
int launch_exclusive_scan_with_cursor(int n, const int32_t *counts, int32_t *start, int32_t *cursor,
                                      hipStream_t st) {
    hipLaunchKernelGGL(scan_kernel<1>, dim3(1), dim3(1024), 0, st, n, counts, start,
                       (int32_t *)nullptr, (int32_t *)nullptr, cursor);
    return check_launch("exclusive scan");
}

}  // namespace gi2d

using namespace gi2d;

extern "C" {

int gi2d_cumsum_tiles_hit(int n, const int32_t *nth, int32_t *cum, int32_t *total,
                          gi2d_stream_t st) {
    if (n < 0) {
        set_error("cumsum: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (n > 0 && (!nth || !cum)) {
        set_error("cumsum: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (n == 0 && !total) return GI2D_OK;
    hipLaunchKernelGGL(scan_kernel<0>, dim3(1), dim3(1024), 0, (hipStream_t)st, n, nth, cum, total,
                       (int32_t *)nullptr, (int32_t *)nullptr);
    return check_launch("cumsum");
}

int gi2d_map_gaussian_to_intersects(int n, int m, const float *xys, const float *depths,
                                    const int32_t *radii, const int32_t *cum, int tiles_x,
                                    int tiles_y, float radius_clip, int64_t *isect_ids,
                                    int32_t *gaussian_ids, gi2d_stream_t st) {
    if (n < 0 || m < 0 || tiles_x < 0 || tiles_y < 0) {
        set_error("map_gaussian_to_intersects: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (n == 0 || m == 0) {
        if (m > 0) {
            hipError_t e = hipMemsetAsync(isect_ids, 0, sizeof(int64_t) * (size_t)m, (hipStream_t)st);
            if (e == hipSuccess)
                e = hipMemsetAsync(gaussian_ids, 0, sizeof(int32_t) * (size_t)m, (hipStream_t)st);
            if (e != hipSuccess) {
                set_error(hipGetErrorString(e));
                return (int)e;
            }
        }
        return GI2D_OK;
    }
    if (!xys || !depths || !radii || !cum || !isect_ids || !gaussian_ids) {
        set_error("map_gaussian_to_intersects: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(map_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, n, m,
                       (const float2 *)xys, depths, radii, cum, tiles_x, tiles_y, radius_clip,
                       isect_ids, gaussian_ids);
    return check_launch("map_gaussian_to_intersects");
}

size_t gi2d_sort_workspace_bytes(int m, int num_tiles) {
    return carve_sort_ws(nullptr, m, num_tiles).bytes;
}

int gi2d_sort_intersects(int m, int num_tiles, const int64_t *isect_ids,
                         const int32_t *gaussian_ids, int64_t *isect_sorted, int32_t *gids_sorted,
                         int32_t *perm, int32_t *inv_perm, int32_t *tile_bins, void *workspace,
                         size_t ws_bytes, gi2d_stream_t st_) {
    hipStream_t st = (hipStream_t)st_;
    if (m < 0 || num_tiles < 0) {
        set_error("sort_intersects: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (num_tiles == 0) return GI2D_OK;
    if (!workspace || ws_bytes < gi2d_sort_workspace_bytes(m, num_tiles)) {
        set_error("sort_intersects: workspace too small");
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    if (m > 0 && (!isect_ids || !gaussian_ids || !gids_sorted)) {
        set_error("sort_intersects: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    SortWs w = carve_sort_ws(workspace, m, num_tiles);
    // flags + counts are adjacent at the head of the workspace: one memset node
    hipError_t e = hipMemsetAsync(w.flags, 0, (size_t)((char *)w.start - (char *)w.flags), st);
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        return (int)e;
    }
    if (m > 0)
        hipLaunchKernelGGL(tile_hist_kernel, dim3((m + 255) / 256), dim3(256), 0, st, m, num_tiles,
                           isect_ids, w.counts, w.flags);
    hipLaunchKernelGGL(scan_kernel<1>, dim3(1), dim3(1024), 0, st, num_tiles, w.counts, w.start,
                       (int32_t *)nullptr, tile_bins, w.cursor);
    if (m > 0) {
        hipLaunchKernelGGL(tile_scatter_kernel, dim3((m + 255) / 256), dim3(256), 0, st, m,
                           num_tiles, isect_ids, w.start, w.cursor, w.slots);
        hipLaunchKernelGGL(tile_order_kernel, dim3(num_tiles), dim3(256), 0, st, m, isect_ids,
                           gaussian_ids, w.start, w.slots, isect_sorted, gids_sorted, perm,
                           inv_perm, w.flags);
    }
    return check_launch("sort_intersects");
}

int gi2d_get_tile_bin_edges(int m, const int64_t *sorted, int rows, int32_t *bins,
                            gi2d_stream_t st) {
    if (m < 0 || rows < 0) {
        set_error("get_tile_bin_edges: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (rows == 0) return GI2D_OK;
    if (!bins || (m > 0 && !sorted)) {
        set_error("get_tile_bin_edges: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipError_t e = hipMemsetAsync(bins, 0, sizeof(int32_t) * 2 * (size_t)rows, (hipStream_t)st);
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        return (int)e;
    }
    if (m > 0)
        hipLaunchKernelGGL(bin_edges_kernel, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)st, m,
                           sorted, rows, bins);
    return check_launch("get_tile_bin_edges");
}

}  // extern "C"
