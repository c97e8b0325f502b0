// Sync-free tile binning straight from the projected gaussians (xys, radii) -- the fast path
// the rasterize wrappers and bench.py use instead of cumsum -> map -> sort -> bin edges.
//
// Result (bit-identical to the reference pipeline followed by a STABLE key sort, i.e. to
// oracle bin_and_sort_gaussians): gaussian_ids_sorted = for every tile, the ids of the gaussians
// whose 3-sigma tile box (helpers.cuh:16-50) covers it, ascending; tile_bins = [start, end).
// What is gone: the inclusive scan over all N gaussians, the 64-bit keys, the Gaussian-major
// (key, id) arrays and the host read-back of the intersection count -- the count lives in
// status[0] on the device and buffers are sized by a caller-chosen capacity.
//
//   count : one lane per gaussian, int atomics into counts[tile][sub]; SUB sub-counters per tile
//           (sub = id mod SUB) cut same-address contention 8x (72 adders per tile at N=50k).
//   scan  : one workgroup, exclusive scan of the T*SUB counters -> start, tile_bins, status.
//   fill  : one lane per gaussian claims a slot in its (tile, sub) segment (returning atomics).
//   order : one workgroup per tile sorts the tile's ids ascending in LDS (rank sort, or a bitmap
//           sweep over the id space for lists longer than 1024).
// Integer work, HBM-light: 12 B read per gaussian twice, 4 B written + 4 B read + 4 B written
// per intersection.
#include "gi2d_common.h"

namespace gi2d {

#define GI2D_SUB 8
#define GI2D_FB_RANK_MAX 1024
#define GI2D_FB_BITMAP_WORDS 8192

__device__ __forceinline__ bool mapped_bbox(int idx, const float2 *__restrict__ xys,
                                            const int32_t *__restrict__ radii, int tiles_x, int tiles_y,
                                            float radius_clip, int &mnx, int &mny, int &mxx, int &mxy) {
    const int rad = radii[idx];
    // forward.cu:161 skips `radii < radius_clip`; radii <= 0 marks a culled gaussian whose slot
    // budget (num_tiles_hit) is zero, so it never owns an intersection either.
    if (rad <= 0 || (float)rad < radius_clip) return false;
    const float2 c = xys[idx];
    tile_bbox(c.x, c.y, (float)rad, tiles_x, tiles_y, mnx, mny, mxx, mxy);
    return mxx > mnx && mxy > mny;
}

__global__ __launch_bounds__(256) void fb_count_kernel(int n, const float2 *__restrict__ xys,
                                                       const int32_t *__restrict__ radii, int tiles_x,
                                                       int tiles_y, float radius_clip,
                                                       int32_t *__restrict__ counts) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int mnx, mny, mxx, mxy;
    if (!mapped_bbox(idx, xys, radii, tiles_x, tiles_y, radius_clip, mnx, mny, mxx, mxy)) return;
    const int sub = idx & (GI2D_SUB - 1);
    for (int i = mny; i < mxy; ++i)
        for (int j = mnx; j < mxx; ++j) atomicAdd(&counts[(i * tiles_x + j) * GI2D_SUB + sub], 1);
}

// One workgroup: exclusive scan of counts[T*SUB] -> start[T*SUB+1]; cursor = 0; tile_bins;
// status = {M, M > capacity, 0, 0}.
__global__ __launch_bounds__(1024) void fb_scan_kernel(int num_tiles, int capacity,
                                                       const int32_t *__restrict__ counts,
                                                       int32_t *__restrict__ start,
                                                       int32_t *__restrict__ cursor,
                                                       int32_t *__restrict__ bins,
                                                       int32_t *__restrict__ status) {
    __shared__ int wave_sums[16];
    __shared__ int carry_s;
    const int n = num_tiles * GI2D_SUB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 8192) {
        // 8 consecutive counters per lane = one tile's sub-counters
        const int i0 = base + tid * GI2D_SUB;
        int v[GI2D_SUB];
        int local = 0;
        if (i0 < n) {
            const int4 a = *reinterpret_cast<const int4 *>(counts + i0);
            const int4 b = *reinterpret_cast<const int4 *>(counts + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
            for (int k = 0; k < GI2D_SUB; ++k) local += v[k];
        } else {
#pragma unroll
            for (int k = 0; k < GI2D_SUB; ++k) v[k] = 0;
        }
        const int incl = wave_inclusive_scan(local);
        if (lane == 63) wave_sums[wv] = incl;
        __syncthreads();
        int wave_off = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) wave_off += (k < wv) ? wave_sums[k] : 0;
        int run = carry_s + wave_off + incl - local;
        if (i0 < n) {
            const int tile = i0 / GI2D_SUB;
            bins[2 * tile] = local > 0 ? min(run, capacity) : 0;
            bins[2 * tile + 1] = local > 0 ? min(run + local, capacity) : 0;
            int s[GI2D_SUB];
#pragma unroll
            for (int k = 0; k < GI2D_SUB; ++k) {
                s[k] = run;
                run += v[k];
            }
            *reinterpret_cast<int4 *>(start + i0) = make_int4(s[0], s[1], s[2], s[3]);
            *reinterpret_cast<int4 *>(start + i0 + 4) = make_int4(s[4], s[5], s[6], s[7]);
            const int4 z = make_int4(0, 0, 0, 0);
            *reinterpret_cast<int4 *>(cursor + i0) = z;
            *reinterpret_cast<int4 *>(cursor + i0 + 4) = z;
        }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) {
        const int total = carry_s;
        start[n] = total;
        status[0] = total;
        status[1] = total > capacity ? 1 : 0;
        status[2] = 0;
        status[3] = 0;
    }
}

__global__ __launch_bounds__(256) void fb_fill_kernel(int n, int capacity, const float2 *__restrict__ xys,
                                                      const int32_t *__restrict__ radii, int tiles_x,
                                                      int tiles_y, float radius_clip,
                                                      const int32_t *__restrict__ start,
                                                      int32_t *__restrict__ cursor,
                                                      int32_t *__restrict__ unsorted) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int mnx, mny, mxx, mxy;
    if (!mapped_bbox(idx, xys, radii, tiles_x, tiles_y, radius_clip, mnx, mny, mxx, mxy)) return;
    const int sub = idx & (GI2D_SUB - 1);
    for (int i = mny; i < mxy; ++i)
        for (int j = mnx; j < mxx; ++j) {
            const int c = (i * tiles_x + j) * GI2D_SUB + sub;
            const int p = start[c] + atomicAdd(&cursor[c], 1);
            if (p < capacity) unsorted[p] = idx;
        }
}

// One workgroup per tile: ascending ids.
__global__ __launch_bounds__(256) void fb_order_kernel(int n, int capacity, const int32_t *__restrict__ start,
                                                       const int32_t *__restrict__ unsorted,
                                                       int32_t *__restrict__ gids_sorted) {
    __shared__ union {
        int ids[GI2D_FB_RANK_MAX];
        uint32_t bits[GI2D_FB_BITMAP_WORDS];
    } sm;
    __shared__ int wsum[4];
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s0 = min(start[tile * GI2D_SUB], capacity);
    const int len = min(start[(tile + 1) * GI2D_SUB], capacity) - s0;
    if (len <= 0) return;
    if (len <= GI2D_FB_RANK_MAX) {
        for (int e = tid; e < len; e += 256) sm.ids[e] = unsorted[s0 + e];
        __syncthreads();
        for (int e = tid; e < len; e += 256) {
            const int mine = sm.ids[e];
            int rank = 0;
            for (int j = 0; j < len; ++j) rank += (sm.ids[j] < mine) ? 1 : 0;
            gids_sorted[s0 + rank] = mine;
        }
        return;
    }
    int emitted = 0;
    for (int win = 0; win < n; win += 32 * GI2D_FB_BITMAP_WORDS) {
        for (int w = tid; w < GI2D_FB_BITMAP_WORDS; w += 256) sm.bits[w] = 0u;
        __syncthreads();
        for (int e = tid; e < len; e += 256) {
            const int rel = unsorted[s0 + e] - win;
            if (rel >= 0 && rel < 32 * GI2D_FB_BITMAP_WORDS) atomicOr(&sm.bits[rel >> 5], 1u << (rel & 31));
        }
        __syncthreads();
        const int w0 = tid * (GI2D_FB_BITMAP_WORDS / 256);
        int cnt = 0;
        for (int w = 0; w < GI2D_FB_BITMAP_WORDS / 256; ++w) cnt += __popc(sm.bits[w0 + w]);
        const int incl = wave_inclusive_scan(cnt);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int off = emitted + incl - cnt;
        for (int k = 0; k < wv; ++k) off += wsum[k];
        const int win_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        for (int w = 0; w < GI2D_FB_BITMAP_WORDS / 256; ++w) {
            uint32_t b = sm.bits[w0 + w];
            while (b) {
                const int bit = __ffs(b) - 1;
                b &= b - 1;
                gids_sorted[s0 + off++] = win + ((w0 + w) << 5) + bit;
            }
        }
        emitted += win_total;
        __syncthreads();
    }
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
struct FbWs {
    int32_t *counts, *start, *cursor, *unsorted;
    size_t bytes;
};
static FbWs carve_fb(void *base, int capacity, int num_tiles) {
    FbWs w;
    char *b = (char *)base;
    size_t off = 0;
    const size_t c = (size_t)(num_tiles > 0 ? num_tiles : 1) * GI2D_SUB;
    w.counts = (int32_t *)(b + off);
    off += align_up(c * sizeof(int32_t));
    w.start = (int32_t *)(b + off);
    off += align_up((c + 1) * sizeof(int32_t));
    w.cursor = (int32_t *)(b + off);
    off += align_up(c * sizeof(int32_t));
    w.unsorted = (int32_t *)(b + off);
    off += align_up((size_t)(capacity > 0 ? capacity : 1) * sizeof(int32_t));
    w.bytes = off;
    return w;
}

}  // namespace gi2d

using namespace gi2d;

extern "C" {

size_t gi2d_bin_workspace_bytes(int capacity, int num_tiles) {
    return carve_fb(nullptr, capacity, num_tiles).bytes;
}

int gi2d_bin_gaussians(int n, int capacity, const float *xys, const int32_t *radii, int tiles_x,
                       int tiles_y, float radius_clip, int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                       int32_t *status, void *workspace, size_t ws_bytes, gi2d_stream_t st_) {
    hipStream_t st = (hipStream_t)st_;
    if (n < 0 || capacity < 0 || tiles_x < 0 || tiles_y < 0) {
        set_error("bin_gaussians: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    const long long t = (long long)tiles_x * tiles_y;
    if (t == 0) return GI2D_OK;
    if (t * GI2D_SUB > 0x7fffffffLL) {
        set_error("bin_gaussians: tile grid too large");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (!tile_bins || !status || (capacity > 0 && !gaussian_ids_sorted) || (n > 0 && (!xys || !radii))) {
        set_error("bin_gaussians: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (!workspace || ws_bytes < gi2d_bin_workspace_bytes(capacity, (int)t)) {
        set_error("bin_gaussians: workspace too small");
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    FbWs w = carve_fb(workspace, capacity, (int)t);
    hipError_t e = hipMemsetAsync(w.counts, 0, sizeof(int32_t) * (size_t)t * GI2D_SUB, st);
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        return (int)e;
    }
    if (n > 0)
        hipLaunchKernelGGL(fb_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, (const float2 *)xys,
                           radii, tiles_x, tiles_y, radius_clip, w.counts);
    hipLaunchKernelGGL(fb_scan_kernel, dim3(1), dim3(1024), 0, st, (int)t, capacity, w.counts, w.start,
                       w.cursor, tile_bins, status);
    if (n > 0 && capacity > 0) {
        hipLaunchKernelGGL(fb_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, capacity,
                           (const float2 *)xys, radii, tiles_x, tiles_y, radius_clip, w.start, w.cursor,
                           w.unsorted);
        hipLaunchKernelGGL(fb_order_kernel, dim3((unsigned)t), dim3(256), 0, st, n, capacity, w.start,
                           w.unsorted, gaussian_ids_sorted);
    }
    return check_launch("bin_gaussians");
}

}  // extern "C"
