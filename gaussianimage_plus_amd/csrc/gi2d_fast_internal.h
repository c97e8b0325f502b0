// Internals of the fused fast path shared by gi2d_fast.hip and gi2d_train.hip: workspace layout, bucket
// fill, partial-row addressing and the ordered per-gaussian reduction.
#pragma once
#include "gi2d_project_core.h"
#include "gi2d_raster_core.h"

namespace gi2d {


#define GI2D_FAST_SUB 4
#define GI2D_FAST_CSUB 256
#define GI2D_FAST_C (GI2D_FAST_SUB * GI2D_FAST_CSUB) /* list slots per tile */
#define GI2D_FAST_EPT (GI2D_FAST_C / 256)            /* bucket entries per lane of the 256-lane tile workgroup */
#define GI2D_FAST_S 16                               /* gaussian-major partial rows per gaussian */
#ifndef GI2D_FILL_BATCH
#define GI2D_FILL_BATCH 16                           /* bucket atomics a lane keeps in flight */
#endif
#ifndef GI2D_REDUCE_BATCH
#define GI2D_REDUCE_BATCH 8                          /* partial rows a lane loads before it starts adding */
#endif
#define GI2D_BIG_TILES_F 32
#define GI2D_FAST_ROW 4 /* float4 per partial row: 48 bytes of data padded to one 64-byte line */
#define GI2D_CURSOR_STRIDE 16 /* ints between cursors: one 64-byte line each, so atomics on different cursors never share a line */

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
// Workgroup size of the one-lane-per-gaussian kernels (project+fill, reduce+project backward, optimizer update):
// single waves spread a small population over many CUs and shorten the dependent atomic / load chains
// (N=2500: project+fill 10.9 -> 5.5 us, reduce 7.4 -> 4.7 us; neutral to slightly worse beyond ~30k gaussians).
#ifndef GI2D_PG_BIG
#define GI2D_PG_BIG 256
#endif
static inline int per_gaussian_block(int n) { return n <= 32768 ? 64 : GI2D_PG_BIG; }
struct FastWs {
    int32_t *cursors;      // [T * SUB]          zero between calls
    int32_t *buckets;      // [T * C]            unsorted ids per (tile, sub)
    int32_t *gids_sorted;  // [T * C]            ascending ids per tile (stride C)
    int32_t *tile_bins;    // [T * 2]            [t*C, t*C + len)
    GaussRec *packed;      // [T * 256]          tile-sorted records of the first 256 entries
    float4 *partial_g;     // [N * S * 4]        gaussian-major partial rows (64 B each)
    float4 *partial_big;   // [T * 256 * 4]      partial rows of gaussians on > S tiles, by (tile, rank)
    int32_t *tile_order;   // [T]                tile handled by workgroup b of the single-pass tile kernel: a
                           //                    permutation that balances tile populations over the CUs
    size_t bytes;
};
static FastWs carve_fast(void *base, int n, int num_tiles) {
    FastWs w;
    char *b = (char *)base;
    size_t off = 0;
    const size_t t = (size_t)(num_tiles > 0 ? num_tiles : 1), nn = (size_t)(n > 0 ? n : 1);
    w.cursors = (int32_t *)(b + off);
    off += align_up(t * GI2D_FAST_SUB * GI2D_CURSOR_STRIDE * sizeof(int32_t));
    w.buckets = (int32_t *)(b + off);
    off += align_up(t * GI2D_FAST_C * sizeof(int32_t));
    w.gids_sorted = (int32_t *)(b + off);
    off += align_up(t * GI2D_FAST_C * sizeof(int32_t));
    w.tile_bins = (int32_t *)(b + off);
    off += align_up(t * 2 * sizeof(int32_t));
    w.packed = (GaussRec *)(b + off);
    off += align_up(t * GI2D_TILE_LIST_CAP * sizeof(GaussRec));
    // cursors and tile_order carry state from call to call: both sit in front of every region whose offset depends
    // on the gaussian count, so a workspace initialised for a capacity can be used with any smaller population
    w.tile_order = (int32_t *)(b + off);
    off += align_up(t * sizeof(int32_t));
    w.partial_g = (float4 *)(b + off);
    off += align_up(nn * GI2D_FAST_S * GI2D_FAST_ROW * sizeof(float4));
    w.partial_big = (float4 *)(b + off);
    off += align_up(t * GI2D_TILE_LIST_CAP * GI2D_FAST_ROW * sizeof(float4));
    w.bytes = off;
    return w;
}

// ----------------------------------------------------------------------------------------- fill
__device__ __forceinline__ void fill_one(int g, int mnx, int mny, int mxx, int mxy, int tiles_x,
                                         int32_t *__restrict__ cursors, int32_t *__restrict__ buckets) {
    const int sub = g & (GI2D_FAST_SUB - 1);
    const int w = mxx - mnx, nt = w * (mxy - mny);
    // GI2D_FILL_BATCH tiles per trip: the returning atomics of a trip are issued back to back, then the stores, so
    // a lane pays one L2 round trip per trip instead of one per tile (dependent atomics cost ~2 us each: a gaussian
    // on 3x3 tiles used to spend 9 of them in a row).
    int di = 0, dj = 0;
    for (int base = 0; base < nt; base += GI2D_FILL_BATCH) {
        int c[GI2D_FILL_BATCH], p[GI2D_FILL_BATCH];
#pragma unroll
        for (int q = 0; q < GI2D_FILL_BATCH; ++q) {
            c[q] = (base + q < nt) ? ((mny + di) * tiles_x + mnx + dj) * GI2D_FAST_SUB + sub : -1;
            if (++dj == w) dj = 0, ++di;
        }
#pragma unroll
        for (int q = 0; q < GI2D_FILL_BATCH; ++q)
            p[q] = c[q] >= 0 ? atomicAdd(&cursors[c[q] * GI2D_CURSOR_STRIDE], 1) : GI2D_FAST_CSUB;
#pragma unroll
        for (int q = 0; q < GI2D_FILL_BATCH; ++q)
            if (p[q] < GI2D_FAST_CSUB) buckets[c[q] * GI2D_FAST_CSUB + p[q]] = g;
    }
}

// partial-row code of gaussian g in tile (tx, ty): >= 0 gaussian-major row, < 0: -(big row) - 1
__device__ __forceinline__ int partial_slot(int g, const float2 xy, int rad, int tiles_x, int tiles_y, int tx,
                                            int ty, int big_row) {
    int mnx, mny, mxx, mxy;
    tile_bbox(xy.x, xy.y, (float)rad, tiles_x, tiles_y, mnx, mny, mxx, mxy);
    const int w = mxx - mnx, ntiles = w * (mxy - mny);
    if (ntiles <= GI2D_FAST_S) return g * GI2D_FAST_S + (ty - mny) * w + (tx - mnx);
    return -big_row - 1;
}

// ------------------------------------------------------------------------------- tile order
// All workgroups of the single-pass tile kernel are resident at once on a 768x512 image (6 per CU), and the
// dispatcher was observed to put workgroups b, b + 256, b + 512, ... on the same CU; a CU is done when its six
// tiles are, and tile populations differ (40...108 gaussians at N=50 000), so the slowest CU finished 5 us after
// the fastest.  This routine -- one extra workgroup of the training update kernel, where it hides behind the other
// workgroups -- sorts the tiles by the population the tile pass has just seen (counting sort, descending) and deals
// them to workgroup indices in snake order over the 256 CU slots, for the NEXT iteration's tile pass (populations
// drift slowly during training).  Any permutation gives the same results; placement is a pure speed choice
// (dispatch order is undefined by contract).  Measured: training iteration at N=50 000 50.2 -> 44.6 us; hot-path
// step 38.3 -> 37.0 us.  Its LDS histogram atomics (many lanes per bin) take ~5 us, so it only rides on kernels that
// are longer than that (the training update kernel; the end-of-step kernel above 32k gaussians).
#define GI2D_ORDER_BINS 1025 /* populations 0 .. GI2D_FAST_C */
#define GI2D_CU_SLOTS 256
#define GI2D_ORDER_MAX_TILES 2048 /* larger grids run in several rounds of resident workgroups and balance themselves */
__device__ __forceinline__ void compute_tile_order(const int2 *__restrict__ tile_bins, int num_tiles,
                                                   int32_t *__restrict__ tile_order) {
    if (num_tiles <= GI2D_CU_SLOTS || num_tiles > GI2D_ORDER_MAX_TILES) return;  // order stays the identity
    __shared__ int hist[GI2D_ORDER_BINS + 1];
    const int tid = threadIdx.x, bs = blockDim.x;  // 64 or 256 lanes
    for (int b = tid; b <= GI2D_ORDER_BINS; b += bs) hist[b] = 0;
    // every lane's (<= 32) tile populations: all loads in flight together, kept in registers for both passes (a
    // lane-serial chain of dependent loads would put this workgroup on the kernel's critical path)
    constexpr int PL = GI2D_ORDER_MAX_TILES / 64;
    int pop[PL];
#pragma unroll
    for (int q = 0; q < PL; ++q) {
        const int t = tid + q * bs;
        const int2 r = t < num_tiles ? tile_bins[t] : make_int2(0, 0);
        pop[q] = min(max(r.y - r.x, 0), GI2D_ORDER_BINS - 1);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PL; ++q)
        if (tid + q * bs < num_tiles) atomicAdd(&hist[GI2D_ORDER_BINS - 1 - pop[q]], 1);  // bin 0 = fullest tiles
    __syncthreads();
    // exclusive scan of the bins by the first wave: 17 bins per lane in registers, one wave scan of the lane sums
    if (tid < 64) {
        constexpr int PER = (GI2D_ORDER_BINS + 63) / 64;
        int v[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int b = tid * PER + q;
            v[q] = b < GI2D_ORDER_BINS ? hist[b] : 0;
            sum += v[q];
        }
        int run = wave_inclusive_scan(sum) - sum;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int b = tid * PER + q;
            if (b < GI2D_ORDER_BINS) hist[b] = run;
            run += v[q];
        }
    }
    __syncthreads();
    const int full_rounds = num_tiles / GI2D_CU_SLOTS;
#pragma unroll
    for (int q = 0; q < PL; ++q) {
        const int t = tid + q * bs;
        if (t >= num_tiles) continue;
        const int rank = atomicAdd(&hist[GI2D_ORDER_BINS - 1 - pop[q]], 1);  // ties: any order
        const int round = rank / GI2D_CU_SLOTS, pos = rank % GI2D_CU_SLOTS;
        const int slot = (round < full_rounds && (round & 1)) ? GI2D_CU_SLOTS - 1 - pos : pos;
        tile_order[round * GI2D_CU_SLOTS + slot] = t;
    }
}

// acc[11] <- ordered sum of gaussian g's partial rows.  Must be called by whole waves.
__device__ __forceinline__ void reduce_one(int g, int n, const float2 *__restrict__ xys,
                                           const int32_t *__restrict__ radii, int tiles_x, int tiles_y,
                                           float radius_clip, const int32_t *__restrict__ gids_sorted,
                                           const int2 *__restrict__ tile_bins, int num_tiles,
                                           const float4 *__restrict__ partial_g,
                                           const float4 *__restrict__ partial_big, float (&acc)[11]) {
    const int lane = threadIdx.x & 63;
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
    if (mapped && ntiles <= GI2D_FAST_S) {
        // GI2D_REDUCE_BATCH rows per trip: their loads are in flight together, the additions stay in ascending
        // tile order
        const float4 *rows = partial_g + GI2D_FAST_ROW * ((size_t)g * GI2D_FAST_S);
        for (int k0 = 0; k0 < ntiles; k0 += GI2D_REDUCE_BATCH) {
            float4 r[GI2D_REDUCE_BATCH][3];
#pragma unroll
            for (int q = 0; q < GI2D_REDUCE_BATCH; ++q)
                if (k0 + q < ntiles) {
                    r[q][0] = rows[GI2D_FAST_ROW * (k0 + q)];
                    r[q][1] = rows[GI2D_FAST_ROW * (k0 + q) + 1];
                    r[q][2] = rows[GI2D_FAST_ROW * (k0 + q) + 2];
                }
#pragma unroll
            for (int q = 0; q < GI2D_REDUCE_BATCH; ++q)
                if (k0 + q < ntiles) add_partial_row(acc, r[q][0], r[q][1], r[q][2]);
        }
    } else if (mapped && ntiles <= GI2D_BIG_TILES_F) {
        for (int i = mny; i < mxy; ++i)
            for (int j = mnx; j < mxx; ++j) {
                const int tile = i * tiles_x + j;
                const int pos = find_in_tile(gids_sorted, tile_bins, tile, num_tiles, g);
                if (pos >= 0)
                    add_partial<GI2D_FAST_ROW>(acc, partial_big, (size_t)tile * GI2D_TILE_LIST_CAP + (pos - tile * GI2D_FAST_C));
            }
    }
    unsigned long long big = __ballot(mapped && ntiles > GI2D_BIG_TILES_F);
    while (big) {  // a gaussian on > 32 tiles: the whole wave strides over its tiles
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
            const int pos = find_in_tile(gids_sorted, tile_bins, tile, num_tiles, bg);
            if (pos >= 0)
                add_partial<GI2D_FAST_ROW>(part, partial_big, (size_t)tile * GI2D_TILE_LIST_CAP + (pos - tile * GI2D_FAST_C));
        }
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            float v = part[q];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == src) acc[q] = v;
        }
    }
}


}  // namespace gi2d
