// Fused fast path of the 2D-Gaussian hot path (what the rasterize wrappers and bench.py run).
//
// Same results as the reference-shaped ops (gi2d_project / gi2d_binning / gi2d_raster), restructured
// around what actually costs time on MI355X at these sizes -- kernel boundaries, device-scope atomics and the
// serial dependent-load chains at the head of every tile workgroup -- rather than HBM bytes:
//
//   fill     one lane per gaussian compares its tile box with the box it was binned with last time and appends its
//            id only to the rows of tiles it has ENTERED (persistent tile lists: gi2d_fast_internal.h).  No count
//            pass, no scan, no keys, no sort launch; fused with the projection itself (project_fill).
//   forward  one workgroup per tile: validates its row against the current boxes, orders it by ascending id in LDS
//            (== the stable key sort of the reference pipeline; a ballot scan when nothing was appended), gathers
//            each gaussian ONCE, writes the tile-sorted 48-byte records ("packed list") + tile_bins for the
//            backward, and rasterizes (same code as the plain forward).
//   backward one workgroup per tile reads the packed records with coalesced 16-byte loads (no
//            gather), runs the strip items, and stores each (tile, gaussian) partial straight into
//            a gaussian-major row (gaussians on <= 16 tiles), so that
//   reduce   is a single level of coalesced loads per gaussian (fixed ascending-tile order: bitwise
//            reproducible); optionally fused with the projection backward (reduce_project) and, in a loop,
//            with the projection + fill of the NEXT step (reduce_project<.., FILL_NEXT>).
//   forward+backward  (fast_fwdbwd_kernel, gi2d_fused_core.h) is forward and backward of a tile in ONE
//            workgroup pass -- what a loop runs when the pixel gradient is given or is the L2-loss gradient
//            of the pixel just rendered: two launches per step.
//
// Capacity contract: at most GI2D_FAST_C (1024) candidates per tile row, of which the 256 lowest ids are
// rasterized (forward.cu:553).  A fuller row sets status[1] and the caller must fall back to the exact path
// (gi2d_bin_gaussians + plain ops) and re-initialise the workspace.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <hip/hip_ext.h>

#include "gi2d_fused_core.h"
#include "gi2d_batch.h"

namespace gi2d {

// ----------------------------------------------------------------------------------------- fill
// Binning step on given projection outputs (what the autograd wrappers have: the projection is a separate operator).
__global__ __launch_bounds__(256) void fast_fill_kernel(int n, const float2 *__restrict__ xys,
                                                        const int32_t *__restrict__ radii,
                                                        const float *__restrict__ conics,
                                                        const float *__restrict__ colors,
                                                        const float *__restrict__ opacities, int tiles_x, int tiles_y,
                                                        float radius_clip, PrevBox *__restrict__ prev_box,
                                                        int32_t *__restrict__ lists, RecSets rs,
                                                        int32_t *__restrict__ status) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    begin_binning(g, status);
    const BinRecs recs = recs_for_binning(rs, g == 0);
    if (g >= n) return;
    // forward.cu:161: culled gaussians are in no tile
    bin_one(g, xys[g], radii[g], true, conics[3 * g], conics[3 * g + 1], conics[3 * g + 2], opacities[g], colors[3 * g],
            colors[3 * g + 1], colors[3 * g + 2], tiles_x, tiles_y, radius_clip, prev_box[g], prev_box, lists, recs);
}

// projection of gaussian g + its binning step (g == 0 also resets the per-call status words)
template <int KIND>
__device__ __forceinline__ void project_fill_one(
    int g, int n, float clip_coe, const float2 *__restrict__ means2d, const float *__restrict__ p0,
    const float *__restrict__ p1, float img_w, float img_h, int tiles_x, int tiles_y, float radius_clip, float2 *xys,
    float *__restrict__ depths, int32_t *radii, float *conics, int32_t *__restrict__ num_tiles_hit,
    const BinTarget &bt) {
    begin_binning(g, bt.status);
    const BinRecs recs = recs_for_binning(bt.recs, g == 0);
    if (g >= n) return;
    // every input is requested before the first store below (the outputs may alias them as far as the compiler knows)
    const PrevBox old_box = bt.prev_box[g];
    const float opac = bt.opacities[g], cr = bt.colors[3 * g], cg = bt.colors[3 * g + 1], cb = bt.colors[3 * g + 2];
    const ProjOut o = project_one<KIND>(g, clip_coe, means2d, p0, p1, img_w, img_h, tiles_x, tiles_y, radius_clip);
    xys[g] = o.xy;
    depths[g] = 0.f;
    radii[g] = o.radius;
    conics[3 * g] = o.k0;
    conics[3 * g + 1] = o.k1;
    conics[3 * g + 2] = o.k2;
    num_tiles_hit[g] = o.tiles_hit;
    bin_projected(g, o, opac, cr, cg, cb, tiles_x, tiles_y, radius_clip, old_box, bt.prev_box, bt.lists, recs);
}

template <int KIND>
__global__ __launch_bounds__(256) void fast_project_fill_kernel(
    int n, float clip_coe, const float2 *__restrict__ means2d, const float *__restrict__ p0,
    const float *__restrict__ p1, float img_w, float img_h, int tiles_x, int tiles_y, float radius_clip,
    float2 *__restrict__ xys, float *__restrict__ depths, int32_t *__restrict__ radii,
    float *__restrict__ conics, int32_t *__restrict__ num_tiles_hit, BinTarget bt) {
    project_fill_one<KIND>(blockIdx.x * blockDim.x + threadIdx.x, n, clip_coe, means2d, p0, p1, img_w, img_h, tiles_x,
                           tiles_y, radius_clip, xys, depths, radii, conics, num_tiles_hit, bt);
}

// -------------------------------------------------------------------------------------- forward
struct FastFwdLds {
    FwdLds f;
};
static_assert(sizeof(float4) * GI2D_FWD_PAIRBUF >= sizeof(int) * GI2D_FAST_C, "the id buffer of the list head overlays the pair buffers");

// WT: packed records and image leave as write-through stores (gi2d_raster_core.h::store16): launches of at most one
// residency round of the chip, whose dirty lines the next kernel of the stream would otherwise wait for.
template <bool WT>
__global__ __launch_bounds__(256) void fast_fwd_kernel(
    int tiles_x, int tiles_y, int img_w, int img_h, RecSets rs, const float *__restrict__ background,
    int32_t *__restrict__ lists, int2 *__restrict__ tile_bins,
    GaussRec *__restrict__ packed, float4 *__restrict__ partial_g, float4 *__restrict__ partial_big,
    int32_t *__restrict__ status, float *__restrict__ final_Ts, int32_t *__restrict__ final_idx,
    float *__restrict__ out_img) {
    __shared__ FastFwdLds sm;
    __shared__ int grp[32];
    int *ids = reinterpret_cast<int *>(sm.f.pairbuf);  // id buffer of the head: dead before the pair buffers are first written
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x;
    const float4 *recs = recs_for_tile_pass(rs, blockIdx.x == 0 && tid == 0);
    if (tid == 0) fwd_stage_dummy(sm.f);
    const float tx0 = (float)(tx * GI2D_TILE), ty0 = (float)(ty * GI2D_TILE);
    const int L = tile_list_head<false>(
        ids, grp, tile, tx, ty, recs, lists, tile_bins, status, [&](int, const BinRec &br) { return br; },
        [&](int rank, int g, const BinRec &br) {
            const GaussRec &r = br.r;
            // the backward kernel finds the entry's partial row by this code; a pool row past the pool's end (the
            // overflow status is raised here) becomes "no row"
            int slot = partial_slot(g, br.box, tx, ty, br.pool);
            float4 *row = partial_row(slot, partial_g, partial_big, tiles_x * tiles_y * GI2D_TILE_LIST_CAP, status);
            if (!row) slot = GI2D_NO_ROW;
            if (rank < GI2D_TILE_LIST_CAP) {
                const AlphaRule ar = alpha_rule(r.gx, r.gy, r.a, r.b, r.c, r.opac);
                const unsigned mask = cull_word_ext(r.gx, r.gy, br.hx, br.hy, tx0, ty0, img_h, ar.clamp);
                fwd_stage_entry(sm.f, rank, r, mask, ar.lim);
                float4 *dst = reinterpret_cast<float4 *>(packed + (size_t)tile * GI2D_TILE_LIST_CAP + rank);
                store16<WT>(dst, make_float4(r.gx, r.gy, r.a, r.b), packed);
                store16<WT>(dst + 1, make_float4(r.c, r.opac, r.cr, r.cg), packed);
                store16<WT>(dst + 2, make_float4(r.cb, __int_as_float(slot), __int_as_float(g), __int_as_float((int)mask)), packed);
            } else if (row) {
                // beyond the 256-entry cap: never rasterized, its gradient row must read as zero
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                row[0] = z;
                row[1] = z;
                row[2] = z;
            }
        }, Inbox{nullptr}, head_row_load(lists, tile, false));
    __syncthreads();
    const int len = L > GI2D_TILE_LIST_CAP ? GI2D_TILE_LIST_CAP : L;
    // "No intersection at all" (image = background) is a global property no single tile can decide: every
    // non-empty tile raises status[0] with a plain store as its LAST memory operation (no barrier waits on
    // it; a contended atomic here would serialise all workgroups) ...
    // ... which the binning step that filled the rows has noted (BinRecs: a gaussian in any tile at all), so the
    // background image of that corner case is written here, not by a second launch
    const bool nothing = background != nullptr && !tile_pass_has_members(rs);
    if (final_idx)
        fwd_rasterize_staged<true, WT>(sm.f, len, list_base(tile), tx, ty, img_w, img_h, nothing, background,
                                       final_Ts, final_idx, out_img);
    else
        fwd_rasterize_staged<false, WT>(sm.f, len, list_base(tile), tx, ty, img_w, img_h, nothing, background,
                                        final_Ts, final_idx, out_img);
    if (tid == 0 && L > 0) status[0] = 1;
}

// rasterize_sum_plus.py:110-118: if there is not a single intersection the image is the background.
__global__ __launch_bounds__(256) void fast_background_kernel(int img_w, int img_h,
                                                              const int32_t *__restrict__ status,
                                                              const float *__restrict__ background,
                                                              float *__restrict__ out_img) {
    if (status[0] > 0) return;
    const size_t npix = (size_t)img_w * img_h;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        out_img[3 * p] = background[0];
        out_img[3 * p + 1] = background[1];
        out_img[3 * p + 2] = background[2];
    }
}

// ------------------------------------------------------------------------------------- backward
template <bool WITH_ABS, bool WT>
__global__ __launch_bounds__(256, WITH_ABS ? 4 : GI2D_BWD_OCC) void fast_bwd_kernel(
    int tiles_x, int tiles_y, int img_w, int img_h, const int2 *__restrict__ tile_bins,
    const GaussRec *__restrict__ packed, const int32_t *__restrict__ final_idx,
    const float *__restrict__ v_output, float4 *__restrict__ partial_g, float4 *__restrict__ partial_big) {
    __shared__ BwdLds<WITH_ABS, false> sm;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x;
    const int2 range = tile_bins[tile];
    const int full_len = range.y - range.x;
    if (full_len <= 0) return;
    const int len = full_len > GI2D_TILE_LIST_CAP ? GI2D_TILE_LIST_CAP : full_len;
    bwd_stage_pixels(sm, tx, ty, img_w, img_h, final_idx, v_output);
    unsigned mask = 0;
    int slot = 0;
    if (tid < len) {
        const float4 *src = reinterpret_cast<const float4 *>(packed + (size_t)tile * GI2D_TILE_LIST_CAP + tid);
        const float4 q0 = src[0], q1 = src[1], q2 = src[2];
        const ConicS cs = scale_conic(q0.z, q0.w, q1.x);
        sm.gA[tid] = make_float4(q0.x, q0.y, cs.ha, cs.hb);
        sm.gB[tid] = make_float4(cs.hc, q1.y, q1.z, q1.w);
        sm.gC[tid] = make_float2(q2.x, __int_as_float((int)alpha_rule(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y).lim));
        if constexpr (WITH_ABS) sm.gRaw[tid] = make_float4(q0.z, q0.w, q1.x, 0.f);
        slot = __float_as_int(q2.y);
        mask = (unsigned)__float_as_int(q2.w);
    }
    float4 *dst = nullptr;
    if (tid < len && slot != GI2D_NO_ROW)  // validated by the forward kernel that wrote the code
        dst = slot >= 0 ? partial_g + GI2D_FAST_ROW * (size_t)slot : partial_big + GI2D_FAST_ROW * (size_t)(-slot - 1);
    bwd_run_tile<WITH_ABS, false, false, WT>(sm, len, mask, range.x, (float)(tx * GI2D_TILE), (float)(ty * GI2D_TILE), dst, 0ull,
                                             nullptr, partial_g);
}

// Empty state of a workspace: every tile row empty, no gaussian binned, identity tile order.
__global__ __launch_bounds__(256) void fast_ws_init_kernel(int num_tiles, int n, int32_t *__restrict__ lists,
                                                           int32_t *__restrict__ tile_order,
                                                           PrevBox *__restrict__ prev_box, int32_t *__restrict__ ver,
                                                           const int32_t *__restrict__ only_if_moved) {
    // only_if_moved: {population after, population before} of a device-side prune (gi2d_densify.hip): the lists refer to
    // gaussian ids, which only change when rows were actually dropped -- one check in a few hundred on a Kodak fit
    if (only_if_moved && (only_if_moved[0] == only_if_moved[1] || only_if_moved[0] == 0)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < GI2D_VER_WORDS) ver[i] = 0;  // the two record-set counters and the any-member stamp (GI2D_VER_ANY)
    if (i == 0) lists[GI2D_POOL_CURSOR] = 0;  // the row pool is empty
    if (i < num_tiles) {
        lists[(size_t)i * GI2D_FAST_LROW] = 0;
        lists[(size_t)i * GI2D_FAST_LROW + 1] = 0;
        tile_order[i] = i;
    }
    // ... and every tile's inbox: 64 bitmap words behind its ids
    for (long long k = i; k < (long long)num_tiles * GI2D_INBOX_WORDS; k += (long long)gridDim.x * blockDim.x)
        inbox_bits_of(lists, (int)(k / GI2D_INBOX_WORDS))[k % GI2D_INBOX_WORDS] = 0;
    if (i < n) prev_box[i] = no_box();
}

// ----------------------------------------------------------------- forward + backward in one pass
// PHASE 0: every tile, general form (256 staged entries: 26 KB of LDS, six workgroups per CU) -- what a launch of at
//          most one residency round of the chip runs (one 768x512 image: 1536 tiles).
// PHASE 1: the SMALL form (GI2D_SMALL_CAP staged entries: 18 KB, 62 registers, EIGHT workgroups per CU) on every tile
//          whose row holds at most that many candidates; a fuller row is only marked (`big_tile`) ...
// PHASE 2: ... and handled by the general form in a second launch whose other workgroups return at once.
// The per-tile code is the same template (fused_tile<MODE, CAP>), so a tile's results do not depend on the phase.
// `mark_big` (batched launches): the general form marks the rows above GI2D_SMALL_CAP too -- batch_pass_end counts them.
// INBOX: the head takes entrants out of the tile's inbox (gi2d_fast_internal.h::Inbox) -- the one launch of one image,
// which is what follows the one kernel that puts any in; every other form is built without that code.
// WTMODE: gradient rows and image leave as write-through stores (gi2d_raster_core.h::store16) -- 0 never (batches, large
// images: their stores overlap other tiles' work and compete for the memory system's slots: +28 % at K = 24), 1 always
// (the INBOX instantiation: one image of at most one residency round), 2 when `wt` says so (the general-form kernel of a
// single image: the launch code knows the tile count).
template <int MODE, int PHASE, bool INBOX = false, int WTMODE = 0>
__device__ __forceinline__ void tile_pass_workgroup(const TilePassArgs &a, int slot, bool first, bool mark_big = false,
                                                    bool wt = false) {
    constexpr int CAP = PHASE == 1 ? GI2D_SMALL_CAP : GI2D_TILE_LIST_CAP;
    __shared__ FusedLdsT<CAP> sm;
    int tile;
    HeadRow hr;
    if (PHASE == 2) {  // (the caller has looked at big_tile: only tiles the small form passed over arrive here)
        tile = __builtin_amdgcn_readfirstlane(a.tile_order[slot]);
        hr = head_row_load(a.lists, tile, INBOX);
    } else {
        hr = head_row_for(a.lists, a.tile_order, slot, tile, INBOX);
    }
    // (ahead of the early return below: the launch's first workgroup records which record set this pass reads --
    // RecSets::ver[0] -- whether or not its own tile is left to phase 2; a pipelined end-of-step kernel reads that word)
    const float4 *recs = recs_for_tile_pass(a.rs, first && threadIdx.x == 0);
    Inbox ib;
    ib.recs = a.inbox;
    if (PHASE == 1 || (PHASE == 0 && mark_big)) {
        const bool big = __builtin_amdgcn_readfirstlane(hr.hdr_count) > GI2D_SMALL_CAP;  // workgroup-uniform
        if (threadIdx.x == 0) a.big_tile[tile] = big ? 1 : 0;
        if (PHASE == 1 && big) return;
    }
    // (phase 2 loops over tiles: its loop keeps the lane's invariants alive, so the forward's trips are not unrolled there)
    if (WTMODE == 1 || (WTMODE == 2 && wt))  // workgroup-uniform (launch-uniform)
        fused_tile<MODE, CAP, PHASE == 2 ? 1 : GI2D_FWD_UNROLL, INBOX, WTMODE != 0>(
            sm, tile, a.tiles_x, a.tiles_y, a.img_w, a.img_h, recs, a.lists, a.tile_bins, a.partial_g, a.partial_big, a.status,
            a.out_img, a.vsrc, a.grad_scale, a.tile_sse, hr, ib);
    else
        fused_tile<MODE, CAP, PHASE == 2 ? 1 : GI2D_FWD_UNROLL, INBOX, false>(
            sm, tile, a.tiles_x, a.tiles_y, a.img_w, a.img_h, recs, a.lists, a.tile_bins, a.partial_g, a.partial_big, a.status,
            a.out_img, a.vsrc, a.grad_scale, a.tile_sse, hr, ib);
}

// Phase 2 finds nothing to do on the scenes the two-phase form is for (large images: sparse rows), and ten thousand
// workgroups that only look and return cost several us of dispatch: a phase-2 workgroup therefore looks at a STRIP of
// GI2D_PHASE2_STRIP slots at once (one per lane of a wave; strided over the launch) and handles the ones that were
// passed over one after the other (a barrier between two tiles: they share its LDS).  64 instead of round 4's 8: the
// launch that finds nothing is 170 workgroups at 2040x1356 instead of 1360.
#ifndef GI2D_PHASE2_STRIP
#define GI2D_PHASE2_STRIP 64
#endif
// a single image of more tiles than one residency round of the general form may run as two launches (see the launch code)
#ifndef GI2D_TWO_PHASE_TILES
#define GI2D_TWO_PHASE_TILES (256 * GI2D_FUSED_OCC) /* development aid: a huge value keeps every launch single-phase */
#endif
static_assert(GI2D_INBOX_MAX_TILES <= GI2D_TWO_PHASE_TILES, "the inboxes are for launches that run the general form throughout");
// workgroups per CU the register allocator leaves room for, by phase: the loop of phase 2 keeps the lane's invariants
// alive across tiles (80 registers and a few dwords of scratch at six per CU -- and a kernel with ANY scratch pays
// ~200 us per dispatch here while the runtime re-arms the queue's scratch: measured) -- so it is built for four (five still left one of the four kernels with 12 bytes of it)
#define GI2D_PHASE_OCC(PHASE) ((PHASE) == 1 ? GI2D_SMALL_OCC : (PHASE) == 2 ? 4 : GI2D_FUSED_OCC)
// `busy`: the last pass that reported found tiles for this launch -- a strip of 8 then (eight times the workgroups, each
// with fewer tiles to handle one after the other: on a Kodak batch the second launch served ~4 tiles per workgroup in
// turn at 64).  The kernels derive the strip from the grid.
#define GI2D_PHASE2_STRIP_BUSY 8 /* measured on the Kodak leg: 1.41-1.43 images/s at 64, 1.45 at 16 and 8, 1.44 at 2 */
static inline unsigned phase2_blocks(long long slots, bool busy = false) {
    const int strip = busy ? GI2D_PHASE2_STRIP_BUSY : GI2D_PHASE2_STRIP;
    return (unsigned)((slots + strip - 1) / strip);
}
__device__ __forceinline__ int phase2_strip(int slots) { return (slots + (int)gridDim.x - 1) / (int)gridDim.x; }

// INBOX: the launch right behind an update kernel that delivers through the inboxes (gi2d_train_steps, iterations 2 ...
// of a call on one image of at most GI2D_INBOX_MAX_TILES tiles); every other launch runs the kernel built without.
template <int MODE, int PHASE, bool INBOX = false>
__global__ __launch_bounds__(256, GI2D_PHASE_OCC(PHASE)) void fast_fwdbwd_kernel(TilePassArgs a) {
    static_assert(PHASE == 0 || !INBOX, "the small form has no room for the entrants' bookkeeping");
    if (PHASE == 2) {
        // which slots of the strip were passed over: lane i looks at slot i (ONE round of two dependent loads for the
        // whole strip -- slot by slot it was sixteen), a ballot makes the answer scalar
        // (the strip is STRIDED -- lane i looks at slot blockIdx + i * gridDim -- so that a run of neighbouring fuller
        // tiles is dealt to as many workgroups as it has tiles instead of queueing up in one)
        const int tiles = a.tiles_x * a.tiles_y, lane = threadIdx.x & 63;
        const int s0 = (int)blockIdx.x, stride = (int)gridDim.x, mine = s0 + lane * stride;
        const bool big = lane < phase2_strip(tiles) && mine < tiles && a.big_tile[a.tile_order[mine]] != 0;
        unsigned long long todo = __ballot(big);
        while (todo) {  // workgroup-uniform: every wave computed the same mask
            const int slot = s0 + __builtin_ctzll(todo) * stride;
            todo &= todo - 1;
            tile_pass_workgroup<MODE, PHASE>(a, slot, false);
            __syncthreads();  // the next tile stages into the same LDS
        }
    } else {
        // (a large image in the general form marks its fuller tiles too: single_pass_end counts them)
        // (written through where the whole launch is one residency round of the chip: see tile_pass_workgroup)
        const int tiles = a.tiles_x * a.tiles_y;
        // (the small form of a single LARGE image -- phase 1 of two launches at 2040x1356 -- measured with write-through
        // rows, image and both: 54.97 -> 55.6-55.7 us; seven residency rounds overlap their stores already)
        tile_pass_workgroup<MODE, PHASE, INBOX, INBOX ? 1 : (PHASE == 0 ? 2 : 0)>(
            a, (int)blockIdx.x, blockIdx.x == 0, tiles > GI2D_TWO_PHASE_TILES, a.write_through != 0);
    }
}

// The same pass over the tiles of K images in ONE launch (gi2d_batch.h): workgroup b belongs to image k with
// tile_start[k] <= b < tile_start[k + 1] and handles that image's tile tile_order[b - tile_start[k]]; everything else
// is the single-image kernel's code, so every image's results are those of its own launch bit for bit.
__device__ __forceinline__ void batched_slot(int b, const int *__restrict__ tile_start, int k_images, int uniform_tiles,
                                             int xcd_map, int &k, int &local) {
    if (b < xcd_map * uniform_tiles) {
        // Images with the same tile count: the first 8 * floor(K / 8) of them (`xcd_map` images) go to the XCDs whole --
        // image k's tiles to the workgroups b with b % 8 == k % 8.  Workgroups are dealt to the eight XCDs round-robin, so
        // one image's records, tile rows and gradient rows stay in ONE XCD's 4 MiB L2 instead of being fetched into all
        // eight (placement is a speed choice only: any mapping is correct).  The K % 8 images left over follow in plain
        // order, spread over all XCDs: giving them an XCD each would leave the other XCDs idle for a whole image (12 images
        // then took the time of 16).
        const int x = b & 7, j = b >> 3;
        const int slot = j / uniform_tiles;
        k = slot * 8 + x;
        local = j - slot * uniform_tiles;
    } else if (uniform_tiles > 0) {
        k = b / uniform_tiles;
        local = b - k * uniform_tiles;
    } else {
        k = batch_find(tile_start, k_images, b);
        local = b - tile_start[k];
    }
    // image and tile are the same for the whole workgroup: say so (the integer divisions above leave them in vector
    // registers, and everything derived from them -- the argument block's loads, the tile's row and column -- would
    // follow: 4 more VGPRs than the single-image kernel, i.e. spills at this kernel's 80-register budget)
    k = __builtin_amdgcn_readfirstlane(k);
    local = __builtin_amdgcn_readfirstlane(local);
}
// The lane's own (image, slot) of workgroup index b -- phase 2 looks at a strip of them at once.
__device__ __forceinline__ void batched_slot_of_lane(int b, const int *__restrict__ tile_start, int k_images,
                                                     int uniform_tiles, int xcd_map, int &k, int &local) {
    if (b < xcd_map * uniform_tiles) {
        const int x = b & 7, j = b >> 3, slot = j / uniform_tiles;
        k = slot * 8 + x;
        local = j - slot * uniform_tiles;
    } else if (uniform_tiles > 0) {
        k = b / uniform_tiles;
        local = b - k * uniform_tiles;
    } else {
        int lo = 0, hi = k_images;  // tile_start[lo] <= b < tile_start[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (tile_start[mid] <= b) lo = mid; else hi = mid;
        }
        k = lo;
        local = b - tile_start[lo];
    }
}
// PHASE as in fast_fwdbwd_kernel.  Measured at K = 8 / 24 of 768x512 images: the small form is 5 ... 8 % faster per
// tile it serves, but a scene whose fuller tiles sit above GI2D_SMALL_CAP (9 % of the tiles of tools/batch_time.py's
// scenes, half of a trained Kodak scene's) hands those to a second launch at lower occupancy (net -6 % at K = 24), so
// the two-phase form is used on a batch only while few rows of it are that full (GI2D_TWO_PHASE_BIG_DIV) -- batch_pass_begin.
template <int MODE, int PHASE>
__global__ __launch_bounds__(256, GI2D_PHASE_OCC(PHASE)) void fast_fwdbwd_batched_kernel(
    const BatchImage *__restrict__ imgs, const BatchHead *__restrict__ head, int k_images, int uniform_tiles,
    int xcd_map) {
    const int *tile_start = head->tile_start;
    int k, local;
    if (PHASE == 2) {
        const int total = uniform_tiles > 0 ? k_images * uniform_tiles : tile_start[k_images], lane = threadIdx.x & 63;
        const int s0 = (int)blockIdx.x, stride = (int)gridDim.x, mine = s0 + lane * stride;  // (strided: see above)
        bool big = false;
        if (lane < phase2_strip(total) && mine < total) {
            batched_slot_of_lane(mine, tile_start, k_images, uniform_tiles, xcd_map, k, local);
            const TilePassArgs &t = imgs[k].t;
            big = t.big_tile[t.tile_order[local]] != 0;
        }
        unsigned long long todo = __ballot(big);
        while (todo) {
            const int b = s0 + __builtin_ctzll(todo) * stride;
            todo &= todo - 1;
            batched_slot(b, tile_start, k_images, uniform_tiles, xcd_map, k, local);
            tile_pass_workgroup<MODE, PHASE>(imgs[k].t, local, false);
            __syncthreads();
        }
    } else {
        batched_slot((int)blockIdx.x, tile_start, k_images, uniform_tiles, xcd_map, k, local);
        // (a SMALL batch in the general form -- K = 3 / 4: three to four residency rounds -- measured with write-through
        // stores: K = 4 12.95 -> 13.75 us per image, the 12-image Kodak shard 1.39 -> 1.31 images/s: only a launch of ONE
        // residency round, whose stores all fall into its last third, gains from them)
        tile_pass_workgroup<MODE, PHASE>(imgs[k].t, local, local == 0, true);
    }
}

// How many tiles of the batch the last tile pass marked as too full for the small form: one number per call, read back
// behind the call's kernels (batch_pass_end).  head->big_seen is zero when the call starts (write_batch_table).
__global__ __launch_bounds__(256) void batch_count_big_kernel(const BatchImage *__restrict__ imgs,
                                                              BatchHead *__restrict__ head, int k_images) {
    const int total = head->tile_start[k_images];
    int mine = 0;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < total; b += gridDim.x * 256) {
        int lo = 0, hi = k_images;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (head->tile_start[mid] <= b) lo = mid; else hi = mid;
        }
        mine += imgs[lo].t.big_tile[b - head->tile_start[lo]] != 0;
    }
    const int upto = wave_inclusive_scan(mine);
    if ((threadIdx.x & 63) == 63 && upto) atomicAdd(&head->big_seen, upto);
}

// the same for one image; *out is zero when the kernel starts (single_pass_end clears it in front)
__global__ __launch_bounds__(256) void count_big_kernel(const int32_t *__restrict__ big_tile, int tiles,
                                                        int32_t *__restrict__ out) {
    int mine = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < tiles; t += gridDim.x * 256) mine += big_tile[t] != 0;
    const int upto = wave_inclusive_scan(mine);
    if ((threadIdx.x & 63) == 63 && upto) atomicAdd(out, upto);
}

// --------------------------------------------------------------------------------------- reduce
__global__ __launch_bounds__(256) void fast_reduce_kernel(
    int n, const PrevBox *__restrict__ prev_box, int tiles_x, int tiles_y, const int32_t *__restrict__ gids_sorted,
    const int2 *__restrict__ tile_bins, const float4 *__restrict__ partial_g, const float4 *__restrict__ partial_big,
    float2 *__restrict__ v_xy, float *__restrict__ v_conic, float *__restrict__ v_rgb, float *__restrict__ v_opacity,
    float4 *__restrict__ v_abs_xy) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    float acc[11];
    const PrevBox pb = g < n ? prev_box[g] : no_box();
    reduce_one(g, make_int2(pb.x, pb.y), pb.z, tiles_x * tiles_y * GI2D_TILE_LIST_CAP, partial_g, partial_big, acc);
    if (g < n) store_grads(g, acc, v_xy, v_conic, v_rgb, v_opacity, v_abs_xy);
}

// What the reduce kernel needs to project and bin the gaussians for the NEXT step of a loop over unchanged or
// caller-updated inputs (FILL_NEXT): the step then starts directly with its tile pass.
struct NextProject {
    float clip_coe;
    const float2 *means2d;
    float *depths;
    int32_t *num_tiles_hit;
    BinTarget bt;
    int32_t *tile_order;  // non-null: one extra workgroup re-balances the next tile pass (large populations only,
                          // where this kernel is long enough to hide it)
};

// FILL_NEXT: the launch holds two kinds of workgroups (plus the tile-ordering one) -- the first `role_blocks` finish
// step i (gradient reduce + projection backward, from the records the tile pass used), the next `role_blocks` start
// step i + 1 (projection + binning step + records into the other set): the two do not depend on each other, and a
// one-lane-per-gaussian kernel this short is all dependent latency, so side by side they take the time of the longer one
// instead of the sum.  The backward lanes therefore take conic / radius / tile box from the RECORD, never from the
// xys / radii / conics arrays the projection lanes overwrite in the same launch.
template <int KIND, bool FILL_NEXT>
__global__ __launch_bounds__(256) void fast_reduce_project_kernel(
    int n, int role_blocks, float2 *xys, int32_t *radii, float *conics, int tiles_x, int tiles_y, float radius_clip,
    PrevBox *prev_box, const int32_t *__restrict__ gids_sorted, const int2 *__restrict__ tile_bins,
    const float4 *__restrict__ partial_g, const float4 *__restrict__ partial_big, const float *__restrict__ p0,
    const float *__restrict__ p1, float img_w, float img_h, float2 *__restrict__ v_xy, float *__restrict__ v_conic,
    float *__restrict__ v_rgb, float *__restrict__ v_opacity, float4 *__restrict__ v_abs_xy,
    float *__restrict__ v_cov2d, float2 *__restrict__ v_mean2d, float *__restrict__ v_p0, float *__restrict__ v_p1,
    NextProject next) {
    int block = blockIdx.x;
    if (FILL_NEXT) {
        if (block >= 2 * role_blocks) {
            compute_tile_order(tile_bins, tiles_x * tiles_y, next.tile_order);  // the extra workgroup (see there)
            return;
        }
        if (block >= role_blocks) {
            project_fill_one<KIND>((block - role_blocks) * blockDim.x + threadIdx.x, n, next.clip_coe, next.means2d, p0,
                                   p1, img_w, img_h, tiles_x, tiles_y, radius_clip, xys, next.depths, radii, conics,
                                   next.num_tiles_hit, next.bt);
            return;
        }
    }
    const int g = block * blockDim.x + threadIdx.x;
    int2 box = make_int2(0, 0);
    int pool = -1;
    float conic[3] = {0.f, 0.f, 0.f};
    int radius = 0;
    if (g < n) {
        if (FILL_NEXT) {
            const float4 *rec = recs_of_last_pass(next.bt.recs) + 4 * (size_t)g;
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
            conic[0] = q0.z, conic[1] = q0.w, conic[2] = q1.x;
            box = make_int2(__float_as_int(q2.w), __float_as_int(q3.x));
            radius = __float_as_int(q3.y);
            pool = __float_as_int(q3.z);
        } else {
            const PrevBox pb = prev_box[g];
            box = make_int2(pb.x, pb.y), pool = pb.z;
            conic[0] = conics[3 * g], conic[1] = conics[3 * g + 1], conic[2] = conics[3 * g + 2];
            radius = radii[g];
        }
    }
    // the projection parameters are requested together with the record, not after the gradient sum they will meet
    float par[3] = {0.f, 0.f, 0.f};
    if (g < n) {
        if (KIND == kScaleRot)
            par[0] = p0[2 * g], par[1] = p0[2 * g + 1], par[2] = p1[g];
        else if (KIND == kCholesky)
            par[0] = p0[3 * g], par[1] = p0[3 * g + 1], par[2] = p0[3 * g + 2];
    }
    float acc[11];
    reduce_one(g, box, pool, tiles_x * tiles_y * GI2D_TILE_LIST_CAP, partial_g, partial_big, acc);
    if (g >= n) return;
    store_grads(g, acc, v_xy, v_conic, v_rgb, v_opacity, v_abs_xy);
    ProjGrad r;
    r.g11 = r.g12 = r.g22 = r.o0 = r.o1 = r.o2 = 0.f;
    r.v_mean = make_float2(0.f, 0.f);
    if (radius > 0) {
        const float vc[3] = {acc[2], acc[3], acc[4]};
        r = project_bwd_one<KIND>(0, par, par + 2, img_w, img_h, conic, make_float2(acc[0], acc[1]), vc);
    }
    store_proj_grad(g, KIND == kScaleRot, r, v_cov2d, v_mean2d, v_p0, v_p1);
}

static int check_ws(const char *what, void *ws, size_t ws_bytes, int n, int tiles_x, int tiles_y) {
    if (n < 0 || tiles_x < 0 || tiles_y < 0) {
        set_error("fast path: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if ((long long)tiles_x * tiles_y * GI2D_FAST_LROW > 0x7fffffffLL || (long long)n * GI2D_FAST_S > 0x7fffffffLL ||
        tiles_x > 0xffff || tiles_y > 0xffff) {
        set_error("fast path: problem too large for 32-bit slot indices");
        return GI2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < carve_fast(nullptr, n, tiles_x * tiles_y).bytes) {
        set_error(what);
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    return GI2D_OK;
}

}  // namespace gi2d

using namespace gi2d;

// Kernel timer (bench.py): start/stop events attached to ONE dispatch with hipExtLaunchKernelGGL read the
// kernel's own begin/end timestamps -- what rocprofv3's kernel trace reports -- instead of the span between
// two stream markers, which also contains marker processing and dispatch latency.
struct KernelTimer {
    hipEvent_t begin, end;
};
static thread_local KernelTimer *g_armed_timer = nullptr;
// gi2d_timer_arm_many: timers for every `stride`-th of the following tile-pass launches of this thread
static thread_local struct {
    KernelTimer **list;
    int count, stride, seen;
} g_timer_plan = {nullptr, 0, 1, 0};
static KernelTimer *next_timer() {
    KernelTimer *t = g_armed_timer;
    g_armed_timer = nullptr;
    if (!t && g_timer_plan.list) {
        const int i = g_timer_plan.seen++;
        if (i % g_timer_plan.stride == 0) {
            t = g_timer_plan.list[i / g_timer_plan.stride];
            if (i / g_timer_plan.stride + 1 >= g_timer_plan.count) g_timer_plan.list = nullptr;  // the last one
        }
    }
    return t;
}
#define GI2D_LAUNCH_TIMED(kernel, grid, block, stream, ...)                                               \
    do {                                                                                                  \
        if (KernelTimer *gi2d_t = next_timer()) {                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, gi2d_t->begin, gi2d_t->end, 0,          \
                                  __VA_ARGS__);                                                           \
        } else {                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                              \
        }                                                                                                 \
    } while (0)

namespace gi2d {
// A single-image launch of more tiles than the chip holds at once (six general-form workgroups per CU x 256 CUs) runs in
// two phases: the small form on every tile it can serve, at eight workgroups per CU, then the general form on the rest
// (2040x1356 at 50 000 gaussians: tile pass 70.2 -> 64.6 us).  One image of up to 1536 tiles (768x512) is a single
// residency round either way and stays one launch; so do batched launches (see fast_fwdbwd_batched_kernel).
static inline bool two_phase_tile_pass(long long tiles) { return tiles > GI2D_TWO_PHASE_TILES; }
// one launch with an optional start / stop event (the two phases of a timed tile pass carry one event each)
template <class K, class... A>
static void launch_between(K kernel, dim3 grid, dim3 block, hipStream_t st, hipEvent_t begin, hipEvent_t end, A... args) {
    if (begin || end)
        hipExtLaunchKernelGGL(kernel, grid, block, 0, st, begin, end, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
}
int launch_tile_pass_batched(int mode, const BatchTable &b, int k_images, int total_blocks, int uniform_tiles,
                             int form, hipStream_t st) {
    if (total_blocks <= 0) return GI2D_OK;
    int xcd_map = 0;  // images placed on the XCDs whole (see the kernel): a multiple of 8
    if (uniform_tiles > 0) xcd_map = k_images & ~7;
    const dim3 grid((unsigned)total_blocks), block(256);
    const BatchImage *imgs = (const BatchImage *)b.img;
    if (form != 0) {
        KernelTimer *tm = next_timer();
        const dim3 grid2(phase2_blocks(total_blocks, form == 2));
        if (mode == 0) {
            launch_between((fast_fwdbwd_batched_kernel<0, 1>), grid, block, st, tm ? tm->begin : nullptr, nullptr, imgs,
                           b.head, k_images, uniform_tiles, xcd_map);
            launch_between((fast_fwdbwd_batched_kernel<0, 2>), grid2, block, st, nullptr, tm ? tm->end : nullptr, imgs,
                           b.head, k_images, uniform_tiles, xcd_map);
        } else {
            launch_between((fast_fwdbwd_batched_kernel<1, 1>), grid, block, st, tm ? tm->begin : nullptr, nullptr, imgs,
                           b.head, k_images, uniform_tiles, xcd_map);
            launch_between((fast_fwdbwd_batched_kernel<1, 2>), grid2, block, st, nullptr, tm ? tm->end : nullptr, imgs,
                           b.head, k_images, uniform_tiles, xcd_map);
        }
    } else if (mode == 0) {
        GI2D_LAUNCH_TIMED((fast_fwdbwd_batched_kernel<0, 0>), grid, block, st, imgs, b.head, k_images, uniform_tiles,
                          xcd_map);
    } else {
        GI2D_LAUNCH_TIMED((fast_fwdbwd_batched_kernel<1, 0>), grid, block, st, imgs, b.head, k_images, uniform_tiles,
                          xcd_map);
    }
    return check_launch("batched tile pass");
}

// ---- which form the next call's tile passes take
// The two-launch form pays only while NO tile is fuller than GI2D_SMALL_CAP: its second launch is as long as one tile's
// whole dependent chain (~15 us) the moment it has a single tile to serve, whatever the size of the first (measured on
// Kodak batches: 8 us for three tiles of 4608, 17 us next to another stream's launch), against 5 ... 10 % of the first.
// So every pass marks its fuller tiles (`big_tile`), a call that may use the form ends with a counting kernel and an
// asynchronous 4-byte copy to pinned memory, and the next call on the same key -- a batch table, or a single image's
// workspace -- looks at the number if its event has fired.  Nothing waits; a stale or missing answer costs speed for
// one call, never correctness (the second launch serves whatever the first passed over).
// One record per key the process has used (a device address; a freed buffer's record is reused by whatever is allocated
// there next).
struct PassHint {
    int *host = nullptr;  // pinned: where the report of the last call lands
    hipEvent_t landed = nullptr;
    bool pending = false, discard = false;  // discard: the report in flight is about an earlier owner of the address
    long long total = 0, min_total = 0;  // tiles when that report was queued; smallest launch the form is used on
    int last = -1;        // tiles above GI2D_SMALL_CAP in the last pass of the last call that reported; -1: nothing known
};
static std::mutex g_hint_mu;
static std::unordered_map<const void *, PassHint> g_hints;
// GI2D_BATCH_TILE_PASS = general | two-phase | auto (default): tests and measurements force either form (batched
// launches and single images of more than GI2D_TWO_PHASE_TILES tiles alike)
static int pass_form_override() {
    static const int v = [] {
        const char *e = std::getenv("GI2D_BATCH_TILE_PASS");
        if (!e) return -1;
        return !std::strcmp(e, "two-phase") ? 1 : !std::strcmp(e, "general") ? 0 : -1;
    }();
    return v;
}
static bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
static void hint_poll(PassHint &h) {
    if (h.pending && hipEventQuery(h.landed) == hipSuccess) {
        if (!h.discard) h.last = *h.host;
        h.pending = h.discard = false;
    }
}
// smallest launch the form is used on: below it the small form's higher occupancy has nothing to fill (one 768x512 image
// is six workgroups per CU in either form) and the second launch's look at every slot costs what the first saves
#define GI2D_BATCH_TWO_PHASE_MIN (8 * GI2D_TWO_PHASE_TILES) /* measured: eight 768x512 images -6 %; one image per
                                                               launch on three streams +40 % in time (a Kodak shard
                                                               of three images 1.10 -> 0.78 images/s) */
// Two launches while at most one tile in GI2D_TWO_PHASE_BIG_DIV was too full for the small form in the last pass that
// reported.  Measured on the Kodak leg (24 images, 50 000 iterations, three batches of eight): 1.376 / 1.396 images/s
// with "none at all" (round 4's rule), 1.408-1.413 with 1/64, 1/16 and 1/6 alike -- a fit spends a stretch of its
// schedule with a handful of crowded tiles per image; tools/batch_time.py's scenes (9 % of the tiles too full) lose 6 %
// in two launches with round 4's second launch, hence not more than 1/16 then; with the second launch shaped for work
// (GI2D_PHASE2_STRIP_BUSY) 1/8 and 1/4 measure the same as 1/16 on both, 1/2 loses 1 % on the Kodak leg.
#define GI2D_TWO_PHASE_BIG_DIV 16
// 0: one launch; 1: two launches, nothing expected for the second; 2: two launches, fuller tiles expected
static int hint_says_two_phase(const PassHint &h, long long total, long long min_total) {
    if (!(h.last >= 0 && (long long)h.last * GI2D_TWO_PHASE_BIG_DIV <= total && total >= min_total)) return 0;
    return h.last > 0 ? 2 : 1;
}
static int pass_form_begin(const void *key, long long total, long long min_total, hipStream_t st) {
    const int forced = pass_form_override();
    if (forced >= 0) return forced == 1 && total >= 1 ? 1 : 0;
    if (total < min_total || stream_is_capturing(st)) return 0;  // a captured call takes the form that is never slow
    std::lock_guard<std::mutex> lock(g_hint_mu);
    PassHint &h = g_hints[key];
    hint_poll(h);
    return hint_says_two_phase(h, total, min_total);
}
// `count_word`: device word that holds the number of marked tiles once everything queued on `st` so far has run
static void pass_form_end(const void *key, const int *count_word, long long total, long long min_total,
                          hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_hint_mu);
    if (g_hints.size() > 256 && !g_hints.count(key)) {  // buffers come and go: forget those with nothing in flight
        for (auto it = g_hints.begin(); it != g_hints.end();) {
            PassHint &old = it->second;
            if (old.pending && hipEventQuery(old.landed) != hipSuccess) {
                ++it;
                continue;
            }
            if (old.landed) (void)hipEventDestroy(old.landed);
            if (old.host) (void)hipHostFree(old.host);
            it = g_hints.erase(it);
        }
    }
    PassHint &h = g_hints[key];
    if (h.pending) {  // the previous report has not been looked at: is it there by now?
        hint_poll(h);
        if (h.pending) return;  // still in flight: its buffer is not ours to overwrite yet
    }
    if (!h.host && (hipHostMalloc((void **)&h.host, sizeof(int), hipHostMallocDefault) != hipSuccess ||
                    hipEventCreateWithFlags(&h.landed, hipEventDisableTiming) != hipSuccess)) {
        (void)hipGetLastError();  // no hint, no two-launch passes: nothing else depends on it
        h.host = nullptr;
        return;
    }
    if (hipMemcpyAsync(h.host, count_word, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
        hipEventRecord(h.landed, st) == hipSuccess)
        h.pending = true, h.total = total, h.min_total = min_total;
    else
        (void)hipGetLastError();
}
static bool hint_wanted(long long total, long long min_total, hipStream_t st) {
    return pass_form_override() < 0 && total >= min_total && !stream_is_capturing(st);
}

int batch_pass_begin(const void *batch, int total_blocks, hipStream_t st) {
    return pass_form_begin(batch, total_blocks, GI2D_BATCH_TWO_PHASE_MIN, st);
}
void batch_pass_end(const void *batch, const BatchTable &b, int k_images, int total_blocks, hipStream_t st) {
    if (!hint_wanted(total_blocks, GI2D_BATCH_TWO_PHASE_MIN, st)) return;
    const int blocks = (total_blocks + 255) / 256;
    hipLaunchKernelGGL(batch_count_big_kernel, dim3((unsigned)(blocks < 64 ? blocks : 64)), dim3(256), 0, st,
                       (const BatchImage *)b.img, b.head, k_images);
    pass_form_end(batch, &b.head->big_seen, total_blocks, GI2D_BATCH_TWO_PHASE_MIN, st);
}
// A single image of more than GI2D_TWO_PHASE_TILES tiles, fitted by gi2d_train_steps: key = its workspace, the count
// lives in a spare word next to the record-set counters.
#define GI2D_WS_BIG_COUNT 16 /* word of FastWs::ver */
int single_pass_begin(const void *ws, long long tiles, hipStream_t st) {
    return pass_form_begin(ws, tiles, GI2D_TWO_PHASE_TILES + 1, st);
}
void single_pass_end(const void *ws, const FastWs &w, long long tiles, hipStream_t st) {
    if (!hint_wanted(tiles, GI2D_TWO_PHASE_TILES + 1, st)) return;
    const int blocks = (int)((tiles + 255) / 256);
    if (hipMemsetAsync(w.ver + GI2D_WS_BIG_COUNT, 0, sizeof(int32_t), st) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    hipLaunchKernelGGL(count_big_kernel, dim3((unsigned)(blocks < 64 ? blocks : 64)), dim3(256), 0, st, w.big_tile,
                       (int)tiles, w.ver + GI2D_WS_BIG_COUNT);
    pass_form_end(ws, w.ver + GI2D_WS_BIG_COUNT, tiles, GI2D_TWO_PHASE_TILES + 1, st);
}
}  // namespace gi2d

namespace gi2d {
// gi2d_fast_workspace_init without its argument checks; `only_if_moved`: see fast_ws_init_kernel
int launch_workspace_init(void *ws, int n, int tiles_x, int tiles_y, const int32_t *only_if_moved, gi2d_stream_t st) {
    if (!only_if_moved) {  // a workspace that starts over: what an earlier owner of this address reported is not about it
        std::lock_guard<std::mutex> lock(g_hint_mu);
        const auto it = g_hints.find(ws);
        if (it != g_hints.end()) it->second.last = -1, it->second.discard = it->second.pending;
    }
    FastWs w = carve_fast(ws, n, tiles_x * tiles_y);
    const int t = tiles_x * tiles_y;
    const int work = t > n ? t : n;
    if (work == 0) return GI2D_OK;
    hipLaunchKernelGGL(fast_ws_init_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)st, t, n,
                       w.lists, w.tile_order, w.prev_box, w.ver, only_if_moved);
    return check_launch("fast workspace init");
}
}  // namespace gi2d

extern "C" {

int gi2d_timer_create(void **timer) {
    if (!timer) return GI2D_ERR_INVALID_ARGUMENT;
    KernelTimer *t = new KernelTimer;
    hipError_t e = hipEventCreate(&t->begin);
    if (e == hipSuccess) e = hipEventCreate(&t->end);
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        delete t;
        return (int)e;
    }
    *timer = t;
    return GI2D_OK;
}
int gi2d_timer_destroy(void *timer) {
    KernelTimer *t = (KernelTimer *)timer;
    if (!t) return GI2D_OK;
    if (g_armed_timer == t) g_armed_timer = nullptr;
    // a plan that still holds this timer must not hand its (about to be destroyed) events to a later launch
    if (g_timer_plan.list)
        for (int i = 0; i < g_timer_plan.count; ++i)
            if (g_timer_plan.list[i] == t) {
                g_timer_plan.list = nullptr;
                break;
            }
    (void)hipEventDestroy(t->begin);
    (void)hipEventDestroy(t->end);
    delete t;
    return GI2D_OK;
}
int gi2d_timer_arm(void *timer) {
    g_armed_timer = (KernelTimer *)timer;
    return GI2D_OK;
}
int gi2d_timer_arm_many(void **timers, int count, int stride) {
    if (count < 0 || stride < 1 || (count > 0 && !timers)) {
        set_error("timer arm many: bad argument");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    g_timer_plan.list = count > 0 ? (KernelTimer **)timers : nullptr;
    g_timer_plan.count = count;
    g_timer_plan.stride = stride;
    g_timer_plan.seen = 0;
    return GI2D_OK;
}
int gi2d_timer_elapsed_us(void *timer, float *us) {
    KernelTimer *t = (KernelTimer *)timer;
    if (!t || !us) return GI2D_ERR_INVALID_ARGUMENT;
    hipError_t e = hipEventSynchronize(t->end);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t->begin, t->end);
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        return (int)e;
    }
    *us = ms * 1e3f;
    return GI2D_OK;
}

size_t gi2d_fast_workspace_bytes(int n, int tiles_x, int tiles_y) {
    return carve_fast(nullptr, n, tiles_x * tiles_y).bytes;
}
int gi2d_fast_tile_capacity(void) { return GI2D_FAST_C; }

int gi2d_fast_workspace_init(void *ws, size_t ws_bytes, int n, int tiles_x, int tiles_y, gi2d_stream_t st) {
    int rc = check_ws("fast workspace init: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    return gi2d::launch_workspace_init(ws, n, tiles_x, tiles_y, nullptr, st);
}

static BinTarget bin_target(const FastWs &w, int n, const float *colors, const float *opac, int32_t *status) {
    BinTarget bt;
    bt.colors = colors;
    bt.opacities = opac;
    bt.prev_box = w.prev_box;
    bt.lists = w.lists;
    bt.recs = rec_sets(w, n);
    bt.status = status;
    return bt;
}

int gi2d_fast_bin(int n, const float *xys, const int32_t *radii, const float *conics, const float *colors,
                  const float *opac, int tiles_x, int tiles_y, float radius_clip, void *ws, size_t ws_bytes,
                  int32_t *status, gi2d_stream_t st) {
    int rc = check_ws("fast bin: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    if (!status || (n > 0 && (!xys || !radii || !conics || !colors || !opac))) {
        set_error("fast bin: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    FastWs w = carve_fast(ws, n, tiles_x * tiles_y);
    const int fbs = per_gaussian_block(n);
    hipLaunchKernelGGL(fast_fill_kernel, dim3((n + fbs - 1) / fbs > 0 ? (n + fbs - 1) / fbs : 1), dim3(fbs), 0,
                       (hipStream_t)st, n, (const float2 *)xys, radii, conics, colors, opac, tiles_x, tiles_y,
                       radius_clip, w.prev_box, w.lists, rec_sets(w, n), status);
    return check_launch("fast bin");
}

int gi2d_fast_project_bin(int kind, int n, float clip_coe, const float *means2d, const float *p0,
                          const float *p1, const float *colors, const float *opac, unsigned h, unsigned w_,
                          int tiles_x, int tiles_y, float radius_clip, float *xys, float *depths, int32_t *radii,
                          float *conics, int32_t *nth, void *ws, size_t ws_bytes, int32_t *status,
                          gi2d_stream_t st) {
    int rc = check_ws("fast project+bin: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    if (kind < 0 || kind > 2 || !status ||
        (n > 0 && (!means2d || !p0 || !colors || !opac || !xys || !depths || !radii || !conics || !nth ||
                   (kind == 2 && !p1)))) {
        set_error("fast project+bin: bad argument");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    FastWs w = carve_fast(ws, n, tiles_x * tiles_y);
    const int bs = per_gaussian_block(n);
    const dim3 grid((n + bs - 1) / bs > 0 ? (n + bs - 1) / bs : 1), block(bs);
    const BinTarget bt = bin_target(w, n, colors, opac, status);
#define GI2D_LAUNCH_PF(K)                                                                                       \
    hipLaunchKernelGGL(fast_project_fill_kernel<K>, grid, block, 0, (hipStream_t)st, n, clip_coe,              \
                       (const float2 *)means2d, p0, p1, (float)w_, (float)h, tiles_x, tiles_y, radius_clip,    \
                       (float2 *)xys, depths, radii, conics, nth, bt)
    if (kind == 0)
        GI2D_LAUNCH_PF(kCholesky);
    else if (kind == 1)
        GI2D_LAUNCH_PF(kCovariance);
    else
        GI2D_LAUNCH_PF(kScaleRot);
#undef GI2D_LAUNCH_PF
    return check_launch("fast project+bin");
}

int gi2d_fast_rasterize_forward(int n, int tiles_x, int tiles_y, unsigned w_, unsigned h, const float *background,
                                void *ws, size_t ws_bytes, int32_t *status, float *final_Ts, int32_t *final_idx,
                                float *out_img, gi2d_stream_t st) {
    int rc = check_ws("fast rasterize forward: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    const long long t = (long long)tiles_x * tiles_y;
    if (t == 0 || w_ == 0 || h == 0) return GI2D_OK;
    if ((unsigned)tiles_x * GI2D_TILE < w_ || (unsigned)tiles_y * GI2D_TILE < h) {
        set_error("fast rasterize forward: tile grid does not cover the image");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (!status || !out_img) {
        set_error("fast rasterize forward: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    FastWs w = carve_fast(ws, n, (int)t);
    if (t <= GI2D_TWO_PHASE_TILES && wt_fits(w, (int)t, (size_t)w_ * h * 12))  // one residency round: written through
        hipLaunchKernelGGL(fast_fwd_kernel<true>, dim3((unsigned)t), dim3(256), 0, (hipStream_t)st, tiles_x, tiles_y,
                           (int)w_, (int)h, rec_sets(w, n), background, w.lists, (int2 *)w.tile_bins, w.packed,
                           w.partial_g, w.partial_big, status, final_Ts, final_idx, out_img);
    else
        hipLaunchKernelGGL(fast_fwd_kernel<false>, dim3((unsigned)t), dim3(256), 0, (hipStream_t)st, tiles_x, tiles_y,
                           (int)w_, (int)h, rec_sets(w, n), background, w.lists, (int2 *)w.tile_bins, w.packed,
                           w.partial_g, w.partial_big, status, final_Ts, final_idx, out_img);
    return check_launch("fast rasterize forward");
}

int gi2d_fast_rasterize_forward_backward(int n, int tiles_x, int tiles_y, unsigned w_, unsigned h,
                                         const float *background, const float *v_output, const float *target,
                                         float grad_scale, float *tile_sse, void *ws, size_t ws_bytes,
                                         int32_t *status, float *out_img, gi2d_stream_t st) {
    return gi2d::fast_forward_backward_form(n, tiles_x, tiles_y, w_, h, background, v_output, target, grad_scale, tile_sse,
                                            ws, ws_bytes, status, out_img, st, -1, nullptr);
}
}  // extern "C"

namespace gi2d {
// form: 1 two launches, 0 one, -1 the rule of a call that knows nothing about the rows (two launches for every image of
// more than GI2D_TWO_PHASE_TILES tiles: right for sparse rows -- a 2040x1356 image at 50 000 gaussians -- and 10 % slow
// when some tile is fuller than the small form; gi2d_train_steps asks single_pass_begin instead)
int fast_forward_backward_form(int n, int tiles_x, int tiles_y, unsigned w_, unsigned h, const float *background,
                               const float *v_output, const float *target, float grad_scale, float *tile_sse, void *ws,
                               size_t ws_bytes, int32_t *status, float *out_img, gi2d_stream_t st, int form,
                               float4 *inbox) {
    int rc = check_ws("fast rasterize forward+backward: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    const long long t = (long long)tiles_x * tiles_y;
    if (t == 0 || w_ == 0 || h == 0) return GI2D_OK;
    if ((unsigned)tiles_x * GI2D_TILE < w_ || (unsigned)tiles_y * GI2D_TILE < h) {
        set_error("fast rasterize forward+backward: tile grid does not cover the image");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (!status || !out_img || ((v_output != nullptr) == (target != nullptr)) || (target && !tile_sse)) {
        set_error("fast rasterize forward+backward: bad argument (exactly one of v_output / target; tile_sse with target)");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    FastWs w = carve_fast(ws, n, (int)t);
    // (the inbox instantiation always stores write-through: it is only picked where that is possible)
    const bool wt = t <= GI2D_TWO_PHASE_TILES && wt_fits(w, (int)t, (size_t)w_ * h * 12);
    w.inbox_recs = wt ? inbox : nullptr;
    inbox = w.inbox_recs;
    TilePassArgs a = tile_pass_args(w, n, tiles_x, tiles_y, (int)w_, (int)h, status, out_img,
                                    v_output ? v_output : target, v_output ? 0.f : grad_scale,
                                    v_output ? nullptr : tile_sse);
    a.write_through = wt ? 1 : 0;
    const dim3 grid((unsigned)t), block(256);
    if (form < 0) form = pass_form_override() >= 0 ? pass_form_override() : 1;
    if (two_phase_tile_pass(t) && form >= 1) {
        KernelTimer *tm = next_timer();
        const dim3 grid2(phase2_blocks(t, form == 2));
        if (v_output) {
            launch_between(fast_fwdbwd_kernel<0, 1>, grid, block, (hipStream_t)st, tm ? tm->begin : nullptr, nullptr, a);
            launch_between(fast_fwdbwd_kernel<0, 2>, grid2, block, (hipStream_t)st, nullptr,
                           tm ? tm->end : nullptr, a);
        } else {
            launch_between(fast_fwdbwd_kernel<1, 1>, grid, block, (hipStream_t)st, tm ? tm->begin : nullptr, nullptr, a);
            launch_between(fast_fwdbwd_kernel<1, 2>, grid2, block, (hipStream_t)st, nullptr,
                           tm ? tm->end : nullptr, a);
        }
    } else if (v_output) {
        GI2D_LAUNCH_TIMED((fast_fwdbwd_kernel<0, 0>), grid, block, (hipStream_t)st, a);
    } else if (inbox != nullptr && t <= GI2D_INBOX_MAX_TILES) {
        GI2D_LAUNCH_TIMED((fast_fwdbwd_kernel<1, 0, true>), grid, block, (hipStream_t)st, a);
    } else {
        GI2D_LAUNCH_TIMED((fast_fwdbwd_kernel<1, 0>), grid, block, (hipStream_t)st, a);
    }
    if (background)
        hipLaunchKernelGGL(fast_background_kernel, dim3(256), dim3(256), 0, (hipStream_t)st, (int)w_, (int)h,
                           status, background, out_img);
    return check_launch("fast rasterize forward+backward");
}
}  // namespace gi2d

extern "C" {
int gi2d_batch_tile_pass_form(const void *batch) {
    const int forced = pass_form_override();
    if (forced >= 0) return forced;
    std::lock_guard<std::mutex> lock(g_hint_mu);
    const auto it = g_hints.find(batch);
    if (it == g_hints.end()) return 0;
    hint_poll(it->second);
    return hint_says_two_phase(it->second, it->second.total, it->second.min_total) != 0 ? 1 : 0;
}

int gi2d_fast_rasterize_forward_backward_batched(int num_images, const gi2d_fast_image *images, void *batch,
                                                 size_t batch_bytes, gi2d_stream_t st) {
    if (num_images < 1 || num_images > GI2D_BATCH_MAX || !images) {
        set_error("fast rasterize forward+backward (batched): 1 .. 64 images per launch");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (!batch || batch_bytes < carve_batch(nullptr, num_images).bytes || ((uintptr_t)batch & 15)) {
        set_error("fast rasterize forward+backward (batched): batch table too small (gi2d_batch_bytes) or misaligned");
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    BatchHead head;
    std::memset(&head, 0, sizeof(head));
    std::vector<BatchImage> host_imgs((size_t)num_images);
    std::memset(host_imgs.data(), 0, host_imgs.size() * sizeof(BatchImage));
    int blocks = 0, tiles0 = 0;
    bool uniform = true;
    const bool given = images[0].v_output != nullptr;
    for (int k = 0; k < num_images; ++k) {
        const gi2d_fast_image &im = images[k];
        int rc = check_ws("fast rasterize forward+backward (batched): workspace too small", im.workspace,
                          im.workspace_bytes, im.num_points, im.tiles_x, im.tiles_y);
        if (rc != GI2D_OK) return rc;
        const long long t = (long long)im.tiles_x * im.tiles_y;
        if (t == 0 || im.img_width == 0 || im.img_height == 0 || (unsigned)im.tiles_x * GI2D_TILE < im.img_width ||
            (unsigned)im.tiles_y * GI2D_TILE < im.img_height) {
            set_error("fast rasterize forward+backward (batched): empty image or tile grid that does not cover it");
            return GI2D_ERR_INVALID_ARGUMENT;
        }
        if (!im.status || !im.out_img || ((im.v_output != nullptr) == (im.target != nullptr)) ||
            (im.target && !im.tile_sse) || (im.v_output != nullptr) != given) {
            set_error("fast rasterize forward+backward (batched): exactly one of v_output / target per image, the same "
                      "kind for the whole batch; tile_sse with target");
            return GI2D_ERR_INVALID_ARGUMENT;
        }
        FastWs w = carve_fast(im.workspace, im.num_points, (int)t);
        host_imgs[k].t = tile_pass_args(w, im.num_points, im.tiles_x, im.tiles_y, (int)im.img_width, (int)im.img_height,
                                        im.status, im.out_img, given ? im.v_output : im.target,
                                        given ? 0.f : im.grad_scale, given ? nullptr : im.tile_sse);
        head.tile_start[k] = blocks;
        blocks += (int)t;
        if (k == 0) tiles0 = (int)t;
        uniform = uniform && (int)t == tiles0;
    }
    head.tile_start[num_images] = blocks;
    BatchTable b = carve_batch(batch, num_images);
    write_batch_table(b, host_imgs.data(), num_images, head, (hipStream_t)st);
    const int two_phase = batch_pass_begin(batch, blocks, (hipStream_t)st);
    const int rc = launch_tile_pass_batched(given ? 0 : 1, b, num_images, blocks, uniform ? tiles0 : 0, two_phase,
                                            (hipStream_t)st);
    batch_pass_end(batch, b, num_images, blocks, (hipStream_t)st);
    return rc;
}

int gi2d_fast_rasterize_backward_tiles(int n, int tiles_x, int tiles_y, unsigned w_, unsigned h,
                                       const int32_t *final_idx, const float *v_output, int with_abs, void *ws,
                                       size_t ws_bytes, gi2d_stream_t st) {
    int rc = check_ws("fast rasterize backward: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    const long long t = (long long)tiles_x * tiles_y;
    if (t == 0) return GI2D_OK;
    if (!v_output) {
        set_error("fast rasterize backward: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    FastWs w = carve_fast(ws, n, (int)t);
#define GI2D_LAUNCH_BWD(ABS, WT)                                                                                       \
    hipLaunchKernelGGL((fast_bwd_kernel<ABS, WT>), dim3((unsigned)t), dim3(256), 0, (hipStream_t)st, tiles_x, tiles_y, \
                       (int)w_, (int)h, (const int2 *)w.tile_bins, w.packed, final_idx, v_output, w.partial_g,        \
                       w.partial_big)
    const bool wt = t <= GI2D_TWO_PHASE_TILES && wt_fits(w, (int)t, 0);  // one residency round: the gradient rows are written through
    if (with_abs) {
        if (wt) GI2D_LAUNCH_BWD(true, true); else GI2D_LAUNCH_BWD(true, false);
    } else {
        if (wt) GI2D_LAUNCH_BWD(false, true); else GI2D_LAUNCH_BWD(false, false);
    }
#undef GI2D_LAUNCH_BWD
    return check_launch("fast rasterize backward tiles");
}

int gi2d_fast_rasterize_backward_reduce(int n, int tiles_x, int tiles_y, void *ws, size_t ws_bytes, float *v_xy,
                                        float *v_conic, float *v_rgb, float *v_opacity, float *v_abs_xy,
                                        gi2d_stream_t st) {
    int rc = check_ws("fast rasterize backward: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    if (n == 0) return GI2D_OK;
    if (!v_xy || !v_conic || !v_rgb || !v_opacity) {
        set_error("fast rasterize backward reduce: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    FastWs w = carve_fast(ws, n, tiles_x * tiles_y);
    const int rbs = per_gaussian_block(n);
    hipLaunchKernelGGL(fast_reduce_kernel, dim3((n + rbs - 1) / rbs), dim3(rbs), 0, (hipStream_t)st, n,
                       (const PrevBox *)w.prev_box, tiles_x, tiles_y, w.gids_sorted, (const int2 *)w.tile_bins, w.partial_g,
                       w.partial_big, (float2 *)v_xy, v_conic, v_rgb, v_opacity, (float4 *)v_abs_xy);
    return check_launch("fast rasterize backward reduce");
}

static int reduce_project_impl(int kind, int n, const float *p0, const float *p1, unsigned h, unsigned w_, float *xys,
                               int32_t *radii, float *conics, int tiles_x, int tiles_y, float radius_clip, void *ws,
                               size_t ws_bytes, float *v_xy, float *v_conic, float *v_rgb, float *v_opacity,
                               float *v_abs_xy, float *v_cov2d, float *v_mean2d, float *v_p0, float *v_p1,
                               const NextProject *next, gi2d_stream_t st) {
    int rc = check_ws("fast reduce+project backward: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    if (n == 0) return GI2D_OK;
    if (kind < 0 || kind > 2 || !p0 || !xys || !radii || !conics || !v_xy || !v_conic || !v_rgb || !v_opacity ||
        !v_mean2d || !v_p0 || (kind == 2 && (!p1 || !v_p1))) {
        set_error("fast reduce+project backward: bad argument");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    FastWs w = carve_fast(ws, n, tiles_x * tiles_y);
    NextProject np;
    np.clip_coe = 0.f;
    np.means2d = nullptr;
    np.depths = nullptr;
    np.num_tiles_hit = np.tile_order = nullptr;
    np.bt = bin_target(w, n, nullptr, nullptr, nullptr);
    if (next) {
        np = *next;
        np.bt = bin_target(w, n, next->bt.colors, next->bt.opacities, next->bt.status);
    }
    const int bs = per_gaussian_block(n), role_blocks = (n + bs - 1) / bs;
    np.tile_order = (next && n > 32768) ? w.tile_order : nullptr;
    const dim3 grid((next ? 2 : 1) * role_blocks + (np.tile_order ? 1 : 0)), block(bs);
#define GI2D_LAUNCH_RP(K, F)                                                                                        \
    hipLaunchKernelGGL((fast_reduce_project_kernel<K, F>), grid, block, 0, (hipStream_t)st, n, role_blocks,          \
                       (float2 *)xys, radii, conics, tiles_x, tiles_y, radius_clip, w.prev_box, w.gids_sorted,       \
                       (const int2 *)w.tile_bins,                                                                    \
                       w.partial_g, w.partial_big, p0, p1, (float)w_, (float)h, (float2 *)v_xy, v_conic, v_rgb,      \
                       v_opacity, (float4 *)v_abs_xy, v_cov2d, (float2 *)v_mean2d, v_p0, v_p1, np)
#define GI2D_LAUNCH_RP2(K)           \
    do {                             \
        if (next)                    \
            GI2D_LAUNCH_RP(K, true); \
        else                         \
            GI2D_LAUNCH_RP(K, false); \
    } while (0)
    if (kind == 0)
        GI2D_LAUNCH_RP2(kCholesky);
    else if (kind == 1)
        GI2D_LAUNCH_RP2(kCovariance);
    else
        GI2D_LAUNCH_RP2(kScaleRot);
#undef GI2D_LAUNCH_RP2
#undef GI2D_LAUNCH_RP
    return check_launch("fast reduce+project backward");
}

int gi2d_fast_reduce_project_backward(int kind, int n, const float *p0, const float *p1, unsigned h, unsigned w_,
                                      const float *xys, const int32_t *radii, const float *conics, int tiles_x,
                                      int tiles_y, float radius_clip, void *ws, size_t ws_bytes, float *v_xy,
                                      float *v_conic, float *v_rgb, float *v_opacity, float *v_abs_xy,
                                      float *v_cov2d, float *v_mean2d, float *v_p0, float *v_p1,
                                      gi2d_stream_t st) {
    return reduce_project_impl(kind, n, p0, p1, h, w_, (float *)xys, (int32_t *)radii, (float *)conics, tiles_x, tiles_y,
                               radius_clip, ws, ws_bytes, v_xy, v_conic, v_rgb, v_opacity, v_abs_xy, v_cov2d, v_mean2d,
                               v_p0, v_p1, nullptr, st);
}

int gi2d_fast_reduce_project_backward_project_bin(int kind, int n, float clip_coe, const float *means2d,
                                                  const float *p0, const float *p1, const float *colors,
                                                  const float *opac, unsigned h, unsigned w_, float *xys,
                                                  float *depths, int32_t *radii, float *conics, int32_t *nth,
                                                  int tiles_x, int tiles_y, float radius_clip, void *ws,
                                                  size_t ws_bytes, int32_t *status, float *v_xy, float *v_conic,
                                                  float *v_rgb, float *v_opacity, float *v_abs_xy, float *v_cov2d,
                                                  float *v_mean2d, float *v_p0, float *v_p1, gi2d_stream_t st) {
    if (n > 0 && (!means2d || !depths || !nth || !status || !colors || !opac)) {
        set_error("fast reduce+project backward + project+bin: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    NextProject np;
    np.clip_coe = clip_coe;
    np.means2d = (const float2 *)means2d;
    np.depths = depths;
    np.num_tiles_hit = nth;
    np.tile_order = nullptr;
    np.bt.colors = colors;
    np.bt.opacities = opac;
    np.bt.status = status;
    np.bt.prev_box = nullptr;
    np.bt.lists = nullptr;
    return reduce_project_impl(kind, n, p0, p1, h, w_, xys, radii, conics, tiles_x, tiles_y, radius_clip, ws, ws_bytes,
                               v_xy, v_conic, v_rgb, v_opacity, v_abs_xy, v_cov2d, v_mean2d, v_p0, v_p1, &np, st);
}

#ifdef GI2D_FUSED_TRACE
int gi2d_debug_set_trace(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_trace), &buf, sizeof(buf));
}
#endif

// Views into the workspace for callers that want the binning result itself (tests, debugging).
int gi2d_fast_workspace_views(void *ws, size_t ws_bytes, int n, int tiles_x, int tiles_y,
                              int32_t **gaussian_ids_sorted, int32_t **tile_bins) {
    int rc = check_ws("fast workspace views: workspace too small", ws, ws_bytes, n, tiles_x, tiles_y);
    if (rc != GI2D_OK) return rc;
    FastWs w = carve_fast(ws, n, tiles_x * tiles_y);
    if (gaussian_ids_sorted) *gaussian_ids_sorted = w.gids_sorted;
    if (tile_bins) *tile_bins = w.tile_bins;
    return GI2D_OK;
}

}  // extern "C"
