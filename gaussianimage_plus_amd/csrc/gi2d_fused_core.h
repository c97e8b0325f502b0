// One workgroup = one 16x16 tile, forward AND backward of the additive rasterizer in a single pass.
//
// The fused fast path's backward does not consume anything the forward computes except the tile's ordered
// gaussian list (it re-evaluates every pair with the same instructions), and the gradient of a pixel depends on
// that pixel alone -- either it is given (a gradient image), or it is the L2-loss gradient of the pixel the
// forward has just produced.  So the tile's gaussians are ranked, gathered and staged in LDS ONCE, the forward
// runs on them, each lane turns its pixel into its gradient, and the backward items run on the same staged
// records: no second launch, no packed-list round trip through HBM, no second cursor / list / record load chain.
//
// LDS: records (raw conic), cull words and partial-row codes by list position are shared by both phases; the
// sort buffer, the forward's lists + pair buffers and the backward's pixel / item / hand-off buffers overlay
// each other (they are live in disjoint phases, separated by workgroup barriers).
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

// Development aid (-DGI2D_FUSED_TRACE): thread 0 of every workgroup stamps the 100 MHz wall clock at the phase
// boundaries into a buffer set with gi2d_debug_set_trace (tools/trace_fused.py).
#ifdef GI2D_FUSED_TRACE
namespace gi2d {
__device__ unsigned long long *g_fused_trace = nullptr;  // [tiles][16]
}
#define GI2D_TRACE_AT(wg, i)                                                                        \
    do {                                                                                            \
        if (threadIdx.x == 0 && g_fused_trace) g_fused_trace[(wg) * 16 + (i)] = wall_clock64();     \
    } while (0)
#define GI2D_TRACE(i) GI2D_TRACE_AT(tile, i)
#define GI2D_TRACE_VALUE(i, v)                                                        \
    do {                                                                              \
        if (threadIdx.x == 0 && g_fused_trace) g_fused_trace[tile * 16 + (i)] = (v);  \
    } while (0)
#define GI2D_BWD_TRACE(i) GI2D_TRACE_AT(blockIdx.x, i)
#define GI2D_HEAD_TRACE(i) GI2D_TRACE_AT(tile, i)
#else
#define GI2D_TRACE(i) \
    do {              \
    } while (0)
#define GI2D_TRACE_VALUE(i, v) \
    do {                       \
    } while (0)
#endif

#include "gi2d_fast_internal.h"

namespace gi2d {

#ifndef GI2D_FUSED_OCC
#define GI2D_FUSED_OCC 6 /* workgroups per CU the register allocator leaves room for: 1536 tiles of a 768x512 image
                            are then resident at once (measured: a 27 KB workgroup still gets only 5 per CU and a
                            second, 256-workgroup round; 23.7 KB gets 6) */
#endif

// CAP: list entries the staging arrays hold.  GI2D_TILE_LIST_CAP (256: forward.cu:553) for the general kernel; the
// small form (GI2D_SMALL_CAP) serves the tiles whose ROW holds at most that many candidates -- every tile of a
// 2040x1356 image at 50 000 gaussians, every tile of the uniform bench scene, every tile of a fit's first 45 000
// iterations -- from 18 KB of LDS and 62 registers instead of 26 KB and 76: EIGHT workgroups per CU instead of six.
// The tile pass is short of runnable waves, not of issue slots: launches with more tiles than the chip
// holds at once run the small form first and the general one on the tiles it passed over (gi2d_fast.hip).
#ifndef GI2D_SMALL_CAP
#define GI2D_SMALL_CAP 128
#endif
template <int CAP>
struct FusedLdsT {
    static constexpr int PSTR = 9;
    static constexpr bool HAS_FIDX = false;
    static constexpr bool HAS_RAW = false;
    static constexpr int PART_ROWS = GI2D_BWD_PART_ROWS;
    static constexpr int LISTLEN = CAP + 8;
    static constexpr int IDS = CAP == GI2D_TILE_LIST_CAP ? GI2D_FAST_C : CAP;  // candidates of the row the head sorts
    float4 gA[CAP + 1];  // gx, gy, ha, hb      (entry CAP: the forward's never-contributing padding;
    float4 gB[CAP + 1];  // hc, opac, cr, cg     conic pre-scaled: gi2d_common.h::scale_conic)
    float2 gC[CAP + 2];  // cb, lim (gi2d_common.h::AlphaRule)
    unsigned cullw[CAP]; // cull_word() of the entry
    float sse_w[4];
    int scan_w[8];  // per-wave totals of the backward's item scan (outside the overlay: written during the forward phase)
    int grp[32];    // tile_list_head: survivors per 64 entries (ascending part, appended part)
    union {
        struct {
            int ids[IDS];
        };
        struct {  // forward phase
            unsigned char lists[4][2 * LISTLEN];  // per wave: left-half list, right-half list
            float4 pairbuf[GI2D_FWD_PAIRBUF];  // 4 waves x GI2D_FWD_PAIRBUF floats
        };
        struct {  // backward phase (member names as BwdLds: bwd_run_tile is shared)
            float4 pix[2 * GI2D_BWD_PIXRECS];
            unsigned short span[CAP];
            unsigned short item[8 * CAP];
            float part[GI2D_BWD_PART_ROWS * PSTR];
            int wsum[8];
            int n_items;
        };
        struct {  // partial-row code of the entry (see fast path: >= 0 gaussian-major, < 0 big): written by the head,
                  // read when the backward starts -- in bytes the forward's buffers do not reach and the backward only
                  // writes in its hand-off (`part`), behind the barrier that follows the reads
            char fwd_reach[sizeof(unsigned char) * 4 * 2 * LISTLEN + sizeof(float4) * GI2D_FWD_PAIRBUF];
            int slot[CAP];
        };
    };
    // rows / columns of entry k's box: its cull word stays staged through both phases
    __device__ __forceinline__ void set_box(int, unsigned) {}
    __device__ __forceinline__ unsigned box_of(int k) const { return cullw[k] >> 8; }
};
typedef FusedLdsT<GI2D_TILE_LIST_CAP> FusedLds;
typedef FusedLdsT<GI2D_SMALL_CAP> FusedLdsSmall;
#ifndef GI2D_SMALL_OCC
#define GI2D_SMALL_OCC 8 /* workgroups per CU of the small form: 62 registers, 18.2 KB */
#endif

// measured: 26.5 KB still leaves room for six workgroups per CU, 27.1 KB does not
static_assert(GI2D_FUSED_OCC < 6 || sizeof(FusedLds) <= 26624,
              "FusedLds: the sixth workgroup per CU needs <= 26 KB (see GI2D_FUSED_OCC)");
static_assert(sizeof(FusedLdsSmall) <= 160 * 1024 / GI2D_SMALL_OCC - 512, "FusedLdsSmall: eight workgroups per CU need <= 19.5 KB");
// slot[] must lie inside `part` (dead until the hand-off) in both forms
static_assert(offsetof(FusedLds, slot) >= offsetof(FusedLds, part) &&
                  offsetof(FusedLds, slot) + sizeof(int) * GI2D_TILE_LIST_CAP <= offsetof(FusedLds, wsum),
              "FusedLds: the partial-row codes must overlay the hand-off buffer only");
static_assert(offsetof(FusedLdsSmall, slot) >= offsetof(FusedLdsSmall, part) &&
                  offsetof(FusedLdsSmall, slot) + sizeof(int) * GI2D_SMALL_CAP <= offsetof(FusedLdsSmall, wsum),
              "FusedLdsSmall: the partial-row codes must overlay the hand-off buffer only");

// MODE 0: `vsrc` is the gradient image v_output[H,W,3].
// MODE 1: `vsrc` is the target image gt[H,W,3]; the pixel gradient is that of mean((clamp(out,0,1) - gt)^2):
//         grad_scale * (clamp(out) - gt) where the clamp passes gradient (models/gaussianimage_cholesky.py:307-310
//         with loss_type "L2"), and tile_sse[tile] receives the tile's sum of squared errors (fixed order).
// INBOX: see tile_list_head.
// WT: gradient rows and image leave as write-through stores (gi2d_raster_core.h::store16): the single-image launches.
template <int MODE, int CAP = GI2D_TILE_LIST_CAP, int FWD_UNROLL = GI2D_FWD_UNROLL, bool INBOX = false, bool WT = false>
__device__ __forceinline__ void fused_tile(
    FusedLdsT<CAP> &sm, int tile, int tiles_x, int tiles_y, int img_w, int img_h, const float4 *__restrict__ recs,
    int32_t *__restrict__ lists, int2 *__restrict__ tile_bins,
    float4 *__restrict__ partial_g, float4 *__restrict__ partial_big, int32_t *__restrict__ status,
    float *__restrict__ out_img, const float *__restrict__ vsrc, float grad_scale, float *__restrict__ tile_sse,
    const HeadRow head_row, const Inbox &ib) {
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int pool_rows = tiles_x * tiles_y * GI2D_TILE_LIST_CAP;  // rows of `partial_big`, the row pool (PrevBox)
    static_assert(CAP <= GI2D_TILE_LIST_CAP, "at most the reference's 256 entries of a tile are rasterized");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // wave wv owns pixel rows 4wv..4wv+3; within a row of 16 lanes the pixel a lane holds after the forward is
    // fwd_lane_col (gi2d_raster_core.h: one entry for two adjacent pixels per lane, the parities meet at the end)
    const int lx = fwd_lane_col(tid), ly = tid >> 4;
    const int j = tx * GI2D_TILE + lx, i = ty * GI2D_TILE + ly;
    const bool inside = (i < img_h) && (j < img_w);
    const size_t pix = (size_t)i * img_w + j;

    GI2D_TRACE(0);
    // this lane's pixel of the gradient / target image: issued first, consumed after the forward
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (inside) {
        p0 = vsrc[3 * pix];
        p1 = vsrc[3 * pix + 1];
        p2 = vsrc[3 * pix + 2];
    }

    // ---- the tile's row -> validated, ordered, staged list (gi2d_fast_internal.h::tile_list_head)
    if (tid == 0) {
        sm.gA[CAP] = make_float4(0.f, 0.f, 0.f, 0.f);
        sm.gB[CAP] = make_float4(0.f, 0.f, 0.f, 0.f);  // opacity 0: alpha = 0 < 1/255
        sm.gC[CAP] = make_float2(0.f, 0.f);            // lim 0: never lands
    }
    GI2D_TRACE_VALUE(1, (unsigned long long)blockIdx.x);  // the launch slot (single-image launches)
    const float tx0 = (float)(tx * GI2D_TILE), ty0 = (float)(ty * GI2D_TILE);
    struct Staged {  // one entry as both phases consume it
        float4 A, B;
        float2 C;
        unsigned cull;
        int slot;
    };
    const int L = tile_list_head<true, INBOX>(
        sm.ids, sm.grp, tile, tx, ty, recs, lists, tile_bins, status,
        [&](int g, const BinRec &br) {
            const GaussRec &r = br.r;
            Staged s;
            s.slot = partial_slot(g, br.box, tx, ty, br.pool);
            const ConicS cs = scale_conic(r.a, r.b, r.c);
            s.A = make_float4(r.gx, r.gy, cs.ha, cs.hb);
            s.B = make_float4(cs.hc, r.opac, r.cr, r.cg);
            const AlphaRule ar = alpha_rule(r.gx, r.gy, r.a, r.b, r.c, r.opac);
            s.C = make_float2(r.cb, __int_as_float((int)ar.lim));
            s.cull = cull_word_ext(r.gx, r.gy, br.hx, br.hy, tx0, ty0, img_h, ar.clamp);
            return s;
        },
        [&](int rank, int, const Staged &s) {
            if (rank < CAP) {  // (the small form only sees rows of at most CAP candidates: always)
                sm.gA[rank] = s.A;
                sm.gB[rank] = s.B;
                sm.gC[rank] = s.C;
                sm.cullw[rank] = s.cull;
                sm.slot[rank] = s.slot;
            } else if (float4 *row = partial_row(s.slot, partial_g, partial_big, pool_rows, status)) {
                // beyond the 256-entry cap: never rasterized, its gradient row must read as zero
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                store16<WT>(row, z, partial_g);  // (the row pool lies behind the gaussian-major rows: carve_fast)
                store16<WT>(row + 1, z, partial_g);
                store16<WT>(row + 2, z, partial_g);
            }
        }, ib, head_row);
    GI2D_TRACE(2);
#if defined(GI2D_STOP_AFTER) && GI2D_STOP_AFTER == 1 /* development aid: instruction / time budget of the phases */
    if (L >= 0) return;
#endif
    // records staged and every lane done with sm.ids (tile_list_head<true> returns behind its last barrier): the overlay
    // may now hold the forward's buffers
    const int len = L > CAP ? CAP : L;
    GI2D_TRACE(3);

    // ---- forward (the routine every forward kernel shares: gi2d_raster_core.h::fwd_pixel_half_lists)
    float *mybuf = reinterpret_cast<float *>(sm.pairbuf) + wv * GI2D_FWD_PAIRBUF;
    float o0, o1, o2;
    int last_unused;
    GI2D_TRACE(4);
    fwd_pixel_half_lists<false, CAP, FWD_UNROLL>(
        sm.lists[wv], mybuf, len, [&](int k) { return sm.cullw[k]; },
        [&](int k) {
            const float4 A = sm.gA[k], B = sm.gB[k];
            FwdRec r;
            r.gx = A.x, r.gy = A.y, r.ha = A.z, r.hb = A.w, r.hc = B.x, r.op = B.y, r.cr = B.z, r.cg = B.w;
            const float2 c = sm.gC[k];
            r.cb = c.x, r.lim = (unsigned)__float_as_int(c.y);
            return r;
        },
        tx0, (float)i, o0, o1, o2, last_unused);
    GI2D_TRACE(5);
#if defined(GI2D_STOP_AFTER) && GI2D_STOP_AFTER == 2
    if (L >= 0) {
        if (o0 + o1 + o2 == 12345.f) out_img[0] = o0;
        return;
    }
#endif
    if (WT && (img_w & 3) == 0 && (reinterpret_cast<uintptr_t>(out_img) & 15) == 0 && (tx + 1) * GI2D_TILE <= img_w &&
        (ty + 1) * GI2D_TILE <= img_h)  // tile-uniform: rows of 16-byte pieces, a tile inside the image
        fwd_store_pixels_wt(o0, o1, o2, tx, ty, img_w, mybuf, out_img);
    else
        fwd_store_pixels(o0, o1, o2, tx, ty, img_w, img_h, out_img);

    // ---- this pixel's gradient
    float v0 = p0, v1 = p1, v2 = p2, sse = 0.f;
    if (MODE == 1) {
        v0 = v1 = v2 = 0.f;
        if (inside) {
            const float o[3] = {o0, o1, o2}, gtv[3] = {p0, p1, p2};
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float oc = __builtin_amdgcn_fmed3f(o[c], 0.f, 1.f);  // torch.clamp(out_img, 0, 1): one v_med3_f32
                                                                            // (a NaN pixel reads as 0 either way)
                const float d = oc - gtv[c];
                sse += d * d;
                v[c] = (oc == o[c]) ? grad_scale * d : 0.f;  // clamp passes the gradient on [0, 1] (<=> it changed nothing)
            }
            v0 = v[0], v1 = v[1], v2 = v[2];
        }
        // summed in COLUMN order within each pixel row, as every round has (the per-tile squared errors decide the
        // best-model snapshot: a fit stays bit-identical to round 5's): lane r of a row fetches column r's term
        sse = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * ((tid & 48) | fwd_col_lane(tid & 15)), __float_as_int(sse)));
        sse = wave_sum_dpp(sse);
        if (lane == 0) sm.sse_w[wv] = sse;
    }
    // the backward's item scan starts here, ahead of the barrier that is needed anyway
    const unsigned cull = tid < len ? sm.cullw[tid] : 0u;
    const unsigned long long scan_incl = bwd_prescan(sm.scan_w, cull);
    __syncthreads();  // every wave is done with its list / pair buffer: the overlay becomes the backward's buffers

    GI2D_TRACE(6);
#if defined(GI2D_STOP_AFTER) && GI2D_STOP_AFTER == 3
    if (L >= 0) {
        if (v0 + v1 + v2 + (float)(unsigned)scan_incl == 12345.f) out_img[0] = v0;
        return;
    }
#endif
    // ---- backward on the same staged records
    bwd_publish_pixel(sm, lx, ly, v0, v1, v2, 0.f);
    float4 *dst = nullptr;
    if (tid < len) dst = partial_row(sm.slot[tid], partial_g, partial_big, pool_rows, status);
    bwd_run_tile<false, false, true, WT>(sm, len, cull, 0, tx0, ty0, dst, scan_incl, sm.scan_w, partial_g);
    GI2D_TRACE(10);
    GI2D_TRACE_VALUE(14, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));   // HW_REG_HW_ID
    GI2D_TRACE_VALUE(15, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));  // HW_REG_XCC_ID
    GI2D_TRACE_VALUE(4, (unsigned long long)L);
    if (MODE == 1 && tid == 0) tile_sse[tile] = (sm.sse_w[0] + sm.sse_w[1]) + (sm.sse_w[2] + sm.sse_w[3]);
    // "No intersection at all" is a global property: see fast_fwd_kernel
    if (tid == 0 && L > 0) status[0] = 1;
}

}  // namespace gi2d
