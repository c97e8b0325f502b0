// Device-side core of the additive tile rasterizer, shared by the reference-shaped ops
// (gi2d_raster.hip) and the fused fast path (gi2d_fast.hip).  See gi2d_raster.hip for the design notes.
#pragma once
#include "gi2d_common.h"

namespace gi2d {

#define GI2D_ALPHA_MIN (1.f / 255.f)
typedef float v2f __attribute__((ext_vector_type(2)));

// One gaussian as the rasterizer consumes it (48 bytes): the "packed record" of the fast path.
struct GaussRec {
    float gx, gy, a, b;      // centre (pixels), conic a, b
    float c, opac, cr, cg;   // conic c, opacity, colour r, g
    float cb;                // colour b
    int slot;                // fast path: row of the gaussian-major partial buffer (-1: none)
    int gid;                 // gaussian id
    int pad;
};
static_assert(sizeof(GaussRec) == 48, "GaussRec must be 3 x float4");

__device__ __forceinline__ GaussRec load_gaussian(int g, const float2 *__restrict__ xys,
                                                  const float *__restrict__ conics,
                                                  const float *__restrict__ colors,
                                                  const float *__restrict__ opacities) {
    GaussRec r;
    const float2 xy = xys[g];
    r.gx = xy.x;
    r.gy = xy.y;
    r.a = conics[3 * g];
    r.b = conics[3 * g + 1];
    r.c = conics[3 * g + 2];
    r.opac = opacities[g];
    r.cr = colors[3 * g];
    r.cg = colors[3 * g + 1];
    r.cb = colors[3 * g + 2];
    r.slot = -1;
    r.gid = g;
    r.pad = 0;
    return r;
}

// Where inside a tile can the gaussian reach alpha >= 1/255?  Packed "cull word":
//   bits 0-3   the 4-row strips it reaches (forward: one wave per strip)
//   bits 8-11  first row r0, bits 12-15 last row r1, bits 16-19 first column c0, bits 20-23 last column c1
//   (rows / columns of the 16x16 tile, from the conservative box of gi2d_common.h::cull_box, clipped to the
//   image height); 0 when it reaches nothing.  Evaluating more pixels than necessary never changes a
//   result (they fail the alpha test); the box guarantees none that passes is left out.
//   bit 24     min(1, opac * vis) can bind for this gaussian (gi2d_common.h::AlphaRule::clamp)
#define GI2D_CULL_CLAMP (1u << 24)
__device__ __forceinline__ unsigned cull_word_of(const CullBox &box, float tx0, float ty0, int img_h, bool clamp) {
    const float last_row = fminf(15.f, (float)(img_h - 1) - ty0);
    const float r0f = fmaxf(ceilf(box.y0 - ty0), 0.f), r1f = fminf(floorf(box.y1 - ty0), last_row);
    const float c0f = fmaxf(ceilf(box.x0 - tx0), 0.f), c1f = fminf(floorf(box.x1 - tx0), 15.f);
    if (!(r1f >= r0f) || !(c1f >= c0f)) return 0u;
    const unsigned r0 = (unsigned)r0f, r1 = (unsigned)r1f, c0 = (unsigned)c0f, c1 = (unsigned)c1f;
    const unsigned s0 = r0 >> 2, s1 = r1 >> 2;
    const unsigned strips = ((2u << s1) - 1u) & ~((1u << s0) - 1u);  // bits s0 .. s1
    return strips | (r0 << 8) | (r1 << 12) | (c0 << 16) | (c1 << 20) | (clamp ? GI2D_CULL_CLAMP : 0u);
}
__device__ __forceinline__ unsigned cull_word(const GaussRec &r, float tx0, float ty0, int img_h, bool clamp) {
    CullBox box;
    if (!cull_box(r.gx, r.gy, r.a, r.b, r.c, r.opac, box)) return 0u;
    return cull_word_of(box, tx0, ty0, img_h, clamp);
}
// the same from extents computed once per gaussian (fast path records)
__device__ __forceinline__ unsigned cull_word_ext(float gx, float gy, float hx, float hy, float tx0, float ty0,
                                                  int img_h, bool clamp) {
    CullBox box;
    if (!cull_box_of(gx, gy, hx, hy, box)) return 0u;
    return cull_word_of(box, tx0, ty0, img_h, clamp);
}

// WT: WRITE-THROUGH stores (`sc1`: the 16 bytes go to memory now and the line is dropped from this XCD's L2).  A
// kernel's dirty L2 lines are written back when it ends and the next kernel of the stream waits for that (the guide's
// price list: + bytes / 6 TB/s per dependent boundary -- 2 us behind the ~12 MB of gradient rows and image a single-image
// tile pass leaves); written through, the bytes leave while the kernel still computes.  The consumer of the rows (the
// update kernel: other CUs, all XCDs) reads them from memory either way.  Measured (tools/ab_trace.sh): the single-image
// tile pass -0.56 us, the update kernel behind it -0.24 us -- and a 24-image launch, whose stores already overlap
// other tiles' work and which is short of memory-system slots, +28 %: the batched kernels store plainly.
// The stores are raw BUFFER stores with the sc1 cache-policy bit (the compiler sees them: hazards and waits are its
// business, which they are not with inline assembly); `base`: a wave-uniform pointer at or below every address stored to
// (the buffer's resource lives in scalar registers), within 4 GB of it.
typedef unsigned int wt_u4 __attribute__((ext_vector_type(4)));
#define GI2D_WT_AUX 16 /* cache policy of the raw buffer store builtins on gfx94x / gfx950: bit 4 = sc1 */
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_resource(const void *base) {
    // (num_records = the whole 32-bit offset range: the launch code only picks a write-through kernel where every address
    // stored to lies within 4 GB of its base -- gi2d_fast_internal.h::wt_fits)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ unsigned wt_offset(const void *p, const void *base) {
    return (unsigned)(reinterpret_cast<const char *>(p) - reinterpret_cast<const char *>(base));
}
template <bool WT>
__device__ __forceinline__ void store16(float4 *p, const float4 v, const void *base) {
    if constexpr (WT) {
        const wt_u4 r = {(unsigned)__float_as_int(v.x), (unsigned)__float_as_int(v.y), (unsigned)__float_as_int(v.z),
                         (unsigned)__float_as_int(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(r, wt_resource(base), (int)wt_offset(p, base), 0, GI2D_WT_AUX);
    } else {
        *p = v;
    }
}
// (before a lane re-reads a row it stored write-through earlier in the same kernel: the store and the load travel
// different ways)
template <bool WT>
__device__ __forceinline__ void wt_drain() {
    if constexpr (WT) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
}

// =========================================================================================== forward
// Wave w owns pixel rows 4w..4w+3 (a "strip").  A gaussian's alpha >= 1/255 box (cull_word) typically spans ~8 columns
// at 50 000 gaussians per 768x512 image, i.e. one half of a strip more often than both, so each wave keeps TWO ascending
// lists -- entries whose box reaches the left half (columns 0..7), entries whose box reaches the right half (an entry
// may sit in both) -- and the two halves are walked side by side: a wave needs max(|left|, |right|) / 2 trips instead of
// |left u right| / 2 (-37 % at that size).
//
// Lane layout (round 6): ONE entry for TWO horizontally adjacent pixels per lane and trip.  The 16 lanes of a DPP row
// serve one pixel row of the strip: lane r (0..15) of the row = (parity r >> 3, half (r >> 2) & 1, column pair r & 3)
// evaluates, for its half's list, the entries at positions 2t + parity on the pixels (8 half + 2 pair, + 1) of its row.
// The row terms dy, b dy, c dy^2 are formed ONCE per lane and trip with plain fp32 instructions (the two pixels share
// them), the two dx and everything behind them with packed fp32 (v_pk_fma_f32 & co) -- 4 plain + 7 packed + 2 v_exp +
// 2 compares + 2 selects per trip, where rounds 3-5 (two entries for one pixel) spent 11 packed ones: the plain
// instructions issue in half a packed one's cycles (tools/ubench/valu_rate.hip), and a trip reads ONE staged entry
// (40 bytes: two ds_read_b128 + one ds_read_b64) instead of two interleaved ones (five ds_read_b128).  Even and odd list
// positions are accumulated by different lanes -- exactly the two partial sums of the earlier layout -- and meet at the
// end in one DPP add per channel (row_ror:8 swaps the parities of a row), even + odd, so every pixel keeps its bits.
// After the exchange the lane of row position r holds the pixel of column fwd_lane_col(r) = 8 half + 2 pair + parity:
// that is the lane -> pixel map of every kernel that runs this routine (fused_tile, fwd_rasterize_staged).
// Entries are copied GI2D_FWD_CHUNK at a time into a wave-private buffer, per list three arrays in list order --
// A[e] = (gx gy ha hb), B[e] = (cr op cg hc), C[e] = (cb lim) [+ (k -) where final_idx is wanted] -- written with the
// same wide stores they are read with; the right list's arrays sit 32 bytes out of phase with the left one's, so the
// four addresses of one read (two halves x two parities) never share a bank.  `lim` is the pair test of
// gi2d_common.h::AlphaRule (one unsigned compare per pair); a wave none of whose entries can exceed alpha = 1 runs the
// trips without the two v_min (bit-identical: the min is the identity there).  Every forward kernel (plain, fast,
// single-pass) runs this one routine: identical pixels bit for bit.
#define GI2D_FWD_DUMMY GI2D_TILE_LIST_CAP /* index of a never-contributing entry used as list padding */
#ifndef GI2D_FWD_CHUNK
#define GI2D_FWD_CHUNK 32                 /* list entries per half copied per trip */
#endif
static_assert(GI2D_FWD_CHUNK == 32, "a chunk is what the 32 lanes of a half wave copy in one go");
#define GI2D_FWD_CW(FIDX) ((FIDX) ? 4 : 2)                                   /* floats of a staged entry's C part */
#define GI2D_FWD_LISTF(FIDX) (GI2D_FWD_CHUNK * (8 + GI2D_FWD_CW(FIDX)))      /* floats of one staged list: A, B, C */
#define GI2D_FWD_HALF_OF(FIDX) (GI2D_FWD_LISTF(FIDX) + 8)                    /* floats from the left list to the right one (incl. bank shift) */
#define GI2D_FWD_PAIRBUF_OF(FIDX) (GI2D_FWD_HALF_OF(FIDX) + GI2D_FWD_LISTF(FIDX)) /* floats per wave */
#define GI2D_FWD_PAIRBUF GI2D_FWD_PAIRBUF_OF(false)
#define GI2D_FWD_LISTLEN (GI2D_TILE_LIST_CAP + 8)
static_assert(GI2D_FWD_PAIRBUF_OF(false) % 4 == 0 && GI2D_FWD_PAIRBUF_OF(true) % 4 == 0, "every wave's buffer is 16-byte aligned");

struct FwdRec {  // one staged entry as the pixel loop consumes it (conic pre-scaled: scale_conic)
    float gx, gy, ha, hb, hc, op, cr, cg, cb;
    unsigned lim;  // AlphaRule::lim
};

// column (0..15) of the pixel the lane at position r of a 16-lane row holds once the forward has run
__device__ __forceinline__ int fwd_lane_col(int lane) {
    const int r = lane & 15;
    return ((r & 4) << 1) | ((r & 3) << 1) | (r >> 3);
}
// ... and the position in the row of the lane that holds column c
__device__ __forceinline__ int fwd_col_lane(int c) { return ((c & 1) << 3) | ((c >> 3) << 2) | ((c >> 1) & 3); }

#ifndef GI2D_FWD_UNROLL
#define GI2D_FWD_UNROLL 2 /* trips per loop body: two let the LDS reads of one trip overlap the arithmetic of the other */
#endif
// `mine_a`, `mine_c`: this lane's first entry of the chunk in the A and C arrays of its half's list (the B array lies
// 4 CHUNK floats behind A); m: entries of the chunk (both parities); px2: x of the lane's two pixels, py: their y.
// a0..a2: (pixel A, pixel B) per channel.
template <bool NEED_FIDX, bool CLAMP, int UNROLL = GI2D_FWD_UNROLL>
__device__ __forceinline__ void fwd_trips(const float *mine_a, const float *mine_c, int m, const v2f px2, const float py,
                                          v2f &a0, v2f &a1, v2f &a2, int &last_a, int &last_b) {
    constexpr int CW = GI2D_FWD_CW(NEED_FIDX);
    const float4 *qa = reinterpret_cast<const float4 *>(mine_a);
    const float4 *qb = reinterpret_cast<const float4 *>(mine_a + 4 * GI2D_FWD_CHUNK);
#pragma unroll UNROLL
    for (int t = 0; t < m; t += 2) {
        const float4 q0 = qa[t], q1 = qb[t];  // entry t + parity: the arrays are in list order, one float4 per entry
        float cb, kf = 0.f;
        unsigned lim;
        if (NEED_FIDX) {
            const float4 q2 = *reinterpret_cast<const float4 *>(mine_c + t * CW);
            cb = q2.x, lim = (unsigned)__float_as_int(q2.y), kf = q2.z;
        } else {
            const float2 q2 = *reinterpret_cast<const float2 *>(mine_c + t * CW);
            cb = q2.x, lim = (unsigned)__float_as_int(q2.y);
        }
        // (B is staged as (cr, op, cg, hc): what a packed fma broadcasts sits in the low half of a register pair -- from
        // the high half the compiler copies it first; the packed multiply takes `op` from either half)
        const float gx = q0.x, gy = q0.y, ha = q0.z, hb = q0.w, cr = q1.x, op = q1.y, cg = q1.z, hc = q1.w;
        // the row terms, once for the two pixels: == row_term_b / row_term_c
        const float dy = gy - py;
        const float bdy = hb * dy, cdy2 = __builtin_fmaf(hc * dy, dy, 0.f);
        const v2f dx = (v2f){gx, gx} - px2;
        const v2f sig = __builtin_elementwise_fma(dx, __builtin_elementwise_fma((v2f){ha, ha}, dx, (v2f){bdy, bdy}),
                                                  (v2f){cdy2, cdy2});
        const v2f vis = {pair_vis(sig.x), pair_vis(sig.y)};
        const v2f tt = (v2f){op, op} * vis;
        // forward.cu:539-541
        const bool ok0 = CLAMP ? pair_lands_odd(sig.x, tt.x, lim) : pair_lands(sig.x, lim);
        const bool ok1 = CLAMP ? pair_lands_odd(sig.y, tt.y, lim) : pair_lands(sig.y, lim);
        v2f am = {ok0 ? tt.x : 0.f, ok1 ? tt.y : 0.f};
        if (CLAMP) am = (v2f){fminf(1.f, am.x), fminf(1.f, am.y)};
        a0 = __builtin_elementwise_fma((v2f){cr, cr}, am, a0);
        a1 = __builtin_elementwise_fma((v2f){cg, cg}, am, a1);
        a2 = __builtin_elementwise_fma((v2f){cb, cb}, am, a2);
        if (NEED_FIDX) {  // entries ascend within a list: the last one that lands is the largest
            last_a = ok0 ? __float_as_int(kf) : last_a;
            last_b = ok1 ? __float_as_int(kf) : last_b;
        }
    }
}

// keep + (the partner's `send`): row_ror:8 swaps a row's two parities (`old` = send: a rotation within the row always
// has a source lane, nothing is left to fill in)
__device__ __forceinline__ float fwd_add_partner(float keep, float send) {
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(send), __float_as_int(send), 0x128, 0xf, 0xf, false));
}
__device__ __forceinline__ int fwd_max_partner(int keep, int send) {
    return max(keep, __builtin_amdgcn_update_dpp(send, send, 0x128, 0xf, 0xf, false));
}

// lists: [2][GI2D_FWD_LISTLEN] bytes of this wave (left, right); buf: GI2D_FWD_PAIRBUF_OF(NEED_FIDX) floats of this
// wave (16-byte aligned).  cull_of(k) -> cull_word of entry k, rec_of(k) -> FwdRec of entry k (k == GI2D_FWD_DUMMY must
// give an entry that never lands: lim 0).  tx0: x of the tile's first column, py: y of this lane's pixel row.  Returns,
// for the pixel (tx0 + fwd_lane_col(lane), py), the colour in o0..o2 and, with NEED_FIDX, the last contributing entry
// (-1: none).
// CAP: entries the caller's staging arrays hold (the lists are CAP + 8 bytes each, entry CAP is the padding entry).
template <bool NEED_FIDX, int CAP = GI2D_TILE_LIST_CAP, int UNROLL = GI2D_FWD_UNROLL, class CullOf, class RecOf>
__device__ __forceinline__ void fwd_pixel_half_lists(unsigned char *lists, float *buf, int len, CullOf cull_of,
                                                     RecOf rec_of, float tx0, float py, float &o0, float &o1,
                                                     float &o2, int &last_k) {
    constexpr int CW = GI2D_FWD_CW(NEED_FIDX), HALF = GI2D_FWD_HALF_OF(NEED_FIDX);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned char *left = lists, *right = lists + (CAP + 8);
    int n_left = 0, n_right = 0;
    bool clamp_any = false;  // wave-uniform
    for (int base = 0; base < len; base += 64) {
        const int k = base + lane;
        const unsigned w = k < len ? cull_of(k) : 0u;
        const bool reach = (w >> wv) & 1u;
        const bool tl = reach && ((w >> 16) & 15u) <= 7u;  // first column c0 <= 7: touches columns 0..7
        const bool tr = reach && ((w >> 20) & 15u) >= 8u;  // last column c1 >= 8: touches columns 8..15
        const unsigned long long ml = __ballot(tl), mr = __ballot(tr);
        if (tl) left[n_left + __popcll(ml & lanemask_lt())] = (unsigned char)k;
        if (tr) right[n_right + __popcll(mr & lanemask_lt())] = (unsigned char)k;
        n_left += __popcll(ml);
        n_right += __popcll(mr);
        clamp_any = clamp_any || __ballot(reach && (w & GI2D_CULL_CLAMP)) != 0ull;
    }
    __builtin_amdgcn_wave_barrier();  // wave-private lists: DS ops of one wave complete in order

    const int r = lane & 15;
    const int parity = r >> 3, my_half = (r >> 2) & 1;  // which entries of which list this lane walks
    const float *mine_a = buf + my_half * HALF + parity * 4;
    const float *mine_c = buf + my_half * HALF + 8 * GI2D_FWD_CHUNK + parity * CW;
    const int bh = lane >> 5, be = lane & 31;      // build role: lanes 0-31 copy left entries, 32-63 right entries
    const unsigned char *blist = bh ? right : left;
    const int bcnt = bh ? n_right : n_left;
    float *bdst = buf + bh * HALF;
    const int n_max = max(n_left, n_right);
    v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
    const float pxa = tx0 + (float)(8 * my_half + 2 * (r & 3));  // small integers: exact
    const v2f px2 = {pxa, pxa + 1.f};
    int last_a = -1, last_b = -1;
    for (int c0 = 0; c0 < n_max; c0 += GI2D_FWD_CHUNK) {
        {
            const int e = c0 + be;
            const int k = e < bcnt ? (int)blist[e] : CAP;  // entry CAP: the never-landing padding entry
            FwdRec rec = rec_of(k);
            // (all of the entry in registers before the first store: source and destination are both LDS, and left to
            // itself the compiler copies in 8-byte pieces, each read waited for before its write is issued -- five
            // dependent LDS round trips per chunk)
            asm volatile("" : "+v"(rec.gx), "+v"(rec.gy), "+v"(rec.ha), "+v"(rec.hb), "+v"(rec.hc), "+v"(rec.op), "+v"(rec.cr),
                         "+v"(rec.cg), "+v"(rec.cb), "+v"(rec.lim));
            reinterpret_cast<float4 *>(bdst)[be] = make_float4(rec.gx, rec.gy, rec.ha, rec.hb);
            reinterpret_cast<float4 *>(bdst + 4 * GI2D_FWD_CHUNK)[be] = make_float4(rec.cr, rec.op, rec.cg, rec.hc);
            if (NEED_FIDX)
                reinterpret_cast<float4 *>(bdst + 8 * GI2D_FWD_CHUNK)[be] =
                    make_float4(rec.cb, __int_as_float((int)rec.lim), __int_as_float(k), 0.f);
            else
                reinterpret_cast<float2 *>(bdst + 8 * GI2D_FWD_CHUNK)[be] = make_float2(rec.cb, __int_as_float((int)rec.lim));
        }
        __builtin_amdgcn_wave_barrier();
        const int m = min(GI2D_FWD_CHUNK, n_max - c0);
        if (clamp_any)
            fwd_trips<NEED_FIDX, true, UNROLL>(mine_a, mine_c, m, px2, py, a0, a1, a2, last_a, last_b);
        else
            fwd_trips<NEED_FIDX, false, UNROLL>(mine_a, mine_c, m, px2, py, a0, a1, a2, last_a, last_b);
        __builtin_amdgcn_wave_barrier();
    }
    // even + odd list positions of each pixel (the two partial sums every earlier layout formed; the sum commutes).
    // A lane keeps the pixel whose column has its parity and hands the other one's sum to its partner: one select
    // each way and one DPP add per channel.
    o0 = fwd_add_partner(parity ? a0.y : a0.x, parity ? a0.x : a0.y);
    o1 = fwd_add_partner(parity ? a1.y : a1.x, parity ? a1.x : a1.y);
    o2 = fwd_add_partner(parity ? a2.y : a2.x, parity ? a2.x : a2.y);
    last_k = -1;
    if (NEED_FIDX) last_k = fwd_max_partner(parity ? last_b : last_a, parity ? last_a : last_b);
}

// A tile's RGB leaves as one 12-byte store per lane: the 16 lanes of a pixel row write 192 contiguous bytes = three
// whole 64-byte lines in one instruction (in the lane order of fwd_lane_col), which is as good for the memory system as
// 16-byte stores of rows transposed through LDS (the form of rounds 1-3) and costs a third of its instructions.
__device__ __forceinline__ void fwd_store_pixels(float o0, float o1, float o2, int tx, int ty, int img_w, int img_h,
                                                 float *__restrict__ out_img) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lx = fwd_lane_col(lane), ly = wv * 4 + (lane >> 4);
    const int j = tx * GI2D_TILE + lx, i = ty * GI2D_TILE + ly;
    if (i < img_h && j < img_w) {
        const size_t pix = (size_t)i * img_w + j;
        out_img[3 * pix + 0] = o0;
        out_img[3 * pix + 1] = o1;
        out_img[3 * pix + 2] = o2;
    }
}
// The same WRITTEN THROUGH (store16<true>): whole 16-byte pieces, so the wave's four pixel rows (4 x 192 bytes) go
// through `buf` -- 192 floats of the wave's own pair buffer, free once its trips are done -- and lanes 0..47 store one
// float4 each.  Needs rows that start on 16 bytes and a tile that lies inside the image (the caller checks: img_w % 4 == 0,
// full tile); otherwise fwd_store_pixels.
__device__ __forceinline__ void fwd_store_pixels_wt(float o0, float o1, float o2, int tx, int ty, int img_w,
                                                    float *buf, float *__restrict__ out_img) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *mine = buf + 48 * (lane >> 4) + 3 * fwd_lane_col(lane);
    mine[0] = o0, mine[1] = o1, mine[2] = o2;
    __builtin_amdgcn_wave_barrier();
    if (lane < 48) {
        const int row = lane / 12, piece = lane - 12 * row;
        const float4 v = *reinterpret_cast<const float4 *>(buf + 48 * row + 4 * piece);
        const size_t pix = (size_t)(ty * GI2D_TILE + wv * 4 + row) * img_w + tx * GI2D_TILE;
        store16<true>(reinterpret_cast<float4 *>(out_img + 3 * pix) + piece, v, out_img);
    }
}

// LDS of the stand-alone forward kernels: staged entries (conic pre-scaled), their cull words, per-wave lists and
// pair buffers.
struct FwdLds {
    float4 AB[2 * (GI2D_TILE_LIST_CAP + 1)];  // [k]: (gx, gy, ha, hb), (hc, opac, cr, cg)
    float2 C[GI2D_TILE_LIST_CAP + 2];         // cb, lim (AlphaRule)
    unsigned cullw[GI2D_TILE_LIST_CAP];       // cull_word() of the entry
    unsigned char lists[4][2 * GI2D_FWD_LISTLEN];
    float4 pairbuf[GI2D_FWD_PAIRBUF_OF(true)];  // 4 waves x GI2D_FWD_PAIRBUF_OF(true) floats
};

// phase 1 helper: lane `k` publishes its gaussian (list position k of the tile)
__device__ __forceinline__ void fwd_stage_entry(FwdLds &sm, int k, const GaussRec &r, unsigned cull, unsigned lim) {
    const ConicS s = scale_conic(r.a, r.b, r.c);
    sm.AB[2 * k] = make_float4(r.gx, r.gy, s.ha, s.hb);
    sm.AB[2 * k + 1] = make_float4(s.hc, r.opac, r.cr, r.cg);
    sm.C[k] = make_float2(r.cb, __int_as_float((int)lim));
    sm.cullw[k] = cull;
}
__device__ __forceinline__ void fwd_stage_dummy(FwdLds &sm) {
    // padding entry: opacity 0 and lim 0 -> never lands
    sm.AB[2 * GI2D_FWD_DUMMY] = make_float4(0.f, 0.f, 0.f, 0.f);
    sm.AB[2 * GI2D_FWD_DUMMY + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    sm.C[GI2D_FWD_DUMMY] = make_float2(0.f, 0.f);
}

// phases 2-4 of the forward for one tile whose `len` (<= 256) entries are staged in ascending order.
// Must be called by all 256 lanes after a __syncthreads() that follows the staging.
// NEED_FIDX=false (fast path: nobody consumes final_idx) drops the per-pair index tracking and the store.
// WT: the image leaves written through (store16) where its rows allow 16-byte pieces.
template <bool NEED_FIDX = true, bool WT = false>
__device__ __forceinline__ void fwd_rasterize_staged(FwdLds &sm, int len, int list_base, int tx, int ty,
                                                     int img_w, int img_h, bool background_fill,
                                                     const float *__restrict__ background,
                                                     float *__restrict__ final_Ts,
                                                     int32_t *__restrict__ final_idx,
                                                     float *__restrict__ out_img) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lx = fwd_lane_col(lane), ly = wv * 4 + (lane >> 4);  // the pixel this lane holds after the forward
    const int j = tx * GI2D_TILE + lx, i = ty * GI2D_TILE + ly;
    const bool inside = (i < img_h) && (j < img_w);
    float *mybuf = reinterpret_cast<float *>(sm.pairbuf) + wv * GI2D_FWD_PAIRBUF_OF(NEED_FIDX);
    float o0, o1, o2;
    int last_k;
    fwd_pixel_half_lists<NEED_FIDX>(
        sm.lists[wv], mybuf, len, [&](int k) { return sm.cullw[k]; },
        [&](int k) {
            const float4 A = sm.AB[2 * k], B = sm.AB[2 * k + 1];
            FwdRec r;
            r.gx = A.x, r.gy = A.y, r.ha = A.z, r.hb = A.w, r.hc = B.x, r.op = B.y, r.cr = B.z, r.cg = B.w;
            const float2 c = sm.C[k];
            r.cb = c.x, r.lim = (unsigned)__float_as_int(c.y);
            return r;
        },
        (float)(tx * GI2D_TILE), (float)i, o0, o1, o2, last_k);
    int cur_idx = last_k < 0 ? 0 : list_base + last_k;  // forward.cu:497,550: 0 when nothing landed
    if (background_fill) {
        // rasterize_sum_plus.py:110-118: no intersections at all -> image = background
        o0 = background[0];
        o1 = background[1];
        o2 = background[2];
        cur_idx = 0;
    }
    if (inside) {
        const int pix = i * img_w + j;
        if (final_Ts) final_Ts[pix] = 1.f;  // forward.cu:558: T is never updated
        if (NEED_FIDX) final_idx[pix] = cur_idx;
    }
    if (WT && (img_w & 3) == 0 && (reinterpret_cast<uintptr_t>(out_img) & 15) == 0 && (tx + 1) * GI2D_TILE <= img_w &&
        (ty + 1) * GI2D_TILE <= img_h)  // tile-uniform: rows of 16-byte pieces, a tile inside the image
        fwd_store_pixels_wt(o0, o1, o2, tx, ty, img_w, mybuf, out_img);
    else
        fwd_store_pixels(o0, o1, o2, tx, ty, img_w, img_h, out_img);
}

// ========================================================================================== backward
// Work item = (gaussian k, ALIGNED row pair p) for the pixel rows 2p, 2p+1 the gaussian's alpha >= 1/255 box reaches:
// one lane walks the box's COLUMNS c0..c1, one column per trip, the two rows of the pair side by side in packed fp32
// (v_pk_fma_f32 & co: dx is common to the two rows, the row terms b*dy, c*dy^2 are the packed pair), keeping every
// running sum in registers.  There is no row loop and no cross-lane reduction: per row the sums
//   S0 = sum w, S1 = sum w dx, S2 = sum w dx^2,  w = opac*vis*v_alpha = -v_sigma  (backward.cu:948)
// sit in the two halves of three packed registers, and an item hands on six MOMENTS of w over its pixels
//   T0 = sum w, T1 = sum w dx, T2 = sum w dx^2, U0 = sum w dy, U1 = sum w dx dy, V0 = sum w dy^2
// (dx, dy relative to the gaussian's centre, so moments of different items simply add) next to the three colour sums.
// The lane that owns the gaussian adds the moments of its (<= 8) items in row order and turns them into the
// gradients ONCE per (tile, gaussian) -- v_xy = -(a T1 + b U0, b T1 + c U0), v_conic = -1/2 (T2, U1, V0),
// v_opacity = T0 / opacity -- instead of once per row as the first design did (its per-row epilogue and per-item
// division were 40 % of the item loop's instructions).  All items of a gaussian cost the same number of trips (its
// columns), so the length classes that balance a wave are per GAUSSIAN and an item's slot is slot0(gaussian) + j.
// A row of the pair outside the box (first / last pair of a box that starts on an odd / ends on an even row) rides
// along masked.
#ifndef GI2D_BWD_ITEMS
#define GI2D_BWD_ITEMS 256 /* items per round (<= 256 = one per lane) */
#endif
static_assert(GI2D_BWD_ITEMS == 256, "an item's row in the hand-off buffer is its lane");
#ifndef GI2D_BWD_PART_ROWS
#define GI2D_BWD_PART_ROWS 176 /* item rows of the LDS hand-off buffer: most tiles of a 50 000-gaussian 768x512 image
                                  (155 items on average) hand over in one pass; 192 rows (27.1 KB in the single-pass tile
                                  kernel) cost it the sixth workgroup per CU: 128 / 160 / 176 / 192 rows measure
                                  20.6 / 20.4 / 20.2 / 21.9 us */
#endif
#ifndef GI2D_BWD_OCC
#define GI2D_BWD_OCC 5 /* waves per SIMD the register allocator must leave room for; measured: 5 (96 VGPRs) beats 6 (80) */
#endif

template <int PSTR, bool WT = false>
__device__ __forceinline__ void store_partial_row(float4 *__restrict__ dst, const float (&acc)[PSTR], int tag,
                                                  const void *wt_base = nullptr);

// Pixel gradients in LDS: one 32-byte record per (row pair p, column c) at pix[2 * (16 p + c)]:
//   (vox_A, vox_B, voy_A, voy_B) (voz_A, voz_B, fidx_A, fidx_B)        A = row 2p, B = row 2p + 1
// -- exactly the packed operands of an item's trip, two LDS reads (the second one b64 where final_idx is not consulted).
#define GI2D_BWD_PIXRECS (GI2D_TILE / 2 * GI2D_TILE)

template <bool WITH_ABS, bool FIDX = true>
struct BwdLds {
    static constexpr int PSTR = WITH_ABS ? 11 : 9;  // odd: conflict-free hand-off rows
    static constexpr bool HAS_FIDX = FIDX;
    static constexpr bool HAS_RAW = WITH_ABS;
    float4 pix[2 * GI2D_BWD_PIXRECS];
    float4 gA[GI2D_TILE_LIST_CAP];  // gx, gy, ha, hb          (conic pre-scaled: scale_conic)
    float4 gB[GI2D_TILE_LIST_CAP];  // hc, opac, cr, cg
    float2 gC[GI2D_TILE_LIST_CAP];  // cb, lim (gi2d_common.h::AlphaRule)
    float4 gRaw[WITH_ABS ? GI2D_TILE_LIST_CAP : 1];  // a, b, c as given (the |v_xy| sums are per pixel: backward.cu:959)
    unsigned short span[GI2D_TILE_LIST_CAP];        // slot0 | n << 11: the gaussian's items are [slot0, slot0 + n)
    unsigned short item[8 * GI2D_TILE_LIST_CAP];    // k | j << 8: the item's gaussian and which of its row pairs
    static constexpr int PART_ROWS = GI2D_BWD_PART_ROWS;
    float part[PART_ROWS * PSTR];
    unsigned xr[GI2D_TILE_LIST_CAP];  // r0 | r1 << 4 | c0 << 8 | c1 << 12 | clamp << 16 per gaussian
    int wsum[8];
    int n_items;
    // rows / columns of gaussian k's box and its clamp flag: bits 8..24 of its cull word
    __device__ __forceinline__ void set_box(int k, unsigned cull) { xr[k] = cull >> 8; }
    __device__ __forceinline__ unsigned box_of(int k) const { return xr[k]; }
};

// lane (lx, ly) publishes pixel (v_out, final_idx); pixels outside the image carry v_out = 0 and final_idx = -1
template <class Lds>
__device__ __forceinline__ void bwd_publish_pixel(Lds &sm, int lx, int ly, float vx, float vy, float vz, float fi) {
    float *r = reinterpret_cast<float *>(&sm.pix[2 * ((ly >> 1) * GI2D_TILE + lx)]) + (ly & 1);
    r[0] = vx;
    r[2] = vy;
    r[4] = vz;
    if (Lds::HAS_FIDX) r[6] = fi;
}
template <class Lds>
__device__ __forceinline__ void bwd_stage_pixels(Lds &sm, int tx, int ty, int img_w, int img_h,
                                                 const int32_t *__restrict__ final_idx,
                                                 const float *__restrict__ v_output) {
    const int tid = threadIdx.x;
    const int lx = tid & 15, ly = tid >> 4;
    const int j = tx * GI2D_TILE + lx, i = ty * GI2D_TILE + ly;
    float vx = 0.f, vy = 0.f, vz = 0.f, fi = __int_as_float(-1);
    if (i < img_h && j < img_w) {
        const size_t pix = (size_t)i * img_w + j;
        vx = v_output[3 * pix];
        vy = v_output[3 * pix + 1];
        vz = v_output[3 * pix + 2];
        if (Lds::HAS_FIDX) fi = __int_as_float(final_idx ? final_idx[pix] : 0x7fffffff);
    }
    bwd_publish_pixel(sm, lx, ly, vx, vy, vz, fi);
}

template <class Lds>
__device__ __forceinline__ void bwd_stage_entry(Lds &sm, int k, const GaussRec &r, unsigned lim) {
    const ConicS s = scale_conic(r.a, r.b, r.c);
    sm.gA[k] = make_float4(r.gx, r.gy, s.ha, s.hb);
    sm.gB[k] = make_float4(s.hc, r.opac, r.cr, r.cg);
    sm.gC[k] = make_float2(r.cb, __int_as_float((int)lim));
    if constexpr (Lds::HAS_RAW) sm.gRaw[k] = make_float4(r.a, r.b, r.c, 0.f);
}

// After pixels and the first `len` gaussians are staged (no barrier needed before the call): builds the
// item list from each lane's cull word (cull_word()), runs the items, and stores to `dst` (lanes tid < len; three
// float4) the gradient partial of gaussian `tid` for this tile:
//   (v_x, v_y, v_conic[3], v_rgb[3], v_opacity [, sum|v_x|, sum|v_y|]).
// Items are processed 256 per round; nearly every tile needs one round, so the running sums are NOT kept in
// registers across rounds (that costs the register budget of the sixth workgroup per CU): a gaussian whose items
// straddle two rounds re-reads its own row from `dst` in the later round.
// `list_base` + k is the entry's position in the sorted list (compared with final_idx, backward.cu:903).
// USE_FIDX=false (fast path): the forward that produced the lists evaluates every pair with the same
// instructions, so "idx <= final_idx" is implied by the alpha test and is not re-checked (pixels outside the
// image still carry v_out = 0 and contribute nothing).
#ifndef GI2D_BWD_TRACE  /* gi2d_fused_core.h defines it for its phase trace (development aid) */
#define GI2D_BWD_TRACE(i) \
    do {                  \
    } while (0)
#endif

// An item costs (columns of the box) trips of the pixel loop, 1 .. 16, the same for every item of a gaussian.  A wave
// runs as long as its longest item, so a tile's items are handed to the lanes by length class -- 9..16, 5..8, 3..4,
// 1..2 trips, longest first -- instead of in gaussian order (41 % of the issued lane-trips useful in gaussian order,
// 64 % by class; a full sort by length: 66 %).
__device__ __forceinline__ int bwd_len_class(int trips) { return trips >= 9 ? 0 : trips >= 5 ? 1 : trips >= 3 ? 2 : 3; }
struct BwdItemsOf {  // the items of one gaussian
    int n;                      // row pairs, 0 .. 8
    int cls;                    // length class of all of them
    unsigned long long counts;  // n in the 16-bit field of its class (class c in bits 16c .. 16c+15)
};
__device__ __forceinline__ BwdItemsOf bwd_items_of(unsigned cull) {
    BwdItemsOf it;
    it.n = 0, it.cls = 0, it.counts = 0ull;
    if (!(cull & 15u)) return it;
    const int r0 = (int)((cull >> 8) & 15u), r1 = (int)((cull >> 12) & 15u);
    const int nc = (int)((cull >> 20) & 15u) - (int)((cull >> 16) & 15u) + 1;
    it.n = (r1 >> 1) - (r0 >> 1) + 1;
    it.cls = bwd_len_class(nc);
    it.counts = (unsigned long long)(unsigned)it.n << (16 * it.cls);
    return it;
}
// First half of the item scan: this wave's inclusive scan of the per-class counts, its totals published in
// `wsum[2 * wave .. 2 * wave + 1]`.  A caller whose next workgroup barrier comes anyway may run it ahead of that barrier
// and pass the result to bwd_run_tile (PRESCANNED), which then needs one barrier less; wsum must not live in memory
// other waves still use.
__device__ __forceinline__ unsigned long long bwd_prescan(int *wsum, unsigned cull) {
    if (__ballot((cull & 15u) != 0u) == 0ull) {  // a wave without items (short lists leave three of four like that)
        if ((threadIdx.x & 63) == 63) wsum[2 * (threadIdx.x >> 6)] = 0, wsum[2 * (threadIdx.x >> 6) + 1] = 0;
        return 0ull;
    }
    const unsigned long long c = bwd_items_of(cull).counts;
    // the 16-bit fields never carry into each other: a tile has at most 8 * 256 items
    const unsigned lo = (unsigned)wave_inclusive_scan((int)(unsigned)c);
    const unsigned hi = (unsigned)wave_inclusive_scan((int)(unsigned)(c >> 32));
    if ((threadIdx.x & 63) == 63) wsum[2 * (threadIdx.x >> 6)] = (int)lo, wsum[2 * (threadIdx.x >> 6) + 1] = (int)hi;
    return (unsigned long long)hi << 32 | lo;
}
__device__ __forceinline__ int bwd_field_sum(unsigned long long packed) {
    return (int)((packed & 0xffffu) + ((packed >> 16) & 0xffffu) + ((packed >> 32) & 0xffffu) + (packed >> 48));
}

template <bool USE_FIDX>
__device__ __forceinline__ bool fidx_admits(int idx, float fidx_bits) {
    return !USE_FIDX || idx <= __float_as_int(fidx_bits);
}

// What one item accumulates, and the loop over its columns (CLAMP: min(1, .) can bind for some item of this wave, see
// gi2d_common.h::AlphaRule).
struct BwdItemAcc {
    v2f S0, S1, S2, gr, gg, gb, ax, ay;
};
struct BwdItemIn {
    float gx, px0;             // centre x, x coordinate of the first column
    v2f dy, bdy, cdy2;         // the two rows of the pair
    v2f ha2, opac2, cr2, cg2, cb2;
    float4 raw;                // a, b, c as given (WITH_ABS)
    unsigned lim_a, lim_b;     // AlphaRule::lim per row of the pair; 0 ("never lands") for a row outside the box: a lane
                               // mask ANDed into the pair test every trip is a VALU -> SALU -> VALU round trip through
                               // VCC on the loop's critical path (tools/ubench/valu_rate.hip: ~20 cycles a trip)
    int idx;                   // position in the sorted list (USE_FIDX)
    const float4 *rec, *rec_end;
};
template <bool WITH_ABS, bool USE_FIDX, bool CLAMP>
__device__ __forceinline__ void bwd_item_columns(const BwdItemIn &in, BwdItemAcc &o) {
    const float4 *rec = in.rec;
    // pixel x coordinates exactly as the forward forms them: (float)j (small integers: stepping by 1.0 stays exact)
    float px = in.px0;
    // (bottom-tested, every lane-varying update unconditional: the exit is then one compare and three scalar
    // instructions per trip; with the test in the middle of the body the exec-mask bookkeeping was ten)
#pragma unroll 1
    do {
        const float4 P0 = rec[0];
        float4 P1;
        if (USE_FIDX) {
            P1 = rec[1];
        } else {
            const float2 h = *reinterpret_cast<const float2 *>(rec + 1);
            P1 = make_float4(h.x, h.y, 0.f, 0.f);
        }
        const v2f vox = {P0.x, P0.y}, voy = {P0.z, P0.w}, voz = {P1.x, P1.y};
        const float dx1 = in.gx - px;
        px += 1.f;
        const v2f dx = {dx1, dx1};
        // == pair_sigma() of the forward, the two rows of the pair per instruction
        const v2f sig = __builtin_elementwise_fma(dx, __builtin_elementwise_fma(in.ha2, dx, in.bdy), in.cdy2);
        const v2f vis = {pair_vis(sig.x), pair_vis(sig.y)};
        const v2f t = in.opac2 * vis;
        // backward.cu:903 (idx <= final_idx) and :922-926 (sigma < 0 || alpha < 1/255: the forward's pair test)
        const bool l0 = CLAMP ? pair_lands_odd(sig.x, t.x, in.lim_a) : pair_lands(sig.x, in.lim_a);
        const bool l1 = CLAMP ? pair_lands_odd(sig.y, t.y, in.lim_b) : pair_lands(sig.y, in.lim_b);
        const bool ok0 = fidx_admits<USE_FIDX>(in.idx, P1.z) & l0;
        const bool ok1 = fidx_admits<USE_FIDX>(in.idx, P1.w) & l1;
        const v2f tz = {ok0 ? t.x : 0.f, ok1 ? t.y : 0.f};
        v2f am = tz;
        if (CLAMP) am = (v2f){fminf(1.f, tz.x), fminf(1.f, tz.y)};
        // backward.cu:940-946
        const v2f v_alpha =
            __builtin_elementwise_fma(in.cb2, voz, __builtin_elementwise_fma(in.cg2, voy, in.cr2 * vox));
        o.gr = __builtin_elementwise_fma(am, vox, o.gr);
        o.gg = __builtin_elementwise_fma(am, voy, o.gg);
        o.gb = __builtin_elementwise_fma(am, voz, o.gb);
        const v2f w = tz * v_alpha;  // = -v_sigma (backward.cu:948), 0 when the pair is invalid
        o.S0 += w;
        const v2f wdx = w * dx;
        o.S1 += wdx;
        o.S2 = __builtin_elementwise_fma(wdx, dx, o.S2);
        if (WITH_ABS) {  // backward.cu:959-960 (commented in the shipped kernel): sum |v_xy|
            const v2f a2 = {in.raw.x, in.raw.x}, b2 = {in.raw.y, in.raw.y}, c2 = {in.raw.z, in.raw.z};
            const v2f ux = w * __builtin_elementwise_fma(a2, dx, b2 * in.dy);
            const v2f uy = w * __builtin_elementwise_fma(b2, dx, c2 * in.dy);
            o.ax += __builtin_elementwise_abs(ux);
            o.ay += __builtin_elementwise_abs(uy);
        }
        rec += 2;
    } while (rec <= in.rec_end);
}

template <bool WITH_ABS, bool USE_FIDX = true, bool PRESCANNED = false, bool WT = false, class Lds = BwdLds<WITH_ABS, USE_FIDX>>
__device__ __forceinline__ void bwd_run_tile(Lds &sm, int len, unsigned cull, int list_base, float tx0, float ty0,
                                             float4 *__restrict__ dst, unsigned long long prescan_incl = 0ull,
                                             const int *prescan_wsum = nullptr, const void *wt_base = nullptr) {
    constexpr int PSTR = Lds::PSTR;
    static_assert(!USE_FIDX || Lds::HAS_FIDX, "final_idx is not staged in this LDS layout");
    static_assert(!WITH_ABS || Lds::HAS_RAW, "the |v_xy| sums need the conic as given");
    const int tid = threadIdx.x, wv = tid >> 6;
    unsigned long long incl = prescan_incl;
    const int *wsum = prescan_wsum;
    if (!PRESCANNED) {
        incl = bwd_prescan(sm.wsum, cull);
        wsum = sm.wsum;
        __syncthreads();
    }
    // waves without entries have nothing to place (wave 0 publishes the item count)
    const bool placing = (tid & ~63) < len || wv == 0;
    if (placing) {
        // the per-wave totals are the same for every lane: keep them, and everything derived from them, in scalar
        // registers (class c's first slot = items of the classes before it)
        unsigned long long before = 0ull, total = 0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(wsum[2 * k]);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(wsum[2 * k + 1]);
            const unsigned long long w = (unsigned long long)hi << 32 | lo;
            total += w;
            if (k < __builtin_amdgcn_readfirstlane(wv)) before += w;
        }
        const unsigned long long class_base = (total << 16) + (total << 32) + (total << 48);  // field c: classes < c
        const BwdItemsOf mine = bwd_items_of(cull);
        const unsigned long long pos = before + class_base + (incl - mine.counts);
        const int slot0 = (int)((pos >> (16 * mine.cls)) & 0xffffu);
        if (tid < len) {
            sm.set_box(tid, cull);
            sm.span[tid] = (unsigned short)(slot0 | mine.n << 11);
        }
        // the gaussian's item codes, without a divergent loop (its exec-mask bookkeeping was three times the stores): a
        // lane past its count repeats its first store
        const bool more = __ballot(mine.n > 4) != 0ull;  // wave-uniform
        if (tid < len && mine.n > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int jj = j < mine.n ? j : 0;
                sm.item[slot0 + jj] = (unsigned short)(tid | jj << 8);
            }
            if (more) {
#pragma unroll
                for (int j = 4; j < 8; ++j) {
                    const int jj = j < mine.n ? j : 0;
                    sm.item[slot0 + jj] = (unsigned short)(tid | jj << 8);
                }
            }
        }
        if (tid == 0) sm.n_items = bwd_field_sum(total);
    }
    __syncthreads();
    GI2D_BWD_TRACE(7);
    const int n_items = sm.n_items;
#if defined(GI2D_STOP_AFTER) && GI2D_STOP_AFTER == 4 /* development aid: budget of the item build */
    if (n_items >= 0) return;
#endif
    // this lane's gaussian: its items are the slots [my_lo, my_hi)
    int my_lo = 0, my_hi = 0;
    if (dst != nullptr) {
        const int sp = sm.span[tid];
        my_lo = sp & 0x7ff, my_hi = my_lo + (sp >> 11);
    }

    int round0 = 0;
    do {
        const int round1 = min(n_items, round0 + GI2D_BWD_ITEMS);
        const int it = round0 + tid;
        float res[PSTR];
        if (it < round1) {
            const int code = sm.item[it], k = code & 255;
            const unsigned xr = sm.box_of(k);  // r0 | r1 << 4 | c0 << 8 | c1 << 12 | clamp << 16
            const int r0 = (int)(xr & 15u), r1 = (int)((xr >> 4) & 15u);
            const int p = (r0 >> 1) + (code >> 8);               // items of k: its row pairs in order
            const int c_lo = (int)((xr >> 8) & 15u);
            const int c_hi = (int)((xr >> 12) & 15u);
            const float4 A = sm.gA[k], B = sm.gB[k];
            const float2 C = sm.gC[k];
            BwdItemIn in;
            const bool in_a = 2 * p >= r0, in_b = 2 * p + 1 <= r1;  // a box may start on row B / end on row A of a pair
            in.gx = A.x;
            const float gy = A.y;
            ConicS s;
            s.ha = A.z, s.hb = A.w, s.hc = B.x;
            // row terms exactly as the forward forms them: dy = gy - (float)i, b*dy, c*dy*dy
            const float py_a = ty0 + (float)(2 * p);
            in.dy = (v2f){gy - py_a, gy - (py_a + 1.f)};
            in.bdy = (v2f){row_term_b(s, in.dy.x), row_term_b(s, in.dy.y)};
            in.cdy2 = (v2f){row_term_c(s, in.dy.x), row_term_c(s, in.dy.y)};
            in.ha2 = (v2f){s.ha, s.ha}, in.opac2 = (v2f){B.y, B.y};
            in.cr2 = (v2f){B.z, B.z}, in.cg2 = (v2f){B.w, B.w}, in.cb2 = (v2f){C.x, C.x};
            in.lim_a = in_a ? (unsigned)__float_as_int(C.y) : 0u;
            in.lim_b = in_b ? (unsigned)__float_as_int(C.y) : 0u;
            in.idx = list_base + k;
            in.raw = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (WITH_ABS) in.raw = sm.gRaw[k];
            in.rec = &sm.pix[2 * (p * GI2D_TILE + c_lo)];
            in.rec_end = in.rec + 2 * (c_hi - c_lo);
            in.px0 = tx0 + (float)c_lo;
            const v2f zero2 = {0.f, 0.f};
            BwdItemAcc o;
            o.S0 = o.S1 = o.S2 = o.gr = o.gg = o.gb = o.ax = o.ay = zero2;
            if (__ballot((xr >> 16) & 1u) != 0ull)  // wave-uniform
                bwd_item_columns<WITH_ABS, USE_FIDX, true>(in, o);
            else
                bwd_item_columns<WITH_ABS, USE_FIDX, false>(in, o);
            const v2f dy = in.dy, S0 = o.S0, S1 = o.S1, S2 = o.S2, gr = o.gr, gg = o.gg, gb = o.gb, ax = o.ax, ay = o.ay;
            // the item's moments (see the head of this section)
            const float u0a = dy.x * S0.x, u0b = dy.y * S0.y;
            res[0] = S1.x + S1.y;                              // T1 = sum w dx
            res[1] = u0a + u0b;                                // U0 = sum w dy
            res[2] = S2.x + S2.y;                              // T2 = sum w dx^2
            res[3] = __builtin_fmaf(dy.x, S1.x, dy.y * S1.y);  // U1 = sum w dx dy
            res[4] = __builtin_fmaf(dy.x, u0a, dy.y * u0b);    // V0 = sum w dy^2
            res[5] = gr.x + gr.y;
            res[6] = gg.x + gg.y;
            res[7] = gb.x + gb.y;
            res[8] = S0.x + S0.y;                              // T0 = sum w = sum opac*vis*v_alpha
            if (WITH_ABS) {
                res[PSTR - 2] = ax.x + ax.y;
                res[PSTR - 1] = ay.x + ay.y;
            }
        }
        GI2D_BWD_TRACE(8);  // this lane's item is done (lane 0: not the slowest one)
#if defined(GI2D_STOP_AFTER) && GI2D_STOP_AFTER == 5 /* development aid: budget of the item loop */
        if (n_items >= 0) {
            float all = 0.f;  // every component stays live: the cut must not let the compiler drop part of the loop
#pragma unroll
            for (int q = 0; q < PSTR; ++q) all += res[q];
            if (it < round1 && all == 12345.678f) sm.part[0] = all;
            return;
        }
#endif
        // hand-off: the lane that owns gaussian `tid` adds the moments of its (<= 8) items of this round, in row
        // order.  The LDS exchange buffer holds PART_ROWS item rows, so a round is handed over in
        // GI2D_BWD_ITEMS / PART_ROWS passes (half the buffer = two more barriers, 4.5 KB less LDS per workgroup).
        constexpr int PROWS = Lds::PART_ROWS;
        const int lo = max(my_lo, round0), hi = min(my_hi, round1);
        // a gaussian without items still owes a zero row
        const bool owner = dst != nullptr && (hi > lo || (round0 == 0 && my_hi == my_lo));
        float acc[PSTR];
#pragma unroll
        for (int q = 0; q < PSTR; ++q) acc[q] = 0.f;
#pragma unroll
        for (int h0 = 0; h0 < GI2D_BWD_ITEMS; h0 += PROWS) {
            if (h0 > 0 && round0 + h0 >= round1) break;  // nothing left in this round (tile-uniform)
            if (tid >= h0 && tid < h0 + PROWS && it < round1) {
                float *out = &sm.part[(tid - h0) * PSTR];
#pragma unroll
                for (int q = 0; q < PSTR; ++q) out[q] = res[q];
            }
            __syncthreads();
            if (owner) {
                const int e0 = max(lo, round0 + h0), e1 = min(hi, round0 + h0 + PROWS);
                for (int e = e0; e < e1; ++e) {
                    const float *in = &sm.part[(e - round0 - h0) * PSTR];
#pragma unroll
                    for (int q = 0; q < PSTR; ++q) acc[q] += in[q];
                }
            }
            // the hand-off buffer is rewritten by the next pass / round: wait, unless this was the last one
            const bool last = (round0 + h0 + PROWS >= round1) && (round0 + GI2D_BWD_ITEMS >= n_items);
            if (!last) __syncthreads();
        }
        GI2D_BWD_TRACE(9);
        if (owner) {
            if (hi > lo) {
                // moments -> gradients, once per (tile, gaussian) and round: the map is linear, so the rounds' rows add.
                // The conic staged in LDS is the pre-scaled one (a log2e / 2, b log2e, c log2e / 2).
                const float4 A = sm.gA[tid], B = sm.gB[tid];
                const float ln2 = 0.6931471805599453f;
                const float T1 = acc[0], U0 = acc[1], opac = B.y;
                acc[0] = -ln2 * __builtin_fmaf(2.f * A.z, T1, A.w * U0);  // -(a T1 + b U0) = sum v_sigma (a dx + b dy)
                acc[1] = -ln2 * __builtin_fmaf(A.w, T1, 2.f * B.x * U0);  // -(b T1 + c U0) = sum v_sigma (b dx + c dy)
                acc[2] *= -0.5f;                                          // sum 0.5 v_sigma dx dx
                acc[3] *= -0.5f;                                          // sum 0.5 v_sigma dx dy
                acc[4] *= -0.5f;                                          // sum 0.5 v_sigma dy dy
                acc[8] = (acc[8] != 0.f) ? acc[8] / opac : 0.f;           // v_opacity = sum vis*v_alpha (backward.cu:961)
            }
            if (my_lo < round0) {  // an earlier round already stored part of this row
                wt_drain<WT>();
                const float4 d0 = dst[0], d1 = dst[1], d2 = dst[2];
                acc[0] += d0.x, acc[1] += d0.y, acc[2] += d0.z, acc[3] += d0.w;
                acc[4] += d1.x, acc[5] += d1.y, acc[6] += d1.z, acc[7] += d1.w;
                acc[8] += d2.x;
                if (PSTR > 9) acc[PSTR - 2] += d2.y, acc[PSTR - 1] += d2.z;
            }
            store_partial_row<PSTR, WT>(dst, acc, tid + 1, wt_base);
        }
        round0 += GI2D_BWD_ITEMS;
    } while (round0 < n_items);
}

// `tag`: the entry's rank in its tile's staged list, plus one -- the spare word of the row hands it to the gaussian's
// lane of the update kernel, which needs it to enter a neighbouring tile through that tile's inbox
// (gi2d_fast_internal.h::Inbox); rows written as zeros elsewhere carry 0 = no rank.
template <int PSTR, bool WT>
__device__ __forceinline__ void store_partial_row(float4 *__restrict__ dst, const float (&acc)[PSTR], int tag,
                                                  const void *wt_base) {
    store16<WT>(dst, make_float4(acc[0], acc[1], acc[2], acc[3]), wt_base);
    store16<WT>(dst + 1, make_float4(acc[4], acc[5], acc[6], acc[7]), wt_base);
    store16<WT>(dst + 2, make_float4(acc[8], PSTR > 9 ? acc[PSTR - 2] : 0.f, PSTR > 9 ? acc[PSTR - 1] : 0.f, __int_as_float(tag)),
                wt_base);
}

__device__ __forceinline__ void add_partial_row(float acc[11], const float4 &p0, const float4 &p1, const float4 &p2);
// STRIDE = float4 per row: 3 (packed 48-byte rows) or 4 (rows padded to one 64-byte line each)
template <int STRIDE = 3>
__device__ __forceinline__ void add_partial(float acc[11], const float4 *__restrict__ partials, size_t row) {
    add_partial_row(acc, partials[STRIDE * row], partials[STRIDE * row + 1], partials[STRIDE * row + 2]);
}
__device__ __forceinline__ void add_partial_row(float acc[11], const float4 &p0, const float4 &p1, const float4 &p2) {
    acc[0] += p0.x;
    acc[1] += p0.y;
    acc[2] += p0.z;
    acc[3] += p0.w;
    acc[4] += p1.x;
    acc[5] += p1.y;
    acc[6] += p1.z;
    acc[7] += p1.w;
    acc[8] += p2.x;
    acc[9] += p2.y;
    acc[10] += p2.z;
}
__device__ __forceinline__ void store_grads(int g, const float acc[11], float2 *v_xy, float *v_conic,
                                            float *v_rgb, float *v_opacity, float4 *v_abs_xy) {
    if (v_abs_xy) v_abs_xy[g] = make_float4(acc[0], acc[1], acc[9], acc[10]);
    v_xy[g] = make_float2(acc[0], acc[1]);
    v_conic[3 * g] = acc[2];
    v_conic[3 * g + 1] = acc[3];
    v_conic[3 * g + 2] = acc[4];
    v_rgb[3 * g] = acc[5];
    v_rgb[3 * g + 1] = acc[6];
    v_rgb[3 * g + 2] = acc[7];
    v_opacity[g] = acc[8];
}

// binary search of gaussian id g in the ascending id list of a tile (entries past the cap carry no gradient)
__device__ __forceinline__ int find_in_tile(const int32_t *__restrict__ gids_sorted,
                                            const int2 *__restrict__ tile_bins, int tile, int rows, int g) {
    if (tile >= rows) return -1;
    const int2 r = tile_bins[tile];
    int lo = r.x, hi = min(r.y, r.x + GI2D_TILE_LIST_CAP);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int v = gids_sorted[mid];
        if (v == g) return mid;
        if (v < g)
            lo = mid + 1;
        else
            hi = mid;
    }
    return -1;
}

}  // namespace gi2d
