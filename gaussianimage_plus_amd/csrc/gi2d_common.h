// Shared device helpers for the gfx950 2D-Gaussian path.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gi2d.h"

#define GI2D_WAVE 64
#define GI2D_BLOCK 256 /* threads per workgroup = one 16x16 tile = 4 waves */

namespace gi2d {

void set_error(const char *msg);
int check_launch(const char *what);

// float -> int with the semantics of CUDA's cvt.rzi.s32.f32 (truncate, saturate, NaN -> 0),
// which get_bbox (helpers.cuh:26-29) relies on for huge / non-finite centres.
__device__ __forceinline__ int cvt_rzi(float x) {
    if (x != x) return 0;
    x = fminf(fmaxf(x, -2147483648.f), 2147483520.f);
    return (int)x;
}

// helpers.cuh:16-50 get_bbox/get_tile_bbox: inclusive min, exclusive max, clamped to the grid.
__device__ __forceinline__ void tile_bbox(float cx, float cy, float pix_radius, int tiles_x,
                                          int tiles_y, int &min_x, int &min_y, int &max_x,
                                          int &max_y) {
#pragma clang fp contract(off)
    const float tcx = cx / (float)GI2D_TILE, tcy = cy / (float)GI2D_TILE;
    const float tr = pix_radius / (float)GI2D_TILE;
    min_x = min(max(0, cvt_rzi(tcx - tr)), tiles_x);
    max_x = min(max(0, cvt_rzi(tcx + tr + 1)), tiles_x);
    min_y = min(max(0, cvt_rzi(tcy - tr)), tiles_y);
    max_y = min(max(0, cvt_rzi(tcy + tr + 1)), tiles_y);
}

// helpers.cuh:179-206 compute_cov2d_bounds.
__device__ __forceinline__ bool cov2d_bounds(float cxx, float cxy, float cyy, float clip_coe,
                                             float &k0, float &k1, float &k2, float &rad_major,
                                             float &rad_minor) {
#pragma clang fp contract(off)
    const float det = cxx * cyy - cxy * cxy;
    if (det == 0.f) return false;
    const float inv_det = 1.f / det;
    k0 = cyy * inv_det;
    k1 = -cxy * inv_det;
    k2 = cxx * inv_det;
    const float b = 0.5f * (cxx + cyy);
    const float s = sqrtf(fmaxf(0.1f, b * b - det));
    const float v1 = b + s, v2 = b - s;
    rad_major = ceilf(clip_coe * sqrtf(fmaxf(v1, v2)));
    rad_minor = ceilf(clip_coe * sqrtf(fminf(v1, v2)));
    return true;
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
    const unsigned lane = threadIdx.x & 63u;
    return lane == 0 ? 0ull : (~0ull >> (64u - lane));
}

// Inclusive scan of one int per lane across the 64-lane wave: six DPP adds (row shifts 1, 2, 4, 8 inside the rows of
// 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into the upper half) -- register to register,
// where the shuffle form (ds_bpermute) pays six dependent LDS round trips.
__device__ __forceinline__ int wave_inclusive_scan(int v) {
#define GI2D_DPP_ADD(ctrl, row_mask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, row_mask, 0xf, false)
    GI2D_DPP_ADD(0x111, 0xf);  // row_shr:1
    GI2D_DPP_ADD(0x112, 0xf);  // row_shr:2
    GI2D_DPP_ADD(0x114, 0xf);  // row_shr:4
    GI2D_DPP_ADD(0x118, 0xf);  // row_shr:8
    GI2D_DPP_ADD(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3
    GI2D_DPP_ADD(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3
#undef GI2D_DPP_ADD
    return v;
}
// Sum of one float per lane over the wave, the same value in every lane (fixed order: the scan's).
__device__ __forceinline__ float wave_sum_dpp(float x) {
    int v = __float_as_int(x);
#define GI2D_DPP_FADD(ctrl, row_mask) \
    v = __float_as_int(__int_as_float(v) + __int_as_float(__builtin_amdgcn_update_dpp(0, v, ctrl, row_mask, 0xf, false)))
    GI2D_DPP_FADD(0x111, 0xf);
    GI2D_DPP_FADD(0x112, 0xf);
    GI2D_DPP_FADD(0x114, 0xf);
    GI2D_DPP_FADD(0x118, 0xf);
    GI2D_DPP_FADD(0x142, 0xa);
    GI2D_DPP_FADD(0x143, 0xc);
#undef GI2D_DPP_FADD
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
// value of `v` in lane `lane` (wave-uniform index): a scalar read, no LDS
__device__ __forceinline__ int wave_read_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// ---- pair evaluation shared (bitwise) by the forward and backward rasterizers -------------
// The conic is pre-scaled by log2(e) so the exponential is a bare v_exp_f32:
//   sigma' = log2e * (0.5*(a dx^2 + c dy^2) + b dx dy) = dx*(ha*dx + hb*dy) + hc*dy*dy.
// sigma' < 0  <=>  sigma < 0 (forward.cu:539).  Both kernels call these two functions with
// the same operands so a pair that contributed in the forward pass contributes in the backward.
struct ConicS {
    float ha, hb, hc;  // 0.5*a*log2e, b*log2e, 0.5*c*log2e
};
__device__ __forceinline__ ConicS scale_conic(float a, float b, float c) {
    const float l2e = 1.4426950408889634f;
    ConicS s;
    s.ha = 0.5f * a * l2e;
    s.hb = b * l2e;
    s.hc = 0.5f * c * l2e;
    return s;
}
__device__ __forceinline__ float row_term_b(const ConicS &s, float dy) { return s.hb * dy; }
// (the closing "+ 0": sigma' must come out as +0, not -0, for a pixel exactly on the centre of a negative conic -- the
// reference's `sigma < 0.f` (forward.cu:539) is false for -0, so that pair lands -- and -0 + +0 = +0 in round-to-nearest,
// while every non-zero product is unchanged; same instruction count: the multiply becomes a multiply-add)
__device__ __forceinline__ float row_term_c(const ConicS &s, float dy) { return __builtin_fmaf(s.hc * dy, dy, 0.f); }
__device__ __forceinline__ float pair_sigma(const ConicS &s, float dx, float bdy, float cdy2) {
    return __builtin_fmaf(dx, __builtin_fmaf(s.ha, dx, bdy), cdy2);
}
__device__ __forceinline__ float pair_vis(float sigma_l2) { return __builtin_amdgcn_exp2f(-sigma_l2); }

// Which pairs of a gaussian land (forward.cu:539-541: `sigma < 0 || alpha < 1/255` are skipped, alpha = min(1, opac *
// vis)).  With sigma' = log2e * sigma the two tests are ONE range test, 0 <= sigma' <= log2(255 * opac), and for
// non-negative floats "<=" is the order of their bit patterns while every negative float (sign bit set) compares
// above all of them as an unsigned integer:
//     lands  <=>  (unsigned)bits(sigma') < lim,      lim = bits(log2(255 * opac)) + 1      (one v_cmp instead of two)
// lim = 0: never (255 * opac < 1, opac <= 0).  A gaussian with a NaN or an infinity among its parameters ("odd") keeps
// the reference's two comparisons as they are written -- every comparison with NaN is false, so a NaN sigma is not
// skipped and alpha = fminf(1, NaN) = 1, while sigma = +-inf (an infinite conic away from the centre) is skipped by one
// test or the other -- with lim = 0xffffffff, which the CLAMP form of the pixel loops reads as "evaluate the
// reference's expression" (pair_lands_odd; such an entry always sets `clamp`).  `clamp`: min(1, .) can bind (opac > 1,
// or odd): with 0 <= opac <= 1 a landing pair has vis <= 1 (v_exp_f32 of a non-positive argument never exceeds 1:
// tools/ubench/exp_le_one.hip sweeps it), so opac * vis <= 1 and the min is the identity -- loops over entries none of
// which needs it run without the two v_min.
struct AlphaRule {
    unsigned lim;
    bool clamp;
};
#define GI2D_LIM_ODD 0xffffffffu
__device__ __forceinline__ AlphaRule alpha_rule(float gx, float gy, float a, float b, float c, float opac) {
    AlphaRule r;
    // NaN, +-inf, inf - inf -- or six finite values whose sum overflows, which the exact comparisons serve just as well
    const float s = gx + gy + a + b + c + opac;
    const bool odd = !(__builtin_fabsf(s) <= 3.4028235e38f);
    r.clamp = odd || !(opac <= 1.f);
    const float smax = __builtin_amdgcn_logf(opac * 255.f);  // v_log_f32 = log2
    r.lim = odd ? GI2D_LIM_ODD : (smax >= 0.f ? (unsigned)__float_as_int(smax) + 1u : 0u);
    return r;
}
__device__ __forceinline__ bool pair_lands(float sigma_l2, unsigned lim) { return (unsigned)__float_as_int(sigma_l2) < lim; }
// the same for the loops that serve entries with `clamp` set: an odd entry is tested as forward.cu:539-541 writes it
// (`alpha_unclamped` = opac * vis)
__device__ __forceinline__ bool pair_lands_odd(float sigma_l2, float alpha_unclamped, unsigned lim) {
    // Integer flags behind an empty asm: written as booleans, the per-lane choice between the two tests comes out of
    // the compiler as divergent branches inside the pixel loops; this way it is two v_cndmask.
    unsigned as_written = (!(sigma_l2 < 0.f) & !(fminf(1.f, alpha_unclamped) < (1.f / 255.f))) ? 1u : 0u;
    unsigned fast = pair_lands(sigma_l2, lim) ? 1u : 0u;
    asm volatile("" : "+v"(as_written), "+v"(fast));
    return (lim == GI2D_LIM_ODD ? as_written : fast) != 0u;
}

// Conservative pixel-space bounding box of {sigma <= ln(255*opac)} (the only place where a
// pair can pass `alpha >= 1/255`, forward.cu:541), widened by a safety margin.  Returns false
// when the gaussian can never contribute; full=true when no finite box exists (non positive
// definite conic or non-finite inputs) and every pixel must be evaluated.
struct CullBox {
    float x0, x1, y0, y1;  // inclusive pixel-coordinate range
};
// Half extents (hx, hy) of that box around the centre: hx < 0 -- the gaussian can never contribute; hx = GI2D_CULL_FULL --
// no finite box exists (non positive definite conic or non-finite inputs) and every pixel must be evaluated.  One
// gaussian's extents do not depend on the tile, so the fast path computes them once per gaussian in its binning step.
#define GI2D_CULL_FULL 3.0e38f
#ifndef GI2D_CULL_MARGIN
#define GI2D_CULL_MARGIN 0.0625f
#endif
__device__ __forceinline__ void cull_extent(float gx, float gy, float a, float b, float c, float opac, float &hx,
                                            float &hy) {
    const float big = GI2D_CULL_FULL;
    hx = hy = big;
    if (!(opac == opac)) return;             // NaN opacity: min(1, NaN) = 1 in the reference
    if (!(opac * 255.f >= 1.f)) {            // alpha <= opac < 1/255 whenever sigma >= 0
        hx = hy = -1.f;
        return;
    }
    const float det = a * c - b * b;
    if (!(a > 0.f) || !(c > 0.f) || !(det > 0.f)) return;  // not PD / NaN: no box
    // hardware sqrt / reciprocal (1 ulp) are enough: what has to be covered is the rounding of the fp32 quadratic
    // form next to the cut-off, which is relative to its terms -- 4e-4 relative + 1e-3 absolute on the threshold,
    // i.e. 2e-4 of the extent -- plus GI2D_CULL_MARGIN pixels of slack (pixels sit at integer coordinates, so every
    // 1/16 px of margin adds 1/16 of a row and of a column to the average box; 0.75 px, the round-1 value, cost
    // a third of all evaluated pairs)
    const float tau2 = 2.f * __logf(opac * 255.f) * 1.0002f + 1e-3f;
    const float t = tau2 * __builtin_amdgcn_rcpf(det);
    const float ex = __builtin_amdgcn_sqrtf(t * c) * 1.0002f + GI2D_CULL_MARGIN;
    const float ey = __builtin_amdgcn_sqrtf(t * a) * 1.0002f + GI2D_CULL_MARGIN;
    if (!(ex < big) || !(ey < big) || !(gx == gx) || !(gy == gy)) return;
    hx = ex;
    hy = ey;
}
__device__ __forceinline__ bool cull_box_of(float gx, float gy, float hx, float hy, CullBox &box) {
    const float big = GI2D_CULL_FULL;
    box.x0 = -big;
    box.x1 = big;
    box.y0 = -big;
    box.y1 = big;
    if (hx < 0.f) return false;
    if (hx >= big) return true;
    box.x0 = gx - hx;
    box.x1 = gx + hx;
    box.y0 = gy - hy;
    box.y1 = gy + hy;
    return true;
}
__device__ __forceinline__ bool cull_box(float gx, float gy, float a, float b, float c, float opac,
                                         CullBox &box) {
    float hx, hy;
    cull_extent(gx, gy, a, b, c, opac, hx, hy);
    return cull_box_of(gx, gy, hx, hy, box);
}

}  // namespace gi2d
