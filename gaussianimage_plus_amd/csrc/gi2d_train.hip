// One whole fitting iteration of GaussianImage++ in four launches ("fused optimizer +
// train-step glue"), for the Cholesky and covariance models with L2 loss and Adam:
//
//   train_project_fill   activations (tanh / +bound: models/gaussianimage_cholesky.py:137,150) + projection +
//                        tile bucket fill
//   fast_fwd_kernel      (gi2d_fast.hip, unchanged)                       -> out_img (pre-clamp)
//   train_bwd_kernel     the backward tile kernel with the loss gradient formed while staging the pixels:
//                        v_out = 2 (clamp(out,0,1) - gt) / (3 H W) inside the clamp range, 0 outside
//                        (models/gaussianimage_cholesky.py:220,305 + models/utils.py:65 "L2"); per-tile sum of
//                        squared errors for the PSNR (:308-309) without a host round trip
//   train_reduce_update  per-gaussian sum of the gradient partials + projection backward + activation
//                        backward + torch.optim.Adam's update (single-tensor form) on xyz / cholesky / colour
//
// The glue the reference runs as ~25 small PyTorch kernels plus two host syncs per iteration
// (models/gaussianimage_cholesky.py:302-317) is folded into the kernels either side of the rasterizer.
#include <cstring>
#include <vector>

#include "gi2d_batch.h"
#include "gi2d_quant_core.h"

namespace gi2d {

// The live population: the host's `n` is an upper bound when the count lives on the device.
__device__ __forceinline__ int live_n(const TrainParams &P, int n) { return P.n_dev ? min(n, *P.n_dev) : n; }

// Whole-row accesses of the [N,2] / [N,3] parameter and moment arrays: one 8- or 12-byte memory instruction per row
// instead of one per float (the arrays may alias as far as the compiler knows, which keeps it from merging them).
// The row types are may_alias: the same memory is also read and written as plain floats (snapshot copies, parked
// entries), and without the attribute type-based alias analysis may move such a float access across a row access.
struct __attribute__((may_alias)) Row3 {
    float a, b, c;
};
struct __attribute__((may_alias, aligned(8))) Row2 {
    float x, y;
};
__device__ __forceinline__ Row3 load_row3(const float *p, int g) { return *reinterpret_cast<const Row3 *>(p + 3 * (size_t)g); }
__device__ __forceinline__ void store_row3(float *p, int g, float a, float b, float c) {
    Row3 r;
    r.a = a, r.b = b, r.c = c;
    *reinterpret_cast<Row3 *>(p + 3 * (size_t)g) = r;
}
__device__ __forceinline__ float2 load_row2(const float *p, int g) {
    const Row2 r = reinterpret_cast<const Row2 *>(p)[g];
    return make_float2(r.x, r.y);
}
__device__ __forceinline__ void store_row2(float *p, int g, float a, float b) {
    Row2 r;
    r.x = a, r.y = b;
    reinterpret_cast<Row2 *>(p)[g] = r;
}

// Activations of one gaussian from its raw parameter rows (already in registers)
template <int KIND>
__device__ __forceinline__ void activate_rows(float2 xy, Row3 raw, const float *bd, float2 &mean, float (&par)[3]) {
    if (KIND == kCholesky)
        mean = make_float2(tanhf(xy.x), tanhf(xy.y));  // get_xyz
    else
        mean = xy;
    par[0] = raw.a + bd[0], par[1] = raw.b + bd[1], par[2] = raw.c + bd[2];  // get_cholesky_elements / get_cov2d_elements
    if (KIND == kScaleRot) {
        // models/gaussianimage_rs.py:166-172: scaling = |_scaling + bound|, rotation = sigmoid(_rotation) * 2 pi;
        // `chol` holds (_scaling.x, _scaling.y, _rotation), `bound` (0.5, 0.5, unused)
        par[0] = fabsf(par[0]);
        par[1] = fabsf(par[1]);
        par[2] = (1.f / (1.f + __expf(-raw.c))) * 6.283185307179586f;
    }
}
template <int KIND>
__device__ __forceinline__ void activate(const TrainParams &P, int g, float2 &mean, float (&par)[3]) {
    activate_rows<KIND>(load_row2(P.xyz, g), load_row3(P.chol, g), P.bound + (size_t)P.bound_stride * g, mean, par);
}
// p1 argument of the projection routines: the rotation of the scale-rot model sits in par[2]
template <int KIND>
__device__ __forceinline__ const float *rot_of(const float (&par)[3]) {
    return KIND == kScaleRot ? &par[2] : nullptr;
}

// Activations + projection + binning step of gaussian `g` of one image (`u.next`: the lists / boxes / records / status
// the binning step works on).
template <int KIND>
__device__ __forceinline__ void train_project_fill_body(int g, const UpdateArgs &u) {
    const TrainParams &P = u.P;
    const int n = live_n(P, u.n);
    begin_binning(g, u.next.status);
    const BinRecs recs = recs_for_binning(u.next.recs, g == 0);
    if (g >= n) return;
    const PrevBox old_box = u.next.prev_box[g];  // with the other inputs, ahead of the stores
    const Row3 col = load_row3(P.feat, g);
    const float opac = P.opacity[g];
    float2 mean;
    float par[3];
    activate<KIND>(P, g, mean, par);
    const ProjOut o = project_one<KIND>(0, u.next.clip_coe, &mean, par, rot_of<KIND>(par), u.img_w, u.img_h, u.tiles_x,
                                        u.tiles_y, u.radius_clip);
    u.xys[g] = o.xy;
    u.radii[g] = o.radius;
    u.conics[3 * g] = o.k0;
    u.conics[3 * g + 1] = o.k1;
    u.conics[3 * g + 2] = o.k2;
    u.next.num_tiles_hit[g] = o.tiles_hit;
    bin_projected(g, o, opac, col.a, col.b, col.c, u.tiles_x, u.tiles_y, u.radius_clip, old_box, u.next.prev_box,
                  u.next.lists, recs);
}
template <int KIND>
__global__ __launch_bounds__(256) void train_project_fill_kernel(UpdateArgs u) {
    train_project_fill_body<KIND>(blockIdx.x * blockDim.x + threadIdx.x, u);
}
// K images in one launch (gi2d_batch.h): workgroup b works on image k with pg_start[k] <= b < pg_start[k + 1]; an
// image's last workgroup is the tile-ordering one of the update kernel and has nothing to do here.
template <int KIND>
__global__ __launch_bounds__(256) void train_project_fill_batched_kernel(const BatchImage *__restrict__ imgs,
                                                                         const int *__restrict__ pg_start,
                                                                         int k_images) {
    const int k = batch_find(pg_start, k_images, (int)blockIdx.x);
    const int local = __builtin_amdgcn_readfirstlane((int)blockIdx.x - pg_start[k]);
    train_project_fill_body<KIND>(local * blockDim.x + threadIdx.x, imgs[k].u);
}

// torch/optim/adam.py::_single_tensor_adam (non-capturable, no amsgrad, no weight decay)
__device__ __forceinline__ float adam(float p, float g, float &m, float &v, const AdamStep &a) {
    m = m + (g - m) * a.one_minus_b1;                 // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.b2 + a.one_minus_b2 * (g * g);          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    return p - a.step_size * (m / denom);             // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// torch.optim.Adam on one gaussian's (xyz, chol, feat) rows; every row is read and written whole.
struct AdamRows {
    float2 x, mx, vx;
    Row3 c, f, mc, vc, mf, vf;
};
// All nine rows of one gaussian.  The update kernel issues these loads before its gradient gather, so they are in
// flight during the gather's dependent memory rounds instead of forming one more round after it.
__device__ __forceinline__ AdamRows adam_load_rows(const TrainParams &P, int g) {
    AdamRows r;
    r.x = load_row2(P.xyz, g), r.mx = load_row2(P.m_xyz, g), r.vx = load_row2(P.v_xyz, g);
    r.c = load_row3(P.chol, g), r.f = load_row3(P.feat, g);
    r.mc = load_row3(P.m_chol, g), r.vc = load_row3(P.v_chol, g), r.mf = load_row3(P.m_feat, g), r.vf = load_row3(P.v_feat, g);
    return r;
}
// The update in registers (r: rows in, updated rows out) ...
__device__ __forceinline__ void adam_update_rows(AdamRows &r, float gx, float gy, const float (&gp)[3],
                                                 const float (&gf)[3], const AdamStep &a_xyz, const AdamStep &a_chol,
                                                 const AdamStep &a_feat) {
    r.x.x = adam(r.x.x, gx, r.mx.x, r.vx.x, a_xyz), r.x.y = adam(r.x.y, gy, r.mx.y, r.vx.y, a_xyz);
    r.c.a = adam(r.c.a, gp[0], r.mc.a, r.vc.a, a_chol), r.c.b = adam(r.c.b, gp[1], r.mc.b, r.vc.b, a_chol),
    r.c.c = adam(r.c.c, gp[2], r.mc.c, r.vc.c, a_chol);
    r.f.a = adam(r.f.a, gf[0], r.mf.a, r.vf.a, a_feat), r.f.b = adam(r.f.b, gf[1], r.mf.b, r.vf.b, a_feat),
    r.f.c = adam(r.f.c, gf[2], r.mf.c, r.vf.c, a_feat);
}
// ... and its nine row stores (issued by the caller where they delay nothing: see fill_diff_begin)
__device__ __forceinline__ void adam_store_rows(const TrainParams &P, int g, const AdamRows &r) {
    store_row2(P.xyz, g, r.x.x, r.x.y);
    store_row2(P.m_xyz, g, r.mx.x, r.mx.y);
    store_row2(P.v_xyz, g, r.vx.x, r.vx.y);
    store_row3(P.chol, g, r.c.a, r.c.b, r.c.c);
    store_row3(P.m_chol, g, r.mc.a, r.mc.b, r.mc.c);
    store_row3(P.v_chol, g, r.vc.a, r.vc.b, r.vc.c);
    store_row3(P.feat, g, r.f.a, r.f.b, r.f.c);
    store_row3(P.m_feat, g, r.mf.a, r.mf.b, r.mf.c);
    store_row3(P.v_feat, g, r.vf.a, r.vf.b, r.vf.c);
}

__device__ __forceinline__ float adan(float p, float g, float &m, float &n, float &d, float &pg, const AdamStep &a);
// Adan on one gaussian's rows, every row read and written whole (five state arrays per parameter group).
__device__ __forceinline__ void adan_rows(const TrainParams &P, int g, float gx, float gy, const float (&gp)[3],
                                          const float (&gf)[3], const AdamStep &a_xyz, const AdamStep &a_chol,
                                          const AdamStep &a_feat, float2 &new_xy, Row3 &new_chol, Row3 &new_feat) {
    const float2 x = load_row2(P.xyz, g);
    float2 m = load_row2(P.m_xyz, g), v = load_row2(P.v_xyz, g), d = load_row2(P.d_xyz, g), pg = load_row2(P.pg_xyz, g);
    const Row3 c = load_row3(P.chol, g), f = load_row3(P.feat, g);
    Row3 mc = load_row3(P.m_chol, g), vc = load_row3(P.v_chol, g), dc = load_row3(P.d_chol, g), pc = load_row3(P.pg_chol, g);
    Row3 mf = load_row3(P.m_feat, g), vf = load_row3(P.v_feat, g), df = load_row3(P.d_feat, g), pf = load_row3(P.pg_feat, g);
    const float nx = adan(x.x, gx, m.x, v.x, d.x, pg.x, a_xyz), ny = adan(x.y, gy, m.y, v.y, d.y, pg.y, a_xyz);
    const float c0 = adan(c.a, gp[0], mc.a, vc.a, dc.a, pc.a, a_chol), c1 = adan(c.b, gp[1], mc.b, vc.b, dc.b, pc.b, a_chol),
                c2 = adan(c.c, gp[2], mc.c, vc.c, dc.c, pc.c, a_chol);
    const float f0 = adan(f.a, gf[0], mf.a, vf.a, df.a, pf.a, a_feat), f1 = adan(f.b, gf[1], mf.b, vf.b, df.b, pf.b, a_feat),
                f2 = adan(f.c, gf[2], mf.c, vf.c, df.c, pf.c, a_feat);
    new_xy = make_float2(nx, ny);
    new_chol.a = c0, new_chol.b = c1, new_chol.c = c2;
    new_feat.a = f0, new_feat.b = f1, new_feat.c = f2;
    store_row2(P.xyz, g, nx, ny);
    store_row2(P.m_xyz, g, m.x, m.y);
    store_row2(P.v_xyz, g, v.x, v.y);
    store_row2(P.d_xyz, g, d.x, d.y);
    store_row2(P.pg_xyz, g, pg.x, pg.y);
    store_row3(P.chol, g, c0, c1, c2);
    store_row3(P.m_chol, g, mc.a, mc.b, mc.c);
    store_row3(P.v_chol, g, vc.a, vc.b, vc.c);
    store_row3(P.d_chol, g, dc.a, dc.b, dc.c);
    store_row3(P.pg_chol, g, pc.a, pc.b, pc.c);
    store_row3(P.feat, g, f0, f1, f2);
    store_row3(P.m_feat, g, mf.a, mf.b, mf.c);
    store_row3(P.v_feat, g, vf.a, vf.b, vf.c);
    store_row3(P.d_feat, g, df.a, df.b, df.c);
    store_row3(P.pg_feat, g, pf.a, pf.b, pf.c);
}

// optimizer.py::_multi_tensor_adan / _single_tensor_adan (weight_decay 0, no gradient clipping), operation by
// operation in fp32.  `pg` holds the previous gradient (the reference keeps its negative, neg_pre_grad).
__device__ __forceinline__ float adan(float p, float g, float &m, float &n, float &d, float &pg, const AdamStep &a) {
    const float diff = a.first ? 0.f : (-pg) + g;     // neg_pre_grad.add_(grad); step 1: neg_pre_grad = -grad
    m = m * a.b1 + g * a.one_minus_b1;                // exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
    d = d * a.b2 + diff * a.one_minus_b2;             // exp_avg_diff.mul_(beta2).add_(diff, alpha=1 - beta2)
    const float u = diff * a.b2 + g;                  // neg_pre_grad.mul_(beta2).add_(grad)
    n = n * a.b3 + a.one_minus_b3 * (u * u);          // exp_avg_sq.mul_(beta3).addcmul_(u, u, value=1 - beta3)
    const float denom = sqrtf(n) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);                // addcdiv_(exp_avg, denom, value=-lr/bc1)
    p = p - a.step_size_diff * (d / denom);           // addcdiv_(exp_avg_diff, denom, value=-lr*beta2/bc2)
    pg = g;                                           // neg_pre_grad = -grad
    return p;
}

// Is the render of THIS step (made with the pre-update parameters) the best so far?  Every WAVE sums the per-tile
// squared errors itself, in the same fixed order -- lane l adds the float4s l, l + 64, ... in ascending order, then one
// DPP wave sum -- so all waves of all workgroups take the same decision without a barrier, an LDS round or a host round
// trip, whatever the workgroup size (64 or 256 lanes, single-image or batched launch).  (The workgroup-wide tree this
// replaces -- one LDS array, nine barriers -- cost the update kernel 3 of its 14 us.)
// Images of more than 2048 tiles would need several dependent rounds of loads per wave that way (six at 2040x1356: the
// update kernel of an adaptive fit at that size took 23 us instead of 12).  Their sum is DEFINED in four chunks -- chunk
// c = float4s [c L, (c + 1) L), L = ceil(T / 16), each summed by one wave as above; total = (s0 + s1) + (s2 + s3) -- which a
// 256-lane workgroup forms with its four waves side by side (one LDS exchange, one barrier) and a 64-lane workgroup one
// chunk after the other: the same bits either way.
// best_sse_loads: the loads, issued with the kernel's other first-round loads; best_decision: the sum and the verdict.
#define GI2D_SSE_ROUND 8 /* float4 loads a lane keeps in flight: 8 x 64 x 4 = 2048 tiles per round */
struct SseLoads {
    float4 v[GI2D_SSE_ROUND];
};
struct SseChunks {  // how the image's float4s are split (chunked: more than one round of loads per wave)
    int n4, len;
    bool chunked, parallel;
};
__device__ __forceinline__ SseChunks sse_chunks(const BestSnap &best) {
    SseChunks c;
    c.n4 = best.num_tiles >> 2;
    c.chunked = c.n4 > 64 * GI2D_SSE_ROUND;
    c.len = c.chunked ? (c.n4 + 3) >> 2 : c.n4;
    c.parallel = c.chunked && blockDim.x == 256;
    return c;
}
// one round of loads of chunk [lo, hi) starting at float4 t0
__device__ __forceinline__ void sse_round(const float4 *__restrict__ sse4, int t0, int hi, float4 (&v)[GI2D_SSE_ROUND]) {
    const int lane = threadIdx.x & 63;
    // Every load unconditional, at a clamped index (a lane past the end re-reads the chunk's last float4 and drops it):
    // with `t < hi ? sse4[t] : 0` each load sat in a block of its own and the compiler closed every block with
    // s_waitcnt vmcnt(0) -- the eight loads, and every other load of the kernel's first round, went one after the
    // other: eight dependent round trips at the top of every wave of the update kernel.
    if (hi <= t0) {  // (nothing to read: wave-uniform)
#pragma unroll
        for (int q = 0; q < GI2D_SSE_ROUND; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
#pragma unroll
    for (int q = 0; q < GI2D_SSE_ROUND; ++q) v[q] = sse4[min(t0 + lane + 64 * q, hi - 1)];
#pragma unroll
    for (int q = 0; q < GI2D_SSE_ROUND; ++q)
        if (t0 + lane + 64 * q >= hi) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float sse_round_sum(const float4 (&v)[GI2D_SSE_ROUND]) {
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < GI2D_SSE_ROUND; ++q) part += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    return part;
}
__device__ __forceinline__ SseLoads best_sse_loads(const BestSnap &best) {
    SseLoads s;
    if (best.sse != nullptr) {
        const SseChunks c = sse_chunks(best);
        const int mine = c.parallel ? (int)(threadIdx.x >> 6) : 0;  // the chunk this wave starts with
        sse_round(reinterpret_cast<const float4 *>(best.tile_sse), mine * c.len, min(c.n4, (mine + 1) * c.len), s.v);
    }
    return s;
}
__device__ __forceinline__ bool best_decision(const BestSnap &best, const SseLoads &first, int n, int g) {
    bool snapshot = false;
    if (best.sse != nullptr) {
        const float4 *sse4 = reinterpret_cast<const float4 *>(best.tile_sse);
        const SseChunks c = sse_chunks(best);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        // chunk `k`, its first round of loads given or not; the (< 4) tiles behind the last float4 ride on the last chunk
        const auto chunk_sum = [&](int k, bool have_first) {
            const int lo = k * c.len, hi = min(c.n4, lo + c.len);
            float part = 0.f;
            int t0 = lo;
            if (have_first) {
                part = sse_round_sum(first.v);
                t0 += 64 * GI2D_SSE_ROUND;
            }
            for (; t0 < hi; t0 += 64 * GI2D_SSE_ROUND) {
                float4 v[GI2D_SSE_ROUND];
                sse_round(sse4, t0, hi, v);
                part += sse_round_sum(v);
            }
            if (!c.chunked || k == 3)
                for (int t = (c.n4 << 2) + lane; t < best.num_tiles; t += 64) part += best.tile_sse[t];
            return wave_sum_dpp(part);
        };
        float total;
        if (!c.chunked) {
            total = chunk_sum(0, true);
        } else {
            float s0, s1, s2, s3;
            if (c.parallel) {
                __shared__ float chunk_s[4];
                const float mine = chunk_sum(wv, true);
                if (lane == 0) chunk_s[wv] = mine;
                __syncthreads();
                s0 = chunk_s[0], s1 = chunk_s[1], s2 = chunk_s[2], s3 = chunk_s[3];
            } else {
                s0 = chunk_sum(0, true), s1 = chunk_sum(1, false), s2 = chunk_sum(2, false), s3 = chunk_sum(3, false);
            }
            total = (s0 + s1) + (s2 + s3);
        }
        const float prev = best.sse[best.step & 1];
        snapshot = total < prev;  // train.py:134 `best_psnr < psnr`
        if (g == 0) {
            best.sse[(best.step + 1) & 1] = snapshot ? total : prev;
            if (snapshot) {
                best.info[0] = n;
                best.info[1] = best.step;
            }
        }
    }
    return snapshot;
}

// Gradient reduce + projection backward + activation backward + optimizer update (+ next iteration's activation,
// projection and binning step) of one image's gaussians: workgroup `block` of the image's launch share, or its extra
// workgroup (`order_block`) that computes the next iteration's tile order (gi2d_fast_internal.h).
// INBOX: entered tiles take the gaussian through their inbox where that is possible (the single-image kernel).
// ALONE: the launch serves one image (its waves have things to wait for and nothing to do meanwhile).
template <int KIND, bool FILL_NEXT, bool ADAN, bool INBOX = false, bool ALONE = INBOX>
__device__ __forceinline__ void train_reduce_update_body(int block, bool order_block, const UpdateArgs &u,
                                                         const AdamStep &a_xyz, const AdamStep &a_chol,
                                                         const AdamStep &a_feat, int step) {
#pragma clang fp contract(off)
    const int tiles_x = u.tiles_x, tiles_y = u.tiles_y;
    if (order_block) {
        // Tile populations drift slowly along a fit: the order is renewed every 16th step, and then unconditionally --
        // what the dealing balances is the SUM of the six tiles a CU holds (371 .. 516 gaussians around a mean of 436
        // on the uniform bench scene when tiles are taken in index order: the slowest CU finishes the tile pass 1.8 us
        // after the median one), which is uneven long before a single tile stands out.
        // (round 6, measured against the identity order, whose speculative row request in the head always hits: the tile
        // pass is the same 18.4 us either way, the iteration 0.2 us slower without the dealing)
        if ((step & 15) == 1) compute_tile_order(u.tile_bins, tiles_x * tiles_y, u.next.tile_order, true);
        return;
    }
    const TrainParams &P = u.P;
    const NextFill &next = u.next;
    const PrevBox *prev_box = u.next.prev_box;
    float2 *xys = u.xys;
    int32_t *radii = u.radii;
    float *conics = u.conics;
    const float img_w = u.img_w, img_h = u.img_h, radius_clip = u.radius_clip;
    float *dbg_grads = u.dbg_grads;
    BestSnap best = u.best;
    best.step = step;
    const int g = block * blockDim.x + threadIdx.x;
    // First round of loads: everything whose address is known now, for every row below the host's upper bound `u.n` (the
    // arrays are that long; with the population on the device the live count is itself one of these loads, and waiting
    // for it first would put a round trip in front of all the others).  The box is the first link of the gradient's
    // load chain (box -> partial rows), radius and conic feed the projection backward.
    const bool in_rows = g < u.n;
    AdamRows rows;
    if (!ADAN && in_rows) rows = adam_load_rows(P, g);
    const PrevBox box_ld = in_rows ? prev_box[g] : no_box();
    const int radius = in_rows ? radii[g] : 0;
    float conic[3] = {0.f, 0.f, 0.f};
    if (in_rows) conic[0] = conics[3 * g], conic[1] = conics[3 * g + 1], conic[2] = conics[3 * g + 2];
    const float opac_next = (FILL_NEXT && in_rows) ? P.opacity[g] : 0.f;
    // ... and the gaussian's first gradient rows, whose addresses need no box (reduce_one)
    // (a single image's kernels only: 9.05 -> 8.65 us at 768x512, 11.5 -> 11.05 at 2040x1356; a batch's launch is short
    // of memory slots, not of things to wait for -- 15.1 -> 15.9 us per image-iteration at K = 24 with four rows ahead)
    constexpr int AHEAD = ALONE ? GI2D_UPDATE_ROWS_AHEAD : 0;
    const RowsAhead<AHEAD> ahead = rows_ahead<AHEAD>(u.partial_g, in_rows ? g : 0);
    // the additive bound of this gaussian (one row for all when bound_stride == 0): used by both activations and the snapshot
    const Row3 bound_row = in_rows ? load_row3(P.bound, P.bound_stride ? g : 0) : Row3{0.f, 0.f, 0.f};
    const float bound3[3] = {bound_row.a, bound_row.b, bound_row.c};
    const int n = live_n(P, u.n);
    const PrevBox pbox = g < n ? box_ld : no_box();
    const int2 box = make_int2(pbox.x, pbox.y);
    const SseLoads sse_first = best_sse_loads(best);
    const bool snapshot = best_decision(best, sse_first, n, g);
    float acc[11];
    // (FILL_NEXT: with the sums come the ranks the tile pass staged this gaussian at -- what it needs to enter a
    // neighbouring tile through that tile's inbox instead of waiting for an atomic's answer: gi2d_fast_internal.h::Inbox)
    InboxFill inbox;
    reduce_one<AHEAD>(g, box, pbox.z, tiles_x * u.tiles_y * GI2D_TILE_LIST_CAP, u.partial_g, u.partial_big,
                                       acc, FILL_NEXT && INBOX ? &inbox.src : nullptr, &ahead);
    if (g >= n) return;
#if defined(GI2D_UPDATE_STOP) && GI2D_UPDATE_STOP == 1 /* development aid: the kernel's time up to the end of its two load rounds */
    {
        float all = rows.x.x + rows.mx.x + rows.vx.y + rows.c.a + rows.f.b + rows.mc.c + rows.vc.a + rows.mf.b + rows.vf.c +
                    conic[0] + conic[1] + conic[2] + opac_next + bound3[0] + (float)radius + (snapshot ? 1.f : 0.f);
#pragma unroll
        for (int q = 0; q < 11; ++q) all += acc[q];
        if (all == 12345.678f) dbg_grads[0] = all;
        return;
    }
#endif
    float2 mean;
    float par[3];
    if (ADAN)
        activate<KIND>(P, g, mean, par);
    else
        activate_rows<KIND>(rows.x, rows.c, bound3, mean, par);
    ProjGrad r;
    r.g11 = r.g12 = r.g22 = r.o0 = r.o1 = r.o2 = 0.f;
    r.v_mean = make_float2(0.f, 0.f);
    // (opaque until here: left to itself the compiler turns `radius > 0` into a lane mask right behind the load -- behind
    // an s_waitcnt vmcnt(0) in the middle of the first load round, with half of that round's loads not yet issued)
    int radius_now = radius;
    asm volatile("" : "+v"(radius_now));
    if (radius_now > 0) {
        const float vc[3] = {acc[2], acc[3], acc[4]};
        r = project_bwd_one<KIND>(0, par, rot_of<KIND>(par), img_w, img_h, conic, make_float2(acc[0], acc[1]), vc);
    }
    // activation backward: tanh' = 1 - tanh^2 (Cholesky model), identity otherwise; the bound is a constant
    float gx = r.v_mean.x, gy = r.v_mean.y;
    if (KIND == kCholesky) {
        gx = gx * (1.f - mean.x * mean.x);
        gy = gy * (1.f - mean.y * mean.y);
    }
    float gp[3] = {r.o0, r.o1, r.o2};
    if (KIND == kScaleRot) {  // through |.| (torch.abs: sign, 0 at 0) and sigmoid * 2 pi
        const float *bd = bound3;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float pre = P.chol[3 * g + q] + bd[q];
            gp[q] = pre > 0.f ? gp[q] : (pre < 0.f ? -gp[q] : 0.f);
        }
        const float sg = par[2] * (1.f / 6.283185307179586f);
        gp[2] = gp[2] * (6.283185307179586f * sg * (1.f - sg));
    }
    const float gf[3] = {acc[5], acc[6], acc[7]};
    if (dbg_grads) {  // [N,8]: gradients w.r.t. the raw parameters (tests)
        float *d = dbg_grads + 8 * (size_t)g;
        d[0] = gx;
        d[1] = gy;
        d[2] = gp[0];
        d[3] = gp[1];
        d[4] = gp[2];
        d[5] = gf[0];
        d[6] = gf[1];
        d[7] = gf[2];
    }
    float2 new_xy;
    Row3 new_chol, new_feat;
    if (ADAN) {
        adan_rows(P, g, gx, gy, gp, gf, a_xyz, a_chol, a_feat, new_xy, new_chol, new_feat);
    } else {
        adam_update_rows(rows, gx, gy, gp, gf, a_xyz, a_chol, a_feat);
        new_xy = rows.x, new_chol = rows.c, new_feat = rows.f;
    }
#if defined(GI2D_UPDATE_STOP) && GI2D_UPDATE_STOP == 2 /* ... up to the optimizer update (nothing stored) */
    {
        const float all = rows.x.x + rows.x.y + rows.mx.x + rows.mx.y + rows.vx.x + rows.vx.y + rows.c.a + rows.c.b + rows.c.c +
                          rows.f.a + rows.f.b + rows.f.c + rows.mc.a + rows.mc.b + rows.mc.c + rows.vc.a + rows.vc.b + rows.vc.c +
                          rows.mf.a + rows.mf.b + rows.mf.c + rows.vf.a + rows.vf.b + rows.vf.c + (snapshot ? 1.f : 0.f);
        if (all == 12345.678f) dbg_grads[0] = all;
        return;
    }
#endif
    // Everything this lane still has to store that does not wait for the binning step's atomics: the optimizer's rows
    // (Adam; Adan stored its own above) and the best-model snapshot -- the state dict after this step's update
    // (train.py:137 copies it after train_iter returned), from the registers the update left, not read back through
    // memory, one 8- or 12-byte store per row.
    const auto store_rest = [&] {
        // (stored right behind the update instead, or written through: both measured, neither faster -- DESIGN.md 3.5)
        if (!ADAN) adam_store_rows(P, g, rows);
        if (snapshot) {
            store_row2(best.xyz, g, new_xy.x, new_xy.y);
            store_row3(best.chol, g, new_chol.a, new_chol.b, new_chol.c);
            store_row3(best.feat, g, new_feat.a, new_feat.b, new_feat.c);
            if (best.bound) store_row3(best.bound, g, bound3[0], bound3[1], bound3[2]);
        }
    };
    if (FILL_NEXT) {
        // same code path as train_project_fill_kernel, on the values just computed
        // (The record-set lookup stays HERE.  Hoisted to the top of the kernel together with begin_binning -- either
        // one alone is fine -- the build keeps one more SGPR alive across the whole kernel, spills SGPRs to VGPR lanes,
        // and a stretch of iterations stops being equal to the same iterations issued one by one: measured,
        // deterministic, and worth nothing in time.)
        begin_binning(g, next.status);
        const BinRecs recs = recs_for_binning(next.recs, g == 0);
        if constexpr (INBOX) {
            inbox.ib.recs = next.inbox;
            // (boxes of at most eight tiles: the ranks reduce_one kept)
            const int old_w = (int)((unsigned)pbox.x >> 16) - (pbox.x & 0xffff), old_h = (int)((unsigned)pbox.y >> 16) - (pbox.y & 0xffff);
            inbox.src.on = old_w > 0 && old_h > 0 && old_w * old_h <= 8;
        }
        // From the rows just computed, still in registers (no store -> load round trip).  The empty asm makes them
        // opaque values, as if loaded: otherwise the compiler fuses the optimizer's last multiply-add into the
        // activation / projection arithmetic in THIS kernel only, and a stretch of iterations issued as one call
        // would no longer be bitwise equal to the same iterations issued one by one (1-ulp differences, measured).
        asm volatile("" : "+v"(new_xy.x), "+v"(new_xy.y), "+v"(new_chol.a), "+v"(new_chol.b), "+v"(new_chol.c));
        float2 mean2;
        float par2[3];
        activate_rows<KIND>(new_xy, new_chol, bound3, mean2, par2);
        const ProjOut o =
            project_one<KIND>(0, next.clip_coe, &mean2, par2, rot_of<KIND>(par2), img_w, img_h, tiles_x, tiles_y,
                              radius_clip);
#if defined(GI2D_UPDATE_STOP) && GI2D_UPDATE_STOP == 3 /* ... up to the next iteration's projection (nothing stored) */
        {
            const float all = o.xy.x + o.xy.y + o.k0 + o.k1 + o.k2 + (float)o.radius + (float)o.tiles_hit + rows.mx.x +
                              rows.vx.y + rows.mc.c + rows.vc.a + rows.mf.b + rows.vf.c + new_feat.a + new_feat.b + new_feat.c;
            if (all == 12345.678f) dbg_grads[0] = all;
            return;
        }
#endif
#if defined(GI2D_UPDATE_STOP) && GI2D_UPDATE_STOP == 6 /* ... up to the record and the box comparison (nothing stored) */
        {
            int mnx, mny, mxx, mxy;
            const bool member = bin_box(o.xy, o.radius, radius_clip, tiles_x, tiles_y, mnx, mny, mxx, mxy) && o.tiles_hit > 0;
            const int2 nw = member ? pack_box(mnx, mny, mxx, mxy) : make_int2(0, 0);
            float4 q[4];
            make_record(q, g, o.xy, o.k0, o.k1, o.k2, opac_next, new_feat.a, new_feat.b, new_feat.c, nw, o.radius, pbox.z);
            float all = (nw.x != pbox.x || nw.y != pbox.y) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) all += q[k].x + q[k].y + q[k].z + q[k].w;
            all += rows.mx.x + rows.vx.y + rows.mc.c + rows.vc.a + rows.mf.b + rows.vf.c + (float)o.tiles_hit;
            if (all == 12345.678f) dbg_grads[0] = all;
            return;
        }
#endif
        // `box` is what prev_box[g] holds: the binning step of THIS iteration left it there (prev_box == next.prev_box).
        // Order of the tail: the binning step's returning atomics (gaussians that entered a tile), then every other
        // store of the lane, then the list stores that need the atomics' results.
        bin_projected<INBOX>(g, o, opac_next, new_feat.a, new_feat.b, new_feat.c, tiles_x, tiles_y, radius_clip, pbox,
                            next.prev_box, next.lists, recs, [&] {
                                xys[g] = o.xy;
                                radii[g] = o.radius;
                                store_row3(conics, g, o.k0, o.k1, o.k2);
                                next.num_tiles_hit[g] = o.tiles_hit;
                                store_rest();
                            }, &inbox);
    } else {
        store_rest();
    }
}

// INBOX: an image of at most GI2D_INBOX_MAX_TILES tiles with another iteration to follow (the launch code picks: a
// kernel of its own, not a run-time switch -- with the switch the box-changing lanes of a large image walked their
// tiles twice, +0.6 us at 2040x1356)
template <int KIND, bool FILL_NEXT, bool ADAN, bool INBOX = false>
__global__ __launch_bounds__(256) void train_reduce_update_kernel(UpdateArgs u, AdamStep a_xyz, AdamStep a_chol,
                                                                  AdamStep a_feat, int step) {
    static_assert(FILL_NEXT || !INBOX, "only a binning update kernel has entrants to deliver");
    train_reduce_update_body<KIND, FILL_NEXT, ADAN, INBOX, true>((int)blockIdx.x, blockIdx.x == gridDim.x - 1, u, a_xyz, a_chol,
                                                           a_feat, step);
}
// K images in one launch (gi2d_batch.h): image k owns workgroups [pg_start[k], pg_start[k + 1]), the last of them its
// tile-ordering workgroup.  All images are at the same optimizer step with the same learning rates.
template <int KIND, bool FILL_NEXT, bool ADAN>
__global__ __launch_bounds__(256) void train_reduce_update_batched_kernel(const BatchImage *__restrict__ imgs,
                                                                          const int *__restrict__ pg_start,
                                                                          int k_images, AdamStep a_xyz, AdamStep a_chol,
                                                                          AdamStep a_feat, int step) {
    const int k = batch_find(pg_start, k_images, (int)blockIdx.x);
    const int local = __builtin_amdgcn_readfirstlane((int)blockIdx.x - pg_start[k]);
    const int last = __builtin_amdgcn_readfirstlane(pg_start[k + 1] - pg_start[k] - 1);
    train_reduce_update_body<KIND, FILL_NEXT, ADAN>(local, local == last, imgs[k].u, a_xyz, a_chol, a_feat, step);
}


// =====================================================================================================================
// Quantisation-aware iteration: GaussianImage_Covariance.train_iter_quantize
// (models/gaussianimage_covariance.py:219-247, forward_quantize :384-410) -- after train_quantize.py's warm-up the
// positions go through an LSQ quantiser, the covariance rows through HybirdQuant (log quantiser on the variances,
// LSQ on the covariance) and the colours through an LSQ quantiser before projection / rasterization, and the twelve
// learned quantiser values (scale, beta per LSQ channel) are trained by their own Adam optimizers.
//
// Launches per iteration: project+fill on the quantised values, the tile pass (colours = dequantised colours), the
// update kernel (gradient reduce, projection backward, quantiser backward, Adam on the gaussians, partial sums for
// everything that needs a whole-array reduction) and one single-workgroup finish kernel that closes the reductions:
//   * v_scale / v_beta of the six LSQ channels -> Adam on the quantiser values;
//   * the log quantiser's range is min()/max() of the CURRENT variances and stays in the autograd graph, so the
//     elements that attain the extremes receive sum-type gradients.  The update kernel cannot finish those elements
//     (their gradient needs the global sums) in its main pass: it parks them in a short list for the closing step;
//   * the log range of the NEXT iteration (min / max / tie counts of the updated variances).

struct QuantVals {
    float xs[2], xb[2], cs, cb, fs[3], fb[3];
    float lbeta, lmax, lscale;
};
__device__ __forceinline__ QuantVals load_quant(const QuantTrain &Q) {
    QuantVals v;
    v.xs[0] = Q.qparams[0], v.xs[1] = Q.qparams[1], v.xb[0] = Q.qparams[2], v.xb[1] = Q.qparams[3];
    v.cs = Q.qparams[4], v.cb = Q.qparams[5];
#pragma unroll
    for (int q = 0; q < 3; ++q) v.fs[q] = Q.qparams[6 + q], v.fb[q] = Q.qparams[9 + q];
    v.lbeta = Q.range[0];
    v.lmax = Q.range[1];
    v.lscale = quant_log_scale(v.lbeta, v.lmax, 0.f, Q.qmax_cov);
    return v;
}

struct QuantRow {
    QuantEval xy[2], cov[3], col[3];
    float covx[3];  // _cov2d + bound: what the covariance quantiser sees
};
__device__ __forceinline__ void quantise_row(const TrainParams &P, const QuantTrain &Q, const QuantVals &v, int g,
                                             QuantRow &r) {
    const float *bd = P.bound + (size_t)P.bound_stride * g;
    const float2 xy = load_row2(P.xyz, g);
    r.xy[0] = quant_eval<GI2D_QUANT_LSQ>(xy.x, v.xs[0], v.xb[0], 0.f, Q.qmax_xy);
    r.xy[1] = quant_eval<GI2D_QUANT_LSQ>(xy.y, v.xs[1], v.xb[1], 0.f, Q.qmax_xy);
    const Row3 raw = load_row3(P.chol, g), col = load_row3(P.feat, g);
    r.covx[0] = raw.a + bd[0], r.covx[1] = raw.b + bd[1], r.covx[2] = raw.c + bd[2];
    r.cov[0] = quant_eval<GI2D_QUANT_LOG>(r.covx[0], v.lscale, v.lbeta, 0.f, Q.qmax_cov);
    r.cov[1] = quant_eval<GI2D_QUANT_LSQ>(r.covx[1], v.cs, v.cb, 0.f, Q.qmax_cov);
    r.cov[2] = quant_eval<GI2D_QUANT_LOG>(r.covx[2], v.lscale, v.lbeta, 0.f, Q.qmax_cov);
    const float cin[3] = {col.a, col.b, col.c};
#pragma unroll
    for (int q = 0; q < 3; ++q) r.col[q] = quant_eval<GI2D_QUANT_LSQ>(cin[q], v.fs[q], v.fb[q], 0.f, Q.qmax_col);
}

// The kernels of a quantisation-aware iteration as functions of (workgroup index within the image, the image's argument
// blocks): the single-image kernels pass blockIdx.x and their kernel arguments, the batched ones their image's entry of
// the batch table (gi2d_batch.h).
__device__ __forceinline__ void project_fill_quant_body(int block, const UpdateArgs &u, const QuantTrain &Q) {
    const TrainParams &P = u.P;
    const int n = live_n(P, u.n);
    const float clip_coe = u.next.clip_coe, img_w = u.img_w, img_h = u.img_h, radius_clip = u.radius_clip;
    const int tiles_x = u.tiles_x, tiles_y = u.tiles_y;
    float2 *xys = u.xys;
    int32_t *radii = u.radii, *num_tiles_hit = u.next.num_tiles_hit, *lists = u.next.lists, *status = u.next.status;
    float *conics = u.conics;
    PrevBox *prev_box = u.next.prev_box;
    const RecSets &rs = u.next.recs;
    const int g = block * blockDim.x + threadIdx.x;
    begin_binning(g, status);
    const BinRecs recs = recs_for_binning(rs, g == 0);
    if (g >= n) return;
    const PrevBox old_box = prev_box[g];  // with the other inputs, ahead of the stores
    const float opac = P.opacity[g];
    const QuantVals v = load_quant(Q);
    QuantRow r;
    quantise_row(P, Q, v, g, r);
    const float2 mean = make_float2(r.xy[0].dequant, r.xy[1].dequant);
    const float par[3] = {r.cov[0].dequant, r.cov[1].dequant, r.cov[2].dequant};
#pragma unroll
    for (int q = 0; q < 3; ++q) Q.qfeat[3 * g + q] = r.col[q].dequant;
    const ProjOut o =
        project_one<kCovariance>(0, clip_coe, &mean, par, nullptr, img_w, img_h, tiles_x, tiles_y, radius_clip);
    xys[g] = o.xy;
    radii[g] = o.radius;
    conics[3 * g] = o.k0;
    conics[3 * g + 1] = o.k1;
    conics[3 * g + 2] = o.k2;
    num_tiles_hit[g] = o.tiles_hit;
    bin_projected(g, o, opac, r.col[0].dequant, r.col[1].dequant, r.col[2].dequant, tiles_x, tiles_y, radius_clip,
                  old_box, prev_box, lists, recs);
}

// (min, count) / (max, count) combination: equal extremes add their counts
__device__ __forceinline__ void range_min_combine(float &m, float &c, float m2, float c2) {
    if (m2 < m) {
        m = m2;
        c = c2;
    } else if (m2 == m) {
        c += c2;
    }
}
__device__ __forceinline__ void range_max_combine(float &m, float &c, float m2, float c2) {
    if (m2 > m) {
        m = m2;
        c = c2;
    } else if (m2 == m) {
        c += c2;
    }
}
__device__ __forceinline__ void wave_range_reduce(float &mn, float &cmn, float &mx, float &cmx) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float a = __shfl_xor(mn, d, 64), ac = __shfl_xor(cmn, d, 64);
        const float b = __shfl_xor(mx, d, 64), bc = __shfl_xor(cmx, d, 64);
        range_min_combine(mn, cmn, a, ac);
        range_max_combine(mx, cmx, b, bc);
    }
}

// One partial row per WAVE (row = the wave's index among the image's waves): fixed-order wave sums, written by the
// wave's first lane.  Rows per wave, not per workgroup, so that the closing kernel's sums do not depend on the workgroup
// size a launch happens to use -- single-image launches pick 64 or 256 lanes by population, a batch by its total.
__device__ __forceinline__ int quant_wave_row(int block) { return block * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6); }
__device__ __forceinline__ void wave_partial_row(float (&sums)[14], float mn, float cmn, float mx, float cmx,
                                                 float *row) {
#pragma unroll
    for (int k = 0; k < 14; ++k) sums[k] = wave_sum(sums[k]);
    wave_range_reduce(mn, cmn, mx, cmx);
    if ((threadIdx.x & 63) == 0) {
        float4 *r4 = reinterpret_cast<float4 *>(row);
        r4[0] = make_float4(sums[0], sums[1], sums[2], sums[3]);
        r4[1] = make_float4(sums[4], sums[5], sums[6], sums[7]);
        r4[2] = make_float4(sums[8], sums[9], sums[10], sums[11]);
        r4[3] = make_float4(sums[12], sums[13], mn, cmn);
        r4[4] = make_float4(mx, cmx, 0.f, 0.f);
    }
}

// Closes the reductions of an iteration (RANGE_ONLY: just the log range, start of a call); one workgroup of any size
// that is a multiple of 64 up to 256.
template <bool RANGE_ONLY>
__device__ __forceinline__ void quant_finish(int blocks, const TrainParams &P, const QuantTrain &Q,
                                             const AdamStep &a_chol, const AdamStep &a_qxy, const AdamStep &a_qcov,
                                             const AdamStep &a_qcol, float *__restrict__ dbg_grads,
                                             const BestSnap &best) {
#pragma clang fp contract(off)
    __shared__ double lane_acc[RANGE_ONLY ? 1 : 256][15];  // 15: odd stride in 8-byte words
    __shared__ double dred[16][16];
    __shared__ double tot[14];
    __shared__ float rred[4][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = (int)blockDim.x >> 6;
    float mn = INFINITY, cmn = 0.f, mx = -INFINITY, cmx = 0.f;
    // Everything that does not depend on the sums is fetched first, so the dependent load chains (parking list ->
    // entry -> its parameter / moments / bound; quantiser values and their moments) overlap the partial-row pass.
    int parked = 0, idx = 0, pg = 0, pq = 0;
    float vt0 = 0.f, p_chol = 0.f, p_m = 0.f, p_v = 0.f, p_bd = 0.f;
    float q_p = 0.f, q_m = 0.f, q_v = 0.f, lbeta = 0.f, lmax = 0.f, n_min = 0.f, n_max = 0.f;
    bool snap = false;
    if (!RANGE_ONLY) {
        // the first round of parked entries is read before the count is known (slots past the count hold stale
        // data that is never used), so count, entries, quantiser values and partial rows are ONE round of loads
        if ((int)threadIdx.x < Q.defer_cap) {
            const float4 *e = reinterpret_cast<const float4 *>(Q.defer + 8 + 8 * threadIdx.x);
            const float4 e0 = e[0], e1 = e[1];
            idx = __float_as_int(e0.x), vt0 = e0.y, p_chol = e0.z, p_m = e0.w, p_v = e1.x, p_bd = e1.y;
        }
        parked = min(Q.defer[0], Q.defer_cap);
        if (threadIdx.x < 12) q_p = Q.qparams[threadIdx.x], q_m = Q.qm[threadIdx.x], q_v = Q.qv[threadIdx.x];
        lbeta = Q.range[0], lmax = Q.range[1], n_min = Q.range[2], n_max = Q.range[3];
        snap = best.sse != nullptr && best.info[1] == best.step;
    }
    // one pass over the partial rows, whole rows per lane (five 16-byte loads in flight), sums in double
    double acc[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) acc[k] = 0.0;
    for (int b = threadIdx.x; b < blocks; b += (int)blockDim.x) {
        const float4 *row = reinterpret_cast<const float4 *>(Q.partial + (size_t)b * GI2D_QT_ROW);
        float4 r[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) r[k] = row[k];
        const float *f = reinterpret_cast<const float *>(r);
        if (!RANGE_ONLY) {
#pragma unroll
            for (int k = 0; k < 14; ++k) acc[k] += (double)f[k];
        }
        range_min_combine(mn, cmn, f[14], f[15]);
        range_max_combine(mx, cmx, f[16], f[17]);
    }
    if (!RANGE_ONLY) {
        // cross-lane sums through LDS in two 16-way steps (fourteen 6-step double shuffle chains cost 4 us here)
#pragma unroll
        for (int k = 0; k < 14; ++k) lane_acc[threadIdx.x][k] = acc[k];
        __syncthreads();
        const int k = threadIdx.x & 15, j = threadIdx.x >> 4, nj = (int)blockDim.x >> 4;
        if (k < 14) {
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) part += lane_acc[j * 16 + i][k];
            dred[j][k] = part;
        }
        __syncthreads();
        if (threadIdx.x < 14) {
            double t = dred[0][threadIdx.x];
            for (int w = 1; w < nj; ++w) t += dred[w][threadIdx.x];
            tot[threadIdx.x] = t;
        }
        __syncthreads();
        // Adam on the twelve quantiser values: qparams = xs[2] xb[2] cs cb fs[3] fb[3]; tot = (s,b) pairs per channel
        if (threadIdx.x < 12) {
            const int k = threadIdx.x;
            int ch, isb;  // channel and which of (scale, beta) this slot is
            if (k < 4) ch = k & 1, isb = k >> 1;
            else if (k < 6) ch = 2, isb = k - 4;
            else ch = 3 + (k - 6) % 3, isb = (k - 6) / 3;
            const float grad = (float)tot[2 * ch + isb];
            const AdamStep &a = k < 4 ? a_qxy : (k < 6 ? a_qcov : a_qcol);
            const float nv = adam(q_p, grad, q_m, q_v, a);
            Q.qparams[k] = nv;
            Q.qm[k] = q_m;
            Q.qv[k] = q_v;
            if (Q.dbg_q) Q.dbg_q[k] = grad;
            if (snap && Q.best_q) Q.best_q[k] = nv;
        }
        // range gradient: scale = (max - beta)/Q is in the graph, so beta gets -v_scale/Q too and max +v_scale/Q,
        // spread evenly over the elements that attain them
        const double qrange = (double)Q.qmax_cov;
        const float e_min = n_min > 0.f ? (float)((tot[13] - tot[12] / qrange) / (double)n_min) : 0.f;
        const float e_max = n_max > 0.f ? (float)((tot[12] / qrange) / (double)n_max) : 0.f;
        if (threadIdx.x == 0 && Q.dbg_q) {
            Q.dbg_q[12] = (float)tot[12];
            Q.dbg_q[13] = (float)tot[13];
            Q.dbg_q[14] = e_min;
            Q.dbg_q[15] = e_max;
        }
        for (int e = threadIdx.x; e < parked; e += (int)blockDim.x) {
            if (e >= (int)blockDim.x) {  // beyond the prefetched first round (rare: more parked entries than lanes)
                const float4 *en = reinterpret_cast<const float4 *>(Q.defer + 8 + 8 * e);
                const float4 e0 = en[0], e1 = en[1];
                idx = __float_as_int(e0.x), vt0 = e0.y, p_chol = e0.z, p_m = e0.w, p_v = e1.x, p_bd = e1.y;
            }
            pg = idx / 3, pq = idx - 3 * pg;
            const float x = p_chol + p_bd, t = quant_log_of(x);
            float vt = vt0;
            if (t == lbeta) vt = vt + e_min;
            if (t == lmax) vt = vt + e_max;
            const float gxv = vt * quant_log_chain(x);
            const float nv = adam(p_chol, gxv, p_m, p_v, a_chol);
            P.chol[idx] = nv;
            P.m_chol[idx] = p_m;
            P.v_chol[idx] = p_v;
            if (dbg_grads) dbg_grads[8 * (size_t)pg + 2 + pq] = gxv;
            if (snap) best.chol[idx] = nv;
            const float t2 = quant_log_of(nv + p_bd);
            range_min_combine(mn, cmn, t2, 1.f);
            range_max_combine(mx, cmx, t2, 1.f);
        }
    }
    wave_range_reduce(mn, cmn, mx, cmx);
    if (lane == 0) rred[wave][0] = mn, rred[wave][1] = cmn, rred[wave][2] = mx, rred[wave][3] = cmx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = rred[0][0], ac = rred[0][1], b = rred[0][2], bc = rred[0][3];
        for (int w = 1; w < waves; ++w) {
            range_min_combine(a, ac, rred[w][0], rred[w][1]);
            range_max_combine(b, bc, rred[w][2], rred[w][3]);
        }
        Q.range[0] = a, Q.range[1] = b, Q.range[2] = ac, Q.range[3] = bc;
        Q.defer[0] = 0;
    }
}

// Range of the variance channels of the current parameters (start of a call, after the host touched them)
__device__ __forceinline__ void quant_range_body(int block, const UpdateArgs &u, const QuantTrain &Q) {
    const TrainParams &P = u.P;
    const int n = live_n(P, u.n);
    const int g = block * blockDim.x + threadIdx.x;
    float sums[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) sums[k] = 0.f;
    float mn = INFINITY, cmn = 0.f, mx = -INFINITY, cmx = 0.f;
    if (g < n) {
        const float *bd = P.bound + (size_t)P.bound_stride * g;
#pragma unroll
        for (int q = 0; q < 3; q += 2) {
            const float t = quant_log_of(P.chol[3 * g + q] + bd[q]);
            range_min_combine(mn, cmn, t, 1.f);
            range_max_combine(mx, cmx, t, 1.f);
        }
    }
    wave_partial_row(sums, mn, cmn, mx, cmx, Q.partial + (size_t)quant_wave_row(block) * GI2D_QT_ROW);
}

template <bool ALONE>
__device__ __forceinline__ void reduce_update_quant_body(int block, const UpdateArgs &u, const QuantTrain &Q,
                                                         const AdamStep &a_xyz, const AdamStep &a_chol,
                                                         const AdamStep &a_feat, int step) {
#pragma clang fp contract(off)
    const TrainParams &P = u.P;
    const int n = live_n(P, u.n);
    const float img_w = u.img_w, img_h = u.img_h;
    const int tiles_x = u.tiles_x, tiles_y = u.tiles_y;
    const int32_t *radii = u.radii;
    const float *conics = u.conics;
    const PrevBox *prev_box = u.next.prev_box;
    const float4 *partial_g = u.partial_g, *partial_big = u.partial_big;
    float *dbg_grads = u.dbg_grads;
    BestSnap best = u.best;
    best.step = step;
    const int g = block * blockDim.x + threadIdx.x;
    int32_t *status = u.next.status;
    // (one image alone: its first gradient rows are asked for before the box says how many count -- reduce_one)
    constexpr int AHEAD = ALONE ? GI2D_UPDATE_ROWS_AHEAD : 0;
    const RowsAhead<AHEAD> ahead = rows_ahead<AHEAD>(partial_g, g < u.n ? g : 0);
    const PrevBox box_ld = g < u.n ? prev_box[g] : no_box();
    const bool snapshot = best_decision(best, best_sse_loads(best), n, g);
    float acc[11];
    const PrevBox pbox = g < n ? box_ld : no_box();
    reduce_one<AHEAD>(g, make_int2(pbox.x, pbox.y), pbox.z, tiles_x * tiles_y * GI2D_TILE_LIST_CAP, partial_g, partial_big,
                      acc, nullptr, &ahead);
    float sums[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) sums[k] = 0.f;
    float mn = INFINITY, cmn = 0.f, mx = -INFINITY, cmx = 0.f;
    if (g < n) {
        const QuantVals v = load_quant(Q);
        QuantRow qr;
        quantise_row(P, Q, v, g, qr);
        const float par[3] = {qr.cov[0].dequant, qr.cov[1].dequant, qr.cov[2].dequant};
        ProjGrad r;
        r.g11 = r.g12 = r.g22 = r.o0 = r.o1 = r.o2 = 0.f;
        r.v_mean = make_float2(0.f, 0.f);
        if (radii[g] > 0) {
            const float conic[3] = {conics[3 * g], conics[3 * g + 1], conics[3 * g + 2]};
            const float vc[3] = {acc[2], acc[3], acc[4]};
            r = project_bwd_one<kCovariance>(0, par, nullptr, img_w, img_h, conic, make_float2(acc[0], acc[1]), vc);
        }
        // quantiser backward: sums[0..11] = (v_scale, v_beta) of xy.x, xy.y, cov, colour r, g, b; sums[12,13] = log
        const float gx = quant_grad<GI2D_QUANT_LSQ>(qr.xy[0], r.v_mean.x, v.xs[0], sums[0], sums[1]);
        const float gy = quant_grad<GI2D_QUANT_LSQ>(qr.xy[1], r.v_mean.y, v.xs[1], sums[2], sums[3]);
        const float go[3] = {r.o0, r.o1, r.o2};
        float gp[3];
        gp[1] = quant_grad<GI2D_QUANT_LSQ>(qr.cov[1], go[1], v.cs, sums[4], sums[5]);
        float gf[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            gf[q] = quant_grad<GI2D_QUANT_LSQ>(qr.col[q], acc[5 + q], v.fs[q], sums[6 + 2 * q], sums[7 + 2 * q]);
        bool parked[3] = {false, false, false};
#pragma unroll
        for (int q = 0; q < 3; q += 2) {
            const float vt = quant_grad<GI2D_QUANT_LOG>(qr.cov[q], go[q], v.lscale, sums[12], sums[13]);
            const float t = quant_log_of(qr.covx[q]);
            gp[q] = vt * quant_log_chain(qr.covx[q]);
            if (t == v.lbeta || t == v.lmax) {  // its gradient also needs the global sums: the finish kernel's job
                parked[q] = true;
                const int slot = atomicAdd(&Q.defer[0], 1);
                if (slot < Q.defer_cap) {  // the entry carries its operands: one load round in the finish kernel
                    float4 *e = reinterpret_cast<float4 *>(Q.defer + 8 + 8 * slot);
                    e[0] = make_float4(__int_as_float(3 * g + q), vt, P.chol[3 * g + q], P.m_chol[3 * g + q]);
                    e[1] = make_float4(P.v_chol[3 * g + q], P.bound[(size_t)P.bound_stride * g + q], 0.f, 0.f);
                } else {
                    atomicOr(&status[2], 2);
                }
            }
        }
        if (dbg_grads) {
            float *d = dbg_grads + 8 * (size_t)g;
            d[0] = gx, d[1] = gy, d[2] = gp[0], d[3] = gp[1], d[4] = gp[2], d[5] = gf[0], d[6] = gf[1], d[7] = gf[2];
        }
        // Adam on whole rows; a parked variance keeps its value and moments (the finish kernel updates it)
        const float *bd = P.bound + (size_t)P.bound_stride * g;
        {
            const float2 x = load_row2(P.xyz, g);
            float2 m_xy = load_row2(P.m_xyz, g), v_xy = load_row2(P.v_xyz, g);
            const Row3 c = load_row3(P.chol, g), f = load_row3(P.feat, g);
            Row3 mc = load_row3(P.m_chol, g), vc = load_row3(P.v_chol, g), mf = load_row3(P.m_feat, g),
                 vf = load_row3(P.v_feat, g);
            const float nx = adam(x.x, gx, m_xy.x, v_xy.x, a_xyz), ny = adam(x.y, gy, m_xy.y, v_xy.y, a_xyz);
            float cn[3] = {c.a, c.b, c.c}, cm[3] = {mc.a, mc.b, mc.c}, cv[3] = {vc.a, vc.b, vc.c};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (parked[q]) continue;
                cn[q] = adam(cn[q], gp[q], cm[q], cv[q], a_chol);
                if (q != 1) {  // this variance's place in the NEXT iteration's log range
                    const float t = quant_log_of(cn[q] + bd[q]);
                    range_min_combine(mn, cmn, t, 1.f);
                    range_max_combine(mx, cmx, t, 1.f);
                }
            }
            const float f0 = adam(f.a, gf[0], mf.a, vf.a, a_feat), f1 = adam(f.b, gf[1], mf.b, vf.b, a_feat),
                        f2 = adam(f.c, gf[2], mf.c, vf.c, a_feat);
            store_row2(P.xyz, g, nx, ny);
            store_row2(P.m_xyz, g, m_xy.x, m_xy.y);
            store_row2(P.v_xyz, g, v_xy.x, v_xy.y);
            store_row3(P.chol, g, cn[0], cn[1], cn[2]);
            store_row3(P.m_chol, g, cm[0], cm[1], cm[2]);
            store_row3(P.v_chol, g, cv[0], cv[1], cv[2]);
            store_row3(P.feat, g, f0, f1, f2);
            store_row3(P.m_feat, g, mf.a, mf.b, mf.c);
            store_row3(P.v_feat, g, vf.a, vf.b, vf.c);
            if (snapshot) {  // from registers, not read back through memory
                best.xyz[2 * g] = nx;
                best.xyz[2 * g + 1] = ny;
                const float nf[3] = {f0, f1, f2};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    best.chol[3 * g + q] = cn[q];  // parked entries are patched by the finish kernel
                    best.feat[3 * g + q] = nf[q];
                    if (best.bound) best.bound[3 * g + q] = bd[q];
                }
            }
        }
    }
    wave_partial_row(sums, mn, cmn, mx, cmx, Q.partial + (size_t)quant_wave_row(block) * GI2D_QT_ROW);
}


// =====================================================================================================================
// Quantisation-aware iteration of the rotation-scale model (BASELINE config 5): GaussianImage_RS.forward_quantize /
// train_iter_quantize (models/gaussianimage_rs.py:443-485) with the quantiser set of training_setup (:131-163): four
// LSQ+ UniformQuantizers -- positions 12 bit unsigned (2 channels), the RAW `_scaling` 6 bit unsigned (2 channels; as
// written, forward_quantize feeds the dequantised raw scaling to the projection without |. + bound|, :451-453),
// rotation = sigmoid(_rotation) * 2 pi 6 bit SIGNED (1 channel), colours 6 bit unsigned (3 channels).  Every range is a
// learned (scale, beta) pair, so there is no data-dependent range and nothing to park: the update kernel leaves 16
// partial sums per workgroup and a one-workgroup kernel closes them with Adam on the 16 quantiser values
//   qparams = xy scale[2], xy beta[2], scaling scale[2], scaling beta[2], rotation scale, beta, colour scale[3], beta[3]
// (optimizer groups as the model file creates them: positions eps 1e-8; scaling + rotation together, eps 1e-15;
// colours eps 1e-15 -- stepped every iteration like GaussianImage_Covariance.optimizer_step does, :235-247 there).
#define GI2D_QT_RS_SUMS 16
struct QuantValsRS {
    float xs[2], xb[2], ss[2], sb[2], rs, rb, fs[3], fb[3];
};
__device__ __forceinline__ QuantValsRS load_quant_rs(const QuantTrain &Q) {
    QuantValsRS v;
    const float4 *q4 = reinterpret_cast<const float4 *>(Q.qparams);  // 64-byte aligned by contract
    const float4 a = q4[0], b = q4[1], c = q4[2], d = q4[3];
    v.xs[0] = a.x, v.xs[1] = a.y, v.xb[0] = a.z, v.xb[1] = a.w;
    v.ss[0] = b.x, v.ss[1] = b.y, v.sb[0] = b.z, v.sb[1] = b.w;
    v.rs = c.x, v.rb = c.y, v.fs[0] = c.z, v.fs[1] = c.w;
    v.fs[2] = d.x, v.fb[0] = d.y, v.fb[1] = d.z, v.fb[2] = d.w;
    return v;
}
struct QuantRowRS {
    QuantEval xy[2], sc[2], rot, col[3];
    float sig;  // sigmoid(_rotation)
};
__device__ __forceinline__ void quantise_row_rs(const QuantTrain &Q, const QuantValsRS &v, float2 xy, Row3 raw,
                                                Row3 col, QuantRowRS &r) {
#pragma clang fp contract(off)
    r.xy[0] = quant_eval<GI2D_QUANT_LSQ>(xy.x, v.xs[0], v.xb[0], 0.f, Q.qmax_xy);
    r.xy[1] = quant_eval<GI2D_QUANT_LSQ>(xy.y, v.xs[1], v.xb[1], 0.f, Q.qmax_xy);
    r.sc[0] = quant_eval<GI2D_QUANT_LSQ>(raw.a, v.ss[0], v.sb[0], 0.f, Q.qmax_cov);
    r.sc[1] = quant_eval<GI2D_QUANT_LSQ>(raw.b, v.ss[1], v.sb[1], 0.f, Q.qmax_cov);
    r.sig = 1.f / (1.f + expf(-raw.c));                                   // get_rotation: torch.sigmoid(_rotation)
    r.rot = quant_eval<GI2D_QUANT_LSQ>(r.sig * 2.f * 3.14159265358979323846f, v.rs, v.rb, Q.qmin_rot, Q.qmax_rot);
    const float cin[3] = {col.a, col.b, col.c};
#pragma unroll
    for (int q = 0; q < 3; ++q) r.col[q] = quant_eval<GI2D_QUANT_LSQ>(cin[q], v.fs[q], v.fb[q], 0.f, Q.qmax_col);
}

__device__ __forceinline__ void project_fill_quant_rs_body(int block, const UpdateArgs &u, const QuantTrain &Q) {
    const TrainParams &P = u.P;
    const int n = live_n(P, u.n);
    const float clip_coe = u.next.clip_coe, img_w = u.img_w, img_h = u.img_h, radius_clip = u.radius_clip;
    const int tiles_x = u.tiles_x, tiles_y = u.tiles_y;
    float2 *xys = u.xys;
    int32_t *radii = u.radii, *num_tiles_hit = u.next.num_tiles_hit, *lists = u.next.lists, *status = u.next.status;
    float *conics = u.conics;
    PrevBox *prev_box = u.next.prev_box;
    const RecSets &rs = u.next.recs;
    const int g = block * blockDim.x + threadIdx.x;
    begin_binning(g, status);
    const BinRecs recs = recs_for_binning(rs, g == 0);
    if (g >= n) return;
    const PrevBox old_box = prev_box[g];  // with the other inputs, ahead of the stores
    const float opac = P.opacity[g];
    const QuantValsRS v = load_quant_rs(Q);
    QuantRowRS r;
    quantise_row_rs(Q, v, load_row2(P.xyz, g), load_row3(P.chol, g), load_row3(P.feat, g), r);
    const float2 mean = make_float2(r.xy[0].dequant, r.xy[1].dequant);
    const float par[3] = {r.sc[0].dequant, r.sc[1].dequant, r.rot.dequant};
    store_row3(Q.qfeat, g, r.col[0].dequant, r.col[1].dequant, r.col[2].dequant);
    const ProjOut o =
        project_one<kScaleRot>(0, clip_coe, &mean, par, &par[2], img_w, img_h, tiles_x, tiles_y, radius_clip);
    xys[g] = o.xy;
    radii[g] = o.radius;
    conics[3 * g] = o.k0;
    conics[3 * g + 1] = o.k1;
    conics[3 * g + 2] = o.k2;
    num_tiles_hit[g] = o.tiles_hit;
    bin_projected(g, o, opac, r.col[0].dequant, r.col[1].dequant, r.col[2].dequant, tiles_x, tiles_y, radius_clip,
                  old_box, prev_box, lists, recs);
}

template <bool ALONE>
__device__ __forceinline__ void reduce_update_quant_rs_body(int block, const UpdateArgs &u, const QuantTrain &Q,
                                                            const AdamStep &a_xyz, const AdamStep &a_chol,
                                                            const AdamStep &a_feat, int step) {
#pragma clang fp contract(off)
    const TrainParams &P = u.P;
    const int n = live_n(P, u.n);
    const float img_w = u.img_w, img_h = u.img_h;
    const int tiles_x = u.tiles_x, tiles_y = u.tiles_y;
    const int32_t *radii = u.radii;
    const float *conics = u.conics;
    const PrevBox *prev_box = u.next.prev_box;
    const float4 *partial_g = u.partial_g, *partial_big = u.partial_big;
    float *dbg_grads = u.dbg_grads;
    BestSnap best = u.best;
    best.step = step;
    const int g = block * blockDim.x + threadIdx.x;
    // first round of loads: for every row below the host's bound (the live count is itself one of these loads); one image
    // alone also asks for its first gradient rows before the box says how many count (reduce_one)
    AdamRows rows;
    if (g < u.n) rows = adam_load_rows(P, g);
    constexpr int AHEAD = ALONE ? GI2D_UPDATE_ROWS_AHEAD : 0;
    const RowsAhead<AHEAD> ahead = rows_ahead<AHEAD>(partial_g, g < u.n ? g : 0);
    const PrevBox box_ld = g < u.n ? prev_box[g] : no_box();
    const bool snapshot = best_decision(best, best_sse_loads(best), n, g);
    float acc[11];
    const PrevBox pbox = g < n ? box_ld : no_box();
    reduce_one<AHEAD>(g, make_int2(pbox.x, pbox.y), pbox.z, tiles_x * tiles_y * GI2D_TILE_LIST_CAP, partial_g, partial_big,
                      acc, nullptr, &ahead);
    float sums[GI2D_QT_RS_SUMS];
#pragma unroll
    for (int k = 0; k < GI2D_QT_RS_SUMS; ++k) sums[k] = 0.f;
    if (g < n) {
        const QuantValsRS v = load_quant_rs(Q);
        QuantRowRS qr;
        quantise_row_rs(Q, v, rows.x, rows.c, rows.f, qr);
        const float par[3] = {qr.sc[0].dequant, qr.sc[1].dequant, qr.rot.dequant};
        ProjGrad r;
        r.g11 = r.g12 = r.g22 = r.o0 = r.o1 = r.o2 = 0.f;
        r.v_mean = make_float2(0.f, 0.f);
        if (radii[g] > 0) {
            const float conic[3] = {conics[3 * g], conics[3 * g + 1], conics[3 * g + 2]};
            const float vc[3] = {acc[2], acc[3], acc[4]};
            r = project_bwd_one<kScaleRot>(0, par, &par[2], img_w, img_h, conic, make_float2(acc[0], acc[1]), vc);
        }
        // quantiser backward; sums = (v_scale, v_beta) of xy.x, xy.y, scaling.x, scaling.y, rotation, colour r, g, b
        const float gx = quant_grad<GI2D_QUANT_LSQ>(qr.xy[0], r.v_mean.x, v.xs[0], sums[0], sums[1]);
        const float gy = quant_grad<GI2D_QUANT_LSQ>(qr.xy[1], r.v_mean.y, v.xs[1], sums[2], sums[3]);
        float gp[3];
        gp[0] = quant_grad<GI2D_QUANT_LSQ>(qr.sc[0], r.o0, v.ss[0], sums[4], sums[5]);
        gp[1] = quant_grad<GI2D_QUANT_LSQ>(qr.sc[1], r.o1, v.ss[1], sums[6], sums[7]);
        const float g_act = quant_grad<GI2D_QUANT_LSQ>(qr.rot, r.o2, v.rs, sums[8], sums[9]);
        gp[2] = (g_act * (2.f * 3.14159265358979323846f)) * (qr.sig * (1.f - qr.sig));  // through * 2 pi, then sigmoid
        float gf[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            gf[q] = quant_grad<GI2D_QUANT_LSQ>(qr.col[q], acc[5 + q], v.fs[q], sums[10 + 2 * q], sums[11 + 2 * q]);
        if (dbg_grads) {
            float *d = dbg_grads + 8 * (size_t)g;
            d[0] = gx, d[1] = gy, d[2] = gp[0], d[3] = gp[1], d[4] = gp[2], d[5] = gf[0], d[6] = gf[1], d[7] = gf[2];
        }
        float2 new_xy;
        Row3 new_chol, new_feat;
        adam_update_rows(rows, gx, gy, gp, gf, a_xyz, a_chol, a_feat);
        adam_store_rows(P, g, rows);
        new_xy = rows.x, new_chol = rows.c, new_feat = rows.f;
        if (snapshot) {  // from registers, not read back through memory
            best.xyz[2 * g] = new_xy.x;
            best.xyz[2 * g + 1] = new_xy.y;
            const float nc[3] = {new_chol.a, new_chol.b, new_chol.c}, nf[3] = {new_feat.a, new_feat.b, new_feat.c};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                best.chol[3 * g + q] = nc[q];
                best.feat[3 * g + q] = nf[q];
                if (best.bound) best.bound[3 * g + q] = P.bound[(size_t)P.bound_stride * g + q];
            }
        }
    }
    // this wave's partial row (see wave_partial_row)
#pragma unroll
    for (int k = 0; k < GI2D_QT_RS_SUMS; ++k) sums[k] = wave_sum(sums[k]);
    if ((threadIdx.x & 63) == 0) {
        float4 *r4 = reinterpret_cast<float4 *>(Q.partial + (size_t)quant_wave_row(block) * GI2D_QT_ROW);
#pragma unroll
        for (int k = 0; k < GI2D_QT_RS_SUMS / 4; ++k)
            r4[k] = make_float4(sums[4 * k], sums[4 * k + 1], sums[4 * k + 2], sums[4 * k + 3]);
    }
}

// One workgroup: sums the partial rows in double (fixed order) and runs Adam on the 16 quantiser values.
__device__ __forceinline__ void quant_finish_rs(int blocks, const QuantTrain &Q, const AdamStep &a_qxy,
                                                const AdamStep &a_qcov, const AdamStep &a_qcol, const BestSnap &best) {
#pragma clang fp contract(off)
    __shared__ double lane_acc[256][GI2D_QT_RS_SUMS + 1];  // odd stride in 8-byte words
    __shared__ double dred[16][GI2D_QT_RS_SUMS];
    float q_p = 0.f, q_m = 0.f, q_v = 0.f;
    if (threadIdx.x < GI2D_QT_RS_SUMS)
        q_p = Q.qparams[threadIdx.x], q_m = Q.qm[threadIdx.x], q_v = Q.qv[threadIdx.x];
    const bool snap = best.sse != nullptr && best.info[1] == best.step;
    double acc[GI2D_QT_RS_SUMS];
#pragma unroll
    for (int k = 0; k < GI2D_QT_RS_SUMS; ++k) acc[k] = 0.0;
    for (int b = threadIdx.x; b < blocks; b += (int)blockDim.x) {
        const float4 *row = reinterpret_cast<const float4 *>(Q.partial + (size_t)b * GI2D_QT_ROW);
        float4 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = row[k];
        const float *f = reinterpret_cast<const float *>(r);
#pragma unroll
        for (int k = 0; k < GI2D_QT_RS_SUMS; ++k) acc[k] += (double)f[k];
    }
#pragma unroll
    for (int k = 0; k < GI2D_QT_RS_SUMS; ++k) lane_acc[threadIdx.x][k] = acc[k];
    __syncthreads();
    const int k = threadIdx.x & 15, j = threadIdx.x >> 4;
    {
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) part += lane_acc[j * 16 + i][k];
        dred[j][k] = part;
    }
    __syncthreads();
    if (threadIdx.x < GI2D_QT_RS_SUMS) {
        const int q = threadIdx.x;
        // slot q of qparams -> (channel, scale or beta): xs xs xb xb | ss ss sb sb | rs rb | fs fs fs fb fb fb
        int ch, isb;
        if (q < 4) ch = q & 1, isb = q >> 1;
        else if (q < 8) ch = 2 + (q & 1), isb = (q - 4) >> 1;
        else if (q < 10) ch = 4, isb = q - 8;
        else ch = 5 + (q - 10) % 3, isb = (q - 10) / 3;
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += dred[w][2 * ch + isb];
        const float grad = (float)t;
        const AdamStep &a = q < 4 ? a_qxy : (q < 10 ? a_qcov : a_qcol);
        const float nv = adam(q_p, grad, q_m, q_v, a);
        Q.qparams[q] = nv;
        Q.qm[q] = q_m;
        Q.qv[q] = q_v;
        if (Q.dbg_q) Q.dbg_q[q] = grad;
        if (snap && Q.best_q) Q.best_q[q] = nv;
    }
}

// ------------------------------------------------------------------ kernels of the quantisation-aware iteration
// MODEL: 1 covariance, 2 rotation-scale.  Single-image forms take the image's argument blocks as kernel arguments; the
// batched forms find theirs in the batch table by workgroup index (gi2d_batch.h), like the plain fitting kernels.
// The closing step is a launch of its own (one workgroup per image).  Running it in the last workgroup of the producing
// kernel instead (ticket + device-scope fences) was measured and dropped: on this 8-XCD part every workgroup's release
// fence writes its L2 back, which made the update kernel 17 us slower to save a 4 us launch.
template <int MODEL>
__global__ __launch_bounds__(256) void train_project_fill_quant_kernel(UpdateArgs u, QuantTrain Q) {
    if (MODEL == 2)
        project_fill_quant_rs_body((int)blockIdx.x, u, Q);
    else
        project_fill_quant_body((int)blockIdx.x, u, Q);
}
template <int MODEL>
__global__ __launch_bounds__(256) void train_project_fill_quant_batched_kernel(const BatchImage *__restrict__ imgs,
                                                                               const int *__restrict__ pg_start,
                                                                               int k_images) {
    const int k = batch_find(pg_start, k_images, (int)blockIdx.x);
    const int local = __builtin_amdgcn_readfirstlane((int)blockIdx.x - pg_start[k]);
    if (MODEL == 2)
        project_fill_quant_rs_body(local, imgs[k].u, imgs[k].q);
    else
        project_fill_quant_body(local, imgs[k].u, imgs[k].q);
}
template <int MODEL>
__global__ __launch_bounds__(256) void train_reduce_update_quant_kernel(UpdateArgs u, QuantTrain Q, AdamStep a_xyz,
                                                                        AdamStep a_chol, AdamStep a_feat, int step) {
    if (MODEL == 2)
        reduce_update_quant_rs_body<true>((int)blockIdx.x, u, Q, a_xyz, a_chol, a_feat, step);
    else
        reduce_update_quant_body<true>((int)blockIdx.x, u, Q, a_xyz, a_chol, a_feat, step);
}
template <int MODEL>
__global__ __launch_bounds__(256) void train_reduce_update_quant_batched_kernel(const BatchImage *__restrict__ imgs,
                                                                                const int *__restrict__ pg_start,
                                                                                int k_images, AdamStep a_xyz,
                                                                                AdamStep a_chol, AdamStep a_feat,
                                                                                int step) {
    const int k = batch_find(pg_start, k_images, (int)blockIdx.x);
    const int local = __builtin_amdgcn_readfirstlane((int)blockIdx.x - pg_start[k]);
    if (MODEL == 2)
        reduce_update_quant_rs_body<false>(local, imgs[k].u, imgs[k].q, a_xyz, a_chol, a_feat, step);
    else
        reduce_update_quant_body<false>(local, imgs[k].u, imgs[k].q, a_xyz, a_chol, a_feat, step);
}
__global__ __launch_bounds__(256) void train_quant_range_kernel(UpdateArgs u, QuantTrain Q) {
    quant_range_body((int)blockIdx.x, u, Q);
}
__global__ __launch_bounds__(256) void train_quant_range_batched_kernel(const BatchImage *__restrict__ imgs,
                                                                        const int *__restrict__ pg_start, int k_images) {
    const int k = batch_find(pg_start, k_images, (int)blockIdx.x);
    const int local = __builtin_amdgcn_readfirstlane((int)blockIdx.x - pg_start[k]);
    quant_range_body(local, imgs[k].u, imgs[k].q);
}
// closing step; `rows`: partial rows (= waves) of the image's per-gaussian launch.  MODE 0: range only (covariance model,
// start of a call), 1: covariance model, 2: rotation-scale model.
template <int MODE>
__device__ __forceinline__ void quant_finish_any(int rows, const UpdateArgs &u, const QuantTrain &Q, const AdamStep &a_chol,
                                                 const AdamStep &a_qxy, const AdamStep &a_qcov, const AdamStep &a_qcol,
                                                 int step) {
    BestSnap best = u.best;
    best.step = step;
    if (MODE == 0) {
        best.sse = nullptr;
        quant_finish<true>(rows, u.P, Q, a_chol, a_qxy, a_qcov, a_qcol, nullptr, best);
    } else if (MODE == 1) {
        quant_finish<false>(rows, u.P, Q, a_chol, a_qxy, a_qcov, a_qcol, u.dbg_grads, best);
    } else {
        quant_finish_rs(rows, Q, a_qxy, a_qcov, a_qcol, best);
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void train_quant_finish_kernel(int rows, UpdateArgs u, QuantTrain Q, AdamStep a_chol,
                                                                 AdamStep a_qxy, AdamStep a_qcov, AdamStep a_qcol,
                                                                 int step) {
    quant_finish_any<MODE>(rows, u, Q, a_chol, a_qxy, a_qcov, a_qcol, step);
}
// one workgroup per image; rows of image k = its workgroups in the per-gaussian launches x waves per workgroup
template <int MODE>
__global__ __launch_bounds__(256) void train_quant_finish_batched_kernel(const BatchImage *__restrict__ imgs,
                                                                         const int *__restrict__ pg_start,
                                                                         int waves_per_block, AdamStep a_chol,
                                                                         AdamStep a_qxy, AdamStep a_qcov, AdamStep a_qcol,
                                                                         int step) {
    const int k = (int)blockIdx.x;
    quant_finish_any<MODE>((pg_start[k + 1] - pg_start[k]) * waves_per_block, imgs[k].u, imgs[k].q, a_chol, a_qxy, a_qcov,
                           a_qcol, step);
}

// The batch table is written by kernels that carry the argument blocks as kernel arguments: stream-ordered, no host
// buffer whose lifetime anybody has to think about, capturable in a graph.
#define GI2D_BATCH_PACK 6 /* argument blocks per writer launch (kernel arguments are limited to 4 KB) */
struct BatchPack {
    BatchImage img[GI2D_BATCH_PACK];
};
static_assert(sizeof(BatchPack) <= 3840, "BatchPack must fit the kernel-argument segment");
__global__ __launch_bounds__(256) void batch_write_images_kernel(BatchPack p, int count, BatchImage *__restrict__ dst) {
    constexpr int WORDS = (int)(sizeof(BatchImage) / sizeof(int));
    static_assert(sizeof(BatchImage) % sizeof(int) == 0, "BatchImage is copied word by word");
    const int *src = reinterpret_cast<const int *>(&p);
    int *out = reinterpret_cast<int *>(dst);
    for (int i = threadIdx.x; i < count * WORDS; i += blockDim.x) out[i] = src[i];
}
__global__ __launch_bounds__(256) void batch_write_head_kernel(BatchHead h, BatchHead *__restrict__ dst) {
    const int *src = reinterpret_cast<const int *>(&h);
    int *out = reinterpret_cast<int *>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(BatchHead) / sizeof(int)); i += blockDim.x) out[i] = src[i];
}

// Host side of the writers: `imgs` (host) -> table.img[0 .. k_images), `head` -> table.head.
void write_batch_table(const BatchTable &table, const BatchImage *imgs, int k_images, const BatchHead &head,
                       hipStream_t st) {
    BatchPack pack;
    for (int k0 = 0; k0 < k_images; k0 += GI2D_BATCH_PACK) {
        const int cnt = k_images - k0 < GI2D_BATCH_PACK ? k_images - k0 : GI2D_BATCH_PACK;
        for (int i = 0; i < cnt; ++i) pack.img[i] = imgs[k0 + i];
        hipLaunchKernelGGL(batch_write_images_kernel, dim3(1), dim3(256), 0, st, pack, cnt, table.img + k0);
    }
    hipLaunchKernelGGL(batch_write_head_kernel, dim3(1), dim3(256), 0, st, head, table.head);
}

}  // namespace gi2d

using namespace gi2d;

// kind x (more iterations follow: the update kernel also starts the next one) x optimizer -> GI2D_LAUNCH_RU(K, F, A)
#define GI2D_DISPATCH_RU2(K, more, adan) \
    do {                                  \
        if (more) {                       \
            if (adan)                     \
                GI2D_LAUNCH_RU(K, true, true);   \
            else                          \
                GI2D_LAUNCH_RU(K, true, false);  \
        } else {                          \
            if (adan)                     \
                GI2D_LAUNCH_RU(K, false, true);  \
            else                          \
                GI2D_LAUNCH_RU(K, false, false); \
        }                                 \
    } while (0)
#define GI2D_DISPATCH_RU(kind, more, adan)           \
    do {                                             \
        if ((kind) == 2)                             \
            GI2D_DISPATCH_RU2(kScaleRot, more, adan);   \
        else if ((kind) == 0)                        \
            GI2D_DISPATCH_RU2(kCholesky, more, adan);   \
        else                                         \
            GI2D_DISPATCH_RU2(kCovariance, more, adan); \
    } while (0)

extern "C" {

static TrainParams params_of(const gi2d_train_state *s) {
    TrainParams P;
    P.xyz = s->xyz;
    P.chol = s->chol;
    P.feat = s->feat;
    P.opacity = s->opacity;
    P.bound = s->bound;
    P.bound_stride = s->bound_stride;
    P.n_dev = s->num_points_dev;
    P.m_xyz = s->m_xyz;
    P.v_xyz = s->v_xyz;
    P.m_chol = s->m_chol;
    P.v_chol = s->v_chol;
    P.m_feat = s->m_feat;
    P.v_feat = s->v_feat;
    P.d_xyz = s->d_xyz;
    P.d_chol = s->d_chol;
    P.d_feat = s->d_feat;
    P.pg_xyz = s->pg_xyz;
    P.pg_chol = s->pg_chol;
    P.pg_feat = s->pg_feat;
    return P;
}

static int train_check(const gi2d_train_state *s, int &tx, int &ty) {
    if (!s || s->kind < 0 || s->kind > 2 || s->num_points < 0 || s->img_height <= 0 || s->img_width <= 0) {
        set_error("train: bad state");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    tx = (s->img_width + GI2D_TILE - 1) / GI2D_TILE;
    ty = (s->img_height + GI2D_TILE - 1) / GI2D_TILE;
    if (!s->xyz || !s->chol || !s->feat || !s->opacity || !s->bound || !s->m_xyz || !s->v_xyz || !s->m_chol ||
        !s->v_chol || !s->m_feat || !s->v_feat || !s->gt || !s->xys || !s->conics || !s->radii ||
        !s->num_tiles_hit || !s->out_img || !s->tile_sse || !s->status) {
        set_error("train: null pointer in state");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (!s->workspace || s->workspace_bytes < gi2d_fast_workspace_bytes(s->num_points, tx, ty)) {
        set_error("train: workspace too small");
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    return GI2D_OK;
}

// One image's argument block of the per-gaussian fitting kernels (gi2d_batch.h).
static UpdateArgs update_args_of(const gi2d_train_state *s, const FastWs &w, int tx, int ty) {
    UpdateArgs u;
    const int n = s->num_points;
    u.n = n;
    u.P = params_of(s);
    u.xys = (float2 *)s->xys;
    u.radii = s->radii;
    u.conics = s->conics;
    u.tiles_x = tx, u.tiles_y = ty;
    u.radius_clip = s->radius_clip;
    u.gids_sorted = w.gids_sorted;
    u.tile_bins = (const int2 *)w.tile_bins;
    u.partial_g = w.partial_g;
    u.partial_big = w.partial_big;
    u.img_w = (float)s->img_width, u.img_h = (float)s->img_height;
    u.dbg_grads = s->dbg_grads;
    u.best.xyz = s->best_xyz;
    u.best.chol = s->best_chol;
    u.best.feat = s->best_feat;
    u.best.bound = s->bound_stride ? s->best_bound : nullptr;
    u.best.sse = s->best_sse;
    u.best.info = s->best_info;
    u.best.tile_sse = s->tile_sse;
    u.best.num_tiles = tx * ty;
    u.best.step = 0;
    u.next.clip_coe = s->clip_coe;
    u.next.num_tiles_hit = s->num_tiles_hit;
    u.next.lists = w.lists;
    // the tiles' inboxes: the caller's own buffer, for an image small enough to use them (single-image calls only look)
    // (... and whose tile pass can store write-through, which the inbox instantiation always does: wt_fits)
    u.next.inbox = (s->inbox != nullptr && inbox_bytes((long long)tx * ty) > 0 && s->inbox_bytes >= inbox_bytes((long long)tx * ty) &&
                    wt_fits(w, tx * ty, (size_t)s->img_width * s->img_height * 12))
                       ? (float4 *)s->inbox
                       : nullptr;
    u.next.prev_box = w.prev_box;
    u.next.recs = rec_sets(w, n);
    u.next.status = s->status;
    u.next.tile_order = w.tile_order;
    return u;
}

static void train_launch_project_fill(const gi2d_train_state *s, const FastWs &w, const TrainParams &, int tx,
                                      int ty, hipStream_t st) {
    const int n = s->num_points;
    const int bs = per_gaussian_block(n);
    const dim3 gg((n + bs - 1) / bs), bb(bs);
    const UpdateArgs u = update_args_of(s, w, tx, ty);
    if (s->kind == 2)
        hipLaunchKernelGGL(train_project_fill_kernel<kScaleRot>, gg, bb, 0, st, u);
    else if (s->kind == 0)
        hipLaunchKernelGGL(train_project_fill_kernel<kCholesky>, gg, bb, 0, st, u);
    else
        hipLaunchKernelGGL(train_project_fill_kernel<kCovariance>, gg, bb, 0, st, u);
}


static AdamStep make_adam_step(double lr, double beta1, double beta2, double beta3, float eps, int step, bool adan_opt) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double bc3 = 1.0 - pow(beta3, (double)step);
    AdamStep a;
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(adan_opt ? bc3 : bc2);
    a.one_minus_b1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.one_minus_b2 = (float)(1.0 - beta2);
    a.eps = eps;
    a.b1 = (float)beta1;
    a.b3 = (float)beta3;
    a.one_minus_b3 = (float)(1.0 - beta3);
    a.step_size_diff = (float)(lr * beta2 / bc2);
    a.first = step == 1;
    return a;
}

static int quant_of(const gi2d_train_state *s, QuantTrain &Q) {
    const gi2d_train_quant *q = s->quant;
    if ((s->kind != 1 && s->kind != 2) || s->optimizer != 0) {
        set_error("train: quantisation is wired for the covariance and the rotation-scale model with Adam "
                  "(train_quantize.py; models/gaussianimage_rs.py:131-163)");
        return GI2D_ERR_UNSUPPORTED;
    }
    const bool rs = s->kind == 2;
    if (q->xy_bits < 1 || q->xy_bits > 16 || q->cov_bits < 1 || q->cov_bits > 16 || q->color_bits < 1 ||
        q->color_bits > 16 || !q->qparams || !q->qm || !q->qv || !q->qfeat || !q->partial ||
        (!rs && (!q->range || !q->defer || q->defer_capacity < 2)) || (rs && (q->rot_bits < 2 || q->rot_bits > 16)) ||
        (rs && ((uintptr_t)q->qparams & 15))) {
        set_error("train: bad quantisation state");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    Q.qmin_rot = rs ? -(float)(1 << (q->rot_bits - 1)) : 0.f;
    Q.qmax_rot = rs ? (float)((1 << (q->rot_bits - 1)) - 1) : 0.f;
    Q.qmax_xy = (float)((1 << q->xy_bits) - 1);
    Q.qmax_cov = (float)((1 << q->cov_bits) - 1);
    Q.qmax_col = (float)((1 << q->color_bits) - 1);
    Q.qparams = q->qparams;
    Q.qm = q->qm;
    Q.qv = q->qv;
    Q.range = q->range;
    Q.qfeat = q->qfeat;
    Q.partial = q->partial;
    Q.defer = q->defer;
    Q.defer_cap = q->defer_capacity;
    Q.best_q = q->best_qparams;
    Q.dbg_q = q->dbg_qgrads;
    return GI2D_OK;
}

// launch shape of the per-gaussian kernels of a quantisation-aware iteration; rows = partial rows (waves) they leave
struct QuantLaunch {
    int bs, blocks, rows;
};
static QuantLaunch quant_launch_of(int n) {
    QuantLaunch l;
    l.bs = per_gaussian_block(n);
    l.blocks = (n + l.bs - 1) / l.bs;
    l.rows = l.blocks * (l.bs / 64);
    return l;
}
// log range of the current variances (covariance model, start of a call)
static void train_launch_quant_range(const UpdateArgs &u, const QuantTrain &Q, hipStream_t st) {
    const QuantLaunch l = quant_launch_of(u.n);
    const AdamStep z = make_adam_step(0.0, 0.9, 0.999, 0.0, 1.f, 1, false);
    hipLaunchKernelGGL(train_quant_range_kernel, dim3(l.blocks), dim3(l.bs), 0, st, u, Q);
    hipLaunchKernelGGL(train_quant_finish_kernel<0>, dim3(1), dim3(256), 0, st, l.rows, u, Q, z, z, z, z, 0);
}
// activations / quantisers + projection + fill
static void train_launch_project_fill_quant(int model, const UpdateArgs &u, const QuantTrain &Q, hipStream_t st) {
    const QuantLaunch l = quant_launch_of(u.n);
    if (model == 2)
        hipLaunchKernelGGL(train_project_fill_quant_kernel<2>, dim3(l.blocks), dim3(l.bs), 0, st, u, Q);
    else
        hipLaunchKernelGGL(train_project_fill_quant_kernel<1>, dim3(l.blocks), dim3(l.bs), 0, st, u, Q);
}

// Forward only (render): activations + projection + fill + rasterize into state->out_img.
int gi2d_train_render(const gi2d_train_state *s, gi2d_stream_t st_) {
    int tx, ty;
    int rc = train_check(s, tx, ty);
    if (rc != GI2D_OK) return rc;
    hipStream_t st = (hipStream_t)st_;
    const int n = s->num_points;
    if (n == 0) return GI2D_OK;
    FastWs w = carve_fast(s->workspace, n, tx * ty);
    const TrainParams P = params_of(s);
    if (s->quant) {  // forward_quantize (models/gaussianimage_covariance.py:384-410)
        QuantTrain Q;
        rc = quant_of(s, Q);
        if (rc != GI2D_OK) return rc;
        const UpdateArgs uq = update_args_of(s, w, tx, ty);
        if (s->kind != 2) train_launch_quant_range(uq, Q, st);  // (the RS model's ranges are learned values)
        train_launch_project_fill_quant(s->kind, uq, Q, st);  // GaussianImage_RS.forward_quantize: models/gaussianimage_rs.py:443-471
        return gi2d_fast_rasterize_forward(n, tx, ty, (unsigned)s->img_width, (unsigned)s->img_height, nullptr,
                                           s->workspace, s->workspace_bytes, s->status, nullptr, nullptr, s->out_img,
                                           st_);
    }
    train_launch_project_fill(s, w, P, tx, ty, st);
    return gi2d_fast_rasterize_forward(n, tx, ty, (unsigned)s->img_width, (unsigned)s->img_height, nullptr, s->workspace,
                                       s->workspace_bytes, s->status, nullptr, nullptr, s->out_img, st_);
}

// `count` full iterations: render, L2 loss gradient + backward, Adam update.  lr[3] / first_step are host values:
// learning rates of the xyz / cholesky / colour groups (constant over the call) and the 1-based Adam step count of
// the first iteration.  Launches: project+fill once, then per iteration the tile pass and the update kernel, which
// also projects and bins the updated gaussians for the following iteration (all but the last) -- 2*count + 1.
int gi2d_train_steps(const gi2d_train_state *s, const double *lr, double beta1, double beta2, float eps,
                     int first_step, int count, gi2d_stream_t st_) {
    int tx, ty;
    int rc = train_check(s, tx, ty);
    if (rc != GI2D_OK) return rc;
    hipStream_t st = (hipStream_t)st_;
    const int n = s->num_points;
    if (n == 0 || count <= 0) return GI2D_OK;
    if (!lr || first_step < 1) {
        set_error("train steps: bad lr/step");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    FastWs w = carve_fast(s->workspace, n, tx * ty);
    const TrainParams P = params_of(s);
    const float grad_scale = 2.f / (3.f * (float)s->img_height * (float)s->img_width);
    BestSnap best;
    best.xyz = s->best_xyz;
    best.chol = s->best_chol;
    best.feat = s->best_feat;
    best.bound = s->bound_stride ? s->best_bound : nullptr;
    best.sse = s->best_sse;
    best.info = s->best_info;
    best.tile_sse = s->tile_sse;
    best.num_tiles = tx * ty;
    if (best.sse && (!best.xyz || !best.chol || !best.feat || !best.info || (s->bound_stride && !s->best_bound))) {
        set_error("train steps: best_sse given without the snapshot buffers");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    const bool adan_opt = s->optimizer == 1;
    if (s->optimizer < 0 || s->optimizer > 1 ||
        (adan_opt && (!s->d_xyz || !s->d_chol || !s->d_feat || !s->pg_xyz || !s->pg_chol || !s->pg_feat))) {
        set_error("train steps: unknown optimizer, or Adan without its extra state (d_*, pg_*)");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    // a large image's tile passes run as two launches while the previous call on this workspace saw at most one row in
    // sixteen above the small form's capacity (gi2d_fast.hip: pass_form_begin)
    const long long tiles = (long long)tx * ty;
    const int form = single_pass_begin(s->workspace, tiles, st);
    if (s->quant) {
        QuantTrain Q;
        rc = quant_of(s, Q);
        if (rc != GI2D_OK) return rc;
        const gi2d_train_quant *q = s->quant;
        if (q->first_step < 1) {
            set_error("train steps: quantiser optimizer step must be >= 1");
            return GI2D_ERR_INVALID_ARGUMENT;
        }
        const UpdateArgs uq = update_args_of(s, w, tx, ty);
        const QuantLaunch l = quant_launch_of(n);
        const int model = s->kind == 2 ? 2 : 1;
        if (model == 1) train_launch_quant_range(uq, Q, st);
        for (int it = 0; it < count; ++it) {
            const int step = first_step + it, qstep = q->first_step + it;
            train_launch_project_fill_quant(model, uq, Q, st);
            rc = fast_forward_backward_form(n, tx, ty, (unsigned)s->img_width, (unsigned)s->img_height, nullptr, nullptr,
                                            s->gt, grad_scale, s->tile_sse, s->workspace, s->workspace_bytes, s->status,
                                            s->out_img, st_, form);
            if (rc != GI2D_OK) return rc;
            AdamStep a[3], aq[3];
            for (int k = 0; k < 3; ++k) {
                a[k] = make_adam_step(lr[k], beta1, beta2, 0.0, eps, step, false);
                aq[k] = make_adam_step(q->lr[k], q->beta1, q->beta2, 0.0, q->eps[k], qstep, false);
            }
            if (model == 2) {
                hipLaunchKernelGGL(train_reduce_update_quant_kernel<2>, dim3(l.blocks), dim3(l.bs), 0, st, uq, Q, a[0],
                                   a[1], a[2], step);
                hipLaunchKernelGGL(train_quant_finish_kernel<2>, dim3(1), dim3(256), 0, st, l.rows, uq, Q, a[1], aq[0],
                                   aq[1], aq[2], step);
            } else {
                hipLaunchKernelGGL(train_reduce_update_quant_kernel<1>, dim3(l.blocks), dim3(l.bs), 0, st, uq, Q, a[0],
                                   a[1], a[2], step);
                hipLaunchKernelGGL(train_quant_finish_kernel<1>, dim3(1), dim3(256), 0, st, l.rows, uq, Q, a[1], aq[0],
                                   aq[1], aq[2], step);
            }
        }
        single_pass_end(s->workspace, w, tiles, st);
        return check_launch("train steps (quantised)");
    }
    const UpdateArgs u = update_args_of(s, w, tx, ty);
    const int bs = per_gaussian_block(n);
    const dim3 gg((n + bs - 1) / bs + 1), bb(bs);  // + 1: the workgroup that orders the tiles
    train_launch_project_fill(s, w, P, tx, ty, st);
    for (int it = 0; it < count; ++it) {
        const int step = first_step + it;
        // (the inboxes: one image of at most GI2D_INBOX_MAX_TILES tiles, i.e. a tile pass of the general form throughout;
        // the update kernel delivers through them when another iteration follows, and the tile pass of that iteration --
        // never the first of a call, which follows the projection kernel -- is the one built to take entrants in)
        // ... and whose caller brought the inboxes' buffer: gi2d_train_state::inbox)
        const bool small_image = u.next.inbox != nullptr;
        rc = fast_forward_backward_form(n, tx, ty, (unsigned)s->img_width, (unsigned)s->img_height, nullptr, nullptr,
                                        s->gt, grad_scale, s->tile_sse, s->workspace, s->workspace_bytes, s->status,
                                        s->out_img, st_, form, it > 0 ? u.next.inbox : nullptr);
        if (rc != GI2D_OK) return rc;
        AdamStep a[3];
        for (int q = 0; q < 3; ++q) a[q] = make_adam_step(lr[q], beta1, beta2, s->beta3, eps, step, adan_opt);
        const bool more = it + 1 < count;
        const bool inbox = more && small_image;
#define GI2D_LAUNCH_RU(K, F, A)                                                                                      \
    do {                                                                                                             \
        if (F && inbox)                                                                                              \
            hipLaunchKernelGGL((train_reduce_update_kernel<K, F, A, F>), gg, bb, 0, st, u, a[0], a[1], a[2], step);  \
        else                                                                                                         \
            hipLaunchKernelGGL((train_reduce_update_kernel<K, F, A>), gg, bb, 0, st, u, a[0], a[1], a[2], step);     \
    } while (0)
        GI2D_DISPATCH_RU(s->kind, more, adan_opt);
#undef GI2D_LAUNCH_RU
    }
    single_pass_end(s->workspace, w, tiles, st);
    return check_launch("train steps");
}

int gi2d_train_step(const gi2d_train_state *s, const double *lr, double beta1, double beta2, float eps, int step,
                    gi2d_stream_t st) {
    return gi2d_train_steps(s, lr, beta1, beta2, eps, step, 1, st);
}

// ---------------------------------------------------------------------------------------------- several images per launch
size_t gi2d_batch_bytes(int num_images) { return carve_batch(nullptr, num_images).bytes; }
size_t gi2d_train_inbox_bytes(int tiles_x, int tiles_y) { return inbox_bytes((long long)tiles_x * tiles_y); }

int gi2d_train_steps_batched(int num_images, const gi2d_train_state *const *states, void *batch, size_t batch_bytes,
                             const double *lr, double beta1, double beta2, float eps, int first_step, int count,
                             gi2d_stream_t st_) {
    hipStream_t st = (hipStream_t)st_;
    if (num_images < 1 || num_images > GI2D_BATCH_MAX || !states) {
        set_error("train steps (batched): 1 .. 64 images per launch");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (!batch || batch_bytes < gi2d_batch_bytes(num_images) || ((uintptr_t)batch & 15)) {
        set_error("train steps (batched): batch table too small (gi2d_batch_bytes) or not 16-byte aligned");
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    if (count <= 0) return GI2D_OK;
    if (!lr || first_step < 1) {
        set_error("train steps (batched): bad lr/step");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    const gi2d_train_state *s0 = states[0];
    BatchHead head;
    std::memset(&head, 0, sizeof(head));
    long long total_n = 0;
    bool uniform = true;
    int tiles0 = 0;
    for (int k = 0; k < num_images; ++k) {
        const gi2d_train_state *s = states[k];
        int tx, ty;
        int rc = train_check(s, tx, ty);
        if (rc != GI2D_OK) return rc;
        if (s->kind != s0->kind || s->optimizer != s0->optimizer || s->beta3 != s0->beta3 || s->optimizer < 0 ||
            s->optimizer > 1 || (s->quant != nullptr) != (s0->quant != nullptr)) {
            set_error("train steps (batched): the images of a batch share model kind and optimizer, and are all "
                      "quantisation-aware or none");
            return GI2D_ERR_UNSUPPORTED;
        }
        if (s->quant) {
            const gi2d_train_quant *q = s->quant, *q0 = s0->quant;
            bool same = q->xy_bits == q0->xy_bits && q->cov_bits == q0->cov_bits && q->color_bits == q0->color_bits &&
                        q->rot_bits == q0->rot_bits && q->beta1 == q0->beta1 && q->beta2 == q0->beta2 &&
                        q->first_step == q0->first_step;
            for (int c = 0; c < 3; ++c) same = same && q->lr[c] == q0->lr[c] && q->eps[c] == q0->eps[c];
            if (!same) {
                set_error("train steps (batched): the quantisers of a batch share bit depths, learning rates, eps, betas "
                          "and step count");
                return GI2D_ERR_UNSUPPORTED;
            }
            if (q->first_step < 1) {
                set_error("train steps (batched): quantiser optimizer step must be >= 1");
                return GI2D_ERR_INVALID_ARGUMENT;
            }
            if (s->optimizer != 0) {
                set_error("train steps (batched): quantisation-aware iterations use Adam");
                return GI2D_ERR_UNSUPPORTED;
            }
        }
        if (s->optimizer == 1 && (!s->d_xyz || !s->d_chol || !s->d_feat || !s->pg_xyz || !s->pg_chol || !s->pg_feat)) {
            set_error("train steps (batched): Adan without its extra state (d_*, pg_*)");
            return GI2D_ERR_INVALID_ARGUMENT;
        }
        if (s->best_sse && (!s->best_xyz || !s->best_chol || !s->best_feat || !s->best_info ||
                            (s->bound_stride && !s->best_bound))) {
            set_error("train steps (batched): best_sse given without the snapshot buffers");
            return GI2D_ERR_INVALID_ARGUMENT;
        }
        total_n += s->num_points;
        if (k == 0) tiles0 = tx * ty;
        uniform = uniform && tx * ty == tiles0;
    }
    const bool adan_opt = s0->optimizer == 1, quantised = s0->quant != nullptr;
    const int bs = per_gaussian_block((int)(total_n > 0x7fffffff ? 0x7fffffff : total_n));
    BatchTable b = carve_batch(batch, num_images);
    // the table: tile-pass and per-gaussian workgroup ranges, one argument block per image
    int tile_blocks = 0, pg_blocks = 0;
    std::vector<BatchImage> host_imgs((size_t)num_images);
    for (int k = 0; k < num_images; ++k) {
        const gi2d_train_state *s = states[k];
        const int tx = (s->img_width + GI2D_TILE - 1) / GI2D_TILE, ty = (s->img_height + GI2D_TILE - 1) / GI2D_TILE;
        const int n = s->num_points;
        FastWs w = carve_fast(s->workspace, n, tx * ty);
        const float grad_scale = 2.f / (3.f * (float)s->img_height * (float)s->img_width);
        host_imgs[k].t = tile_pass_args(w, n, tx, ty, s->img_width, s->img_height, s->status, s->out_img, s->gt,
                                        grad_scale, s->tile_sse);
        host_imgs[k].u = update_args_of(s, w, tx, ty);
        std::memset(&host_imgs[k].q, 0, sizeof(QuantTrain));
        if (quantised) {
            int rc = quant_of(s, host_imgs[k].q);
            if (rc != GI2D_OK) return rc;
        }
        head.tile_start[k] = tile_blocks;
        head.pg_start[k] = pg_blocks;
        tile_blocks += tx * ty;
        // plain fitting: + 1, the workgroup that orders the tiles (the quantised update kernels have none)
        pg_blocks += (n + bs - 1) / bs + (quantised ? 0 : 1);
    }
    head.tile_start[num_images] = tile_blocks;
    head.pg_start[num_images] = pg_blocks;
    write_batch_table(b, host_imgs.data(), num_images, head, st);
    const int two_phase = batch_pass_begin(batch, tile_blocks, st);
    const BatchImage *imgs = b.img;
    const int *pg_start = b.head->pg_start;
    const dim3 gg((unsigned)pg_blocks), bb(bs);
    if (quantised) {
        // train_iter_quantize for every image of the batch: four launches per iteration, whatever the number of images
        const gi2d_train_quant *q = s0->quant;
        const int model = s0->kind == 2 ? 2 : 1, wpb = bs / 64;
        const dim3 gi((unsigned)num_images), b256(256);
        if (model == 1) {
            const AdamStep z = make_adam_step(0.0, 0.9, 0.999, 0.0, 1.f, 1, false);
            hipLaunchKernelGGL(train_quant_range_batched_kernel, gg, bb, 0, st, imgs, pg_start, num_images);
            hipLaunchKernelGGL(train_quant_finish_batched_kernel<0>, gi, b256, 0, st, imgs, pg_start, wpb, z, z, z, z, 0);
        }
        for (int it = 0; it < count; ++it) {
            const int step = first_step + it, qstep = q->first_step + it;
            if (model == 2)
                hipLaunchKernelGGL(train_project_fill_quant_batched_kernel<2>, gg, bb, 0, st, imgs, pg_start, num_images);
            else
                hipLaunchKernelGGL(train_project_fill_quant_batched_kernel<1>, gg, bb, 0, st, imgs, pg_start, num_images);
            int rc = launch_tile_pass_batched(1, b, num_images, tile_blocks, uniform ? tiles0 : 0, two_phase, st);
            if (rc != GI2D_OK) return rc;
            AdamStep a[3], aq[3];
            for (int k = 0; k < 3; ++k) {
                a[k] = make_adam_step(lr[k], beta1, beta2, 0.0, eps, step, false);
                aq[k] = make_adam_step(q->lr[k], q->beta1, q->beta2, 0.0, q->eps[k], qstep, false);
            }
            if (model == 2) {
                hipLaunchKernelGGL(train_reduce_update_quant_batched_kernel<2>, gg, bb, 0, st, imgs, pg_start, num_images,
                                   a[0], a[1], a[2], step);
                hipLaunchKernelGGL(train_quant_finish_batched_kernel<2>, gi, b256, 0, st, imgs, pg_start, wpb, a[1], aq[0],
                                   aq[1], aq[2], step);
            } else {
                hipLaunchKernelGGL(train_reduce_update_quant_batched_kernel<1>, gg, bb, 0, st, imgs, pg_start, num_images,
                                   a[0], a[1], a[2], step);
                hipLaunchKernelGGL(train_quant_finish_batched_kernel<1>, gi, b256, 0, st, imgs, pg_start, wpb, a[1], aq[0],
                                   aq[1], aq[2], step);
            }
        }
        batch_pass_end(batch, b, num_images, tile_blocks, st);
        return check_launch("train steps (batched, quantised)");
    }
    if (s0->kind == 2)
        hipLaunchKernelGGL(train_project_fill_batched_kernel<kScaleRot>, gg, bb, 0, st, imgs, pg_start, num_images);
    else if (s0->kind == 0)
        hipLaunchKernelGGL(train_project_fill_batched_kernel<kCholesky>, gg, bb, 0, st, imgs, pg_start, num_images);
    else
        hipLaunchKernelGGL(train_project_fill_batched_kernel<kCovariance>, gg, bb, 0, st, imgs, pg_start, num_images);
    for (int it = 0; it < count; ++it) {
        const int step = first_step + it;
        int rc = launch_tile_pass_batched(1, b, num_images, tile_blocks, uniform ? tiles0 : 0, two_phase, st);
        if (rc != GI2D_OK) return rc;
        AdamStep a[3];
        for (int q = 0; q < 3; ++q) a[q] = make_adam_step(lr[q], beta1, beta2, s0->beta3, eps, step, adan_opt);
        const bool more = it + 1 < count;
#define GI2D_LAUNCH_RU(K, F, A)                                                                                       \
    hipLaunchKernelGGL((train_reduce_update_batched_kernel<K, F, A>), gg, bb, 0, st, imgs, pg_start, num_images, a[0], \
                       a[1], a[2], step)
        GI2D_DISPATCH_RU(s0->kind, more, adan_opt);
#undef GI2D_LAUNCH_RU
    }
    batch_pass_end(batch, b, num_images, tile_blocks, st);
    return check_launch("train steps (batched)");
}

}  // extern "C"
