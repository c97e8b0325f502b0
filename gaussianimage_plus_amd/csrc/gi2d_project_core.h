// Per-gaussian projection math as device functions, shared by the
// reference-shaped kernels (gi2d_project.hip) and the fused fast path (gi2d_fast.hip).
#pragma once
#include "gi2d_common.h"

namespace gi2d {

enum ProjKind { kCholesky = 0, kCovariance = 1, kScaleRot = 2 };

// glm::mat2 product in glm's evaluation order; m = {col0.row0, col0.row1, col1.row0, col1.row1}.
struct M2 {
    float v[4];
};
// fp contraction is switched off in this header: every product/sum is rounded on its own, exactly as the
// CPU oracle evaluates it, so projection results do not depend on the kernel a function is inlined into.
__device__ __forceinline__ M2 mul(const M2 &a, const M2 &b) {
#pragma clang fp contract(off)
    M2 r;
    r.v[0] = a.v[0] * b.v[0] + a.v[2] * b.v[1];
    r.v[1] = a.v[1] * b.v[0] + a.v[3] * b.v[1];
    r.v[2] = a.v[0] * b.v[2] + a.v[2] * b.v[3];
    r.v[3] = a.v[1] * b.v[2] + a.v[3] * b.v[3];
    return r;
}
__device__ __forceinline__ M2 tr(const M2 &a) { return M2{{a.v[0], a.v[2], a.v[1], a.v[3]}}; }

struct ProjOut {
    float2 xy;
    float k0, k1, k2;
    int radius, tiles_hit;
    int mnx, mny, mxx, mxy;  // tile box (valid when tiles_hit > 0)
};

// foward2d.cu:12-69 / :130-187 / :192-288 for gaussian idx; culled gaussians give all-zero outputs.
template <int KIND>
__device__ __forceinline__ ProjOut project_one(int idx, float clip_coe, const float2 *__restrict__ means2d,
                                               const float *__restrict__ p0, const float *__restrict__ p1,
                                               float img_w, float img_h, int tiles_x, int tiles_y,
                                               float radius_clip) {
#pragma clang fp contract(off)
    const float2 m = means2d[idx];
    float cx, cy, cxx, cxy, cyy;
    if (KIND == kCholesky) {  // foward2d.cu:41-48
        cx = 0.5f * img_w * m.x + 0.5f * img_w;
        cy = 0.5f * img_h * m.y + 0.5f * img_h;
        const float l11 = p0[3 * idx], l21 = p0[3 * idx + 1], l22 = p0[3 * idx + 2];
        cxx = l11 * l11;
        cxy = l11 * l21;
        cyy = l21 * l21 + l22 * l22;
    } else if (KIND == kCovariance) {  // foward2d.cu:226-236
        cx = m.x;
        cy = m.y;
        cxx = p0[3 * idx];
        cxy = p0[3 * idx + 1];
        cyy = p0[3 * idx + 2];
    } else {  // foward2d.cu:155-164, helpers.cuh:579-598
        cx = m.x;
        cy = m.y;
        const float rot = p1[idx];
        const float c = cosf(rot), s = sinf(rot);
        const M2 R{{c, -s, s, c}};
        const M2 S{{p0[2 * idx], 0.f, 0.f, p0[2 * idx + 1]}};
        const M2 M = mul(R, S);
        const M2 T = mul(M, tr(M));
        cxx = T.v[0];
        cxy = T.v[1];
        cyy = T.v[3];
    }
    ProjOut o;
    o.xy = make_float2(0.f, 0.f);
    o.k0 = o.k1 = o.k2 = 0.f;
    o.radius = o.tiles_hit = 0;
    o.mnx = o.mny = o.mxx = o.mxy = 0;
    float rmaj, rmin, k0, k1, k2;
    if (cov2d_bounds(cxx, cxy, cyy, clip_coe, k0, k1, k2, rmaj, rmin) && !(rmin < radius_clip)) {
        o.k0 = k0;
        o.k1 = k1;
        o.k2 = k2;
        o.xy = make_float2(cx, cy);
        o.radius = cvt_rzi(rmaj);
        // scale-rot passes the int radius (foward2d.cu:177), the others radius.x (:60, :277)
        tile_bbox(cx, cy, KIND == kScaleRot ? (float)o.radius : rmaj, tiles_x, tiles_y, o.mnx, o.mny, o.mxx,
                  o.mxy);
        const int area = (int)((unsigned)(o.mxx - o.mnx) * (unsigned)(o.mxy - o.mny));
        if (area > 0) o.tiles_hit = area;
    }
    return o;
}

// helpers.cuh:384-395 cov2d_to_conic_vjp
__device__ __forceinline__ void conic_vjp(const float *conic, const float *vc, float &g11, float &g12,
                                          float &g22) {
    const M2 X{{conic[0], conic[1], conic[1], conic[2]}};
    const M2 nX{{-conic[0], -conic[1], -conic[1], -conic[2]}};
    const M2 G{{vc[0], vc[1], vc[1], vc[2]}};
    const M2 s = mul(mul(nX, G), X);
    g11 = s.v[0];
    g12 = s.v[2] + s.v[1];
    g22 = s.v[3];
}

struct ProjGrad {
    float g11, g12, g22;  // v_cov2d
    float2 v_mean;
    float o0, o1, o2;     // v_L / v_cov (3) or v_scale (2) + v_rot
};

// backward2d.cu:8-51 / :53-101 / :157-214 for one gaussian (radius > 0), given its conic and the
// rasterizer's v_xy / v_conic.  The Cholesky and scale-rot forms double-count the off-diagonal on purpose.
template <int KIND>
__device__ __forceinline__ ProjGrad project_bwd_one(int idx, const float *__restrict__ p0,
                                                    const float *__restrict__ p1, float img_w, float img_h,
                                                    const float conic[3], float2 vxy, const float v_conic[3]) {
#pragma clang fp contract(off)
    ProjGrad r;
    conic_vjp(conic, v_conic, r.g11, r.g12, r.g22);
    const float g11 = r.g11, g12 = r.g12, g22 = r.g22;
    if (KIND == kCholesky) {  // backward2d.cu:39-49
        const float l11 = p0[3 * idx], l21 = p0[3 * idx + 1], l22 = p0[3 * idx + 2];
        r.o0 = 2 * l11 * g11 + 2 * g12 * l21;
        r.o1 = 2 * l11 * g12 + 2 * l21 * g22;
        r.o2 = 2 * l22 * g22;
        r.v_mean = make_float2(vxy.x * (0.5f * img_w), vxy.y * (0.5f * img_h));
    } else if (KIND == kCovariance) {  // backward2d.cu:194-206
        r.o0 = g11;
        r.o1 = g12;
        r.o2 = g22;
        r.v_mean = vxy;
    } else {  // backward2d.cu:72-99
        const float rot = p1[idx];
        const float c = cosf(rot), s = sinf(rot);
        const float sx = p0[2 * idx], sy = p0[2 * idx + 1];
        const M2 R{{c, -s, s, c}}, Rg{{-s, -c, c, -s}}, S{{sx, 0.f, 0.f, sy}};
        const M2 M = mul(R, S);
        const M2 A = mul(mul(Rg, S), tr(M));
        const M2 B = mul(mul(M, tr(S)), tr(Rg));
        const M2 sgx = mul(mul(R, M2{{2.f * sx, 0.f, 0.f, 0.f}}), tr(R));
        const M2 sgy = mul(mul(R, M2{{0.f, 0.f, 0.f, 2.f * sy}}), tr(R));
        r.o0 = g11 * sgx.v[0] + 2 * g12 * sgx.v[1] + g22 * sgx.v[3];
        r.o1 = g11 * sgy.v[0] + 2 * g12 * sgy.v[1] + g22 * sgy.v[3];
        r.o2 = g11 * (A.v[0] + B.v[0]) + 2 * g12 * (A.v[1] + B.v[1]) + g22 * (A.v[3] + B.v[3]);
        r.v_mean = vxy;
    }
    return r;
}

__device__ __forceinline__ void store_proj_grad(int idx, bool scale_rot, const ProjGrad &r,
                                                float *__restrict__ v_cov2d, float2 *__restrict__ v_mean2d,
                                                float *__restrict__ v_p0, float *__restrict__ v_p1) {
    if (v_cov2d) {
        v_cov2d[3 * idx] = r.g11;
        v_cov2d[3 * idx + 1] = r.g12;
        v_cov2d[3 * idx + 2] = r.g22;
    }
    v_mean2d[idx] = r.v_mean;
    if (scale_rot) {
        v_p0[2 * idx] = r.o0;
        v_p0[2 * idx + 1] = r.o1;
        v_p1[idx] = r.o2;
    } else {
        v_p0[3 * idx] = r.o0;
        v_p0[3 * idx + 1] = r.o1;
        v_p0[3 * idx + 2] = r.o2;
    }
}

}  // namespace gi2d
