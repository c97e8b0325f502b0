// Per-gaussian projection kernels: one lane per gaussian.
// HBM-bound elementwise work: 20 B in / 32 B out per gaussian (fwd), 48 B in / 32 B out (bwd).
// Unlike the reference (bindings.cu:1347-1356: five torch::zeros memsets + kernel) every
// output element is written here, culled rows as zeros, so no memset launches are needed.
#include "gi2d_common.h"

namespace gi2d {

enum ProjKind { kCholesky = 0, kCovariance = 1, kScaleRot = 2 };

// glm::mat2 product in glm's evaluation order; m = {col0.row0, col0.row1, col1.row0, col1.row1}.
struct M2 {
    float v[4];
};
__device__ __forceinline__ M2 mul(const M2 &a, const M2 &b) {
    M2 r;
    r.v[0] = a.v[0] * b.v[0] + a.v[2] * b.v[1];
    r.v[1] = a.v[1] * b.v[0] + a.v[3] * b.v[1];
    r.v[2] = a.v[0] * b.v[2] + a.v[2] * b.v[3];
    r.v[3] = a.v[1] * b.v[2] + a.v[3] * b.v[3];
    return r;
}
__device__ __forceinline__ M2 tr(const M2 &a) { return M2{{a.v[0], a.v[2], a.v[1], a.v[3]}}; }

template <int KIND>
__global__ __launch_bounds__(256) void project_fwd_kernel(
    int n, float clip_coe, const float2 *__restrict__ means2d, const float *__restrict__ p0,
    const float *__restrict__ p1, float img_w, float img_h, int tiles_x, int tiles_y,
    float radius_clip, float2 *__restrict__ xys, float *__restrict__ depths,
    int32_t *__restrict__ radii, float *__restrict__ conics, int32_t *__restrict__ num_tiles_hit) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
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
    float2 o_xy = make_float2(0.f, 0.f);
    float k0 = 0.f, k1 = 0.f, k2 = 0.f;
    int o_rad = 0, o_hit = 0;
    float rmaj, rmin;
    if (cov2d_bounds(cxx, cxy, cyy, clip_coe, k0, k1, k2, rmaj, rmin) && !(rmin < radius_clip)) {
        o_xy = make_float2(cx, cy);
        o_rad = cvt_rzi(rmaj);
        int mnx, mny, mxx, mxy;
        // scale-rot passes the int radius (foward2d.cu:177), the others radius.x (:60, :277)
        tile_bbox(cx, cy, KIND == kScaleRot ? (float)o_rad : rmaj, tiles_x, tiles_y, mnx, mny, mxx,
                  mxy);
        const int area = (int)((unsigned)(mxx - mnx) * (unsigned)(mxy - mny));
        if (area > 0) o_hit = area;
    } else {
        k0 = k1 = k2 = 0.f;
    }
    xys[idx] = o_xy;
    depths[idx] = 0.f;
    radii[idx] = o_rad;
    conics[3 * idx] = k0;
    conics[3 * idx + 1] = k1;
    conics[3 * idx + 2] = k2;
    num_tiles_hit[idx] = o_hit;
}

// helpers.cuh:384-395 cov2d_to_conic_vjp
__device__ __forceinline__ void conic_vjp(const float *conic, const float *vc, float &g11,
                                          float &g12, float &g22) {
    const M2 X{{conic[0], conic[1], conic[1], conic[2]}};
    const M2 nX{{-conic[0], -conic[1], -conic[1], -conic[2]}};
    const M2 G{{vc[0], vc[1], vc[1], vc[2]}};
    const M2 s = mul(mul(nX, G), X);
    g11 = s.v[0];
    g12 = s.v[2] + s.v[1];
    g22 = s.v[3];
}

template <int KIND>
__global__ __launch_bounds__(256) void project_bwd_kernel(
    int n, const float *__restrict__ p0, const float *__restrict__ p1, float img_w, float img_h,
    const int32_t *__restrict__ radii, const float *__restrict__ conics,
    const float2 *__restrict__ v_xy, const float *__restrict__ v_conic, float *__restrict__ v_cov2d,
    float2 *__restrict__ v_mean2d, float *__restrict__ v_p0, float *__restrict__ v_p1) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float g11 = 0.f, g12 = 0.f, g22 = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    float2 vm = make_float2(0.f, 0.f);
    if (radii[idx] > 0) {
        conic_vjp(conics + 3 * idx, v_conic + 3 * idx, g11, g12, g22);
        const float2 vxy = v_xy[idx];
        if (KIND == kCholesky) {  // backward2d.cu:39-49 (double-counted off-diagonal, on purpose)
            const float l11 = p0[3 * idx], l21 = p0[3 * idx + 1], l22 = p0[3 * idx + 2];
            o0 = 2 * l11 * g11 + 2 * g12 * l21;
            o1 = 2 * l11 * g12 + 2 * l21 * g22;
            o2 = 2 * l22 * g22;
            vm = make_float2(vxy.x * (0.5f * img_w), vxy.y * (0.5f * img_h));
        } else if (KIND == kCovariance) {  // backward2d.cu:194-206
            o0 = g11;
            o1 = g12;
            o2 = g22;
            vm = vxy;
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
            o0 = g11 * sgx.v[0] + 2 * g12 * sgx.v[1] + g22 * sgx.v[3];
            o1 = g11 * sgy.v[0] + 2 * g12 * sgy.v[1] + g22 * sgy.v[3];
            o2 = g11 * (A.v[0] + B.v[0]) + 2 * g12 * (A.v[1] + B.v[1]) + g22 * (A.v[3] + B.v[3]);
            vm = vxy;
        }
    }
    v_cov2d[3 * idx] = g11;
    v_cov2d[3 * idx + 1] = g12;
    v_cov2d[3 * idx + 2] = g22;
    v_mean2d[idx] = vm;
    if (KIND == kScaleRot) {
        v_p0[2 * idx] = o0;
        v_p0[2 * idx + 1] = o1;
        v_p1[idx] = o2;
    } else {
        v_p0[3 * idx] = o0;
        v_p0[3 * idx + 1] = o1;
        v_p0[3 * idx + 2] = o2;
    }
}

// bindings.cu:21-39 compute_cov2d_bounds_kernel (zeros where the reference leaves garbage).
__global__ __launch_bounds__(256) void cov2d_bounds_kernel(int n, float clip_coe,
                                                           const float *__restrict__ cov,
                                                           float *__restrict__ conics,
                                                           float *__restrict__ radii) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float k0 = 0.f, k1 = 0.f, k2 = 0.f, rmaj = 0.f, rmin = 0.f;
    if (!cov2d_bounds(cov[3 * idx], cov[3 * idx + 1], cov[3 * idx + 2], clip_coe, k0, k1, k2, rmaj,
                      rmin)) {
        k0 = k1 = k2 = rmaj = 0.f;
    }
    conics[3 * idx] = k0;
    conics[3 * idx + 1] = k1;
    conics[3 * idx + 2] = k2;
    radii[idx] = rmaj;
}

template <int KIND>
static int launch_fwd(int n, float clip_coe, const float *means2d, const float *p0, const float *p1,
                      unsigned h, unsigned w, int tiles_x, int tiles_y, float radius_clip, float *xys,
                      float *depths, int32_t *radii, float *conics, int32_t *nth, gi2d_stream_t st) {
    if (n < 0 || tiles_x < 0 || tiles_y < 0) {
        set_error("project forward: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return GI2D_OK;
    if (!means2d || !p0 || !xys || !depths || !radii || !conics || !nth) {
        set_error("project forward: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(project_fwd_kernel<KIND>, dim3((n + 255) / 256), dim3(256), 0,
                       (hipStream_t)st, n, clip_coe, (const float2 *)means2d, p0, p1, (float)w,
                       (float)h, tiles_x, tiles_y, radius_clip, (float2 *)xys, depths, radii, conics,
                       nth);
    return check_launch("project forward");
}

template <int KIND>
static int launch_bwd(int n, const float *p0, const float *p1, unsigned h, unsigned w,
                      const int32_t *radii, const float *conics, const float *v_xy,
                      const float *v_conic, float *v_cov2d, float *v_mean2d, float *v_p0, float *v_p1,
                      gi2d_stream_t st) {
    if (n < 0) {
        set_error("project backward: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return GI2D_OK;
    if (!p0 || !radii || !conics || !v_xy || !v_conic || !v_cov2d || !v_mean2d || !v_p0) {
        set_error("project backward: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(project_bwd_kernel<KIND>, dim3((n + 255) / 256), dim3(256), 0,
                       (hipStream_t)st, n, p0, p1, (float)w, (float)h, radii, conics,
                       (const float2 *)v_xy, v_conic, v_cov2d, (float2 *)v_mean2d, v_p0, v_p1);
    return check_launch("project backward");
}

}  // namespace gi2d

using namespace gi2d;

extern "C" {

int gi2d_project_gaussians_2d_forward(int n, float clip_coe, const float *means2d, const float *L,
                                      unsigned h, unsigned w, int tiles_x, int tiles_y,
                                      float clip_thresh, float radius_clip, float *xys,
                                      float *depths, int32_t *radii, float *conics,
                                      int32_t *num_tiles_hit, gi2d_stream_t st) {
    (void)clip_thresh;
    return launch_fwd<kCholesky>(n, clip_coe, means2d, L, nullptr, h, w, tiles_x, tiles_y,
                                 radius_clip, xys, depths, radii, conics, num_tiles_hit, st);
}
int gi2d_project_gaussians_2d_covariance_forward(int n, float clip_coe, const float *means2d,
                                                 const float *cov, unsigned h, unsigned w,
                                                 int tiles_x, int tiles_y, float clip_thresh,
                                                 float radius_clip, float *xys, float *depths,
                                                 int32_t *radii, float *conics,
                                                 int32_t *num_tiles_hit, gi2d_stream_t st) {
    (void)clip_thresh;
    return launch_fwd<kCovariance>(n, clip_coe, means2d, cov, nullptr, h, w, tiles_x, tiles_y,
                                   radius_clip, xys, depths, radii, conics, num_tiles_hit, st);
}
int gi2d_project_gaussians_2d_scale_rot_forward(int n, float clip_coe, const float *means2d,
                                                const float *scales, const float *rot, unsigned h,
                                                unsigned w, int tiles_x, int tiles_y,
                                                float clip_thresh, float radius_clip, float *xys,
                                                float *depths, int32_t *radii, float *conics,
                                                int32_t *num_tiles_hit, gi2d_stream_t st) {
    (void)clip_thresh;
    if (n > 0 && !rot) {
        set_error("project scale_rot forward: null rotation");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    return launch_fwd<kScaleRot>(n, clip_coe, means2d, scales, rot, h, w, tiles_x, tiles_y,
                                 radius_clip, xys, depths, radii, conics, num_tiles_hit, st);
}

int gi2d_project_gaussians_2d_backward(int n, const float *means2d, const float *L, unsigned h,
                                       unsigned w, const int32_t *radii, const float *conics,
                                       const float *v_xy, const float *v_depth,
                                       const float *v_conic, float *v_cov2d, float *v_mean2d,
                                       float *v_L, gi2d_stream_t st) {
    (void)means2d;
    (void)v_depth;
    return launch_bwd<kCholesky>(n, L, nullptr, h, w, radii, conics, v_xy, v_conic, v_cov2d,
                                 v_mean2d, v_L, nullptr, st);
}
int gi2d_project_gaussians_2d_covariance_backward(int n, const float *means2d, const float *cov,
                                                  unsigned h, unsigned w, const int32_t *radii,
                                                  const float *conics, const float *v_xy,
                                                  const float *v_depth, const float *v_conic,
                                                  float *v_cov2d, float *v_mean2d, float *v_cov,
                                                  gi2d_stream_t st) {
    (void)means2d;
    (void)v_depth;
    return launch_bwd<kCovariance>(n, cov, nullptr, h, w, radii, conics, v_xy, v_conic, v_cov2d,
                                   v_mean2d, v_cov, nullptr, st);
}
int gi2d_project_gaussians_2d_scale_rot_backward(int n, const float *means2d, const float *scales,
                                                 const float *rot, unsigned h, unsigned w,
                                                 const int32_t *radii, const float *conics,
                                                 const float *v_xy, const float *v_depth,
                                                 const float *v_conic, float *v_cov2d,
                                                 float *v_mean2d, float *v_scale, float *v_rot,
                                                 gi2d_stream_t st) {
    (void)means2d;
    (void)v_depth;
    if (n > 0 && (!rot || !v_rot)) {
        set_error("project scale_rot backward: null rotation");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    return launch_bwd<kScaleRot>(n, scales, rot, h, w, radii, conics, v_xy, v_conic, v_cov2d,
                                 v_mean2d, v_scale, v_rot, st);
}

int gi2d_compute_cov2d_bounds(int n, float clip_coe, const float *cov, float *conics, float *radii,
                              gi2d_stream_t st) {
    if (n < 0) {
        set_error("compute_cov2d_bounds: negative size");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return GI2D_OK;
    if (!cov || !conics || !radii) {
        set_error("compute_cov2d_bounds: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(cov2d_bounds_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, n,
                       clip_coe, cov, conics, radii);
    return check_launch("compute_cov2d_bounds");
}

}  // extern "C"
