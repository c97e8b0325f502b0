// Per-gaussian projection kernels: one lane per gaussian.
// HBM-bound elementwise work: 20 B in / 32 B out per gaussian (fwd), 48 B in / 32 B out (bwd).
// Unlike the reference (bindings.cu:1347-1356: five torch::zeros memsets + kernel) every
// output element is written here, culled rows as zeros, so no memset launches are needed.
#include "gi2d_project_core.h"

namespace gi2d {

template <int KIND>
__global__ __launch_bounds__(256) void project_fwd_kernel(
    int n, float clip_coe, const float2 *__restrict__ means2d, const float *__restrict__ p0,
    const float *__restrict__ p1, float img_w, float img_h, int tiles_x, int tiles_y,
    float radius_clip, float2 *__restrict__ xys, float *__restrict__ depths,
    int32_t *__restrict__ radii, float *__restrict__ conics, int32_t *__restrict__ num_tiles_hit) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const ProjOut o = project_one<KIND>(idx, clip_coe, means2d, p0, p1, img_w, img_h, tiles_x, tiles_y, radius_clip);
    xys[idx] = o.xy;
    depths[idx] = 0.f;
    radii[idx] = o.radius;
    conics[3 * idx] = o.k0;
    conics[3 * idx + 1] = o.k1;
    conics[3 * idx + 2] = o.k2;
    num_tiles_hit[idx] = o.tiles_hit;
}

template <int KIND>
__global__ __launch_bounds__(256) void project_bwd_kernel(
    int n, const float *__restrict__ p0, const float *__restrict__ p1, float img_w, float img_h,
    const int32_t *__restrict__ radii, const float *__restrict__ conics,
    const float2 *__restrict__ v_xy, const float *__restrict__ v_conic, float *__restrict__ v_cov2d,
    float2 *__restrict__ v_mean2d, float *__restrict__ v_p0, float *__restrict__ v_p1) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    ProjGrad r;
    r.g11 = r.g12 = r.g22 = r.o0 = r.o1 = r.o2 = 0.f;
    r.v_mean = make_float2(0.f, 0.f);
    if (radii[idx] > 0) {
        const float conic[3] = {conics[3 * idx], conics[3 * idx + 1], conics[3 * idx + 2]};
        const float vc[3] = {v_conic[3 * idx], v_conic[3 * idx + 1], v_conic[3 * idx + 2]};
        r = project_bwd_one<KIND>(idx, p0, p1, img_w, img_h, conic, v_xy[idx], vc);
    }
    store_proj_grad(idx, KIND == kScaleRot, r, v_cov2d, v_mean2d, v_p0, v_p1);
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
