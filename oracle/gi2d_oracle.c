/*
 * gi2d_oracle.c -- CPU restatement of GaussianImage++'s 2D Gaussian hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gaussianimage_plus_amd/)
 * may import, link or call this file.  It is the checker used by tests/, by
 * __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg.
 *
 * PARITY STATUS: pinned by outputs of the reference itself run in the dev container (the
 * reference holds no test or fixture of the 2D path and its kernels are CUDA-only, SURVEY.md
 * section 8c; what it can produce on a CPU is its Python layer gsplat/gsplat/_torch_impl.py):
 *   - tests/golden/ref_vectors.npz (make_ref_vectors.py): compute_cov2d_bounds, get_tile_bbox,
 *     map_gaussian_to_intersects, get_tile_bin_edges -- the CPU side of the reference's
 *     test_cov2d_bounds / test_map_gaussians / test_get_tile_bin_edges -- and the projection
 *     forward of the three parameterisations through them, committed as arrays;
 *   - tests/golden/refras_vectors.npz (make_refras_vectors.py): the rasterizer forward AND backward
 *     from the reference's own CPU rasterizer, _torch_impl.rasterize_forward (:354-421), called
 *     once per gaussian (one contributor per list: T = 1, its term of the sum rasterizer), the
 *     images summed, the gradients by torch autograd through the same calls;
 *   - float64 autograd of the inverse-covariance map for the projection VJPs, with the
 *     documented double count of the off-diagonal term.
 * tests/test_ref_vectors_cpu.py holds this file to those arrays.  Outside any CPU pin: the rounding
 * of CUDA's __expf next to the 1/255 cut-off (pairs inside a band around it are flagged by
 * pair_eval and skipped by the comparisons).  All citations are relative to
 * /root/reference/gsplat/gsplat/cuda/csrc/.
 *
 * Plain C11 + OpenMP.  fp32 arithmetic follows the CUDA expressions operand by
 * operand (no FMA contraction is requested); gradient accumulation, which the
 * reference performs with order-unspecified float atomics, is done in double and
 * rounded once so that the oracle is the centre of the set of valid float results.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GI2D_BLOCK_X 16 /* config.h:1 */
#define GI2D_BLOCK_Y 16 /* config.h:2 */
#define GI2D_BLOCK_SIZE 256 /* config.h:3 */

/* CUDA float->int conversion (cvt.rzi.s32.f32): truncates, saturates, NaN -> 0.
 * AMD v_cvt_i32_f32 behaves the same; plain C would be UB out of range. */
static inline int cvt_rzi(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int)x;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* helpers.cuh:16-30 get_bbox + :32-50 get_tile_bbox.  Inclusive min, exclusive max,
 * truncation toward zero, clamp to [0, bound]. */
static inline void tile_bbox(float cx, float cy, float pix_radius, int tiles_x, int tiles_y,
                             unsigned *min_x, unsigned *min_y, unsigned *max_x, unsigned *max_y) {
    float tcx = cx / (float)GI2D_BLOCK_X, tcy = cy / (float)GI2D_BLOCK_Y;
    float trx = pix_radius / (float)GI2D_BLOCK_X, try_ = pix_radius / (float)GI2D_BLOCK_Y;
    *min_x = (unsigned)imin(imax(0, cvt_rzi(tcx - trx)), tiles_x);
    *max_x = (unsigned)imin(imax(0, cvt_rzi(tcx + trx + 1)), tiles_x);
    *min_y = (unsigned)imin(imax(0, cvt_rzi(tcy - try_)), tiles_y);
    *max_y = (unsigned)imin(imax(0, cvt_rzi(tcy + try_ + 1)), tiles_y);
}

/* helpers.cuh:179-206 compute_cov2d_bounds.  det==0 -> false; no clamp of det
 * (unlike _torch_impl.py:197-199); fmaxf semantics of CUDA max(float,float). */
static inline int cov2d_bounds(float cx, float cy, float cz, float clip_coe, float conic[3],
                               float radius[2]) {
    float det = cx * cz - cy * cy;
    if (det == 0.f) return 0;
    float inv_det = 1.f / det;
    conic[0] = cz * inv_det;
    conic[1] = -cy * inv_det;
    conic[2] = cx * inv_det;
    float b = 0.5f * (cx + cz);
    float v1 = b + sqrtf(fmaxf(0.1f, b * b - det));
    float v2 = b - sqrtf(fmaxf(0.1f, b * b - det));
    radius[0] = ceilf(clip_coe * sqrtf(fmaxf(v1, v2)));
    radius[1] = ceilf(clip_coe * sqrtf(fminf(v1, v2))); /* NaN when min(v1,v2) < 0 */
    return 1;
}

/* bindings.cu:21-39 compute_cov2d_bounds_kernel (radii = radius.x as float [N,1]).
 * The reference ignores the det==0 return and stores uninitialised values; the
 * oracle stores zeros there. */
void gi2d_oracle_compute_cov2d_bounds(int n, float clip_coe, const float *cov2d, float *conics,
                                      float *radii) {
    for (int i = 0; i < n; ++i) {
        float c[3] = {0, 0, 0}, r[2] = {0, 0};
        cov2d_bounds(cov2d[3 * i], cov2d[3 * i + 1], cov2d[3 * i + 2], clip_coe, c, r);
        conics[3 * i] = c[0];
        conics[3 * i + 1] = c[1];
        conics[3 * i + 2] = c[2];
        radii[i] = r[0];
    }
}

/* Common tail of the three projection kernels (foward2d.cu:51-68 / :169-185 / :241-285).
 * `bbox_uses_int_radius`: the scale-rot kernel passes radii[idx] (int) to get_tile_bbox
 * (foward2d.cu:177) while the other two pass radius.x (float, already ceil'd) (:60, :277);
 * the values coincide whenever radius.x fits an int. */
static inline void project_tail(int idx, float centre_x, float centre_y, float cxx, float cxy,
                                float cyy, float clip_coe, float radius_clip, int tiles_x,
                                int tiles_y, int bbox_uses_int_radius, float *xys, int *radii,
                                float *conics, int *num_tiles_hit) {
    float conic[3], radius[2];
    if (!cov2d_bounds(cxx, cxy, cyy, clip_coe, conic, radius)) return;
    if (radius[1] < radius_clip) return; /* false for NaN: gaussian kept */
    conics[3 * idx] = conic[0];
    conics[3 * idx + 1] = conic[1];
    conics[3 * idx + 2] = conic[2];
    xys[2 * idx] = centre_x;
    xys[2 * idx + 1] = centre_y;
    radii[idx] = cvt_rzi(radius[0]);
    unsigned mnx, mny, mxx, mxy;
    tile_bbox(centre_x, centre_y, bbox_uses_int_radius ? (float)radii[idx] : radius[0], tiles_x,
              tiles_y, &mnx, &mny, &mxx, &mxy);
    int32_t area = (int32_t)((mxx - mnx) * (mxy - mny));
    if (area <= 0) return;
    num_tiles_hit[idx] = area;
}

/* foward2d.cu:12-69 project_gaussians_2d_forward_kernel (Cholesky, means in NDC (-1,1)).
 * Outputs are zero-initialised as bindings.cu:1347-1356 does. */
void gi2d_oracle_project_cholesky_fwd(int n, float clip_coe, const float *means2d, const float *L,
                                      int img_h, int img_w, int tiles_x, int tiles_y,
                                      float radius_clip, float *xys, float *depths, int *radii,
                                      float *conics, int *num_tiles_hit) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        xys[2 * i] = xys[2 * i + 1] = 0.f;
        conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0.f;
        radii[i] = 0;
        num_tiles_hit[i] = 0;
        depths[i] = 0.f;
        float cx = 0.5f * (float)img_w * means2d[2 * i] + 0.5f * (float)img_w;     /* :41 */
        float cy = 0.5f * (float)img_h * means2d[2 * i + 1] + 0.5f * (float)img_h; /* :42 */
        float l11 = L[3 * i], l21 = L[3 * i + 1], l22 = L[3 * i + 2];
        project_tail(i, cx, cy, l11 * l11, l11 * l21, l21 * l21 + l22 * l22, clip_coe, radius_clip,
                     tiles_x, tiles_y, 0, xys, radii, conics, num_tiles_hit); /* :48 */
    }
}

/* foward2d.cu:192-288 project_gaussians_2d_covariance_forward_kernel (means in pixels,
 * covariance given directly, :226,:236). */
void gi2d_oracle_project_covariance_fwd(int n, float clip_coe, const float *means2d,
                                        const float *cov, int img_h, int img_w, int tiles_x,
                                        int tiles_y, float radius_clip, float *xys, float *depths,
                                        int *radii, float *conics, int *num_tiles_hit) {
    (void)img_h;
    (void)img_w;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        xys[2 * i] = xys[2 * i + 1] = 0.f;
        conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0.f;
        radii[i] = 0;
        num_tiles_hit[i] = 0;
        depths[i] = 0.f;
        project_tail(i, means2d[2 * i], means2d[2 * i + 1], cov[3 * i], cov[3 * i + 1],
                     cov[3 * i + 2], clip_coe, radius_clip, tiles_x, tiles_y, 0, xys, radii, conics,
                     num_tiles_hit);
    }
}

/* 2x2 product in glm's evaluation order (column-major m[col][row]); a,b,out are
 * {m00,m01,m10,m11} = {col0.row0, col0.row1, col1.row0, col1.row1}. */
static inline void mat2_mul(const float a[4], const float b[4], float out[4]) {
    out[0] = a[0] * b[0] + a[2] * b[1];
    out[1] = a[1] * b[0] + a[3] * b[1];
    out[2] = a[0] * b[2] + a[2] * b[3];
    out[3] = a[1] * b[2] + a[3] * b[3];
}
static inline void mat2_t(const float a[4], float out[4]) {
    out[0] = a[0];
    out[1] = a[2];
    out[2] = a[1];
    out[3] = a[3];
}
/* helpers.cuh:587-598 rotmat2d: R[0][0]=R[1][1]=cos, R[0][1]=-sin, R[1][0]=sin. */
static inline void rotmat2d(float rot, float r[4]) {
    float c = cosf(rot), s = sinf(rot);
    r[0] = c;
    r[1] = -s;
    r[2] = s;
    r[3] = c;
}
/* helpers.cuh:600-611 rotmat2d_gradient. */
static inline void rotmat2d_grad(float rot, float r[4]) {
    float c = cosf(rot), s = sinf(rot);
    r[0] = -s;
    r[1] = -c;
    r[2] = c;
    r[3] = -s;
}

/* foward2d.cu:130-187 project_gaussians_2d_scale_rot_forward_kernel (means in pixels,
 * Sigma = (R S)(R S)^T, :158-164). */
void gi2d_oracle_project_scale_rot_fwd(int n, float clip_coe, const float *means2d,
                                       const float *scales, const float *rot, int img_h, int img_w,
                                       int tiles_x, int tiles_y, float radius_clip, float *xys,
                                       float *depths, int *radii, float *conics,
                                       int *num_tiles_hit) {
    (void)img_h;
    (void)img_w;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        xys[2 * i] = xys[2 * i + 1] = 0.f;
        conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0.f;
        radii[i] = 0;
        num_tiles_hit[i] = 0;
        depths[i] = 0.f;
        float R[4], S[4] = {scales[2 * i], 0.f, 0.f, scales[2 * i + 1]}, M[4], Mt[4], T[4];
        rotmat2d(rot[i], R);
        mat2_mul(R, S, M);
        mat2_t(M, Mt);
        mat2_mul(M, Mt, T);
        /* cov2d = (tmp[0][0], tmp[0][1], tmp[1][1]) */
        project_tail(i, means2d[2 * i], means2d[2 * i + 1], T[0], T[1], T[3], clip_coe, radius_clip,
                     tiles_x, tiles_y, 1, xys, radii, conics, num_tiles_hit);
    }
}

/* helpers.cuh:384-395 cov2d_to_conic_vjp: v_Sigma = -X G X, off-diagonal summed. */
static inline void conic_vjp(const float conic[3], const float vc[3], float v_cov2d[3]) {
    float X[4] = {conic[0], conic[1], conic[1], conic[2]};
    float G[4] = {vc[0], vc[1], vc[1], vc[2]};
    float nX[4] = {-X[0], -X[1], -X[2], -X[3]};
    float t[4], s[4];
    mat2_mul(nX, G, t);
    mat2_mul(t, X, s);
    v_cov2d[0] = s[0];
    v_cov2d[1] = s[2] + s[1]; /* v_Sigma[1][0] + v_Sigma[0][1] */
    v_cov2d[2] = s[3];
}

/* backward2d.cu:8-51 project_gaussians_2d_backward_kernel.  Reference-faithful:
 * G_12 already holds the summed off-diagonal and is used with a factor 2 again
 * (:39-40) -- not the true gradient; replicated on purpose.
 * Outputs zero-initialised (bindings.cu:1537-1542); rows with radii<=0 stay zero. */
void gi2d_oracle_project_cholesky_bwd(int n, const float *L, int img_h, int img_w,
                                      const int *radii, const float *conics, const float *v_xy,
                                      const float *v_conic, float *v_cov2d, float *v_mean2d,
                                      float *v_L) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        v_cov2d[3 * i] = v_cov2d[3 * i + 1] = v_cov2d[3 * i + 2] = 0.f;
        v_L[3 * i] = v_L[3 * i + 1] = v_L[3 * i + 2] = 0.f;
        v_mean2d[2 * i] = v_mean2d[2 * i + 1] = 0.f;
        if (radii[i] <= 0) continue;
        conic_vjp(conics + 3 * i, v_conic + 3 * i, v_cov2d + 3 * i);
        float G11 = v_cov2d[3 * i], G12 = v_cov2d[3 * i + 1], G22 = v_cov2d[3 * i + 2];
        float l11 = L[3 * i], l21 = L[3 * i + 1], l22 = L[3 * i + 2];
        v_L[3 * i] = 2 * l11 * G11 + 2 * G12 * l21;
        v_L[3 * i + 1] = 2 * l11 * G12 + 2 * l21 * G22;
        v_L[3 * i + 2] = 2 * l22 * G22;
        v_mean2d[2 * i] = v_xy[2 * i] * (0.5f * (float)img_w);
        v_mean2d[2 * i + 1] = v_xy[2 * i + 1] * (0.5f * (float)img_h);
    }
}

/* backward2d.cu:157-214 project_gaussians_2d_covariance_backward_kernel (exact). */
void gi2d_oracle_project_covariance_bwd(int n, const int *radii, const float *conics,
                                        const float *v_xy, const float *v_conic, float *v_cov2d,
                                        float *v_mean2d, float *v_cov) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        v_cov2d[3 * i] = v_cov2d[3 * i + 1] = v_cov2d[3 * i + 2] = 0.f;
        v_cov[3 * i] = v_cov[3 * i + 1] = v_cov[3 * i + 2] = 0.f;
        v_mean2d[2 * i] = v_mean2d[2 * i + 1] = 0.f;
        if (radii[i] <= 0) continue;
        conic_vjp(conics + 3 * i, v_conic + 3 * i, v_cov2d + 3 * i);
        v_cov[3 * i] = v_cov2d[3 * i];
        v_cov[3 * i + 1] = v_cov2d[3 * i + 1];
        v_cov[3 * i + 2] = v_cov2d[3 * i + 2];
        v_mean2d[2 * i] = v_xy[2 * i];
        v_mean2d[2 * i + 1] = v_xy[2 * i + 1];
    }
}

/* backward2d.cu:53-101 project_gaussians_2d_scale_rot_backward_kernel. */
void gi2d_oracle_project_scale_rot_bwd(int n, const float *scales, const float *rot,
                                       const int *radii, const float *conics, const float *v_xy,
                                       const float *v_conic, float *v_cov2d, float *v_mean2d,
                                       float *v_scale, float *v_rot) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        v_cov2d[3 * i] = v_cov2d[3 * i + 1] = v_cov2d[3 * i + 2] = 0.f;
        v_scale[2 * i] = v_scale[2 * i + 1] = 0.f;
        v_rot[i] = 0.f;
        v_mean2d[2 * i] = v_mean2d[2 * i + 1] = 0.f;
        if (radii[i] <= 0) continue;
        conic_vjp(conics + 3 * i, v_conic + 3 * i, v_cov2d + 3 * i);
        float R[4], Rg[4], Rt[4], Rgt[4], S[4] = {scales[2 * i], 0.f, 0.f, scales[2 * i + 1]};
        float M[4], Mt[4], St[4], a[4], b[4], c[4], d[4], theta_g[4];
        rotmat2d(rot[i], R);
        rotmat2d_grad(rot[i], Rg);
        mat2_mul(R, S, M);
        mat2_t(M, Mt);
        mat2_t(S, St);
        mat2_t(R, Rt);
        mat2_t(Rg, Rgt);
        /* theta_g = R_g*S*M^T + M*S^T*R_g^T  (:76) */
        mat2_mul(Rg, S, a);
        mat2_mul(a, Mt, b);
        mat2_mul(M, St, c);
        mat2_mul(c, Rgt, d);
        for (int k = 0; k < 4; ++k) theta_g[k] = b[k] + d[k];
        float sxg[4] = {2.f * scales[2 * i], 0.f, 0.f, 0.f};
        float syg[4] = {0.f, 0.f, 0.f, 2.f * scales[2 * i + 1]};
        float sigx[4], sigy[4];
        mat2_mul(R, sxg, a);
        mat2_mul(a, Rt, sigx);
        mat2_mul(R, syg, a);
        mat2_mul(a, Rt, sigy);
        float G11 = v_cov2d[3 * i], G12 = v_cov2d[3 * i + 1], G22 = v_cov2d[3 * i + 2];
        v_scale[2 * i] = G11 * sigx[0] + 2 * G12 * sigx[1] + G22 * sigx[3];
        v_scale[2 * i + 1] = G11 * sigy[0] + 2 * G12 * sigy[1] + G22 * sigy[3];
        v_rot[i] = G11 * theta_g[0] + 2 * G12 * theta_g[1] + G22 * theta_g[3];
        v_mean2d[2 * i] = v_xy[2 * i];
        v_mean2d[2 * i + 1] = v_xy[2 * i + 1];
    }
}

/* gsplat/gsplat/utils.py:248-249 compute_cumulative_intersects: inclusive int32 cumsum,
 * returns the total. */
int gi2d_oracle_cumsum(int n, const int *num_tiles_hit, int *cum_tiles_hit) {
    int32_t acc = 0;
    for (int i = 0; i < n; ++i) {
        acc += num_tiles_hit[i];
        cum_tiles_hit[i] = acc;
    }
    return n > 0 ? acc : 0;
}

/* forward.cu:141-206 map_gaussian_to_intersects (radius_clip overload); outputs
 * zero-initialised (bindings.cu:304-307).  radii (int) is compared with radius_clip
 * (float) (:161).  `capacity` guards the oracle against the out-of-bounds writes the
 * reference would perform for inconsistent inputs. */
void gi2d_oracle_map_gaussian_to_intersects(int n, int capacity, const float *xys,
                                            const float *depths, const int *radii,
                                            const int *cum_tiles_hit, int tiles_x, int tiles_y,
                                            float radius_clip, int64_t *isect_ids,
                                            int32_t *gaussian_ids) {
    memset(isect_ids, 0, sizeof(int64_t) * (size_t)capacity);
    memset(gaussian_ids, 0, sizeof(int32_t) * (size_t)capacity);
    for (int idx = 0; idx < n; ++idx) {
        if ((float)radii[idx] < radius_clip) continue;
        unsigned mnx, mny, mxx, mxy;
        tile_bbox(xys[2 * idx], xys[2 * idx + 1], (float)radii[idx], tiles_x, tiles_y, &mnx, &mny,
                  &mxx, &mxy);
        int32_t cur = idx == 0 ? 0 : cum_tiles_hit[idx - 1];
        int32_t bits;
        memcpy(&bits, &depths[idx], 4);
        int64_t depth_id = (int64_t)bits; /* sign-extended, as (int64_t)*(int32_t*)& */
        for (int i = (int)mny; i < (int)mxy; ++i)
            for (int j = (int)mnx; j < (int)mxx; ++j) {
                int64_t tile_id = (int64_t)(i * tiles_x + j);
                if (cur >= 0 && cur < capacity) {
                    isect_ids[cur] = (tile_id << 32) | depth_id;
                    gaussian_ids[cur] = idx;
                }
                ++cur;
            }
    }
}

/* gsplat/gsplat/utils.py:301-302 torch.sort(isect_ids) + gather.  torch.sort is not
 * declared stable; the oracle (and the HIP path) define the order as the STABLE sort
 * by the signed 64-bit key, i.e. ascending original position = ascending gaussian id
 * inside a tile when depths are equal.  LSD radix, 8 passes of 8 bits. */
void gi2d_oracle_sort_intersects(int m, const int64_t *isect_ids, const int32_t *gaussian_ids,
                                 int64_t *isect_sorted, int32_t *gaussian_sorted) {
    if (m <= 0) return;
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)m);
    uint64_t *kb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)m);
    int32_t *va = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    int32_t *vb = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    for (int i = 0; i < m; ++i) {
        ka[i] = (uint64_t)isect_ids[i] ^ 0x8000000000000000ull; /* signed order */
        va[i] = gaussian_ids[i];
    }
    for (int pass = 0; pass < 8; ++pass) {
        size_t cnt[257] = {0};
        int sh = pass * 8;
        for (int i = 0; i < m; ++i) cnt[((ka[i] >> sh) & 0xff) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (int i = 0; i < m; ++i) {
            size_t p = cnt[(ka[i] >> sh) & 0xff]++;
            kb[p] = ka[i];
            vb[p] = va[i];
        }
        uint64_t *tk = ka;
        ka = kb;
        kb = tk;
        int32_t *tv = va;
        va = vb;
        vb = tv;
    }
    for (int i = 0; i < m; ++i) {
        isect_sorted[i] = (int64_t)(ka[i] ^ 0x8000000000000000ull);
        gaussian_sorted[i] = va[i];
    }
    free(ka);
    free(kb);
    free(va);
    free(vb);
}

/* forward.cu:211-233 get_tile_bin_edges.  tile_bins has `rows` rows indexed by TILE ID
 * (the reference allocates num_intersects rows, bindings.cu:372-374, and writes out of
 * bounds when a tile id >= num_intersects occurs; the oracle drops such writes). */
void gi2d_oracle_get_tile_bin_edges(int m, const int64_t *isect_sorted, int rows,
                                    int32_t *tile_bins) {
    memset(tile_bins, 0, sizeof(int32_t) * 2 * (size_t)rows);
    for (int idx = 0; idx < m; ++idx) {
        int32_t cur = (int32_t)(isect_sorted[idx] >> 32);
        if (idx == 0 && cur >= 0 && cur < rows) tile_bins[2 * cur] = 0;
        if (idx == m - 1 && cur >= 0 && cur < rows) tile_bins[2 * cur + 1] = m;
        if (idx == 0) continue;
        int32_t prev = (int32_t)(isect_sorted[idx - 1] >> 32);
        if (prev != cur) {
            if (prev >= 0 && prev < rows) tile_bins[2 * prev + 1] = idx;
            if (cur >= 0 && cur < rows) tile_bins[2 * cur] = idx;
        }
    }
}

/* Relative half-width of the band around the alpha = 1/255 cut-off inside which a
 * (pixel, gaussian) pair is reported as "ambiguous": device exp (v_exp_f32, ~1 ulp,
 * FMA-contracted sigma) and host expf may land on different sides of the threshold
 * there (SURVEY section 7 "Threshold flips"). */
#define GI2D_AMBIG_REL 2e-5f
#define GI2D_AMBIG_SIGMA 1e-6f

/* Conditioning weight of a pair for the tolerance scales reported next to the results:
 * T = |a dx^2|/2 + |c dy^2|/2 + |b dx dy| is the magnitude of the terms sigma is summed
 * from; every fp32 evaluation order of sigma (this file, nvcc's FMA-contracted code, the
 * HIP kernels) carries a rounding error of a few ulp(T), i.e. a relative error of a few
 * 6e-8*T in alpha.  The scale is |term| * max(1, T/16): "1e-5 relative" where the
 * quadratic form is well conditioned, proportionally wider where it is not. */
static inline float pair_weight(float a, float b, float c, float dx, float dy) {
    float T = 0.5f * fabsf(a * dx * dx) + 0.5f * fabsf(c * dy * dy) + fabsf(b * dx * dy);
    return fmaxf(1.f, T / 16.f);
}

static inline int pair_eval(float a, float b, float c, float gx, float gy, float opac, float px,
                            float py, float *dx_, float *dy_, float *vis_, float *alpha_,
                            int *ambig) {
    /* forward.cu:534-541 == backward.cu:917-926 */
    float dx = gx - px, dy = gy - py;
    float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
    float vis = expf(-sigma);
    float alpha = fminf(1.f, opac * vis);
    *dx_ = dx;
    *dy_ = dy;
    *vis_ = vis;
    *alpha_ = alpha;
    if (ambig) {
        float thr = 1.f / 255.f;
        /* d alpha / alpha = d sigma, and every fp32 evaluation order of sigma is off by a few ulp of the terms it is
         * summed from, T = |a dx^2|/2 + |c dy^2|/2 + |b dx dy| (pair_weight above): the band is the wider of the fixed
         * relative one and 8 ulp(T) -- for a gaussian 200 times longer than wide T reaches 10^3..10^4 next to the
         * cut-off (sigma = 5.5 there is what is left of terms that large), where two correct machines differ by more
         * than the fixed band. */
        float T = 0.5f * fabsf(a * dx * dx) + 0.5f * fabsf(c * dy * dy) + fabsf(b * dx * dy);
        float band = fmaxf(GI2D_AMBIG_REL * (1.f + fabsf(sigma)), 8.f * 1.1920929e-7f * T);
        if (fabsf(alpha - thr) <= band * thr) *ambig = 1;
        if (fabsf(sigma) <= GI2D_AMBIG_SIGMA * (fabsf(a * dx * dx) + fabsf(c * dy * dy) +
                                                 fabsf(b * dx * dy)) &&
            sigma != 0.f)
            *ambig = 1;
    }
    return !(sigma < 0.f || alpha < 1.f / 255.f);
}

/* forward.cu:452-567 rasterize_forward_sum == :570-691 rasterize_sum_plus_forward.
 * One 16x16 block per tile; `done = true` after batch 0 (:553) means only the first
 * BLOCK_SIZE=256 list entries of a tile are ever consumed.  Pixel sample point is the
 * integer coordinate (:477-478).  final_Ts is always 1 (:497,:558); final_idx is the
 * absolute sorted-list index of the last contributor, 0 if none (:550,:559).
 * Optional (may be NULL): `ambig` u8[H*W] flags pixels touched by a near-threshold
 * pair; `abs_img` f32[H*W*3] receives sum |colour*alpha|*pair_weight for tolerance scaling. */
void gi2d_oracle_rasterize_forward_sum(int tiles_x, int tiles_y, int img_w, int img_h,
                                       const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                                       int tile_bins_rows, const float *xys, const float *conics,
                                       const float *colors, const float *opacities, float *final_Ts,
                                       int32_t *final_idx, float *out_img, uint8_t *ambig,
                                       float *abs_img) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < tiles_x * tiles_y; ++tile) {
        int ty = tile / tiles_x, tx = tile % tiles_x;
        int rx = 0, ry = 0;
        if (tile < tile_bins_rows) {
            rx = tile_bins[2 * tile];
            ry = tile_bins[2 * tile + 1];
        }
        int end = ry;
        if (end - rx > GI2D_BLOCK_SIZE) end = rx + GI2D_BLOCK_SIZE; /* the 256 cap */
        for (int ly = 0; ly < GI2D_BLOCK_Y; ++ly)
            for (int lx = 0; lx < GI2D_BLOCK_X; ++lx) {
                int i = ty * GI2D_BLOCK_Y + ly, j = tx * GI2D_BLOCK_X + lx;
                if (i >= img_h || j >= img_w) continue;
                float px = (float)j, py = (float)i;
                float o0 = 0.f, o1 = 0.f, o2 = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
                int cur_idx = 0, amb = 0;
                for (int idx = rx; idx < end; ++idx) {
                    int g = gaussian_ids_sorted[idx];
                    float dx, dy, vis, alpha;
                    int ok = pair_eval(conics[3 * g], conics[3 * g + 1], conics[3 * g + 2],
                                       xys[2 * g], xys[2 * g + 1], opacities[g], px, py, &dx, &dy,
                                       &vis, &alpha, ambig ? &amb : NULL);
                    if (!ok) continue;
                    o0 = o0 + colors[3 * g] * alpha;
                    o1 = o1 + colors[3 * g + 1] * alpha;
                    o2 = o2 + colors[3 * g + 2] * alpha;
                    if (abs_img) {
                        float wgt = pair_weight(conics[3 * g], conics[3 * g + 1], conics[3 * g + 2], dx, dy);
                        a0 += fabsf(colors[3 * g] * alpha) * wgt;
                        a1 += fabsf(colors[3 * g + 1] * alpha) * wgt;
                        a2 += fabsf(colors[3 * g + 2] * alpha) * wgt;
                    }
                    cur_idx = idx;
                }
                int pix = i * img_w + j;
                final_Ts[pix] = 1.f;
                final_idx[pix] = cur_idx;
                out_img[3 * pix] = o0;
                out_img[3 * pix + 1] = o1;
                out_img[3 * pix + 2] = o2;
                if (ambig) ambig[pix] = (uint8_t)amb;
                if (abs_img) {
                    abs_img[3 * pix] = a0;
                    abs_img[3 * pix + 1] = a1;
                    abs_img[3 * pix + 2] = a2;
                }
            }
    }
}

/* backward.cu:813-991 rasterize_backward_sum_kernel == :1168-1350.
 * A (pixel, list entry idx) pair contributes iff the pixel is inside, idx <= final_idx
 * of that pixel (:903), sigma >= 0 and alpha >= 1/255 (:925).  v_sigma ignores the
 * min(1,.) clamp (:948).  Outputs are zero-initialised (bindings.cu:1212-1216) and
 * accumulated here in double (the reference uses order-unspecified float atomics).
 * Optional (may be NULL): `ambig` u8[N] flags gaussians touched by a near-threshold
 * pair; `abs9` f32[N*9] = tolerance scales: sum of |term|*pair_weight for (v_xy[2],
 * v_conic[3], v_rgb[3], v_opacity), with the two v_xy terms taken part by part
 * (|v_sigma a dx| + |v_sigma b dy|, ...);
 * `v_abs_xy` f32[N*4] = (sum v_x, sum v_y, sum |v_x|, sum |v_y|) over pixels, the quantity
 * rasterize_sum.py:308,328 returns for `screenspace_points` (backward.cu:932,959-960).
 * `amb9` f32[N*9] (optional): sum of |term| over the FLAGGED pairs of the gaussian alone, landed or not -- what a
 * correct machine that puts those pairs on the other side of a cut-off may differ by at most (the bound the tests hold
 * the masked gaussians to). */
void gi2d_oracle_rasterize_backward_sum_ex(int n, int tiles_x, int tiles_y, int img_w, int img_h,
                                           const int32_t *gaussian_ids_sorted,
                                           const int32_t *tile_bins, int tile_bins_rows,
                                           const float *xys, const float *conics, const float *rgbs,
                                           const float *opacities, const int32_t *final_idx,
                                           const float *v_output, float *v_xy, float *v_conic,
                                           float *v_rgb, float *v_opacity, uint8_t *ambig,
                                           float *abs9, float *v_abs_xy, float *amb9) {
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    size_t stride = (size_t)n * 9;
    double *acc = (double *)calloc(stride * (size_t)nthreads, sizeof(double));
    double *aacc = v_abs_xy ? (double *)calloc(stride * (size_t)nthreads, sizeof(double)) : NULL;
    double *wacc = abs9 ? (double *)calloc(stride * (size_t)nthreads, sizeof(double)) : NULL;
    uint8_t *amb_t = ambig ? (uint8_t *)calloc((size_t)n * (size_t)nthreads, 1) : NULL;
    double *bacc = (amb9 && ambig) ? (double *)calloc(stride * (size_t)nthreads, sizeof(double)) : NULL;
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *A = acc + stride * (size_t)tid;
        double *AA = aacc ? aacc + stride * (size_t)tid : NULL;
        double *AW = wacc ? wacc + stride * (size_t)tid : NULL;
        uint8_t *AM = amb_t ? amb_t + (size_t)n * (size_t)tid : NULL;
        double *AB = bacc ? bacc + stride * (size_t)tid : NULL;
#pragma omp for schedule(dynamic, 4)
        for (int tile = 0; tile < tiles_x * tiles_y; ++tile) {
            int ty = tile / tiles_x, tx = tile % tiles_x;
            int rx = 0, ry = 0;
            if (tile < tile_bins_rows) {
                rx = tile_bins[2 * tile];
                ry = tile_bins[2 * tile + 1];
            }
            for (int idx = rx; idx < ry; ++idx) {
                int g = gaussian_ids_sorted[idx];
                float a = conics[3 * g], b = conics[3 * g + 1], c = conics[3 * g + 2];
                float gx = xys[2 * g], gy = xys[2 * g + 1], opac = opacities[g];
                float r0 = rgbs[3 * g], r1 = rgbs[3 * g + 1], r2 = rgbs[3 * g + 2];
                double s[9] = {0}, sa[9] = {0}, sw[9] = {0}, sb[9] = {0};
                int amb = 0, any = 0;
                for (int ly = 0; ly < GI2D_BLOCK_Y; ++ly)
                    for (int lx = 0; lx < GI2D_BLOCK_X; ++lx) {
                        int i = ty * GI2D_BLOCK_Y + ly, j = tx * GI2D_BLOCK_X + lx;
                        if (i >= img_h || j >= img_w) continue;
                        int pix = i * img_w + j;
                        float dx, dy, vis, alpha;
                        /* backward.cu:903: idx > final_idx -- not a contributor of this pixel in the forward.  It may be
                         * one by a hair: a pair whose alpha the forward put just BELOW 1/255 never extends final_idx, so
                         * the gate skips it before its alpha is looked at -- the ambiguity report must not depend on
                         * that (found at 2040x1356: one gaussian in 50 000 escaped the flag this way). */
                        const int gated = idx > final_idx[pix];
                        int pamb = 0;
                        if (gated && !AM) continue;
                        const int lands = pair_eval(a, b, c, gx, gy, opac, (float)j, (float)i, &dx, &dy, &vis, &alpha,
                                                    AM ? &pamb : NULL);
                        if (pamb) {
                            amb = 1;
                            if (AB) { /* what the flagged pair adds if it lands: the bound for the masked gaussian */
                                float vo0 = v_output[3 * pix], vo1 = v_output[3 * pix + 1], vo2 = v_output[3 * pix + 2];
                                double va = fabs((double)r0 * vo0 + (double)r1 * vo1 + (double)r2 * vo2);
                                double vs = fabs((double)opac * vis) * va, al = fabs((double)alpha);
                                sb[0] += vs * (fabs((double)a * dx) + fabs((double)b * dy));
                                sb[1] += vs * (fabs((double)b * dx) + fabs((double)c * dy));
                                sb[2] += 0.5 * vs * dx * dx;
                                sb[3] += 0.5 * vs * fabs((double)dx * dy);
                                sb[4] += 0.5 * vs * dy * dy;
                                sb[5] += al * fabs((double)vo0);
                                sb[6] += al * fabs((double)vo1);
                                sb[7] += al * fabs((double)vo2);
                                sb[8] += fabs((double)vis) * va;
                            }
                        }
                        if (gated || !lands) continue;
                        float vo0 = v_output[3 * pix], vo1 = v_output[3 * pix + 1],
                              vo2 = v_output[3 * pix + 2];
                        float t[9];
                        float v_alpha = 0.f;
                        v_alpha += r0 * vo0;
                        v_alpha += r1 * vo1;
                        v_alpha += r2 * vo2;
                        float v_sigma = -opac * vis * v_alpha;
                        t[0] = v_sigma * (a * dx + b * dy);
                        t[1] = v_sigma * (b * dx + c * dy);
                        t[2] = 0.5f * v_sigma * dx * dx;
                        t[3] = 0.5f * v_sigma * dx * dy;
                        t[4] = 0.5f * v_sigma * dy * dy;
                        t[5] = alpha * vo0;
                        t[6] = alpha * vo1;
                        t[7] = alpha * vo2;
                        t[8] = vis * v_alpha;
                        for (int k = 0; k < 9; ++k) {
                            s[k] += (double)t[k];
                            sa[k] += fabs((double)t[k]);
                        }
                        if (AW) {
                            double wgt = (double)pair_weight(a, b, c, dx, dy);
                            sw[0] += wgt * (fabs((double)(v_sigma * a * dx)) + fabs((double)(v_sigma * b * dy)));
                            sw[1] += wgt * (fabs((double)(v_sigma * b * dx)) + fabs((double)(v_sigma * c * dy)));
                            for (int k = 2; k < 9; ++k) sw[k] += wgt * fabs((double)t[k]);
                        }
                        any = 1;
                    }
                if (any)
                    for (int k = 0; k < 9; ++k) {
                        A[(size_t)g * 9 + k] += s[k];
                        if (AA) AA[(size_t)g * 9 + k] += sa[k];
                        if (AW) AW[(size_t)g * 9 + k] += sw[k];
                    }
                if (AM && amb) AM[g] = 1;
                if (AB && amb)
                    for (int k = 0; k < 9; ++k) AB[(size_t)g * 9 + k] += sb[k];
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int g = 0; g < n; ++g) {
        double s[9] = {0}, sa[9] = {0}, sw[9] = {0};
        int amb = 0;
        for (int t = 0; t < nthreads; ++t) {
            for (int k = 0; k < 9; ++k) {
                s[k] += acc[stride * (size_t)t + (size_t)g * 9 + k];
                if (aacc) sa[k] += aacc[stride * (size_t)t + (size_t)g * 9 + k];
                if (wacc) sw[k] += wacc[stride * (size_t)t + (size_t)g * 9 + k];
            }
            if (amb_t) amb |= amb_t[(size_t)n * (size_t)t + g];
        }
        v_xy[2 * g] = (float)s[0];
        v_xy[2 * g + 1] = (float)s[1];
        v_conic[3 * g] = (float)s[2];
        v_conic[3 * g + 1] = (float)s[3];
        v_conic[3 * g + 2] = (float)s[4];
        v_rgb[3 * g] = (float)s[5];
        v_rgb[3 * g + 1] = (float)s[6];
        v_rgb[3 * g + 2] = (float)s[7];
        v_opacity[g] = (float)s[8];
        if (abs9)
            for (int k = 0; k < 9; ++k) abs9[(size_t)g * 9 + k] = (float)sw[k];
        if (v_abs_xy) {
            v_abs_xy[4 * g] = (float)s[0];
            v_abs_xy[4 * g + 1] = (float)s[1];
            v_abs_xy[4 * g + 2] = (float)sa[0];
            v_abs_xy[4 * g + 3] = (float)sa[1];
        }
        if (ambig) ambig[g] = (uint8_t)amb;
        if (bacc)
            for (int k = 0; k < 9; ++k) {
                double v = 0.0;
                for (int t = 0; t < nthreads; ++t) v += bacc[stride * (size_t)t + (size_t)g * 9 + k];
                amb9[(size_t)g * 9 + k] = (float)v;
            }
    }
    free(acc);
    free(aacc);
    free(wacc);
    free(amb_t);
    free(bacc);
}
void gi2d_oracle_rasterize_backward_sum(int n, int tiles_x, int tiles_y, int img_w, int img_h,
                                        const int32_t *gaussian_ids_sorted,
                                        const int32_t *tile_bins, int tile_bins_rows,
                                        const float *xys, const float *conics, const float *rgbs,
                                        const float *opacities, const int32_t *final_idx,
                                        const float *v_output, float *v_xy, float *v_conic,
                                        float *v_rgb, float *v_opacity, uint8_t *ambig,
                                        float *abs9, float *v_abs_xy) {
    gi2d_oracle_rasterize_backward_sum_ex(n, tiles_x, tiles_y, img_w, img_h, gaussian_ids_sorted, tile_bins,
                                          tile_bins_rows, xys, conics, rgbs, opacities, final_idx, v_output, v_xy,
                                          v_conic, v_rgb, v_opacity, ambig, abs9, v_abs_xy, NULL);
}

int gi2d_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void gi2d_oracle_set_num_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}
