// Stand-alone quantiser operators of the quantisation-aware front end: what
// /root/reference/quantize.py runs as a few dozen elementwise torch kernels plus min()/max() reductions per
// quantiser and per iteration.  Rows of up to four channels, one thread per row; every channel is either an LSQ+
// UniformQuantizer (learned scale/beta) or a LogQuantizer(learned=False) channel.  All log channels of a spec share
// ONE range in forward/backward (LogQuantizer.forward takes min()/max() over its whole input; HybirdQuant hands it
// the two variance columns together, quantize.py:354-366) and per-channel ranges in init/compress (:192-201,243-255).
//
// Reductions are two-stage and order-fixed (per-workgroup partials, then one workgroup sums them in double), so
// results are bitwise reproducible.  The fused fitting iteration (gi2d_train.hip) uses the same element functions
// (gi2d_quant_core.h) inside its own kernels; these entry points serve the torch-facing quantiser modules, the
// compress / decompress step and the tests.
#include "gi2d_quant_core.h"

namespace gi2d {

struct QuantSpecDev {
    int channels;
    int kind[4];
    float qmin[4], qmax[4];
};

#define GI2D_QROW 16  // floats per workgroup partial row: [channel][4]

// ---- ranges ---------------------------------------------------------------------------------------------------
// per-workgroup (min, max) of t per channel: t = log(|x|+1e-6) on log channels, x on LSQ channels
__global__ __launch_bounds__(256) void quant_range_kernel(QuantSpecDev sp, int n, const float *__restrict__ x,
                                                          float *__restrict__ partial) {
    __shared__ float red[4][8];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = 0; c < sp.channels; ++c) {
        float lo = INFINITY, hi = -INFINITY;
        if (r < n) {
            const float v = x[(size_t)r * sp.channels + c];
            lo = hi = sp.kind[c] == GI2D_QUANT_LOG ? quant_log_of(v) : v;
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        if (lane == 0) {
            red[wave][2 * c] = lo;
            red[wave][2 * c + 1] = hi;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * sp.channels) {
        const int k = threadIdx.x;
        float v = red[0][k];
        for (int w = 1; w < 4; ++w) v = (k & 1) ? fmaxf(v, red[w][k]) : fminf(v, red[w][k]);
        partial[(size_t)blockIdx.x * GI2D_QROW + k] = v;
    }
}

// one workgroup: combine the partial ranges into params[c] = {scale, beta, max t, 0}.
// SHARED_LOG: all log channels get one common range and LSQ channels are left untouched (training forward);
// otherwise every channel gets its own (the reference's _init_data).
template <bool SHARED_LOG>
__global__ __launch_bounds__(256) void quant_range_finish_kernel(QuantSpecDev sp, int blocks,
                                                                 const float *__restrict__ partial,
                                                                 float *__restrict__ params) {
#pragma clang fp contract(off)
    __shared__ float red[4][8];
    __shared__ float ext[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < 2 * sp.channels; ++k) {
        float v = (k & 1) ? -INFINITY : INFINITY;
        for (int b = threadIdx.x; b < blocks; b += 256) {
            const float p = partial[(size_t)b * GI2D_QROW + k];
            v = (k & 1) ? fmaxf(v, p) : fminf(v, p);
        }
        v = (k & 1) ? wave_max(v) : wave_min(v);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2 * sp.channels) {
        const int k = threadIdx.x;
        float v = red[0][k];
        for (int w = 1; w < 4; ++w) v = (k & 1) ? fmaxf(v, red[w][k]) : fminf(v, red[w][k]);
        ext[k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (SHARED_LOG) {
            float lo = INFINITY, hi = -INFINITY;
            for (int c = 0; c < sp.channels; ++c)
                if (sp.kind[c] == GI2D_QUANT_LOG) {
                    lo = fminf(lo, ext[2 * c]);
                    hi = fmaxf(hi, ext[2 * c + 1]);
                }
            for (int c = 0; c < sp.channels; ++c)
                if (sp.kind[c] == GI2D_QUANT_LOG) {
                    params[4 * c] = quant_log_scale(lo, hi, sp.qmin[c], sp.qmax[c]);
                    params[4 * c + 1] = lo;
                    params[4 * c + 2] = hi;
                    params[4 * c + 3] = 0.f;
                }
        } else {
            for (int c = 0; c < sp.channels; ++c) {
                const float lo = ext[2 * c], hi = ext[2 * c + 1];
                const float scale = (hi - lo) / (sp.qmax[c] - sp.qmin[c]);
                params[4 * c] = scale;
                // quantize.py:74 beta = t_min - qmin*scale (LSQ);  :198 beta = t_min (log)
                params[4 * c + 1] = sp.kind[c] == GI2D_QUANT_LOG ? lo : lo - sp.qmin[c] * scale;
                params[4 * c + 2] = hi;
                params[4 * c + 3] = 0.f;
            }
        }
    }
}

// ---- forward / compress ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_apply_kernel(QuantSpecDev sp, int n, const float *__restrict__ x,
                                                          const float *__restrict__ params,
                                                          float *__restrict__ dequant, float *__restrict__ code) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    for (int c = 0; c < sp.channels; ++c) {
        const size_t i = (size_t)r * sp.channels + c;
        const float s = params[4 * c], b = params[4 * c + 1];
        const QuantEval e = sp.kind[c] == GI2D_QUANT_LOG
                                ? quant_eval<GI2D_QUANT_LOG>(x[i], s, b, sp.qmin[c], sp.qmax[c])
                                : quant_eval<GI2D_QUANT_LSQ>(x[i], s, b, sp.qmin[c], sp.qmax[c]);
        if (dequant) dequant[i] = e.dequant;
        if (code) code[i] = e.code;
    }
}

__global__ __launch_bounds__(256) void quant_decompress_kernel(QuantSpecDev sp, int n,
                                                               const float *__restrict__ code,
                                                               const float *__restrict__ params,
                                                               float *__restrict__ out) {
#pragma clang fp contract(off)
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    for (int c = 0; c < sp.channels; ++c) {
        const size_t i = (size_t)r * sp.channels + c;
        const float lin = code[i] * params[4 * c] + params[4 * c + 1];
        out[i] = sp.kind[c] == GI2D_QUANT_LOG ? expf(lin) : lin;
    }
}

__global__ __launch_bounds__(256) void quant_half_kernel(size_t count, const float *__restrict__ x,
                                                         float *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = (float)(_Float16)x[i];  // x.half().float(): round to nearest even, overflow to inf
}

// ---- backward -------------------------------------------------------------------------------------------------
// phase 1: elementwise gradient (without the range terms of the log channels) + per-workgroup partial sums
// [c][4] = {sum_s, sum_b, #elements at the range minimum, #at the maximum}
__global__ __launch_bounds__(256) void quant_bwd_kernel(QuantSpecDev sp, int n, const float *__restrict__ x,
                                                        const float *__restrict__ params,
                                                        const float *__restrict__ v_dequant,
                                                        float *__restrict__ v_x, float *__restrict__ partial) {
    __shared__ float red[4][GI2D_QROW];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = 0; c < sp.channels; ++c) {
        float ss = 0.f, sb = 0.f, cmin = 0.f, cmax = 0.f;
        if (r < n) {
            const size_t i = (size_t)r * sp.channels + c;
            const float s = params[4 * c], b = params[4 * c + 1], xv = x[i], g = v_dequant[i];
            if (sp.kind[c] == GI2D_QUANT_LOG) {
                const QuantEval e = quant_eval<GI2D_QUANT_LOG>(xv, s, b, sp.qmin[c], sp.qmax[c]);
                const float vt = quant_grad<GI2D_QUANT_LOG>(e, g, s, ss, sb);
                v_x[i] = vt * quant_log_chain(xv);
                const float t = quant_log_of(xv);
                cmin = t == b ? 1.f : 0.f;
                cmax = t == params[4 * c + 2] ? 1.f : 0.f;
            } else {
                const QuantEval e = quant_eval<GI2D_QUANT_LSQ>(xv, s, b, sp.qmin[c], sp.qmax[c]);
                v_x[i] = quant_grad<GI2D_QUANT_LSQ>(e, g, s, ss, sb);
            }
        }
        ss = wave_sum(ss);
        sb = wave_sum(sb);
        cmin = wave_sum(cmin);
        cmax = wave_sum(cmax);
        if (lane == 0) {
            red[wave][4 * c] = ss;
            red[wave][4 * c + 1] = sb;
            red[wave][4 * c + 2] = cmin;
            red[wave][4 * c + 3] = cmax;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4 * sp.channels) {
        const int k = threadIdx.x;
        partial[(size_t)blockIdx.x * GI2D_QROW + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

// phase 2 (one workgroup): totals in double, fixed order.  v_params[c] = {v_scale, v_beta} for LSQ channels (0 for
// log channels, whose range is not a parameter); extras = {v_beta_total / #min, v_max / #max} for the log channels.
__global__ __launch_bounds__(256) void quant_bwd_finish_kernel(QuantSpecDev sp, int blocks,
                                                               const float *__restrict__ partial,
                                                               float *__restrict__ v_params,
                                                               float *__restrict__ extras) {
    __shared__ double red[4][GI2D_QROW];
    __shared__ double tot[GI2D_QROW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < 4 * sp.channels; ++k) {
        double v = 0.0;
        for (int b = threadIdx.x; b < blocks; b += 256) v += (double)partial[(size_t)b * GI2D_QROW + k];
        v = wave_sum_d(v);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4 * sp.channels) {
        const int k = threadIdx.x;
        tot[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double vs = 0.0, vb = 0.0, nmin = 0.0, nmax = 0.0, qr = 1.0;
        for (int c = 0; c < sp.channels; ++c) {
            if (sp.kind[c] == GI2D_QUANT_LOG) {
                vs += tot[4 * c];
                vb += tot[4 * c + 1];
                nmin += tot[4 * c + 2];
                nmax += tot[4 * c + 3];
                qr = (double)(sp.qmax[c] - sp.qmin[c]);
                v_params[2 * c] = 0.f;
                v_params[2 * c + 1] = 0.f;
            } else {
                v_params[2 * c] = (float)tot[4 * c];
                v_params[2 * c + 1] = (float)tot[4 * c + 1];
            }
        }
        // scale = (max - beta) / (qmax - qmin) is part of the graph: beta also receives -v_scale/qr, max +v_scale/qr;
        // torch.min()/max() spread their gradient evenly over every element that attains the extreme
        extras[0] = nmin > 0.0 ? (float)((vb - vs / qr) / nmin) : 0.f;
        extras[1] = nmax > 0.0 ? (float)((vs / qr) / nmax) : 0.f;
    }
}

// phase 3: the elements at the extremes of the shared log range receive the range gradient
__global__ __launch_bounds__(256) void quant_bwd_ties_kernel(QuantSpecDev sp, int n, const float *__restrict__ x,
                                                             const float *__restrict__ params,
                                                             const float *__restrict__ extras,
                                                             float *__restrict__ v_x) {
#pragma clang fp contract(off)
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    for (int c = 0; c < sp.channels; ++c) {
        if (sp.kind[c] != GI2D_QUANT_LOG) continue;
        const size_t i = (size_t)r * sp.channels + c;
        const float xv = x[i], t = quant_log_of(xv);
        float add = 0.f;
        if (t == params[4 * c + 1]) add += extras[0];
        if (t == params[4 * c + 2]) add += extras[1];
        if (add != 0.f) v_x[i] = v_x[i] + add * quant_log_chain(xv);
    }
}

static int spec_to_dev(const gi2d_quant_spec *spec, QuantSpecDev &d, bool &any_log) {
    if (!spec || spec->channels < 1 || spec->channels > GI2D_QUANT_MAX_CHANNELS) {
        set_error("quant: spec missing or channels outside 1..4");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    d.channels = spec->channels;
    any_log = false;
    for (int c = 0; c < 4; ++c) {
        d.kind[c] = c < spec->channels ? spec->kind[c] : 0;
        d.qmin[c] = spec->qmin[c];
        d.qmax[c] = spec->qmax[c];
        if (c < spec->channels) {
            if (d.kind[c] != GI2D_QUANT_LSQ && d.kind[c] != GI2D_QUANT_LOG) {
                set_error("quant: unknown channel kind");
                return GI2D_ERR_INVALID_ARGUMENT;
            }
            if (!(d.qmax[c] > d.qmin[c])) {
                set_error("quant: qmax must exceed qmin");
                return GI2D_ERR_INVALID_ARGUMENT;
            }
            any_log |= d.kind[c] == GI2D_QUANT_LOG;
        }
    }
    return GI2D_OK;
}

}  // namespace gi2d

using namespace gi2d;

extern "C" {

size_t gi2d_quant_workspace_bytes(int num_rows) {
    const size_t blocks = num_rows > 0 ? ((size_t)num_rows + 255) / 256 : 1;
    return (blocks * GI2D_QROW + 16) * sizeof(float);
}

static int quant_common(const gi2d_quant_spec *spec, int n, const void *a, const void *b, void *ws, size_t ws_bytes,
                        bool need_ws, QuantSpecDev &sp, bool &any_log) {
    const int rc = spec_to_dev(spec, sp, any_log);
    if (rc != GI2D_OK) return rc;
    if (n < 0 || (n > 0 && (!a || !b))) {
        set_error("quant: bad size or null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (need_ws && n > 0 && (!ws || ws_bytes < gi2d_quant_workspace_bytes(n))) {
        set_error("quant: workspace too small");
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    return GI2D_OK;
}

int gi2d_quant_init(const gi2d_quant_spec *spec, int n, const float *x, float *params, void *ws, size_t ws_bytes,
                    gi2d_stream_t st_) {
    QuantSpecDev sp;
    bool any_log;
    const int rc = quant_common(spec, n, x, params, ws, ws_bytes, true, sp, any_log);
    if (rc != GI2D_OK || n == 0) return rc;
    hipStream_t st = (hipStream_t)st_;
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(quant_range_kernel, dim3(blocks), dim3(256), 0, st, sp, n, x, (float *)ws);
    hipLaunchKernelGGL(quant_range_finish_kernel<false>, dim3(1), dim3(256), 0, st, sp, blocks, (const float *)ws,
                       params);
    return check_launch("quant init");
}

int gi2d_quant_forward(const gi2d_quant_spec *spec, int n, const float *x, float *params, float *dequant,
                       float *code, void *ws, size_t ws_bytes, gi2d_stream_t st_) {
    QuantSpecDev sp;
    bool any_log;
    const int rc = quant_common(spec, n, x, params, ws, ws_bytes, true, sp, any_log);
    if (rc != GI2D_OK || n == 0) return rc;
    hipStream_t st = (hipStream_t)st_;
    const int blocks = (n + 255) / 256;
    if (any_log) {
        hipLaunchKernelGGL(quant_range_kernel, dim3(blocks), dim3(256), 0, st, sp, n, x, (float *)ws);
        hipLaunchKernelGGL(quant_range_finish_kernel<true>, dim3(1), dim3(256), 0, st, sp, blocks,
                           (const float *)ws, params);
    }
    hipLaunchKernelGGL(quant_apply_kernel, dim3(blocks), dim3(256), 0, st, sp, n, x, (const float *)params, dequant,
                       code);
    return check_launch("quant forward");
}

int gi2d_quant_backward(const gi2d_quant_spec *spec, int n, const float *x, const float *params,
                        const float *v_dequant, float *v_x, float *v_params, void *ws, size_t ws_bytes,
                        gi2d_stream_t st_) {
    QuantSpecDev sp;
    bool any_log;
    const int rc = quant_common(spec, n, x, params, ws, ws_bytes, true, sp, any_log);
    if (rc != GI2D_OK) return rc;
    if (!v_params || (n > 0 && (!v_dequant || !v_x))) {
        set_error("quant backward: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = (hipStream_t)st_;
    const int blocks = (n + 255) / 256;  // n == 0: the finish kernel still zeroes v_params
    float *partial = (float *)ws, *extras = partial ? partial + (size_t)(blocks > 0 ? blocks : 1) * GI2D_QROW : nullptr;
    if (n == 0) {
        (void)hipMemsetAsync(v_params, 0, sizeof(float) * 2 * sp.channels, st);
        return check_launch("quant backward");
    }
    hipLaunchKernelGGL(quant_bwd_kernel, dim3(blocks), dim3(256), 0, st, sp, n, x, params, v_dequant, v_x, partial);
    hipLaunchKernelGGL(quant_bwd_finish_kernel, dim3(1), dim3(256), 0, st, sp, blocks, (const float *)partial,
                       v_params, extras);
    if (any_log)
        hipLaunchKernelGGL(quant_bwd_ties_kernel, dim3(blocks), dim3(256), 0, st, sp, n, x, params,
                           (const float *)extras, v_x);
    return check_launch("quant backward");
}

int gi2d_quant_compress(const gi2d_quant_spec *spec, int n, const float *x, const float *params, float *dequant,
                        float *code, gi2d_stream_t st_) {
    QuantSpecDev sp;
    bool any_log;
    const int rc = quant_common(spec, n, x, params, nullptr, 0, false, sp, any_log);
    if (rc != GI2D_OK || n == 0) return rc;
    hipLaunchKernelGGL(quant_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st_, sp, n, x, params,
                       dequant, code);
    return check_launch("quant compress");
}

int gi2d_quant_decompress(const gi2d_quant_spec *spec, int n, const float *code, const float *params, float *out,
                          gi2d_stream_t st_) {
    QuantSpecDev sp;
    bool any_log;
    const int rc = quant_common(spec, n, code, params, nullptr, 0, false, sp, any_log);
    if (rc != GI2D_OK || n == 0) return rc;
    if (!out) {
        set_error("quant decompress: null output");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(quant_decompress_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st_, sp, n, code,
                       params, out);
    return check_launch("quant decompress");
}

int gi2d_quant_half(size_t count, const float *x, float *y, gi2d_stream_t st_) {
    if (count > 0 && (!x || !y)) {
        set_error("quant half: null pointer");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (count == 0) return GI2D_OK;
    hipLaunchKernelGGL(quant_half_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)st_,
                       count, x, y);
    return check_launch("quant half");
}

}  // extern "C"
