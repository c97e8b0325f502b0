// Element-level arithmetic of the quantisation-aware front end, shared by the stand-alone
// quantiser kernels (gi2d_quant.hip) and the fused quantised fitting iteration (gi2d_train.hip) so both produce
// the same bits.  Restates /root/reference/quantize.py:
//   UniformQuantizer.forward :123-141 (LSQ+), LogQuantizer.forward learned=False :219-233, ste :23-24.
// Operation order is torch's (sub, div, clamp, round-half-even, mul, add -- no fused multiply-add), so LSQ codes
// and dequantised values are bit-exact with the reference; the log quantiser goes through logf/expf (<= 1 ulp).
#pragma once
#include "gi2d_common.h"

#define GI2D_QUANT_LOG_EPS 1e-6f

namespace gi2d {

__device__ __forceinline__ float quant_log_of(float x) { return logf(fabsf(x) + GI2D_QUANT_LOG_EPS); }

struct QuantEval {
    float raw;      // (t - beta) / scale before the clamp, t = x (LSQ) or log(|x| + 1e-6) (log)
    float code;     // round(clamp(raw))  -- the value ste() yields
    float dequant;  // code * scale + beta (LSQ) or exp of that (log)
    bool inside;    // clamp passes the gradient: qmin <= raw <= qmax
};

template <int KIND>
__device__ __forceinline__ QuantEval quant_eval(float x, float scale, float beta, float qmin, float qmax) {
#pragma clang fp contract(off)
    QuantEval e;
    const float t = KIND == GI2D_QUANT_LOG ? quant_log_of(x) : x;
    e.raw = (t - beta) / scale;
    e.inside = e.raw >= qmin && e.raw <= qmax;
    e.code = rintf(fminf(fmaxf(e.raw, qmin), qmax));
    const float lin = e.code * scale + beta;
    e.dequant = KIND == GI2D_QUANT_LOG ? expf(lin) : lin;
    return e;
}

// Gradient of one element.  Returns d loss / d t (t as above) WITHOUT the range terms of the log quantiser, and adds
// this element's share of d loss / d scale and d loss / d beta (direct paths only) to sum_s / sum_b:
//   gc = inside ? gq * scale : 0;  v_t = gc / scale;  sum_s += gq * code - (gc * raw) / scale;  sum_b += gq - v_t
// with gq = g (LSQ) or g * dequant (log: through exp).
template <int KIND>
__device__ __forceinline__ float quant_grad(const QuantEval &e, float g, float scale, float &sum_s, float &sum_b) {
#pragma clang fp contract(off)
    const float gq = KIND == GI2D_QUANT_LOG ? g * e.dequant : g;
    const float gc = e.inside ? gq * scale : 0.f;
    const float vt = gc / scale;
    sum_s += gq * e.code - (gc * e.raw) / scale;
    sum_b += gq - vt;
    return vt;
}

// d t / d x of the log quantiser: torch.abs' sign (0 at 0) over |x| + 1e-6
__device__ __forceinline__ float quant_log_chain(float x) {
#pragma clang fp contract(off)
    const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    return sg / (fabsf(x) + GI2D_QUANT_LOG_EPS);
}

// Range of the log quantiser from the extremes of log(|x| + 1e-6): beta = min, scale = (max - min) / (qmax - qmin)
__device__ __forceinline__ float quant_log_scale(float lmin, float lmax, float qmin, float qmax) {
#pragma clang fp contract(off)
    return (lmax - lmin) / (qmax - qmin);
}

// Wave (64 lanes) and workgroup tree reductions used for the quantiser sums; `red` holds one float per wave.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

}  // namespace gi2d
