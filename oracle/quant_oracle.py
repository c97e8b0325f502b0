"""CPU restatement (numpy) of the reference's quantisation-aware front end -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(gaussianimage_plus_amd/) never does.  Pinned by tests/golden/quant_reference.npz, which was produced by running the
reference's own classes under torch autograd (tests/golden/make_quant_golden.py).

Follows /root/reference/quantize.py:
  UniformQuantizer (LSQ+)  :39-156   forward :123-141, _init_data :69-77, compress/decompress :149-156
  LogQuantizer             :158-259  forward (learned=False branch) :219-233, _init_data :192-201, compress :243-255
  HybirdQuant              :336-389  channels 0 and 2 -> LogQuantizer, channel 1 -> UniformQuantizer
  FakeQuantizationHalf     :27-37
  ste() :23-24  (round half to even, gradient 1)
and the size arithmetic of models/gaussianimage_covariance.py:469-509 (analysis_wo_ec, lsq branches).

Forward values are computed in float32 operation by operation like the torch code; gradient sums are accumulated in
float64 (torch sums in float32 in an unspecified order, so tests compare sums with a condition-aware tolerance).

Gradients (derived from the autograd graph of the forward as written, with `ste` passing the gradient through):
  LSQ   code = clamp((x-b)/s, qmin, qmax), y = round(code)*s + b, m = [qmin <= (x-b)/s <= qmax]
        v_x = m * (g*s)/s;  v_s = sum g*round(code) - sum m*(g*s)*((x-b)/s)/s;  v_b = sum g - sum m*(g*s)/s
  Log   L = log(|x|+1e-6), b = min L, s = (max L - b)/(qmax-qmin) (one scalar range over ALL channels passed in),
        D = round(code)*s + b, y = exp(D), gD = g*y.  b and s stay attached to the graph, so
        v_s = sum gD*round(code) - sum m*(gD*s)*((L-b)/s)/s,  v_b = sum gD - sum m*(gD*s)/s - v_s/(qmax-qmin),
        v_max = v_s/(qmax-qmin); torch.min()/max() spread v_b / v_max evenly over all elements that attain the
        extreme; v_x = v_L * sign(x)/(|x|+1e-6) with torch.abs's sign(0)=0.
"""
import numpy as np

F = np.float32
LOG_EPS = F(1e-6)


def qrange(bits, signed=False):
    """(qmin, qmax) of quantize.py:48-59 / :168-176."""
    if signed:
        return float(-2 ** (bits - 1)), float(2 ** (bits - 1) - 1)
    return 0.0, float(2 ** bits - 1)


# ---------------------------------------------------------------------------------------------- LSQ (UniformQuantizer)
def lsq_init(x, qmin, qmax):
    """_init_data (quantize.py:69-77): per-channel range of the data."""
    x = np.asarray(x, F)
    t_min, t_max = x.min(axis=0), x.max(axis=0)
    scale = ((t_max - t_min) / F(qmax - qmin)).astype(F)
    beta = (t_min - F(qmin) * scale).astype(F)
    return scale, beta


def lsq_forward(x, scale, beta, qmin, qmax):
    """forward (quantize.py:123-141): returns (dequant, code) with code already rounded (the value `ste` yields)."""
    x, scale, beta = np.asarray(x, F), np.asarray(scale, F), np.asarray(beta, F)
    raw = ((x - beta) / scale).astype(F)
    code = np.rint(np.clip(raw, F(qmin), F(qmax))).astype(F)
    return (code * scale + beta).astype(F), code


def lsq_backward(x, scale, beta, qmin, qmax, g):
    x, scale, beta, g = np.asarray(x, F), np.asarray(scale, F), np.asarray(beta, F), np.asarray(g, F)
    raw = ((x - beta) / scale).astype(F)
    m = (raw >= F(qmin)) & (raw <= F(qmax))
    code = np.rint(np.clip(raw, F(qmin), F(qmax))).astype(F)
    gc = np.where(m, (g * scale).astype(F), F(0))
    v_x = (gc / scale).astype(F)
    v_scale = (g.astype(np.float64) * code).sum(0) - ((gc * raw).astype(F) / scale).astype(np.float64).sum(0)
    v_beta = g.astype(np.float64).sum(0) - v_x.astype(np.float64).sum(0)
    return v_x, v_scale, v_beta


def lsq_compress(x, scale, beta, qmin, qmax):
    """compress (quantize.py:149-152): (dequantised, integer codes)."""
    deq, code = lsq_forward(x, scale, beta, qmin, qmax)
    return deq, code


def lsq_decompress(code, scale, beta):
    return (np.asarray(code, F) * np.asarray(scale, F) + np.asarray(beta, F)).astype(F)


# ------------------------------------------------------------------------------------------------ Log (LogQuantizer)
def log_of(x):
    return np.log(np.abs(np.asarray(x, F)) + LOG_EPS).astype(F)


def log_forward(x, qmin, qmax):
    """forward, learned=False (quantize.py:219-233): ONE scalar range over the whole tensor, recomputed every call,
    and no sign in the result.  Returns (dequant, code, beta, scale)."""
    L = log_of(x)
    beta, mx = L.min(), L.max()
    scale = F((mx - beta) / F(qmax - qmin))
    raw = ((L - beta) / scale).astype(F)
    code = np.rint(np.clip(raw, F(qmin), F(qmax))).astype(F)
    return np.exp((code * scale + beta).astype(F)).astype(F), code, beta, scale


def log_backward(x, qmin, qmax, g):
    x, g = np.asarray(x, F), np.asarray(g, F)
    L = log_of(x)
    beta, mx = L.min(), L.max()
    qr = F(qmax - qmin)
    scale = F((mx - beta) / qr)
    raw = ((L - beta) / scale).astype(F)
    m = (raw >= F(qmin)) & (raw <= F(qmax))
    code = np.rint(np.clip(raw, F(qmin), F(qmax))).astype(F)
    y = np.exp((code * scale + beta).astype(F)).astype(F)
    gD = (g * y).astype(F)
    gc = np.where(m, (gD * scale).astype(F), F(0))
    gL = (gc / scale).astype(np.float64)
    v_s = (gD.astype(np.float64) * code).sum() - ((gc * raw).astype(F) / scale).astype(np.float64).sum()
    v_b = gD.astype(np.float64).sum() - gL.sum() - v_s / float(qr)
    v_mx = v_s / float(qr)
    at_min, at_max = L == beta, L == mx
    gL = gL + at_min * (v_b / at_min.sum()) + at_max * (v_mx / at_max.sum())
    return (gL * np.sign(x) / (np.abs(x) + LOG_EPS).astype(np.float64)).astype(F)


def log_init(x, qmin, qmax):
    """_init_data (quantize.py:192-201): per-CHANNEL log range, used by compress()/decompress()."""
    L = log_of(x)
    t_min, t_max = L.min(axis=0), L.max(axis=0)
    return ((t_max - t_min) / F(qmax - qmin)).astype(F), t_min.astype(F)


def log_compress(x, qmin, qmax):
    """compress (quantize.py:243-255): re-initialises per channel first (learned=False); magnitudes only."""
    scale, beta = log_init(x, qmin, qmax)
    raw = ((log_of(x) - beta) / scale).astype(F)
    code = np.rint(np.clip(raw, F(qmin), F(qmax))).astype(F)
    return np.exp((code * scale + beta).astype(F)).astype(F), code, scale, beta


def log_decompress(code, scale, beta):
    return np.exp((np.asarray(code, F) * np.asarray(scale, F) + np.asarray(beta, F)).astype(F)).astype(F)


# ------------------------------------------------------------------------------------------------ Hybrid (HybirdQuant)
def hybrid_forward(x, cov_scale, cov_beta, bits, cov_bits):
    """forward (quantize.py:354-366) on [N,3] rows (a, b, c): a and c share one log range, b is LSQ."""
    x = np.asarray(x, F)
    lq = qrange(bits)
    cq = qrange(cov_bits)
    dv, cv, lbeta, lscale = log_forward(x[:, ::2], *lq)
    dc, cc = lsq_forward(x[:, 1:2], cov_scale, cov_beta, *cq)
    return (np.concatenate([dv[:, :1], dc, dv[:, 1:]], 1), np.concatenate([cv[:, :1], cc, cv[:, 1:]], 1),
            lbeta, lscale)


def hybrid_backward(x, cov_scale, cov_beta, bits, cov_bits, g):
    x, g = np.asarray(x, F), np.asarray(g, F)
    gv = log_backward(x[:, ::2], *qrange(bits), g[:, ::2])
    gc, v_scale, v_beta = lsq_backward(x[:, 1:2], cov_scale, cov_beta, *qrange(cov_bits), g[:, 1:2])
    return np.concatenate([gv[:, :1], gc, gv[:, 1:]], 1), v_scale, v_beta


def hybrid_compress(x, cov_scale, cov_beta, bits, cov_bits):
    """compress (quantize.py:375-382). Returns (dequant, codes, var_scale[2], var_beta[2])."""
    x = np.asarray(x, F)
    dv, cv, vs, vb = log_compress(x[:, ::2], *qrange(bits))
    dc, cc = lsq_compress(x[:, 1:2], cov_scale, cov_beta, *qrange(cov_bits))
    return (np.concatenate([dv[:, :1], dc, dv[:, 1:]], 1), np.concatenate([cv[:, :1], cc, cv[:, 1:]], 1), vs, vb)


def hybrid_decompress(code, cov_scale, cov_beta, var_scale, var_beta):
    code = np.asarray(code, F)
    v = log_decompress(code[:, ::2], var_scale, var_beta)
    c = lsq_decompress(code[:, 1:2], cov_scale, cov_beta)
    return np.concatenate([v[:, :1], c, v[:, 1:]], 1)


def hybrid_size(bits, cov_bits):
    """HybirdQuant.size (quantize.py:368-369)."""
    return (cov_bits + bits * 2) / 3


# ------------------------------------------------------------------------------------------------------ half precision
def half_forward(x):
    with np.errstate(over="ignore"):  # beyond 65504 the half is inf, as in torch
        return np.asarray(x, F).astype(np.float16).astype(F)


def half_backward(g):
    return np.asarray(g, F)


# ------------------------------------------------------------------------------------------------------- size analysis
def analysis_bits(num_points, height, width, xy_quant="lsq", xy_bit=12, cov_bit=10, color_bit=6):
    """analysis_wo_ec, lsq branches (models/gaussianimage_covariance.py:469-509): fixed-length code sizes plus the
    quantiser side information (32*3*2 bits for covariance and colour, 32*2*2 for positions)."""
    chol_bits = num_points * 3 * hybrid_size(cov_bit, cov_bit) + 32 * 3 * 2
    feat_bits = num_points * 3 * color_bit + 32 * 3 * 2
    pos_bits = num_points * 2 * xy_bit + 32 * 2 * 2 if xy_quant == "lsq" else num_points * 2 * 16
    hw = height * width
    return {"bpp": (pos_bits + chol_bits + feat_bits) / hw, "position_bpp": pos_bits / hw,
            "cholesky_bpp": chol_bits / hw, "feature_dc_bpp": feat_bits / hw}


def gaussian_code_length_bits(codes):
    """Ideal code length of integer symbols under the quantised Gaussian the reference hands to its ANS coder
    (utils.py:94-110: mean, std clamped to [1e-5, 1e10], support [min, max] of the data).  The coder itself
    (`constriction`, third-party, absent here) adds at most a few 32-bit words on top; this is the estimate used in its
    place and is not a bit-exact restatement of that library."""
    from math import erf, sqrt
    c = np.asarray(codes, np.float64).ravel()
    mean, std = c.mean(), min(max(c.std(ddof=1), 1e-5), 1e10)
    lo, hi = int(c.min()), int(c.max())
    if lo == hi:
        hi = lo + 1
    ks = np.arange(lo, hi + 1)
    cdf = np.array([0.5 * (1 + erf((k - mean) / (std * sqrt(2)))) for k in np.concatenate([ks - 0.5, [hi + 0.5]])])
    p = np.diff(cdf)
    p = p / p.sum()
    p = (p * (1 - len(p) * 2.0 ** -24) + 2.0 ** -24)  # every symbol keeps a non-zero ("leaky") probability
    idx = c.astype(np.int64) - lo
    return float(-np.log2(p[idx]).sum())
