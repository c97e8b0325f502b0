"""Quantisers of the quantisation-aware front end, same class names and call surface as the reference's
quantize.py, backed by the HIP quantiser kernels (csrc/gi2d_quant.hip) through the C ABI (include/gi2d.h,
gi2d_quant_*).  SURVEY 8f rank 4.

    UniformQuantizer      LSQ+ (quantize.py:39-156)
    LogQuantizer          log-domain quantiser, learned=False (quantize.py:158-259)
    HybirdQuant           [N,3] covariance rows: log quantiser on the variances, LSQ on the covariance (:336-389)
    FakeQuantizationHalf  x.half().float() with a straight-through gradient (:27-37)

Each forward is ONE fused operator per quantiser (range pass + elementwise pass) and each backward one operator
(elementwise pass + ordered reduction for the scale / beta gradients and for the gradient that reaches the extremes of
the log range); the reference issues a few dozen elementwise torch kernels and min()/max() reductions for the same
work.  VectorQuantizer (the `vq` colour option) wraps the third-party package `vector_quantize_pytorch` and is not
provided.  Inputs must live on the GPU: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib

LSQ, LOG = 0, 1

# what `from quantize import *` gives the reference's model files (quantize.py has no __all__; these are its public
# names that are part of this path)
__all__ = ["myabs", "mysign", "grad_scale", "ste", "FakeQuantizationHalf", "UniformQuantizer", "LogQuantizer",
           "VectorQuantizer", "HybirdQuant"]


# Torch-level counterparts of the four small tensor helpers at quantize.py:11-24, for code that composes its own
# quantiser in torch.  The quantiser classes below do not use them: rounding and the straight-through gradient live in
# the HIP operators.
def myabs(x):
    """|x| that leaves exact zeros (and their sign bit) untouched."""
    return torch.abs(x).where(x != 0, x)


def mysign(x):
    """sign(x) with sign(0) = +1."""
    return torch.sign(x) + (x == 0).to(x.dtype)


def grad_scale(x, scale):
    """Value of x, gradient multiplied by `scale` (LSQ's step-size gradient scaling)."""
    scaled = x * scale
    return scaled + (x - scaled).detach()


def ste(x):
    """Round half to even in the forward pass, identity gradient."""
    return x + (torch.round(x) - x).detach()


class _QuantSpec(C.Structure):
    """struct gi2d_quant_spec (include/gi2d.h)."""
    _fields_ = [("channels", C.c_int32), ("kind", C.c_int32 * 4), ("qmin", C.c_float * 4), ("qmax", C.c_float * 4)]


def make_spec(kinds, qmins, qmaxs) -> _QuantSpec:
    s = _QuantSpec()
    s.channels = len(kinds)
    for i, (k, lo, hi) in enumerate(zip(kinds, qmins, qmaxs)):
        s.kind[i], s.qmin[i], s.qmax[i] = int(k), float(lo), float(hi)
    return s


def _check_rows(x: torch.Tensor, channels: int) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("gaussianimage_plus_amd.quantize: the tensor must live on the GPU (no CPU fallback)")
    if x.dim() != 2 or x.shape[1] != channels:
        raise ValueError(f"expected a [N, {channels}] tensor, got {tuple(x.shape)}")
    return x.contiguous().float()


def _stream(x):
    return torch.cuda.current_stream(x.device).cuda_stream


def _workspace(x):
    nbytes = _lib.load().gi2d_quant_workspace_bytes(x.shape[0])
    return torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=x.device), nbytes


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _params_of(spec, device, scale=None, beta=None):
    """f32[C][4] = {scale, beta, max t, 0}; LSQ rows filled from the learned tensors."""
    p = torch.zeros(spec.channels, 4, device=device)
    if scale is not None:
        lsq = [c for c in range(spec.channels) if spec.kind[c] == LSQ]
        p[lsq, 0] = scale.detach().float().reshape(-1)
        p[lsq, 1] = beta.detach().float().reshape(-1)
    return p


class _QuantRows(torch.autograd.Function):
    """Rows [N,C] through gi2d_quant_forward / gi2d_quant_backward.  `scale`, `beta` hold the learned values of the
    LSQ channels in channel order (or None when the spec has none)."""

    @staticmethod
    def forward(ctx, x, scale, beta, spec):
        xc = _check_rows(x, spec.channels)
        params = _params_of(spec, xc.device, scale, beta)
        deq, code = torch.empty_like(xc), torch.empty_like(xc)
        ws, nbytes = _workspace(xc)
        with torch.cuda.device(xc.device):
            _lib.call("gi2d_quant_forward", C.byref(spec), xc.shape[0], _ptr(xc), _ptr(params), _ptr(deq), _ptr(code),
                      _ptr(ws), nbytes, _stream(xc))
        ctx.spec = spec
        ctx.save_for_backward(xc, params)
        ctx.has_learned = scale is not None
        ctx.mark_non_differentiable(code, params)
        return deq, code, params

    @staticmethod
    def backward(ctx, v_deq, _v_code, _v_params):
        xc, params = ctx.saved_tensors
        spec = ctx.spec
        g = v_deq.contiguous().float()
        v_x = torch.empty_like(xc)
        v_p = torch.empty(spec.channels, 2, device=xc.device)
        ws, nbytes = _workspace(xc)
        with torch.cuda.device(xc.device):
            _lib.call("gi2d_quant_backward", C.byref(spec), xc.shape[0], _ptr(xc), _ptr(params), _ptr(g), _ptr(v_x),
                      _ptr(v_p), _ptr(ws), nbytes, _stream(xc))
        v_scale = v_beta = None
        if ctx.has_learned:
            lsq = [c for c in range(spec.channels) if spec.kind[c] == LSQ]
            v_scale, v_beta = v_p[lsq, 0].contiguous(), v_p[lsq, 1].contiguous()
        return v_x, v_scale, v_beta, None


def _compress(spec, x, params):
    xc = _check_rows(x, spec.channels)
    deq, code = torch.empty_like(xc), torch.empty_like(xc)
    with torch.cuda.device(xc.device):
        _lib.call("gi2d_quant_compress", C.byref(spec), xc.shape[0], _ptr(xc), _ptr(params), _ptr(deq), _ptr(code),
                  _stream(xc))
    return deq, code


def _decompress(spec, code, params):
    cc = _check_rows(code, spec.channels)
    out = torch.empty_like(cc)
    with torch.cuda.device(cc.device):
        _lib.call("gi2d_quant_decompress", C.byref(spec), cc.shape[0], _ptr(cc), _ptr(params), _ptr(out), _stream(cc))
    return out


def _init_params(spec, x):
    xc = _check_rows(x, spec.channels)
    params = torch.empty(spec.channels, 4, device=xc.device)
    ws, nbytes = _workspace(xc)
    with torch.cuda.device(xc.device):
        _lib.call("gi2d_quant_init", C.byref(spec), xc.shape[0], _ptr(xc), _ptr(params), _ptr(ws), nbytes,
                  _stream(xc))
    return params


class _HalfFn(torch.autograd.Function):
    @staticmethod
    def forward(_, x):
        if not x.is_cuda:
            raise RuntimeError("gaussianimage_plus_amd.quantize: the tensor must live on the GPU (no CPU fallback)")
        xc = x.contiguous().float()
        y = torch.empty_like(xc)
        with torch.cuda.device(xc.device):
            _lib.call("gi2d_quant_half", xc.numel(), _ptr(xc), _ptr(y), _stream(xc))
        return y

    @staticmethod
    def backward(_, grad_output):
        return grad_output


class FakeQuantizationHalf:
    """performs fake quantization for half precision (quantize.py:27-37); use as FakeQuantizationHalf.apply(x)."""
    apply = _HalfFn.apply


def _range(signed, bits):
    if signed:
        return -2 ** (bits - 1), 2 ** (bits - 1) - 1
    return 0, 2 ** bits - 1


class UniformQuantizer(nn.Module):
    """LSQ+ (quantize.py:39-156): per-channel learned scale and offset, initialised from the first batch."""

    def __init__(self, signed=False, bits=8, learned=False, num_channels=1, entropy_type="none", weight=0.0001):
        super().__init__()
        self.bits = bits
        self.init_state = 0
        self.batch_init = 20
        self.qmin, self.qmax = _range(signed, bits)
        self.qm = self.qmax
        self.learned = learned
        self.entropy_type = entropy_type
        self.num_channels = num_channels
        if self.learned:
            self.scale = nn.Parameter(torch.ones(num_channels) / self.qmax, requires_grad=True)
            self.beta = nn.Parameter(torch.ones(num_channels) / self.qmax, requires_grad=True)

    def _spec(self, channels):
        return make_spec([LSQ] * channels, [self.qmin] * channels, [self.qm] * channels)

    def _init_data(self, tensor):
        p = _init_params(self._spec(tensor.shape[1]), tensor.detach())
        self.scale.data = p[:, 0].contiguous()
        self.beta.data = p[:, 1].contiguous()

    def forward(self, x, quant_loss=False):
        bits, entropy_loss = 0, 0
        if self.init_state == 0:
            self._init_data(x)
            self.init_state += 1
        c = x.shape[1]
        scale = self.scale.expand(c) if self.scale.numel() == 1 else self.scale
        beta = self.beta.expand(c) if self.beta.numel() == 1 else self.beta
        dequant, quant, _ = _QuantRows.apply(x, scale, beta, self._spec(c))
        return dequant, entropy_loss, bits, quant

    def size(self):
        return self.bits

    def reset_state(self):
        self.init_state = 0

    def _params(self, c, device):
        return _params_of(self._spec(c), device, self.scale.expand(c), self.beta.expand(c))

    def compress(self, x):
        c = x.shape[1]
        return _compress(self._spec(c), x, self._params(c, x.device))

    def decompress(self, x):
        c = x.shape[1]
        return _decompress(self._spec(c), x, self._params(c, x.device))


class LogQuantizer(nn.Module):
    """log quantizer (quantize.py:158-259), learned=False: the range is the min/max of log(|x|+1e-6) over the WHOLE
    input of each forward (kept in the autograd graph), per channel in compress(); magnitudes only."""

    def __init__(self, signed=True, bits=8, learned=False, num_channels=1, entropy_type="none", weight=0.001):
        super().__init__()
        if learned:
            raise NotImplementedError("LogQuantizer(learned=True) is not on the path train_quantize.py wires "
                                      "(HybirdQuant builds it with learned=False)")
        self.bits = bits
        self.init_state = 0
        self.qmin, self.qmax = _range(signed, bits)
        self.learned = learned
        self.entropy_type = entropy_type
        self.num_channels = num_channels
        self.beta = torch.empty(num_channels)
        self.scale = torch.empty(num_channels)
        self.weight = weight
        self.min_log, self.max_log = 0, 0
        self.sign = None

    def _spec(self, channels):
        return make_spec([LOG] * channels, [self.qmin] * channels, [self.qmax] * channels)

    def _init_data(self, tensor):
        p = _init_params(self._spec(tensor.shape[1]), tensor.detach())
        self.scale, self.beta = p[:, 0].contiguous(), p[:, 1].contiguous()
        self.min_log, self.max_log = self.beta, p[:, 2].contiguous()
        return p

    def forward(self, x, quant_loss=False):
        bits, entropy_loss = 0, 0
        if self.init_state == 0:
            self.init_state += 1
        dequant, quant, params = _QuantRows.apply(x, None, None, self._spec(x.shape[1]))
        self.scale, self.beta, self.max_log = params[0, 0], params[0, 1], params[0, 2]  # scalars, as in the reference
        return dequant, entropy_loss, bits, quant

    def size(self):
        return self.bits

    def reset_state(self):
        self.init_state = 0

    def compress(self, x):
        p = self._init_data(x)
        self.sign = torch.sign(x)
        return _compress(self._spec(x.shape[1]), x, p)

    def decompress(self, x):
        c = x.shape[1]
        p = torch.zeros(c, 4, device=x.device)
        p[:, 0], p[:, 1] = self.scale.to(x.device).expand(c), self.beta.to(x.device).expand(c)
        return _decompress(self._spec(c), x, p)


class HybirdQuant(nn.Module):
    """Covariance rows (a, b, c) (quantize.py:336-389): a and c through ONE LogQuantizer call (shared range), b through
    an LSQ UniformQuantizer with `cov_bits`.  The whole row is one fused operator here."""

    def __init__(self, signed=False, bits=8, cov_bits=10, learned=False, num_channels=1, entropy_type="none",
                 weight=0.001):
        super().__init__()
        self.init_state = 0
        self.var_quantizer = LogQuantizer(False, bits, learned=False, num_channels=2, entropy_type=entropy_type,
                                          weight=weight)
        self.cov_quantizer = UniformQuantizer(signed, cov_bits, learned=True, num_channels=1,
                                              entropy_type=entropy_type, weight=weight)
        self.bits = bits

    def _spec(self):
        v, c = self.var_quantizer, self.cov_quantizer
        return make_spec([LOG, LSQ, LOG], [v.qmin, c.qmin, v.qmin], [v.qmax, c.qm, v.qmax])

    def _init_data(self, tensor):
        p = _init_params(self._spec(), tensor.detach())
        v, c = self.var_quantizer, self.cov_quantizer
        v.scale, v.beta = p[::2, 0].contiguous(), p[::2, 1].contiguous()
        v.min_log, v.max_log = v.beta, p[::2, 2].contiguous()
        c.scale.data, c.beta.data = p[1:2, 0].contiguous(), p[1:2, 1].contiguous()
        c.init_state = 1
        return p

    def forward(self, x, quant_loss=False):
        if self.init_state == 0:
            self._init_data(x)
            self.init_state += 1
        c = self.cov_quantizer
        dequant, code_quant, params = _QuantRows.apply(x, c.scale, c.beta, self._spec())
        v = self.var_quantizer
        v.scale, v.beta, v.max_log = params[0, 0], params[0, 1], params[0, 2]
        return dequant, 0, 0, code_quant

    def size(self):
        return (self.cov_quantizer.size() + self.var_quantizer.size() * 2) / 3

    def reset_state(self):
        self.var_quantizer.reset_state()
        self.cov_quantizer.reset_state()

    def compress(self, x):
        c = self.cov_quantizer
        p = _init_params(self._spec(), x.detach())  # per-channel log ranges (LogQuantizer.compress re-initialises)
        v = self.var_quantizer
        v.scale, v.beta = p[::2, 0].contiguous(), p[::2, 1].contiguous()
        v.sign = torch.sign(x[:, ::2])
        p[1, 0], p[1, 1] = c.scale.detach()[0], c.beta.detach()[0]
        return _compress(self._spec(), x, p)

    def decompress(self, x):
        v, c = self.var_quantizer, self.cov_quantizer
        p = torch.zeros(3, 4, device=x.device)
        p[::2, 0], p[::2, 1] = v.scale.to(x.device).expand(2), v.beta.to(x.device).expand(2)
        p[1, 0], p[1, 1] = c.scale.detach()[0], c.beta.detach()[0]
        return _decompress(self._spec(), x, p)


class VectorQuantizer(nn.Module):
    """The `vq` colour option (quantize.py:262-333) wraps vector_quantize_pytorch's ResidualVQ; that third-party
    package is not part of this path."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("VectorQuantizer needs the third-party package vector_quantize_pytorch; use "
                                  "color_quant='lsq' (train_quantize.py's default)")
