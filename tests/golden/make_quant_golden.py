#!/usr/bin/env python3
"""Dev-container only: drive the REFERENCE's quantisers (/root/reference/quantize.py: UniformQuantizer, LogQuantizer,
HybirdQuant, FakeQuantizationHalf) on CPU with torch autograd over seeded inputs and commit inputs + outputs +
gradients as a fixture (tests/golden/quant_reference.npz).  The fixture pins oracle/quant_oracle.py, which in turn is
what the HIP quantiser kernels are compared with on the GPU.  Nothing of the reference travels: the fixture is data.

quantize.py imports two things this image does not have at module level: the third-party package
`vector_quantize_pytorch` (used only by VectorQuantizer, which is out of scope here) and three ANS helpers from the
reference's top-level utils.py (which itself needs `constriction`).  Neither is touched by the classes driven below, so
the generator registers empty placeholder modules for those two names before the import; the quantiser classes
themselves run as written."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference_quantize():
    vq = types.ModuleType("vector_quantize_pytorch")
    vq.VectorQuantize = vq.ResidualVQ = None
    ut = types.ModuleType("utils")
    ut.compress_matrix_flatten_categorical = ut.decompress_matrix_flatten_categorical = ut.get_np_size = None
    saved = {k: sys.modules.get(k) for k in ("vector_quantize_pytorch", "utils")}
    sys.modules["vector_quantize_pytorch"], sys.modules["utils"] = vq, ut
    sys.path.insert(0, "/root/reference")
    try:
        import quantize  # noqa: E402  (the reference's own file, imported, not copied)
    finally:
        sys.path.pop(0)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return quantize


def lsq_case(Q, out, name, x, bits, g, steps_off=None, perturb=None, signed=False):
    """UniformQuantizer(signed=False, learned=True): data init on first forward, then forward/backward; `perturb`
    moves scale/beta off their initial values so the clamp is active on both ends."""
    c = x.shape[1]
    q = Q.UniformQuantizer(signed=signed, bits=bits, learned=True, num_channels=c)
    xin = x.clone().requires_grad_(True)
    q(xin)  # init_state 0 -> _init_data
    out[f"{name}_init_scale"] = q.scale.detach().numpy().copy()
    out[f"{name}_init_beta"] = q.beta.detach().numpy().copy()
    if perturb is not None:
        with torch.no_grad():
            q.scale.mul_(perturb[0])
            q.beta.add_(perturb[1] * q.scale)
    xin = x.clone().requires_grad_(True)
    deq, _, _, code = q(xin)
    (deq * g).sum().backward()
    out[f"{name}_x"] = x.numpy()
    out[f"{name}_g"] = g.numpy()
    out[f"{name}_bits"] = np.int32(bits)
    out[f"{name}_scale"] = q.scale.detach().numpy().copy()
    out[f"{name}_beta"] = q.beta.detach().numpy().copy()
    out[f"{name}_dequant"] = deq.detach().numpy()
    out[f"{name}_code"] = code.detach().numpy()
    out[f"{name}_v_x"] = xin.grad.numpy()
    out[f"{name}_v_scale"] = q.scale.grad.numpy()
    out[f"{name}_v_beta"] = q.beta.grad.numpy()
    with torch.no_grad():
        cd, cc = q.compress(x)
        out[f"{name}_compress_dequant"] = cd.numpy()
        out[f"{name}_compress_code"] = cc.numpy()
        out[f"{name}_decompress"] = q.decompress(cc).numpy()


def log_case(Q, out, name, x, bits, g):
    q = Q.LogQuantizer(False, bits, learned=False, num_channels=x.shape[1])
    xin = x.clone().requires_grad_(True)
    deq, _, _, code = q(xin)
    (deq * g).sum().backward()
    out[f"{name}_x"] = x.numpy()
    out[f"{name}_g"] = g.numpy()
    out[f"{name}_bits"] = np.int32(bits)
    out[f"{name}_dequant"] = deq.detach().numpy()
    out[f"{name}_code"] = code.detach().numpy()
    out[f"{name}_v_x"] = xin.grad.numpy()
    out[f"{name}_fwd_beta"] = np.float32(q.beta.item())
    out[f"{name}_fwd_scale"] = np.float32(q.scale.item())
    with torch.no_grad():
        cd, cc = q.compress(x)
        out[f"{name}_compress_dequant"] = cd.numpy()
        out[f"{name}_compress_code"] = cc.numpy()
        out[f"{name}_compress_beta"] = q.beta.numpy().copy()
        out[f"{name}_compress_scale"] = q.scale.numpy().copy()
        out[f"{name}_decompress"] = q.decompress(cc).numpy()


def hybrid_case(Q, out, name, x, bits, g, perturb):
    q = Q.HybirdQuant(signed=False, bits=bits, cov_bits=bits, learned=True, weight=1.0)
    xin = x.clone().requires_grad_(True)
    q(xin)
    with torch.no_grad():
        q.cov_quantizer.scale.mul_(perturb[0])
        q.cov_quantizer.beta.add_(perturb[1] * q.cov_quantizer.scale)
    xin = x.clone().requires_grad_(True)
    deq, _, _, code = q(xin)
    (deq * g).sum().backward()
    out[f"{name}_x"] = x.numpy()
    out[f"{name}_g"] = g.numpy()
    out[f"{name}_bits"] = np.int32(bits)
    out[f"{name}_cov_scale"] = q.cov_quantizer.scale.detach().numpy().copy()
    out[f"{name}_cov_beta"] = q.cov_quantizer.beta.detach().numpy().copy()
    out[f"{name}_dequant"] = deq.detach().numpy()
    out[f"{name}_code"] = code.detach().numpy()
    out[f"{name}_v_x"] = xin.grad.numpy()
    out[f"{name}_v_cov_scale"] = q.cov_quantizer.scale.grad.numpy()
    out[f"{name}_v_cov_beta"] = q.cov_quantizer.beta.grad.numpy()
    out[f"{name}_size"] = np.float64(q.size())
    with torch.no_grad():
        cd, cc = q.compress(x)
        out[f"{name}_compress_dequant"] = cd.numpy()
        out[f"{name}_compress_code"] = cc.numpy()
        out[f"{name}_decompress"] = q.decompress(cc).numpy()


def main():
    Q = _import_reference_quantize()
    gen = torch.Generator().manual_seed(2025)
    out = {}
    n = 257
    # positions in pixels (models/gaussianimage_covariance.py:52-54), 12 bit
    xy = torch.rand(n, 2, generator=gen) * torch.tensor([768.0, 512.0])
    lsq_case(Q, out, "xy12", xy, 12, torch.randn(n, 2, generator=gen), perturb=(1.02, 3.0))
    # colours, 6 bit, three channels
    col = torch.randn(n, 3, generator=gen) * 0.4 + 0.3
    lsq_case(Q, out, "col6", col, 6, torch.randn(n, 3, generator=gen), perturb=(1.05, 1.5))
    # unperturbed: nothing clamps right after the data initialisation
    lsq_case(Q, out, "col6_init", col, 6, torch.randn(n, 3, generator=gen))
    # rotation angles through a SIGNED 6-bit quantiser (models/gaussianimage_rs.py:142: the RS model's rotation_quantizer)
    rot = torch.sigmoid(torch.randn(n, 1, generator=gen)) * 2 * 3.141592653589793
    lsq_case(Q, out, "rot6s", rot, 6, torch.randn(n, 1, generator=gen), perturb=(1.04, 1.0), signed=True)
    # variances (two channels, one global log range), 10 bit; include a negative and a zero entry
    var = torch.rand(n, 2, generator=gen) * 40.0 + 0.3
    var[5, 0] = -2.5
    var[9, 1] = 0.0
    log_case(Q, out, "var10", var, 10, torch.randn(n, 2, generator=gen))
    # ties at the extremes of the log range: min()/max() spread their gradient evenly
    var_t = var.clone()
    var_t[9, 1] = 0.25
    var_t[17, 0] = var_t[40, 1] = var_t.abs().max()
    var_t[3, 0] = var_t[77, 1] = var_t[100, 0] = 0.25
    log_case(Q, out, "var10_ties", var_t, 10, torch.randn(n, 2, generator=gen))
    # covariance triplets (a, b, c) = _cov2d + bound
    cov = torch.cat([var[:, :1].abs() + 0.5, torch.randn(n, 1, generator=gen) * 3.0, var[:, 1:].abs() + 0.5], 1)
    hybrid_case(Q, out, "hyb10", cov, 10, torch.randn(n, 3, generator=gen), perturb=(1.03, 2.0))
    # FakeQuantizationHalf
    xh = torch.randn(n, 2, generator=gen) * 300.0
    xin = xh.clone().requires_grad_(True)
    yh = Q.FakeQuantizationHalf.apply(xin)
    gh = torch.randn(n, 2, generator=gen)
    (yh * gh).sum().backward()
    out["half_x"], out["half_y"], out["half_g"], out["half_v_x"] = xh.numpy(), yh.detach().numpy(), gh.numpy(), \
        xin.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "quant_reference.npz"), **out)
    print("wrote quant_reference.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
