"""HIP quantiser operators (csrc/gi2d_quant.hip through gaussianimage_plus_amd.quantize) against the reference fixture
(tests/golden/quant_reference.npz) and the oracle (oracle/quant_oracle.py).

Bar: LSQ codes / dequantised values bit-exact (IEEE sub, div, round, mul, add only); log-quantiser codes exact except
where the pre-round value sits within 2e-3 of a half-integer (logf differs by an ulp between libraries), values 1e-5
relative; reduced gradients within 2e-5 of the sum of absolute terms (torch sums in fp32 in another order)."""
import os

import numpy as np
import pytest
import torch

from oracle import quant_oracle as qo

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "quant_reference.npz"))
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(a, b, rtol=1e-5, atol=1e-7):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def codes_match(code, ref, raw):
    code = code.detach().cpu().numpy() if torch.is_tensor(code) else code
    frac = np.abs(raw - np.floor(raw) - 0.5)
    bad = code != ref
    assert np.all(frac[bad] < 2e-3), "a code differs away from a rounding boundary"
    assert bad.mean() < 0.01
    return ~bad


@pytest.mark.parametrize("name", ["xy12", "col6", "col6_init", "rot6s"])
def test_lsq_bit_exact_with_reference(name):
    from gaussianimage_plus_amd.quantize import UniformQuantizer
    x, g, bits = G[f"{name}_x"], G[f"{name}_g"], int(G[f"{name}_bits"])
    signed = name.endswith("s")  # rot6s: the RS model's signed rotation quantiser (models/gaussianimage_rs.py:142)
    q = UniformQuantizer(signed=signed, bits=bits, learned=True, num_channels=x.shape[1]).to(DEV)
    q(t(x))  # data initialisation on the first call (quantize.py:126-128)
    close(q.scale, G[f"{name}_init_scale"], 1e-6)
    close(q.beta, G[f"{name}_init_beta"], 1e-6)
    q.scale.data, q.beta.data = t(G[f"{name}_scale"]), t(G[f"{name}_beta"])
    xin = t(x).requires_grad_(True)
    deq, el, nb, code = q(xin)
    assert (el, nb) == (0, 0)
    assert np.array_equal(code.cpu().numpy(), G[f"{name}_code"])
    assert np.array_equal(deq.detach().cpu().numpy(), G[f"{name}_dequant"])
    (deq * t(g)).sum().backward()
    close(xin.grad, G[f"{name}_v_x"], 1e-6)
    qmin, qmax = qo.qrange(bits, signed=signed)
    _, v_s, v_b = qo.lsq_backward(x, G[f"{name}_scale"], G[f"{name}_beta"], qmin, qmax, g)
    terms = np.abs(g * G[f"{name}_code"]).sum(0) * 2
    assert np.all(np.abs(q.scale.grad.cpu().numpy() - v_s) <= 2e-6 * terms)
    assert np.all(np.abs(q.beta.grad.cpu().numpy() - v_b) <= 2e-6 * np.abs(g).sum(0))
    assert np.all(np.abs(q.scale.grad.cpu().numpy() - G[f"{name}_v_scale"]) <= 2e-5 * terms)
    assert np.all(np.abs(q.beta.grad.cpu().numpy() - G[f"{name}_v_beta"]) <= 2e-5 * np.abs(g).sum(0))
    cd, cc = q.compress(t(x))
    assert np.array_equal(cc.cpu().numpy(), G[f"{name}_compress_code"])
    assert np.array_equal(cd.cpu().numpy(), G[f"{name}_compress_dequant"])
    assert np.array_equal(q.decompress(cc).cpu().numpy(), G[f"{name}_decompress"])


@pytest.mark.parametrize("name", ["var10", "var10_ties"])
def test_log_quantiser_matches_reference(name):
    from gaussianimage_plus_amd.quantize import LogQuantizer
    x, g, bits = G[f"{name}_x"], G[f"{name}_g"], int(G[f"{name}_bits"])
    q = LogQuantizer(False, bits, learned=False, num_channels=2)
    xin = t(x).requires_grad_(True)
    deq, _, _, code = q(xin)
    close(q.beta, G[f"{name}_fwd_beta"], 1e-6)
    close(q.scale, G[f"{name}_fwd_scale"], 1e-6)
    raw = (qo.log_of(x) - G[f"{name}_fwd_beta"]) / G[f"{name}_fwd_scale"]
    same = codes_match(code, G[f"{name}_code"], raw)
    close(deq.detach().cpu().numpy()[same], G[f"{name}_dequant"][same], 1e-5)
    (deq * t(g)).sum().backward()
    v_x, ref, want = xin.grad.cpu().numpy(), G[f"{name}_v_x"], qo.log_backward(x, *qo.qrange(bits), g)
    L = qo.log_of(x)
    ext = (L == L.min()) | (L == L.max())
    close(v_x[same & ~ext], ref[same & ~ext], 2e-5, 1e-7)
    # the extremes carry a reduced sum: tight against the float64 oracle, condition-aware against torch's fp32 sum
    cond = np.abs(g * G[f"{name}_dequant"]).sum() / (2 ** bits - 1) / (np.abs(x[ext]) + 1e-6)
    assert np.all(np.abs(v_x[ext] - want[ext]) <= 1e-5 * cond + 1e-5 * np.abs(want[ext]))
    assert np.all(np.abs(v_x[ext] - ref[ext]) <= 1e-4 * cond + 1e-5 * np.abs(ref[ext]))
    cd, cc = q.compress(t(x))
    close(q.scale, G[f"{name}_compress_scale"], 1e-6)
    close(q.beta, G[f"{name}_compress_beta"], 1e-6)
    rawc = (qo.log_of(x) - G[f"{name}_compress_beta"]) / G[f"{name}_compress_scale"]
    same = codes_match(cc, G[f"{name}_compress_code"], rawc)
    close(cd.cpu().numpy()[same], G[f"{name}_compress_dequant"][same], 1e-5)
    close(q.decompress(t(G[f"{name}_compress_code"])), G[f"{name}_decompress"], 1e-5)


def test_hybrid_quantiser_matches_reference():
    from gaussianimage_plus_amd.quantize import HybirdQuant
    n = "hyb10"
    x, g, bits = G[f"{n}_x"], G[f"{n}_g"], int(G[f"{n}_bits"])
    q = HybirdQuant(signed=False, bits=bits, cov_bits=bits, learned=True, weight=1.0).to(DEV)
    q(t(x))
    q.cov_quantizer.scale.data, q.cov_quantizer.beta.data = t(G[f"{n}_cov_scale"]), t(G[f"{n}_cov_beta"])
    xin = t(x).requires_grad_(True)
    deq, _, _, code = q(xin)
    _, _, lbeta, lscale = qo.hybrid_forward(x, G[f"{n}_cov_scale"], G[f"{n}_cov_beta"], bits, bits)
    raw = np.zeros_like(x)
    raw[:, ::2] = (qo.log_of(x[:, ::2]) - lbeta) / lscale
    same = codes_match(code, G[f"{n}_code"], raw)
    assert np.array_equal(code.cpu().numpy()[:, 1], G[f"{n}_code"][:, 1])
    assert np.array_equal(deq.detach().cpu().numpy()[:, 1], G[f"{n}_dequant"][:, 1])
    close(deq.detach().cpu().numpy()[same], G[f"{n}_dequant"][same], 1e-5)
    (deq * t(g)).sum().backward()
    L = qo.log_of(x[:, ::2])
    ext = np.zeros(x.shape, bool)
    ext[:, ::2] = (L == L.min()) | (L == L.max())
    close(xin.grad.cpu().numpy()[same & ~ext], G[f"{n}_v_x"][same & ~ext], 2e-5, 1e-7)
    want_x, want_s, want_b = qo.hybrid_backward(x, G[f"{n}_cov_scale"], G[f"{n}_cov_beta"], bits, bits, g)
    cond = np.abs(g[:, ::2] * G[f"{n}_dequant"][:, ::2]).sum() / (2 ** bits - 1) / (np.abs(x[ext]) + 1e-6)
    assert np.all(np.abs(xin.grad.cpu().numpy()[ext] - want_x[ext]) <= 1e-5 * cond + 1e-5 * np.abs(want_x[ext]))
    terms = np.abs(g[:, 1] * G[f"{n}_code"][:, 1]).sum() * 2
    assert abs(q.cov_quantizer.scale.grad.item() - want_s[0]) <= 2e-6 * terms
    assert abs(q.cov_quantizer.beta.grad.item() - want_b[0]) <= 2e-6 * np.abs(g[:, 1]).sum()
    assert abs(q.cov_quantizer.scale.grad.item() - G[f"{n}_v_cov_scale"][0]) <= 2e-5 * terms
    assert q.size() == float(G[f"{n}_size"])
    assert sorted(q.state_dict()) == ["cov_quantizer.beta", "cov_quantizer.scale"]  # as the reference's module
    cd, cc = q.compress(t(x))
    _, _, vs, vb = qo.hybrid_compress(x, G[f"{n}_cov_scale"], G[f"{n}_cov_beta"], bits, bits)
    rawc = np.zeros_like(x)
    rawc[:, ::2] = (qo.log_of(x[:, ::2]) - vb) / vs
    same = codes_match(cc, G[f"{n}_compress_code"], rawc)
    close(cd.cpu().numpy()[same], G[f"{n}_compress_dequant"][same], 1e-5)
    close(q.decompress(t(G[f"{n}_compress_code"])), G[f"{n}_decompress"], 1e-5)


def test_half_bit_exact():
    from gaussianimage_plus_amd.quantize import FakeQuantizationHalf
    xin = t(G["half_x"]).requires_grad_(True)
    y = FakeQuantizationHalf.apply(xin)
    assert np.array_equal(y.detach().cpu().numpy(), G["half_y"])
    (y * t(G["half_g"])).sum().backward()
    assert np.array_equal(xin.grad.cpu().numpy(), G["half_v_x"])
    big = torch.tensor([[7e4, -7e4, 65504.0, 1e-8]], device=DEV)  # overflow to inf, underflow to 0 like .half()
    assert np.array_equal(FakeQuantizationHalf.apply(big).cpu().numpy(), qo.half_forward(big.cpu().numpy()))


def test_config5_sizes_against_oracle_and_reproducible():
    """N = 30 000 rows (BASELINE config 5): every quantiser of forward_quantize at its default bit depth."""
    from gaussianimage_plus_amd.quantize import HybirdQuant, UniformQuantizer
    rng = np.random.default_rng(5)
    n = 30000
    cov = np.stack([rng.uniform(0.4, 60, n), rng.normal(0, 4, n), rng.uniform(0.4, 60, n)], 1).astype(np.float32)
    g = rng.normal(0, 1, (n, 3)).astype(np.float32)
    q = HybirdQuant(signed=False, bits=10, cov_bits=10, learned=True, weight=1.0).to(DEV)
    xin = t(cov).requires_grad_(True)
    deq, _, _, code = q(xin)
    s, b = q.cov_quantizer.scale.detach().cpu().numpy(), q.cov_quantizer.beta.detach().cpu().numpy()
    wd, wc, lbeta, lscale = qo.hybrid_forward(cov, s, b, 10, 10)
    raw = np.zeros_like(cov)
    raw[:, ::2] = (qo.log_of(cov[:, ::2]) - lbeta) / lscale
    same = codes_match(code, wc, raw)
    close(deq.detach().cpu().numpy()[same], wd[same], 1e-5)
    (deq * t(g)).sum().backward()
    g1 = xin.grad.clone()
    gs1 = q.cov_quantizer.scale.grad.clone()
    want_x, want_s, want_b = qo.hybrid_backward(cov, s, b, 10, 10, g)
    L = qo.log_of(cov[:, ::2])
    ext = np.zeros(cov.shape, bool)
    ext[:, ::2] = (L == L.min()) | (L == L.max())
    close(g1.cpu().numpy()[same & ~ext], want_x[same & ~ext], 2e-5, 1e-7)
    assert abs(gs1.item() - want_s[0]) <= 2e-6 * np.abs(g[:, 1] * wc[:, 1]).sum() * 2
    xin.grad = None
    q.cov_quantizer.scale.grad = None
    deq2, _, _, _ = q(xin)
    (deq2 * t(g)).sum().backward()
    assert torch.equal(xin.grad, g1) and torch.equal(q.cov_quantizer.scale.grad, gs1)  # ordered sums: bitwise

    xy = (rng.uniform(0, 1, (n, 2)) * [768, 512]).astype(np.float32)
    qx = UniformQuantizer(signed=False, bits=12, learned=True, num_channels=2).to(DEV)
    dx, _, _, cx = qx(t(xy))
    wdx, wcx = qo.lsq_forward(xy, qx.scale.detach().cpu().numpy(), qx.beta.detach().cpu().numpy(), 0, 4095)
    assert np.array_equal(cx.cpu().numpy(), wcx) and np.array_equal(dx.detach().cpu().numpy(), wdx)
    assert cx.min().item() == 0 and cx.max().item() == 4095


def test_errors_and_empty():
    from gaussianimage_plus_amd.quantize import LogQuantizer, UniformQuantizer, VectorQuantizer
    q = UniformQuantizer(signed=False, bits=6, learned=True, num_channels=3).to(DEV)
    with pytest.raises(RuntimeError):
        q(torch.zeros(4, 3))  # CPU tensor: no fallback
    with pytest.raises(NotImplementedError):
        LogQuantizer(False, 8, learned=True)
    with pytest.raises(NotImplementedError):
        VectorQuantizer()
    q.init_state = 1
    d, _, _, c = q(torch.zeros(0, 3, device=DEV))
    assert d.shape == (0, 3) and c.shape == (0, 3)


def test_rs_model_forward_quantize_composition():
    """models/gaussianimage_rs.py:443-471 forward_quantize written with the package's pieces: LSQ quantisers (positions
    12 bit, scaling 6 bit, SIGNED 6-bit rotation, colours 6 bit) in front of project_gaussians_2d_scale_rot and
    rasterize_gaussians_sum; gradients reach every parameter and every quantiser value."""
    import math
    import gaussianimage_plus_amd.gsplat as gs
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.quantize import UniformQuantizer
    n, h, w = 1200, 64, 96
    g = torch.Generator().manual_seed(8)
    xyz = (torch.rand(n, 2, generator=g) * torch.tensor([float(w), float(h)])).to(DEV).requires_grad_(True)
    scaling = (torch.rand(n, 2, generator=g) * 3 + 0.5).to(DEV).requires_grad_(True)
    rotation = torch.rand(n, 1, generator=g).to(DEV).requires_grad_(True)
    feat = (torch.rand(n, 3, generator=g) * 0.4).to(DEV).requires_grad_(True)
    gt = synthetic_image(h, w, 2).to(DEV)
    xyq = UniformQuantizer(signed=False, bits=12, learned=True, num_channels=2).to(DEV)
    sq = UniformQuantizer(signed=False, bits=6, learned=True, num_channels=2).to(DEV)
    rq = UniformQuantizer(signed=True, bits=6, learned=True, num_channels=1).to(DEV)
    fq = UniformQuantizer(signed=False, bits=6, learned=True, num_channels=3).to(DEV)
    means, _, _, cxy = xyq(xyz)
    sc, _, _, cs = sq(scaling)
    rot, _, _, cr = rq(torch.sigmoid(rotation) * 2 * math.pi)
    col, _, _, cc = fq(feat)
    assert cs.max() <= 63 and cs.min() >= 0 and cr.min() >= -32 and cr.max() <= 31 and cxy.max() <= 4095
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    xys, depths, radii, conics, nth = gs.project_gaussians_2d_scale_rot(means, sc, rot, h, w, tb)
    sp = torch.zeros(n, 4, device=DEV)
    img, _, _ = gs.rasterize_gaussians_sum(xys, sp, depths, radii, conics, nth, col, torch.ones(n, 1, device=DEV), h, w,
                                           16, 16, background=torch.ones(3, device=DEV))
    loss = torch.nn.functional.mse_loss(img.clamp(0, 1), gt)
    loss.backward()
    for p in (xyz, scaling, rotation, feat, xyq.scale, xyq.beta, sq.scale, sq.beta, rq.scale, rq.beta, fq.scale, fq.beta):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0
    # dequantised values sit on the grid code * scale + beta (bit for bit)
    assert torch.equal(sc.detach(), cs * sq.scale.detach() + sq.beta.detach())
