"""The quantiser oracle (oracle/quant_oracle.py) against the fixture recorded from the reference's own classes under
torch autograd (tests/golden/quant_reference.npz, made by tests/golden/make_quant_golden.py).

Tolerances: integer codes are exact except where the pre-round value sits within 2e-3 of a half-integer AND the
quantiser goes through log() (libm vs torch differ by an ulp there, which can flip the rounding); dequantised values
1e-5 relative; reduced gradients (sums of N terms with cancellation) 2e-5 of the sum of absolute terms."""
import os

import numpy as np
import pytest

from oracle import quant_oracle as qo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "quant_reference.npz"))


def close(a, b, rtol=1e-5, atol=1e-7):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ["xy12", "col6", "col6_init", "rot6s"])
def test_lsq_matches_reference(name):
    x, g, bits = G[f"{name}_x"], G[f"{name}_g"], int(G[f"{name}_bits"])
    qmin, qmax = qo.qrange(bits, signed=name.endswith("s"))  # rot6s: the RS model's signed rotation quantiser
    s0, b0 = qo.lsq_init(x, qmin, qmax)
    close(s0, G[f"{name}_init_scale"], 1e-6)
    close(b0, G[f"{name}_init_beta"], 1e-6)
    s, b = G[f"{name}_scale"], G[f"{name}_beta"]
    deq, code = qo.lsq_forward(x, s, b, qmin, qmax)
    assert np.array_equal(code, G[f"{name}_code"])  # IEEE sub/div/round only: bit exact
    assert np.array_equal(deq, G[f"{name}_dequant"])
    v_x, v_s, v_b = qo.lsq_backward(x, s, b, qmin, qmax, g)
    close(v_x, G[f"{name}_v_x"], 1e-6)
    scale_terms = np.abs(g * code).sum(0) + np.abs(g * (x - b) / s).sum(0)
    assert np.all(np.abs(v_s - G[f"{name}_v_scale"]) <= 2e-5 * scale_terms)
    assert np.all(np.abs(v_b - G[f"{name}_v_beta"]) <= 2e-5 * np.abs(g).sum(0))
    if name in ("xy12", "col6"):  # these perturbed cases clamp at both ends: the sums are not just rounding noise
        assert np.all(np.abs(G[f"{name}_v_beta"]) > 1e-3)
    cd, cc = qo.lsq_compress(x, s, b, qmin, qmax)
    assert np.array_equal(cc, G[f"{name}_compress_code"])
    assert np.array_equal(cd, G[f"{name}_compress_dequant"])
    assert np.array_equal(qo.lsq_decompress(cc, s, b), G[f"{name}_decompress"])


def _codes_match(code, ref, raw):
    frac = np.abs(raw - np.floor(raw) - 0.5)
    bad = code != ref
    assert np.all(frac[bad] < 2e-3), "a code differs away from a rounding boundary"
    assert bad.mean() < 0.01
    return ~bad


@pytest.mark.parametrize("name", ["var10", "var10_ties"])
def test_log_matches_reference(name):
    x, g, bits = G[f"{name}_x"], G[f"{name}_g"], int(G[f"{name}_bits"])
    qmin, qmax = qo.qrange(bits)
    deq, code, beta, scale = qo.log_forward(x, qmin, qmax)
    close(beta, G[f"{name}_fwd_beta"], 1e-6)
    close(scale, G[f"{name}_fwd_scale"], 1e-6)
    raw = (qo.log_of(x) - beta) / scale
    same = _codes_match(code, G[f"{name}_code"], raw)
    close(deq[same], G[f"{name}_dequant"][same], 1e-5)
    assert np.all(deq > 0)  # the sign is dropped (quantize.py:231-232)
    v_x = qo.log_backward(x, qmin, qmax, g)
    ref = G[f"{name}_v_x"]
    # elementwise part to 1e-5; the entries at the extremes of the range also carry a reduced sum
    L = qo.log_of(x)
    ext = (L == L.min()) | (L == L.max())
    close(v_x[same & ~ext], ref[same & ~ext], 2e-5, 1e-7)
    cond = np.abs(g * deq).sum() / (qmax - qmin) / (np.abs(x[ext]) + 1e-6)
    assert np.all(np.abs(v_x[ext] - ref[ext]) <= 1e-4 * cond + 1e-5 * np.abs(ref[ext]))
    assert ext.sum() == (2 if name == "var10" else 7)  # ties: 4 at the minimum, 3 at the maximum
    assert v_x[9, 1] == 0 or name != "var10"  # x == 0: torch.abs has zero gradient there, even at the range minimum
    cd, cc, cs, cb = qo.log_compress(x, qmin, qmax)
    close(cs, G[f"{name}_compress_scale"], 1e-6)
    close(cb, G[f"{name}_compress_beta"], 1e-6)
    rawc = (qo.log_of(x) - cb) / cs
    same = _codes_match(cc, G[f"{name}_compress_code"], rawc)
    close(cd[same], G[f"{name}_compress_dequant"][same], 1e-5)
    close(qo.log_decompress(G[f"{name}_compress_code"], cs, cb), G[f"{name}_decompress"], 1e-5)


def test_hybrid_matches_reference():
    n = "hyb10"
    x, g, bits = G[f"{n}_x"], G[f"{n}_g"], int(G[f"{n}_bits"])
    s, b = G[f"{n}_cov_scale"], G[f"{n}_cov_beta"]
    deq, code, lbeta, lscale = qo.hybrid_forward(x, s, b, bits, bits)
    raw = np.zeros_like(x)
    raw[:, ::2] = (qo.log_of(x[:, ::2]) - lbeta) / lscale
    same = _codes_match(code, G[f"{n}_code"], raw)
    assert np.array_equal(code[:, 1], G[f"{n}_code"][:, 1])
    close(deq[same], G[f"{n}_dequant"][same], 1e-5)
    v_x, v_s, v_b = qo.hybrid_backward(x, s, b, bits, bits, g)
    L = qo.log_of(x[:, ::2])
    ext = np.zeros(x.shape, bool)
    ext[:, ::2] = (L == L.min()) | (L == L.max())
    close(v_x[same & ~ext], G[f"{n}_v_x"][same & ~ext], 2e-5, 1e-7)
    assert abs(v_s[0] - G[f"{n}_v_cov_scale"][0]) <= 2e-5 * (np.abs(g[:, 1] * code[:, 1]).sum() * 2)
    assert abs(v_b[0] - G[f"{n}_v_cov_beta"][0]) <= 2e-5 * np.abs(g[:, 1]).sum()
    assert qo.hybrid_size(bits, bits) == float(G[f"{n}_size"])
    cd, cc, vs, vb = qo.hybrid_compress(x, s, b, bits, bits)
    rawc = np.zeros_like(x)
    rawc[:, ::2] = (qo.log_of(x[:, ::2]) - vb) / vs
    same = _codes_match(cc, G[f"{n}_compress_code"], rawc)
    close(cd[same], G[f"{n}_compress_dequant"][same], 1e-5)
    close(qo.hybrid_decompress(G[f"{n}_compress_code"], s, b, vs, vb), G[f"{n}_decompress"], 1e-5)


def test_half_matches_reference():
    assert np.array_equal(qo.half_forward(G["half_x"]), G["half_y"])
    assert np.array_equal(qo.half_backward(G["half_g"]), G["half_v_x"])


def test_size_arithmetic():
    # models/gaussianimage_covariance.py:469-509 with the default bit depths (train_quantize.py:330-332)
    a = qo.analysis_bits(30000, 512, 768)
    hw = 512 * 768
    assert a["position_bpp"] == (30000 * 2 * 12 + 128) / hw
    assert a["cholesky_bpp"] == (30000 * 3 * 10 + 192) / hw
    assert a["feature_dc_bpp"] == (30000 * 3 * 6 + 192) / hw
    assert abs(a["bpp"] - (a["position_bpp"] + a["cholesky_bpp"] + a["feature_dc_bpp"])) < 1e-12
    assert qo.analysis_bits(100, 16, 16, xy_quant="fp16")["position_bpp"] == 100 * 2 * 16 / 256
    rng = np.random.default_rng(0)
    codes = np.rint(rng.normal(30, 6, 5000)).clip(0, 63)
    bits = qo.gaussian_code_length_bits(codes)
    assert 4.0 < bits / codes.size < 5.2  # entropy of a std-6 discretised gaussian is ~4.63 bits
