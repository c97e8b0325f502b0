"""CPU: the product's entropy-coded size estimate against the oracle's statement of the same model."""
import numpy as np
import torch

from oracle import quant_oracle as qo


def test_code_length_estimate_matches_oracle():
    from gaussianimage_plus_amd.trainer import quantized_gaussian_code_length_bits
    rng = np.random.default_rng(3)
    for codes in (np.rint(rng.normal(500, 80, (4000, 3))).clip(0, 1023), np.rint(rng.normal(20, 9, 3000)).clip(0, 63),
                  np.full(100, 7.0)):
        want = qo.gaussian_code_length_bits(codes)
        got = quantized_gaussian_code_length_bits(torch.from_numpy(codes))
        assert abs(got - want) <= 1e-9 * max(want, 1.0) + 1e-6, (got, want)
