"""CPU: the pre-processing oracle against outputs of Pillow itself (tests/golden/prep_*.npz) — bit-exact."""
import os

import numpy as np
import pytest

from fixture_inputs import prep_inputs
from oracle import preprocess_oracle as po

CASES = ["prep_s120_r32_train", "prep_s100_r37_flat", "prep_s50_r80_train", "prep_s1200_r300_train"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_pillow_chain(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    rois = prep_inputs(z)
    train, res, pad = bool(int(z["train"])), int(z["res"]), int(z["pad"])
    for t in range(rois.shape[0]):
        got = po.finalize_tile(rois[t], res, z["params"][t] if train else None, pad=pad)
        if "out" in z.files:
            assert np.array_equal(got, z["out"][t]), (name, t)
        else:
            assert np.array_equal(got, po.to_tensor_normalize(z["out_u8"][t])), (name, t)


def test_coefficients_are_normalised_fixed_point():
    for s, r in ((1200, 256), (1200, 300), (50, 80), (64, 64)):
        bounds, kk = po.resize_coeffs(s, r)
        assert np.all(np.abs(kk.sum(axis=1) - (1 << po.PRECISION_BITS)) <= kk.shape[1])     # rounding of each weight only
        assert np.all(bounds[:, 0] >= 0) and np.all(bounds[:, 0] + bounds[:, 1] <= s)
