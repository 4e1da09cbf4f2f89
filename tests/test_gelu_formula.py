"""CPU: the erf-GELU polynomial of the encoder GEMMs' epilogue (bergen_amd/csrc/gemm_f16_kernel.h, bh_gemm::gelu_erf) restated in numpy with
the constants read from the header, against the exact 0.5 x (1 + erf(x / sqrt 2)) the reference's BertIntermediate applies
(transformers activations.py GELUActivation; reached through AutoModel from models/retrievers/dense.py:16).  Bound written here: 2e-5 absolute for
|x| <= 12, 1e-6 for x < -6, 1e-6 relative for x > 6 — a fortieth of an fp16 half-ulp at |y| ~ 1."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_constants():
    src = open(os.path.join(ROOT, "bergen_amd", "csrc", "gemm_f16_kernel.h")).read()
    c = float(re.search(r"GELU_C = ([0-9.]+)f", src).group(1))
    body = re.search(r"GELU_R\[(\d+)\] = \{([^}]*)\}", src)
    coef = [float(v.strip().rstrip("f")) for v in body.group(2).split(",")]
    assert len(coef) == int(body.group(1))
    assert re.search(r"GELU_TA = 2\.0f / \(4\.5f \* 4\.5f\)", src) and c == 4.5
    return c, coef


def test_gelu_polynomial_is_within_its_stated_error():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from fit_gelu import Phi, gelu_f32
    c, coef = header_constants()
    xs = np.concatenate([np.linspace(-12, 12, 400_001), -np.logspace(0, 4.8, 5000), np.logspace(0, 4.8, 5000), [0.0, -0.0, c, -c]])
    ref = xs * Phi(xs)
    got = gelu_f32(xs, coef, c).astype(np.float64)
    err = np.abs(got - ref)
    assert err[np.abs(xs) <= 12].max() <= 2e-5
    assert err[xs < -6].max() <= 1e-6
    assert (err[xs > 6] / ref[xs > 6]).max() <= 1e-6
    assert gelu_f32(np.array([0.0]), coef, c)[0] == 0.0


def test_gelu_polynomial_is_monotone_where_gelu_is():
    """x Phi(x) increases for x > -0.7518; the approximation must not wiggle there by more than its error allows (an fp16 output cannot see it)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from fit_gelu import gelu_f32
    c, coef = header_constants()
    xs = np.linspace(-0.75, 12, 200_001)
    y = gelu_f32(xs, coef, c).astype(np.float64)
    assert (np.diff(y) >= -1e-6).all()
