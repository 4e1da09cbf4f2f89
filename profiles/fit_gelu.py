#!/usr/bin/env python
"""The coefficients of gemm_f16_kernel.h's erf-GELU polynomial (bh_gemm::GELU_R):
    Phi(x) - 1/2 ~= xc R(t),  xc = clamp(x, -c, c),  t = 2 xc^2 / c^2 - 1,  gelu(x) ~= max(x, -c) (1/2 + xc R(t))
R = weighted minimax (Lawson-iterated least squares on a Chebyshev basis, weight x^2 = the GELU error an error of R causes) of
(Phi(x) - 1/2) / x over (0, c] under the constraint c R(1) = 1/2, converted to the power basis in t (Horner in t is well conditioned in
fp32; Horner in x^2 is not: 6e-5 at the same degree).  Prints, per (c, degree), the error of the fp32 evaluation over a dense grid.
    python profiles/fit_gelu.py"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf


def Phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def fit_t(c, deg):
    x = np.linspace(1e-4, c, 40001)
    target = (Phi(x) - 0.5) / x
    V = C.chebvander(2 * x * x / (c * c) - 1, deg)
    Vc = V[:, 1:] - V[:, [0]]  # constraint R(t = 1) = 1 / (2 c): coefficient 0 = 1 / (2 c) - the sum of the others (T_k(1) = 1)
    tc = target - 0.5 / c
    w = x * x
    lw = np.ones_like(x)
    for _ in range(80):
        W = w * lw
        cf, *_ = np.linalg.lstsq(Vc * W[:, None], tc * W, rcond=None)
        err = np.abs(w * (Vc @ cf - tc))
        lw = lw * np.maximum(err / err.mean(), 1e-6) ** 0.5
        lw /= lw.mean()
    return C.cheb2poly(np.concatenate([[0.5 / c - cf.sum()], cf]))


def gelu_f32(x, pt, c):
    """the kernel's arithmetic: fp32, every fma rounded once (computed in fp64, rounded to fp32)"""
    f = np.float32
    x = x.astype(f)
    xc = np.clip(x, f(-c), f(c))
    s = (xc * xc).astype(f)
    t = (s.astype(np.float64) * np.float64(f(2.0 / (c * c))) - 1.0).astype(f)
    r = np.full_like(x, f(pt[-1]))
    for k in range(len(pt) - 2, -1, -1):
        r = (r.astype(np.float64) * t.astype(np.float64) + np.float64(f(pt[k]))).astype(f)
    p = (xc.astype(np.float64) * r.astype(np.float64) + 0.5).astype(f)
    return (np.maximum(x, f(-c)) * p).astype(f)


if __name__ == "__main__":
    xs = np.concatenate([np.linspace(-12, 12, 2_000_001), -np.logspace(0, 4.8, 20000), np.logspace(0, 4.8, 20000)])
    ref = xs * Phi(xs)
    for c, deg in [(4.5, 8), (4.5, 9), (4.75, 9), (4.75, 10), (5.0, 10), (5.0, 11)]:
        pt = fit_t(c, deg)
        err = np.abs(gelu_f32(xs, pt, c).astype(np.float64) - ref)
        rel = err / np.maximum(np.abs(ref), 1e-30)
        print(f"c {c} degree {deg}: max |error| {err[np.abs(xs) <= 12].max():.3e} (|x| <= 12), {err[xs < -6].max():.3e} (x < -6), relative {rel[xs > 6].max():.3e} (x > 6)")
        print("   R (t^0 ..):", ", ".join("%.9e" % v for v in pt))
