#!/usr/bin/env python3
"""Fit of the packed GELU in csrc/gemm.hip (gelu_pk): gelu(x) = x / (1 + exp(-x q(x^2))) with x q(x^2) ~ logit(Phi(x)).
q is a polynomial in x^2, fitted by weighted linear least squares with Lawson reweighting towards the minimax ABSOLUTE error
of gelu on |x| <= 7 (sensitivity d gelu / d q = x^2 sigma (1 - sigma)); prints the coefficients and the fp32-evaluated error."""
import numpy as np
from scipy.special import erf, log_ndtr

x = np.linspace(1e-3, 7, 70001)
g = 0.5 * x * (1 + erf(x / np.sqrt(2)))
p_true = log_ndtr(x) - log_ndtr(-x)
q_true = p_true / x
sig = 1 / (1 + np.exp(-p_true))
sens = x * x * sig * (1 - sig)
gn_true = 0.5 * (-x) * (1 + erf(-x / np.sqrt(2)))
for nterm in (5, 6, 7):
    V = np.vander(x * x, nterm, increasing=True)
    lw = np.ones_like(x)
    best = None
    for it in range(200):
        W = sens * lw
        c, *_ = np.linalg.lstsq(V * W[:, None], q_true * W, rcond=None)
        q = V @ c
        e = np.maximum(np.abs(x / (1 + np.exp(-(x * q))) - g), np.abs(-x / (1 + np.exp(x * q)) - gn_true))
        if best is None or e.max() < best[0]:
            best = (e.max(), c.copy())
        lw = lw * (1 + 2 * e / e.max())
        lw /= lw.mean()
    emax, c = best
    cf = (-(c * 1.4426950408889634)).astype(np.float32)
    xf = np.linspace(-30, 30, 600001).astype(np.float32)
    xc = np.clip(xf, -12, 12)
    x2 = xc * xc
    p = cf[-1]
    for k in range(len(cf) - 2, -1, -1):
        p = p * x2 + cf[k]
    with np.errstate(over="ignore"):
        y = xf * (np.float32(1) / (np.float32(1) + np.exp2(p * xc)))
    xd = xf.astype(np.float64)
    e32 = np.abs(y.astype(np.float64) - 0.5 * xd * (1 + erf(xd / np.sqrt(2)))).max()
    print("%d terms: max abs err %.3e (fp32 evaluation, clamped: %.3e); coefficients x -log2(e): %s" %
          (nterm, emax, e32, ", ".join("%.9e" % v for v in cf)))
