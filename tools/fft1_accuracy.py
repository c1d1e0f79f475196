#!/usr/bin/env python3
"""Accuracy of the batched 1-D transform (ipde_fft1_c2c) by length: random data against numpy, and a
SMOOTH periodic function (spectrum decaying to rounding level, what the annular solvers transform)
against a long-double dense DFT — lengths with a large prime factor go through rocFFT's Bluestein
path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ipde_amd.spectral import fft1

rng = np.random.default_rng(0)
for n in (2048, 2400, 2392, 2390, 1370, 1195, 4780, 9560, 9568):
    a = rng.standard_normal((14, n)) + 1j * rng.standard_normal((14, n))
    f = np.asarray(fft1(a, -1))
    ref = np.fft.fft(a, axis=1)
    t = 2 * np.pi * np.arange(n) / n
    s = (np.exp(np.cos(3 * t)) * np.sin(t + 0.3) + 1j * np.exp(np.sin(2 * t)))[None, :]
    fs = np.asarray(fft1(s, -1))[0]
    if n <= 5000:     # long-double dense DFT of the smooth function
        k = np.arange(n, dtype=np.longdouble)
        ang = (np.outer(np.arange(n), np.arange(n)) % n).astype(np.longdouble) * (-2 * np.pi / np.longdouble(n))
        W = np.cos(ang) + 1j * np.sin(ang)
        exact = (W @ s[0].astype(np.clongdouble)).astype(complex)
    else:
        exact = np.fft.fft(s[0])
    e = np.abs(fs - exact)
    print("n = %5d   random: rel err %.2e   smooth: max abs err / max|F| %.2e, rms %.2e, numpy's own %.2e" %
          (n, np.abs(f - ref).max() / np.abs(ref).max(), e.max() / np.abs(exact).max(),
           np.sqrt(np.mean(e ** 2)) / np.abs(exact).max(),
           np.abs(np.fft.fft(s[0]) - exact).max() / np.abs(exact).max()), flush=True)
