"""Synthetic clouds of BASELINE.json's configs.

Platform independent: raw 32-bit words of numpy's Philox counter generator (a specified, stable
stream) are mapped to float32 by (w >> 8) * 2^-24, never through a library distribution.
"""
import numpy as np


def _words(gen, n):
    """n raw 32-bit words (high halves of Philox's 64-bit outputs)."""
    return (gen.random_raw(n) >> np.uint64(32)).astype(np.uint32)


def _unit(words):
    return ((words >> np.uint32(8)).astype(np.float32)) * np.float32(2.0 ** -24)


def uniform_cloud(n, seed):
    """n points i.i.d. U[0,1)^3, float32, shape (n, 3).  C2: (1e6, 42); C3: (1e7, 43); C5: (5e7, 45)."""
    gen = np.random.Philox(key=seed)
    return _unit(_words(gen, 3 * n)).reshape(n, 3)


def clustered_cloud(n, seed=44, components=64):
    """C4: Gaussian mixture, `components` isotropic clusters, centres U[0.1,0.9]^3, sigma log-uniform in
    [0.004, 0.04], equal weights, Box-Muller on the same stream, samples outside [0,1]^3 re-drawn."""
    gen = np.random.Philox(key=seed)
    centres = (0.1 + 0.8 * _unit(_words(gen, 3 * components))).reshape(components, 3).astype(np.float32)
    u = _unit(_words(gen, components))
    sigma = (0.004 * np.exp(u.astype(np.float64) * np.log(10.0))).astype(np.float32)
    out = np.empty((n, 3), np.float32)
    comp = (np.arange(n, dtype=np.int64) * components // n).astype(np.int64)  # equal weights, contiguous blocks
    todo = np.arange(n, dtype=np.int64)
    while len(todo):
        m = len(todo)
        u1 = _unit(_words(gen, 3 * m)).astype(np.float64).reshape(m, 3)
        u2 = _unit(_words(gen, 3 * m)).astype(np.float64).reshape(m, 3)
        g = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)  # 1-u1 in (0,1]
        c = comp[todo]
        pts = (centres[c].astype(np.float64) + g * sigma[c, None].astype(np.float64)).astype(np.float32)
        ok = np.all((pts >= 0.0) & (pts < 1.0), axis=1)
        out[todo[ok]] = pts[ok]
        todo = todo[~ok]
    # interleave the clusters so that input order is not spatially sorted
    perm = np.random.Generator(np.random.Philox(key=seed + 1000)).permutation(n)
    return out[perm]


def jitter(points, seed, amplitude=1e-3):
    """C5 streaming step: add U(-amplitude, amplitude) per coordinate."""
    gen = np.random.Philox(key=seed)
    n = points.shape[0]
    u = _unit(_words(gen, 3 * n)).reshape(n, 3)
    return (points + (2.0 * u - 1.0).astype(np.float32) * np.float32(amplitude)).astype(np.float32)
