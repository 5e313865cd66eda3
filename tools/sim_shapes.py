"""CPU model: leaf size and tree arity under the Hilbert order.  Cost model per 64-query group in wave VALU instructions:
13 per candidate test, 12.5 per child-box test.  usage: python tools/sim_shapes.py [uniform|clustered] [n] [groups]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.argv_backup = list(sys.argv)
import importlib
syn = importlib.import_module("point-cloud-processing_amd.synthetic")
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ngroups = int(sys.argv[3]) if len(sys.argv) > 3 else 100
K = 15
BITS = 13
pts = syn.uniform_cloud(n, 43) if kind == "uniform" else syn.clustered_cloud(n, 44)
lo, hi = pts.min(0), pts.max(0)
q = np.minimum(((pts - lo) / (hi - lo) * (1 << BITS)).astype(np.int64), (1 << BITS) - 1).astype(np.uint32)
exec(open("tools/sim_hilbert.py").read().split("def build(order):")[0].split("def interleave(X):")[1].join(["def interleave(X):", ""]))
order = np.argsort(hilbert_key(q), kind="stable")
sp = pts[order].astype(np.float64)

def boxd2(lo_, hi_, qq):
    d = np.maximum(np.maximum(lo_[:, None, :] - qq[None, :, :], qq[None, :, :] - hi_[:, None, :]), 0.0)
    return (d * d).sum(-1)

for LEAF, W in ((8, 4), (8, 8), (8, 2), (4, 4), (16, 4), (16, 2), (4, 8)):
    nleaves = (n + LEAF - 1) // LEAF
    pad = nleaves * LEAF - n
    spp = np.concatenate([sp, np.full((pad, 3), np.nan)]) if pad else sp
    L = spp.reshape(nleaves, LEAF, 3)
    depth = 0
    while W ** depth < nleaves:
        depth += 1
    blo = np.full((W ** depth, 3), np.inf); bhi = np.full((W ** depth, 3), -np.inf)
    blo[:nleaves] = np.nanmin(L, axis=1); bhi[:nleaves] = np.nanmax(L, axis=1)
    levels = {depth: (blo, bhi)}
    for d in range(depth - 1, -1, -1):
        clo, chi = levels[d + 1]
        levels[d] = (clo.reshape(-1, W, 3).min(1), chi.reshape(-1, W, 3).max(1))
    rng = np.random.default_rng(1)
    G = n // 64
    lv_, ex_, real_ = [], [], []
    lpg = 64 // LEAF
    extra = 16 // LEAF  # two 8-point leaves' worth of extra seeds on either side
    for g in rng.integers(2, G - 2, ngroups):
        qs = sp[g * 64:(g + 1) * 64]
        s0, s1 = g * lpg - extra, g * lpg + lpg + extra
        seedpts = sp[s0 * LEAF:s1 * LEAF]
        dd = ((seedpts[None, :, :] - qs[:, None, :]) ** 2).sum(-1)
        dd[dd < 1e-20] = np.inf
        seeded = np.sort(dd, axis=1)[:, K - 1]
        cap = 1.375 * np.median(seeded[1::4])
        tau = np.minimum(seeded, cap) * (1 + 1e-12)
        frontier = np.array([0]); nexp = 0; nreal = 0
        for d in range(depth):
            nexp += len(frontier)
            ch = (frontier[:, None] * W + np.arange(W)[None, :]).ravel()
            clo, chi = levels[d + 1]
            nreal += np.isfinite(clo[ch, 0]).sum()
            need = (boxd2(clo[ch], chi[ch], qs) <= tau[None, :]).any(1)
            frontier = ch[need]
        lv = frontier[(frontier < s0) | (frontier >= s1)]
        lv_.append(len(lv)); ex_.append(nexp); real_.append(nreal)
    cand = (np.mean(lv_) + (s1 - s0)) * LEAF
    cost = cand * 13 + np.mean(ex_) * W * 12.5
    print("LEAF %2d W %d depth %2d: %.1f leaves + %d seed, %.1f expansions (%.0f child boxes) -> %4.0f candidates/query, model VALU %5.0f"
          % (LEAF, W, depth, np.mean(lv_), s1 - s0, np.mean(ex_), np.mean(ex_) * W, cand, cost))
