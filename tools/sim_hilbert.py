"""CPU model: does a Hilbert order (instead of Morton) make the leaves and 64-query groups compact enough to shrink the
union a wave walks?  Same tree (leaves of 8, 4-ary heap of tight boxes), same seeded radii; only the sort key changes.
usage: python tools/sim_hilbert.py [uniform|clustered] [n] [groups] [bits]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import importlib
syn = importlib.import_module("point-cloud-processing_amd.synthetic")
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ngroups = int(sys.argv[3]) if len(sys.argv) > 3 else 150
BITS = int(sys.argv[4]) if len(sys.argv) > 4 else 13
K = 15
pts = syn.uniform_cloud(n, 43) if kind == "uniform" else syn.clustered_cloud(n, 44)
lo, hi = pts.min(0), pts.max(0)
q = np.minimum(((pts - lo) / (hi - lo) * (1 << BITS)).astype(np.int64), (1 << BITS) - 1).astype(np.uint32)

def interleave(X):  # X: (n,3) uint32 with BITS bits -> key, x most significant
    key = np.zeros(len(X), np.uint64)
    for b in range(BITS - 1, -1, -1):
        for a in range(3):
            key = (key << np.uint64(1)) | ((X[:, a] >> np.uint32(b)) & np.uint32(1)).astype(np.uint64)
    return key

def hilbert_key(Xin):
    """Skilling, 'Programming the Hilbert curve' (2004): axes -> transposed Hilbert index, vectorised"""
    X = Xin.copy()
    M = np.uint32(1 << (BITS - 1))
    Q = M
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(3):
            hit = (X[:, i] & Q) != 0
            X[hit, 0] ^= P                       # invert
            t = (X[:, 0] ^ X[:, i]) & P          # exchange
            t[hit] = 0
            X[:, 0] ^= t
            X[:, i] ^= t
        Q = np.uint32(Q >> 1)
    for i in range(1, 3):
        X[:, i] ^= X[:, i - 1]
    t = np.zeros(len(X), np.uint32)
    Q = M
    while Q > 1:
        hit = (X[:, 2] & Q) != 0
        t[hit] ^= np.uint32(Q - 1)
        Q = np.uint32(Q >> 1)
    for i in range(3):
        X[:, i] ^= t
    return interleave(X)

def build(order):
    sp = pts[order].astype(np.float64)
    nleaves = (n + 7) // 8
    pad = nleaves * 8 - n
    spp = np.concatenate([sp, np.full((pad, 3), np.nan)]) if pad else sp
    L = spp.reshape(nleaves, 8, 3)
    depth = 0
    while 4 ** depth < nleaves:
        depth += 1
    blo = np.full((4 ** depth, 3), np.inf); bhi = np.full((4 ** depth, 3), -np.inf)
    blo[:nleaves] = np.nanmin(L, axis=1); bhi[:nleaves] = np.nanmax(L, axis=1)
    levels = {depth: (blo, bhi)}
    for d in range(depth - 1, -1, -1):
        clo, chi = levels[d + 1]
        levels[d] = (clo.reshape(-1, 4, 3).min(1), chi.reshape(-1, 4, 3).max(1))
    return sp, levels, depth, nleaves

def boxd2(lo_, hi_, qq):
    d = np.maximum(np.maximum(lo_[:, None, :] - qq[None, :, :], qq[None, :, :] - hi_[:, None, :]), 0.0)
    return (d * d).sum(-1)

def stats(name, order):
    sp, levels, depth, nleaves = build(order)
    blo, bhi = levels[depth]
    ext = (bhi[:nleaves] - blo[:nleaves])
    vol = np.prod(np.maximum(ext, 1e-12), axis=1)
    rng = np.random.default_rng(1)
    G = n // 64
    leaves_v, exp_v, app_v = [], [], []
    for g in rng.integers(2, G - 2, ngroups):
        qs = sp[g * 64:(g + 1) * 64]
        s0, s1 = g * 8 - 2, g * 8 + 8 + 2
        seedpts = sp[s0 * 8:s1 * 8]
        dd = ((seedpts[None, :, :] - qs[:, None, :]) ** 2).sum(-1)
        dd[dd < 1e-20] = np.inf
        seeded = np.sort(dd, axis=1)[:, K - 1]
        cap = 1.375 * np.median(seeded[1::4])
        tau = np.minimum(seeded, cap) * (1 + 1e-12)
        frontier = np.array([0]); nexp = 0
        for d in range(depth):
            nexp += len(frontier)
            ch = (frontier[:, None] * 4 + np.arange(4)[None, :]).ravel()
            clo, chi = levels[d + 1]
            need = (boxd2(clo[ch], chi[ch], qs) <= tau[None, :]).any(1)
            frontier = ch[need]
        lv = frontier[(frontier < s0) | (frontier >= s1)]
        leaves_v.append(len(lv)); exp_v.append(nexp)
    print("%-8s leaf box: mean diagonal %.5f, mean volume %.3e | per 64-query group: %.1f leaves (p90 %.0f), %.1f expansions (p90 %.0f), %d candidates/query"
          % (name, np.sqrt((ext ** 2).sum(1)).mean(), vol.mean(), np.mean(leaves_v), np.percentile(leaves_v, 90), np.mean(exp_v),
             np.percentile(exp_v, 90), np.mean(leaves_v) * 8 + 96))

stats("morton", np.argsort(interleave(q), kind="stable"))
stats("hilbert", np.argsort(hilbert_key(q), kind="stable"))
