"""CPU model: how many of a group's 64 lanes need each leaf its walk visits (k = 15, cap 1.25 x median, 2 seed leaves either side left out)?
Decided the threshold range of k_knn's point-per-lane leaf form (pcpx_query.hip: sparse_leaf).  Builds on tools/sim_wave_box.py's tree.
usage: python tools/sim_leaf_need.py   ->  1 M uniform points: 62.7 walk leaves per group, m <= 1: 16 %, <= 2: 28 %, <= 3: 37 %, <= 4: 45 %, <= 8: 68 %, mean 7.0"""
import sys, numpy as np, importlib
sys.path.insert(0, ".")
exec(open('tools/sim_wave_box.py').read().split("def walk(qq, tau):")[0])
def walk_m(qq, tau, s0, s1):
    frontier = np.array([0])
    for d in range(depth):
        ch = (frontier[:, None] * 4 + np.arange(4)[None, :]).ravel()
        clo, chi = levels[d + 1]
        ok = np.isfinite(clo[ch, 0])
        nd = (boxd2(clo[ch], chi[ch], qq) <= tau[None, :]) & ok[:, None]
        need = nd.any(1)
        if d == depth - 1:
            ms = nd.sum(1)[need]
            lv = ch[need]
            keep = (lv < s0) | (lv >= s1)
            return ms[keep]
        frontier = ch[need]
rng = np.random.default_rng(1); G = n // 64
tree = cKDTree(sp)
hist = np.zeros(65, int)
for g in rng.integers(2, G - 2, 150):
    qs = sp[g * 64:(g + 1) * 64]
    d, _ = tree.query(qs, k=K + 1); kth2 = d[:, K] ** 2
    tau = np.minimum(kth2 * 1.0, 1.25 * np.median(kth2)) * 1.15
    ms = walk_m(qs, tau, g * 8 - 2, g * 8 + 10)
    for m in ms: hist[m] += 1
tot = hist.sum()
print("leaves/group", tot / 150)
c = np.cumsum(hist) / tot
for m in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32): print("m <=", m, round(float(c[m]), 3))
print("mean m", (np.arange(65) * hist).sum() / tot)
