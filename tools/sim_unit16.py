"""CPU model: the walk of k_knn with the tree built over leaves of 8 points (as shipped) against units of 16 points (two leaf records
under one box; one publish per unit, steps of 8 needing lanes x 8 points per record): events per 64-query group and the instruction
budget they imply (expansion 70 V / 50 S, leaf overhead 22 / 35, step 14 / 11, dense record 93 / 30).
usage: python tools/sim_unit16.py [uniform|clustered]"""
import sys, numpy as np
exec(open('tools/sim_wave_box.py').read().split("def walk(qq, tau):")[0])
def build(unit):
    nu = (n + unit - 1) // unit
    pad = nu * unit - n
    P = np.concatenate([sp, np.full((pad, 3), np.nan)]) if pad else sp
    Lu = P.reshape(nu, unit, 3)
    dep = 0
    while 4 ** dep < nu: dep += 1
    blo = np.full((4 ** dep, 3), np.inf); bhi = np.full((4 ** dep, 3), -np.inf)
    blo[:nu] = np.nanmin(Lu, 1); bhi[:nu] = np.nanmax(Lu, 1)
    lv = {dep: (blo, bhi)}
    for d in range(dep - 1, -1, -1):
        clo, chi = lv[d + 1]; lv[d] = (clo.reshape(-1, 4, 3).min(1), chi.reshape(-1, 4, 3).max(1))
    return dep, lv
def walk_events(dep, lv, unit, qq, tau, s0, s1):
    frontier = np.array([0]); exp = 0
    for d in range(dep):
        exp += len(frontier)
        ch = (frontier[:, None] * 4 + np.arange(4)[None, :]).ravel()
        clo, chi = lv[d + 1]
        ok = np.isfinite(clo[ch, 0])
        nd = (boxd2(clo[ch], chi[ch], qq) <= tau[None, :]) & ok[:, None]
        need = nd.any(1)
        if d == dep - 1:
            ms = nd.sum(1)[need]; u = ch[need]
            keep = (u < s0) | (u >= s1)
            return exp, ms[keep]
        frontier = ch[need]
rng = np.random.default_rng(1); G = n // 64
tree = cKDTree(sp)
res = {}
for unit in (8, 16):
    dep, lv = build(unit)
    E = U = ST = DENSE = 0; V = S = 0
    groups = rng.integers(2, G - 2, 120) if unit == 8 else groups
    for g in groups:
        qs = sp[g * 64:(g + 1) * 64]
        d, _ = tree.query(qs, k=K + 1); kth2 = d[:, K] ** 2
        tau = np.minimum(kth2, 1.25 * np.median(kth2)) * 1.15
        per = 64 // unit
        exp, ms = walk_events(dep, lv, unit, qs, tau, g * per - 2 * 8 // unit, g * per + per + 2 * 8 // unit)
        recs = unit // 8
        packed = ms <= 24
        steps = (np.ceil(ms[packed] / 8) * recs).sum()
        dense = (~packed).sum() * recs
        E += exp; U += len(ms); ST += steps; DENSE += dense
        V += exp * 70 + packed.sum() * 22 + steps * 14 + dense * 93
        S += exp * 50 + packed.sum() * 35 + steps * 11 + dense * 30
    k = len(groups)
    res[unit] = dict(depth=dep, expansions=E / k, walk_units=U / k, steps=ST / k, dense_records=DENSE / k, V=V / k, S=S / k)
    print(unit, {a: round(b, 1) for a, b in res[unit].items()})
print("unit 16 / unit 8:  walk V %.2f  S %.2f" % (res[16]['V'] / res[8]['V'], res[16]['S'] / res[8]['S']))
