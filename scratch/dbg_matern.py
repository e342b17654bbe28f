import sys, numpy as np
sys.path.insert(0, '/root/repo')
import gptools_amd as g
from oracle import oracle as O
d = 1
rs = np.random.RandomState(100 + d)
for nu in (0.8, 2.5, 3.0, 6.3):
    M = 300
    p = np.concatenate(([1.1, nu], 0.3 + 0.5 * rs.rand(d)))
    k = g.MaternKernel(num_dim=d, initial_params=list(p), param_bounds=[(0.0, 1e3)] * (d + 2))
    Xi, Xj = rs.rand(M, d), rs.rand(M, d)
    Xj[:20] = Xi[:20]
    Xj[20:40] = Xi[20:40] + 1e-3 * (rs.rand(20, d) - 0.5)
    ni, nj = np.zeros((M, d), int), np.zeros((M, d), int)
    for m in range(M):
        for _ in range(rs.randint(0, 3 if 20 <= m < 40 else 9)):
            (ni if rs.rand() < 0.5 else nj)[m, rs.randint(d)] += 1
    got = k(Xi, Xj, ni, nj); ref = O.kpairs("matern", p, Xi, Xj, ni, nj)
    fin = np.isfinite(ref) & np.isfinite(got)
    rel = np.zeros(M); rel[fin] = np.abs(got[fin]-ref[fin])/np.maximum(np.abs(ref[fin]), 1e-300)
    bad = np.where((rel > 1e-9) | (np.isnan(got) != np.isnan(ref)))[0]
    print("nu", nu, "bad", len(bad))
    for i in bad[:8]:
        y = 2*nu*(((Xi[i]-Xj[i])/p[2:])**2).sum()
        print("  row", i, "n", (ni[i]+nj[i]), "y %.4e" % y, "z %.4f" % np.sqrt(y), "got", got[i], "ref", ref[i])
