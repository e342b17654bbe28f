import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib as L
N = int(sys.argv[1]); la = int(sys.argv[2]); nb = int(sys.argv[3])
ctx = L.Context(0)
ctx.set_option("timing", 1); ctx.set_option("lookahead", la); ctx.set_option("nb_outer", nb)
rs = np.random.RandomState(1234)
d = 3
X = rs.rand(N, d); n = np.zeros((N, d), int)
for i in range(3*N//4, N): n[i, i % d] = 1
y = np.sin(X.sum(1)) + 0.05*rs.randn(N)
p = np.concatenate(([1.0], 0.3*np.ones(d))); err = 0.05*np.ones(N)
ctx.set_data(X, n)
for it in range(3):
    ll, ld = ctx.fit(1, p, 0.0, y, err, 2.2e-14)
print(ctx.last_timings(), ll)
