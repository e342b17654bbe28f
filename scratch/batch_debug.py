import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
X, n, y, err, params = bench.synth("se", N, 2, False)
rs = np.random.RandomState(0)
B = 4
P = params[None, :] * (1.0 + 0.1 * rs.rand(B, 3))
c = _lib.Context(0); c.set_data(X, n)
Y = np.tile(y, (B, 1)); nv = np.zeros(B)
ll, ld, info = c.fit_batch(0, P, nv, Y, err, 2.2e-14)
for b in range(B):
    l1, d1 = c.fit(0, P[b], 0.0, y, err, 2.2e-14)
    print(b, info[b], "ll", l1, ll[b], (ll[b] - l1) / abs(l1), "ld", d1, ld[b], (ld[b] - d1) / abs(d1))
for opts in ({"lookahead": 0}, {"fuse_trsm": 0}, {"lookahead": 0, "fuse_trsm": 0}):
    for k, v in opts.items(): c.set_option(k, v)
    print(opts, c.fit(0, P[0], 0.0, y, err, 2.2e-14))
    c.set_option("lookahead", 1); c.set_option("fuse_trsm", 8192)
