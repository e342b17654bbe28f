"""A few gpt_fit_batch calls at the end of the process (for scratch/trace_any.sh): python scratch/batch_one.py <N> <B>"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
N, B = int(sys.argv[1]), int(sys.argv[2])
X, n, y, err, params = bench.synth("se", N, 2, False)
rs = np.random.RandomState(0)
P = params[None, :] * (1.0 + 0.1 * rs.rand(B, 3))
c = _lib.Context(0); c.set_data(X, n)
Y = np.tile(y, (B, 1)); nv = np.zeros(B)
for _ in range(4): r = c.fit_batch(0, P, nv, Y, err, 2.2e-14)
print(r[0][:2])
