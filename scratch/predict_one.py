"""One predict call at the end of the process (for scratch/trace_any.sh): python scratch/predict_one.py <workload> <M> <want>"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl, M, want = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.Context(0)
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
rs = np.random.RandomState(3)
Xs = rs.rand(M, d); ns = np.zeros((M, d), int)
for _ in range(4):
    t0 = time.perf_counter(); r = ctx.predict(Xs, ns, want); t = time.perf_counter() - t0
print("predict M=%d want=%d: %.3f ms wall" % (M, want, t * 1e3))
