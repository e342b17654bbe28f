import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
Ms = [int(v) for v in sys.argv[2:]] or [64, 1024, 4096]
ctx = _lib.Context(0)
import os
if os.environ.get('GPT_TILE'): ctx.set_option('tile', int(os.environ['GPT_TILE']))
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
rs = np.random.RandomState(3)
for M in Ms:
    Xs = rs.rand(M, d); ns = np.zeros((M, d), int)
    for want in (0, 1, 2):
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); r = ctx.predict(Xs, ns, want); ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        fl = 2.0 * N * M + (0 if want == 0 else N * N * M + (N * M if want == 1 else N * M * M))
        print("%s N=%d M=%5d want=%d: %.3f ms  (%.1f TFLOP/s on N^2 M [+ N M^2])" % (wl, N, M, want, t * 1e3, fl / t * 1e-12))
