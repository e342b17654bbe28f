import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl, M, sk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.Context(0)
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
rs = np.random.RandomState(3)
Xs = rs.rand(M, d); ns = np.zeros((M, d), int)
ctx.set_option('splitk', 0); ref = ctx.predict(Xs, ns, 2)
ctx.set_option('splitk', sk)
out = []
for want in (1, 2):
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); r = ctx.predict(Xs, ns, want); ts.append(time.perf_counter() - t0)
    out.append(min(ts[1:]) * 1e3)
print("M=%d std %.3f ms cov %.3f ms  (max |cov - unsplit| %.1e)" % (M, out[0], out[1], np.abs(r[2] - ref[2]).max()))
