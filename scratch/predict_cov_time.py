"""predict(cov) at M points: wall per call over several calls (the pinned result buffers are pooled): python scratch/predict_cov_time.py c3 4096"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl, M = sys.argv[1], int(sys.argv[2])
ctx = _lib.Context(0)
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
rs = np.random.RandomState(3)
Xs = rs.rand(M, d); ns = np.zeros((M, d), int)
for want in (1, 2):
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); r = ctx.predict(Xs, ns, want); ts.append((time.perf_counter() - t0) * 1e3)
    print("want=%d:" % want, " ".join("%.2f" % t for t in ts))
# pageable destination through the C ABI directly
import ctypes as C
mean = np.empty(M); std = np.empty(M); cov = np.empty((M, M))
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    _lib.check(ctx._lib.gpt_predict(ctx.handle, _lib.dptr(_lib.f64(Xs)), _lib.iptr(_lib.i32(ns)), M, 2, None, None, _lib.dptr(mean), _lib.dptr(std), _lib.dptr(cov)))
    ts.append((time.perf_counter() - t0) * 1e3)
print("pageable cov:", " ".join("%.2f" % t for t in ts), " max|cov - pinned| %.1e" % np.abs(cov - r[2]).max())
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    _lib.check(ctx._lib.gpt_predict(ctx.handle, _lib.dptr(_lib.f64(Xs)), _lib.iptr(_lib.i32(ns)), M, 2, None, None, _lib.dptr(mean), _lib.dptr(std), None))
    ts.append((time.perf_counter() - t0) * 1e3)
print("device-resident cov:", " ".join("%.2f" % t for t in ts), " max|std - sqrt(diag)| %.1e" % np.abs(std - np.sqrt(np.diag(cov))).max())
print("symmetric:", np.abs(cov - cov.T).max())
