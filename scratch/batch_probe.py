"""Aggregate LML evaluations/s when B independent evaluations run concurrently on ONE GPU (B contexts, B host threads)."""
import sys, time, threading, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
kid = bench.KID[kernel]
for B in [int(v) for v in sys.argv[2:]] or [1, 2, 3, 4]:
    ctxs = [_lib.Context(0) for _ in range(B)]
    for c in ctxs:
        c.set_data(X, n)
        c.fit(kid, params, 0.0, y, err, 2.2e-14)
    reps = 12
    out = [None] * B
    def work(i):
        for r in range(reps):
            out[i] = ctxs[i].fit(kid, params * (1.0 + 0.01 * i), 0.0, y, err, 2.2e-14)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(B)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print("%s B=%d: %.1f evaluations/s aggregate (%.3f ms per evaluation amortised), ll[0]=%.6f" % (wl, B, B * reps / dt, dt / (B * reps) * 1e3, out[0][0]))
    del ctxs
