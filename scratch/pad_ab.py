"""option defer_pad (the pad leaf beside the substitution of an eager alpha) on / off, same process: time, bits of ll / alpha / L"""
import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
import bench
for wl in sys.argv[1:] or ["c3", "c2"]:
    kernel, N, d, deriv = bench.WORKLOADS[wl]
    X, n, y, err, params = bench.synth(kernel, N, d, deriv)
    ctx = _lib.Context(0); ctx.set_data(X, n)
    ctx.set_option("eager_alpha", 1)
    OPT = os.environ.get("AB_OPT", "defer_pad")
    V0, V1 = (int(v) for v in os.environ.get("AB_VALS", "0,1").split(","))
    def run(defer, reps=(40 if N <= 8192 else 6 if N <= 16384 else 2)):
        ctx.set_option(OPT, defer)
        for _ in range(5): ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); ctx.get_alpha(N)
        ts = []
        for r in range(8):
            t0 = time.perf_counter()
            for _ in range(reps):
                res = ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
                a = ctx.get_alpha(N)
            ts.append((time.perf_counter() - t0) / reps)
        return np.median(ts) * 1e3, min(ts) * 1e3, res, a, ctx.get_L(N)
    for rep in range(3):
        m0, b0, r0, a0, L0 = run(V0); m1, b1, r1, a1, L1 = run(V1)
        print(("%s " + OPT + " " + str(V0) + ": %.4f (min %.4f)  " + str(V1) + ": %.4f (min %.4f) ms  gain %.1f us  ll equal %s alpha equal %s (max diff %.1e) L equal %s")
              % (wl, m0, b0, m1, b1, (m0 - m1) * 1e3, r0 == r1, np.array_equal(a0, a1), np.abs(a0 - a1).max(), np.array_equal(L0, L1)), flush=True)
    del ctx
