"""Many evaluations with flag edges on, with and without random launch delays, two contexts in two threads: looks for hangs
(run under `timeout`) and for any change of the result.  python scratch/stress_flags.py"""
import sys, os, time, threading, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
def run(wl, reps, nctx):
    kernel, N, d, deriv = bench.WORKLOADS[wl]
    X, n, y, err, params = bench.synth(kernel, N, d, deriv)
    ctxs = [_lib.Context(0) for _ in range(nctx)]
    for c in ctxs: c.set_data(X, n)
    ref = ctxs[0].fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
    bad = []
    def loop(c, scale):
        p = params * scale
        r0 = c.fit(bench.KID[kernel], p, 0.0, y, err, 2.2e-14)
        for _ in range(reps):
            r = c.fit(bench.KID[kernel], p, 0.0, y, err, 2.2e-14)
            if r != r0: bad.append((r, r0))
    th = [threading.Thread(target=loop, args=(c, 1.0 + 0.01 * i)) for i, c in enumerate(ctxs)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    print("%s x %d evaluations on %d context(s): %.1f s, edges raised %s, mismatches %d" % (wl, reps, nctx, time.time() - t0, [c.edge_count for c in ctxs], len(bad)))
    assert not bad
import sys
sys.stdout.reconfigure(line_buffering=True)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
run("c2", R, 1)
run("c2", R, 2)
run("c3", R // 4, 1)
run("c3", R // 4, 2)
