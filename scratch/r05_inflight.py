# throughput of K independent LML evaluations in flight on one GPU (one context + host thread each): python scratch/r05_inflight.py c3 1 2 3 4
import sys, time, threading, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1]
Ks = [int(v) for v in sys.argv[2:]] or [1, 2, 3, 4]
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
steps = 20
# (all contexts are created before any of them runs: streams created after another context has been busy share hardware queues
# with it on ROCm 7, bench.py)
ctxs = [_lib.Context(0) for _ in range(max(Ks))]
for i, c in enumerate(ctxs):
    c.set_data(X, n); c.set_option('timing', 0); c.set_option('profile_gemm', 0)
    c.fit(bench.KID[kernel], params * (1 + 0.01 * i), 0.0, y, err, 2.2e-14)
for K in Ks:
    def run(c, p):
        for _ in range(steps):
            c.fit(bench.KID[kernel], p, 0.0, y, err, 2.2e-14)
    best = 0
    for rep in range(5):
        th = [threading.Thread(target=run, args=(ctxs[i], params * (1 + 0.01 * i))) for i in range(K)]
        if K > 1:
            with _lib.concurrent_evaluations():
                t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t = time.perf_counter() - t0
        else:
            t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t = time.perf_counter() - t0
        best = max(best, K * steps / t)
    fl = bench.flops_fit(N)
    print("%s N=%d, %d in flight: %.1f evals/s, %.1f TFLOP/s (%.1f %% of 78.6)" % (wl, N, K, best, best * fl * 1e-12, 100 * best * fl * 1e-12 / 78.6), flush=True)
