"""Throughput of two contexts / two host threads with and without the concurrency hint, against one thread: python scratch/two_thread_modes.py [wl] [reps]"""
import sys, time, threading, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
cs = [_lib.Context(0), _lib.Context(0)]
for c in cs:
    c.set_data(X, n); c.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
def loop(c, reps):
    for _ in range(reps): c.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
def two(reps):
    th = [threading.Thread(target=loop, args=(c, reps)) for c in cs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0
for rnd in range(2):
    t0 = time.perf_counter(); loop(cs[0], R); t1 = time.perf_counter() - t0
    e0 = [c.edge_count for c in cs]
    tu = two(R)
    e1 = [c.edge_count for c in cs]
    with _lib.concurrent_evaluations():
        th_ = two(R)
    time.sleep(0.15)
    print("%s: one thread %.0f evals/s | two threads un-hinted %.0f evals/s (flag edges raised %s) | hinted %.0f evals/s" % (
        wl, R / t1, 2 * R / tu, [b - a for a, b in zip(e0, e1)], 2 * R / th_))
