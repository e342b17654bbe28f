"""ll_batch throughput at small N: python scratch/batch_grid_bench.py [N] [B]  -> evaluations/s of one gpt_fit per vector, of
two contexts in two threads (round 2's ll_batch) and of gpt_fit_batch."""
import sys, time, threading, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
X, n, y, err, params = bench.synth("se", N, 2, False)
rs = np.random.RandomState(0)
P = params[None, :] * (1.0 + 0.1 * rs.rand(B, 3))
c1, c2 = _lib.Context(0), _lib.Context(0)
for c in (c1, c2):
    c.set_data(X, n)
    c.fit(0, params, 0.0, y, err, 2.2e-14)
def seq():
    return [c1.fit(0, p, 0.0, y, err, 2.2e-14)[0] for p in P]
def two():
    out = [None] * B
    def run(c, idx):
        for i in idx: out[i] = c.fit(0, P[i], 0.0, y, err, 2.2e-14)[0]
    th = [threading.Thread(target=run, args=(c1, range(0, B, 2))), threading.Thread(target=run, args=(c2, range(1, B, 2)))]
    with _lib.concurrent_evaluations():
        for t in th: t.start()
        for t in th: t.join()
    return out
Y = np.tile(y, (B, 1)); nv = np.zeros(B)
def grid():
    return list(c1.fit_batch(0, P, nv, Y, err, 2.2e-14)[0])
ref = seq()
for name, fn in (("one gpt_fit per vector", seq), ("two contexts, two threads", two), ("gpt_fit_batch", grid)):
    fn(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    print("N=%d B=%d %-28s %8.3f ms per batch  %9.0f evals/s  identical=%s" % (N, B, name, min(ts) * 1e3, B / min(ts), r == ref))
