"""ll_batch throughput for the models gpt_fit_batch_terms added in round 4: a linear TRANSFORM and a PRODUCT term,
python scratch/batch_terms_bench.py [Nx] [B] -> evaluations/s of one gpt_fit_terms per vector, of two contexts in two threads
(what such models got before), and of gpt_fit_batch_terms; identical = bit-identical ll."""
import sys, time, threading, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rs = np.random.RandomState(0)
d = 2
X = rs.rand(Nx, d)
n = np.zeros((Nx, d), dtype=np.int32)
f = np.sin(3 * X.sum(1))
for case in ("transform (Ny = Nx / 2), SE", "k1 * k2 + k3 (SE * Matern52 + RQ)"):
    with_T = case.startswith("transform")
    Ny = Nx // 2 if with_T else Nx
    T = rs.rand(Ny, Nx) / Nx if with_T else None
    y = (T.dot(f) if with_T else f) + 1e-2 * rs.randn(Ny)
    err = np.full(Ny, 0.02)

    def terms(b):
        s = 1.0 + 0.002 * b
        if with_T:
            return [(0, np.array([1.1 * s, 0.4, 0.6]))]
        return [(0, np.array([1.1 * s, 0.4, 0.6]), 1, np.array([0.9, 1.5 * s, 2.0])), (4, np.array([0.3, 1.7, 0.8 * s, 0.9]))]
    cs = [_lib.Context(0), _lib.Context(0)]
    for c in cs:
        c.set_data(X, n)
        if with_T:
            c.set_T(T)
        c.fit_terms(terms(0), 1e-3, y, err, 2.2e-14)
    nv = np.full(B, 1e-3)
    Y = np.tile(y, (B, 1))

    def seq():
        return [cs[0].fit_terms(terms(b), nv[b], Y[b], err, 2.2e-14)[0] for b in range(B)]

    def two():
        out = [None] * B
        def run(c, idx):
            for i in idx:
                out[i] = c.fit_terms(terms(i), nv[i], Y[i], err, 2.2e-14)[0]
        th = [threading.Thread(target=run, args=(cs[0], range(0, B, 2))), threading.Thread(target=run, args=(cs[1], range(1, B, 2)))]
        with _lib.concurrent_evaluations():
            for t in th: t.start()
            for t in th: t.join()
        return out

    def grid():
        return list(cs[0].fit_batch_terms([terms(b) for b in range(B)], nv, Y, err, 2.2e-14)[0])
    ref = seq()
    for name, fn in (("one gpt_fit_terms per vector", seq), ("two contexts, two threads", two), ("gpt_fit_batch_terms", grid)):
        fn(); ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        print("%-36s Nx=%d B=%d %-30s %8.3f ms per batch  %9.0f evals/s  identical=%s" % (case, Nx, B, name, min(ts) * 1e3, B / min(ts), r == ref), flush=True)
    for c in cs:
        c.close()
