# alpha after a fit: rounds 2-4's form (binv_launches=1), the new one, and the new one enqueued by the fit itself (eager_alpha)
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
ctx = _lib.Context(0)
kernel, N, d, deriv = bench.WORKLOADS[wl]
if len(sys.argv) > 2: N = int(sys.argv[2])
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
res = {}
for name, opts in (("old", {"binv_launches": 1, "eager_alpha": 0}), ("new", {"binv_launches": 0, "eager_alpha": 0}),
                   ("eager", {"binv_launches": 0, "eager_alpha": 1}), ("old", {"binv_launches": 1, "eager_alpha": 0}),
                   ("new", {"binv_launches": 0, "eager_alpha": 0}), ("eager", {"binv_launches": 0, "eager_alpha": 1})):
    for k, v in opts.items(): ctx.set_option(k, v)
    tf, ta = [], []
    for _ in range(8):
        t0 = time.perf_counter(); ll = ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); t1 = time.perf_counter()
        a = ctx.get_alpha(N); t2 = time.perf_counter()
        tf.append(t1 - t0); ta.append(t2 - t0)
    res.setdefault(name, []).append(a.copy())
    print("%-6s fit %.3f ms, fit + alpha %.3f ms (min of 8)" % (name, 1e3 * min(tf), 1e3 * min(ta)), flush=True)
ctx.set_option("eager_alpha", 0)
tf = []
for _ in range(8):
    t0 = time.perf_counter(); ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); tf.append(time.perf_counter() - t0)
print("fit alone %.3f ms" % (1e3 * min(tf)))
print("alpha new vs old: %.3e, eager vs new: %.3e (max |alpha| %.3e); repeat: %s" % (
    np.abs(res["new"][0] - res["old"][0]).max(), np.abs(res["eager"][0] - res["new"][0]).max(), np.abs(res["old"][0]).max(),
    all(np.array_equal(res[k][0], res[k][1]) for k in res)))
