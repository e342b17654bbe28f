import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
ctx = _lib.Context(0); ctx.set_option("timing", 1)
for N in [int(v) for v in sys.argv[1:]]:
    X, n, y, err, params = bench.synth("se", N, 2, False)
    ctx.set_data(X, n)
    res = []
    for nb in (256, 384, 512, 640):
        ctx.set_option("nb_outer", nb)
        best = 1e9
        for _ in range(5):
            ctx.fit(0, params, 0.0, y, err, 2.2e-14); best = min(best, ctx.last_timings()["total"])
        res.append("nb %d: %.3f ms" % (nb, best))
    print("N=%d  " % N + "   ".join(res))
