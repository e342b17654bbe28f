"""N beyond the BASELINE sizes: single-context path vs the block-cyclic path (independent schedules) on one GPU."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
from gptools_amd.dist import DistributedLML
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
X, n, y, err, params = bench.synth("se", N, 4, False)
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
ctx.set_data(X, n)
for _ in range(2):
    ll, ld = ctx.fit(0, params, 0.0, y, err, 2.2e-14)
t = ctx.last_timings()["total"]
print("single path  N=%d: %.1f ms  %.1f TF/s  ll %.10e logdet/2 %.10e" % (N, t, bench.flops_fit(N) / t * 1e-9, ll, ld))
a = ctx.get_alpha(N)
print("alpha finite:", bool(np.isfinite(a).all()), " |alpha|max %.3e" % np.abs(a).max())
del ctx
plan = DistributedLML(X, n, nb=512, device=0)
t0 = time.perf_counter(); ll2, ld2 = plan.fit(0, params, y, err); t1 = time.perf_counter()
print("block-cyclic N=%d: %.1f ms  ll %.10e logdet/2 %.10e  rel diff ll %.2e logdet %.2e" % (N, (t1 - t0) * 1e3, ll2, ld2, abs(ll - ll2) / abs(ll), abs(ld - ld2) / abs(ld)))
