import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from gptools_amd.dist import DistributedLML
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
plan = DistributedLML(X, n, nb=512, device=0)
for _ in range(3):
    t0 = time.perf_counter(); plan.fit(bench.KID[kernel], params, y, err); t1 = time.perf_counter()
    print("%s: fit %.1f ms, host enqueue %.1f ms" % (wl, (t1 - t0) * 1e3, plan.timings["host_enqueue_s"] * 1e3))
