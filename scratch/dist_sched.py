"""World-size-1 timing of the block-cyclic engine's schedules on one GPU: python scratch/dist_sched.py c5 c4"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from gptools_amd.dist import DistributedLML
for wl in sys.argv[1:] or ["c5"]:
    kernel, N, d, deriv = bench.WORKLOADS[wl]
    X, n, y, err, params = bench.synth(kernel, N, d, deriv)
    plan = DistributedLML(X, n, nb=512, device=0)
    for sched, cb in (("bcast", (2, 8, 32)), ("pipelined", (2, 8, 32)), ("pipelined", (2, 16)), ("pipelined", (4, 16))):
        plan.schedule, plan.chunk_blocks = sched, cb
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ll, ld = plan.fit(bench.KID[kernel], params, y, err)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("%s N=%d %-9s chunks %-12s: %.2f ms (host enqueue %.2f ms)  ll %.12g" % (
            wl, N, sched, cb, min(ts) * 1e3, plan.timings["host_enqueue_s"] * 1e3, ll), flush=True)
    del plan
