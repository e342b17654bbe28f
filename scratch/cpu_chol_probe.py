"""Host-side probe (GPU box): how fast is scipy.linalg.cholesky at each OpenBLAS thread count, and how much RAM is there?
Decides the thread count used by the full-size oracle comparisons (tests) and by bench.py's cpu_baseline."""
import os, sys, time
import numpy as np
import scipy.linalg
from threadpoolctl import threadpool_limits, threadpool_info
print("cores", len(os.sched_getaffinity(0)))
print(os.popen("free -g | head -2").read())
for i in threadpool_info():
    print(i)
rs = np.random.RandomState(0)
for N in (8192, 16384):
    A = rs.rand(N, 64)
    K = A.dot(A.T) + N * np.eye(N)
    for nt in ((16, 32, 64, 128, 256) if N == 8192 else (32, 64, 128)):
        with threadpool_limits(limits=nt):
            t0 = time.perf_counter()
            L = scipy.linalg.cholesky(K, lower=True, check_finite=False)
            t = time.perf_counter() - t0
        print("N=%d threads=%d  %.2f s  %.0f GFLOP/s" % (N, nt, t, N ** 3 / 3 / t * 1e-9), flush=True)
