import sys, time, threading, numpy as np
sys.path.insert(0, '/root/repo')
mode = sys.argv[1]
if mode in ("torch", "torchsync"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
from gptools_amd import _lib
import bench
kernel, N, d, deriv = bench.WORKLOADS["c3"]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
kid = bench.KID[kernel]
ctxs = [_lib.Context(0) for _ in range(2)]
if mode == "lateuse":                  # both created up front, the second one first USED after the first one's loop
    ctxs[0].set_data(X, n)
    for _ in range(14): ctxs[0].fit(kid, params, 0.0, y, err, 2.2e-14)
    ctxs[1].set_data(X, n); ctxs[1].fit(kid, params, 0.0, y, err, 2.2e-14)
elif mode == "latecreate":             # like `late`, nothing destroyed
    ctxs = [ctxs[0]]
    ctxs[0].set_data(X, n)
    for _ in range(14): ctxs[0].fit(kid, params, 0.0, y, err, 2.2e-14)
    ctxs.append(_lib.Context(0)); ctxs[1].set_data(X, n); ctxs[1].fit(kid, params, 0.0, y, err, 2.2e-14)
else:
  for c in ctxs:
    c.set_data(X, n); c.fit(kid, params, 0.0, y, err, 2.2e-14)
if mode == "warm":                      # like bench: a single-context loop first
    for _ in range(13): ctxs[0].fit(kid, params, 0.0, y, err, 2.2e-14)
if mode == "late":
    for _ in range(13): ctxs[0].fit(kid, params, 0.0, y, err, 2.2e-14)
    ctxs[1] = _lib.Context(0); ctxs[1].set_data(X, n); ctxs[1].fit(kid, params * 1.01, 0.0, y, err, 2.2e-14)
if mode == "opts":
    ctxs[0].set_option("timing", 1); ctxs[0].set_option("profile_gemm", 1)
    for _ in range(13): ctxs[0].fit(kid, params, 0.0, y, err, 2.2e-14)
    ctxs[0].gemm_profile_read(); ctxs[0].set_option("timing", 0); ctxs[0].set_option("profile_gemm", 0)
reps = 12
def work(i):
    for r in range(reps):
        ctxs[i].fit(kid, params * (1.0 + 0.01 * i), 0.0, y, err, 2.2e-14)
ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
if mode == "torchsync": torch.cuda.synchronize()
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
if mode == "torchsync": torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(mode, "%.1f evaluations/s" % (2 * reps / dt))
