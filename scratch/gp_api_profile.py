"""Where the host time of GaussianProcess.update_hyperparameters goes (cProfile, C3; the device work is ~4.4 ms per call)."""
import sys, time, cProfile, pstats, io, warnings, numpy as np
sys.path.insert(0, "/root/repo")
import gptools_amd as g
import bench
kernel, N, d, deriv = bench.WORKLOADS["c3"]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    gp = g.GaussianProcess(g.Matern52Kernel(num_dim=d, initial_params=params, param_bounds=[(1e-3, 10.0)] * (d + 1)), X=X, y=y, err_y=err, n=n)
for _ in range(5): gp.update_hyperparameters(params)
t0 = time.perf_counter()
for _ in range(50): gp.update_hyperparameters(params)
print("%.3f ms per call" % ((time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): gp.update_hyperparameters(params)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
