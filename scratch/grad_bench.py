"""Cost of one analytic-gradient evaluation (gpt_fit + gpt_ll_grad) against p+1 plain evaluations (finite differences)."""
import sys, time, warnings, numpy as np
sys.path.insert(0, '/root/repo')
warnings.simplefilter("ignore")
import gptools_amd as g
for N, d in ((4096, 2), (8192, 3), (16384, 2), (16384, 6)):
    rs = np.random.RandomState(1)
    X = rs.rand(N, d); y = np.sin(X.sum(1)) + 0.05 * rs.randn(N)
    k = g.SquaredExponentialKernel(num_dim=d, initial_params=[1.0] + [0.3] * d, param_bounds=[(1e-3, 10.0)] * (d + 1))
    gp = g.GaussianProcess(k, X=X, y=y, err_y=0.05, use_hyper_deriv=True)
    th = np.array([1.0] + [0.3] * d)
    gp.update_hyperparameters(th)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); v, gr = gp.update_hyperparameters(th * (1 + 1e-3 * len(ts))); ts.append(time.perf_counter() - t0)
    gp.use_hyper_deriv = False
    tv = []
    for _ in range(3):
        t0 = time.perf_counter(); gp.update_hyperparameters(th * (1 + 1e-3 * len(tv))); tv.append(time.perf_counter() - t0)
    p = d + 1
    print("N=%5d d=%d (p=%d): value+gradient %.1f ms; value only %.1f ms; finite differences would cost %.1f ms (p+1 evaluations)" % (
        N, d, p, min(ts) * 1e3, min(tv) * 1e3, (p + 1) * min(tv) * 1e3))
