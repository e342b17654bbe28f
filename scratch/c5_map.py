"""BASELINE config 5: update_hyperparameters MAP loop, SE kernel, N=16384, d=2, L-BFGS-B, maxiter=50, no jac."""
import sys, time, warnings, numpy as np
sys.path.insert(0, '/root/repo')
warnings.simplefilter("ignore")
import gptools_amd as g
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rs = np.random.RandomState(1234)
X = rs.rand(N, 2); y = np.sin(X.sum(1)) + 0.05 * rs.randn(N)
k = g.SquaredExponentialKernel(num_dim=2, initial_params=[1.0, 0.3, 0.3], param_bounds=[(1e-3, 10.0)] * 3)
gp = g.GaussianProcess(k, X=X, y=y, err_y=0.05)
t0 = time.perf_counter(); v0 = gp.update_hyperparameters([1.0, 0.3, 0.3]); t1 = time.perf_counter()
print("first evaluation %.1f ms, -ll = %.6f" % ((t1 - t0) * 1e3, v0))
nev = [0]
orig = gp.update_hyperparameters
def counted(p, **kw):
    nev[0] += 1
    return orig(p, **kw)
gp.update_hyperparameters = counted
t0 = time.perf_counter()
res, nres = gp.optimize_hyperparameters(method='L-BFGS-B', opt_kwargs={'options': {'maxiter': 50}}, random_starts=0, num_proc=0)
t1 = time.perf_counter()
print("MAP: %d iterations, %d objective evaluations, %.2f s total, %.1f ms per evaluation, params %s, -ll %.6f" % (
    res.nit, nev[0], t1 - t0, (t1 - t0) / nev[0] * 1e3, np.array2string(res.x, precision=6), res.fun))
