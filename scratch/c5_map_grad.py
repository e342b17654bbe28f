"""BASELINE config 5 (MAP loop, SE, N=16384, d=2, L-BFGS-B, maxiter=50) two ways: finite-difference gradients (the
reference's default: no jac; here the perturbed points are evaluated two at a time) against the analytic gradient on
the device (use_hyper_deriv=True: one factorisation + gpt_ll_grad per objective call).  Prints wall time,
objective calls, factorisations and the optimum of both runs."""
import sys, time, warnings, numpy as np
sys.path.insert(0, '/root/repo')
warnings.simplefilter("ignore")
import gptools_amd as g
from gptools_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rs = np.random.RandomState(1234)
X = rs.rand(N, 2); y = np.sin(X.sum(1)) + 0.05 * rs.randn(N)
fits = [0]
orig_fit = _lib.Context.fit
def counting_fit(self, *a, **kw):
    fits[0] += 1
    return orig_fit(self, *a, **kw)
_lib.Context.fit = counting_fit
for mode in ("fd", "analytic"):
    k = g.SquaredExponentialKernel(num_dim=2, initial_params=[1.0, 0.3, 0.3], param_bounds=[(1e-3, 10.0)] * 3)
    gp = g.GaussianProcess(k, X=X, y=y, err_y=0.05, use_hyper_deriv=(mode == "analytic"))
    gp.update_hyperparameters([1.0, 0.3, 0.3])          # warm-up: allocations, first launches
    fits[0] = 0
    calls = [0]
    inner = gp.update_hyperparameters
    def counted(p, *a, **kw):
        calls[0] += 1
        return inner(p, *a, **kw)
    gp.update_hyperparameters = counted
    t0 = time.perf_counter()
    res, _ = gp.optimize_hyperparameters(method='L-BFGS-B', opt_kwargs={'options': {'maxiter': 50}}, random_starts=0, num_proc=0)
    t = time.perf_counter() - t0
    print("%-8s: %2d iterations, %3d objective calls, %3d factorisations, %.2f s wall, %.1f ms per iteration; params %s, -ll %.6f"
          % (mode, res.nit, calls[0], fits[0], t, t / max(res.nit, 1) * 1e3, np.array2string(res.x, precision=6), res.fun))
