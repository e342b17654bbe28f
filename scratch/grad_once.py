import sys, warnings, numpy as np
sys.path.insert(0, '/root/repo')
warnings.simplefilter("ignore")
import gptools_amd as g
N, d = 8192, 3
rs = np.random.RandomState(1)
X = rs.rand(N, d); y = np.sin(X.sum(1)) + 0.05 * rs.randn(N)
k = g.SquaredExponentialKernel(num_dim=d, initial_params=[1.0] + [0.3] * d, param_bounds=[(1e-3, 10.0)] * (d + 1))
gp = g.GaussianProcess(k, X=X, y=y, err_y=0.05, use_hyper_deriv=True)
th = np.array([1.0] + [0.3] * d)
for i in range(3):
    gp.update_hyperparameters(th * (1 + 1e-3 * i))
