import sys, faulthandler, numpy as np
faulthandler.enable()
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gptools_amd.dist import DistributedLML, HipPanelOps
from test_gpu_parity import c3_inputs
X, n, y = c3_inputs(1500, 3)
p = np.array([1.0, 0.3, 0.3, 0.3]); err = 0.05 * np.ones(1500)
ops = HipPanelOps(0)
for mode in (False, "python", "native"):
    plan = DistributedLML(X, n, nb=256, ops=ops, compiled=mode)
    print(mode, plan.fit(1, p, y, err), flush=True)
    print(mode, plan.fit(1, p, y, err), plan.timings, flush=True)
