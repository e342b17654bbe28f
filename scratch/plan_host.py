"""Host enqueue time of one rank's evaluation of the 1-D block-cyclic engine: the Python step loop against the compiled plan
(gpt_plan_run), rank `r` of a W-rank layout replayed on one GPU (no collectives: foreign panels are whatever the buffers hold --
the GPU's results are meaningless here, the host's enqueue time is not), and the whole evaluation at world size 1 against gpt_fit.
usage: plan_host.py [N] [W] [r]"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
from gptools_amd.dist import DistributedLML, HipPanelOps
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = int(sys.argv[3]) if len(sys.argv) > 3 else 3
kernel, d = "se", 4
X, n, y, err, params = bench.synth(kernel, N, d, False)
ops = HipPanelOps(0)


class OneRankOfMany(DistributedLML):
    """rank r of a W-rank layout on one GPU, every exchange a no-op (what arrives is whatever the buffer holds)"""
    def _collectives_on(self):
        return False


print("N = %d" % N)
for mode in (False, "python", "native"):
    plan = OneRankOfMany(X, n, nb=512, ops=ops, layout=(r, W), compiled=mode)
    ts, tn = [], []
    for rep in range(4):
        try:
            plan.fit(bench.KID[kernel], params, y, err)
        except np.linalg.LinAlgError:          # (garbage panels: not positive definite is the expected outcome)
            pass
        ts.append(plan.timings.get("host_enqueue_s", 0) * 1e3)
        tn.append(plan.timings.get("native_enqueue_ms", float("nan")))
    print("rank %d of %d, compiled=%-7s: host enqueue of one evaluation %.2f ms (best of 4; ops %s; gpt_plan_run alone %.2f ms)"
          % (r, W, mode, min(ts[1:]), plan.timings.get("plan_ops", "-"), np.nanmin(tn[1:]) if mode == "native" else float("nan")), flush=True)
    del plan
# world size 1: the whole evaluation through the engine against gpt_fit
ctx = _lib.Context(0)
ctx.set_data(X, n)
t_fit = []
for rep in range(3):
    t0 = time.perf_counter(); ref = ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); t_fit.append(time.perf_counter() - t0)
del ctx
for mode in (False, "native"):
    plan = DistributedLML(X, n, nb=512, ops=ops, compiled=mode)
    tt = []
    for rep in range(3):
        t0 = time.perf_counter(); res = plan.fit(bench.KID[kernel], params, y, err); tt.append(time.perf_counter() - t0)
    print("world 1, compiled=%-7s: %.1f ms per evaluation (gpt_fit %.1f ms: ratio %.3f), host enqueue %.2f ms, ll rel diff %.1e"
          % (mode, min(tt) * 1e3, min(t_fit) * 1e3, min(tt) / min(t_fit), plan.timings["host_enqueue_s"] * 1e3, abs(res[0] - ref[0]) / abs(ref[0])), flush=True)
    del plan
