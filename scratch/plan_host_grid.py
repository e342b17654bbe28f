"""Host enqueue time of one rank's evaluation of the 2-D block-cyclic engine (GridLML): the Python step loop against the compiled
plan (the Python interpreter of the op list, gpt_plan_run), rank `r` of a P_r x P_c grid replayed on one GPU.  The exchanges are in
the op list (five channels) but nothing is sent: the step loop's `_xbcast` returns no handles, the interpreter's collectives are
no-ops, gpt_plan_run skips the ops of a channel without a communicator -- what arrives is whatever the buffers hold, so the GPU's
results are meaningless here (the three drivers still have to agree on them bit for bit: same kernels, same buffers, same order);
the host's enqueue time is not.  Then the whole evaluation at 1 x 1 against gpt_fit.
usage: plan_host_grid.py [N] [P_r] [P_c] [r]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
from gptools_amd.dist import GridLML, HipPanelOps
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
Pr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
Pc = int(sys.argv[3]) if len(sys.argv) > 3 else 4
r = int(sys.argv[4]) if len(sys.argv) > 4 else 5
kernel, d = "se", 4
X, n, y, err, params = bench.synth(kernel, N, d, False)
ops = HipPanelOps(0)


class Placed(GridLML):
    """rank r of the grid on one GPU: collectives "on" (they are recorded), nothing sent"""
    def _on(self, size):
        return size > 1

    def _xbcast(self, kind, k, buf, src, group, size):
        return GridLML._xbcast(self, kind, k, buf, src, group, size) if self._rec is not None else []

    def _plan_collective(self, opcode, buf, what):
        return []

    def _allreduce(self, t, op):
        pass


print("N = %d, grid %d x %d" % (N, Pr, Pc))
bits = {}
for mode in (False, "python", "native"):
    plan = Placed(X, n, (Pr, Pc), nb=512, ops=ops, layout=r, compiled=mode)
    for t in plan.R + plan.C + plan.H + plan.W + [p_ for ps in plan.piece for p_ in ps]:
        t.fill_(1e-3)
    plan.A.zero_()
    ts, tn = [], []
    for rep in range(4):
        try:
            plan.fit(bench.KID[kernel], params, y, err)
        except np.linalg.LinAlgError:          # (garbage panels: not positive definite is the expected outcome)
            pass
        ts.append(plan.timings.get("host_enqueue_s", 0) * 1e3)
        tn.append(plan.timings.get("native_enqueue_ms", float("nan")))
        if rep == 0:
            torch.cuda.synchronize()
            bits[mode] = (plan.A.view(torch.int64).clone(), plan.red.clone())
    ncoll = sum(1 for o in plan._plans[next(iter(plan._plans))].ops if o[0] == 10) if mode else "-"
    print("rank %d of %d x %d, compiled=%-7s: host enqueue of one evaluation %.2f ms (best of 3; ops %s, of which broadcasts %s; "
          "gpt_plan_run alone %.2f ms)" % (r, Pr, Pc, mode, min(ts[1:]), plan.timings.get("plan_ops", "-"), ncoll,
                                           np.nanmin(tn[1:]) if mode == "native" else float("nan")), flush=True)
    del plan
same = all(torch.equal(bits[False][0], bits[m][0]) for m in ("python", "native"))
print("the rank's matrix after the first evaluation, step loop / interpreter / gpt_plan_run: %s" % ("bit-identical" if same else "DIFFERENT"))
# 1 x 1: the whole evaluation through the engine against gpt_fit
ctx = _lib.Context(0)
ctx.set_data(X, n)
t_fit = []
for rep in range(3):
    t0 = time.perf_counter(); ref = ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); t_fit.append(time.perf_counter() - t0)
del ctx
for mode in (False, "native"):
    plan = GridLML(X, n, (1, 1), nb=512, ops=ops, compiled=mode)
    tt = []
    for rep in range(3):
        t0 = time.perf_counter(); res = plan.fit(bench.KID[kernel], params, y, err); tt.append(time.perf_counter() - t0)
    print("1 x 1, compiled=%-7s: %.1f ms per evaluation (gpt_fit %.1f ms: ratio %.3f), host enqueue %.2f ms, ll rel diff %.1e"
          % (mode, min(tt) * 1e3, min(t_fit) * 1e3, min(tt) / min(t_fit), plan.timings["host_enqueue_s"] * 1e3, abs(res[0] - ref[0]) / abs(ref[0])), flush=True)
    del plan
