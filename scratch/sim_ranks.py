"""Compute-side bound of the 8-GPU factorisation, measured on ONE GPU: replay the schedule of one rank of a W-rank
block-cyclic job with every panel that rank does not own injected (device-to-device copy of the true panel, i.e. an
infinitely fast interconnect).  What is timed is exactly the GPU work rank r would do: its K columns, its panels, its
staircase updates.  max_r T_r is a lower bound of the W-GPU time; the gap to the real time is broadcast + chain stalls.

  python scratch/sim_ranks.py c4 8 [pipelined|bcast] [ranks ...]       # workload, world size, schedule
"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from gptools_amd.dist import DistributedLML, HipPanelOps

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sched = sys.argv[3] if len(sys.argv) > 3 else "pipelined"
ranks = [int(v) for v in sys.argv[4:]] or list(range(W))
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
kid = bench.KID[kernel]
ops = HipPanelOps(0)
if os.environ.get("SIM_FUSE") is not None:
    ops.ctx_panel.set_option("fuse_trsm", int(os.environ["SIM_FUSE"]))


class Recorder(DistributedLML):
    """World size 1, keeps a copy of every factored panel."""
    def _factor_staged(self, k, buf):
        DistributedLML._factor_staged(self, k, buf)
        self.saved[k] = buf[:self.NP - k * self.nb].clone()


class CopyDone(object):
    """Work-handle stand-in: wait() orders the current stream behind the injected copy."""
    def __init__(self):
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream())

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class OneRank(DistributedLML):
    """Rank r of W; foreign panels are copied in from the recorder."""
    def _exchange(self, buf, src, group=None, tag=None):
        k, lo = tag
        if src != self.rank:
            # (a receiver posts its side on the "recv" queue; here that is where the copy runs)
            buf.copy_(self.panels[k][lo:lo + buf.shape[0]])
            return [CopyDone()]
        return []

    def _allreduce(self, t, op):
        pass


rec = Recorder(X, n, nb=int(os.environ.get("SIM_NB", "512")), ops=ops, schedule="bcast")
rec.saved = {}
ll_ref, ld_ref = rec.fit(kid, params, y, err)
t0 = time.perf_counter(); rec.saved = {}; rec.fit(kid, params, y, err); torch.cuda.synchronize(); t1 = time.perf_counter()
print("world 1 (block-cyclic engine): %.1f ms" % ((t1 - t0) * 1e3))
panels = rec.saved
tot = []
for r in ranks:
    plan = OneRank(X, n, nb=int(os.environ.get("SIM_NB", "512")), ops=ops, layout=(r, W), schedule="bcast",      # (the row-chunked schedule left the product in round 4)
                   owner_first=bool(int(os.environ.get("SIM_OWNER_FIRST", "1"))),
                   inv_trsm=bool(int(os.environ.get("SIM_INV_TRSM", "1"))))
    plan.force_collectives = True          # takes the world > 1 code path (scalar reduction included, as a no-op)
    plan.panels = panels
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ll, ld = plan.fit(kid, params, y, err)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    tot.append(min(ts))
    print("rank %d of %d: %.1f ms (owns %d block columns)" % (r, W, min(ts) * 1e3, plan.nloc))
    del plan
T = max(tot)
print("%s N=%d, %d ranks, schedule %s: max over ranks %.1f ms -> compute-side bound %.1f TFLOP/s = %.1f %% of %d x 78.6" % (
    wl, N, W, sched, T * 1e3, bench.flops_fit(N) / T * 1e-12, 100 * bench.flops_fit(N) / T * 1e-12 / (78.6 * W), W))
