"""Where a grid rank's time goes (one GPU, no interconnect model): GridLML (1, 1) against DistributedLML at world size 1, and
grid position r of a Pr x Pc job replayed with every foreign piece present at once, with the main queue's per-step marks.
  python scratch/grid_diag.py c4 4 2 0"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ.setdefault("SIM_NB", "512")
import bench
from gptools_amd.dist import DistributedLML, GridLML, HipPanelOps
wl, Pr, Pc, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
kid = bench.KID[kernel]
ops = HipPanelOps(0)
for name, mk in (("DistributedLML world 1", lambda: DistributedLML(X, n, nb=512, ops=ops)),
                 ("GridLML 1x1", lambda: GridLML(X, n, (1, 1), nb=512, ops=ops))):
    p = mk()
    p.fit(kid, params, y, err)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        ll = p.fit(kid, params, y, err)
    torch.cuda.synchronize()
    print("%s: %.1f ms per evaluation, ll %.6f" % (name, (time.perf_counter() - t) / 3 * 1e3, ll[0]), flush=True)
    if name.startswith("Grid"):
        p.trace = True
        p.fit(kid, params, y, err)
        st = p.timings["steps_ms"]
        arr = {k: t_ for k, tag, t_ in st if tag == "arrived"}
        app = {k: t_ for k, tag, t_ in st if tag == "applied"}
        print("  step: arrived | applied | update ms   (1x1)")
        for k in sorted(arr)[::4]:
            print("  %2d: %7.2f %7.2f %6.2f" % (k, arr[k], app[k], app[k] - arr[k]))
    del p
    torch.cuda.empty_cache()


# ---- host cost of one grid rank's step loop: rank r of Pr x Pc, every foreign piece "there" (no data moved: garbage in, the
# kernels' durations do not depend on values), cProfile over one evaluation
class NoComm(GridLML):
    class H(object):
        def wait(self):
            pass

    def _xbcast(self, kind, k, buf, src, group, size):
        if size <= 1 or src == (self.pr, self.pc) or buf.numel() == 0:
            return []
        return [NoComm.H()]

    def _allreduce(self, t, op):
        pass


import cProfile, pstats
p = NoComm(X, n, (Pr, Pc), nb=512, ops=ops, layout=r)
for rep in range(2):
    try:
        p.fit(kid, params, y, err)
    except np.linalg.LinAlgError:
        pass
torch.cuda.synchronize()
pr_ = cProfile.Profile()
t = time.perf_counter()
pr_.enable()
try:
    p.fit(kid, params, y, err)
except np.linalg.LinAlgError:
    pass
pr_.disable()
torch.cuda.synchronize()
print("grid %dx%d rank %d, no communication: %.1f ms per evaluation, host enqueue %.1f ms" % (
    Pr, Pc, r, (time.perf_counter() - t) * 1e3, p.timings["host_enqueue_s"] * 1e3), flush=True)
pstats.Stats(pr_).sort_stats("tottime").print_stats(18)
p.trace = True
try:
    p.fit(kid, params, y, err)
except np.linalg.LinAlgError:
    pass
st = p.timings.get("steps_ms", [])
arr_ = {k: t for k, tag, t in st if tag == "arrived"}
app_ = {k: t for k, tag, t in st if tag == "applied"}
other = {}
for k, tag, t in st:
    if tag not in ("arrived", "applied"):
        other.setdefault(k, []).append("%s %.2f" % (tag, t))
print("step: arrived | applied | marks")
for k in sorted(arr_):
    if k < 6 or 28 <= k < 40:
        print("  %2d: %7.2f | %7.2f | %s" % (k, arr_[k], app_.get(k, 0.0), "  ".join(other.get(k, []))))
sys.stdout.flush()
os._exit(0)
