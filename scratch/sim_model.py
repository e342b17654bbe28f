"""Modelled 8-rank time of the partitioned factorisation, from ONE GPU.

Every rank's GPU work is MEASURED: rank r of a W-rank block-cyclic job is replayed on this GPU with its own kernels, streams
and events (gptools_amd.dist, unchanged).  What cannot run here -- the other ranks and the links -- is MODELLED:
  * a foreign chunk (panel k, rows lo..) becomes readable on rank r at  arrive[k, lo] = end of its transfer + latency, where the
    transfer starts when its OWNER has produced it (a time measured in the owner's own replay) and the owner's outgoing link
    is free (FIFO per source, `bw` bytes/s): broadcast = bytes / bw, scatter + all-gather = 2 bytes / (W bw) + one more latency;
  * the receiver pays nothing for the data movement (an ideal DMA engine: every panel buffer exists up front, NBUF = number of
    panels, the foreign ones pre-filled), it only may not read the chunk earlier: a one-wave gate kernel holds the receiving
    queue until the device clock reaches  t0 + arrive[k, lo]  (scratch/simgate.hip).
Production times depend on arrival times and vice versa: the W replays are iterated to a fixed point (arrivals of iteration
i from the productions measured in iteration i - 1, starting from an infinitely fast interconnect).

  python scratch/sim_model.py c4 8 <schedule> <exchange> <latency_us> [chunks] [owner_first] [iterations]
"""
import ctypes, os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from gptools_amd.dist import DistributedLML, HipPanelOps, _StreamEvent

wl, W, sched, exch = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
lat = float(sys.argv[5]) * 1e-3                                     # ms
chunks = tuple(int(v) for v in (sys.argv[6] if len(sys.argv) > 6 else "2,3,8,32").split(","))
ofirst = sys.argv[7] if len(sys.argv) > 7 else "1"
ofirst = "head" if ofirst == "head" else bool(int(ofirst))
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 4
BW = float(os.environ.get("SIM_BW", "153e9"))
NB = int(os.environ.get("SIM_NB", "512"))
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
kid = bench.KID[kernel]
gate = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsimgate.so"))
gate.gate_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
gate.gate_wait.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]


class TimedOps(HipPanelOps):
    def new_event(self):
        return _StreamEvent(self, timing=True)


ops = TimedOps(0)


class Recorder(DistributedLML):
    def _factor_staged(self, k, buf):
        DistributedLML._factor_staged(self, k, buf)
        self.saved[k] = buf[:self.NP - k * self.nb].clone()


class Done(object):
    def __init__(self):
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream())

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class ModelRank(DistributedLML):
    """Rank r of W: own panels computed, foreign panels pre-filled, their chunks gated by the modelled arrival times."""
    def _exchange(self, buf, src, group=None, tag=None):
        k, lo = tag
        nbytes = buf.numel() * buf.element_size()
        st = torch.cuda.current_stream()
        if src == self.rank:
            e = torch.cuda.Event(enable_timing=True)          # production of this chunk is complete HERE on the panel queue
            e.record(st)
            self.produced[(k, lo)] = (e, nbytes)
            return []
        t = self.arrive.get((k, lo))
        if t is not None and t > 0.0:
            gate.gate_wait(ctypes.c_void_p(st.cuda_stream), ctypes.c_void_p(self.t0_dev.data_ptr()), int(t * 1e5))   # ms -> 10 ns ticks
        return [Done()]

    def _allreduce(self, t, op):
        pass

    def _begin(self, *a, **kw):
        with self.ops.queue("main"):
            gate.gate_stamp(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(self.t0_dev.data_ptr()))
            self.t0_ev = torch.cuda.Event(enable_timing=True)
            self.t0_ev.record(torch.cuda.current_stream())
        return DistributedLML._begin(self, *a, **kw)


rec = Recorder(X, n, nb=NB, ops=ops, schedule="bcast", compiled=False)
rec.saved = {}
ll_ref, ld_ref = rec.fit(kid, params, y, err)
panels = rec.saved
nblk = rec.nblk
del rec
torch.cuda.empty_cache()
plans = []
for r in range(W):
    ModelRank.NBUF = nblk
    p = ModelRank(X, n, nb=NB, ops=ops, layout=(r, W), schedule="bcast", exchange="bcast", owner_first=bool(ofirst), compiled=False)      # (the row-chunked schedule left the product in round 4: scratch/attic)
    p.force_collectives = True
    p.t0_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    p.arrive, p.produced = {}, {}
    for k in range(nblk):
        if k % W != r:
            p.P[k][:panels[k].shape[0]].copy_(panels[k])
    plans.append(p)
    # (the rank's own buffers are rewritten by every replay; the foreign ones stay)
torch.cuda.synchronize()
arrive = {}
hist = []


def link_model(prod):
    """Arrival times of ONE source rank's chunks: FIFO on its outgoing link in order of production."""
    out, free = {}, 0.0
    for (k, lo), (tp, nbytes) in sorted(prod.items(), key=lambda kv: kv[1][0]):
        start = max(tp, free)
        if exch == "scatter_gather" and nbytes >= (8 << 20):
            dur_x, extra = 2.0 * nbytes / (W * BW) * 1e3, lat
        else:
            dur_x, extra = nbytes / BW * 1e3, 0.0
        free = start + dur_x
        out[(k, lo)] = start + dur_x + lat + extra
    return out


# Gauss-Seidel over the ranks: the arrivals of rank r's panels are refreshed right after ITS replay, so one sweep carries a
# delay through up to W consecutive panels of the chain (a Jacobi sweep: through one)
for it in range(iters):
    ends, delta = [], 0.0
    for r, p in enumerate(plans):
        p.arrive, p.produced = arrive, {}
        torch.cuda.synchronize()
        ll, ld = p.fit(kid, params, y, err)
        e_end = torch.cuda.Event(enable_timing=True)           # (fit() returns with the rank's queues drained)
        e_end.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        prod = {key: (p.t0_ev.elapsed_time(e), nb_) for key, (e, nb_) in p.produced.items()}
        ends.append(p.t0_ev.elapsed_time(e_end))
        ld_sum = ld if r == 0 else ld_sum + ld               # every rank returns the log-determinant part of ITS panels
        fresh = link_model(prod)
        delta = max([delta] + [abs(fresh[key] - arrive.get(key, 0.0)) for key in fresh])
        arrive.update(fresh)
    assert abs(ld_sum - ld_ref) <= 1e-9 * abs(ld_ref), (ld_sum, ld_ref)
    hist.append((max(ends), delta))
    print("sweep %d: rank end times (ms) %s -> max %.2f; arrivals moved by up to %.2f ms" % (
        it, " ".join("%.1f" % e for e in ends), max(ends), delta), flush=True)
    if it >= 2 and delta < 0.15:
        break
if os.environ.get("SIM_TRACE"):
    # one more replay of one rank with its main-queue marks: when panel i had arrived on the main queue, when its update was done
    r = int(os.environ["SIM_TRACE"])
    p = plans[r]
    p.trace = True
    p.arrive, p.produced = arrive, {}
    p.fit(kid, params, y, err)
    torch.cuda.synchronize()
    st = p.timings.get("steps_ms", [])
    arr_ = {k: t for k, tag, t in st if tag == "arrived"}
    app_ = {k: t for k, tag, t in st if tag == "applied"}
    off = p._t0.elapsed_ms(_StreamEvent.__new__(_StreamEvent)) if False else 0.0
    heads = {k: min(t for (kk, lo), t in arrive.items() if kk == k) for k in range(nblk)}
    print("rank %d main queue: step: head arrival (model) | all chunks arrived on main | update done   (ms)" % r)
    for k in sorted(arr_):
        print("  %2d%s: %7.2f | %7.2f | %7.2f" % (k, "*" if k % W == r else " ", heads.get(k, 0.0), arr_[k], app_.get(k, 0.0)))
if os.environ.get("SIM_DUMP"):
    # the converged chain: per panel, when each chunk was produced by its owner and when it arrived (ms from the start)
    allprod = {}
    for r, p in enumerate(plans):
        for key, (e, nb_) in p.produced.items():
            allprod[key] = (p.t0_ev.elapsed_time(e), nb_)
    for k in range(0, nblk, int(os.environ.get("SIM_DUMP"))):
        row = sorted((lo, tp, arrive[(kk, lo)], nb_) for (kk, lo), (tp, nb_) in allprod.items() if kk == k)
        print("panel %2d (owner %d): " % (k, k % W) + "  ".join("[%d: prod %.2f arr %.2f %dMB]" % (lo // NB, tp, ta, nb_ >> 20) for lo, tp, ta, nb_ in row))
T = hist[-1][0]
print("MODEL %s N=%d W=%d schedule=%s exchange=%s latency=%.0fus bw=%.0fGB/s chunks=%s owner_first=%s nb=%d: %.1f ms -> %.1f TFLOP/s = %.1f %% of %d x 78.6" % (
    wl, N, W, sched, exch, lat * 1e3, BW * 1e-9, ",".join(map(str, chunks)), ofirst, NB, T, bench.flops_fit(N) / T * 1e-9,
    100 * bench.flops_fit(N) / T * 1e-9 / (78.6 * W), W))
