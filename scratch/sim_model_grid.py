"""Modelled W-rank time of the 2-D block-cyclic factorisation (gptools_amd.dist.GridLML), from ONE GPU.

Same idea as scratch/sim_model.py (1-D): every rank's GPU work is MEASURED -- grid position (pr, pc) of a P_r x P_c job is
replayed on this GPU with its own kernels, queues and events, gptools_amd.dist unchanged -- and what cannot run here, the other
ranks and the links, is MODELLED:
  * a broadcast produced by another rank becomes readable here at  arrive = max(production, link free) + bytes / bw + latency,
    production = the time the OWNER's replay recorded behind the kernels that produce it; one FIFO per directed link
    (source -> destination: xGMI is point to point, a root feeds its peers over different links at the same time);
  * the receiving side: the payload (taken from a complete factor computed first) is copied into the buffer the broadcast would
    fill on a delivery stream of its own, as soon as that buffer may be written -- HBM traffic beside the compute, as a receive
    is; every queue that waits for the broadcast waits for that copy and is then held by a one-wave gate kernel of its own until
    the device clock reaches the arrival time.  (Versions that did not work: the copy on the waiting queue at arrival time -- 15 ms
    of strided copies per rank on the critical queues; one buffer slot per panel, all payloads up front -- every step touches
    fresh hundreds of MB, replays of 100+ ms; one gate on the first waiter with an event for the others -- ties a rank's queues
    together, the sweeps ratchet upwards.)
Production depends on arrivals and vice versa: the W replays are swept (Gauss-Seidel over the ranks) to the fixed point.
NOT modelled: RCCL's own launch overhead and CU usage, contention between concurrent transfers on the fabric, host jitter.

  python scratch/sim_model_grid.py c4 <Pr> <Pc> <latency_us> [sweeps]        env: SIM_BW (bytes/s, 153e9), SIM_NB (512), SIM_LA (1)
"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from gptools_amd.dist import DistributedLML, GridLML, HipPanelOps, _StreamEvent

wl, Pr, Pc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lat = float(sys.argv[4]) * 1e-3                                     # ms
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
BW = float(os.environ.get("SIM_BW", "153e9"))
NB = int(os.environ.get("SIM_NB", "512"))
LA = bool(int(os.environ.get("SIM_LA", "1")))
W = Pr * Pc
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
kid = bench.KID[kernel]
gate = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsimgate.so"))
gate.gate_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
gate.gate_wait.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]


ops = HipPanelOps(0)          # (plain events on the queues, as in the product: a timing-capable event is a barrier packet, ~6 us of
                              #  command-processor time each, and the engine records about ten per step)


class Recorder(DistributedLML):
    def _factor_staged(self, k, buf):
        DistributedLML._factor_staged(self, k, buf)
        self.saved[k] = buf[:self.NP - k * self.nb].clone()


rec = Recorder(X, n, nb=NB, ops=ops, schedule="bcast")
rec.saved = {}
ll_ref, ld_ref = rec.fit(kid, params, y, err)
panels = rec.saved                               # panels[k]: rows [k nb, NP) of block column k of L
nblk = rec.nblk
del rec
torch.cuda.empty_cache()
eye = torch.eye(NB, dtype=torch.float64, device="cuda")
Winv = {k: torch.linalg.solve_triangular(torch.tril(panels[k][:NB]), eye, upper=False) for k in range(nblk)}


deliver = torch.cuda.Stream()          # the "DMA engine": payloads land on it, beside the compute queues


class Handle(object):
    """Receive side of one modelled broadcast.  The payload is copied into the buffer on the delivery stream as soon as the
    buffer may be written (the issue point of the broadcast on the rank's queue) -- HBM traffic beside the compute, as a
    receive is; EVERY queue that waits for the broadcast then waits for that copy and is held, by a gate kernel of its own,
    until the device clock reaches the arrival time (each queue waits for the RCCL work by itself)."""
    def __init__(self, plan, kind, k, buf, src, t):
        self.plan, self.t = plan, t
        cur = torch.cuda.current_stream()
        deliver.wait_stream(cur)
        with torch.cuda.stream(deliver):
            plan._payload(kind, k, buf, src)
            self.copied = torch.cuda.Event()
            self.copied.record(deliver)

    def wait(self):
        cur = torch.cuda.current_stream()
        cur.wait_event(self.copied)
        if self.t is not None and self.t > 0.0:
            gate.gate_wait(ctypes.c_void_p(cur.cuda_stream), ctypes.c_void_p(self.plan.t0_dev.data_ptr()), int(self.t * 1e5))   # ms -> 10 ns ticks


class ModelRank(GridLML):
    """Grid position `layout` of the P_r x P_c job: own pieces computed, foreign ones gated by their modelled arrival."""

    def _assemble(self, *a, **kw):
        st = torch.cuda.current_stream()
        gate.gate_stamp(ctypes.c_void_p(st.cuda_stream), ctypes.c_void_p(self.t0_dev.data_ptr()))
        self.t0_ev = torch.cuda.Event(enable_timing=True)
        self.t0_ev.record(st)
        return GridLML._assemble(self, *a, **kw)

    def _allreduce(self, t, op):
        pass

    def _payload(self, kind, k, buf, src):
        """What the broadcast would have delivered, from the complete factor."""
        nb = self.nb
        P = panels[k]
        if kind == "W":
            buf.copy_(Winv[k])
        elif kind == "H":
            buf.copy_(P[nb:2 * nb])
        elif kind in ("R", "R0"):
            # the rank's rows I >= k + 2 of panel k ("R0": the first of them alone, "R": the rest where an R0 exists, else all)
            li0 = self.li_ge(k + 2)
            has0 = self.pr == (k + 2) % self.Pr and k + 2 < nblk
            rows = self.my_rows[li0:]
            rows = rows[:1] if kind == "R0" else (rows[1:] if has0 else rows)
            if rows:
                buf.view(len(rows), nb, nb).copy_(P.view(-1, nb, nb)[rows[0] - k::self.Pr][:len(rows)])
        else:
            # process row src[0]'s share of this rank's columns J >= k + 2: J = J0 + t lcm
            lj0 = self.lj_ge(k + 2)
            sc = self.lcm // self.Pc
            J0 = next(J for J in self.my_cols[lj0:lj0 + sc] if J % self.Pr == src[0])
            nt = buf.shape[0] // nb
            buf.view(nt, nb, nb).copy_(P.view(-1, nb, nb)[J0 - k::self.lcm][:nt])

    def _xbcast(self, kind, k, buf, src, group, size):
        if size <= 1 or buf.numel() == 0:
            return []
        nbytes = buf.numel() * buf.element_size()
        key = (kind, k, src)
        cur = torch.cuda.current_stream()
        if src == (self.pr, self.pc):
            e = torch.cuda.Event(enable_timing=True)          # production is complete HERE, on the producing queue
            e.record(cur)
            self.produced[key] = (e, nbytes)
            return []
        return [Handle(self, kind, k, buf, src, self.arrive.get(key + ((self.pr, self.pc),)))]


def members(kind, k, src):
    """Grid positions that receive broadcast (kind, k) from src."""
    if kind == "W":
        return [(r, src[1]) for r in range(Pr) if r != src[0]]
    if kind == "H":
        return [(r, c) for r in range(Pr) for c in range(Pc) if (r, c) != src]
    if kind in ("R", "R0"):
        return [(src[0], c) for c in range(Pc) if c != src[1]]
    return [(r, src[1]) for r in range(Pr) if r != src[0]]


plans = []
for r in range(W):
    p = ModelRank(X, n, (Pr, Pc), nb=NB, ops=ops, layout=r, lookahead=LA)
    p.t0_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    p.arrive, p.produced = {}, {}
    plans.append(p)
torch.cuda.synchronize()
arrive, link_free_hist, hist = {}, None, []
for it in range(iters):
    ends, delta = [], 0.0
    for r, p in enumerate(plans):
        p.arrive, p.produced = arrive, {}
        torch.cuda.synchronize()
        ll, ld = p.fit(kid, params, y, err)
        e_end = torch.cuda.Event(enable_timing=True)
        e_end.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ends.append(p.t0_ev.elapsed_time(e_end))
        ld_sum = ld if r == 0 else ld_sum + ld
        # link model for everything THIS rank produced: FIFO per directed link, in order of production
        free = {}
        prod = sorted(((p.t0_ev.elapsed_time(e), key, nb_) for key, (e, nb_) in p.produced.items()))
        for tp, key, nb_ in prod:
            kind, k, src = key
            for dst in members(kind, k, src):
                start = max(tp, free.get(dst, 0.0))
                free[dst] = start + nb_ / BW * 1e3
                ta = free[dst] + lat
                akey = key + (dst,)
                delta = max(delta, abs(ta - arrive.get(akey, 0.0)))
                arrive[akey] = ta
    assert abs(ld_sum - ld_ref) <= 1e-9 * abs(ld_ref), (ld_sum, ld_ref)
    hist.append((max(ends), delta))
    print("sweep %d: rank end times (ms) %s -> max %.2f; arrivals moved by up to %.2f ms" % (
        it, " ".join("%.1f" % e for e in ends), max(ends), delta), flush=True)
    if it >= 2 and delta < 0.15:
        break
if os.environ.get("SIM_TRACE"):
    # one more replay of one rank with its main-queue marks and the times its own pieces were produced
    r = int(os.environ["SIM_TRACE"])
    p = plans[r]
    p.trace = True
    p.arrive, p.produced = arrive, {}
    p.fit(kid, params, y, err)
    torch.cuda.synchronize()
    st = p.timings.get("steps_ms", [])
    arr_ = {k: t for k, tag, t in st if tag == "arrived"}
    app_ = {k: t for k, tag, t in st if tag == "applied"}
    other = {}
    for k, tag, t in st:
        if tag not in ("arrived", "applied"):
            other.setdefault(k, []).append("%s %.2f" % (tag, t))
    print("rank %d = (%d, %d): step: panel k arrived on main | update done | marks of the panel / recv queues (ms from the start)" % (r, p.pr, p.pc))
    for k in sorted(arr_):
        print("  %2d: %7.2f | %7.2f | %s" % (k, arr_[k], app_.get(k, 0.0), "  ".join(other.get(k, []))))
    sys.stdout.flush()
    os._exit(0)
if os.environ.get("SIM_DUMP"):
    # the converged chain: when W / H of every SIM_DUMP-th panel were produced, and rank 0's main-queue marks
    allprod = {}
    for r, p in enumerate(plans):
        for key, (e, nb_) in p.produced.items():
            allprod[key] = p.t0_ev.elapsed_time(e)
    for k in range(0, nblk, int(os.environ["SIM_DUMP"])):
        row = sorted((kind, src, t) for (kind, kk, src), t in allprod.items() if kk == k)
        print("panel %2d: " % k + "  ".join("%s%s %.2f" % (kind, src, t) for kind, src, t in row))
T = hist[-1][0]
print("MODEL-GRID %s N=%d grid=%dx%d lookahead=%d latency=%.0fus bw=%.0fGB/s nb=%d: %.1f ms -> %.1f TFLOP/s = %.1f %% of %d x 78.6" % (
    wl, N, Pr, Pc, int(LA), lat * 1e3, BW * 1e-9, NB, T, bench.flops_fit(N) / T * 1e-9, 100 * bench.flops_fit(N) / T * 1e-9 / (78.6 * W), W))
