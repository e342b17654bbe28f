"""Missing-edge detector for gptools_amd.dist (run by tests/test_gpu_a_dist_processes.py; not a test module itself).

`world` gloo ranks share cuda:0 and run the product ops with a random delay kernel at the start of every queue block
(main / panel / recv) and in front of every dense operation, so the relative timing of the queues changes from call to
call.  With every dependency expressed as an event the results do not move; a missing edge shows up as a wrong ll.
(Validated against a real one: with the "every queue waits for this rank's K build" edge of the row-chunked schedule
removed, 4 of 18 evaluations per rank came out wrong; the two-rank test without jitter had passed by timing luck.)

    python tests/dist_jitter_worker.py <world>            spawns the ranks and reports
    python tests/dist_jitter_worker.py <world> <rank>     one rank"""
import contextlib, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 2:
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), sys.argv[1], str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(int(sys.argv[1]))]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        print(so.strip() or se[-1500:])
    sys.exit(max(p.returncode for p in procs))
import faulthandler; faulthandler.dump_traceback_later(400, exit=True)
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
world, rank = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29540 + world))
torch.cuda.set_device(0)
if world == 1:
    # one rank, RCCL, collectives forced on: the NCCL streams' ordering against the jittered queues, both exchanges
    os.environ["GPT_DIST_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
else:
    dist.init_process_group('gloo', rank=rank, world_size=world)
os.environ["GPT_JITTER"] = "60"          # ... and inside the library: in front of every launch of the panel routines
from gptools_amd.dist import DistributedLML, GridLML, HipPanelOps
from test_gpu_parity import c3_inputs
rng = random.Random(100 + rank)


def nap(scale=1200000):
    torch.cuda._sleep(int(rng.random() ** 3 * scale))       # on the current stream: mostly short, up to ~0.5 ms


class JitterOps(HipPanelOps):
    """Random delay kernels at the start of every queue block and in front of every dense operation."""

    def queue(self, q):
        cm = HipPanelOps.queue(self, q)

        @contextlib.contextmanager
        def wrap():
            with cm:
                nap()
                yield
        return wrap()

    def kbuild_block(self, *a, **k):
        nap(4000000)
        return HipPanelOps.kbuild_block(self, *a, **k)

    def potrf_panel(self, *a, **k):
        nap()
        return HipPanelOps.potrf_panel(self, *a, **k)

    def trsm_rlt(self, *a, **k):
        nap()
        return HipPanelOps.trsm_rlt(self, *a, **k)

    def trinv(self, *a, **k):
        nap()
        return HipPanelOps.trinv(self, *a, **k)

    def gemm_nt(self, *a, **k):
        with torch.cuda.stream(self._stream[k.get("q", "main")]):
            nap()
        return HipPanelOps.gemm_nt(self, *a, **k)

    def gemm_nt_stair(self, *a, **k):
        with torch.cuda.stream(self._stream[k.get("q", "main")]):
            nap()
        return HipPanelOps.gemm_nt_stair(self, *a, **k)

    def gemm_nt_gridstair(self, *a, **k):
        with torch.cuda.stream(self._stream[k.get("q", "main")]):
            nap()
        return HipPanelOps.gemm_nt_gridstair(self, *a, **k)

    def kbuild_rect(self, *a, **k):
        nap(4000000)
        return HipPanelOps.kbuild_rect(self, *a, **k)

    def copy2d(self, *a, **k):
        with torch.cuda.stream(self._stream[k.get("q", "panel")]):
            nap(300000)
        return HipPanelOps.copy2d(self, *a, **k)


X, n, y = c3_inputs(2500, 3)
p = np.array([1.0, 0.3, 0.3, 0.3])
ref = DistributedLML(X, n, nb=128, device=0, schedule="bcast").fit(1, p, y, 0.05 * np.ones(2500))
jops = JitterOps(0)
plan = DistributedLML(X, n, nb=128, ops=jops, owner_first=(rank % 2 == 0) if world == 2 else None, sag_min_bytes=0)
# the process grids this many ranks allow (world 1: the collectives of a 1 x 1 grid are forced on)
grids = {1: [(1, 1)], 2: [(1, 2), (2, 1)], 3: [(3, 1), (1, 3)], 4: [(2, 2)]}[world]
cases = [("1-D bcast", plan, "bcast"), ("1-D scatter+all-gather" if world == 1 else "1-D bcast again", plan,
                                        "scatter_gather" if world == 1 else "bcast")]
cases += [("grid %d x %d" % g, GridLML(X, n, g, nb=128, ops=jops), None) for g in grids]
bad = 0
reps = int(os.environ.get("JITTER_REPS", "6"))
for name, pl, exch in cases:
    if exch is not None:
        pl.exchange = exch
    for rep in range(reps):
        try:
            r = pl.fit(1, p, y, 0.05 * np.ones(2500))
            ok = abs(r[0] - ref[0]) <= 1e-10 * abs(ref[0]) and abs(r[1] - ref[1]) <= 1e-11 * abs(ref[1])
        except np.linalg.LinAlgError as e:
            r, ok = ("LinAlgError", str(e)[:40]), False
        bad += (not ok)
        if not ok:
            print("rank %d %s rep %d: %s  (reference %s)" % (rank, name, rep, r, ref), flush=True)
print("rank %d: %d bad of %d" % (rank, bad, len(cases) * reps), flush=True)
dist.destroy_process_group()
sys.exit(1 if bad else 0)
