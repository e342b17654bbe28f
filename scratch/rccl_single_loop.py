"""Repeat the single-rank RCCL exercise of tests/test_gpu_parity.py to catch intermittent hangs (faulthandler dump after 40 s).
Variants through the environment: V_ONECOMM=1 (tail chunks on the main communicator), V_NOINV=1, V_ONECHUNK=1, V_SCHEDS=..."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(int(os.environ.get("V_TIMEOUT", "40")), exit=True)
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], GPT_DIST_FORCE_COLLECTIVES='1')
torch.cuda.set_device(0)
t0 = time.time()
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from gptools_amd.dist import DistributedLML
from test_gpu_parity import c3_inputs
X, n, y = c3_inputs(1500, 3)
kw = {}
if os.environ.get("V_NOINV"): kw["inv_trsm"] = False
if os.environ.get("V_ONECHUNK"): kw["chunk_blocks"] = (1000,)
plan = DistributedLML(X, n, nb=128, device=0, sag_min_bytes=0, **kw)
if os.environ.get("V_ONECOMM"): plan.group_tail = plan.group
print("init %.1f s" % (time.time() - t0), flush=True)
scheds = os.environ.get("V_SCHEDS", "bcast+bcast,bcast+scatter_gather,pipelined+bcast,pipelined+scatter_gather").split(",")
for se in scheds:
    sched, exch = se.split("+")
    plan.schedule, plan.exchange = sched, exch
    for rep in range(3):
        t1 = time.time()
        r = plan.fit(1, np.array([1.0, 0.3, 0.3, 0.3]), y, 0.05 * np.ones(1500))
        print(sched, exch, "fit %d: %.2f s" % (rep, time.time() - t1), flush=True)
dist.destroy_process_group()
print("done %.1f s" % (time.time() - t0), flush=True)
