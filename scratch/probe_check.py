"""Exercise bench.comm_probe / the tuned N>1 bench path on one GPU: a 1-rank nccl group with the collectives forced on."""
import os, sys, json
sys.path.insert(0, '/root/repo')
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29531', GPT_DIST_FORCE_COLLECTIVES='1')
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
import bench
print(json.dumps(bench.comm_probe(torch, dist, 1, 0)))
from gptools_amd.dist import DistributedLML
kernel, N, d, deriv = bench.WORKLOADS["c5"]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
plan = DistributedLML(X, n, nb=512, device=0)
for sched, exch in (("pipelined", "bcast"), ("pipelined", "scatter_gather"), ("bcast", "bcast"), ("bcast", "scatter_gather")):
    plan.schedule, plan.exchange = sched, exch
    step = lambda: plan.fit(bench.KID[kernel], params, y, err)
    print(sched, exch, step(), step())
    tr = bench.dist_trace(plan, step)
    print({k: v for k, v in tr.items() if not k.endswith("ed_ms")})
dist.destroy_process_group()
