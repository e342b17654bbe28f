"""A second process with a live HIP context makes the pipelined schedule + RCCL crawl: which ingredient?"""
import os, subprocess, sys, time
sys.path.insert(0, '/root/repo')
import torch
x = torch.ones(10, device="cuda"); torch.cuda.synchronize()
port = 29700
for name, env in (("bb,pb", {"V_SCHEDS": "bcast+bcast,pipelined+bcast"}),
                  ("bs,pb", {"V_SCHEDS": "bcast+scatter_gather,pipelined+bcast"}),
                  ("ps", {"V_SCHEDS": "pipelined+scatter_gather"}),
                  ("pb,ps", {"V_SCHEDS": "pipelined+bcast,pipelined+scatter_gather"}),
                  ("full", {"V_SCHEDS": "bcast+bcast,bcast+scatter_gather,pipelined+bcast,pipelined+scatter_gather"}),
                  ("full, one comm", {"V_ONECOMM": "1", "V_SCHEDS": "bcast+bcast,bcast+scatter_gather,pipelined+bcast,pipelined+scatter_gather"})):
    port += 1
    t0 = time.time()
    e = dict(os.environ, V_TIMEOUT="40", **env)
    out = subprocess.run([sys.executable, "/root/repo/scratch/rccl_single_loop.py", str(port)], capture_output=True, text=True, env=e)
    lines = [l for l in out.stdout.splitlines() if l.startswith(("init", "done", "pipelined", "bcast"))]
    print("%-18s child rc %d in %.1f s: %s" % (name, out.returncode, time.time() - t0, " | ".join(l.replace("pipelined", "p").replace("scatter_gather", "s").replace("bcast", "b").replace("fit ", "") for l in lines)), flush=True)
