"""Round-robin comparison of environment settings in separate processes on one box:
   python scratch/env_ab.py <workload> <reps> <rounds> VAR v1 v2 ...   -> best / median GPU-timeline ms per value"""
import sys, os, subprocess, numpy as np
wl, reps, rounds, var = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
vals = sys.argv[5:]
res = {v: [] for v in vals}
for rnd in range(rounds):
    for v in vals:
        out = subprocess.run([sys.executable, "/root/repo/scratch/fit_loop.py", wl, reps], env=dict(os.environ, **{var: v}),
                             capture_output=True, text=True)
        try:
            res[v].append(float(out.stdout.strip().splitlines()[-1].split("best")[1].split("ms")[0]))
        except Exception:
            print(v, out.stdout[-200:], out.stderr[-300:])
for v in vals:
    if res[v]: print("%s=%-8s best %.3f  median %.3f ms" % (var, v, min(res[v]), np.median(res[v])))
