"""Round-robin comparison of several environment SETTINGS (each "A=1,B=2" or "-" for none) in separate processes:
   python scratch/env_multi.py <workload> <reps> <rounds> set1 set2 ..."""
import sys, os, subprocess, numpy as np
wl, reps, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
sets = sys.argv[4:]
res = {v: [] for v in sets}
for rnd in range(rounds):
    for v in sets:
        env = dict(os.environ)
        if v != "-":
            env.update(dict(kv.split("=") for kv in v.split(",")))
        out = subprocess.run([sys.executable, "/root/repo/scratch/fit_loop.py", wl, reps], env=env, capture_output=True, text=True)
        try:
            res[v].append(float(out.stdout.strip().splitlines()[-1].split("best")[1].split("ms")[0]))
        except Exception:
            print(v, out.stdout[-200:], out.stderr[-300:])
for v in sets:
    if res[v]: print("%-40s best %.3f  median %.3f ms" % (v, min(res[v]), np.median(res[v])))
