"""python scratch/ab_wall.py <workload> <reps> lib1.so lib2.so ...: wall time per evaluation of several library builds, round robin"""
import sys, os, subprocess
wl, reps, libs = sys.argv[1], sys.argv[2], sys.argv[3:]
best = {l: 1e9 for l in libs}
med = {l: 1e9 for l in libs}
for rnd in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, "/root/repo/scratch/run_with_lib.py", l, "/root/repo/scratch/wall_loop.py", wl, reps], capture_output=True, text=True)
        try:
            line = out.stdout.strip().splitlines()[-1]
            best[l] = min(best[l], float(line.split("best")[1].split("ms")[0]))
            med[l] = min(med[l], float(line.split("median")[1].split(")")[0]))
        except Exception: print(l, out.stdout[-200:], out.stderr[-300:])
for l in libs: print("%-28s best %.3f ms wall, best median %.3f" % (l, best[l], med[l]))
