"""Round-robin comparison of several library builds on one box: python scratch/ab_multi.py <workload> <reps> lib1.so lib2.so ..."""
import sys, os, subprocess
wl, reps, libs = sys.argv[1], sys.argv[2], sys.argv[3:]
code = """
import sys; sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import os, ctypes
_lib.LIB_PATH = os.environ['GPT_AB_LIB']
probe = ctypes.CDLL(_lib.LIB_PATH)
for name in list(_lib.SIGNATURES):
    if not hasattr(probe, name): del _lib.SIGNATURES[name]
import runpy; sys.argv = ['fit_loop.py', %r, %r]; runpy.run_path('/root/repo/scratch/fit_loop.py', run_name='__main__')
"""
best = {l: 1e9 for l in libs}
for rnd in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code % (wl, reps)], env=dict(os.environ, GPT_AB_LIB=os.path.abspath(l)), capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-200:]
        try: best[l] = min(best[l], float(line.split("best")[1].split("ms")[0]))
        except Exception: print(l, line)
for l in libs: print("%-36s best %.3f ms" % (l, best[l]))
