"""A/B two builds of the library on the same box: python scratch/ab_lib.py <workload> <reps> [path/to/other.so]"""
import sys, os, subprocess
wl, reps = sys.argv[1], sys.argv[2]
other = sys.argv[3] if len(sys.argv) > 3 else "scratch/libgpt_hip_old.so"
code = """
import sys; sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import os
if os.environ.get('GPT_AB_LIB'):
    _lib.LIB_PATH = os.environ['GPT_AB_LIB']
    import ctypes
    probe = ctypes.CDLL(_lib.LIB_PATH)
    for name in list(_lib.SIGNATURES):
        if not hasattr(probe, name): del _lib.SIGNATURES[name]      # an older build lacks the newer entry points
import runpy; sys.argv = ['fit_loop.py', %r, %r] + %r; runpy.run_path('/root/repo/scratch/fit_loop.py', run_name='__main__')
"""
for rnd in range(2):
    for name, lib in (("current", ""), ("other", os.path.abspath(other))):
        env = dict(os.environ, GPT_AB_LIB=lib)
        out = subprocess.run([sys.executable, "-c", code % (wl, reps, sys.argv[4:])], env=env, capture_output=True, text=True)
        print(name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
