"""Same-box A/B of two builds of the library in separate processes, round-robin: python scratch/lib_ab.py <workload> <reps> <rounds>
(scratch/lib_prev.so = the build before the change, copied by hand; the other one is the in-tree library)"""
import sys, os, subprocess, shutil, numpy as np
wl, reps, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = '/root/repo'
tmp = '/tmp/lib_ab_prev'
if os.path.isdir(tmp): shutil.rmtree(tmp)
shutil.copytree(root + '/gptools_amd', tmp + '/gptools_amd', ignore=shutil.ignore_patterns('csrc'))
shutil.copy(root + '/scratch/lib_prev.so', tmp + '/gptools_amd/libgpt_hip.so')
shutil.copy(root + '/bench.py', tmp + '/bench.py')
code = open(root + '/scratch/fit_loop.py').read()
res = {"prev": [], "new": []}
for rnd in range(rounds):
    for name, base in (("prev", tmp), ("new", root)):
        out = subprocess.run([sys.executable, "-c", code.replace("/root/repo", base), wl, reps], capture_output=True, text=True, cwd=base)
        try:
            res[name].append(float(out.stdout.strip().splitlines()[-1].split("best")[1].split("ms")[0]))
        except Exception:
            print(out.stdout[-300:], out.stderr[-300:])
for k, v in res.items():
    print("%-5s best %.3f  median %.3f ms  (%s)" % (k, min(v), float(np.median(v)), " ".join("%.3f" % x for x in v)))
