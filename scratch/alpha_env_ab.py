"""A/B of the eager-alpha variants by environment switch, one process per variant (same box)."""
import os, subprocess, sys, json
code = r'''
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
import bench
kernel, N, d, deriv = bench.WORKLOADS[sys.argv[1]]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx = _lib.Context(0); ctx.set_data(X, n)
def run(eager, reps=40):
    ctx.set_option("eager_alpha", eager)
    for _ in range(5): ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); ctx.get_alpha(N)
    best = 1e9
    for r in range(6):
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
            if eager: a = ctx.get_alpha(N)
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3
l = run(0); e = run(1)
print("RES %.4f %.4f %.4f" % (l, e, e - l))
'''
for env in ({}, {"GPT_ALPHA_MIRROR": "1"}, {"GPT_BINV_EARLY_OFF": "1"}, {"GPT_ALPHA_TAIL_OFF": "1"}, {"GPT_BINV_EARLY_OFF": "1", "GPT_ALPHA_TAIL_OFF": "1"},
            {"GPT_BINV_EARLY_OFF": "1", "GPT_ALPHA_MIRROR": "1"}, {}):
    out = subprocess.run([sys.executable, "-c", code, sys.argv[1] if len(sys.argv) > 1 else "c3"], env=dict(os.environ, **env), capture_output=True, text=True)
    r = [l for l in out.stdout.splitlines() if l.startswith("RES")]
    print("%-60s lazy / eager / extra ms: %s" % (env or "(default: early inverses, alpha on the tail stream, D2H copy)", r[-1][4:] if r else out.stderr[-300:]), flush=True)
