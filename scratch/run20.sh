cd $GRAFT_REPO_ROOT
O=gpurun_out/mixed; mkdir -p $O
for p in 0 40 60 100; do echo "== GPT_GEMM_MIXED=$p"; GPT_GEMM_MIXED=$p timeout 200 python scratch/gemm_curve.py 2>&1 | grep "m=\|sum" | awk '{print $1, $2, $5, $6, $8, $9}' | tr '\n' ';'; echo; done > $O/curve.txt 2>&1
timeout 600 python scratch/env_ab.py c3 20 3 GPT_GEMM_MIXED 0 40 60 100 > $O/ab_c3.txt 2>&1
python - <<'PY' > $O/bits.txt 2>&1
import os, subprocess, sys
code = """
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gptools_amd import _lib
from test_gpu_parity import c3_inputs
c = _lib.Context(0)
for N in (8192, 5000):
    X, n, y = c3_inputs(N, 3)
    c.set_data(X, n)
    ll, ld = c.fit(1, np.array([1.0, 0.3, 0.3, 0.3]), 0.0, y, 0.05 * np.ones(N), 1e2 * sys.float_info.epsilon)
    print('RESULT', N, repr(ll), repr(ld))
"""
res = {}
for p in ("0", "60"):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPT_GEMM_MIXED=p), capture_output=True, text=True)
    res[p] = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(p, res[p], out.stderr[-300:] if out.returncode else "")
print("identical bits:", res["0"] == res["60"])
PY
cat $O/curve.txt; cat $O/ab_c3.txt; cat $O/bits.txt
