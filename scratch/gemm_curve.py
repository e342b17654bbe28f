"""TFLOP/s of the trailing-update launch (lower trapezoid m x m x 384, one launch) over the sizes an N = 8192 factorisation meets,
with the tile count and the rounds over the stream's workgroup slots (224 CUs x 4)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
lib = _lib.load()
c1 = _lib.Context(0)
k = 384
dev = torch.device("cuda:0")
def run(m, reps=15):
    A = torch.randn(m, k, dtype=torch.float64, device=dev) * 0.01
    C = torch.zeros(m, m, dtype=torch.float64, device=dev)
    best = 1e9
    for rep in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        _lib.check(lib.gpt_dev_gemm_nt(c1.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
        c1.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e6
tot = 0
for j in range(1, 21):
    m = 8192 - 384 * j
    if m < 512: break
    us = run(m)
    nt = (m // 64) * (m // 64 + 1) // 2
    tot += us
    print("m=%5d tiles %5d rounds %5.2f  %7.1f us  %5.1f TFLOP/s (wall incl. launch + sync ~8 us)" % (m, nt, nt / 896.0, us, m * m * k / us * 1e-6), flush=True)
print("sum %.1f us" % tot)
