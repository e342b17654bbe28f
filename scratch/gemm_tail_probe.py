"""Does a trailing update lose its last half tile-time to quantisation?  The lower trapezoid m x m x k as ONE launch of 64x64 tiles,
against the same work as two concurrent launches on two streams with the same CU mask: the top rows with 64x64 tiles, the bottom
h rows (a full-width rectangle) with 32x32 tiles -- the small tiles fill the big launch's tail."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
lib = _lib.load()
c1, c2 = _lib.Context(0), _lib.Context(0)
k = 384
dev = torch.device("cuda:0")
def run(m, h, reps=12):
    A = torch.randn(m, k, dtype=torch.float64, device=dev) * 0.01
    C = torch.zeros(m, m, dtype=torch.float64, device=dev)
    best = 1e9
    for rep in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        if h == 0:
            c1.set_option("tile", 0)
            _lib.check(lib.gpt_dev_gemm_nt(c1.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
        else:
            mt = m - h
            c1.set_option("tile", 0)
            c2.set_option("tile", 32)
            _lib.check(lib.gpt_dev_gemm_nt(c1.handle, mt, mt, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
            _lib.check(lib.gpt_dev_gemm_nt(c2.handle, h, m, k, -1.0, A[mt:].data_ptr(), k, A.data_ptr(), k, 1.0, C[mt:].data_ptr(), m, 0))
        c1.synchronize(); c2.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e6
for m in (7808, 6272, 4736, 3200, 2176):
    base = run(m, 0)
    row = ["m=%d  one launch %.1f us |" % (m, base)]
    for h in (128, 256, 384, 512, 768):
        row.append("h=%d %.1f" % (h, run(m, h)))
    print(" ".join(row), flush=True)
