# How long does a skinny gemm_nt launch take as a function of k?  (m rows against n = 2048 rows of an 8192-wide matrix)
import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
ctx = _lib.Context(0)
lib = ctx._lib
dev = torch.device('cuda:0')
LD = 8192
Bm = torch.randn(8192, LD, dtype=torch.float64, device=dev)
Am = torch.randn(256, LD, dtype=torch.float64, device=dev)
Cm = torch.zeros(256, LD, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
def run(m, n, k, reps=200):
    for it in range(2):
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            rc = lib.gpt_dev_gemm_nt(ctx.handle, m, n, k, 1.0, Am.data_ptr(), LD, Bm.data_ptr(), LD, 0.0, Cm.data_ptr(), LD, 0)
            assert rc == 0
        ctx.synchronize(); t = (time.perf_counter() - t0) / reps
    return t * 1e6
for m in (32, 64, 128):
    for n in (2048, 6144):
        print("m=%d n=%d: " % (m, n) + "  ".join("k=%d %.1f us" % (k, run(m, n, k)) for k in (64, 128, 256, 512, 1024, 2048)), flush=True)
