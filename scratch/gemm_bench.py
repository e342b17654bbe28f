import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
st = torch.cuda.Stream()
ctx = _lib.Context(0, stream=st.cuda_stream)
lib = _lib.load()
def run(m, n, k, tri, tile, reps=5):
    ctx.set_option("tile", tile)
    with torch.cuda.stream(st):
        A = torch.randn(m, k, dtype=torch.float64, device='cuda')
        B = A if tri else torch.randn(n, k, dtype=torch.float64, device='cuda')
        C = torch.randn(m, n, dtype=torch.float64, device='cuda')
        for _ in range(2):
            _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, n, k, -1.0, A.data_ptr(), k, B.data_ptr(), k, BETA, C.data_ptr(), n, tri))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, n, k, -1.0, A.data_ptr(), k, B.data_ptr(), k, BETA, C.data_ptr(), n, tri))
        e1.record(st); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
    elems = (n * (n + 1) / 2 + (m - n) * n) if tri else m * n
    fl = 2.0 * k * elems
    print("m %5d n %5d k %4d tri %d tile %3d: %8.3f ms  %6.2f TF/s" % (m, n, k, tri, tile, ms, fl / ms * 1e-9))
cases = [(8192, 8192, 256, 0), (8192, 8192, 512, 0), (8192, 8192, 1024, 0), (8192, 8192, 256, 1), (8192, 8192, 512, 1),
         (4096, 4096, 256, 1), (4096, 4096, 512, 1), (2048, 2048, 256, 1), (16384, 16384, 512, 1), (8192, 256, 256, 1), (8192, 128, 128, 1), (4096, 128, 128, 1)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
import os
BETA = float(os.environ.get("BETA", "1.0"))
for (m, n, k, tri) in cases:
    for tile in [int(v) for v in os.environ.get("TILES", "128,129,64").split(",")]:
        run(m, n, k, tri, tile)
