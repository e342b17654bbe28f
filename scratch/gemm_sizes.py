"""Stand-alone rate of the trailing-update launch (lower trapezoid, k = 384 / 512) by size, on the library's own
CU-masked main stream (224 CUs) -- where does the tail of a launch start to hurt?"""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
lib = _lib.load()
ctx = _lib.Context(0)                      # own masked stream
st = torch.cuda.ExternalStream(int(ctx.stream))
ctx.set_option("lookahead", 0)
import os
ctx.set_option("tile", int(os.environ.get("TILE", "0")))
for k in (384, 512):
    for m in (1024, 1536, 2048, 3072, 4096, 5120, 6144, 7168, 8192, 12288, 16384):
        with torch.cuda.stream(st):
            A = torch.randn(m, k, dtype=torch.float64, device="cuda")
            C = torch.randn(m, m, dtype=torch.float64, device="cuda")
            for _ in range(2):
                _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
            e1.record()
        st.synchronize()
        t = e0.elapsed_time(e1) / 5
        tiles = (m // 64) * (m // 64 + 1) // 2
        print("k=%d m=%5d: %8.1f us  %5.1f TFLOP/s  (%d tiles = %.2f x 896 slots)" % (k, m, t * 1e3, m * (m + 1) * k / t * 1e-9, tiles, tiles / 896.0))
        del A, C
