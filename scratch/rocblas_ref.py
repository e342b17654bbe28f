"""Vendor reference for the trailing-update shape: torch (rocBLAS / hipBLASLt) fp64 C -= A A^T against gemm_nt_kernel."""
import sys, torch, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
ctx = _lib.Context(0)
for (m, k) in ((4096, 512), (8192, 512), (8192, 384), (16384, 512), (24576, 512)):
    A = torch.randn(m, k, dtype=torch.float64, device="cuda")
    C = torch.randn(m, m, dtype=torch.float64, device="cuda")
    for name, fn in (("torch.addmm (full m x m)", lambda: torch.addmm(C, A, A.t(), beta=1.0, alpha=-1.0, out=C)),):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        print("m=%5d k=%3d %-28s %.3f ms  %.1f TFLOP/s" % (m, k, name, t, 2.0 * m * m * k / t * 1e-9))
    # ours: lower trapezoid only (SYRK-style), flops counted for the computed half
    lib = _lib.load()
    st = torch.cuda.Stream(); c2 = _lib.Context(0, stream=st.cuda_stream)
    with torch.cuda.stream(st):
        for _ in range(2): _lib.check(lib.gpt_dev_gemm_nt(c2.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): _lib.check(lib.gpt_dev_gemm_nt(c2.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
        e1.record()
    st.synchronize()
    t = e0.elapsed_time(e1) / 5
    print("m=%5d k=%3d %-28s %.3f ms  %.1f TFLOP/s (lower half: m(m+1)k flops)" % (m, k, "gemm_nt_kernel tri=1", t, 1.0 * m * (m + 1) * k / t * 1e-9))
    del A, C
