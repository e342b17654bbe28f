"""Write-pattern ceiling of the K-builder: the same launch with the ZeroKernel pair function (no arithmetic)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import ctypes as C
lib = _lib.load()
ctx = _lib.Context(0)
import torch
N, D = 16384, 2
X = torch.rand(N, D, dtype=torch.float64, device="cuda"); n = torch.zeros(N, D, dtype=torch.int32, device="cuda")
K = torch.empty(N, N, dtype=torch.float64, device="cuda")
err = torch.zeros(N, dtype=torch.float64, device="cuda")
p = _lib.f64(np.array([1.0, 0.3, 0.3]))
st = torch.cuda.Stream()
ctx2 = _lib.Context(0, stream=st.cuda_stream)
for kid, name in ((3, "zero"), (0, "se"), (1, "m52")):
    pp = p if kid != 3 else _lib.f64(np.array([1.0]))
    for lower in (1, 0):
        ts = []
        for _ in range(5):
            with torch.cuda.stream(st):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.gpt_dev_kbuild(ctx2.handle, kid, _lib.dptr(pp), len(pp), X.data_ptr(), n.data_ptr(), N,
                                              X.data_ptr(), n.data_ptr(), N, D, -1, 1, None, lower, 0, 0, err.data_ptr(), 0.0, 0.0, K.data_ptr(), N))
                e1.record()
            st.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = min(ts)
        byts = 8.0 * N * (N + 1) / 2 if lower else 8.0 * N * N
        print("%-5s lower=%d: %.3f ms  %.2f TB/s written" % (name, lower, t, byts / t * 1e-9))
