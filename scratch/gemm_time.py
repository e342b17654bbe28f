"""Times one trailing-update shape on the masked main stream: python scratch/gemm_time.py [m] [k] [reps]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
lib = _lib.load()
ctx = _lib.Context(0)
st = torch.cuda.ExternalStream(int(ctx.stream))
ctx.set_option("lookahead", 1)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 7168
k = int(sys.argv[2]) if len(sys.argv) > 2 else 384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if len(sys.argv) > 4:
    ctx.set_option("tile", int(sys.argv[4]))
with torch.cuda.stream(st):
    A = torch.randn(m, k, dtype=torch.float64, device="cuda")
    C = torch.randn(m, m, dtype=torch.float64, device="cuda")
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
        e1.record(st)
        st.synchronize()
        best = min(best, e0.elapsed_time(e1))
fl = k * (m * (m + 1.0))
print("gemm tri m=%d k=%d tile=%s: best %.1f us -> %.1f TF/s" % (m, k, sys.argv[4] if len(sys.argv) > 4 else "0", best * 1e3, fl / best * 1e-9))
