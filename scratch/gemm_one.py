"""One trailing-update shape launched a few times on the masked main stream (for rocprofv3 --pmc passes)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
lib = _lib.load()
ctx = _lib.Context(0)
st = torch.cuda.ExternalStream(int(ctx.stream))
ctx.set_option("lookahead", 1)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 7168
k = int(sys.argv[2]) if len(sys.argv) > 2 else 384
with torch.cuda.stream(st):
    A = torch.randn(m, k, dtype=torch.float64, device="cuda")
    C = torch.randn(m, m, dtype=torch.float64, device="cuda")
    for _ in range(6):
        _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
st.synchronize()
