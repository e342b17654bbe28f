import sys, time, torch
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
lib = _lib.load(); c1 = _lib.Context(0); k = 384; dev = torch.device("cuda:0")
out = []
for m in (7808, 7424, 6656, 4736):
    A = torch.randn(m, k, dtype=torch.float64, device=dev) * 0.01
    C = torch.zeros(m, m, dtype=torch.float64, device=dev)
    best = 1e9
    for rep in range(15):
        torch.cuda.synchronize(); t = time.perf_counter()
        _lib.check(lib.gpt_dev_gemm_nt(c1.handle, m, m, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), m, 1))
        c1.synchronize(); best = min(best, time.perf_counter() - t)
    out.append("%d: %.1f us %.1f TF" % (m, best * 1e6, m * m * k / best * 1e-12))
print(" | ".join(out))
