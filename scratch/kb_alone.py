"""Stand-alone K-builder rates (gpt_dev_kbuild, lower triangle + fused diagonal, HIP events, best of 8):
python scratch/kb_alone.py [N]"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = 3
ctx = _lib.Context(0)
lib = _lib.load()
rs = np.random.RandomState(0)
X = rs.rand(N, d)
err = 0.05 * np.ones(N)
st = torch.cuda.ExternalStream(int(ctx.stream))
for kname, params in (("m52", np.array([1.0, 0.3, 0.3, 0.3])), ("se", np.array([1.0, 0.3, 0.3, 0.3]))):
    for deriv in ("none", "quarter", "all"):
        n = np.zeros((N, d), dtype=np.int32)
        if deriv == "quarter": n[3 * N // 4:, 0] = 1
        if deriv == "all": n[:, 1] = 1
        with torch.cuda.stream(st):
            dX = torch.from_numpy(X).cuda(); dn = torch.from_numpy(n).cuda(); de = torch.from_numpy(err).cuda()
            dK = torch.empty((N, N), dtype=torch.float64, device="cuda")
            best = 1e9
            for _ in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                _lib.check(lib.gpt_dev_kbuild(ctx.handle, bench.KID[kname], _lib.dptr(params), len(params), dX.data_ptr(), dn.data_ptr(), N,
                                              dX.data_ptr(), dn.data_ptr(), N, d, -1, 1, None, 1, 0, 0, de.data_ptr(), 0.0, 2.2e-14, dK.data_ptr(), N))
                e1.record(st)
                st.synchronize()
                best = min(best, e0.elapsed_time(e1))
        b = 8.0 * N * (N + 1) / 2
        print("%-4s deriv=%-8s N=%d: %.4f ms  %.2f TB/s written  %.0f Gpairs/s" % (kname, deriv, N, best, b / best * 1e-9, N * (N + 1) / 2 / best * 1e-6))
