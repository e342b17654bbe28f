"""Stand-alone durations of what sits on the serial chain of a partitioned factorisation (one idle MI355X, HIP events): the
512 x 512 diagonal block (gpt_dev_potrf_panel), its inverse (gpt_dev_trinv), a 512^3 product, and the per-step bulk operations
of one grid rank (slice / look-ahead GEMMs of m x 512 x 512).  For the per-term budget of DESIGN.md section 5.
  python scratch/chain_terms.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from gptools_amd.dist import HipPanelOps
ops = HipPanelOps(0)
rs = np.random.RandomState(0)


def timed(fn, reps=20, q="panel"):
    with ops.queue(q):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


for nb in (256, 512, 1024):
    M = rs.rand(nb, nb)
    A0 = torch.from_numpy(M.dot(M.T) + nb * np.eye(nb)).cuda()
    A = A0.clone()
    W = torch.empty((nb, nb), dtype=torch.float64, device="cuda")
    invd = torch.empty(((nb // 128) * 9216,), dtype=torch.float64, device="cuda")
    info = torch.zeros((1,), dtype=torch.int32, device="cuda")

    def potrf():
        ops.copy2d(A, A0)
        ops.potrf_panel(nb, nb, A.data_ptr(), nb, invd, info, 0)
    t_copy = timed(lambda: ops.copy2d(A, A0))
    t_potrf = timed(potrf) - t_copy
    t_inv = timed(lambda: ops.trinv(nb, A.data_ptr(), nb, invd, W.data_ptr(), nb))
    B = torch.from_numpy(rs.rand(nb, nb)).cuda()
    Cc = torch.zeros((nb, nb), dtype=torch.float64, device="cuda")
    t_g = timed(lambda: ops.gemm_nt(nb, nb, nb, 1.0, B.data_ptr(), nb, W.data_ptr(), nb, 0.0, Cc.data_ptr(), nb, 0, q="panel"))
    t_t = timed(lambda: ops.trsm_rlt(nb, nb, A.data_ptr(), nb, invd, B.data_ptr(), nb))
    print("nb = %4d: diagonal block %6.1f us, inverse %6.1f us, nb^3 product %5.1f us, nb x nb right-TRSM %6.1f us" % (
        nb, t_potrf, t_inv, t_g, t_t), flush=True)
nb = 512
W = torch.from_numpy(rs.rand(nb, nb)).cuda()
for m in (1024, 2048, 4096, 8192, 16384):
    S = torch.from_numpy(rs.rand(m, nb)).cuda()
    R = torch.empty((m, nb), dtype=torch.float64, device="cuda")
    t = timed(lambda: ops.gemm_nt(m, nb, nb, 1.0, S.data_ptr(), nb, W.data_ptr(), nb, 0.0, R.data_ptr(), nb, 0, q="panel"), reps=10)
    print("slice / look-ahead GEMM %5d x 512 x 512: %6.1f us (%.1f TFLOP/s)" % (m, t, 2.0 * m * nb * nb / t * 1e-6), flush=True)
import os
os._exit(0)
