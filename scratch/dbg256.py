import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
ctx = _lib.Context(0)
rs = np.random.RandomState(1)
for N in (256, 384, 512, 640):
    A = rs.randn(N, N); A = A.dot(A.T) + N * np.eye(N)
    Lr = np.linalg.cholesky(A)
    for nb in (256, 384):
        ctx.set_option("nb_outer", nb)
        for l256 in (0, 1):
            ctx.set_option("leaf256", l256)
            try:
                L = np.tril(ctx.potrf_host(A))
            except Exception as e:
                print(N, nb, l256, "EXC", e); continue
            E = np.abs(L - Lr)
            blocks = [[E[i:i+128, j:j+128].max() for j in range(0, i + 128, 128)] for i in range(0, N, 128)]
            print("N=%d nb=%d leaf256=%d max err %.2e  blocks: %s" % (N, nb, l256, E.max(), [["%.0e" % v for v in row] for row in blocks]))
