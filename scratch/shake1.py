import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib as L
from oracle import oracle as O
ctx = L.Context(0)
rs = np.random.RandomState(1)
def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
# --- gemm
for (m, n, k) in [(64, 64, 64), (128, 192, 64), (300, 200, 100), (512, 512, 256), (1024, 1024, 128)]:
    A = rs.randn(m, k); B = rs.randn(n, k); C0 = rs.randn(m, n)
    for tile in (64, 128):
        ctx.set_option("tile", tile)
        C = ctx.gemm_nt_host(-1.0, A, B, 1.0, C0)
        print("gemm", m, n, k, tile, rel(C, C0 - A @ B.T))
ctx.set_option("tile", 0)
# --- kpairs / kbuild vs oracle
for kern, kid in (("se", 0), ("m52", 1)):
    for d in (1, 2, 3, 4):
        M = 1000
        Xi = rs.rand(M, d); Xj = rs.rand(M, d); Xj[:50] = Xi[:50]
        mo = 1 if kern == "m52" else 3
        ni = np.zeros((M, d), int); nj = np.zeros((M, d), int)
        for r in range(M):
            if rs.rand() < 0.5: ni[r, rs.randint(d)] = rs.randint(1, mo + 1)
            if rs.rand() < 0.5: nj[r, rs.randint(d)] = rs.randint(1, mo + 1)
        p = np.concatenate(([1.3], 0.2 + 0.5 * rs.rand(d)))
        a = ctx.kpairs(kid, p, Xi, Xj, ni, nj); b = O.kpairs(kern, p, Xi, Xj, ni, nj)
        print("kpairs", kern, d, rel(a, b), np.abs(a - b).max())
        if kern == "se":
            for hd in range(d + 1):
                a = ctx.kpairs(kid, p, Xi, Xj, ni, nj, hyper_deriv=hd)
                with np.errstate(all="ignore"):
                    b = O.kpairs(kern, p, Xi, Xj, ni, nj, hyper_deriv=hd)
                ok = np.isfinite(b)
                print("   hd", hd, rel(a[ok], b[ok]))
        N, P = 300, 170
        X = rs.rand(N, d); Xs = rs.rand(P, d)
        n = np.zeros((N, d), int); n[N // 2:, 0] = 1
        ns = np.zeros((P, d), int); ns[::3, d - 1] = 1
        print("kbuild sym", rel(ctx.kbuild(kid, p, X, n), O.kbuild(kern, p, X, n)),
              "rect", rel(ctx.kbuild(kid, p, X, n, Xs, ns), O.kbuild(kern, p, X, n, Xs, ns)))
# --- potrf
for N in (50, 128, 200, 256, 600, 1100):
    A = rs.randn(N, N); A = A @ A.T + N * np.eye(N)
    Lg = ctx.potrf_host(A)
    Lr = np.linalg.cholesky(A)
    print("potrf", N, rel(Lg, Lr))
# --- fit / predict
for kern, kid in (("se", 0), ("m52", 1)):
    for N, d in ((100, 1), (500, 2), (1500, 3)):
        X = rs.rand(N, d); n = np.zeros((N, d), int); n[3 * N // 4:, 0] = 1
        y = np.sin(X.sum(1)) + 0.05 * rs.randn(N)
        p = np.concatenate(([1.0], 0.3 * np.ones(d))); err = 0.05 * np.ones(N)
        ref = O.fit(kern, p, X, n, y, err, chol="scipy")
        ctx.set_data(X, n)
        for la in (0, 1):
            ctx.set_option("lookahead", la)
            ll, ld = ctx.fit(kid, p, 0.0, y, err, 1e2 * np.finfo(float).eps)
            print("fit", kern, N, d, "la", la, "ll", ll, ref["ll_data"], abs(ll - ref["ll_data"]) / abs(ref["ll_data"]),
                  "ld", abs(ld - ref["logdet_half"]) / abs(ref["logdet_half"]))
        Lg = ctx.get_L(N); al = ctx.get_alpha(N)
        print("   L", rel(Lg, ref["L"]), "alpha", rel(al, ref["alpha"]))
        M = 70
        Xs = rs.rand(M, d); ns = np.zeros((M, d), int); ns[M // 2:, 0] = 1
        mr, sr, cr = O.predict(kern, p, X, n, ref["L"], ref["alpha"], Xs, ns)
        m2, s2, c2 = ctx.predict(Xs, ns, 2)
        m1, s1, _ = ctx.predict(Xs, ns, 1)
        print("   predict mean", rel(m2, mr), "cov", rel(c2, cr), "std", np.abs(s1 - sr).max(), np.abs(s2 - sr).max())
print("done")
