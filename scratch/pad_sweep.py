"""defer_pad on / off over the padded orders that end a factorisation differently (last panel = pad leaf alone / one real leaf + pad / two + pad,
panel widths 256 / 384 / 640, helper stream on / off): ll, alpha and L must be the same bits, alpha must match the oracle.  (test infrastructure)"""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
from oracle import oracle as O
O.build()
rs = np.random.RandomState(11)
bad = 0
for N in (1536, 3072, 4608, 5120, 5632, 6144, 6656, 7168, 7680, 9216, 12288, 12800, 13312):
    d = 2
    X = rs.rand(N, d); n = np.zeros((N, d), dtype=np.int32); n[3 * N // 4:, 0] = 1
    y = np.where(n.sum(1) > 0, np.cos(X.sum(1)), np.sin(X.sum(1))) + 0.05 * rs.randn(N); err = 0.05 * np.ones(N)
    p = np.array([1.0, 0.3, 0.3])
    ctx = _lib.Context(0); ctx.set_data(X, n); ctx.set_option("eager_alpha", 1)
    out = []
    for mode in (0, 1, 1, 0):
        ctx.set_option("defer_pad", mode)
        ll = ctx.fit(_lib.KERNEL_M52, p, 0.0, y, err, 2.2e-14)
        out.append((ll, ctx.get_alpha(N), ctx.get_L(N) if N <= 7680 else None))
    same = all(out[0][0] == o[0] and np.array_equal(out[0][1], o[1]) and (o[2] is None or np.array_equal(out[0][2], o[2])) for o in out[1:])
    msg = ""
    if N <= 5632:
        ref = O.fit("m52", p, X, n, y, err, chol="scipy")
        e1 = abs(out[1][0][0] - ref["ll_data"]) / abs(ref["ll_data"]); e2 = np.abs(out[1][1] - ref["alpha"]).max() / np.abs(ref["alpha"]).max()
        msg = "  oracle: ll %.1e alpha %.1e" % (e1, e2)
        same = same and e1 < 1e-9 and e2 < 2e-7
    NP = (N + 1 + 127) // 128 * 128
    print("N %5d (padded %5d): %s%s" % (N, NP, "same bits with and without" if same else "DIFFERENT", msg), flush=True)
    bad += not same
    ctx.close()
print("pad sweep:", "OK" if bad == 0 else "%d FAILED" % bad)
