"""Repeatability of the leaf kernels (round 5; round 6: the explicit publication stores, the eager alpha on the tail stream): the same evaluation N times must return the same bits every time (flag timing,
pipelined publication and the strips' batch overlap may not leak into the numbers), at sizes that cover the 64-row and the 128-row
consumer workgroups, timeout 600 python scratch/repeat.py [reps]"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = _lib.Context(0)
rs = np.random.RandomState(5)
bad = 0
for (kid, N, d) in ((0, 700, 2), (1, 2047, 3), (0, 3000, 2), (1, 4096, 3), (1, 5000, 3), (1, 8192, 3)):
    X = rs.rand(N, d); n = np.zeros((N, d), int)
    if kid == 1:
        n[3 * N // 4:, 0] = 1
    y = np.sin(X.sum(1)) + 0.05 * rs.randn(N); err = 0.05 * np.ones(N); p = np.concatenate(([1.0], 0.3 * np.ones(d)))
    ctx.set_data(X, n)
    for opts in ({}, {"eager_alpha": 1}, {"fuse_rows64": 8192, "fuse_rows32": 0}, {"fuse_rows32": 8192}, {"fuse_rows16": 8192}):
        for k_, v_ in (("eager_alpha", 0), ("fuse_rows64", 2048), ("fuse_rows32", 2048), ("fuse_rows16", 0)):
            ctx.set_option(k_, opts.get(k_, v_))
        r = reps if not opts else max(reps // 6, 20)
        first = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
        L0 = np.tril(ctx.get_L(N)) if N <= 3000 else None
        diff = 0
        a0 = ctx.get_alpha(N) if opts.get("eager_alpha") else None
        for it in range(r):
            got = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
            diff += got != first
            if a0 is not None:          # (the eager alpha: with N a multiple of 512 its substitution runs beside the pad leaf)
                diff += not np.array_equal(ctx.get_alpha(N), a0)
        if L0 is not None:
            diff += not np.array_equal(np.tril(ctx.get_L(N)), L0)
        print("N %5d kid %d %-22s %4d evaluations: %s" % (N, kid, opts or "(default)", r, "all identical" if diff == 0 else "%d DIFFER" % diff), flush=True)
        bad += diff
print("repeatability:", "OK" if bad == 0 else "FAILED")
