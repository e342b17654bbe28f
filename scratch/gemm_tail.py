"""Where does the time of one trailing-update launch go?  Time (5 back-to-back launches, so launch latency is hidden) of the
lower-trapezoid update at k = 384 for tile counts just below / above multiples of the 896 workgroup slots of the masked
main stream, and the per-tile time from a launch that fills exactly one wave."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
lib = _lib.load()
ctx = _lib.Context(0)
st = torch.cuda.ExternalStream(int(ctx.stream))
ctx.set_option("lookahead", 1)             # trailing updates with the 1 KiB LDS pad (4 workgroups per CU)
k = 384
def run(m, n, tri, reps=6):
    with torch.cuda.stream(st):
        A = torch.randn(m, k, dtype=torch.float64, device="cuda")
        C = torch.randn(m, n, dtype=torch.float64, device="cuda")
        for _ in range(2):
            _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, n, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), n, tri))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _lib.check(lib.gpt_dev_gemm_nt(ctx.handle, m, n, k, -1.0, A.data_ptr(), k, A.data_ptr(), k, 1.0, C.data_ptr(), n, tri))
        e1.record()
    st.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
# rectangular launches with an exact number of tiles: n = 64 * c columns, m = 64 * r rows -> r * c tiles
for tiles_target in (224, 448, 896, 897, 1120, 1344, 1792, 1793, 2688, 2689, 3584, 4480, 5376, 6272, 7168, 7169):
    # factor into r x c with c <= r
    best = None
    for c in range(1, 200):
        if tiles_target % c == 0:
            r = tiles_target // c
            if r >= c and r <= 400: best = (r, c)
    if best is None:
        # prime-ish: use a single column strip
        best = (tiles_target, 1)
    r, c = best
    t = run(64 * r, 64 * c, 0)
    fl = 2.0 * 64 * r * 64 * c * k
    print("%5d tiles (%3d x %3d) = %5.2f waves of 896: %7.1f us  %5.1f TFLOP/s  %.1f us per wave" % (r * c, r, c, r * c / 896.0, t, fl / t * 1e-6, t / np.ceil(r * c / 896.0)))
