"""The serial chain of a partitioned factorisation BESIDE a trailing update: per 512-column panel the chain is
  [rank-512 update of the diagonal block] -> [diagonal block 512 x 512: four fused leaves] -> [its inverse] -> [two 512^3 products]
(0.19 ms alone, 0.45-0.77 ms in the replay traces).  This script isolates it on one GPU: the chain runs on the
high-priority panel queue of a HipPanelOps (as in the engines) while the CU-masked main queue runs the trailing update of a
C4-sized step, in several shapes:
   alone            nothing on the main queue
   default          one lower-trapezoid launch m x m x 512 (what the engines issue)
   pad=<bytes>      the same with more dummy LDS per workgroup: 3 / 2 / 1 workgroups of the update per CU instead of 4
   pieces=<n>       the same tiles as n launches of column slices, back to back (the chip drains between them)
   reserve=<n>      (separate process, GPT_RESERVE_CUS) n CUs left to the panel queue instead of 32
For every shape: the chain's wall time per panel (HIP events on the panel queue, median of the panels that ran fully inside the
update) and the update's own duration (what the shape costs the main queue).
usage: chain_contention.py [m=16384] [variant ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
from gptools_amd.dist import HipPanelOps  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variants = sys.argv[2:] or ["alone", "default", "pad=8192", "pad=16384", "pad=24576", "pad=32768", "pad=49152", "pad=65536", "pieces=4", "pieces=16"]
nb = 512
ops = HipPanelOps(0)
rs = np.random.RandomState(0)
Mh = rs.rand(nb, nb)
A0 = torch.from_numpy(Mh.dot(Mh.T) + nb * np.eye(nb)).cuda()
A = A0.clone()
H = torch.from_numpy(rs.rand(nb, nb)).cuda()
W = torch.empty((nb, nb), dtype=torch.float64, device="cuda")
R = torch.empty((nb, nb), dtype=torch.float64, device="cuda")
invd = torch.empty(((nb // 128) * 9216,), dtype=torch.float64, device="cuda")
info = torch.zeros((1,), dtype=torch.int32, device="cuda")
P = torch.from_numpy(rs.rand(m, nb)).cuda()                 # the panel the update applies
C = torch.zeros((m, m), dtype=torch.float64, device="cuda")


def chain_once():
    """one panel's worth of chain work on the panel queue (the 1 x 1 corner of GridLML's step)"""
    ops.copy2d(A, A0)
    ops.gemm_nt(nb, nb, nb, -1.0, H.data_ptr(), nb, H.data_ptr(), nb, 1.0, A.data_ptr(), nb, 1, q="panel")      # A_kk -= H H^T
    ops.potrf_panel(nb, nb, A.data_ptr(), nb, invd, info, 0)
    ops.trinv(nb, A.data_ptr(), nb, invd, W.data_ptr(), nb)
    ops.gemm_nt(nb, nb, nb, -1.0, H.data_ptr(), nb, H.data_ptr(), nb, 1.0, R.data_ptr(), nb, 0, q="panel")      # head block update
    ops.gemm_nt(nb, nb, nb, 1.0, R.data_ptr(), nb, W.data_ptr(), nb, 0.0, H.data_ptr(), nb, 0, q="panel")        # H = A W^T


def update(pieces=1):
    if pieces == 1:
        ops.gemm_nt(m, m, nb, -1.0, P.data_ptr(), nb, P.data_ptr(), nb, 1.0, C.data_ptr(), m, 1, q="main")
        return
    w = (m // pieces) // 64 * 64
    c0 = 0
    while c0 < m:
        cw = min(w, m - c0)
        # columns [c0, c0 + cw): rows from c0 down (lower trapezoid of the slice)
        ops.gemm_nt(m - c0, cw, nb, -1.0, P.data_ptr() + 8 * c0 * nb, nb, P.data_ptr() + 8 * c0 * nb, nb, 1.0,
                    C.data_ptr() + 8 * (c0 * m + c0), m, 1, q="main")
        c0 += cw


def run(variant):
    pieces, pad = 1, 1024
    if variant.startswith("pad="):
        pad = int(variant[4:])
    if variant.startswith("pieces="):
        pieces = int(variant[7:])
    ops.ctx_main.set_option("lookahead", 1)                  # (the LDS pad of main-queue GEMMs applies with look-ahead on)
    ops.ctx_main.set_option("gemm_pad", pad)
    reps_u = 1 if variant == "alone" else 3
    nchain = 40
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nchain + 1)]
    u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with ops.queue("panel"):
        for _ in range(3):
            chain_once()
    if variant != "alone":
        with ops.queue("main"):
            update(pieces)                                   # (warm-up: tile-order tables, first touch)
    torch.cuda.synchronize()
    with ops.queue("main"):
        u0.record()
        if variant != "alone":
            for _ in range(reps_u):
                update(pieces)
        u1.record()
    with ops.queue("panel"):
        ev[0].record()
        for i in range(nchain):
            chain_once()
            ev[i + 1].record()
    torch.cuda.synchronize()
    t_upd = u0.elapsed_time(u1) / reps_u
    # panels that ran while the update was running: those that ended before the update did
    t0 = u0.elapsed_time(ev[0])
    per = []
    for i in range(nchain):
        end_i = u0.elapsed_time(ev[i + 1])
        if variant == "alone" or end_i < t_upd * reps_u:
            per.append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
    per = per[1:] or [float("nan")]
    print("%-14s chain per panel: median %6.1f us  min %6.1f  max %6.1f  (%2d panels inside the update)   update %8.1f us (%5.1f TFLOP/s)"
          % (variant, float(np.median(per)), min(per), max(per), len(per), t_upd * 1e3,
             (nb * (m * (m + 1.0))) / (t_upd * 1e-3) * 1e-12 if variant != "alone" else 0.0), flush=True)


print("m = %d, nb = %d, GPT_RESERVE_CUS = %s" % (m, nb, os.environ.get("GPT_RESERVE_CUS", "32 (default)")))
for v in variants:
    run(v)
os._exit(0)
